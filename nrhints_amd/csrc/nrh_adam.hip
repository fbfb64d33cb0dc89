// Adam over a fixed list of parameter tensors in ONE launch (trainer/trainer.py:99-102 builds torch.optim.Adam over two parameter
// groups; torch's capturable implementation issues ~100 small kernels per step for the 46 tensors of the renderer - 0.5 ms of a
// 7 ms training step).  Arithmetic of torch.optim.Adam as the reference runs it (default, non-capturable branch of
// torch.optim.adam._multi_tensor_adam; no amsgrad / weight decay / maximize): the bias corrections are host-precision scalars
// there (python floats), the tensor operations float32 in this order:
//   step += 1;  m += (g - m) (1 - b1);  v = v b2 + (1 - b2) g g
//   bc1 = 1 - b1^step;  bc2 = 1 - b2^step   (double);   step_size = lr / bc1
//   p += -step_size * (m / (sqrt(v) / sqrt(bc2) + eps))
// (torch's CAPTURABLE branch divides by the learning rate - sqrt(v) / (sqrt(bc2) * -step_size) - which is 0 / -0 = NaN for an
// entry with zero gradient history under lr = 0, the first step of the reference's warm-up schedule; not reproduced here.)
// One block = one chunk of 2048 elements of one tensor (chunk table built by the host once per parameter list); the step
// counters are bumped by a second, one-block launch after every chunk has read them.
#include "nrh_common.h"

namespace nrhadam {

constexpr int CHUNK = 2048;
constexpr int MAX_GROUPS = 4;

struct Tensor {          // mirrors NrhAdamTensor (include/nrhints_hip.h)
  float* p;
  const float* g;
  float* m;
  float* v;
  float* step;           // device scalar, float32 (torch's capturable state layout)
  long long n;
  int group;
  int pad;
};

struct Args {
  const Tensor* tensors;
  const int* chunks;      // [nchunks][2]: tensor index, element offset / CHUNK
  const float* lr_ptr[MAX_GROUPS];   // device learning rate per group (overrides lr) or null
  double b1d[MAX_GROUPS], b2d[MAX_GROUPS], lrd[MAX_GROUPS];
  float b2[MAX_GROUPS], w1[MAX_GROUPS], w2[MAX_GROUPS], eps[MAX_GROUPS];
  int ntensors;
};

__global__ __launch_bounds__(256) void adam_kernel(const Args a) {
  const int ti = a.chunks[2 * blockIdx.x], off = a.chunks[2 * blockIdx.x + 1];
  const Tensor t = a.tensors[ti];
  const int gi = t.group;
  const double lr = a.lr_ptr[gi] ? (double)a.lr_ptr[gi][0] : a.lrd[gi];
  const float b2 = a.b2[gi], eps = a.eps[gi];
  const double step = (double)t.step[0] + 1.0;
  const double bc1 = 1.0 - pow(a.b1d[gi], step), bc2 = 1.0 - pow(a.b2d[gi], step);
  const float neg_step_size = -(float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2);
  const float w1 = a.w1[gi], w2 = a.w2[gi];
  const long long base = (long long)off * CHUNK;
#pragma unroll
  for (int k = 0; k < CHUNK / 256; ++k) {
    const long long i = base + k * 256 + threadIdx.x;
    if (i < t.n) {
      const float g = t.g[i];
      const float m = t.m[i] + w1 * (g - t.m[i]);                  // lerp_(g, 1 - b1), weight < 0.5 form
      const float v = __fmul_rn(t.v[i], b2) + __fmul_rn(__fmul_rn(w2, g), g);   // mul_(b2).addcmul_(g, g, value = 1 - b2)
      t.m[i] = m;
      t.v[i] = v;
      t.p[i] = t.p[i] + neg_step_size * (m / (sqrtf(v) / bc2_sqrt + eps));     // addcdiv_(m, denom, value = -step_size)
    }
  }
}

__global__ void adam_bump_kernel(const Tensor* tensors, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) tensors[i].step[0] += 1.0f;
}

}  // namespace nrhadam
