// Weight-norm fold of all linears in one launch, and its adjoint in one launch.
//   W[r][:] = v[r][:] * g[r] / ||v[r][:]||_2        (old-style nn.utils.weight_norm, dim = 0;
//                                                    reference: fields/sdf_field.py:81-82, fields/reflectance_network.py:61-62)
//   adjoint:  gbar[r] = <Wbar[r], v[r]> / ||v[r]||;   vbar[r] = (g[r]/||v[r]||) (Wbar[r] - <Wbar[r], v[r]> v[r] / ||v[r]||^2)
// A training step folds the 15 linears and differentiates the fold: ~200 small PyTorch kernels per step otherwise.
// One wavefront per matrix row (rows of all layers are enumerated consecutively).
#include "nrh_common.h"

namespace nrh {

constexpr int FOLD_MAX_LAYERS = 16;
constexpr int FOLD_MAX_COLS = 384;   // 6 elements per lane

struct FoldArgs {
  const float* v[FOLD_MAX_LAYERS];
  const float* g[FOLD_MAX_LAYERS];
  float* w[FOLD_MAX_LAYERS];          // forward out
  const float* wbar[FOLD_MAX_LAYERS]; // adjoint in (null: layer skipped, zero gradients)
  float* vbar[FOLD_MAX_LAYERS];       // adjoint out
  float* gbar[FOLD_MAX_LAYERS];       // adjoint out
  int cols[FOLD_MAX_LAYERS];
  int row_start[FOLD_MAX_LAYERS + 1];
  int nlayers;
};

template <bool ADJOINT>
__global__ __launch_bounds__(256) void fold_kernel(const FoldArgs a) {
  const int lane = threadIdx.x & 63;
  const int row_g = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row_g >= a.row_start[a.nlayers]) return;
  int l = 0;
#pragma unroll 1
  while (l + 1 < a.nlayers && row_g >= a.row_start[l + 1]) ++l;
  const int r = row_g - a.row_start[l];
  const int C = a.cols[l];
  const float* v = a.v[l] + (size_t)r * C;
  float x[6], ss = 0.0f;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int c = lane + 64 * i;
    x[i] = (c < C) ? v[c] : 0.0f;
    ss += x[i] * x[i];
  }
  const float nrm = sqrtf(wave_sum(ss));
  const float g = a.g[l][r];
  if (!ADJOINT) {
    const float s = g / nrm;
    float* w = a.w[l] + (size_t)r * C;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int c = lane + 64 * i;
      if (c < C) w[c] = x[i] * s;
    }
    return;
  }
  float* vb = a.vbar[l] + (size_t)r * C;
  if (a.wbar[l] == nullptr) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int c = lane + 64 * i;
      if (c < C) vb[c] = 0.0f;
    }
    if (lane == 0) a.gbar[l][r] = 0.0f;
    return;
  }
  const float* wb = a.wbar[l] + (size_t)r * C;
  float y[6], dot = 0.0f;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int c = lane + 64 * i;
    y[i] = (c < C) ? wb[c] : 0.0f;
    dot += y[i] * x[i];
  }
  dot = wave_sum(dot);
  const float inv = 1.0f / nrm;
  const float s = g * inv, k = dot * inv * inv;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int c = lane + 64 * i;
    if (c < C) vb[c] = s * (y[i] - k * x[i]);
  }
  if (lane == 0) a.gbar[l][r] = dot * inv;
}

}  // namespace nrh
