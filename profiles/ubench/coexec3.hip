// Micro-benchmark 3 (measurement aid): MFMA wave + VALU wave sharing a SIMD, per MFMA shape.
// 512-thread blocks, one per CU: waves 0-3 run 4 independent MFMAs per iteration, waves 4-7 run NV v_fma per iteration.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define VOP(x) x = __builtin_fmaf(x, 1.0001f, 0.5f)
#define TOP(x) x = __builtin_amdgcn_exp2f(x)

// SHAPE 0: 16x16x32 f16 (4 pass), 1: 32x32x16 f16 (8 pass), 2: 16x16x4 f32 (8 pass), 3: 32x32x2 f32 (16 pass)
template <int SHAPE, int MODE, int TRANS>  // MODE 0 both, 1 mfma waves only, 2 valu waves only
__global__ __launch_bounds__(512, 2) void kB(float* out, long long* clk, int iters) {
  const int wave = threadIdx.x >> 6;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  const float fa = threadIdx.x * 0.01f, fb = 1.5f;
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  f32x16 d0 = {}, d1 = {};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
  long long t0 = __builtin_readcyclecounter();
  if (wave < 4 && MODE != 2) {
    for (int it = 0; it < iters; ++it) {
      if (SHAPE == 0) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
      } else if (SHAPE == 1) {
        d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d1, 0, 0, 0);
      } else if (SHAPE == 2) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, c3, 0, 0, 0);
      } else {
        d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, d1, 0, 0, 0);
      }
    }
  } else if (wave >= 4 && MODE != 1) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) { if (TRANS) TOP(v[j & 7]); else VOP(v[j & 7]); }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[1];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
  if (threadIdx.x == 256 && blockIdx.x == 0) clk[1] = t1 - t0;
}
template <typename K>
static void run(const char* name, K kern, int iters) {
  float* o; long long* c; hipMalloc(&o, 256 * 512 * sizeof(float)); hipMalloc(&c, 16);
  hipMemset(c, 0, 16);
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, o, c, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, o, c, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long hc[2]; hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
  printf("%-52s %8.3f ms  mfma-wave %6.1f clk/iter   valu-wave %6.1f clk/iter\n", name, ms, (double)hc[0] / iters, (double)hc[1] / iters);
  hipFree(o); hipFree(c);
}
#define TRI(S, name) \
  run(name " mfma alone", kB<S, 1, 0>, it); run(name " + 16 v_fma wave", kB<S, 0, 0>, it); run(name " + 16 v_exp wave", kB<S, 0, 1>, it);
int main() {
  const int it = 100000;
  run("16 v_fma alone", kB<0, 2, 0>, it);
  run("16 v_exp alone", kB<0, 2, 1>, it);
  TRI(0, "16x16x32 f16 x4")
  TRI(1, "32x32x16 f16 x4")
  TRI(2, "16x16x4  f32 x4")
  TRI(3, "32x32x2  f32 x4")
  return 0;
}
