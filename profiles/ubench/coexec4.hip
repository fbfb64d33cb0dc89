// Micro-benchmark 4 (measurement aid): SAME-wave interleave - per iteration 4 independent MFMAs, each followed by NV
// independent v_fma (or v_exp).  WPS = waves per SIMD (1: 256-thread blocks, 2: 512-thread blocks), one block per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define VOP(x) x = __builtin_fmaf(x, 1.0001f, 0.5f)
#define TOP(x) x = __builtin_amdgcn_exp2f(x)

template <int NV, int SHAPE, int TRANS, int THREADS>
__global__ __launch_bounds__(THREADS, 2) void kA(float* out, long long* clk, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  f32x16 d0 = {}, d1 = {};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (SHAPE == 0) {
        if (m == 0) c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        if (m == 1) c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
        if (m == 2) c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
        if (m == 3) c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
      } else {
        if (m & 1) d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d1, 0, 0, 0);
        else d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d0, 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < NV; ++j) { if (TRANS) TOP(v[(m * NV + j) & 7]); else VOP(v[(m * NV + j) & 7]); }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[1];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
template <typename K>
static void run(const char* name, K kern, int threads, int iters) {
  float* o; long long* c; hipMalloc(&o, 256 * 512 * sizeof(float)); hipMalloc(&c, 16);
  hipMemset(c, 0, 16);
  hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, o, c, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, o, c, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long hc[2]; hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
  printf("%-50s %8.3f ms   %6.1f clk/iter (4 mfma)\n", name, ms, (double)hc[0] / iters);
  hipFree(o); hipFree(c);
}
#define ROW(S, T, TH, label) \
  run(label " nv=0", kA<0, S, T, TH>, TH, it); run(label " nv=1", kA<1, S, T, TH>, TH, it); run(label " nv=2", kA<2, S, T, TH>, TH, it); \
  run(label " nv=3", kA<3, S, T, TH>, TH, it); run(label " nv=4", kA<4, S, T, TH>, TH, it); run(label " nv=6", kA<6, S, T, TH>, TH, it); \
  run(label " nv=8", kA<8, S, T, TH>, TH, it);
int main() {
  const int it = 100000;
  ROW(0, 0, 256, "16x16x32 1w/SIMD fma")
  ROW(0, 0, 512, "16x16x32 2w/SIMD fma")
  ROW(1, 0, 256, "32x32x16 1w/SIMD fma")
  ROW(1, 0, 512, "32x32x16 2w/SIMD fma")
  ROW(0, 1, 512, "16x16x32 2w/SIMD exp")
  ROW(1, 1, 512, "32x32x16 2w/SIMD exp")
  return 0;
}
