// Micro-benchmark 2 (measurement aid): can VALU work hide behind MFMA on one SIMD of gfx950?
//   A<NV,SHAPE>: ONE wave per SIMD, per iteration 4 MFMAs each followed by NV independent VALU ops (compile-time).
//   B: two waves per SIMD, one pure-MFMA, one pure-VALU (compile-time roles by wave id).
// Reports ns/iter and cycles/iter from s_memtime deltas (shader clock) so DVFS does not blur the comparison.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VOP(x) x = __builtin_fmaf(x, 1.0001f, 0.5f)

template <int NV, int SHAPE>
__global__ __launch_bounds__(256, 1) void kA(float* out, long long* clk, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  f32x16 d0 = {}, d1 = {}, d2 = {}, d3 = {};
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (SHAPE == 16) {
        if (m == 0) c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        if (m == 1) c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
        if (m == 2) c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
        if (m == 3) c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
      } else {
        if (m == 0) d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d0, 0, 0, 0);
        if (m == 1) d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d1, 0, 0, 0);
        if (m == 2) d2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d2, 0, 0, 0);
        if (m == 3) d3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d3, 0, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < NV; ++j) VOP(v[(m * NV + j) & 7]);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = c0[0] + c1[1] + c2[2] + c3[3] + d0[0] + d1[1] + d2[2] + d3[3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int MODE>  // 0: 8 waves all MFMA; 1: 8 waves all VALU; 2: waves 0-3 MFMA, 4-7 VALU; 3: 4 waves MFMA; 4: 4 waves VALU
__global__ __launch_bounds__(512, 1) void kB(float* out, long long* clk, int iters) {
  const int wave = threadIdx.x >> 6;
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
  const bool do_mfma = (MODE == 0) || (MODE == 2 && wave < 4) || (MODE == 3 && wave < 4);
  const bool do_valu = (MODE == 1) || (MODE == 2 && wave >= 4) || (MODE == 4 && wave < 4);
  long long t0 = __builtin_readcyclecounter();
  if (do_mfma) {
    for (int it = 0; it < iters; ++it) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
    }
  } else if (do_valu) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) VOP(v[j & 7]);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = c0[0] + c1[1] + c2[2] + c3[3];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
  if (threadIdx.x == 256 && blockIdx.x == 0) clk[1] = t1 - t0;
}

template <typename K>
static void run(const char* name, K kern, int threads, int iters) {
  float* o; long long* c; hipMalloc(&o, 256 * 512 * sizeof(float)); hipMalloc(&c, 16);
  hipMemset(c, 0, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, o, c, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, o, c, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long hc[2]; hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
  printf("%-44s %8.3f ms  %7.2f ns/iter  clk(w0)=%6.1f clk(w4)=%6.1f per iter\n", name, ms, ms * 1e6 / iters,
         (double)hc[0] / iters, (double)hc[1] / iters);
  hipFree(o); hipFree(c);
}
int main() {
  const int it = 100000;
  run("A 16x16x32: 4 mfma + 0 valu  (1 wave/SIMD)", kA<0, 16>, 256, it);
  run("A 16x16x32: 4 mfma + 4x1 valu", kA<1, 16>, 256, it);
  run("A 16x16x32: 4 mfma + 4x2 valu", kA<2, 16>, 256, it);
  run("A 16x16x32: 4 mfma + 4x3 valu", kA<3, 16>, 256, it);
  run("A 16x16x32: 4 mfma + 4x4 valu", kA<4, 16>, 256, it);
  run("A 16x16x32: 4 mfma + 4x6 valu", kA<6, 16>, 256, it);
  run("A 32x32x16: 4 mfma + 0 valu", kA<0, 32>, 256, it);
  run("A 32x32x16: 4 mfma + 4x2 valu", kA<2, 32>, 256, it);
  run("A 32x32x16: 4 mfma + 4x4 valu", kA<4, 32>, 256, it);
  run("A 32x32x16: 4 mfma + 4x6 valu", kA<6, 32>, 256, it);
  run("A 32x32x16: 4 mfma + 4x8 valu", kA<8, 32>, 256, it);
  run("B 8 waves all MFMA (4/iter)", kB<0>, 512, it);
  run("B 8 waves all VALU (16 fma/iter)", kB<1>, 512, it);
  run("B 4 waves MFMA only", kB<3>, 512, it);
  run("B 4 waves VALU only", kB<4>, 512, it);
  run("B 4 MFMA waves + 4 VALU waves (same SIMDs)", kB<2>, 512, it);
  return 0;
}
