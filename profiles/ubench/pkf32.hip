// Micro-benchmark (measurement aid, not product): does v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 issue at the rate of the scalar
// forms on gfx950, alone and between 32x32x16 MFMAs of the same wave (one wave per SIMD, as the wide kernels run)?
//   mode 0: N scalar v_fma_f32 on 16 independent registers      mode 1: N/2 v_pk_fma_f32 on the same 16 registers (8 pairs)
//   mode 2: per MFMA  K scalar fma                                mode 3: per MFMA  K/2 packed fma            (K = 8, 16)
// Cycles from s_memtime around the loop, wave 0 of every block; 256 blocks x 256 threads.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define FMA1(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m), "v"(c))
#define PK(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(m2), "v"(c2))

template <int MODE, int K>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  f32x16 acc = {}, acc2 = {};
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i;
  f32x2 p[8];
  for (int i = 0; i < 8; ++i) p[i] = f32x2{v[2 * i], v[2 * i + 1]};
  const float m = 1.0001f, c = 0.5f;
  const f32x2 m2 = {m, m}, c2 = {c, c};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) FMA1(v[i]);
    } else if (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) PK(p[i]);
    } else {
      // four MFMAs per iteration on three accumulators in the wide kernels' pattern (hh, hh, cc, ...), K VALU lanes-ops after each
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (r % 3 == 2) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc2, 0, 0, 0);
        else acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        if (MODE == 2) {
#pragma unroll
          for (int i = 0; i < K; ++i) FMA1(v[(4 * r + i) % 16]);
        } else {
#pragma unroll
          for (int i = 0; i < K / 2; ++i) PK(p[(2 * r + i) % 8]);
        }
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += v[i] + acc[i] + acc2[i];
  for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int K>
static void run(const char* name, int iters, int flops_per_iter) {
  float* o; unsigned long long* c;
  hipMalloc(&o, 256 * 256 * sizeof(float)); hipMalloc(&c, 256 * sizeof(unsigned long long));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, K>), dim3(256), dim3(256), 0, 0, o, c, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, K>), dim3(256), dim3(256), 0, 0, o, c, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256]; hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < 256; ++i) avg += (double)h[i]; avg /= 256;
  printf("%-44s %8.3f ms   %8.2f counter ticks per iteration\n", name, ms, avg / iters);
  hipFree(o); hipFree(c);
}

int main() {
  const int it = 200000;
  run<0, 0>("64 x v_fma_f32", it, 64);
  run<1, 0>("32 x v_pk_fma_f32 (same 64 lane-ops)", it, 64);
  run<2, 0>("4 MFMA 32x32x16", it, 0);
  run<2, 2>("4 x (MFMA + 2 v_fma_f32)", it, 8);
  run<3, 2>("4 x (MFMA + 1 v_pk_fma_f32)", it, 8);
  run<2, 4>("4 x (MFMA + 4 v_fma_f32)", it, 16);
  run<3, 4>("4 x (MFMA + 2 v_pk_fma_f32)", it, 16);
  run<2, 8>("4 x (MFMA + 8 v_fma_f32)", it, 32);
  run<3, 8>("4 x (MFMA + 4 v_pk_fma_f32)", it, 32);
  run<2, 12>("4 x (MFMA + 12 v_fma_f32)", it, 48);
  run<3, 12>("4 x (MFMA + 6 v_pk_fma_f32)", it, 48);
  return 0;
}
