// What does gfx950 SUSTAIN in v_mfma_f32_32x32x16_f16 under its package power limit?  (The roofline's 2.5 PFLOP/s is the boost-clock
// number; the wide f16x3 kernels run at 1.8-2.0 GHz and 1.3-1.4 kW, profiles/r06/launch_length_probe.log.)
// One wave per SIMD (1 024 waves, like the wide kernels), operands in registers, no LDS / memory traffic inside the loop:
//   mode 0  operands all zero            (the multipliers do not toggle)
//   mode 1  random fp16 operands, fixed  (the same A / B every iteration: operand buses quiet, arrays busy)
//   mode 2  random fp16 operands, 8 different A fragments rotating (closer to a K loop's operand traffic)
//   mode 3  v_mfma_f32_32x32x16_bf16 (nrh_dw_gemm's instruction) on bf16 values, 8 A fragments rotating
// Prints TFLOP/s over ~0.4 s per mode (long enough for the power controller to settle) and the cycle count per MFMA from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 profiles/ubench/mfma_sustained.hip -o profiles/ubench/bin/mfma_sustained
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NA, bool BF>
__global__ __launch_bounds__(256) void mfma_loop(const f16x8* __restrict__ ops, float* __restrict__ out, int iters, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63;
  f16x8 a[NA], b[2];
#pragma unroll
  for (int i = 0; i < NA; ++i) a[i] = ops[i * 64 + lane];
  b[0] = ops[8 * 64 + lane]; b[1] = ops[9 * 64 + lane];
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if constexpr (BF) {       // the same bits read as bf16: v_mfma_f32_32x32x16_bf16 (the weight-gradient kernel's instruction)
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(k) % NA]), __builtin_bit_cast(bf16x8, b[0]), c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(k + 1) % NA]), __builtin_bit_cast(bf16x8, b[1]), c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(k + 2) % NA]), __builtin_bit_cast(bf16x8, b[0]), c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(k + 3) % NA]), __builtin_bit_cast(bf16x8, b[1]), c3, 0, 0, 0);
      } else {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(k) % NA], b[0], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(k + 1) % NA], b[1], c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(k + 2) % NA], b[0], c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(k + 3) % NA], b[1], c3, 0, 0, 0);
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 0.4;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  std::vector<_Float16> h(10 * 64 * 8);
  f16x8* d_ops; float* d_out; unsigned long long* d_cyc;
  CK(hipMalloc(&d_ops, h.size() * 2)); CK(hipMalloc(&d_out, cus * 256 * 4)); CK(hipMalloc(&d_cyc, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("device %s, %d CUs, one wave per SIMD, v_mfma_f32_32x32x16_f16 (32 768 FLOP each), register operands\n", prop.gcnArchName, cus);
  const char* names[4] = {"zero operands", "random operands, one A fragment", "random operands, 8 A fragments rotating",
                          "bf16: random bit patterns, 8 A fragments rotating"};
  for (int rep = 0; rep < 2; ++rep)
  for (int mode = 0; mode < 4; ++mode) {
    srand(1);
    for (auto& x : h) {
      const float v = ((rand() & 0xffff) / 32768.0f - 1.0f) * 0.5f;
      if (mode == 3) { const __bf16 bv = (__bf16)v; x = __builtin_bit_cast(_Float16, bv); }      // 16 bits of a bf16 value
      else x = mode == 0 ? (_Float16)0.f : (_Float16)v;
    }
    CK(hipMemcpy(d_ops, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    int iters = 20000;
    for (int pass = 0; pass < 2; ++pass) {     // pass 0 calibrates the iteration count, pass 1 is the measurement
      CK(hipEventRecord(e0, 0));
      if (mode == 3) hipLaunchKernelGGL((mfma_loop<8, true>), dim3(cus), dim3(256), 0, 0, d_ops, d_out, iters, d_cyc);
      else if (mode == 2) hipLaunchKernelGGL((mfma_loop<8, false>), dim3(cus), dim3(256), 0, 0, d_ops, d_out, iters, d_cyc);
      else hipLaunchKernelGGL((mfma_loop<1, false>), dim3(cus), dim3(256), 0, 0, d_ops, d_out, iters, d_cyc);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long cyc; CK(hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost));
      const double n = (double)iters * 32;                      // MFMAs per wave
      if (pass == 1)
        printf("  %-42s %8.1f ms  %7.1f TFLOP/s  %5.2f cycles per MFMA  -> %.3f GHz average\n", names[mode], ms,
               n * 32768.0 * cus * 4 / (ms * 1e-3) / 1e12, (double)cyc / n, (double)cyc / (ms * 1e-3) / 1e9);
      iters = (int)(iters * secs * 1e3 / ms);
    }
  }
  return 0;
}
