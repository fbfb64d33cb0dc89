// What limits the MFMA issue rate of the wide MLP machinery (csrc/nrh_mlp32.h)?  One wave per SIMD, 4 waves per workgroup,
// one workgroup per CU; every variant runs ITERS x 2 windows of 48 v_mfma_f32_32x32x16_f16 and reports shader cycles per MFMA
// (s_memtime, clock-independent) and the wall-clock rate.  Variants: profiles/ubench/gen_kloop_ubench.py.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form \
//         -I nrhints_amd/csrc -I profiles/ubench profiles/ubench/kloop_ubench.hip -o profiles/ubench/data/kloop_ubench
#include "nrh_mlp32.h"

#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define W32_QSTORE(c, half, val) __builtin_nontemporal_store((val), reinterpret_cast<nrh32::u32x4*>(scr + ((c) * 2 + (half)) * 1024 + lane16))

using namespace nrh32;

// DMA: 0 none, 1 eight pieces per wave per window + chunk barrier (the production pattern), 2 barrier only
template <int V, int DMA>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void kb(unsigned long long* out, float* sink, const char* w,
                                                                                           char* scratch, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t lane16 = lane * 16;
  char* const scr = scratch + (size_t)(blockIdx.x * WAVES + wave) * 16384;
  // LDS: two ring slots of small fp16 numbers; AGPRs: the same
  for (int i = threadIdx.x; i < 3 * SLOT_BYTES / 4; i += THREADS) reinterpret_cast<uint32_t*>(smem)[i] = 0x2c002e00u + (i & 0xff);
  {
    const uint32_t v = 0x30003400u + lane;
#include "gen/fill_agpr.inc"
  }
  __syncthreads();
  const uint32_t ring_lds = lds_off(smem);
  const uint32_t wlane = ring_lds + lane16;
  int n = 0;
  f32x16 h0, c0, d0, h1, c1, d1;
  for (int r = 0; r < 16; ++r) { h0[r] = 0.01f * r; c0[r] = 0.f; d0[r] = 0.f; h1[r] = -0.02f * r; c1[r] = 0.f; d1[r] = 0.f; }
#define UB_WADDR() (wlane + (n % 3) * SLOT_BYTES)
  // DMA 1: the production pattern (chunk barrier, 8 pieces of block n + 2 spread over MFMA slots by W32_DMA); 2: barrier only;
  // 3: barrier + all 8 pieces in a burst at the top of the window (what the first version of the kernel did)
  const char* fg0 = uni(w + wave * 8192); const char* fg1 = fg0 + 4096;
  uint32_t fm0 = 0, fm1 = 0;
#define W32_DMA(i) do { if (DMA == 1) dma_piece<((i) & 3) * 1024>(((i) < 4) ? fg0 : fg1, ((i) < 4) ? fm0 : fm1, lane16); } while (0)
#define UB_TOP() do { if (DMA) { chunk_sync<8>(); fm0 = uni(ring_lds + ((n + 2) % 3) * SLOT_BYTES + wave * 8192); fm1 = fm0 + 4096; \
    if (DMA == 3) dma_block(fg0, fm0, lane16); } ++n; } while (0)
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if constexpr (V == 0) {
#include "gen/bare2.inc"
    } else if constexpr (V == 1) {
#include "gen/bare3.inc"
    } else if constexpr (V == 2) {
#include "gen/nolds2.inc"
    } else if constexpr (V == 3) {
#include "gen/nolds3.inc"
    } else if constexpr (V == 4) {
#include "gen/pf3_2.inc"
    } else if constexpr (V == 5) {
#include "gen/epi2.inc"
    } else if constexpr (V == 6) {
#include "gen/epi3.inc"
    } else if constexpr (V == 7) {
#include "gen/epi2d.inc"
    } else if constexpr (V == 8) {
#include "gen/epi2_nv6.inc"
    } else if constexpr (V == 9) {
#include "gen/bare2_dma.inc"
    } else if constexpr (V == 10) {
#include "gen/epi2_dma.inc"
    } else {
#include "gen/epi2d_dma.inc"
    }
  }
  asm volatile("s_nop 7\n\ts_nop 7" : "+v"(h0), "+v"(c0), "+v"(h1), "+v"(c1));
  const unsigned long long t1 = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) out[blockIdx.x * WAVES + wave] = t1 - t0;
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += h0[r] + c0[r] + h1[r] + c1[r] + d0[r] + d1[r];
  if (s == 123.456f) sink[0] = s;
}

template <int V, int DMA>
static void run(const char* name, unsigned long long* d_out, float* d_sink, const char* d_w, char* d_scr, int iters) {
  CK(hipFuncSetAttribute((const void*)kb<V, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  const int grid = 256;
  hipLaunchKernelGGL((kb<V, DMA>), dim3(grid), dim3(THREADS), LDS_BYTES, 0, d_out, d_sink, d_w, d_scr, 4);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((kb<V, DMA>), dim3(grid), dim3(THREADS), LDS_BYTES, 0, d_out, d_sink, d_w, d_scr, iters);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(grid * WAVES);
  CK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
  double avg = 0, mx = 0;
  for (auto v : h) { avg += (double)v; if ((double)v > mx) mx = (double)v; }
  avg /= h.size();
  const double mf = 96.0 * iters;
  printf("%-14s dma %d: %7.2f cycles/MFMA (max wave %7.2f)  %8.3f ms  -> %6.0f MHz effective, %7.1f TFLOP/s MFMA rate\n", name, DMA, avg / mf, mx / mf, ms,
         avg / (ms * 1e3), mf * grid * WAVES * 32768.0 / (ms * 1e9));
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  unsigned long long* d_out;
  float* d_sink;
  char *d_w, *d_scr;
  CK(hipMalloc(&d_out, 1024 * 8));
  CK(hipMalloc(&d_sink, 4));
  CK(hipMalloc(&d_w, 4 << 20));
  CK(hipMemset(d_w, 0x2c, 4 << 20));
  CK(hipMalloc(&d_scr, (size_t)1024 * 16384));
  run<2, 0>("nolds2", d_out, d_sink, d_w, d_scr, iters);
  run<0, 0>("bare2", d_out, d_sink, d_w, d_scr, iters);
  run<0, 2>("bare2", d_out, d_sink, d_w, d_scr, iters);
  run<0, 3>("bare2 burst", d_out, d_sink, d_w, d_scr, iters);
  run<9, 1>("bare2 spread", d_out, d_sink, d_w, d_scr, iters);
  run<5, 0>("epi2", d_out, d_sink, d_w, d_scr, iters);
  run<7, 0>("epi2d", d_out, d_sink, d_w, d_scr, iters);
  run<5, 3>("epi2 burst", d_out, d_sink, d_w, d_scr, iters);
  run<10, 1>("epi2 spread", d_out, d_sink, d_w, d_scr, iters);
  run<11, 1>("epi2d spread", d_out, d_sink, d_w, d_scr, iters);
  return 0;
}
