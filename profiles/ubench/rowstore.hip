// Micro-benchmark (measurement aid, not product): which HBM rate do the access patterns of the training sweeps' saved arrays
// reach?  One array = [L = 8 layers][npts][256] float32 (1.07 GB at 131 072 points), written / read once by 1 024 waves
// (256 workgroups x 4 waves), every lane moving 16 bytes per instruction as the MLP kernels' epilogues do.
//   pattern 0  "16-point row-major"   lane (j = lane & 15, q = lane >> 4), block b: row j, bytes 64 b + 16 q   (nrh_mlp.h kernels)
//   pattern 1  "32-point row-major"   lane (j = lane & 31, hf = lane >> 5), chunk c, group g: row j, floats 32 c + 8 g + 4 hf
//   pattern 2  "tile-native"          [tile of 32 points][c][g][lane] x 16 B: 1 KiB contiguous per wave instruction
//   pattern 3  "32-point row-major, 32-byte runs"  as 1 after a permlane32 swap: lane owns floats 32 c + 16 (g >> 1) + 8 hf + 4 (g & 1)
// op 0: non-temporal stores, 1: plain stores, 2: non-temporal loads, 3: plain loads, 4: read two arrays + write one (nt), 5: copy (nt)
// Build: hipcc -O3 --offload-arch=gfx950 rowstore.hip -o rowstore;  run: ./rowstore [npts]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ size_t addr(int pat, int lane, long long tile32, int c, int g, long long npts, int layer) {
  // float index of this lane's 16 bytes of (32-point tile, chunk c in 0..7, group g in 0..3) in layer `layer`
  const size_t base = (size_t)layer * (size_t)npts * 256;
  if (pat == 0) {   // two 16-point tiles; (c, g) enumerate the 32 (tile half, block) pairs: half = c >> 2, block = 4 (c & 3) + g
    const int j = lane & 15, q = lane >> 4, half = c >> 2, b = 4 * (c & 3) + g;
    return base + (size_t)(tile32 * 32 + 16 * half + j) * 256 + 16 * b + 4 * q;
  } else if (pat == 1) {
    const int j = lane & 31, hf = lane >> 5;
    return base + (size_t)(tile32 * 32 + j) * 256 + 32 * c + 8 * g + 4 * hf;
  } else if (pat == 2) {
    return base + (size_t)tile32 * 8192 + (size_t)((c * 4 + g) * 64 + lane) * 4;
  } else {
    const int j = lane & 31, hf = lane >> 5;
    return base + (size_t)(tile32 * 32 + j) * 256 + 32 * c + 16 * (g >> 1) + 8 * hf + 4 * (g & 1);
  }
}

template <int PAT, int OP>
__global__ __launch_bounds__(256) void k(float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, long long npts,
                                         float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long ntiles = npts / 32;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long long t = (long long)blockIdx.x * 4 + wave; t < ntiles; t += (long long)gridDim.x * 4) {
    for (int layer = 0; layer < 8; ++layer) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const size_t a = addr(PAT, lane, t, c, g, npts, layer);
          if (OP == 0) __builtin_nontemporal_store(f32x4{(float)lane, (float)c, (float)g, (float)layer}, reinterpret_cast<f32x4*>(x + a));
          else if (OP == 1) *reinterpret_cast<f32x4*>(x + a) = f32x4{(float)lane, (float)c, (float)g, (float)layer};
          else if (OP == 2) acc += __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(y + a));
          else if (OP == 3) acc += *reinterpret_cast<const f32x4*>(y + a);
          else if (OP == 4) {
            const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(y + a)) * __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(z + a));
            __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(x + a));
          } else __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const f32x4*>(y + a)), reinterpret_cast<f32x4*>(x + a));
        }
      }
    }
  }
  if (OP == 2 || OP == 3) {
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f) sink[0] = acc[0];
  }
}

template <int PAT, int OP>
static void run(float* x, float* y, float* z, long long npts, float* sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 6;
  k<PAT, OP><<<256, 256>>>(x, y, z, npts, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) k<PAT, OP><<<256, 256>>>(x, y, z, npts, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double unit = 8.0 * npts * 1024.0;
  const double bytes = unit * (OP == 4 ? 3 : (OP == 5 ? 2 : 1));
  static const char* pn[] = {"16pt row-major", "32pt row-major", "tile-native", "32pt row-major 32B runs"};
  static const char* on[] = {"store nt", "store", "load nt", "load", "2 loads + 1 store (nt)", "copy (nt)"};
  printf("%-26s %-24s %7.3f ms  %6.2f TB/s\n", pn[PAT], on[OP], ms, bytes / ms * 1e-9);
}

int main(int argc, char** argv) {
  const long long npts = argc > 1 ? atoll(argv[1]) : 131072;
  const size_t bytes = (size_t)8 * npts * 1024;
  float *x, *y, *z, *sink;
  hipMalloc(&x, bytes); hipMalloc(&y, bytes); hipMalloc(&z, bytes); hipMalloc(&sink, 64);
  hipMemset(x, 0, bytes); hipMemset(y, 0, bytes); hipMemset(z, 0, bytes);
  printf("npts %lld: one array = %.2f GB\n", npts, bytes * 1e-9);
#define ALLOPS(P) run<P, 0>(x, y, z, npts, sink); run<P, 1>(x, y, z, npts, sink); run<P, 2>(x, y, z, npts, sink); run<P, 3>(x, y, z, npts, sink); \
  run<P, 4>(x, y, z, npts, sink); run<P, 5>(x, y, z, npts, sink);
  ALLOPS(0) ALLOPS(1) ALLOPS(2) ALLOPS(3)
  return 0;
}
