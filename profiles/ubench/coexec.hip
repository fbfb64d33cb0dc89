// Micro-benchmark: do MFMA and VALU work of two waves on ONE SIMD overlap on gfx950?  (measurement aid, not product)
// Block = 512 threads = 8 waves: waves w and w+4 share a SIMD.  role[w]: 0 idle, 1 MFMA loop, 2 VALU loop, 3 TRANS
// loop, 4 interleaved MFMA+VALU in one wave.  One block per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512, 2) void k(const int* roles, float* out, int iters, int nvalu) {
  const int wave = threadIdx.x >> 6;
  const int role = roles[wave];
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
  f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  float v0 = threadIdx.x * 1e-3f, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
  if (role == 1) {
    for (int it = 0; it < iters; ++it) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
    }
  } else if (role == 2) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        v0 = v0 * 1.0001f + 0.5f; v1 = v1 * 1.0001f + 0.5f; v2 = v2 * 1.0001f + 0.5f; v3 = v3 * 1.0001f + 0.5f;
        v4 = v4 * 1.0001f + 0.5f; v5 = v5 * 1.0001f + 0.5f; v6 = v6 * 1.0001f + 0.5f; v7 = v7 * 1.0001f + 0.5f;
      }
    }
  } else if (role == 3) {
    for (int it = 0; it < iters; ++it) {
      v0 = __builtin_amdgcn_exp2f(v0); v1 = __builtin_amdgcn_exp2f(v1); v2 = __builtin_amdgcn_logf(v2); v3 = __builtin_amdgcn_logf(v3);
      v4 = __builtin_amdgcn_exp2f(v4); v5 = __builtin_amdgcn_exp2f(v5); v6 = __builtin_amdgcn_rcpf(v6); v7 = __builtin_amdgcn_rcpf(v7);
    }
  } else if (role == 4) {
    for (int it = 0; it < iters; ++it) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
      if (nvalu > 0) { v0 = v0 * 1.0001f + 0.5f; v1 = v1 * 1.0001f + 0.5f; }
      if (nvalu > 2) { v2 = v2 * 1.0001f + 0.5f; v3 = v3 * 1.0001f + 0.5f; }
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
      if (nvalu > 0) { v4 = v4 * 1.0001f + 0.5f; v5 = v5 * 1.0001f + 0.5f; }
      if (nvalu > 2) { v6 = v6 * 1.0001f + 0.5f; v7 = v7 * 1.0001f + 0.5f; }
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
      if (nvalu > 0) { v0 = v0 * 1.0001f + 0.25f; v1 = v1 * 1.0001f + 0.25f; }
      if (nvalu > 2) { v2 = v2 * 1.0001f + 0.25f; v3 = v3 * 1.0001f + 0.25f; }
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
      if (nvalu > 0) { v4 = v4 * 1.0001f + 0.25f; v5 = v5 * 1.0001f + 0.25f; }
      if (nvalu > 2) { v6 = v6 * 1.0001f + 0.25f; v7 = v7 * 1.0001f + 0.25f; }
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}

static float run(const int* hroles, int iters, int nvalu) {
  int* d; float* o;
  hipMalloc(&d, 8 * sizeof(int)); hipMalloc(&o, 256 * 512 * sizeof(float));
  hipMemcpy(d, hroles, 8 * sizeof(int), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, o, iters, nvalu);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d, o, iters, nvalu);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(d); hipFree(o);
  return ms;
}
int main() {
  const int it = 200000;
  struct { const char* name; int r[8]; int nv; } cases[] = {
    {"mfma x4 (1/SIMD)          ", {1,1,1,1,0,0,0,0}, 0},
    {"mfma x8 (2/SIMD)          ", {1,1,1,1,1,1,1,1}, 0},
    {"valu x4 (16 fma/iter)     ", {2,2,2,2,0,0,0,0}, 0},
    {"valu x8                   ", {2,2,2,2,2,2,2,2}, 0},
    {"mfma x4 + valu x4 sameSIMD", {1,1,1,1,2,2,2,2}, 0},
    {"trans x4 (8 trans/iter)   ", {3,3,3,3,0,0,0,0}, 0},
    {"mfma x4 + trans x4        ", {1,1,1,1,3,3,3,3}, 0},
    {"interleaved 4mfma+0valu x4", {4,4,4,4,0,0,0,0}, 0},
    {"interleaved 4mfma+8valu x4", {4,4,4,4,0,0,0,0}, 2},
    {"interleaved 4mfma+16valu x4", {4,4,4,4,0,0,0,0}, 4},
    {"interleaved 4mfma+16valu x8", {4,4,4,4,4,4,4,4}, 4},
  };
  for (auto& c : cases) {
    float ms = run(c.r, it, c.nv);
    printf("%s  %8.3f ms   %6.2f ns/iter  (~%5.1f cycles/iter @2.4GHz)\n", c.name, ms, ms * 1e6 / it, ms * 1e6 / it * 2.4);
  }
  return 0;
}
