// Does v_mfma_f32_16x16x32_f16 honour fp16 subnormal inputs on gfx950? (measurement aid)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, float av, float bv) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.0f; b[i] = (_Float16)0.0f; }
  a[0] = (_Float16)av;  // every lane: A[i][k-slot 0 of its q]
  b[0] = (_Float16)bv;
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  out[threadIdx.x] = c[0];
}
int main() {
  float* o; hipMalloc(&o, 64 * 4);
  float tests[][2] = {{9.5367431640625e-07f /*2^-20 subnormal*/, 1024.0f}, {6.0e-8f, 16384.0f}, {3.0e-5f, 3.0e-5f}, {1.0f, 1.0f}};
  for (auto& t : tests) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, t[0], t[1]);
    float h[64]; hipMemcpy(h, o, 256, hipMemcpyDeviceToHost);
    printf("a=%.6e b=%.6e  -> D[0][0]=%.9e  (expected 4 * a16*b16 = %.9e)\n", t[0], t[1], h[0], 4.0 * (double)(float)(_Float16)t[0] * (double)(float)(_Float16)t[1]);
  }
  return 0;
}
