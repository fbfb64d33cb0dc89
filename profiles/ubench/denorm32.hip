// Do v_mfma_f32_32x32x16_f16 and v_cvt_pkrtz_f16_f32 honour fp16 subnormals on gfx950?  (the wide kernels keep the activation
// residual unscaled: values below 2^-14 reach the MFMA as subnormal fp16)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float av, float bv) {
  // A[row][k]: lane (row = lane & 31, kgroup = lane >> 5) holds k = 8 kgroup + j;  B likewise for columns
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.0f; b[i] = (_Float16)0.0f; }
  const h16x2 pa = __builtin_amdgcn_cvt_pkrtz(av, 0.0f);    // the instruction the epilogues use
  a[0] = (_Float16)pa[0];
  b[0] = (_Float16)bv;
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  out[threadIdx.x] = c[0];
  if (threadIdx.x == 0) out[64] = (float)pa[0];
}
int main() {
  float* o; hipMalloc(&o, 65 * 4);
  float tests[][2] = {{9.5367431640625e-07f /*2^-20*/, 1024.0f}, {3.0e-5f, 1024.0f}, {6.0e-8f, 16384.0f}, {1.0f, 1.0f}};
  for (auto& t : tests) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, t[0], t[1]);
    float h[65]; hipMemcpy(h, o, 65 * 4, hipMemcpyDeviceToHost);
    printf("a=%.6e b=%.6e  cvt_pkrtz(a)=%.9e  D[0][0]=%.9e  (2 lanes-groups contribute: expected 2 * a16 * b16 = %.9e)\n", t[0], t[1], h[64], h[0],
           2.0 * (double)(float)(_Float16)t[0] * (double)(float)(_Float16)t[1]);
  }
  return 0;
}
