// Does the instruction offset of global_load_lds_dwordx4 move the LDS destination as well as the global source?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
__global__ void k(const uint32_t* src, uint32_t* out) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)lds;
  const uint32_t voff = lane * 16;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\ts_waitcnt vmcnt(0)" ::"s"(base), "v"(voff), "s"(src) : "memory");
  __syncthreads();
  for (int i = lane; i < 1024; i += 64) out[i] = lds[i];
}
int main() {
  std::vector<uint32_t> h(2048);
  for (int i = 0; i < 2048; ++i) h[i] = i;
  uint32_t *d_src, *d_out;
  hipMalloc(&d_src, 8192); hipMalloc(&d_out, 4096);
  hipMemcpy(d_src, h.data(), 8192, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d_src, d_out);
  std::vector<uint32_t> o(1024);
  hipMemcpy(o.data(), d_out, 4096, hipMemcpyDeviceToHost);
  printf("lds[0]=%u lds[255]=%u lds[256]=%u lds[511]=%u lds[512]=%x\n", o[0], o[255], o[256], o[511], o[512]);
  if (o[0] == 256) printf("RESULT: inst offset moves the GLOBAL address only (LDS dest = M0 + lane*16)\n");
  else if (o[256] == 256) printf("RESULT: inst offset moves BOTH global and LDS addresses\n");
  else printf("RESULT: unexpected\n");
  return 0;
}
