// Translation unit of the wide (32-point tile, one wave per SIMD) f16x3 kernels.  Built separately from nrh_api.hip because
// these kernels need their own code generation flags (-fno-slp-vectorize: no packed f32 VALU beside MFMAs;
// -mllvm -amdgpu-mfma-vgpr-form: accumulators in arch VGPRs, the AGPR file belongs to the hand-placed B operands).
#include "nrh_sdf32.hip"
#include "nrh_color32.hip"
#include "nrh_wide.h"

namespace nrh32 {

static bool g_attr[16] = {};   // per device id: dynamic-LDS attribute set (idempotent; a race sets it twice)

int wide_sdf_launch(const WideSdfCall& c, hipStream_t st) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -2;
  if (dev >= 0 && dev < 16 && !g_attr[dev]) {
    const void* fns[4] = {(const void*)sdf32_kernel<0>, (const void*)sdf32_kernel<1>, (const void*)sdf32_kernel<2>,
                          (const void*)sdf32_kernel<3>};
    for (const void* f : fns)
      if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return -2;
    g_attr[dev] = true;
  }
  Sdf32Args a;
  long long off = 0;
  if (c.mode < 0 || c.mode > 3) return -1;
  const int smode = c.mode == 3 ? 0 : c.mode;     // mode 3 (forward-mode derivative along the ray) consumes the forward-only stream
  for (int m = 0; m < smode; ++m) off += sdf32_stream_bytes(m);
  a.w = reinterpret_cast<const char*>(c.streams) + off;
  a.tab = c.tables; a.ro = c.ro; a.rd = c.rd; a.t = c.t; a.sdf = c.sdf; a.grad = c.grad; a.feat = c.feat;
  a.scratch = reinterpret_cast<uint32_t*>(c.scratch);
  a.npts = c.npts; a.n_per_ray = c.n_per_ray; a.t_stride = c.t_stride; a.sdf_stride = c.sdf_stride;
  const int group = c.mode == 3 ? GROUP / 2 : GROUP;                  // mode 3: 16 points + 16 tangents per wave tile
  const long long groups = (c.npts + group - 1) / group;
  if (groups > 0x7fffffffLL) return -1;
  a.ngroups = (int)groups;
  a.dbg = nullptr; a.dbg_stage = 99;
  const int grid = (int)(groups < c.max_grid ? groups : c.max_grid);
  if (grid <= 0) return -2;
  if (c.mode == 0) hipLaunchKernelGGL(sdf32_kernel<0>, dim3(grid), dim3(THREADS), LDS_BYTES, st, a);
  else if (c.mode == 1) hipLaunchKernelGGL(sdf32_kernel<1>, dim3(grid), dim3(THREADS), LDS_BYTES, st, a);
  else if (c.mode == 2) hipLaunchKernelGGL(sdf32_kernel<2>, dim3(grid), dim3(THREADS), LDS_BYTES, st, a);
  else hipLaunchKernelGGL(sdf32_kernel<3>, dim3(grid), dim3(THREADS), LDS_BYTES, st, a);
  return 0;
}

int wide_color_launch(const WideColorCall& c, hipStream_t st) {
  static bool attr[16] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -2;
  if (dev >= 0 && dev < 16 && !attr[dev]) {
    if (hipFuncSetAttribute((const void*)color32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, COL32_LDS_BYTES) != hipSuccess) return -2;
    attr[dev] = true;
  }
  if (c.nrays > 0x7fffffffLL) return -1;
  Color32Args a;
  a.w = reinterpret_cast<const char*>(c.stream); a.tab = c.tables; a.part = c.part; a.ro = c.ro; a.rd = c.rd; a.tmid = c.tmid;
  a.nhat = c.nhat; a.raymisc = c.raymisc; a.color = c.color; a.nrays = (int)c.nrays; a.raymisc_stride = c.raymisc_stride;
  const int grid = (int)(c.nrays < c.max_grid ? c.nrays : c.max_grid);
  if (grid <= 0) return -2;
  hipLaunchKernelGGL(color32_kernel, dim3(grid), dim3(THREADS), COL32_LDS_BYTES, st, a);
  return 0;
}
long long wide_color_stream_bytes() { return color32_stream_bytes(); }

long long wide_sdf_stream_bytes_total() { return sdf32_stream_bytes(0) + sdf32_stream_bytes(1) + sdf32_stream_bytes(2); }
long long wide_sdf_scratch_bytes(int grid) { return (long long)grid * WAVES * SCRATCH_WORDS_PER_WAVE * 4; }

}  // namespace nrh32
