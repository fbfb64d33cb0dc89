// The SDF network's training kernels (nrh_sdf.hip MODE 3, nrh_sdf_train.hip) with FOUR waves per workgroup, for small batches.
//
// The 16-point kernels share one weight stream among the 8 waves of a workgroup (two per SIMD): the right shape when there are
// many more tiles than SIMDs.  A 64- or 128-ray training batch (the reference's per-rank share under 8-way DDP,
// trainer/trainer.py:116-123) has 512 - 1 024 tiles for 1 024 SIMDs: with 8-wave workgroups they sit two to a SIMD on 64 - 128
// CUs while the other CUs idle.  The same source compiled with NRH_WG_WAVES = 4 puts one wave on every SIMD of twice as many CUs:
// 0.258 -> 0.202 ms (training forward), 0.172 -> 0.109 (tangent sweep), 0.150 -> 0.102 (value sweep) at 64 rays
// (profiles/r04/wg4_ab.log); per-tile arithmetic and results are unchanged bit for bit.  nrh_api.hip takes these builds while
// the batch has at most 4 tiles per CU and the default ones above (at 256 rays the 8-wave build wins again).
#include <hip/hip_runtime.h>
#include <string.h>
#define NRH_WG_WAVES 4
#define nrh nrh4
#include "nrh_sdf.hip"
#include "nrh_sdf_train.hip"
#undef nrh
#include "nrh_small.h"

namespace nrh4s {

static_assert(nrh4::WG_WAVES == WAVES, "this unit is the 4-wave build");
static bool g_attr[64];

static int ensure() {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return -2;
  if (g_attr[dev]) return 0;
  const void* fns[] = {(const void*)nrh4::sdf_kernel<3, 0>, (const void*)nrh4::sdf_kernel<3, 1>,
                       (const void*)nrh4::sdf_tangent_kernel<0>, (const void*)nrh4::sdf_tangent_kernel<1>,
                       (const void*)nrh4::sdf_adjoint_kernel<0>, (const void*)nrh4::sdf_adjoint_kernel<1>};
  for (const void* f : fns)
    if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, nrh4::MLP_LDS_BYTES) != hipSuccess) return -2;
  g_attr[dev] = true;
  return 0;
}

static int geometry(long long npts, int max_grid, int& groups, int& grid) {
  const long long g = (npts + 16 * WAVES - 1) / (16 * WAVES);
  if (g > 0x7fffffffLL || g <= 0 || max_grid <= 0) return -1;
  groups = (int)g;
  grid = (int)(g < max_grid ? g : max_grid);
  return 0;
}

int launch_sdf_train_forward(int precision, const void* args, size_t bytes, int max_grid, hipStream_t st) {
  if (bytes != sizeof(nrh4::SdfArgs) || (precision != 0 && precision != 1)) return -1;
  nrh4::SdfArgs a;
  memcpy(&a, args, sizeof(a));
  int grid = 0;
  if (geometry(a.npts, max_grid, a.ntile_groups, grid)) return -1;
  if (ensure()) return -2;
  if (precision == 0) hipLaunchKernelGGL((nrh4::sdf_kernel<3, 0>), dim3(grid), dim3(nrh4::MLP_THREADS), nrh4::MLP_LDS_BYTES, st, a);
  else hipLaunchKernelGGL((nrh4::sdf_kernel<3, 1>), dim3(grid), dim3(nrh4::MLP_THREADS), nrh4::MLP_LDS_BYTES, st, a);
  return 0;
}

int launch_sdf_train_sweeps(int precision, const void* args, size_t bytes, int max_grid, hipStream_t st) {
  if (bytes != sizeof(nrh4::SdfTrainArgs) || (precision != 0 && precision != 1)) return -1;
  nrh4::SdfTrainArgs a;
  memcpy(&a, args, sizeof(a));
  int grid = 0;
  if (geometry(a.npts, max_grid, a.ntile_groups, grid)) return -1;
  if (ensure()) return -2;
  const dim3 g(grid), blk(nrh4::MLP_THREADS);
  if (precision == 0) hipLaunchKernelGGL((nrh4::sdf_tangent_kernel<0>), g, blk, nrh4::MLP_LDS_BYTES, st, a);
  else hipLaunchKernelGGL((nrh4::sdf_tangent_kernel<1>), g, blk, nrh4::MLP_LDS_BYTES, st, a);
  if (hipGetLastError() != hipSuccess) return -2;
  if (precision == 0) hipLaunchKernelGGL((nrh4::sdf_adjoint_kernel<0>), g, blk, nrh4::MLP_LDS_BYTES, st, a);
  else hipLaunchKernelGGL((nrh4::sdf_adjoint_kernel<1>), g, blk, nrh4::MLP_LDS_BYTES, st, a);
  return 0;
}

}  // namespace nrh4s
