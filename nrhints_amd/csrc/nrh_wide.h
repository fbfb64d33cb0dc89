// Host-side interface between nrh_api.hip and the separately compiled wide-kernel translation unit (nrh_wide.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace nrh32 {

struct WideSdfCall {
  int mode;                 // 0 sdf | 1 + gradient | 2 + feature tiles | 3 sdf + derivative along the ray
  const void* streams;      // packing32.pack_sdf32: the three mode streams back to back
  const float* tables;      // [11][256]
  const float *ro, *rd, *t;
  float *sdf, *grad, *feat;
  void* scratch;            // >= wide_sdf_scratch_bytes(max_grid) bytes (modes 1, 2)
  long long npts;
  int n_per_ray, t_stride, sdf_stride;
  int max_grid;             // workgroups (one per CU)
};
int wide_sdf_launch(const WideSdfCall& c, hipStream_t st);   // 0 ok, -1 invalid, -2 launch/device error
long long wide_sdf_stream_bytes_total();
long long wide_sdf_scratch_bytes(int grid);

struct WideColorCall {
  const void* stream;       // packing32.pack_color32: 33 blocks
  const float* tables;      // [5][256]
  const float *part, *ro, *rd, *tmid, *nhat, *raymisc;
  float* color;
  long long nrays;
  int raymisc_stride;
  int max_grid;
};
int wide_color_launch(const WideColorCall& c, hipStream_t st);   // 0 ok, -1 invalid, -2 launch/device error
long long wide_color_stream_bytes();

}  // namespace nrh32

// the one-term builds of the wide SDF kernels (nrh_wide1.hip: precision "f16", a single fp16 MFMA pass per K step)
namespace nrh32t {
int wide_sdf_launch(const nrh32::WideSdfCall& c, hipStream_t st);
}
