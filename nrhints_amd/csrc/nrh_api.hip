// C ABI of libnrhints_hip.so (declared in include/nrhints_hip.h).  Single translation unit: the kernel sources
// are included here so one hipcc invocation builds the whole library for gfx950.
#include "nrh_sdf.hip"
#include "nrh_sdf_train.hip"
#include "nrh_sdf_split.hip"
#include "nrh_sdf_train_split.hip"
#include "nrh_color.hip"
#include "nrh_color_split.hip"
#include "nrh_outside.hip"
#include "nrh_rays.hip"
#include "nrh_rays_train.hip"
#include "nrh_fold.hip"
#include "nrh_dw.hip"
#include "nrh_train_fused.hip"
#include "nrh_adam.hip"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/nrhints_hip.h"
#include "nrh_wide.h"
#include "nrh_small.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, const char* a = "", long long b = 0) {
  snprintf(g_err, sizeof(g_err), fmt, a, b);
  return code;
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: launch failed: %s", what, hipGetErrorString(e));
    return NRH_E_LAUNCH;
  }
  return NRH_OK;
}

// Per-device caches, indexed by the CURRENT device id (the caller makes the tensors' device current; nrhints_amd/_lib.py does).
// Plain ints / bools written with the same value by every thread that races on them: idempotent, no lock needed.
constexpr int MAX_DEVICES = 64;
int g_cus[MAX_DEVICES] = {};
bool g_attr_done[MAX_DEVICES] = {};

int current_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return -1;
  return dev;
}

int device_cus() {
  const int dev = current_device();
  if (dev < 0) return 0;
  if (g_cus[dev] == 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    g_cus[dev] = prop.multiProcessorCount;
  }
  return g_cus[dev];
}

int mlp_grid() {
  const int cus = device_cus();
  return cus > 0 ? cus * (8 / nrh::WG_WAVES) : 0;  // 8 resident waves per CU (2 per SIMD at 256 VGPRs)
}

int ensure_attrs() {
  const int dev = current_device();
  if (dev < 0) return fail(NRH_E_LAUNCH, "no HIP device%s", "");
  if (g_attr_done[dev]) return NRH_OK;
  hipError_t e;
  const void* fns[] = {(const void*)nrh::sdf_kernel<0, 0>, (const void*)nrh::sdf_kernel<1, 0>, (const void*)nrh::sdf_kernel<2, 0>,
                       (const void*)nrh::sdf_kernel<0, 1>, (const void*)nrh::sdf_kernel<1, 1>, (const void*)nrh::sdf_kernel<2, 1>,
                       (const void*)nrh::sdf_kernel<3, 0>, (const void*)nrh::sdf_kernel<3, 1>,
                       (const void*)nrh::sdf_tangent_kernel<0>, (const void*)nrh::sdf_tangent_kernel<1>,
                       (const void*)nrh::sdf_adjoint_kernel<0>, (const void*)nrh::sdf_adjoint_kernel<1>,
                       (const void*)nrh::color_kernel<0, 8, true>, (const void*)nrh::color_kernel<1, 8, true>,
                       (const void*)nrh::color_kernel<0, 4, true>, (const void*)nrh::color_kernel<1, 4, true>,
                       (const void*)nrh::color_adjoint_kernel<0, 8>, (const void*)nrh::color_adjoint_kernel<1, 8>,
                       (const void*)nrh::color_adjoint_kernel<0, 4>, (const void*)nrh::color_adjoint_kernel<1, 4>,
                       (const void*)nrh::color_kernel<0, 8>, (const void*)nrh::color_kernel<1, 8>,
                       (const void*)nrh::color_kernel<0, 4>, (const void*)nrh::color_kernel<1, 4>,
                       (const void*)nrh::color_kernel<1, 8, false, true>, (const void*)nrh::color_kernel<1, 4, false, true>};
  e = hipSuccess;
  for (const void* f : fns)
    if (e == hipSuccess) e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, nrh::MLP_LDS_BYTES);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)nrh::sdf_split_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, nrh::split_lds_bytes(1));
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)nrh::sdf_split_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, nrh::split_lds_bytes(2));
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)nrh::sdf_train_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, nrh::SPLT_LDS_BYTES);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)nrh::sdf_grad_split_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, nrh::SPLG_LDS_BYTES);
  if (e != hipSuccess) return fail(NRH_E_LAUNCH, "hipFuncSetAttribute failed: %s", hipGetErrorString(e));
  g_attr_done[dev] = true;
  return NRH_OK;
}

inline long long round64(long long x) { return (x + 63) / 64 * 64; }

// ---- optional live kernel timing (bench.py's roofline leg): HIP events around one kernel family ----
struct TimedLaunch { hipEvent_t a, b; };
int g_time_kind = -1;  // -1 off; 0/1/2 = sdf_kernel<mode>; 3 = color_kernel
std::vector<TimedLaunch> g_timed;
void timing_begin(int kind, hipStream_t st, TimedLaunch& t, bool& on) {
  on = (kind == g_time_kind);
  if (!on) return;
  (void)hipEventCreate(&t.a);
  (void)hipEventCreate(&t.b);
  (void)hipEventRecord(t.a, st);
}
void timing_end(hipStream_t st, TimedLaunch& t, bool on) {
  if (!on) return;
  (void)hipEventRecord(t.b, st);
  g_timed.push_back(t);
}

// wide f16x3 evaluation kernels (nrh_sdf32.hip): taken when the caller supplies their packed streams
struct WideNet {
  const void* streams = nullptr;
  const float* tables = nullptr;
};
// NrhNet.precision 2 ("f16": the single-pass, one-term builds of the wide SDF kernels, nrh_wide1.hip): set for the duration of one
// render call by render_forward_impl / nrh_sdf_eval_wide_f16 (a call runs on the caller's thread from entry to return)
static thread_local int t_wide_one_term = 0;
struct OneTermScope {
  int saved;
  explicit OneTermScope(int on) : saved(t_wide_one_term) { t_wide_one_term = on; }
  ~OneTermScope() { t_wide_one_term = saved; }
};

int sdf_eval_impl(int prec, int mode, const float* w, const float* b, const float* head, const float* ro, const float* rd,
                  const float* t, int t_stride, int n_per_ray, long long nrays, float* sdf, int sdf_stride, float* grad,
                  float* feat, float* scratch, hipStream_t st, const WideNet wide = WideNet()) {
  if (prec == 1 && wide.streams && wide.tables) {
    if (mode < 0 || mode > 3) return fail(NRH_E_INVALID, "nrh_sdf_eval: mode must be 0, 1, 2 or (wide kernels) 3%s", "");
    if (!ro || !rd || !t || !sdf) return fail(NRH_E_INVALID, "nrh_sdf_eval: null pointer%s", "");
    if (mode == 3 && !grad) return fail(NRH_E_INVALID, "nrh_sdf_eval: mode 3 needs grad%s", "");
    if ((mode == 1 || mode == 2) && (!grad || !scratch)) return fail(NRH_E_INVALID, "nrh_sdf_eval: mode %s needs grad and scratch", mode == 1 ? "1" : "2");
    if (mode == 2 && !feat) return fail(NRH_E_INVALID, "nrh_sdf_eval: mode 2 needs feat%s", "");
    if (n_per_ray <= 0 || nrays < 0 || t_stride < n_per_ray || sdf_stride < n_per_ray)
      return fail(NRH_E_INVALID, "nrh_sdf_eval: bad n_per_ray/stride%s", "");
    if (nrays == 0) return NRH_OK;
    nrh32::WideSdfCall c;
    c.mode = mode; c.streams = wide.streams; c.tables = wide.tables; c.ro = ro; c.rd = rd; c.t = t; c.sdf = sdf; c.grad = grad;
    c.feat = feat; c.scratch = scratch; c.npts = nrays * n_per_ray; c.n_per_ray = n_per_ray; c.t_stride = t_stride;
    c.sdf_stride = sdf_stride; c.max_grid = device_cus();
    TimedLaunch tl;
    bool timed;
    timing_begin(mode == 3 ? 1 : mode, st, tl, timed);
    const int wrc = t_wide_one_term ? nrh32t::wide_sdf_launch(c, st) : nrh32::wide_sdf_launch(c, st);
    timing_end(st, tl, timed);
    if (wrc == -1) return fail(NRH_E_INVALID, "nrh_sdf_eval: too many points%s", "");
    if (wrc) return fail(NRH_E_LAUNCH, "wide sdf kernel: no HIP device / attribute error%s", "");
    return check_launch("sdf32_kernel");
  }
  if (mode < 0 || mode > 2) return fail(NRH_E_INVALID, "nrh_sdf_eval: mode must be 0, 1 or 2%s", "");
  if (prec < 0 || prec > 1) return fail(NRH_E_INVALID, "nrh_sdf_eval: precision must be 0 (f32) or 1 (f16x3)%s", "");
  if (!w || !b || !head || !ro || !rd || !t || !sdf) return fail(NRH_E_INVALID, "nrh_sdf_eval: null pointer%s", "");
  if (mode >= 1 && (!grad || !scratch)) return fail(NRH_E_INVALID, "nrh_sdf_eval: mode %s needs grad and scratch", mode == 1 ? "1" : "2");
  if (mode == 2 && !feat) return fail(NRH_E_INVALID, "nrh_sdf_eval: mode 2 needs feat%s", "");
  if (n_per_ray <= 0 || nrays < 0 || t_stride < n_per_ray || sdf_stride < n_per_ray)
    return fail(NRH_E_INVALID, "nrh_sdf_eval: bad n_per_ray/stride%s", "");
  if (nrays == 0) return NRH_OK;
  int rc = ensure_attrs();
  if (rc) return rc;
  nrh::SdfArgs a;
  a.w = w; a.b = b; a.head = head; a.ro = ro; a.rd = rd; a.t = t; a.sdf = sdf; a.grad = grad; a.feat = feat;
  a.scratch = scratch;
  a.npts = nrays * n_per_ray;
  a.n_per_ray = n_per_ray; a.t_stride = t_stride; a.sdf_stride = sdf_stride;
  const long long groups = (a.npts + 16 * nrh::WG_WAVES - 1) / (16 * nrh::WG_WAVES);
  if (groups > 0x7fffffffLL) return fail(NRH_E_INVALID, "nrh_sdf_eval: too many points%s", "");
  a.ntile_groups = (int)groups;
  const int grid = (int)(groups < mlp_grid() ? groups : mlp_grid());
  if (grid <= 0) return fail(NRH_E_LAUNCH, "no HIP device%s", "");
  TimedLaunch tl;
  bool timed;
  timing_begin(mode, st, tl, timed);
  const dim3 g(grid), blk(nrh::MLP_THREADS);
  const int lds = nrh::MLP_LDS_BYTES;
  if (prec == 0) {
    if (mode == 0) hipLaunchKernelGGL((nrh::sdf_kernel<0, 0>), g, blk, lds, st, a);
    else if (mode == 1) hipLaunchKernelGGL((nrh::sdf_kernel<1, 0>), g, blk, lds, st, a);
    else hipLaunchKernelGGL((nrh::sdf_kernel<2, 0>), g, blk, lds, st, a);
  } else {
    if (mode == 0) hipLaunchKernelGGL((nrh::sdf_kernel<0, 1>), g, blk, lds, st, a);
    else if (mode == 1) hipLaunchKernelGGL((nrh::sdf_kernel<1, 1>), g, blk, lds, st, a);
    else hipLaunchKernelGGL((nrh::sdf_kernel<2, 1>), g, blk, lds, st, a);
  }
  timing_end(st, tl, timed);
  return check_launch("sdf_kernel");
}

// launch geometry shared by the per-point MLP kernels
int mlp_launch_geometry(long long npts, int& groups_out, int& grid_out, const char* who) {
  const long long groups = (npts + 16 * nrh::WG_WAVES - 1) / (16 * nrh::WG_WAVES);
  if (groups > 0x7fffffffLL) return fail(NRH_E_INVALID, "%s: too many points", who);
  groups_out = (int)groups;
  grid_out = (int)(groups < mlp_grid() ? groups : mlp_grid());
  if (grid_out <= 0) return fail(NRH_E_LAUNCH, "no HIP device%s", "");
  return NRH_OK;
}

// Kernel selection is a function of the call's arguments, not of the process environment: the A/B overrides below are read ONLY
// when NRH_PROFILING=1 is set as well (profiles/*.sh set it); without it a stray variable changes nothing.
static const char* prof_env(const char* name) {
  static const bool profiling = getenv("NRH_PROFILING") && atoi(getenv("NRH_PROFILING")) != 0;
  return profiling ? getenv(name) : nullptr;
}

// Small batches (at most 4 sixteen-point tiles per CU): the SDF training kernels' 4-wave builds (csrc/nrh_small.hip) put one wave on
// every SIMD of twice as many CUs.  NRH_PROFILING=1 NRH_SMALL_WG=0 in the environment keeps the 8-wave builds (A/B runs).
bool small_batch(long long npts) {
  static const bool on = !(prof_env("NRH_SMALL_WG") && atoi(prof_env("NRH_SMALL_WG")) == 0);
  return on && nrh::WG_WAVES == 8 && npts <= 16LL * 4 * device_cus();
}

// ... and below that, for precision f16x3, the channel-split training kernels (csrc/nrh_sdf_train_split.hip): one tile per workgroup,
// its stages' output channels over the four waves.  NRH_PROFILING=1 NRH_SPLIT_TRAIN=0 in the environment keeps the 4-wave builds (A/B runs).
bool split_train(int precision, long long npts) {
  static const bool on = !(prof_env("NRH_SPLIT_TRAIN") && atoi(prof_env("NRH_SPLIT_TRAIN")) == 0);
  return on && precision == 1 && npts <= 16LL * 2 * device_cus();      // (at 4 tiles per CU the 4-wave builds are as fast: profiles/r04/tsplit_ab.log)
}

// ... and the reflectance net's training forward / adjoint sweep (csrc/nrh_color_split.hip) while single tiles fit the CUs twice:
// above that the 8-wave kernels' shared weight stream wins again.  NRH_PROFILING=1 NRH_SPLIT_COLOR_MAX_PTS=n overrides (A/B runs).
bool split_color(int precision, long long npts) {
  static const long long forced = prof_env("NRH_SPLIT_COLOR_MAX_PTS") ? atoll(prof_env("NRH_SPLIT_COLOR_MAX_PTS")) : -1;
  const long long limit = forced >= 0 ? forced : 16LL * 2 * device_cus();
  return precision == 1 && npts % 16 == 0 && npts <= limit;
}

int sampler_step_impl(const nrh::StepArgs& a, hipStream_t st) {
  const int blocks = (a.nrays + nrh::RAYS_PER_BLOCK - 1) / nrh::RAYS_PER_BLOCK;
  hipLaunchKernelGGL(nrh::sampler_step_kernel, dim3(blocks), dim3(256), 0, st, a);
  return check_launch("sampler_step_kernel");
}

// SDF values of a SMALL point set on the channel-split kernel (csrc/nrh_sdf_split.hip; f16x3 stages, bit-identical to sdf_kernel<0, 1>).
// tiles: 16-point tiles per workgroup (1 or 2; 0 = one while single tiles fit the CUs once, two above)
#ifndef NRH_SPLIT_GRAD_MAX_PTS
#define NRH_SPLIT_GRAD_MAX_PTS 12288    // ... and the shadow rays' sdf + gradient pass of a training batch of at most this many points (96 rays; profiles/r04/grad_split_bench.log)
#endif
#ifndef NRH_SPLIT_MAX_PTS
#define NRH_SPLIT_MAX_PTS 16384     // the training sampler takes the split kernel for passes of at most this many points (0: never)
#endif
static int g_sampler_fusion = 1;      // nrh_sampler_fusion: tests / measurements only
int sdf_split_impl(const float* w, const float* b, const float* head, const float* ro, const float* rd, const float* t, int t_stride,
                   int n_per_ray, long long nrays, float* sdf, int sdf_stride, int tiles, hipStream_t st,
                   const nrh::StepArgs* step = nullptr) {
  if (!w || !b || !head || !ro || !rd || !t || !sdf) return fail(NRH_E_INVALID, "nrh_sdf_eval_split: null pointer%s", "");
  if (n_per_ray <= 0 || nrays < 0 || t_stride < n_per_ray || sdf_stride < n_per_ray)
    return fail(NRH_E_INVALID, "nrh_sdf_eval_split: bad n_per_ray/stride%s", "");
  if (tiles < 0 || tiles > 2) return fail(NRH_E_INVALID, "nrh_sdf_eval_split: tiles must be 0 (auto), 1 or 2%s", "");
  if (nrays == 0) return NRH_OK;
  const long long npts = nrays * n_per_ray;
  if (npts > (1LL << 24)) return fail(NRH_E_INVALID, "nrh_sdf_eval_split: at most 16 777 216 points per call%s", "");
  const int arc = ensure_attrs();
  if (arc) return arc;
  if (tiles == 0) {
    static const int forced = prof_env("NRH_SPLIT_TILES") ? atoi(prof_env("NRH_SPLIT_TILES")) : 0;       // profiling override
    tiles = (forced == 1 || forced == 2) ? forced : (npts > 16LL * device_cus()) ? 2 : 1;            // one tile per workgroup while that fills the CUs once
  }
  nrh::SdfSplitArgs a;
  a.w = w; a.b = b; a.head = head; a.ro = ro; a.rd = rd; a.t = t; a.sdf = sdf; a.npts = npts; a.n_per_ray = n_per_ray;
  a.t_stride = t_stride; a.sdf_stride = sdf_stride;
  a.fused_step = 0;
  if (step) {       // the per-ray sampler step in the launch's tail: a tile must be exactly one ray's 16 new samples
    if (n_per_ray != 16 || t_stride != 16 || sdf_stride != 16 || step->n_new != 16 || step->nrays != nrays)
      return fail(NRH_E_INVALID, "sdf_split: a fused sampler step needs 16 samples per ray%s", "");
    a.fused_step = 1;
    a.step = *step;
  }
  const unsigned grid = (unsigned)((npts + 16 * tiles - 1) / (16 * tiles));
  TimedLaunch tl;
  bool timed;
  timing_begin(0, st, tl, timed);
  if (tiles == 1) hipLaunchKernelGGL(nrh::sdf_split_kernel<1>, dim3(grid), dim3(256), nrh::split_lds_bytes(1), st, a);
  else hipLaunchKernelGGL(nrh::sdf_split_kernel<2>, dim3(grid), dim3(256), nrh::split_lds_bytes(2), st, a);
  timing_end(st, tl, timed);
  return check_launch("sdf_split_kernel");
}

// sdf + d sdf / dx of a SMALL point set on the channel-split kernel (csrc/nrh_sdf_train_split.hip sdf_grad_split_kernel; f16x3 stages,
// bit-identical to sdf_kernel<1, 1>)
int sdf_grad_split_impl(const float* w, const float* b, const float* head, const float* ro, const float* rd, const float* t, int t_stride,
                        int n_per_ray, long long nrays, float* sdf, int sdf_stride, float* grad, hipStream_t st) {
  if (!w || !b || !head || !ro || !rd || !t || !sdf || !grad) return fail(NRH_E_INVALID, "nrh_sdf_grad_split: null pointer%s", "");
  if (n_per_ray <= 0 || nrays < 0 || t_stride < n_per_ray || sdf_stride < n_per_ray)
    return fail(NRH_E_INVALID, "nrh_sdf_grad_split: bad n_per_ray/stride%s", "");
  if (nrays == 0) return NRH_OK;
  const long long npts = nrays * n_per_ray;
  if (npts > (1LL << 24)) return fail(NRH_E_INVALID, "nrh_sdf_grad_split: at most 16 777 216 points per call%s", "");
  const int arc = ensure_attrs();
  if (arc) return arc;
  nrh::SdfArgs a;
  memset(&a, 0, sizeof(a));
  a.w = w; a.b = b; a.head = head; a.ro = ro; a.rd = rd; a.t = t; a.sdf = sdf; a.grad = grad; a.npts = npts; a.n_per_ray = n_per_ray;
  a.t_stride = t_stride; a.sdf_stride = sdf_stride;
  TimedLaunch tl;
  bool timed;
  timing_begin(1, st, tl, timed);
  hipLaunchKernelGGL(nrh::sdf_grad_split_kernel, dim3((unsigned)((npts + 15) / 16)), dim3(256), nrh::SPLG_LDS_BYTES, st, a);
  timing_end(st, tl, timed);
  return check_launch("sdf_grad_split_kernel");
}

int color_eval_impl(int prec, int hints, const float* w, const float* b, const float* feat, const float* ro, const float* rd,
                    const float* tmid, const float* nhat, const float* raymisc, long long nrays, float* color,
                    hipStream_t st, int fused = 0, int misc_shift = 7) {
  if (!w || !b || !feat || !ro || !rd || !tmid || !nhat || !raymisc || !color)
    return fail(NRH_E_INVALID, "nrh_color_eval: null pointer%s", "");
  if (nrays == 0) return NRH_OK;
  int rc = ensure_attrs();
  if (rc) return rc;
  nrh::ColorArgs a;
  a.w = w; a.b = b; a.feat = feat; a.ro = ro; a.rd = rd; a.tmid = tmid; a.nhat = nhat; a.raymisc = raymisc;
  a.color = color; a.misc_shift = misc_shift;
  a.npts = nrays * 128;
  const long long groups = (a.npts + 16 * nrh::WG_WAVES - 1) / (16 * nrh::WG_WAVES);
  if (groups > 0x7fffffffLL) return fail(NRH_E_INVALID, "nrh_color_eval: too many points%s", "");
  a.ntile_groups = (int)groups;
  const int grid = (int)(groups < mlp_grid() ? groups : mlp_grid());
  if (grid <= 0) return fail(NRH_E_LAUNCH, "no HIP device%s", "");
  TimedLaunch tl;
  bool timed;
  timing_begin(3, st, tl, timed);
  const dim3 g(grid), blk(nrh::MLP_THREADS);
  const int lds = nrh::MLP_LDS_BYTES;
  if (fused && prec != 1) return fail(NRH_E_INVALID, "fused feature block needs precision 1 (f16x3, wide SDF kernels)%s", "");
  if (fused && hints) hipLaunchKernelGGL((nrh::color_kernel<1, 8, false, true>), g, blk, lds, st, a);
  else if (fused) hipLaunchKernelGGL((nrh::color_kernel<1, 4, false, true>), g, blk, lds, st, a);
  else if (prec == 0 && hints) hipLaunchKernelGGL((nrh::color_kernel<0, 8>), g, blk, lds, st, a);
  else if (prec == 1 && hints) hipLaunchKernelGGL((nrh::color_kernel<1, 8>), g, blk, lds, st, a);
  else if (prec == 0) hipLaunchKernelGGL((nrh::color_kernel<0, 4>), g, blk, lds, st, a);
  else if (prec == 1) hipLaunchKernelGGL((nrh::color_kernel<1, 4>), g, blk, lds, st, a);
  else return fail(NRH_E_INVALID, "nrh_color_eval: precision must be 0 (f32) or 1 (f16x3)%s", "");
  timing_end(st, tl, timed);
  return check_launch("color_kernel");
}

// Sample counts of one family of rays (models/neus_hint_model.py:139-171, :696-713, :373-412): nc coarse samples, `steps`
// up-sampling steps of n_new samples each (lin_new = linspace(0, 1, n_new), at most 16), nc + steps * n_new <= 128 in all.  The
// default {64, 4, 16} is the reference's 64 + 64 / 4.
struct SamplerPlan {
  int nc, steps, n_new;
  const float* lin_new;
  int total() const { return nc + steps * n_new; }
};
constexpr int LIN_TABLE_STRIDE = 128;     // NrhNet.lin_tables: [4][128] floats
// the primary rays' / the shadow rays' plan of a net: NrhNet.n_coarse == 0 -> the defaults with the caller's lin16 table
static SamplerPlan primary_plan(const NrhNet* net, const float* lin16) {
  if (!net || net->n_coarse == 0) return SamplerPlan{64, (net && net->samples == 64) ? 0 : 4, 16, lin16};
  return SamplerPlan{net->n_coarse, net->n_steps, net->n_new, net->lin_tables + 1 * LIN_TABLE_STRIDE};
}
static SamplerPlan shadow_plan(const NrhNet* net, const float* lin16) {
  if (!net || net->n_coarse == 0) return SamplerPlan{64, 4, 16, lin16};
  return SamplerPlan{net->s_coarse, net->s_new > 0 ? 4 : 0, net->s_new > 0 ? net->s_new : 16, net->lin_tables + 3 * LIN_TABLE_STRIDE};
}
static const char* plan_problem(const NrhNet* net) {
  if (!net || net->n_coarse == 0) return nullptr;
  if (!net->lin_tables) return "NrhNet.lin_tables is null";
  if (net->n_coarse < 2 || net->n_coarse > 128 || net->n_steps < 0 || net->n_steps > 8 || net->n_new < 1 || net->n_new > 16 ||
      net->n_coarse + net->n_steps * net->n_new > 128)
    return "primary ray: 2 <= n_coarse, n_new <= 16, n_coarse + n_steps * n_new <= 128";
  if (net->s_coarse < 2 || net->s_coarse > 64 || net->s_new < 0 || net->s_new > 16 || net->s_coarse + 4 * net->s_new > 128)
    return "shadow ray: 2 <= s_coarse <= 64, s_new <= 16, s_coarse + 4 * s_new <= 128";
  return nullptr;
}

// hierarchical sampling of one family of rays: coarse sdf, steps x (up-sample n_new, evaluate, merge), finalise sections
int run_sampler(const NrhNet* net, const float* ro, const float* rd, float* z, float* s, float* znew, float* snew,
                const SamplerPlan plan, const float* last_dist_ray, float last_dist, float* tmid, float* dists,
                long long n, hipStream_t st, bool latency = false) {
  if (plan.steps == 0) {
    // no hierarchical sampling (n_importance_samples = 0, :696): the coarse samples are final
    if (last_dist_ray) return fail(NRH_E_UNSUPPORTED, "run_sampler: a shadow ray without importance samples is finalised by its caller%s", "");
    hipLaunchKernelGGL(nrh::finalize64_kernel, dim3((unsigned)((n * 128 + 255) / 256)), dim3(256), 0, st, (const float*)z, last_dist, tmid,
                       dists, (int)n, plan.nc);
    return check_launch("finalize64_kernel");
  }
  const WideNet wide{net->sdf_w32, net->sdf_tab32};
  // a training step's passes are small and serial: below NRH_SPLIT_MAX_PTS points the channel-split kernel (a tile's MFMA work
  // over the CU's four SIMDs) instead of the one-wave-per-tile evaluation kernels.  Training only: a frame's chunks must not
  // change kernels with their size (bit-equal re-chunking, tests/test_gpu_fullsize.py)
  static const long long split_max = prof_env("NRH_SPLIT_MAX_PTS") ? atoll(prof_env("NRH_SPLIT_MAX_PTS")) : (long long)NRH_SPLIT_MAX_PTS;   // (env: profiling override)
  auto sdf0 = [&](const float* t, int stride, int per_ray, float* out) {
    if (latency && net->precision == 1 && n * per_ray <= split_max)
      return sdf_split_impl(net->sdf_w, net->sdf_b, net->sdf_head, ro, rd, t, stride, per_ray, n, out, stride, 0, st);
    return sdf_eval_impl(net->precision, 0, net->sdf_w, net->sdf_b, net->sdf_head, ro, rd, t, stride, per_ray, n, out, stride, nullptr,
                         nullptr, nullptr, st, wide);
  };
  // ... and with 16 new samples per step (a 16-point tile of such a pass IS one ray's new samples) the per-ray step that follows a
  // pass runs in the tail of the pass's own launch (nrh_sampler_fusion(0) switches it off: tests compare the two forms bit for bit)
  // Taken while a pass is ONE tile per workgroup (at most one ray per CU's worth of tiles: 256 rays on 256 CUs): there the tail costs
  // less than the separate launch's floor (0.90 -> 0.87 ms at 64 rays); with two tiles per workgroup the step serialises behind
  // the MLP of a workgroup that occupies its CU alone and the separate launch wins (6.19 against 6.31 ms at 1 024 rays,
  // profiles/r06/fused_step_ab.log).  nrh_sampler_fusion(2) forces it wherever the kernel supports it (tests).
  auto fuse_step = [&](int n_new) {
    const bool fits = g_sampler_fusion == 2 ? (n * 16 <= split_max) : (n * 16 <= 16LL * device_cus());
    return g_sampler_fusion && latency && net->precision == 1 && n_new == 16 && fits;
  };
  int rc = sdf0(z, 128, plan.nc, s);
  if (rc) return rc;
  // the per-ray step kernel: up-sample 0 | merge i + up-sample i + 1 (i = 0 .. steps - 2), with the SDF pass of the new samples
  // between them; the LAST step's samples are merged without their sdf (:328-329) and the sections finalised inside the launch
  // that drew them (StepArgs.then_finalize: the reference's last two steps have no SDF pass between them)
  nrh::StepArgs a;
  memset(&a, 0, sizeof(a));
  a.ro = ro; a.rd = rd; a.z = z; a.s = s; a.znew_in = znew; a.snew_in = snew; a.znew_out = znew; a.lin16 = plan.lin_new;
  a.last_dist_ray = last_dist_ray; a.tmid = tmid; a.dists = dists; a.last_dist = last_dist;
  a.nrays = (int)n; a.n_new = plan.n_new;
  a.inv_s = 64.0f; a.n = plan.nc; a.do_merge = 0; a.merge_sdf = 0; a.do_upsample = 1; a.do_finalize = 0;
  a.then_finalize = plan.steps == 1 ? 1 : 0;
  rc = sampler_step_impl(a, st);
  if (rc) return rc;
  for (int i = 0; i + 1 < plan.steps; ++i) {
    // sdf of the samples step i drew, then: merge them, up-sample step i + 1 from the merged state (inv_s = 64 * 2^(i+1)) - and, if
    // that was the last up-sample, merge ITS samples (no sdf) and finalise.  For small training batches with 16 new samples per
    // step the SDF pass and the step are ONE launch (sdf_split_kernel's fused tail: a tile of the pass is a ray's 16 samples)
    a.n = plan.nc + plan.n_new * i; a.do_merge = 1; a.merge_sdf = 1; a.do_upsample = 1; a.do_finalize = 0;
    a.inv_s = 64.0f * (float)(1 << (i + 1));
    a.then_finalize = (i + 2 == plan.steps) ? 1 : 0;
    if (fuse_step(plan.n_new)) {
      rc = sdf_split_impl(net->sdf_w, net->sdf_b, net->sdf_head, ro, rd, znew, 16, plan.n_new, n, snew, 16, 0, st, &a);
      if (rc) return rc;
      continue;
    }
    rc = sdf0(znew, 16, plan.n_new, snew);
    if (rc) return rc;
    rc = sampler_step_impl(a, st);
    if (rc) return rc;
  }
  return NRH_OK;
}

// core_alpha_kernel with the renderer's constants (shadow_ray_offset 1e-2, the four specular roughness values of
// models/neus_hint_model.py:161 evaluated in double like Python's scalars)
// 1 - shadow_ray_offset as the reference forms it: a Python double, rounded to float32 when it meets the tensor (:387)
float shadow_one_minus_offset(const NrhNet* net) {
  return (float)(1.0 - ((net && net->custom_consts) ? net->shadow_ray_offset : 1e-2));
}

int launch_core_alpha(nrh::CoreArgs& c, hipStream_t st, const NrhNet* net) {
  c.shadow_om = shadow_one_minus_offset(net);
  const double def[4] = {0.02, 0.05, 0.13, 0.34};
  const double* rough = (net && net->custom_consts) ? net->specular_roughness : def;
  for (int i = 0; i < 4; ++i) {
    const double k = (rough[i] + 1.0) * (rough[i] + 1.0) / 8.0, a2 = rough[i] * rough[i];
    c.kk[i] = (float)k; c.omk[i] = (float)(1.0 - k); c.a2[i] = (float)a2; c.a2m1[i] = (float)(a2 - 1.0);
  }
  hipLaunchKernelGGL(nrh::core_alpha_kernel, dim3((unsigned)((c.nrays + 3) / 4)), dim3(256), 0, st, c);
  return check_launch("core_alpha_kernel");
}

}  // namespace

extern "C" {

int nrh_version(void) { return 148; }
int nrh_train_arrays_tiled(void) { return nrh::arr_tiled(nrh::ARR_H) ? 1 : 0; }
#ifndef NRH_SOURCE_HASH
#define NRH_SOURCE_HASH "unknown"      // (a build outside csrc/Makefile: _lib.load() refuses it)
#endif
#ifndef NRH_BUILD_DEFS
#define NRH_BUILD_DEFS ""
#endif
const char* nrh_source_hash(void) { return NRH_SOURCE_HASH; }
const char* nrh_build_info(void) {
  return "nrhints_hip gfx950 mfma f32 16x16x4 | f16x3 16x16x32 | sources " NRH_SOURCE_HASH " | defs [" NRH_BUILD_DEFS "] | " __DATE__ " " __TIME__;
}
const char* nrh_last_error_string(void) { return g_err; }

int nrh_param_sizes(int* out) {
  if (!out) return fail(NRH_E_INVALID, "nrh_param_sizes: null%s", "");
  out[0] = nrh::SDF_PACKED_FLOATS;
  out[1] = nrh::SDF_BIAS_FLOATS;
  out[2] = nrh::SDF_HEAD_FLOATS;
  out[3] = nrh::COL_PACKED_FLOATS;
  out[4] = nrh::COL_BIAS_FLOATS;
  out[5] = nrh::RAYMISC_STRIDE;
  out[6] = nrh::SDF_SCRATCH_FLOATS_PER_WAVE;
  out[7] = nrh::WG_WAVES;
  return NRH_OK;
}

int nrh_mlp_grid(void) { return mlp_grid(); }

long long nrh_color_transposed_floats(int hints) { return nrh::colt_packed_floats(hints ? 8 : 4); }

int nrh_kernel_timing_select(int kind) {
  if (kind < -1 || kind > 3) return fail(NRH_E_INVALID, "nrh_kernel_timing_select: kind must be -1..3%s", "");
  for (auto& t : g_timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
  g_timed.clear();
  g_time_kind = kind;
  return NRH_OK;
}

int nrh_kernel_timing_read(double* total_ms, long long* launches) {
  if (!total_ms || !launches) return fail(NRH_E_INVALID, "nrh_kernel_timing_read: null%s", "");
  double tot = 0.0;
  for (auto& t : g_timed) {
    float ms = 0.f;
    if (hipEventSynchronize(t.b) != hipSuccess || hipEventElapsedTime(&ms, t.a, t.b) != hipSuccess)
      return fail(NRH_E_LAUNCH, "nrh_kernel_timing_read: event query failed%s", "");
    tot += ms;
    (void)hipEventDestroy(t.a);
    (void)hipEventDestroy(t.b);
  }
  *total_ms = tot;
  *launches = (long long)g_timed.size();
  g_timed.clear();
  return NRH_OK;
}

int nrh_sdf_eval(int precision, int mode, const float* sdf_w, const float* sdf_b, const float* sdf_head, const float* ro,
                 const float* rd, const float* t, int t_stride, int n_per_ray, long long nrays, float* sdf,
                 int sdf_stride, float* grad, float* feat, float* scratch, void* stream) {
  return sdf_eval_impl(precision, mode, sdf_w, sdf_b, sdf_head, ro, rd, t, t_stride, n_per_ray, nrays, sdf, sdf_stride, grad, feat,
                       scratch, (hipStream_t)stream);
}

int nrh_sdf_eval_wide(int mode, const void* sdf_w32, const float* sdf_tab32, const float* ro, const float* rd, const float* t,
                      int t_stride, int n_per_ray, long long nrays, float* sdf, int sdf_stride, float* grad, float* feat,
                      float* scratch, void* stream) {
  if (!sdf_w32 || !sdf_tab32) return fail(NRH_E_INVALID, "nrh_sdf_eval_wide: null pointer%s", "");
  return sdf_eval_impl(1, mode, nullptr, nullptr, nullptr, ro, rd, t, t_stride, n_per_ray, nrays, sdf, sdf_stride, grad, feat, scratch,
                       (hipStream_t)stream, WideNet{sdf_w32, sdf_tab32});
}

// nrh_sdf_eval_wide on the ONE-TERM builds of the wide kernels (precision "f16": one fp16 MFMA pass per K step; same streams / tables)
int nrh_sdf_eval_wide_f16(int mode, const void* sdf_w32, const float* sdf_tab32, const float* ro, const float* rd, const float* t,
                          int t_stride, int n_per_ray, long long nrays, float* sdf, int sdf_stride, float* grad, float* feat,
                          float* scratch, void* stream) {
  OneTermScope scope(1);
  return nrh_sdf_eval_wide(mode, sdf_w32, sdf_tab32, ro, rd, t, t_stride, n_per_ray, nrays, sdf, sdf_stride, grad, feat, scratch, stream);
}

int nrh_sdf_eval_split(const float* sdf_w, const float* sdf_b, const float* sdf_head, const float* ro, const float* rd, const float* t,
                       int t_stride, int n_per_ray, long long nrays, float* sdf, int sdf_stride, int tiles, void* stream) {
  return sdf_split_impl(sdf_w, sdf_b, sdf_head, ro, rd, t, t_stride, n_per_ray, nrays, sdf, sdf_stride, tiles, (hipStream_t)stream);
}

int nrh_sdf_grad_split(const float* sdf_w, const float* sdf_b, const float* sdf_head, const float* ro, const float* rd, const float* t,
                       int t_stride, int n_per_ray, long long nrays, float* sdf, int sdf_stride, float* grad, void* stream) {
  return sdf_grad_split_impl(sdf_w, sdf_b, sdf_head, ro, rd, t, t_stride, n_per_ray, nrays, sdf, sdf_stride, grad, (hipStream_t)stream);
}

long long nrh_sdf_wide_stream_bytes(void) { return nrh32::wide_sdf_stream_bytes_total(); }

// 16-bit hand-offs (NrhTrainSaves.save_h16 / save_t16, nrh_sdf_train_backward_half): the 8- and 4-wave f16x3 kernels
// Two further 16-bit candidates, built, measured and left OFF (profiles/r05/train_coup16_ab.log, train_t16only_ab.log): coup - the
// sweeps' private hand-off - as fp16 (-DNRH_COUP16=1: -1.07 GB per 1 024-ray step, +0.6 %) and layers 1..6 of t as fp16 only
// (-DNRH_T16_ONLY=1: -0.9 GB, +0.9 %).  Both pass the 1 024-ray parity tests, but unlike h / abar / zbar these arrays feed the adjoint
// CHAIN (11-bit roundings inside it, not at its outputs), and the sweeps are no longer bound by their bytes.  They change numerics,
// so they are BUILD-TIME variants (make variant NAME=coup16 DEFS=-DNRH_COUP16=1; reported by nrh_build_info), never a process
// environment switch.
#ifndef NRH_COUP16
#define NRH_COUP16 0
#endif
#ifndef NRH_T16_ONLY
#define NRH_T16_ONLY 0
#endif
static constexpr int coup16_mode() { return NRH_COUP16 ? 1 : 0; }
static constexpr int t16_only_mode() { return NRH_T16_ONLY ? 1 : 0; }
int nrh_train_half_supported(int precision, long long npts) {
  // (the 4-wave builds of csrc/nrh_small.hip are the same source as the 8-wave kernels; the channel-split kernels are not)
  return (precision == 1 && npts > 0 && npts % 32 == 0 && !split_train(precision, npts)) ? 1 : 0;
}

static int sdf_train_forward_impl(int precision, const float* sdf_w, const float* sdf_b, const float* sdf_head, const float* ro,
                                  const float* rd, const float* t, int t_stride, int n_per_ray, long long nrays, float* sdf,
                                  float* grad, float* feat_rows, float* save_h, float* save_s1, float* save_t, float* save_ge,
                                  void* save_h16, void* save_t16, void* stream);
int nrh_sdf_train_forward(int precision, const float* sdf_w, const float* sdf_b, const float* sdf_head, const float* ro,
                          const float* rd, const float* t, int t_stride, int n_per_ray, long long nrays, float* sdf,
                          float* grad, float* feat_rows, float* save_h, float* save_s1, float* save_t, float* save_ge,
                          void* stream) {
  return sdf_train_forward_impl(precision, sdf_w, sdf_b, sdf_head, ro, rd, t, t_stride, n_per_ray, nrays, sdf, grad, feat_rows, save_h,
                                save_s1, save_t, save_ge, nullptr, nullptr, stream);
}
int nrh_sdf_train_forward_half(int precision, const float* sdf_w, const float* sdf_b, const float* sdf_head, const float* ro,
                               const float* rd, const float* t, int t_stride, int n_per_ray, long long nrays, float* sdf,
                               float* grad, float* feat_rows, float* save_h, float* save_s1, float* save_t, float* save_ge,
                               void* save_h16, void* save_t16, void* stream) {
  if (!save_h16 || !save_t16) return fail(NRH_E_INVALID, "nrh_sdf_train_forward_half: null pointer%s", "");
  return sdf_train_forward_impl(precision, sdf_w, sdf_b, sdf_head, ro, rd, t, t_stride, n_per_ray, nrays, sdf, grad, feat_rows, save_h,
                                save_s1, save_t, save_ge, save_h16, save_t16, stream);
}
static int sdf_train_forward_impl(int precision, const float* sdf_w, const float* sdf_b, const float* sdf_head, const float* ro,
                                  const float* rd, const float* t, int t_stride, int n_per_ray, long long nrays, float* sdf,
                                  float* grad, float* feat_rows, float* save_h, float* save_s1, float* save_t, float* save_ge,
                                  void* save_h16, void* save_t16, void* stream) {
  if (precision < 0 || precision > 1) return fail(NRH_E_INVALID, "nrh_sdf_train_forward: precision must be 0 (f32) or 1 (f16x3)%s", "");
  if (!sdf_w || !sdf_b || !sdf_head || !ro || !rd || !t || !sdf || !grad || !feat_rows || !save_h || !save_s1 || !save_t || !save_ge)
    return fail(NRH_E_INVALID, "nrh_sdf_train_forward: null pointer%s", "");
  if (n_per_ray <= 0 || nrays < 0 || t_stride < n_per_ray) return fail(NRH_E_INVALID, "nrh_sdf_train_forward: bad n_per_ray/stride%s", "");
  if ((nrays * n_per_ray) % 16 != 0) return fail(NRH_E_INVALID, "nrh_sdf_train_forward: the number of points must be a multiple of 16%s", "");
  // (32-bit lane offsets inside a layer of the saved arrays, csrc/nrh_mlp.h arr_ptr: npts * 1 KiB must stay below 4 GiB)
  if (nrays * n_per_ray >= (1LL << 22)) return fail(NRH_E_INVALID, "nrh_sdf_train_forward: at most 4 194 303 points per call%s", "");
  if (nrays == 0) return NRH_OK;
  int rc = ensure_attrs();
  if (rc) return rc;
  nrh::SdfArgs a;
  memset(&a, 0, sizeof(a));
  a.w = sdf_w; a.b = sdf_b; a.head = sdf_head; a.ro = ro; a.rd = rd; a.t = t; a.sdf = sdf; a.grad = grad; a.feat = feat_rows;
  a.save_h = save_h; a.save_s1 = save_s1; a.save_t = save_t; a.save_ge = save_ge;
  a.npts = nrays * n_per_ray;
  a.n_per_ray = n_per_ray; a.t_stride = t_stride; a.sdf_stride = n_per_ray;
  if (save_h16 || save_t16) {
    if (!save_h16 || !save_t16 || !nrh_train_half_supported(precision, a.npts) || (((uintptr_t)save_h16 | (uintptr_t)save_t16) & 15))
      return fail(NRH_E_INVALID, "nrh_sdf_train_forward: 16-bit hand-offs need precision f16x3, a batch above the channel-split kernels' range "
                  "(nrh_train_half_supported) and 16-byte aligned arrays%s", "");
    a.save_h16 = save_h16; a.save_t16 = save_t16;
    a.t16_only = t16_only_mode();
  }
  const hipStream_t st = (hipStream_t)stream;
  if (split_train(precision, a.npts)) {
    hipLaunchKernelGGL(nrh::sdf_train_split_kernel, dim3((unsigned)(a.npts / 16)), dim3(256), nrh::SPLT_LDS_BYTES, st, a);
    return check_launch("sdf_train_split_kernel");
  }
  if (small_batch(a.npts)) {
    const int src = nrh4s::launch_sdf_train_forward(precision, &a, sizeof(a), device_cus() * 2, st);
    if (src) return fail(src == -1 ? NRH_E_INVALID : NRH_E_LAUNCH, "nrh_sdf_train_forward: small-batch launch failed%s", "");
    return check_launch("sdf_kernel<3> (4 waves)");
  }
  int grid = 0;
  rc = mlp_launch_geometry(a.npts, a.ntile_groups, grid, "nrh_sdf_train_forward");
  if (rc) return rc;
  if (precision == 0) hipLaunchKernelGGL((nrh::sdf_kernel<3, 0>), dim3(grid), dim3(nrh::MLP_THREADS), nrh::MLP_LDS_BYTES, st, a);
  else hipLaunchKernelGGL((nrh::sdf_kernel<3, 1>), dim3(grid), dim3(nrh::MLP_THREADS), nrh::MLP_LDS_BYTES, st, a);
  return check_launch("sdf_kernel<3>");
}

// adj_scale of the three backward entries: a positive power of two (exact to apply and to undo)
static bool adj_scale_ok(float s) {
  int e = 0;
  return s > 0.0f && s < 3.0e38f && frexpf(s, &e) == 0.5f && e >= -60 && e <= 60;
}

static int sdf_train_backward_impl(int precision, const float* sdf_w, const float* wt_feat, const float* sdf_head, const float* ro,
                                   const float* rd, const float* t, int t_stride, int n_per_ray, long long nrays,
                                   const float* save_s1, const float* save_t, const float* gbar, const float* fbar,
                                   const float* sbar, float* abar, float* coup, float* gebar, float* zbar, float* pbar,
                                   float adj_scale, void* abar16, void* zbar16, float* dyn, const void* save_t16, void* stream);
int nrh_sdf_train_backward(int precision, const float* sdf_w, const float* wt_feat, const float* sdf_head, const float* ro,
                           const float* rd, const float* t, int t_stride, int n_per_ray, long long nrays,
                           const float* save_s1, const float* save_t, const float* gbar, const float* fbar,
                           const float* sbar, float* abar, float* coup, float* gebar, float* zbar, float* pbar,
                           float adj_scale, void* stream) {
  return sdf_train_backward_impl(precision, sdf_w, wt_feat, sdf_head, ro, rd, t, t_stride, n_per_ray, nrays, save_s1, save_t, gbar, fbar,
                                 sbar, abar, coup, gebar, zbar, pbar, adj_scale, nullptr, nullptr, nullptr, nullptr, stream);
}
int nrh_sdf_train_backward_half(int precision, const float* sdf_w, const float* wt_feat, const float* sdf_head, const float* ro,
                                const float* rd, const float* t, int t_stride, int n_per_ray, long long nrays,
                                const float* save_s1, const float* save_t, const float* gbar, const float* fbar,
                                const float* sbar, float* abar, float* coup, float* gebar, float* zbar, float* pbar,
                                void* abar16, void* zbar16, float* dyn, const void* save_t16, void* stream) {
  if (!abar16 || !zbar16 || !dyn) return fail(NRH_E_INVALID, "nrh_sdf_train_backward_half: null pointer%s", "");
  return sdf_train_backward_impl(precision, sdf_w, wt_feat, sdf_head, ro, rd, t, t_stride, n_per_ray, nrays, save_s1, save_t, gbar, fbar,
                                 sbar, abar, coup, gebar, zbar, pbar, 1.0f, abar16, zbar16, dyn, save_t16, stream);
}
static int sdf_train_backward_impl(int precision, const float* sdf_w, const float* wt_feat, const float* sdf_head, const float* ro,
                                   const float* rd, const float* t, int t_stride, int n_per_ray, long long nrays,
                                   const float* save_s1, const float* save_t, const float* gbar, const float* fbar,
                                   const float* sbar, float* abar, float* coup, float* gebar, float* zbar, float* pbar,
                                   float adj_scale, void* abar16, void* zbar16, float* dyn, const void* save_t16, void* stream) {
  if (precision < 0 || precision > 1) return fail(NRH_E_INVALID, "nrh_sdf_train_backward: precision must be 0 (f32) or 1 (f16x3)%s", "");
  if (!adj_scale_ok(adj_scale)) return fail(NRH_E_INVALID, "nrh_sdf_train_backward: adj_scale must be a power of two in [2^-60, 2^60]%s", "");
  if (!sdf_w || !wt_feat || !sdf_head || !ro || !rd || !t || !save_s1 || !save_t || !gbar || !fbar || !sbar || !abar || !coup ||
      !gebar || !zbar || !pbar)
    return fail(NRH_E_INVALID, "nrh_sdf_train_backward: null pointer%s", "");
  if (n_per_ray <= 0 || nrays < 0 || t_stride < n_per_ray) return fail(NRH_E_INVALID, "nrh_sdf_train_backward: bad n_per_ray/stride%s", "");
  if ((nrays * n_per_ray) % 16 != 0) return fail(NRH_E_INVALID, "nrh_sdf_train_backward: the number of points must be a multiple of 16%s", "");
  if (nrays * n_per_ray >= (1LL << 22)) return fail(NRH_E_INVALID, "nrh_sdf_train_backward: at most 4 194 303 points per call%s", "");
  if (nrays == 0) return NRH_OK;
  int rc = ensure_attrs();
  if (rc) return rc;
  nrh::SdfTrainArgs a;
  memset(&a, 0, sizeof(a));
  a.w = sdf_w; a.wt_feat = wt_feat; a.head = sdf_head; a.ro = ro; a.rd = rd; a.t = t; a.s1 = save_s1; a.tt = save_t;
  a.gbar = gbar; a.abar = abar; a.coup = coup; a.gebar = gebar; a.fbar = fbar; a.sbar = sbar; a.zbar = zbar; a.pbar = pbar;
  a.npts = nrays * n_per_ray;
  a.n_per_ray = n_per_ray; a.t_stride = t_stride;
  a.adj_scale = precision == 1 ? adj_scale : 1.0f;
  const hipStream_t st = (hipStream_t)stream;
  if (abar16) {
    // 16-bit hand-offs: the adjoint scale follows the seeds (adjoint_range_kernel), abar / zbar leave as fp16 x S
    if (!nrh_train_half_supported(precision, a.npts) || (((uintptr_t)abar16 | (uintptr_t)zbar16 | (uintptr_t)dyn) & 15))
      return fail(NRH_E_INVALID, "nrh_sdf_train_backward_half: needs precision f16x3, a batch above the channel-split kernels' range "
                  "(nrh_train_half_supported) and 16-byte aligned arrays%s", "");
    nrh::AdjRangeArgs ra;
    ra.sbar = sbar; ra.gbar = gbar; ra.fbar = fbar; ra.dyn = dyn; ra.npts = a.npts;
    const long long want = (a.npts * 64 / 8 + 255) / 256;          // (every row of fbar is read: 1 024 blocks keep HBM busy)
    hipLaunchKernelGGL(nrh::adjoint_range_kernel, dim3((unsigned)(want < 1 ? 1 : (want > 1024 ? 1024 : want))), dim3(256), 0, st, ra);
    rc = check_launch("adjoint_range_kernel");
    if (rc) return rc;
    a.abar16 = abar16; a.zbar16 = zbar16; a.dyn = dyn;
    a.t16 = (t16_only_mode() && save_t16) ? save_t16 : nullptr;
    a.coup16 = coup16_mode();
  }
  if (split_train(precision, a.npts)) {
    const dim3 gs((unsigned)(a.npts / 16)), bs(256);
    hipLaunchKernelGGL(nrh::sdf_tangent_split_kernel, gs, bs, nrh::SPLB_LDS_BYTES, st, a);
    rc = check_launch("sdf_tangent_split_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(nrh::sdf_adjoint_split_kernel, gs, bs, nrh::SPLB_LDS_BYTES, st, a);
    return check_launch("sdf_adjoint_split_kernel");
  }
  if (small_batch(a.npts)) {
    const int src = nrh4s::launch_sdf_train_sweeps(precision, &a, sizeof(a), device_cus() * 2, st);
    if (src) return fail(src == -1 ? NRH_E_INVALID : NRH_E_LAUNCH, "nrh_sdf_train_backward: small-batch launch failed%s", "");
    return check_launch("sdf_tangent_kernel / sdf_adjoint_kernel (4 waves)");
  }
  int grid = 0;
  rc = mlp_launch_geometry(a.npts, a.ntile_groups, grid, "nrh_sdf_train_backward");
  if (rc) return rc;
  const dim3 g(grid), blk(nrh::MLP_THREADS);
  if (precision == 0) hipLaunchKernelGGL((nrh::sdf_tangent_kernel<0>), g, blk, nrh::MLP_LDS_BYTES, st, a);
  else hipLaunchKernelGGL((nrh::sdf_tangent_kernel<1>), g, blk, nrh::MLP_LDS_BYTES, st, a);
  rc = check_launch("sdf_tangent_kernel");
  if (rc) return rc;
  if (precision == 0) hipLaunchKernelGGL((nrh::sdf_adjoint_kernel<0>), g, blk, nrh::MLP_LDS_BYTES, st, a);
  else hipLaunchKernelGGL((nrh::sdf_adjoint_kernel<1>), g, blk, nrh::MLP_LDS_BYTES, st, a);
  return check_launch("sdf_adjoint_kernel");
}

static int fold_launch(bool adjoint, int nlayers, const int* rows, const int* cols, const float* const* v, const float* const* g,
                       float* const* w, const float* const* wbar, float* const* vbar, float* const* gbar, void* stream) {
  const char* who = adjoint ? "nrh_weight_norm_fold_backward" : "nrh_weight_norm_fold";
  if (nlayers <= 0 || nlayers > nrh::FOLD_MAX_LAYERS) return fail(NRH_E_INVALID, "%s: 1..16 layers", who);
  if (!rows || !cols || !v || !g) return fail(NRH_E_INVALID, "%s: null pointer", who);
  nrh::FoldArgs a;
  memset(&a, 0, sizeof(a));
  a.nlayers = nlayers;
  int total = 0;
  for (int l = 0; l < nlayers; ++l) {
    if (rows[l] <= 0 || cols[l] <= 0 || cols[l] > nrh::FOLD_MAX_COLS) return fail(NRH_E_INVALID, "%s: bad layer shape (cols <= 384)", who);
    if (!v[l] || !g[l]) return fail(NRH_E_INVALID, "%s: null layer pointer", who);
    a.v[l] = v[l]; a.g[l] = g[l]; a.cols[l] = cols[l]; a.row_start[l] = total;
    if (adjoint) {
      if (!vbar || !gbar || !vbar[l] || !gbar[l]) return fail(NRH_E_INVALID, "%s: null output pointer", who);
      a.wbar[l] = wbar ? wbar[l] : nullptr; a.vbar[l] = vbar[l]; a.gbar[l] = gbar[l];
    } else {
      if (!w || !w[l]) return fail(NRH_E_INVALID, "%s: null output pointer", who);
      a.w[l] = w[l];
    }
    total += rows[l];
  }
  a.row_start[nlayers] = total;
  const unsigned blocks = (unsigned)((total + 3) / 4);
  if (adjoint) hipLaunchKernelGGL(nrh::fold_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(nrh::fold_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch(who);
}

int nrh_weight_norm_fold(int nlayers, const int* rows, const int* cols, const float* const* v, const float* const* g,
                         float* const* w, void* stream) {
  return fold_launch(false, nlayers, rows, cols, v, g, w, nullptr, nullptr, nullptr, stream);
}

int nrh_weight_norm_fold_backward(int nlayers, const int* rows, const int* cols, const float* const* v, const float* const* g,
                                  const float* const* wbar, float* const* vbar, float* const* gbar, void* stream) {
  return fold_launch(true, nlayers, rows, cols, v, g, nullptr, wbar, vbar, gbar, stream);
}

int nrh_pack_gather(const float* flat, const int* index, const float* factor, long long n, int mode, void* out, void* stream) {
  if (!flat || !index || !out || (mode != 0 && !factor)) return fail(NRH_E_INVALID, "nrh_pack_gather: null pointer%s", "");
  if (mode < 0 || mode > 2 || n < 0) return fail(NRH_E_INVALID, "nrh_pack_gather: mode must be 0, 1 or 2%s", "");
  if (mode != 0 && (n & 511)) return fail(NRH_E_INVALID, "nrh_pack_gather: fp16 packs come in blocks of 512 elements%s", "");
  if (n == 0) return NRH_OK;
  nrh::PackGatherArgs a;
  a.flat = flat; a.index = index; a.factor = factor; a.out = out; a.n = n; a.mode = mode;
  hipLaunchKernelGGL(nrh::pack_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("pack_gather_kernel");
}

int nrh_sdf32_tables(const float* const* sdf_bias, const int* rows, const float* feat_b, const float* head_b, const float* head_w,
                     float* tables, void* stream) {
  if (!sdf_bias || !rows || !feat_b || !head_b || !head_w || !tables) return fail(NRH_E_INVALID, "nrh_sdf32_tables: null pointer%s", "");
  nrh::TablesArgs a;
  for (int l = 0; l < 8; ++l) {
    if (!sdf_bias[l] || rows[l] < 1 || rows[l] > 256) return fail(NRH_E_INVALID, "nrh_sdf32_tables: bad bias row%s", "");
    a.bias[l] = sdf_bias[l]; a.rows[l] = rows[l];
  }
  a.feat_b = feat_b; a.head_b = head_b; a.head_w = head_w; a.out = tables;
  hipLaunchKernelGGL(nrh::sdf32_tables_kernel, dim3(11), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("sdf32_tables_kernel");
}

int nrh_color_train_forward(int precision, int hints, const float* col_w, const float* col_b, const float* feat_rows,
                            const float* pts, const float* normal, const float* raymisc, long long nrays, float* color,
                            float* save_h, float* save_misc, void* stream) {
  return nrh_color_train_forward_grouped(precision, hints, col_w, col_b, feat_rows, pts, normal, raymisc, 128, nrays, color, save_h,
                                         save_misc, stream);
}

static int color_train_forward_impl(int precision, int hints, const float* col_w, const float* col_b, const float* feat_rows,
                                    const float* pts, const float* normal, const float* raymisc, int samples_per_row, long long nrays,
                                    float* color, float* save_h, float* save_misc, void* save_h16, void* stream);
int nrh_color_train_forward_grouped(int precision, int hints, const float* col_w, const float* col_b, const float* feat_rows,
                                    const float* pts, const float* normal, const float* raymisc, int samples_per_row, long long nrays,
                                    float* color, float* save_h, float* save_misc, void* stream) {
  return color_train_forward_impl(precision, hints, col_w, col_b, feat_rows, pts, normal, raymisc, samples_per_row, nrays, color, save_h,
                                  save_misc, nullptr, stream);
}
int nrh_color_train_forward_half(int precision, int hints, const float* col_w, const float* col_b, const float* feat_rows,
                                 const float* pts, const float* normal, const float* raymisc, int samples_per_row, long long nrays,
                                 float* color, float* save_h, float* save_misc, void* save_h16, void* stream) {
  if (!save_h16 || precision != 1 || ((uintptr_t)save_h16 & 15))
    return fail(NRH_E_INVALID, "nrh_color_train_forward_half: precision f16x3 and a 16-byte aligned fp16 array%s", "");
  return color_train_forward_impl(precision, hints, col_w, col_b, feat_rows, pts, normal, raymisc, samples_per_row, nrays, color, save_h,
                                  save_misc, save_h16, stream);
}
static int color_train_forward_impl(int precision, int hints, const float* col_w, const float* col_b, const float* feat_rows,
                                    const float* pts, const float* normal, const float* raymisc, int samples_per_row, long long nrays,
                                    float* color, float* save_h, float* save_misc, void* save_h16, void* stream) {
  int misc_shift = 0;
  while ((1 << misc_shift) < samples_per_row) ++misc_shift;
  if (samples_per_row < 1 || samples_per_row > 128 || (1 << misc_shift) != samples_per_row)
    return fail(NRH_E_INVALID, "nrh_color_train_forward: samples_per_row must be a power of two <= 128%s", "");
  if (precision < 0 || precision > 1) return fail(NRH_E_INVALID, "nrh_color_train_forward: precision must be 0 (f32) or 1 (f16x3)%s", "");
  if (!col_w || !col_b || !feat_rows || !pts || !normal || !raymisc || !color || !save_h || !save_misc)
    return fail(NRH_E_INVALID, "nrh_color_train_forward: null pointer%s", "");
  if (nrays < 0) return fail(NRH_E_INVALID, "nrh_color_train_forward: nrays < 0%s", "");
  if (nrays == 0) return NRH_OK;
  int rc = ensure_attrs();
  if (rc) return rc;
  nrh::ColorArgs a;
  memset(&a, 0, sizeof(a));
  a.w = col_w; a.b = col_b; a.feat = feat_rows; a.pts = pts; a.nhat = normal; a.raymisc = raymisc; a.color = color;
  a.ro = pts; a.rd = pts; a.tmid = pts;  // unused in the training instantiation
  a.save_h = save_h; a.save_misc = save_misc; a.misc_shift = misc_shift; a.save_h16 = save_h16;
  a.npts = nrays * 128;
  if (!save_h16 && split_color(precision, a.npts)) {      // small batch: one tile per workgroup, channels over its four waves
    const dim3 sg((unsigned)(a.npts / 16)), sb(256);
    if (hints) hipLaunchKernelGGL((nrh::color_train_split_kernel<8>), sg, sb, nrh::SPLC_LDS_BYTES, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((nrh::color_train_split_kernel<4>), sg, sb, nrh::SPLC_LDS_BYTES, (hipStream_t)stream, a);
    return check_launch("color_train_split_kernel");
  }
  int grid = 0;
  rc = mlp_launch_geometry(a.npts, a.ntile_groups, grid, "nrh_color_train_forward");
  if (rc) return rc;
  const hipStream_t st = (hipStream_t)stream;
  const dim3 g(grid), blk(nrh::MLP_THREADS);
  const int lds = nrh::MLP_LDS_BYTES;
  if (precision == 0 && hints) hipLaunchKernelGGL((nrh::color_kernel<0, 8, true>), g, blk, lds, st, a);
  else if (precision == 1 && hints) hipLaunchKernelGGL((nrh::color_kernel<1, 8, true>), g, blk, lds, st, a);
  else if (precision == 0) hipLaunchKernelGGL((nrh::color_kernel<0, 4, true>), g, blk, lds, st, a);
  else hipLaunchKernelGGL((nrh::color_kernel<1, 4, true>), g, blk, lds, st, a);
  return check_launch("color_kernel<train>");
}

static int color_train_backward_impl(int precision, int hints, const float* col_wt, const float* zbar4, const float* save_h,
                                     long long nrays, float* zbar, float* fbar, float* mbar, float adj_scale, const void* save_h16,
                                     void* zbar16, float half_gain, void* stream);
int nrh_color_train_backward(int precision, int hints, const float* col_wt, const float* zbar4, const float* save_h,
                             long long nrays, float* zbar, float* fbar, float* mbar, float adj_scale, void* stream) {
  return color_train_backward_impl(precision, hints, col_wt, zbar4, save_h, nrays, zbar, fbar, mbar, adj_scale, nullptr, nullptr, 1.0f, stream);
}
int nrh_color_train_backward_half(int precision, int hints, const float* col_wt, const float* zbar4, const float* save_h,
                                  long long nrays, float* zbar, float* fbar, float* mbar, float adj_scale, const void* save_h16,
                                  void* zbar16, float half_gain, void* stream) {
  if (!save_h16 || !zbar16 || precision != 1 || !adj_scale_ok(half_gain) || (((uintptr_t)save_h16 | (uintptr_t)zbar16) & 15))
    return fail(NRH_E_INVALID, "nrh_color_train_backward_half: precision f16x3, 16-byte aligned fp16 arrays, half_gain a power of two%s", "");
  return color_train_backward_impl(precision, hints, col_wt, zbar4, save_h, nrays, zbar, fbar, mbar, adj_scale, save_h16, zbar16, half_gain,
                                   stream);
}
static int color_train_backward_impl(int precision, int hints, const float* col_wt, const float* zbar4, const float* save_h,
                                     long long nrays, float* zbar, float* fbar, float* mbar, float adj_scale, const void* save_h16,
                                     void* zbar16, float half_gain, void* stream) {
  if (precision < 0 || precision > 1) return fail(NRH_E_INVALID, "nrh_color_train_backward: precision must be 0 (f32) or 1 (f16x3)%s", "");
  if (!adj_scale_ok(adj_scale)) return fail(NRH_E_INVALID, "nrh_color_train_backward: adj_scale must be a power of two in [2^-60, 2^60]%s", "");
  if (!col_wt || !zbar4 || !save_h || !zbar || !fbar || !mbar) return fail(NRH_E_INVALID, "nrh_color_train_backward: null pointer%s", "");
  if (nrays < 0) return fail(NRH_E_INVALID, "nrh_color_train_backward: nrays < 0%s", "");
  if (nrays == 0) return NRH_OK;
  int rc = ensure_attrs();
  if (rc) return rc;
  nrh::ColorAdjArgs a;
  memset(&a, 0, sizeof(a));
  a.wt = col_wt; a.zbar4 = zbar4; a.save_h = save_h; a.zbar = zbar; a.fbar = fbar; a.mbar = mbar;
  a.save_h16 = save_h16; a.zbar16 = zbar16; a.half_gain = half_gain;
  a.npts = nrays * 128;
  a.adj_scale = precision == 1 ? adj_scale : 1.0f;
  if (!save_h16 && !zbar16 && split_color(precision, a.npts)) {
    const dim3 sg((unsigned)(a.npts / 16)), sb(256);
    if (hints) hipLaunchKernelGGL((nrh::color_adjoint_split_kernel<8>), sg, sb, nrh::SPLCA_LDS_BYTES, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((nrh::color_adjoint_split_kernel<4>), sg, sb, nrh::SPLCA_LDS_BYTES, (hipStream_t)stream, a);
    return check_launch("color_adjoint_split_kernel");
  }
  int grid = 0;
  rc = mlp_launch_geometry(a.npts, a.ntile_groups, grid, "nrh_color_train_backward");
  if (rc) return rc;
  const hipStream_t st = (hipStream_t)stream;
  const dim3 g(grid), blk(nrh::MLP_THREADS);
  const int lds = nrh::MLP_LDS_BYTES;
  if (precision == 0 && hints) hipLaunchKernelGGL((nrh::color_adjoint_kernel<0, 8>), g, blk, lds, st, a);
  else if (precision == 1 && hints) hipLaunchKernelGGL((nrh::color_adjoint_kernel<1, 8>), g, blk, lds, st, a);
  else if (precision == 0) hipLaunchKernelGGL((nrh::color_adjoint_kernel<0, 4>), g, blk, lds, st, a);
  else hipLaunchKernelGGL((nrh::color_adjoint_kernel<1, 4>), g, blk, lds, st, a);
  return check_launch("color_adjoint_kernel");
}

// ---- the outside-NeRF background network (csrc/nrh_outside.hip) ----
int nrh_outside_sizes(int* out) {
  if (!out) return fail(NRH_E_INVALID, "nrh_outside_sizes: null%s", "");
  out[0] = nrh::ON_PACKED_FLOATS; out[1] = nrh::ON_BIAS_FLOATS; out[2] = nrh::ONT_PACKED_FLOATS; out[3] = nrh::ON_X; out[4] = nrh::ON_V;
  return NRH_OK;
}

static int outside_attrs() {
  const int dev = current_device();
  if (dev < 0) return fail(NRH_E_LAUNCH, "no HIP device%s", "");
  static bool done[MAX_DEVICES] = {};
  if (done[dev]) return NRH_OK;
  const void* fns[] = {(const void*)nrh::outside_kernel<0, false>, (const void*)nrh::outside_kernel<0, true>,
                       (const void*)nrh::outside_kernel<1, false>, (const void*)nrh::outside_kernel<1, true>,
                       (const void*)nrh::outside_adjoint_kernel<0>, (const void*)nrh::outside_adjoint_kernel<1>};
  for (const void* f : fns)
    if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, nrh::MLP_LDS_BYTES) != hipSuccess)
      return fail(NRH_E_LAUNCH, "hipFuncSetAttribute failed%s", "");
  done[dev] = true;
  return NRH_OK;
}

int nrh_outside_forward(int precision, const float* on_w, const float* on_b, const float* pts4, const float* views, const float* pls,
                        int pts_per_ray, long long npts, float* density, float* rgb, float* save_x, float* save_v, float* save_h,
                        float* save_f, float* save_hv, void* stream) {
  if (precision < 0 || precision > 1) return fail(NRH_E_INVALID, "nrh_outside_forward: precision must be 0 (f32) or 1 (f16x3)%s", "");
  if (!on_w || !on_b || !pts4 || !views || !pls || !density || !rgb) return fail(NRH_E_INVALID, "nrh_outside_forward: null pointer%s", "");
  const bool train = save_x || save_v || save_h || save_f || save_hv;
  if (train && (!save_x || !save_v || !save_h || !save_f || !save_hv))
    return fail(NRH_E_INVALID, "nrh_outside_forward: the training saves come all or none%s", "");
  if (pts_per_ray <= 0 || npts < 0 || npts % pts_per_ray != 0) return fail(NRH_E_INVALID, "nrh_outside_forward: npts must be a multiple of pts_per_ray%s", "");
  if (train && npts % 16 != 0) return fail(NRH_E_INVALID, "nrh_outside_forward: training needs a multiple of 16 points%s", "");
  if (npts == 0) return NRH_OK;
  int rc = outside_attrs();
  if (rc) return rc;
  nrh::OutsideArgs a;
  memset(&a, 0, sizeof(a));
  a.w = on_w; a.b = on_b; a.pts4 = pts4; a.views = views; a.pls = pls; a.density = density; a.rgb = rgb; a.npts = npts;
  a.pts_per_ray = pts_per_ray; a.save_x = save_x; a.save_v = save_v; a.save_h = save_h; a.save_f = save_f; a.save_hv = save_hv;
  int grid = 0;
  rc = mlp_launch_geometry(a.npts, a.ntile_groups, grid, "nrh_outside_forward");
  if (rc) return rc;
  const hipStream_t st = (hipStream_t)stream;
  const dim3 g(grid), blk(nrh::MLP_THREADS);
  const int lds = nrh::MLP_LDS_BYTES;
  if (precision == 0 && train) hipLaunchKernelGGL((nrh::outside_kernel<0, true>), g, blk, lds, st, a);
  else if (precision == 0) hipLaunchKernelGGL((nrh::outside_kernel<0, false>), g, blk, lds, st, a);
  else if (train) hipLaunchKernelGGL((nrh::outside_kernel<1, true>), g, blk, lds, st, a);
  else hipLaunchKernelGGL((nrh::outside_kernel<1, false>), g, blk, lds, st, a);
  return check_launch("outside_kernel");
}

int nrh_outside_backward(int precision, const float* on_wt, const float* alpha_w, const float* density_bar, const float* rgb_bar,
                         const float* save_h, const float* save_hv, long long npts, float* zbar, float* fbar, float* zvbar, float* xbar,
                         float* vbar, float adj_scale, void* stream) {
  if (precision < 0 || precision > 1) return fail(NRH_E_INVALID, "nrh_outside_backward: precision must be 0 (f32) or 1 (f16x3)%s", "");
  if (!adj_scale_ok(adj_scale)) return fail(NRH_E_INVALID, "nrh_outside_backward: adj_scale must be a power of two in [2^-60, 2^60]%s", "");
  if (!on_wt || !alpha_w || !density_bar || !rgb_bar || !save_h || !save_hv || !zbar || !fbar || !zvbar || !xbar || !vbar)
    return fail(NRH_E_INVALID, "nrh_outside_backward: null pointer%s", "");
  if (npts < 0 || npts % 16 != 0) return fail(NRH_E_INVALID, "nrh_outside_backward: the number of points must be a multiple of 16%s", "");
  if (npts == 0) return NRH_OK;
  int rc = outside_attrs();
  if (rc) return rc;
  nrh::OutsideAdjArgs a;
  memset(&a, 0, sizeof(a));
  a.wt = on_wt; a.walpha = alpha_w; a.dbar = density_bar; a.cbar = rgb_bar; a.save_h = save_h; a.save_hv = save_hv; a.zbar = zbar;
  a.fbar = fbar; a.zvbar = zvbar; a.xbar = xbar; a.vbar = vbar; a.npts = npts;
  a.adj_scale = precision == 1 ? adj_scale : 1.0f;
  int grid = 0;
  rc = mlp_launch_geometry(a.npts, a.ntile_groups, grid, "nrh_outside_backward");
  if (rc) return rc;
  const hipStream_t st = (hipStream_t)stream;
  const dim3 g(grid), blk(nrh::MLP_THREADS);
  if (precision == 0) hipLaunchKernelGGL((nrh::outside_adjoint_kernel<0>), g, blk, nrh::MLP_LDS_BYTES, st, a);
  else hipLaunchKernelGGL((nrh::outside_adjoint_kernel<1>), g, blk, nrh::MLP_LDS_BYTES, st, a);
  return check_launch("outside_adjoint_kernel");
}

int nrh_alpha_train_forward(const float* sdf, const float* grad, const float* rd, const float* dists, float inv_s,
                            float cos_anneal, const float* dyn_scalars, long long nrays, float* weights, float* nhat,
                            void* stream) {
  return nrh_alpha_train_forward_n(sdf, grad, rd, dists, inv_s, cos_anneal, dyn_scalars, nrays, 128, weights, nhat, stream);
}

int nrh_alpha_train_forward_n(const float* sdf, const float* grad, const float* rd, const float* dists, float inv_s,
                              float cos_anneal, const float* dyn_scalars, long long nrays, int n_real, float* weights, float* nhat,
                              void* stream) {
  if (n_real < 2 || n_real > 128) return fail(NRH_E_INVALID, "nrh_alpha_train_forward: n_real must lie in 2 .. 128%s", "");
  if (!sdf || !grad || !rd || !dists || !weights || !nhat) return fail(NRH_E_INVALID, "nrh_alpha_train_forward: null pointer%s", "");
  if (nrays < 0 || nrays > 0x7fffffffLL) return fail(NRH_E_INVALID, "nrh_alpha_train_forward: nrays out of range%s", "");
  if (nrays == 0) return NRH_OK;
  nrh::AlphaTrainArgs a;
  memset(&a, 0, sizeof(a));
  a.sdf = sdf; a.grad = grad; a.rd = rd; a.dists = dists; a.inv_s = inv_s; a.cos_anneal = cos_anneal; a.nrays = (int)nrays;
  a.dyn = dyn_scalars; a.nreal = n_real == 128 ? 0 : n_real;
  a.weights = weights; a.nhat = nhat;
  const unsigned blocks = (unsigned)((nrays + nrh::TRAIN_RAYS_PER_BLOCK - 1) / nrh::TRAIN_RAYS_PER_BLOCK);
  hipLaunchKernelGGL(nrh::alpha_train_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("alpha_train_kernel<fwd>");
}

// ---- renderer.use_outside_nerf: the pieces around the caller's background network -------------------------------------------
int nrh_sample_primary(const NrhNet* net, const float* origins, const float* directions, const float* nears, const float* fars,
                       long long nrays, const float* t_rand_primary, const float* lin64, const float* lin16, float* z_vals, float* mid_z,
                       float* dists, float* workspace, long long workspace_floats, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!net || !net->sdf_w || !net->sdf_b || !net->sdf_head || !origins || !directions || !nears || !fars || !lin64 || !lin16 || !z_vals ||
      !mid_z || !dists || !workspace)
    return fail(NRH_E_INVALID, "nrh_sample_primary: null pointer%s", "");
  if (nrays < 0 || nrays > (1LL << 24)) return fail(NRH_E_INVALID, "nrh_sample_primary: nrays out of range%s", "");
  if (nrays == 0) return NRH_OK;
  const long long n = nrays;
  if (workspace_floats < round64(n * 128) + 2 * round64(n * 16)) return fail(NRH_E_WORKSPACE, "nrh_sample_primary: workspace too small%s", "");
  float* sbuf = workspace;
  float* znew = sbuf + round64(n * 128);
  float* snew = znew + round64(n * 16);
  nrh::CoarseArgs c;
  c.near_ = nears; c.far_ = fars; c.lin64 = lin64; c.t_rand = t_rand_primary; c.z = z_vals; c.nrays = (int)n; c.nc = 64;
  hipLaunchKernelGGL(nrh::coarse_z_kernel, dim3((unsigned)((n * 64 + 255) / 256)), dim3(256), 0, st, c);
  int rc = check_launch("coarse_z_kernel");
  if (rc) return rc;
  return run_sampler(net, origins, directions, z_vals, sbuf, znew, snew, SamplerPlan{64, 4, 16, lin16}, nullptr, 2.0f / 64.0f, mid_z, dists, n, st);
}

int nrh_alpha_blend_forward(const float* sdf, const float* grad, const float* rd, const float* dists, const float* inside_sphere,
                            const float* bg_alpha, float inv_s, float cos_anneal, const float* dyn_scalars, long long nrays,
                            float* weights, float* nhat, float* tail_t, void* stream) {
  if (!sdf || !grad || !rd || !dists || !inside_sphere || !bg_alpha || !weights || !nhat || !tail_t)
    return fail(NRH_E_INVALID, "nrh_alpha_blend_forward: null pointer%s", "");
  if (nrays < 0 || nrays > 0x7fffffffLL) return fail(NRH_E_INVALID, "nrh_alpha_blend_forward: nrays out of range%s", "");
  if (nrays == 0) return NRH_OK;
  nrh::AlphaTrainArgs a;
  memset(&a, 0, sizeof(a));
  a.sdf = sdf; a.grad = grad; a.rd = rd; a.dists = dists; a.inv_s = inv_s; a.cos_anneal = cos_anneal; a.nrays = (int)nrays;
  a.dyn = dyn_scalars; a.inside = inside_sphere; a.bg_alpha = bg_alpha; a.weights = weights; a.nhat = nhat; a.tail_t = tail_t;
  const unsigned blocks = (unsigned)((nrays + nrh::TRAIN_RAYS_PER_BLOCK - 1) / nrh::TRAIN_RAYS_PER_BLOCK);
  hipLaunchKernelGGL(nrh::alpha_train_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("alpha_train_kernel<fwd, blend>");
}

int nrh_alpha_blend_backward(const float* sdf, const float* grad, const float* rd, const float* dists, const float* inside_sphere,
                             const float* bg_alpha, float inv_s, float cos_anneal, const float* dyn_scalars, long long nrays,
                             const float* weights_bar, const float* nhat_bar, const float* tail_t_bar, float* sdf_bar, float* grad_bar,
                             float* rd_bar, float* invs_bar, float* bg_alpha_bar, void* stream) {
  if (!sdf || !grad || !rd || !dists || !inside_sphere || !bg_alpha || !weights_bar || !tail_t_bar || !sdf_bar || !grad_bar || !rd_bar ||
      !invs_bar || !bg_alpha_bar)
    return fail(NRH_E_INVALID, "nrh_alpha_blend_backward: null pointer%s", "");
  if (nrays < 0 || nrays > 0x7fffffffLL) return fail(NRH_E_INVALID, "nrh_alpha_blend_backward: nrays out of range%s", "");
  if (nrays == 0) return NRH_OK;
  nrh::AlphaTrainArgs a;
  memset(&a, 0, sizeof(a));
  a.sdf = sdf; a.grad = grad; a.rd = rd; a.dists = dists; a.inv_s = inv_s; a.cos_anneal = cos_anneal; a.nrays = (int)nrays;
  a.dyn = dyn_scalars; a.inside = inside_sphere; a.bg_alpha = bg_alpha; a.weights_bar = weights_bar; a.nhat_bar = nhat_bar;
  a.tail_t_bar = tail_t_bar; a.sdf_bar = sdf_bar; a.grad_bar = grad_bar; a.rd_bar = rd_bar; a.invs_bar = invs_bar;
  a.bg_alpha_bar = bg_alpha_bar;
  const unsigned blocks = (unsigned)((nrays + nrh::TRAIN_RAYS_PER_BLOCK - 1) / nrh::TRAIN_RAYS_PER_BLOCK);
  hipLaunchKernelGGL(nrh::alpha_train_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("alpha_train_kernel<adjoint, blend>");
}

// the shadow ray's alpha stage for renderer.shadow_hint_gradient: visibility = transmittance in front of the last sample
int nrh_shadow_alpha_forward(const float* sdf, const float* grad, const float* shadow_dirs, const float* dists, float inv_s,
                             float cos_anneal, const float* dyn_scalars, long long nrays, int n_real, float* visibilities, void* stream) {
  if (!sdf || !grad || !shadow_dirs || !dists || !visibilities) return fail(NRH_E_INVALID, "nrh_shadow_alpha_forward: null pointer%s", "");
  if (nrays < 0 || nrays > 0x7fffffffLL) return fail(NRH_E_INVALID, "nrh_shadow_alpha_forward: nrays out of range%s", "");
  if (n_real < 2 || n_real > 128) return fail(NRH_E_INVALID, "nrh_shadow_alpha_forward: n_real must be 2..128%s", "");
  if (nrays == 0) return NRH_OK;
  nrh::AlphaTrainArgs a;
  memset(&a, 0, sizeof(a));
  a.sdf = sdf; a.grad = grad; a.rd = shadow_dirs; a.dists = dists; a.inv_s = inv_s; a.cos_anneal = cos_anneal; a.nrays = (int)nrays;
  a.dyn = dyn_scalars; a.tlast = visibilities; a.nreal = n_real == 128 ? 0 : n_real;
  const unsigned blocks = (unsigned)((nrays + nrh::TRAIN_RAYS_PER_BLOCK - 1) / nrh::TRAIN_RAYS_PER_BLOCK);
  hipLaunchKernelGGL(nrh::alpha_train_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("alpha_train_kernel<fwd, shadow>");
}

int nrh_shadow_alpha_backward(const float* sdf, const float* grad, const float* shadow_dirs, const float* dists, float inv_s,
                              float cos_anneal, const float* dyn_scalars, long long nrays, int n_real, const float* visibilities_bar,
                              float* sdf_bar, float* grad_bar, float* dirs_bar, float* invs_bar, void* stream) {
  if (!sdf || !grad || !shadow_dirs || !dists || !visibilities_bar || !sdf_bar || !grad_bar || !dirs_bar || !invs_bar)
    return fail(NRH_E_INVALID, "nrh_shadow_alpha_backward: null pointer%s", "");
  if (nrays < 0 || nrays > 0x7fffffffLL) return fail(NRH_E_INVALID, "nrh_shadow_alpha_backward: nrays out of range%s", "");
  if (n_real < 2 || n_real > 128) return fail(NRH_E_INVALID, "nrh_shadow_alpha_backward: n_real must be 2..128%s", "");
  if (nrays == 0) return NRH_OK;
  nrh::AlphaTrainArgs a;
  memset(&a, 0, sizeof(a));
  a.sdf = sdf; a.grad = grad; a.rd = shadow_dirs; a.dists = dists; a.inv_s = inv_s; a.cos_anneal = cos_anneal; a.nrays = (int)nrays;
  a.dyn = dyn_scalars; a.tlast_bar = visibilities_bar; a.nreal = n_real == 128 ? 0 : n_real;
  a.sdf_bar = sdf_bar; a.grad_bar = grad_bar; a.rd_bar = dirs_bar; a.invs_bar = invs_bar;
  const unsigned blocks = (unsigned)((nrays + nrh::TRAIN_RAYS_PER_BLOCK - 1) / nrh::TRAIN_RAYS_PER_BLOCK);
  hipLaunchKernelGGL(nrh::alpha_train_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("alpha_train_kernel<adjoint, shadow>");
}

int nrh_alpha_train_backward(const float* sdf, const float* grad, const float* rd, const float* dists, float inv_s,
                             float cos_anneal, const float* dyn_scalars, long long nrays, const float* weights_bar, const float* nhat_bar,
                             float* sdf_bar, float* grad_bar, float* rd_bar, float* invs_bar, void* stream) {
  return nrh_alpha_train_backward_fused(sdf, grad, rd, dists, inv_s, cos_anneal, dyn_scalars, nrays, weights_bar, nhat_bar, 3, nullptr,
                                        nullptr, sdf_bar, grad_bar, rd_bar, invs_bar, 128, stream);
}

static int alpha_backward_impl(const float* sdf, const float* grad, const float* rd, const float* dists, float inv_s,
                               float cos_anneal, const float* dyn_scalars, long long nrays, const float* weights_bar,
                               const float* nhat_bar, int nhat_bar_stride, const float* inside_sphere, const float* eikonal_coef,
                               float* sdf_bar, float* grad_bar, float* rd_bar, float* invs_bar, int n_real, void* stream);

int nrh_alpha_train_backward_n(const float* sdf, const float* grad, const float* rd, const float* dists, float inv_s,
                               float cos_anneal, const float* dyn_scalars, long long nrays, int n_real, const float* weights_bar,
                               const float* nhat_bar, float* sdf_bar, float* grad_bar, float* rd_bar, float* invs_bar, void* stream) {
  if (n_real < 2 || n_real > 128) return fail(NRH_E_INVALID, "nrh_alpha_train_backward: n_real must lie in 2 .. 128%s", "");
  return alpha_backward_impl(sdf, grad, rd, dists, inv_s, cos_anneal, dyn_scalars, nrays, weights_bar, nhat_bar, 3, nullptr, nullptr,
                             sdf_bar, grad_bar, rd_bar, invs_bar, n_real, stream);
}

int nrh_alpha_train_backward_fused(const float* sdf, const float* grad, const float* rd, const float* dists, float inv_s,
                                   float cos_anneal, const float* dyn_scalars, long long nrays, const float* weights_bar,
                                   const float* nhat_bar, int nhat_bar_stride, const float* inside_sphere, const float* eikonal_coef,
                                   float* sdf_bar, float* grad_bar, float* rd_bar, float* invs_bar, int n_real, void* stream) {
  if (n_real < 2 || n_real > 128) return fail(NRH_E_INVALID, "nrh_alpha_train_backward_fused: n_real must lie in 2 .. 128%s", "");
  return alpha_backward_impl(sdf, grad, rd, dists, inv_s, cos_anneal, dyn_scalars, nrays, weights_bar, nhat_bar, nhat_bar_stride,
                             inside_sphere, eikonal_coef, sdf_bar, grad_bar, rd_bar, invs_bar, n_real, stream);
}

static int alpha_backward_impl(const float* sdf, const float* grad, const float* rd, const float* dists, float inv_s,
                               float cos_anneal, const float* dyn_scalars, long long nrays, const float* weights_bar,
                               const float* nhat_bar, int nhat_bar_stride, const float* inside_sphere, const float* eikonal_coef,
                               float* sdf_bar, float* grad_bar, float* rd_bar, float* invs_bar, int n_real, void* stream) {
  if (!sdf || !grad || !rd || !dists || !weights_bar || !sdf_bar || !grad_bar || !rd_bar || !invs_bar)
    return fail(NRH_E_INVALID, "nrh_alpha_train_backward: null pointer%s", "");
  if (nrays < 0 || nrays > 0x7fffffffLL) return fail(NRH_E_INVALID, "nrh_alpha_train_backward: nrays out of range%s", "");
  if (nhat_bar && nhat_bar_stride < 3) return fail(NRH_E_INVALID, "nrh_alpha_train_backward_fused: nhat_bar_stride must be >= 3%s", "");
  if ((eikonal_coef != nullptr) != (inside_sphere != nullptr))
    return fail(NRH_E_INVALID, "nrh_alpha_train_backward_fused: inside_sphere and eikonal_coef come together%s", "");
  if (nrays == 0) return NRH_OK;
  nrh::AlphaTrainArgs a;
  memset(&a, 0, sizeof(a));
  a.sdf = sdf; a.grad = grad; a.rd = rd; a.dists = dists; a.inv_s = inv_s; a.cos_anneal = cos_anneal; a.nrays = (int)nrays;
  a.dyn = dyn_scalars;
  a.weights_bar = weights_bar; a.nhat_bar = nhat_bar; a.sdf_bar = sdf_bar; a.grad_bar = grad_bar; a.rd_bar = rd_bar;
  a.invs_bar = invs_bar; a.nbar_stride = nhat_bar_stride; a.inside = inside_sphere; a.eik_coef = eikonal_coef;
  a.nreal = n_real == 128 ? 0 : n_real;
  const unsigned blocks = (unsigned)((nrays + nrh::TRAIN_RAYS_PER_BLOCK - 1) / nrh::TRAIN_RAYS_PER_BLOCK);
  hipLaunchKernelGGL(nrh::alpha_train_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("alpha_train_kernel<bwd>");
}

// ---- weight gradients (csrc/nrh_dw.hip) ----
static int dw_total_slabs(const NrhDwJob* jobs, int njobs) {
  int tot = 0;
  for (int j = 0; j < njobs; ++j) tot += jobs[j].slabs;
  return tot;
}

long long nrh_dw_workspace_floats(const NrhDwJob* jobs, int njobs) {
  if (!jobs || njobs <= 0 || njobs > nrhdw::MAX_JOBS) return -1;
  for (int j = 0; j < njobs; ++j)
    if (jobs[j].slabs <= 0) return -1;
  return (long long)dw_total_slabs(jobs, njobs) * (nrhdw::SLOT_FLOATS + nrhdw::CSUM_FLOATS);
}

int nrh_dw_gemm(const NrhDwJob* jobs, int njobs, long long npts, float* workspace, long long workspace_floats, void* stream) {
  if (!jobs || !workspace) return fail(NRH_E_INVALID, "nrh_dw_gemm: null pointer%s", "");
  if (njobs <= 0 || njobs > nrhdw::MAX_JOBS) return fail(NRH_E_INVALID, "nrh_dw_gemm: 1..24 jobs%s", "");
  if (npts <= 0 || npts % nrhdw::KPTS != 0 || npts / nrhdw::KPTS > 0x7fffffffLL)
    return fail(NRH_E_INVALID, "nrh_dw_gemm: the number of points must be a positive multiple of 32%s", "");
  if (((uintptr_t)workspace & 15) != 0) return fail(NRH_E_INVALID, "nrh_dw_gemm: workspace must be 16-byte aligned%s", "");
  const long long need = nrh_dw_workspace_floats(jobs, njobs);
  if (need < 0 || workspace_floats < need) return fail(NRH_E_WORKSPACE, "nrh_dw_gemm: workspace too small%s (need %lld floats)", "", need);
  nrhdw::DwArgs a;
  nrhdw::ReduceArgs r;
  memset(&a, 0, sizeof(a));
  memset(&r, 0, sizeof(r));
  const int nsteps = (int)(npts / nrhdw::KPTS);
  int slab0 = 0;
  for (int j = 0; j < njobs; ++j) {
    const NrhDwJob& q = jobs[j];
    if (q.npairs < 1 || q.npairs > 2 || q.m < 1 || q.m > 256 || q.n < 1 || q.n > 256 || q.slabs < 1 || q.slabs > nsteps)
      return fail(NRH_E_INVALID, "nrh_dw_gemm: bad job shape%s (job %lld)", "", (long long)j);
    for (int k = 0; k < q.npairs; ++k) {
      if (!q.a[k] || !q.b[k] || q.lda[k] < q.m || q.ldb[k] < q.n) return fail(NRH_E_INVALID, "nrh_dw_gemm: bad operand%s (job %lld)", "", (long long)j);
      a.job[j].a[k] = q.a[k]; a.job[j].b[k] = q.b[k]; a.job[j].lda[k] = q.lda[k]; a.job[j].ldb[k] = q.ldb[k];
      // tiled operands (the layout of h, t, abar, zbar: csrc/nrh_mlp.h): 256-channel arrays only
      if ((q.tiled_a[k] && (q.lda[k] != 256 || q.m != 256)) || (q.tiled_b[k] && (q.ldb[k] != 256 || q.n != 256)))
        return fail(NRH_E_INVALID, "nrh_dw_gemm: a tiled operand has 256 channels%s (job %lld)", "", (long long)j);
      a.job[j].ta[k] = q.tiled_a[k] ? 1 : 0; a.job[j].tb[k] = q.tiled_b[k] ? 1 : 0;
      if (q.half_ops && (q.lda[k] != 256 || q.ldb[k] != 256 || q.m != 256 || q.n != 256 || (((uintptr_t)q.a[k] | (uintptr_t)q.b[k]) & 15)))
        return fail(NRH_E_INVALID, "nrh_dw_gemm: half operands are full 256-channel arrays, 16-byte aligned%s (job %lld)", "", (long long)j);
    }
    if (q.half_ops && (q.colsum_b || npts % 32 != 0 || q.slabs > nsteps / 2))
      return fail(NRH_E_INVALID, "nrh_dw_gemm: a half-operand job has no column sums of B, 32-point stages%s (job %lld)", "", (long long)j);
    a.job[j].half = q.half_ops ? 1 : 0;
    if (q.out && (q.rows < 1 || q.rows > q.m || q.cols < 1 || q.cols > q.n || q.ldo < 1))
      return fail(NRH_E_INVALID, "nrh_dw_gemm: bad output shape%s (job %lld)", "", (long long)j);
    if (q.col_map && q.transpose) return fail(NRH_E_INVALID, "nrh_dw_gemm: col_map with transpose%s", "");
    a.job[j].npairs = q.npairs; a.job[j].m = q.m; a.job[j].n = q.n; a.job[j].slab0 = slab0; a.job[j].slabs = q.slabs;
    a.job[j].colsum = (q.colsum_a ? 1 : 0) | (q.colsum_b ? 2 : 0);
    nrhdw::OutDev& o = r.job[j];
    o.out = q.out; o.col_map = q.col_map; o.colsum_a = q.colsum_a; o.colsum_b = q.colsum_b; o.ldo = q.ldo; o.transpose = q.transpose;
    o.rows = q.out ? q.rows : q.m; o.cols = q.out ? q.cols : q.n; o.scale = q.scale; o.scale_a = q.scale_a; o.scale_b = q.scale_b;
    o.slab0 = slab0; o.slabs = q.slabs; o.dyn = q.dyn_scale;
    slab0 += q.slabs;
  }
  a.njobs = njobs; a.nsteps = nsteps;
  a.partial = workspace;
  a.csum = workspace + (size_t)slab0 * nrhdw::SLOT_FLOATS;
  r.njobs = njobs; r.partial = a.partial; r.csum = a.csum;
  const int dev = current_device();
  if (dev < 0) return fail(NRH_E_LAUNCH, "no HIP device%s", "");
  static bool attr_done[MAX_DEVICES] = {};
  if (!attr_done[dev]) {
    if (hipFuncSetAttribute((const void*)nrhdw::dw_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, nrhdw::FAST_LDS_BYTES) != hipSuccess)
      return fail(NRH_E_LAUNCH, "hipFuncSetAttribute failed%s", "");
    attr_done[dev] = true;
  }
  const hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(nrhdw::dw_kernel, dim3(slab0), dim3(nrhdw::THREADS), nrhdw::FAST_LDS_BYTES, st, a);
  int rc = check_launch("dw_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(nrhdw::dw_reduce_kernel, dim3(256, njobs), dim3(256), 0, st, r);
  return check_launch("dw_reduce_kernel");
}

#ifdef NRH_DW_TIMING
int nrh_dw_debug_read(unsigned long long* out8, int reset) {
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(nrhdw::g_dw_cycles), 64) != hipSuccess) return NRH_E_LAUNCH;
  if (reset) { unsigned long long z[8] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(nrhdw::g_dw_cycles), z, 64); }
  return NRH_OK;
}
#endif

int nrh_embedding_rows(const float* ro, const float* rd, const float* t, int t_stride, int n_per_ray, long long nrays, float* rows,
                       void* stream) {
  if (!ro || !rd || !t || !rows) return fail(NRH_E_INVALID, "nrh_embedding_rows: null pointer%s", "");
  if (n_per_ray <= 0 || nrays < 0 || t_stride < n_per_ray) return fail(NRH_E_INVALID, "nrh_embedding_rows: bad n_per_ray/stride%s", "");
  if (nrays == 0) return NRH_OK;
  nrh::EmbRowsArgs a;
  a.ro = ro; a.rd = rd; a.t = t; a.out = rows; a.npts = nrays * n_per_ray; a.n_per_ray = n_per_ray; a.t_stride = t_stride;
  const long long blocks = (a.npts * 64 + 255) / 256;
  if (blocks > 0x7fffffffLL) return fail(NRH_E_INVALID, "nrh_embedding_rows: too many points%s", "");
  hipLaunchKernelGGL(nrh::emb_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("emb_rows_kernel");
}

int nrh_composite_loss(const float* sampled_color, const float* weights, const float* rgb_gt, const float* background,
                       const float* analytic_normals, const float* inside_sphere, long long nrays, float* rgb, float* zbar_out,
                       float* weights_bar, float* partials, void* stream) {
  if (!sampled_color || !weights || !rgb_gt || !analytic_normals || !inside_sphere || !rgb || !zbar_out || !weights_bar || !partials)
    return fail(NRH_E_INVALID, "nrh_composite_loss: null pointer%s", "");
  if (nrays < 0 || nrays > 0x7fffffffLL) return fail(NRH_E_INVALID, "nrh_composite_loss: nrays out of range%s", "");
  if (nrays == 0) return NRH_OK;
  nrh::CompositeLossArgs a;
  a.color = sampled_color; a.weights = weights; a.gt = rgb_gt; a.bg = background; a.grad = analytic_normals; a.inside = inside_sphere;
  a.rgb = rgb; a.zbar4 = zbar_out; a.wbar = weights_bar; a.partial = partials; a.inv_n = (float)(1.0 / ((double)nrays + 1e-5));
  a.nrays = (int)nrays;
  hipLaunchKernelGGL(nrh::composite_loss_kernel, dim3((unsigned)((nrays + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("composite_loss_kernel");
}

int nrh_ray_adjoint(const float* origins, const float* directions, const float* pl_positions, const float* mid_z, const float* pbar,
                    const float* gbar, const float* save_ge, const float* mbar, int mbar_width, const float* rd_bar, long long nrays,
                    float* origins_bar, float* directions_bar, float* pl_bar, void* stream) {
  if (!origins || !directions || !pl_positions || !mid_z || !pbar || !gbar || !save_ge || !mbar || !rd_bar || !origins_bar ||
      !directions_bar || !pl_bar)
    return fail(NRH_E_INVALID, "nrh_ray_adjoint: null pointer%s", "");
  if (mbar_width < 60) return fail(NRH_E_INVALID, "nrh_ray_adjoint: mbar rows hold at least the 60 columns [p, n, enc4(view), enc4(light)]%s", "");
  if (nrays < 0 || nrays > 0x7fffffffLL) return fail(NRH_E_INVALID, "nrh_ray_adjoint: nrays out of range%s", "");
  if (nrays == 0) return NRH_OK;
  nrh::RayAdjArgs a;
  a.ro = origins; a.rd = directions; a.pl = pl_positions; a.mid = mid_z; a.pbar = pbar; a.gbar = gbar; a.ge = save_ge; a.mbar = mbar;
  a.rd_bar = rd_bar; a.obar = origins_bar; a.dbar = directions_bar; a.plbar = pl_bar; a.mw = mbar_width; a.nrays = (int)nrays;
  hipLaunchKernelGGL(nrh::ray_adjoint_kernel, dim3((unsigned)((nrays + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("ray_adjoint_kernel");
}

int nrh_loss_finish(const float* partials, long long nrays, float inv_s, const float* dyn_scalars, float igr_weight, float* out8,
                    void* stream) {
  if (!partials || !out8) return fail(NRH_E_INVALID, "nrh_loss_finish: null pointer%s", "");
  if (nrays <= 0 || nrays > 0x7fffffffLL) return fail(NRH_E_INVALID, "nrh_loss_finish: nrays out of range%s", "");
  nrh::LossFinishArgs a;
  a.partial = partials; a.out = out8; a.dyn = dyn_scalars; a.inv_s = inv_s; a.igr_weight = igr_weight; a.nrays = (int)nrays;
  hipLaunchKernelGGL(nrh::loss_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("loss_finish_kernel");
}

int nrh_variance_grad(const float* invs_bar, long long nrays, float inv_s, const float* dyn_scalars, float* variance_bar, void* stream) {
  if (!invs_bar || !variance_bar) return fail(NRH_E_INVALID, "nrh_variance_grad: null pointer%s", "");
  if (nrays <= 0 || nrays > 0x7fffffffLL) return fail(NRH_E_INVALID, "nrh_variance_grad: nrays out of range%s", "");
  nrh::VarGradArgs a;
  a.invs_bar = invs_bar; a.dyn = dyn_scalars; a.inv_s = inv_s; a.out = variance_bar; a.nrays = (int)nrays;
  hipLaunchKernelGGL(nrh::variance_grad_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("variance_grad_kernel");
}

int nrh_sampler_fusion(int on) {
  const int was = g_sampler_fusion;
  if (on == 0 || on == 1 || on == 2) g_sampler_fusion = on;
  return was;
}

int nrh_step_scalars(float* const* dst, const float* values, int n, const float* variance, float* inv_s_out, void* stream) {
  if (n < 0 || n > 4 || (n > 0 && (!dst || !values))) return fail(NRH_E_INVALID, "nrh_step_scalars: 0..4 (address, value) pairs%s", "");
  if ((variance != nullptr) != (inv_s_out != nullptr)) return fail(NRH_E_INVALID, "nrh_step_scalars: variance and inv_s_out come together%s", "");
  if (n == 0 && !variance) return NRH_OK;
  nrh::StepScalarsArgs a;
  memset(&a, 0, sizeof(a));
  for (int i = 0; i < n; ++i) { a.dst[i] = dst[i]; a.val[i] = values[i]; }
  a.n = n; a.variance = variance; a.inv_s_out = inv_s_out;
  hipLaunchKernelGGL(nrh::step_scalars_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
  return check_launch("step_scalars_kernel");
}

int nrh_sampler_step(const float* ro, const float* rd, float* z, float* s, const float* znew_in,
                     const float* snew_in, float* znew_out, const float* lin16, const float* last_dist_ray,
                     float* tmid, float* dists, float inv_s, float last_dist, int nrays, int n, int do_merge,
                     int merge_sdf, int do_upsample, int do_finalize, void* stream) {
  if (!ro || !rd || !z || !s) return fail(NRH_E_INVALID, "nrh_sampler_step: null pointer%s", "");
  if (do_merge && (!znew_in || (merge_sdf && !snew_in))) return fail(NRH_E_INVALID, "nrh_sampler_step: merge needs znew_in/snew_in%s", "");
  if (do_upsample && (!znew_out || !lin16)) return fail(NRH_E_INVALID, "nrh_sampler_step: upsample needs znew_out/lin16%s", "");
  if (do_finalize && (!tmid || !dists)) return fail(NRH_E_INVALID, "nrh_sampler_step: finalize needs tmid/dists%s", "");
  const int n_after = n + (do_merge ? 16 : 0);
  if (n < 2 || n_after > 128 || (do_finalize && n_after != 128))
    return fail(NRH_E_INVALID, "nrh_sampler_step: bad sample count%s", "");
  if (nrays <= 0) return nrays == 0 ? NRH_OK : fail(NRH_E_INVALID, "nrh_sampler_step: nrays < 0%s", "");
  nrh::StepArgs a;
  memset(&a, 0, sizeof(a));
  a.ro = ro; a.rd = rd; a.z = z; a.s = s; a.znew_in = znew_in; a.snew_in = snew_in; a.znew_out = znew_out;
  a.lin16 = lin16; a.last_dist_ray = last_dist_ray; a.tmid = tmid; a.dists = dists; a.inv_s = inv_s;
  a.last_dist = last_dist; a.nrays = nrays; a.n = n; a.do_merge = do_merge; a.merge_sdf = merge_sdf;
  a.do_upsample = do_upsample; a.do_finalize = do_finalize; a.n_new = 16;
  return sampler_step_impl(a, (hipStream_t)stream);
}

int nrh_color_eval(int precision, int hints, const float* col_w, const float* col_b, const float* feat, const float* ro, const float* rd,
                   const float* tmid, const float* nhat, const float* raymisc, long long nrays, float* color,
                   void* stream) {
  return color_eval_impl(precision, hints, col_w, col_b, feat, ro, rd, tmid, nhat, raymisc, nrays, color, (hipStream_t)stream);
}

// workspace carve-up (floats per ray, each array rounded up to a multiple of 64 floats)
#define NRH_WS_FIELDS(X)                                                                                             \
  X(zbuf, 128) X(sbuf, 128) X(znew, 16) X(snew, 16) X(tmid, 128) X(dists, 128) X(sdf_c, 128) X(grad_c, 384)          \
  X(feat, 32768) X(weights, 128) X(inside, 128) X(nhat, 384) X(depth, 1) X(wsum, 1) X(cue, 4) X(srd, 3) X(slast, 1)  \
  X(tmid_s, 128) X(dists_s, 128) X(sdf_s, 128) X(grad_s, 384) X(vis, 1) X(raymisc, nrh::RAYMISC_STRIDE)              \
  X(color, 384) X(cue_b, 512) X(raymisc_g, nrh::MAX_SHADOW_CLIP * nrh::RAYMISC_STRIDE)

int nrh_generate_rays(const float* pose /* host, 12 */, const float* pl /* host, 3 */, float cx, float cy, float fx, float fy,
                      int width, int row0, int nrows, float* origins, float* directions, float* pl_positions,
                      float* nears, float* fars, void* stream) {
  if (!pose || !pl || !origins || !directions || !pl_positions || !nears || !fars)
    return fail(NRH_E_INVALID, "nrh_generate_rays: null pointer%s", "");
  if (width <= 0 || nrows < 0 || row0 < 0 || fx == 0.0f || fy == 0.0f) return fail(NRH_E_INVALID, "nrh_generate_rays: bad geometry%s", "");
  if (nrows == 0) return NRH_OK;
  nrh::RayGenArgs a;
  for (int i = 0; i < 12; ++i) a.pose[i] = pose[i];
  for (int i = 0; i < 3; ++i) a.pl[i] = pl[i];
  a.cx = cx; a.cy = cy; a.fx = fx; a.fy = fy; a.width = width; a.row0 = row0; a.nrows = nrows;
  a.origins = origins; a.dirs = directions; a.pls = pl_positions; a.nears = nears; a.fars = fars;
  const long long tot = (long long)nrows * width;
  hipLaunchKernelGGL(nrh::raygen_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("raygen_kernel");
}

static int raygen_indexed_args(nrh::RayGenIdxArgs& a, const char* who, const long long* img_indices, const float* h_indices,
                               const float* w_indices, const float* poses, int pose_stride, const float* pls, long long nrays,
                               const float* delta, const float* pl_delta, int ncam, float cx, float cy, float fx, float fy,
                               int near_far_from_sphere, float zn, float zf) {
  if (nrays < 0 || ncam < 0) return fail(NRH_E_INVALID, "%s: negative size", who);
  if (nrays == 0) return NRH_OK;
  if (!h_indices || !w_indices || !poses || !pls) return fail(NRH_E_INVALID, "%s: null pointer", who);
  if (pose_stride < 12 || fx == 0.0f || fy == 0.0f) return fail(NRH_E_INVALID, "%s: bad geometry (pose_stride >= 12, fx, fy != 0)", who);
  if ((delta || pl_delta) && (!img_indices || ncam == 0)) return fail(NRH_E_INVALID, "%s: per-view deltas need img_indices and ncam > 0", who);
  a = nrh::RayGenIdxArgs{};
  a.img = img_indices; a.hidx = h_indices; a.widx = w_indices; a.poses = poses; a.pls = pls; a.delta = delta; a.pl_delta = pl_delta;
  a.pose_stride = pose_stride; a.ncam = ncam; a.sphere = near_far_from_sphere ? 1 : 0;
  a.cx = cx; a.cy = cy; a.fx = fx; a.fy = fy; a.zn = zn; a.zf = zf; a.n = nrays;
  return 1;   // "go on"
}

int nrh_generate_rays_indexed(const long long* img_indices, const float* h_indices, const float* w_indices, const float* poses,
                              int pose_stride, const float* pls, long long nrays, const float* delta, const float* pl_delta, int ncam,
                              float cx, float cy, float fx, float fy, int near_far_from_sphere, float zn, float zf, float* origins,
                              float* directions, float* pl_positions, float* nears, float* fars, void* stream) {
  nrh::RayGenIdxArgs a;
  const int rc = raygen_indexed_args(a, "nrh_generate_rays_indexed", img_indices, h_indices, w_indices, poses, pose_stride, pls, nrays,
                                     delta, pl_delta, ncam, cx, cy, fx, fy, near_far_from_sphere, zn, zf);
  if (rc <= 0) return rc;
  if (!origins || !directions || !pl_positions || !nears || !fars) return fail(NRH_E_INVALID, "nrh_generate_rays_indexed: null output%s", "");
  a.origins = origins; a.dirs = directions; a.pl_out = pl_positions; a.nears = nears; a.fars = fars;
  hipLaunchKernelGGL(nrh::raygen_indexed_kernel, dim3((unsigned)((nrays + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("raygen_indexed_kernel");
}

int nrh_generate_rays_indexed_backward(const long long* img_indices, const float* h_indices, const float* w_indices, const float* poses,
                                       int pose_stride, const float* pls, long long nrays, const float* delta, const float* pl_delta,
                                       int ncam, float cx, float cy, float fx, float fy, int near_far_from_sphere,
                                       const float* g_origins, const float* g_directions, const float* g_pl_positions,
                                       const float* g_nears, const float* g_fars, float* g_delta, float* g_pl_delta, void* stream) {
  nrh::RayGenIdxArgs a;
  const int rc = raygen_indexed_args(a, "nrh_generate_rays_indexed_backward", img_indices, h_indices, w_indices, poses, pose_stride, pls,
                                     nrays, delta, pl_delta, ncam, cx, cy, fx, fy, near_far_from_sphere, 0.0f, 0.0f);
  if (rc <= 0) return rc;
  if (!img_indices || ncam == 0) return fail(NRH_E_INVALID, "nrh_generate_rays_indexed_backward: needs img_indices and ncam > 0%s", "");
  if (!g_delta && !g_pl_delta) return NRH_OK;
  a.g_o = g_origins; a.g_d = g_directions; a.g_pl = g_pl_positions; a.g_near = g_nears; a.g_far = g_fars;
  a.g_delta = g_delta; a.g_pl_delta = g_pl_delta;    // accumulated INTO (caller zeroes)
  hipLaunchKernelGGL(nrh::raygen_indexed_adjoint_kernel, dim3((unsigned)((nrays + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
  return check_launch("raygen_indexed_adjoint_kernel");
}

long long nrh_color_wide_stream_bytes(void) { return nrh32::wide_color_stream_bytes(); }

int nrh_color_eval_wide(const void* col_w32, const float* col_tab32, const float* part_tiles, const float* ro, const float* rd,
                        const float* tmid, const float* nhat, const float* raymisc, long long nrays, float* color, void* stream) {
  if (!col_w32 || !col_tab32 || !part_tiles || !ro || !rd || !tmid || !nhat || !raymisc || !color)
    return fail(NRH_E_INVALID, "nrh_color_eval_wide: null pointer%s", "");
  if (nrays < 0) return fail(NRH_E_INVALID, "nrh_color_eval_wide: negative size%s", "");
  if (nrays == 0) return NRH_OK;
  nrh32::WideColorCall c;
  c.stream = col_w32; c.tables = col_tab32; c.part = part_tiles; c.ro = ro; c.rd = rd; c.tmid = tmid; c.nhat = nhat;
  c.raymisc = raymisc; c.color = color; c.nrays = nrays; c.raymisc_stride = nrh::RAYMISC_STRIDE; c.max_grid = device_cus();
  const int wrc = nrh32::wide_color_launch(c, (hipStream_t)stream);
  if (wrc) return fail(wrc == -1 ? NRH_E_INVALID : NRH_E_LAUNCH, "wide reflectance kernel: launch failed%s", "");
  return check_launch("color32_kernel");
}

long long nrh_render_workspace_floats(long long nrays) {
  if (nrays < 0) return -1;
  long long tot = 0;
#define X(name, per) tot += round64(nrays * (long long)(per));
  NRH_WS_FIELDS(X)
#undef X
  tot += (long long)mlp_grid() * nrh::WG_WAVES * nrh::SDF_SCRATCH_FLOATS_PER_WAVE;
  return tot;
}

// The reference's sphere_trace (models/neus_hint_model.py:359-372): up to `iterations` rounds of "evaluate the SDF at every ray's
// point, advance the rays that have not stopped".  The loop ends early when a round moved no ray; the flag is read back every
// 4th round (one stream synchronisation each), so this call cannot be captured into a hipGraph.
static int sphere_trace_impl(const NrhNet* net, const float* origins, const float* directions, long long n, int iterations,
                             float threshold, float far_, float* pts, float* depth, float* sdf, float* zero_t, int* moved,
                             hipStream_t st) {
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
    return fail(NRH_E_UNSUPPORTED, "DepthComputationType.SphereTracing reads a flag back per few iterations and cannot be captured into a graph%s", "");
  if (hipMemcpyAsync(pts, origins, sizeof(float) * 3 * n, hipMemcpyDeviceToDevice, st) != hipSuccess ||
      hipMemsetAsync(depth, 0, sizeof(float) * n, st) != hipSuccess || hipMemsetAsync(zero_t, 0, sizeof(float) * n, st) != hipSuccess)
    return fail(NRH_E_LAUNCH, "sphere trace: memset/memcpy failed%s", "");
  nrh::TraceArgs a;
  a.rd = directions; a.sdf = sdf; a.pts = pts; a.depth = depth; a.moved = moved; a.threshold = threshold; a.far_ = far_; a.nrays = (int)n;
  for (int it = 0; it < iterations; ++it) {
    int rc = sdf_eval_impl(net->precision, 0, net->sdf_w, net->sdf_b, net->sdf_head, pts, directions, zero_t, 1, 1, n, sdf, 1, nullptr,
                           nullptr, nullptr, st, WideNet{net->sdf_w32, net->sdf_tab32});
    if (rc) return rc;
    const bool probe = (it & 3) == 3 || it + 1 == iterations;
    if (probe && hipMemsetAsync(moved, 0, sizeof(int), st) != hipSuccess) return fail(NRH_E_LAUNCH, "sphere trace: memset failed%s", "");
    hipLaunchKernelGGL(nrh::sphere_trace_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
    rc = check_launch("sphere_trace_step_kernel");
    if (rc) return rc;
    if (probe) {
      int h = 1;
      if (hipMemcpyAsync(&h, moved, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess)
        return fail(NRH_E_LAUNCH, "sphere trace: flag read-back failed%s", "");
      if (!h) break;
    }
  }
  return NRH_OK;
}

static int render_forward_impl(const NrhNet* net, const float* origins, const float* directions, const float* pl_positions,
                               const float* nears, const float* fars, long long nrays, const float* background,
                               float cos_anneal, const float* t_rand_primary, const float* t_rand_shadow, int zero_hints,
                               const float* lin64, const float* lin16, float* rgb, float* depth, float* weights,
                               float* inside_sphere, float* analytic_normals, float* normalized_normals, float* visibilities,
                               float* specular_cue, float* mid_z, float* dists, float* normal_map, float* normalized_normal_map,
                               const NrhTrainSaves* train, float* workspace, long long workspace_floats, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  if (!net || !net->sdf_w || !net->sdf_b || !net->sdf_head || !net->col_w || !net->col_b)
    return fail(NRH_E_INVALID, "nrh_render_forward: null network pointer%s", "");
  if (net->precision < 0 || net->precision > 2)
    return fail(NRH_E_INVALID, "nrh_render_forward: net->precision must be 0 (f32), 1 (f16x3) or 2 (f16, evaluation only)%s", "");
  // precision 2: everything as precision 1 (same packed buffers), with the wide SDF kernels in their one-term builds
  NrhNet eff;
  const int one_term = net->precision == 2;
  if (one_term) {
    if (!net->sdf_w32 || !net->sdf_tab32) return fail(NRH_E_INVALID, "nrh_render_forward: precision 2 (f16) needs the wide streams sdf_w32 / sdf_tab32%s", "");
    if (train) return fail(NRH_E_UNSUPPORTED, "nrh_render_forward_train: precision 2 (f16, single pass) is an evaluation mode%s", "");
    eff = *net;
    eff.precision = 1;
    net = &eff;
  }
  OneTermScope one_term_scope(one_term);
  if ((net->hints != 0 && net->hints != 1) || (net->normal_type != 0 && net->normal_type != 1) ||
      net->depth_type < 0 || net->depth_type > 2)
    return fail(NRH_E_UNSUPPORTED, "nrh_render_forward: hints / normal_type must be 0 or 1, depth_type 0, 1 or 2%s", "");
  if (net->n_coarse == 0 && net->samples != 0 && net->samples != 64 && net->samples != 128)
    return fail(NRH_E_UNSUPPORTED, "nrh_render_forward: NrhNet.samples must be 0 / 128 (64 + 64 importance) or 64 (no importance samples)%s", "");
  if (const char* why = plan_problem(net)) return fail(NRH_E_UNSUPPORTED, "nrh_render_forward: sample counts: %s", why);
  const SamplerPlan pplan = primary_plan(net, lin16), splan = shadow_plan(net, lin16);
  const int T_primary = pplan.total(), T_shadow = splan.total();
  if (net->n_coarse != 0 && net->samples != T_primary)
    return fail(NRH_E_INVALID, "nrh_render_forward: NrhNet.samples must equal n_coarse + n_steps * n_new%s", "");
  if ((net->bg_alpha || net->shadow_clip > 0) && (T_primary != 128 || T_shadow != 128 || pplan.nc != 64 || splan.nc != 64))
    return fail(NRH_E_UNSUPPORTED, "nrh_render_forward: the outside-NeRF blend and the partial visibility hint need the default sample counts%s", "");
  if (net->bg_alpha && (net->samples == 64 || net->shadow_clip > 0))
    return fail(NRH_E_UNSUPPORTED, "nrh_render_forward: the outside-NeRF blend needs the 128-sample layout and the hit-point shadow mode%s", "");
  if (net->samples == 64 && net->shadow_clip > 0)
    return fail(NRH_E_UNSUPPORTED, "nrh_render_forward: the partial visibility hint needs the 128-sample layout%s", "");
  const int no_hints = zero_hints || !net->hints;  // no shadow march: geometry warm-up, or the pl-naive model
  if (net->shadow_clip != 0 && net->shadow_clip != -1 &&
      (net->shadow_clip < 1 || net->shadow_clip > nrh::MAX_SHADOW_CLIP || (net->shadow_clip & (net->shadow_clip - 1)) != 0))
    return fail(NRH_E_UNSUPPORTED, "nrh_render_forward: shadow_clip (n_shadow_importance_clip) must be -1 / 0 (hit point) or a power of two <= 16%s", "");
  // partial visibility hint (n_shadow_importance_clip > 0): one shadow ray per group of 128 / clip samples instead of one per ray
  const int clip = (!no_hints && net->shadow_clip > 0) ? net->shadow_clip : 0;
  if (!origins || !directions || !pl_positions || !nears || !fars || !lin64 || !lin16 || (!rgb && !train) || !workspace)
    return fail(NRH_E_INVALID, "nrh_render_forward: null pointer%s", "");
  if (train && (!train->sdf || !train->feat_rows || !train->save_h || !train->save_s1 || !train->save_t || !train->save_ge))
    return fail(NRH_E_INVALID, "nrh_render_forward_train: null pointer in NrhTrainSaves%s", "");
  if (nrays < 0 || nrays > (1LL << 24)) return fail(NRH_E_INVALID, "nrh_render_forward: nrays out of range (chunk the call)%s", "");
  if (nrays == 0) return NRH_OK;
  if (workspace_floats < nrh_render_workspace_floats(nrays))
    return fail(NRH_E_WORKSPACE, "nrh_render_forward: workspace too small%s (need %lld floats)", "", nrh_render_workspace_floats(nrays));
  if (((uintptr_t)workspace & 15) != 0) return fail(NRH_E_INVALID, "nrh_render_forward: workspace must be 16-byte aligned%s", "");
  const long long n = nrays;
  float* p = workspace;
#define X(name, per) float* ws_##name = p; p += round64(n * (long long)(per));
  NRH_WS_FIELDS(X)
#undef X
  float* scratch = p;
  // outputs the caller asked for are written in place, the rest live in the workspace
  float* o_weights = weights ? weights : ws_weights;
  float* o_inside = inside_sphere ? inside_sphere : ws_inside;
  float* o_grad = analytic_normals ? analytic_normals : ws_grad_c;
  float* o_nhat = normalized_normals ? normalized_normals : ws_nhat;
  float* o_depth = depth ? depth : ws_depth;
  float* o_vis = visibilities ? visibilities : ws_vis;
  float* o_cue_b = specular_cue;  // optional broadcast copy
  // shadow_hint_gradient: the caller differentiates the shadow ray's alpha itself and needs its section mid-points / lengths
  float* o_tmid_s = (train && train->shadow_mid_z) ? train->shadow_mid_z : ws_tmid_s;
  float* o_dists_s = (train && train->shadow_dists) ? train->shadow_dists : ws_dists_s;
  float* o_tmid = mid_z ? mid_z : ws_tmid;
  float* o_dists = dists ? dists : ws_dists;

  // ---- primary rays: coarse z, hierarchical sampling ----
  {
    nrh::CoarseArgs c;
    c.near_ = nears; c.far_ = fars; c.lin64 = net->n_coarse ? net->lin_tables : lin64; c.t_rand = t_rand_primary; c.z = ws_zbuf; c.nrays = (int)n;
    c.nc = pplan.nc;
    const long long tot = n * pplan.nc;
    hipLaunchKernelGGL(nrh::coarse_z_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, c);
    int rc = check_launch("coarse_z_kernel");
    if (rc) return rc;
  }
  int rc = NRH_OK;
  // (steps = 0, renderer.n_importance_samples = 0: the coarse samples are final, padded to 128 with alpha = 0; the last section's
  // length is sample_dist = 2 / n_samples, :670)
  rc = run_sampler(net, origins, directions, ws_zbuf, ws_sbuf, ws_znew, ws_snew, pplan, nullptr, 2.0f / (float)pplan.nc, o_tmid, o_dists, n, st,
                   train != nullptr);
  if (rc) return rc;
  if (clip && hipMemcpyAsync(ws_cue_b, ws_zbuf, sizeof(float) * 128 * n, hipMemcpyDeviceToDevice, st) != hipSuccess)
    return fail(NRH_E_LAUNCH, "nrh_render_forward: copy of the sample positions failed%s", "");   // z_vals for the group targets
  // ---- render_core: sdf + feature + gradient at the 128 section mid-points ----
  float* sdf_c = train ? train->sdf : ws_sdf_c;
  if (train) {
    // training: the same evaluation with the feature row-major and the arrays the backward sweeps need (16-point kernels: a wide
    // one-wave-per-SIMD form was built in round 4, measured slower - its row stores do not overlap a single wave's MFMA stream,
    // CHANGELOG.md section 7c - and removed in round 5)
    rc = sdf_train_forward_impl(net->precision, net->sdf_w, net->sdf_b, net->sdf_head, origins, directions, o_tmid, 128, 128, n,
                               sdf_c, o_grad, train->feat_rows, train->save_h, train->save_s1, train->save_t, train->save_ge, train->save_h16, train->save_t16, stream);
  } else {
    rc = sdf_eval_impl(net->precision, 2, net->sdf_w, net->sdf_b, net->sdf_head, origins, directions, o_tmid, 128, 128, n, sdf_c, 128,
                       o_grad, ws_feat, scratch, st, WideNet{net->sdf_w32, net->sdf_tab32});
  }
  if (rc) return rc;
  {
    nrh::CoreArgs c;
    c.ro = origins; c.rd = directions; c.pl = pl_positions; c.sdf = sdf_c; c.grad = o_grad; c.dists = o_dists;
    c.tmid = o_tmid; c.lin64 = net->n_coarse ? net->lin_tables + 2 * LIN_TABLE_STRIDE : lin64; c.t_rand_shadow = t_rand_shadow;
    c.s_coarse = splan.nc; c.weights = o_weights; c.inside = o_inside;
    c.nhat = o_nhat; c.depth = o_depth; c.wsum = ws_wsum; c.cue = ws_cue; c.cue_b = o_cue_b; c.srd = ws_srd;
    c.slast = ws_slast; c.zs = ws_zbuf; c.inv_s = net->inv_s; c.cos_anneal = cos_anneal;
    c.dyn = net->dyn_scalars; c.hit = nullptr; c.hit_n = nullptr; c.depth_in = nullptr; c.hit_in = nullptr;
    c.pts = train ? train->pts : nullptr;
    if (net->depth_type == 2) {
      // DepthComputationType.SphereTracing: the shadow-stage arrays are free until the core kernel has run
      float* q = ws_grad_s;
      float* st_pts = q; q += round64(3 * n);
      float* st_depth = q; q += round64(n);
      float* st_sdf = q; q += round64(n);
      float* st_zero = q; q += round64(n);
      rc = sphere_trace_impl(net, origins, directions, n, 2000, 1e-4f, 100.0f, st_pts, st_depth, st_sdf, st_zero, (int*)q, st);
      if (rc) return rc;
      c.depth_in = st_depth; c.hit_in = st_pts;
    }
    c.zero_hints = no_hints;
    c.bg_alpha = net->bg_alpha; c.tail_t = net->tail_t;
    c.nreal = T_primary == 128 ? 0 : T_primary;
    c.depth_max_weight = net->depth_type == 1;
    c.nrays = (int)n;
    rc = launch_core_alpha(c, st, net);
    if (rc) return rc;
  }
  // ---- shadow rays light -> hit point ----
  if (!no_hints && !clip) {
    if (splan.steps == 0) {
      // n_shadow_importance_samples = 0 (:397): the coarse shadow samples are final; the last section is light distance / n_shadow_samples
      // (:383, :417), which core_alpha_kernel left per ray in ws_slast - finalised per ray by the step kernel without merge / up-sample
      nrh::StepArgs fa;
      memset(&fa, 0, sizeof(fa));
      fa.ro = pl_positions; fa.rd = ws_srd; fa.z = ws_zbuf; fa.s = ws_sbuf; fa.last_dist_ray = ws_slast; fa.tmid = o_tmid_s; fa.dists = o_dists_s;
      fa.nrays = (int)n; fa.n = splan.nc; fa.n_new = 16; fa.do_finalize = 1;
      rc = sampler_step_impl(fa, st);
    } else {
      rc = run_sampler(net, pl_positions, ws_srd, ws_zbuf, ws_sbuf, ws_znew, ws_snew, splan, ws_slast, 0.0f, o_tmid_s,
                       o_dists_s, n, st, train != nullptr);
    }
    if (rc) return rc;
    // the shadow ray's alpha only needs <direction, gradient>: with the wide kernels that is mode 3 (forward mode, no scratch)
    const int smode = (net->shadow_jvp && net->precision == 1 && net->sdf_w32 && net->sdf_tab32) ? 3 : 1;
    static const long long gsplit_max = prof_env("NRH_SPLIT_GRAD_MAX_PTS") ? atoll(prof_env("NRH_SPLIT_GRAD_MAX_PTS")) : (long long)NRH_SPLIT_GRAD_MAX_PTS;
    if (train && net->precision == 1 && n * 128 <= gsplit_max)     // small training batch: the channel-split kernel (as the sampler passes)
      rc = sdf_grad_split_impl(net->sdf_w, net->sdf_b, net->sdf_head, pl_positions, ws_srd, o_tmid_s, 128, 128, n, ws_sdf_s, 128, ws_grad_s, st);
    else
      rc = sdf_eval_impl(net->precision, smode, net->sdf_w, net->sdf_b, net->sdf_head, pl_positions, ws_srd, o_tmid_s, 128, 128, n, ws_sdf_s,
                         128, ws_grad_s, nullptr, scratch, st, WideNet{net->sdf_w32, net->sdf_tab32});
    if (rc) return rc;
  }
  {
    nrh::ShadowArgs c;
    c.rd = directions; c.pl = pl_positions; c.srd = ws_srd; c.sdf = ws_sdf_s; c.grad = ws_grad_s; c.dists = o_dists_s;
    c.cue = ws_cue; c.vis = o_vis; c.raymisc = (train && train->raymisc) ? train->raymisc : ws_raymisc; c.inv_s = net->inv_s; c.cos_anneal = cos_anneal;
    c.dyn = net->dyn_scalars;
    c.nrays = (int)n; c.zero_hints = no_hints || clip; c.row_mul = 0; c.row_off = 0;   // clip: rows rewritten per group below
    c.nreal = T_shadow == 128 ? 0 : T_shadow;
    hipLaunchKernelGGL(nrh::shadow_finish_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, c);
    rc = check_launch("shadow_finish_kernel");
    if (rc) return rc;
  }
  // ---- partial visibility hint: one shadow ray per group of 128 / clip samples (n_shadow_importance_clip > 0, :553-575) ----
  const float* raymisc_rows = ws_raymisc;   // what the reflectance kernel reads, and at how many samples per row
  int misc_shift = 7;
  if (clip) {
    const int ratio = 128 / clip;
    float* rows = (train && train->raymisc) ? train->raymisc : ws_raymisc_g;   // [n * clip, RAYMISC_STRIDE]
    float* visg = (train && train->vis_groups) ? train->vis_groups : ws_cue_b + 128 * n;   // [n, clip]
    for (int g = 0; g < clip; ++g) {
      nrh::PartialSetupArgs ps;
      ps.ro = origins; ps.rd = directions; ps.pl = pl_positions; ps.z = ws_cue_b /* the kept z_vals */; ps.lin64 = lin64;
      ps.t_rand_shadow = t_rand_shadow; ps.srd = ws_srd; ps.slast = ws_slast; ps.zs = ws_zbuf; ps.shadow_om = shadow_one_minus_offset(net);
      ps.z_index = g * ratio; ps.clip = clip; ps.group = g; ps.nrays = (int)n;
      hipLaunchKernelGGL(nrh::partial_shadow_setup_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, ps);
      rc = check_launch("partial_shadow_setup_kernel");
      if (rc) return rc;
      rc = run_sampler(net, pl_positions, ws_srd, ws_zbuf, ws_sbuf, ws_znew, ws_snew, splan, ws_slast, 0.0f, ws_tmid_s, ws_dists_s, n, st, train != nullptr);
      if (rc) return rc;
      rc = sdf_eval_impl(net->precision, 1, net->sdf_w, net->sdf_b, net->sdf_head, pl_positions, ws_srd, ws_tmid_s, 128, 128, n, ws_sdf_s,
                         128, ws_grad_s, nullptr, scratch, st, WideNet{net->sdf_w32, net->sdf_tab32});
      if (rc) return rc;
      nrh::ShadowArgs c;
      c.rd = directions; c.pl = pl_positions; c.srd = ws_srd; c.sdf = ws_sdf_s; c.grad = ws_grad_s; c.dists = ws_dists_s;
      c.cue = ws_cue; c.vis = visg; c.raymisc = rows; c.inv_s = net->inv_s; c.cos_anneal = cos_anneal; c.dyn = net->dyn_scalars;
      c.nrays = (int)n; c.zero_hints = 0; c.row_mul = clip; c.row_off = g; c.nreal = 0;
      hipLaunchKernelGGL(nrh::shadow_finish_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, c);
      rc = check_launch("shadow_finish_kernel");
      if (rc) return rc;
    }
    hipLaunchKernelGGL(nrh::partial_shadow_map_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, (const float*)o_weights,
                       (const float*)visg, o_vis, clip, (int)n);
    rc = check_launch("partial_shadow_map_kernel");
    if (rc) return rc;
    raymisc_rows = rows;
    misc_shift = 0;
    while ((1 << misc_shift) < ratio) ++misc_shift;
  }
  if (train) return NRH_OK;  // reflectance + composite are differentiated by the caller
  // ---- reflectance + composite ----
  if (net->sampled_color) ws_color = net->sampled_color;      // outside NeRF: the caller blends the colours itself (:630-633)
  // feat_fused: ws_feat holds W0feat * feature (the wide mode-2 stream was packed with the product matrix)
  const int fused = (net->precision == 1 && net->feat_fused && net->sdf_w32 && net->sdf_tab32) ? 1 : 0;
  if (fused && net->hints && net->col_w32 && net->col_tab32 && !clip) {
    // the reflectance net on the wide machinery too (csrc/nrh_color32.hip; one raymisc row per workgroup pass = per ray, so
    // the partial shadow mode with its row per sample group stays on the 16-point kernel)
    nrh32::WideColorCall c;
    c.stream = net->col_w32; c.tables = net->col_tab32; c.part = ws_feat; c.ro = origins; c.rd = directions; c.tmid = o_tmid;
    c.nhat = net->normal_type ? o_grad : o_nhat; c.raymisc = ws_raymisc; c.color = ws_color; c.nrays = n;
    c.raymisc_stride = nrh::RAYMISC_STRIDE; c.max_grid = device_cus();
    TimedLaunch tl;
    bool timed;
    timing_begin(3, st, tl, timed);
    const int wrc = nrh32::wide_color_launch(c, st);
    timing_end(st, tl, timed);
    if (wrc) return fail(wrc == -1 ? NRH_E_INVALID : NRH_E_LAUNCH, "wide reflectance kernel: launch failed%s", "");
    rc = check_launch("color32_kernel");
  } else {
    rc = color_eval_impl(net->precision, net->hints, net->col_w, net->col_b, ws_feat, origins, directions, o_tmid,
                         net->normal_type ? o_grad : o_nhat, raymisc_rows, n, ws_color, st, fused, misc_shift);
  }
  if (rc) return rc;
  {
    nrh::CompositeArgs c;
    c.color = ws_color; c.weights = o_weights; c.wsum = ws_wsum; c.bg = background; c.rgb = rgb; c.nrays = (int)n;
    c.inside = o_inside; c.grad = o_grad; c.nhat = o_nhat; c.nmap = normal_map; c.nnmap = normalized_normal_map;
    hipLaunchKernelGGL(nrh::composite_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, c);
    rc = check_launch("composite_kernel");
    if (rc) return rc;
  }
  return NRH_OK;
}

int nrh_fuse_feature_head(const float* col_w0, int ld0, const float* feat_w, const float* feat_b, float* out_w, float* out_b, void* stream) {
  if (!col_w0 || !feat_w || !feat_b || !out_w || !out_b) return fail(NRH_E_INVALID, "nrh_fuse_feature_head: null pointer%s", "");
  if (ld0 < 316) return fail(NRH_E_INVALID, "nrh_fuse_feature_head: the first reflectance layer has at least 316 input columns%s", "");
  hipLaunchKernelGGL(nrh::fuse_head_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, col_w0, ld0, feat_w, feat_b, out_w, out_b);
  return check_launch("fuse_head_kernel");
}

// ---- Adam in one launch (csrc/nrh_adam.hip) ----
int nrh_adam_step(const NrhAdamTensor* tensors_dev, int ntensors, const int* chunks_dev, int nchunks, int ngroups, const double* lr,
                  const float* const* lr_dev, const double* beta1, const double* beta2, const double* eps, void* stream) {
  static_assert(sizeof(NrhAdamTensor) == sizeof(nrhadam::Tensor), "NrhAdamTensor layout");
  if (ntensors == 0 || nchunks == 0) return NRH_OK;
  if (!tensors_dev || !chunks_dev || !lr || !beta1 || !beta2 || !eps || ntensors < 0 || nchunks < 0)
    return fail(NRH_E_INVALID, "nrh_adam_step: null pointer / negative count%s", "");
  if (ngroups < 1 || ngroups > nrhadam::MAX_GROUPS) return fail(NRH_E_INVALID, "nrh_adam_step: 1..4 parameter groups%s", "");
  nrhadam::Args a;
  memset(&a, 0, sizeof(a));
  a.tensors = (const nrhadam::Tensor*)tensors_dev; a.chunks = chunks_dev; a.ntensors = ntensors;
  for (int g = 0; g < ngroups; ++g) {
    // scalars rounded to float32 as torch does with python floats (1 - beta in double first)
    a.lrd[g] = lr[g]; a.lr_ptr[g] = lr_dev ? lr_dev[g] : nullptr; a.b1d[g] = beta1[g]; a.b2d[g] = beta2[g]; a.b2[g] = (float)beta2[g];
    a.w1[g] = (float)(1.0 - beta1[g]); a.w2[g] = (float)(1.0 - beta2[g]); a.eps[g] = (float)eps[g];
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(nrhadam::adam_kernel, dim3((unsigned)nchunks), dim3(256), 0, st, a);
  int rc = check_launch("adam_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(nrhadam::adam_bump_kernel, dim3((unsigned)((ntensors + 63) / 64)), dim3(64), 0, st, a.tensors, ntensors);
  return check_launch("adam_bump_kernel");
}

// ---- unit entries of the evaluation render's per-ray stages (the kernels nrh_render_forward launches) ----
long long nrh_sphere_trace_workspace_floats(long long nrays) { return nrays < 0 ? -1 : 2 * round64(nrays) + 64; }

int nrh_sphere_trace(const NrhNet* net, const float* origins, const float* directions, long long nrays, int iterations,
                     float threshold, float far_depth, float* points, float* depths, float* workspace, long long workspace_floats,
                     void* stream) {
  if (!net || !net->sdf_w || !net->sdf_b || !net->sdf_head || !origins || !directions || !points || !depths || !workspace)
    return fail(NRH_E_INVALID, "nrh_sphere_trace: null pointer%s", "");
  if (net->precision < 0 || net->precision > 1) return fail(NRH_E_INVALID, "nrh_sphere_trace: net->precision must be 0 or 1%s", "");
  if (nrays < 0 || nrays > (1LL << 24) || iterations < 0) return fail(NRH_E_INVALID, "nrh_sphere_trace: nrays / iterations out of range%s", "");
  if (workspace_floats < nrh_sphere_trace_workspace_floats(nrays)) return fail(NRH_E_WORKSPACE, "nrh_sphere_trace: workspace too small%s", "");
  if (nrays == 0) return NRH_OK;
  float* sdf = workspace;
  float* zero_t = sdf + round64(nrays);
  return sphere_trace_impl(net, origins, directions, nrays, iterations, threshold, far_depth, points, depths, sdf, zero_t,
                           (int*)(zero_t + round64(nrays)), (hipStream_t)stream);
}

int nrh_alpha_composite(const float* origins, const float* directions, const float* pl_positions, const float* sdf, const float* grad,
                        const float* dists, const float* mid_z, float inv_s, float cos_anneal, int depth_type, int zero_hints,
                        const float* lin64, const float* t_rand_shadow, long long nrays, float* weights, float* inside_sphere,
                        float* normalized_normals, float* depth, float* weight_sum, float* specular_cue, float* hit_points,
                        float* hit_normals, float* shadow_dirs, float* shadow_last_dist, float* shadow_z, void* stream) {
  if (!origins || !directions || !pl_positions || !sdf || !grad || !dists || !mid_z || !lin64 || !weights || !inside_sphere ||
      !normalized_normals || !depth || !weight_sum || !specular_cue || !shadow_dirs || !shadow_last_dist || !shadow_z)
    return fail(NRH_E_INVALID, "nrh_alpha_composite: null pointer%s", "");
  if (nrays < 0 || nrays > 0x7fffffffLL) return fail(NRH_E_INVALID, "nrh_alpha_composite: nrays out of range%s", "");
  if (depth_type != 0 && depth_type != 1) return fail(NRH_E_UNSUPPORTED, "nrh_alpha_composite: depth_type must be 0 (alpha blend) or 1 (maximal weight)%s", "");
  if (nrays == 0) return NRH_OK;
  nrh::CoreArgs c;
  memset(&c, 0, sizeof(c));
  c.ro = origins; c.rd = directions; c.pl = pl_positions; c.sdf = sdf; c.grad = grad; c.dists = dists; c.tmid = mid_z;
  c.lin64 = lin64; c.t_rand_shadow = t_rand_shadow; c.weights = weights; c.inside = inside_sphere; c.nhat = normalized_normals;
  c.depth = depth; c.wsum = weight_sum; c.cue = specular_cue; c.cue_b = nullptr; c.hit = hit_points; c.hit_n = hit_normals;
  c.srd = shadow_dirs; c.slast = shadow_last_dist; c.zs = shadow_z; c.inv_s = inv_s; c.cos_anneal = cos_anneal; c.dyn = nullptr;
  c.zero_hints = zero_hints ? 1 : 0; c.depth_max_weight = depth_type; c.nrays = (int)nrays;
  return launch_core_alpha(c, (hipStream_t)stream, nullptr);   // (unit entry: the reference's default constants)
}

int nrh_visibility(const float* directions, const float* pl_positions, const float* shadow_dirs, const float* sdf, const float* grad,
                   const float* dists, const float* specular_cue, float inv_s, float cos_anneal, int zero_hints, long long nrays,
                   float* visibilities, float* raymisc, void* stream) {
  if (!directions || !pl_positions || !visibilities || !raymisc) return fail(NRH_E_INVALID, "nrh_visibility: null pointer%s", "");
  if (!zero_hints && (!shadow_dirs || !sdf || !grad || !dists || !specular_cue))
    return fail(NRH_E_INVALID, "nrh_visibility: null pointer%s (shadow-ray inputs are required unless zero_hints)", "");
  if (nrays < 0 || nrays > 0x7fffffffLL) return fail(NRH_E_INVALID, "nrh_visibility: nrays out of range%s", "");
  if (nrays == 0) return NRH_OK;
  nrh::ShadowArgs c;
  memset(&c, 0, sizeof(c));
  c.rd = directions; c.pl = pl_positions; c.srd = shadow_dirs; c.sdf = sdf; c.grad = grad; c.dists = dists; c.cue = specular_cue;
  c.vis = visibilities; c.raymisc = raymisc; c.inv_s = inv_s; c.cos_anneal = cos_anneal; c.dyn = nullptr;
  c.nrays = (int)nrays; c.zero_hints = zero_hints ? 1 : 0;
  hipLaunchKernelGGL(nrh::shadow_finish_kernel, dim3((unsigned)((nrays + 3) / 4)), dim3(256), 0, (hipStream_t)stream, c);
  return check_launch("shadow_finish_kernel");
}

int nrh_color_composite(const float* sampled_color, const float* weights, const float* weight_sum, const float* background,
                        const float* inside_sphere, const float* analytic_normals, const float* normalized_normals, long long nrays,
                        float* rgb, float* normal_map, float* normalized_normal_map, void* stream) {
  if (!sampled_color || !weights || !weight_sum || !rgb) return fail(NRH_E_INVALID, "nrh_color_composite: null pointer%s", "");
  if ((normal_map || normalized_normal_map) && (!inside_sphere || !analytic_normals || !normalized_normals))
    return fail(NRH_E_INVALID, "nrh_color_composite: the normal maps need inside_sphere and both normal arrays%s", "");
  if (nrays < 0 || nrays > 0x7fffffffLL) return fail(NRH_E_INVALID, "nrh_color_composite: nrays out of range%s", "");
  if (nrays == 0) return NRH_OK;
  nrh::CompositeArgs c;
  memset(&c, 0, sizeof(c));
  c.color = sampled_color; c.weights = weights; c.wsum = weight_sum; c.bg = background; c.rgb = rgb; c.nrays = (int)nrays;
  c.inside = inside_sphere; c.grad = analytic_normals; c.nhat = normalized_normals; c.nmap = normal_map; c.nnmap = normalized_normal_map;
  hipLaunchKernelGGL(nrh::composite_kernel, dim3((unsigned)((nrays + 3) / 4)), dim3(256), 0, (hipStream_t)stream, c);
  return check_launch("composite_kernel");
}

int nrh_render_forward(const NrhNet* net, const float* origins, const float* directions, const float* pl_positions,
                       const float* nears, const float* fars, long long nrays, const float* background,
                       float cos_anneal, const float* t_rand_primary, const float* t_rand_shadow, int zero_hints,
                       const float* lin64, const float* lin16, float* rgb, float* depth, float* weights,
                       float* inside_sphere, float* analytic_normals, float* normalized_normals, float* visibilities,
                       float* specular_cue, float* mid_z, float* dists, float* normal_map, float* normalized_normal_map,
                       float* workspace, long long workspace_floats, void* stream) {
  return render_forward_impl(net, origins, directions, pl_positions, nears, fars, nrays, background, cos_anneal, t_rand_primary,
                             t_rand_shadow, zero_hints, lin64, lin16, rgb, depth, weights, inside_sphere, analytic_normals,
                             normalized_normals, visibilities, specular_cue, mid_z, dists, normal_map, normalized_normal_map,
                             nullptr, workspace, workspace_floats, stream);
}

int nrh_render_forward_train(const NrhNet* net, const float* origins, const float* directions, const float* pl_positions,
                             const float* nears, const float* fars, long long nrays, float cos_anneal,
                             const float* t_rand_primary, const float* t_rand_shadow, int zero_hints, const float* lin64,
                             const float* lin16, float* depth, float* weights, float* inside_sphere, float* analytic_normals,
                             float* normalized_normals, float* visibilities, float* specular_cue, float* mid_z, float* dists,
                             const NrhTrainSaves* saves, float* workspace, long long workspace_floats, void* stream) {
  if (!saves) return fail(NRH_E_INVALID, "nrh_render_forward_train: saves is null%s", "");
  return render_forward_impl(net, origins, directions, pl_positions, nears, fars, nrays, nullptr, cos_anneal, t_rand_primary,
                             t_rand_shadow, zero_hints, lin64, lin16, nullptr, depth, weights, inside_sphere, analytic_normals,
                             normalized_normals, visibilities, specular_cue, mid_z, dists, nullptr, nullptr, saves, workspace,
                             workspace_floats, stream);
}

}  // extern "C"
