// Weight-norm fold of all linears in one launch, and its adjoint in one launch.
//   W[r][:] = v[r][:] * g[r] / ||v[r][:]||_2        (old-style nn.utils.weight_norm, dim = 0;
//                                                    reference: fields/sdf_field.py:81-82, fields/reflectance_network.py:61-62)
//   adjoint:  gbar[r] = <Wbar[r], v[r]> / ||v[r]||;   vbar[r] = (g[r]/||v[r]||) (Wbar[r] - <Wbar[r], v[r]> v[r] / ||v[r]||^2)
// A training step folds the 15 linears and differentiates the fold: ~200 small PyTorch kernels per step otherwise.
// One wavefront per matrix row (rows of all layers are enumerated consecutively).
#include "nrh_common.h"

namespace nrh {

constexpr int FOLD_MAX_LAYERS = 16;
constexpr int FOLD_MAX_COLS = 384;   // 6 elements per lane

struct FoldArgs {
  const float* v[FOLD_MAX_LAYERS];
  const float* g[FOLD_MAX_LAYERS];
  float* w[FOLD_MAX_LAYERS];          // forward out
  const float* wbar[FOLD_MAX_LAYERS]; // adjoint in (null: layer skipped, zero gradients)
  float* vbar[FOLD_MAX_LAYERS];       // adjoint out
  float* gbar[FOLD_MAX_LAYERS];       // adjoint out
  int cols[FOLD_MAX_LAYERS];
  int row_start[FOLD_MAX_LAYERS + 1];
  int nlayers;
};

template <bool ADJOINT>
__global__ __launch_bounds__(256) void fold_kernel(const FoldArgs a) {
  const int lane = threadIdx.x & 63;
  const int row_g = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row_g >= a.row_start[a.nlayers]) return;
  int l = 0;
#pragma unroll 1
  while (l + 1 < a.nlayers && row_g >= a.row_start[l + 1]) ++l;
  const int r = row_g - a.row_start[l];
  const int C = a.cols[l];
  const float* v = a.v[l] + (size_t)r * C;
  float x[6], ss = 0.0f;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int c = lane + 64 * i;
    x[i] = (c < C) ? v[c] : 0.0f;
    ss += x[i] * x[i];
  }
  const float nrm = sqrtf(wave_sum(ss));
  const float g = a.g[l][r];
  if (!ADJOINT) {
    const float s = g / nrm;
    float* w = a.w[l] + (size_t)r * C;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int c = lane + 64 * i;
      if (c < C) w[c] = x[i] * s;
    }
    return;
  }
  float* vb = a.vbar[l] + (size_t)r * C;
  if (a.wbar[l] == nullptr) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int c = lane + 64 * i;
      if (c < C) vb[c] = 0.0f;
    }
    if (lane == 0) a.gbar[l][r] = 0.0f;
    return;
  }
  const float* wb = a.wbar[l] + (size_t)r * C;
  float y[6], dot = 0.0f;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int c = lane + 64 * i;
    y[i] = (c < C) ? wb[c] : 0.0f;
    dot += y[i] * x[i];
  }
  dot = wave_sum(dot);
  const float inv = 1.0f / nrm;
  const float s = g * inv, k = dot * inv * inv;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int c = lane + 64 * i;
    if (c < C) vb[c] = s * (y[i] - k * x[i]);
  }
  if (lane == 0) a.gbar[l][r] = dot * inv;
}

}  // namespace nrh

// ---- re-packing after an optimiser step: the index plans of nrhints_amd/packing.py / packing32.py as ONE launch each -------------
// A packed buffer is out[e] = f(flat[index[e]]) for a fixed index plan (element permutations, zero padding = index 0) with
//   mode 0: float32 copy;  mode 1: x = flat[i] / factor[e];  mode 2: x = flat[i] * factor[e];  modes 1, 2: x -> fp16 hi + lo with
//   lo = (x - hi) * 2^11, stored as 512-element blocks [hi 512 | lo 512] (the f16x3 kernels' stream layout).
// Same roundings as the torch expressions they replace (round-to-nearest conversions, IEEE division): bit-identical packs.
namespace nrh {

struct PackGatherArgs {
  const float* flat;
  const int* index;
  const float* factor;
  void* out;
  long long n;
  int mode;
};

__global__ __launch_bounds__(256) void pack_gather_kernel(const PackGatherArgs a) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= a.n) return;
  const float v = a.flat[a.index[e]];
  if (a.mode == 0) {
    reinterpret_cast<float*>(a.out)[e] = v;
    return;
  }
  const float x = a.mode == 1 ? v / a.factor[e] : v * a.factor[e];
  const _Float16 hi = (_Float16)x;
  const _Float16 lo = (_Float16)((x - (float)hi) * 2048.0f);
  _Float16* o = reinterpret_cast<_Float16*>(a.out) + (e >> 9) * 1024 + (e & 511);
  o[0] = hi;
  o[512] = lo;
}

// packing32.sdf32_tables: [11][256] float32 - rows 0..7 one packed fp16 pair (hi | lo * 2^11 << 16) of b_l * 100 / ln 2 per output
// row (rows past the layer's size zero), row 8 the same of b_feat (unscaled), row 9 {b_s / 3, 0...}, row 10 w_s / 3
struct TablesArgs {
  const float* bias[8];
  int rows[8];
  const float* feat_b;
  const float* head_b;
  const float* head_w;
  float* out;
};
__device__ __forceinline__ float pack_pair_f16(float x) {
  const _Float16 hi = (_Float16)x;
  const _Float16 lo = (_Float16)((x - (float)hi) * 2048.0f);
  const uint32_t bits = (uint32_t)__builtin_bit_cast(unsigned short, hi) | ((uint32_t)__builtin_bit_cast(unsigned short, lo) << 16);
  return __builtin_bit_cast(float, bits);
}
__global__ __launch_bounds__(256) void sdf32_tables_kernel(const TablesArgs a) {
  const int r = blockIdx.x, c = threadIdx.x;
  float v;
  if (r < 8) v = pack_pair_f16(c < a.rows[r] ? a.bias[r][c] * 144.26950408889634074f : 0.0f);
  else if (r == 8) v = pack_pair_f16(a.feat_b[c]);
  else if (r == 9) v = c == 0 ? a.head_b[0] / 3.0f : 0.0f;
  else v = a.head_w[c] / 3.0f;
  a.out[r * 256 + c] = v;
}

// -------------------------------------------------------------------------------------------------
// packing32.fuse_feature_head on the device: (W0[:, 60:316] W_feat, W0[:, 60:316] b_feat) with float64 accumulation, rounded
// to float32 once - the feature head of the SDF net multiplied into the feature block of the reflectance net's first layer
// (fields/sdf_field.py:119-123 -> fields/reflectance_network.py:77-84).  One thread per output entry, a 256-long dot product
// each (16.8 M FMAs in all: not worth a library GEMM call, and it keeps rocBLAS out of the process).
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fuse_head_kernel(const float* w0, int ld0, const float* feat_w, const float* feat_b,
                                                        float* out_w, float* out_b) {
  const int r = blockIdx.x, c = threadIdx.x;        // out_w[r][c], 256 x 256
  const float* wr = w0 + (long long)r * ld0 + 60;
  double acc = 0.0;
  for (int k = 0; k < 256; ++k) acc += (double)wr[k] * (double)feat_w[k * 256 + c];
  out_w[r * 256 + c] = (float)acc;
  if (c == 0) {
    double b = 0.0;
    for (int k = 0; k < 256; ++k) b += (double)wr[k] * (double)feat_b[k];
    out_b[r] = (float)b;
  }
}

}  // namespace nrh
