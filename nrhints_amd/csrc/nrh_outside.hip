// The outside-NeRF background network on gfx950 (renderer.use_outside_nerf; reference: fields/nerf_density_field.py:30-89, called
// from models/neus_hint_model.py:434-473): per sample point of the inverted-sphere parameterisation (p / |p|, 1 / |p|)
//
//   x  = enc10(pts4)                         84 inputs (padded to 96)
//   h0 = relu(N0 x);  h_l = relu(N_l h_{l-1})  l = 1..7, layer 5 reads cat[x, h_4] (the skip after layer index 4: 340 inputs,
//                                              split along K into N5a over h_4 and N5b over x)
//   density = alpha_linear(h_7)              1 row, carried as row 256 of the feature stage (a ninth 32-row chunk)
//   f  = feature_linear(h_7)                 256, no activation
//   hv = relu(V [f, enc4(cat[view, light])]) 128 rows; K split into VA over f and VB over the 54 (-> 64) encoded entries
//   rgb = rgb_linear(hv)                     3 rows (before the sigmoid)
//
// on the transposed register chain of nrh_mlp.h (16-point tiles, both precision modes), with a training variant that saves the
// operands of the weight gradients row-major, and the adjoint sweep through the transposed stages.  The weight gradients
// themselves are jobs of nrh_dw_gemm.  Everything around the network (sample positions, alpha = 1 - exp(-softplus(density) dist),
// the 32-sample tail composite) is elementwise work on [N,160] arrays and stays with the caller (nrhints_amd/outside.py).
#pragma once
#include "nrh_mlp.h"

namespace nrh {

// ---- packed geometry (floats; precision f16x3: the same number of fp16 hi/lo pairs) ----
constexpr int ON_X = 96, ON_V = 64, ON_XREAL = 84, ON_VREAL = 54;
constexpr int ON_N0_FLOATS = 8 * 2 * 6 * 256;        // 256 x 96
constexpr int ON_HF_FLOATS = 9 * 2 * 16 * 256;       // 288 x 256: feature (256 rows) + density (row 256)
constexpr int ON_VA_FLOATS = 4 * 2 * 16 * 256;       // 128 x 256
constexpr int ON_VB_FLOATS = 4 * 2 * 4 * 256;        // 128 x 64
constexpr int ON_RGB_FLOATS = 1 * 2 * 8 * 256;       // 32 (3 used) x 128
// forward stream: N0 | N1..N4 | N5a | N5b | N6 | N7 | HF | VA | VB | RGB
constexpr int ON_OFF_N0 = 0;
__host__ __device__ constexpr int on_off_N(int l) { return ON_N0_FLOATS + (l - 1) * SDF_REG_FLOATS; }   // l = 1..4
constexpr int ON_OFF_N5A = ON_N0_FLOATS + 4 * SDF_REG_FLOATS;
constexpr int ON_OFF_N5B = ON_OFF_N5A + SDF_REG_FLOATS;
constexpr int ON_OFF_N6 = ON_OFF_N5B + ON_N0_FLOATS;
constexpr int ON_OFF_N7 = ON_OFF_N6 + SDF_REG_FLOATS;
constexpr int ON_OFF_HF = ON_OFF_N7 + SDF_REG_FLOATS;
constexpr int ON_OFF_VA = ON_OFF_HF + ON_HF_FLOATS;
constexpr int ON_OFF_VB = ON_OFF_VA + ON_VA_FLOATS;
constexpr int ON_OFF_RGB = ON_OFF_VB + ON_VB_FLOATS;
constexpr int ON_PACKED_FLOATS = ON_OFF_RGB + ON_RGB_FLOATS;
// biases: [8][256] pts layers | [256] feature | [16] density (1 used) | [128] views | [16] rgb (3 used)
constexpr int ON_B_FEAT = 8 * 256, ON_B_ALPHA = ON_B_FEAT + 256, ON_B_VIEWS = ON_B_ALPHA + 16, ON_B_RGB = ON_B_VIEWS + 128;
constexpr int ON_BIAS_FLOATS = ON_B_RGB + 32;   // (the rgb stage reads both 16-float blocks of its single chunk)
// transposed stream (adjoint sweep): TRGB | TVA | TVB | THF | T7 | T6 | T5x | T5h | T4 | T3 | T2 | T1 | T0
constexpr int ONT_RGB_FLOATS = 4 * 2 * 2 * 256;      // 128 x 32 (3 used)
constexpr int ONT_VA_FLOATS = 8 * 2 * 8 * 256;       // 256 x 128
constexpr int ONT_VB_FLOATS = 2 * 2 * 8 * 256;       // 64 x 128
constexpr int ONT_X_FLOATS = 3 * 2 * 16 * 256;       // 96 x 256
constexpr int ONT_OFF_RGB = 0;
constexpr int ONT_OFF_VA = ONT_RGB_FLOATS;
constexpr int ONT_OFF_VB = ONT_OFF_VA + ONT_VA_FLOATS;
constexpr int ONT_OFF_HF = ONT_OFF_VB + ONT_VB_FLOATS;
constexpr int ONT_OFF_T7 = ONT_OFF_HF + SDF_REG_FLOATS;
constexpr int ONT_OFF_T6 = ONT_OFF_T7 + SDF_REG_FLOATS;
constexpr int ONT_OFF_T5X = ONT_OFF_T6 + SDF_REG_FLOATS;
constexpr int ONT_OFF_T5H = ONT_OFF_T5X + ONT_X_FLOATS;
__host__ __device__ constexpr int ont_off_T(int l) { return ONT_OFF_T5H + SDF_REG_FLOATS + (4 - l) * SDF_REG_FLOATS; }   // l = 4..1
constexpr int ONT_OFF_T0 = ONT_OFF_T5H + 5 * SDF_REG_FLOATS;
constexpr int ONT_PACKED_FLOATS = ONT_OFF_T0 + ONT_X_FLOATS;

struct OutsideArgs {
  const float* w;        // ON_PACKED_FLOATS
  const float* b;        // ON_BIAS_FLOATS
  const float* pts4;     // [npts,4]
  const float* views;    // [nrays,3]
  const float* pls;      // [nrays,3]
  float* density;        // [npts]
  float* rgb;            // [npts,3]  before the sigmoid
  long long npts;
  int pts_per_ray;
  int ntile_groups;
  // TRAIN: operands of the weight gradients, row-major
  float* save_x;         // [npts][96]   enc10(pts4) (84 used)
  float* save_v;         // [npts][64]   enc4(cat[view, light]) (54 used)
  float* save_h;         // [8][npts][256]
  float* save_f;         // [npts][256]
  float* save_hv;        // [npts][128]
};

struct OutsideAdjArgs {
  const float* wt;       // ONT_PACKED_FLOATS
  const float* walpha;   // [256] alpha_linear.weight
  const float* dbar;     // [npts]    adjoint of the density
  const float* cbar;     // [npts,3]  adjoint of rgb (before the sigmoid)
  const float* save_h;   // [8][npts][256]
  const float* save_hv;  // [npts][128]
  float* zbar;           // [8][npts][256]  adjoints of the pre-ReLU outputs of the 8 pts layers
  float* fbar;           // [npts][256]     adjoint of the feature
  float* zvbar;          // [npts][128]     adjoint of the pre-ReLU output of the views layer
  float* xbar;           // [npts][96]      adjoint of enc10(pts4)
  float* vbar;           // [npts][64]      adjoint of enc4(cat[view, light])
  long long npts;
  int ntile_groups;
  float adj_scale;       // f16x3 only: power of two S; the chain runs on S * seeds, outputs leave as 1 / S (see ColorAdjArgs, nrh_color.hip)
};

struct OnBias { f32x4 b0, b1; };
__device__ __forceinline__ f32x4 on_relu4(const f32x4 x) {
  return f32x4{fmaxf(x[0], 0.0f), fmaxf(x[1], 0.0f), fmaxf(x[2], 0.0f), fmaxf(x[3], 0.0f)};
}

template <int PREC, bool TRAIN>
__global__ __launch_bounds__(MLP_THREADS, 2) void outside_kernel(const OutsideArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  int par = 0;
  dma_chunk(a.w + ON_OFF_N0, smem, 12, wave, lane);
  __syncthreads();

  for (int tg = blockIdx.x; tg < a.ntile_groups; tg += gridDim.x) {
    const long long tile = (long long)tg * WG_WAVES + wave;
    const long long P = tile * TILE_PTS + j;
    const bool valid = P < a.npts;
    const long long Pc = valid ? P : a.npts - 1;
    const long long ray = Pc / a.pts_per_ray;
    const bool tile_ok = tile * TILE_PTS < a.npts;   // TRAIN: whole tiles only (npts % 16 == 0)
    auto save_rows = [&](float* base, int l, int width, int ch, const f32x4 v0, const f32x4 v1) {
      if (tile_ok) {
        float* p = base + ((size_t)l * (size_t)a.npts + (size_t)Pc) * width + 4 * q;
        st_stream(reinterpret_cast<f32x4*>(p + (2 * ch) * 16), v0);
        st_stream(reinterpret_cast<f32x4*>(p + (2 * ch + 1) * 16), v1);
      }
    };
    // (the bias pointer is made opaque once per tile: the loads are invariant over the persistent tile loop, and hipcc would
    // otherwise hoist all of them - 9 stages x 16 floats per lane - out of it and spill them)
    const float* bias = a.b;
    asm volatile("" : "+s"(bias));
    auto bias_pre = [&](int off) {
      return [&, off](int ch) {
        OnBias p;
        p.b0 = *reinterpret_cast<const f32x4*>(bias + off + (2 * ch) * 16 + 4 * q);
        p.b1 = *reinterpret_cast<const f32x4*>(bias + off + (2 * ch + 1) * 16 + 4 * q);
        return p;
      };
    };

    // ---- encodings: entry m = 16 b + 4 q + r of enc10(pts4) / enc4(cat[view, light]) ----
    float x4[4], v6[6];
#pragma unroll
    for (int c = 0; c < 4; ++c) x4[c] = a.pts4[Pc * 4 + c];
#pragma unroll
    for (int c = 0; c < 3; ++c) { v6[c] = a.views[ray * 3 + c]; v6[3 + c] = a.pls[ray * 3 + c]; }
    // (the encodings are rebuilt where they are used - layers 0 and 5, the views layer - instead of being kept in registers
    // across the eight 256 x 256 stages in between: 24 + 16 sines against 40 live registers per lane)
    auto make_xe = [&](Act<PREC, 6>& xe, bool save) {
      float xx[4] = {x4[0], x4[1], x4[2], x4[3]};
      asm volatile("" : "+v"(xx[0]), "+v"(xx[1]), "+v"(xx[2]), "+v"(xx[3]));   // not shared between the two call sites
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        float o[8];
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) o[r8] = nerf_enc_entry_q<4, 10>(xx, (2 * ch + (r8 >> 2)) * 16 + (r8 & 3), q);
        xe.set_chunk(ch, o);
        if (TRAIN && save) save_rows(a.save_x, 0, ON_X, ch, f32x4{o[0], o[1], o[2], o[3]}, f32x4{o[4], o[5], o[6], o[7]});
      }
    };
    auto make_ve = [&](Act<PREC, 4>& ve) {
#pragma unroll
      for (int ch = 0; ch < 2; ++ch) {
        float o[8];
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) o[r8] = nerf_enc_entry_q<6, 4>(v6, (2 * ch + (r8 >> 2)) * 16 + (r8 & 3), q);
        ve.set_chunk(ch, o);
        if constexpr (TRAIN) save_rows(a.save_v, 0, ON_V, ch, f32x4{o[0], o[1], o[2], o[3]}, f32x4{o[4], o[5], o[6], o[7]});
      }
    };

    // ---- N0 ----
    Act<PREC, 16> h;
    {
      Act<PREC, 6> xe;
      make_xe(xe, true);
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1, const OnBias& p) {
        const f32x4 h0 = on_relu4(acc0 + p.b0), h1 = on_relu4(acc1 + p.b1);
        h.set_chunk(ch, h0, h1);
        if constexpr (TRAIN) save_rows(a.save_h, 0, 256, ch, h0, h1);
      };
      run_stage<PREC, 6, 8, false, true>(a.w + ON_OFF_N0, a.w + on_off_N(1), 32, smem, par, xe, nullptr, bias_pre(0), epi, wave, lane);
    }
    // ---- N1..N4 ----
    for (int l = 1; l <= 4; ++l) {
      Act<PREC, 16> ho;
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1, const OnBias& p) {
        const f32x4 h0 = on_relu4(acc0 + p.b0), h1 = on_relu4(acc1 + p.b1);
        ho.set_chunk(ch, h0, h1);
        if constexpr (TRAIN) save_rows(a.save_h, l, 256, ch, h0, h1);
      };
      const float* wn = (l < 4) ? a.w + on_off_N(l + 1) : a.w + ON_OFF_N5A;
      run_stage<PREC, 16, 8, false, true>(a.w + on_off_N(l), wn, 32, smem, par, h, nullptr, bias_pre(l * 256), epi, wave, lane);
      h = ho;
    }
    // ---- N5: cat[x, h_4] (fields/nerf_density_field.py:78-79) as a K split ----
    {
      float part[64];
      auto pre0 = [&](int) { return 0; };
      auto epi0 = [&](int ch, f32x4 acc0, f32x4 acc1, int) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { part[ch * 8 + r] = acc0[r]; part[ch * 8 + 4 + r] = acc1[r]; }
      };
      run_stage<PREC, 16, 8, false>(a.w + ON_OFF_N5A, a.w + ON_OFF_N5B, 12, smem, par, h, nullptr, pre0, epi0, wave, lane);
      Act<PREC, 6> xe;
      make_xe(xe, false);
      Act<PREC, 16> ho;
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1, const OnBias& p) {
        const f32x4 h0 = on_relu4(acc0 + p.b0), h1 = on_relu4(acc1 + p.b1);
        ho.set_chunk(ch, h0, h1);
        if constexpr (TRAIN) save_rows(a.save_h, 5, 256, ch, h0, h1);
      };
      run_stage<PREC, 6, 8, true, true>(a.w + ON_OFF_N5B, a.w + ON_OFF_N6, 32, smem, par, xe, part, bias_pre(5 * 256), epi, wave, lane);
      h = ho;
    }
    // ---- N6, N7 ----
    for (int l = 6; l <= 7; ++l) {
      Act<PREC, 16> ho;
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1, const OnBias& p) {
        const f32x4 h0 = on_relu4(acc0 + p.b0), h1 = on_relu4(acc1 + p.b1);
        ho.set_chunk(ch, h0, h1);
        if constexpr (TRAIN) save_rows(a.save_h, l, 256, ch, h0, h1);
      };
      const float* wn = (l < 7) ? a.w + ON_OFF_N7 : a.w + ON_OFF_HF;
      run_stage<PREC, 16, 8, false, true>(a.w + (l == 6 ? ON_OFF_N6 : ON_OFF_N7), wn, 32, smem, par, h, nullptr, bias_pre(l * 256), epi, wave, lane);
      h = ho;
    }
    // ---- HF: feature (chunks 0..7, no activation) + density (row 0 of chunk 8) ----
    Act<PREC, 16> f;
    {
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1, const OnBias& p) {
        if (ch < 8) {
          const f32x4 f0 = acc0 + p.b0, f1 = acc1 + p.b1;
          f.set_chunk(ch, f0, f1);
          if constexpr (TRAIN) save_rows(a.save_f, 0, 256, ch, f0, f1);
        } else if (valid && q == 0) {
          a.density[P] = acc0[0] + p.b0[0];    // ON_B_ALPHA follows the feature bias: chunk 8's b0 of the q = 0 lanes starts there
        }
      };
      run_stage<PREC, 16, 9, false, true>(a.w + ON_OFF_HF, a.w + ON_OFF_VA, 32, smem, par, h, nullptr, bias_pre(ON_B_FEAT), epi, wave, lane);
    }
    // ---- V: cat[f, enc(view, light)] -> 128, ReLU ----
    Act<PREC, 8> hv;
    {
      float part[32];
      auto pre0 = [&](int) { return 0; };
      auto epi0 = [&](int ch, f32x4 acc0, f32x4 acc1, int) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { part[ch * 8 + r] = acc0[r]; part[ch * 8 + 4 + r] = acc1[r]; }
      };
      run_stage<PREC, 16, 4, false>(a.w + ON_OFF_VA, a.w + ON_OFF_VB, 8, smem, par, f, nullptr, pre0, epi0, wave, lane);
      Act<PREC, 4> ve;
      make_ve(ve);
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1, const OnBias& p) {
        const f32x4 h0 = on_relu4(acc0 + p.b0), h1 = on_relu4(acc1 + p.b1);
        hv.set_chunk(ch, h0, h1);
        if constexpr (TRAIN) save_rows(a.save_hv, 0, 128, ch, h0, h1);
      };
      run_stage<PREC, 4, 4, true, true>(a.w + ON_OFF_VB, a.w + ON_OFF_RGB, 16, smem, par, ve, part, bias_pre(ON_B_VIEWS), epi, wave, lane);
    }
    // ---- RGB: 3 rows (block 0, lanes q == 0 hold r = 0..2), before the sigmoid ----
    {
      auto epi = [&](int, f32x4 acc0, f32x4, const OnBias& p) {
        if (valid && q == 0) {
#pragma unroll
          for (int r = 0; r < 3; ++r) a.rgb[P * 3 + r] = acc0[r] + p.b0[r];
        }
      };
      run_stage<PREC, 8, 1, false, true>(a.w + ON_OFF_RGB, a.w + ON_OFF_N0, 12, smem, par, hv, nullptr, bias_pre(ON_B_RGB), epi, wave, lane);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// adjoint sweep:  zvbar = (Wrgb^T cbar)[hv > 0];  fbar = Wva^T zvbar;  vbar = Wvb^T zvbar;
//                 hbar_7 = Wf^T fbar + w_alpha dbar;  zbar_l = hbar_l [h_l > 0];  hbar_{l-1} = W_l^T zbar_l
//                 (layer 5: hbar_4 = W5h^T zbar_5, xbar += W5x^T zbar_5);  xbar += W0^T zbar_0
// ------------------------------------------------------------------------------------------------------------------
template <int PREC>
__global__ __launch_bounds__(MLP_THREADS, 2) void outside_adjoint_kernel(const OutsideAdjArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  int par = 0;
  dma_chunk(a.wt + ONT_OFF_RGB, smem, 4, wave, lane);
  __syncthreads();
  const float S = (PREC == 1) ? a.adj_scale : 1.0f, IS = 1.0f / S;

  for (int tg = blockIdx.x; tg < a.ntile_groups; tg += gridDim.x) {
    const long long tile = (long long)tg * WG_WAVES + wave;
    const bool tile_ok = tile * TILE_PTS < a.npts;
    const long long row = tile_ok ? tile * TILE_PTS + j : j;
    auto rows_ptr = [&](const float* base, int l, int width, int blk) {
      return reinterpret_cast<const f32x4*>(base + ((size_t)l * (size_t)a.npts + (size_t)row) * width + blk * 16 + 4 * q);
    };
    auto store_rows = [&](float* base, int l, int width, int ch, const f32x4 v0, const f32x4 v1) {
      if (tile_ok) {
        float* p = base + ((size_t)l * (size_t)a.npts + (size_t)row) * width + 4 * q;
        if constexpr (PREC == 1) {
          st_stream(reinterpret_cast<f32x4*>(p + (2 * ch) * 16), v0 * IS);
          st_stream(reinterpret_cast<f32x4*>(p + (2 * ch + 1) * 16), v1 * IS);
        } else {
          st_stream(reinterpret_cast<f32x4*>(p + (2 * ch) * 16), v0);
          st_stream(reinterpret_cast<f32x4*>(p + (2 * ch + 1) * 16), v1);
        }
      }
    };
    struct HPre { f32x4 h0, h1; };
    auto mask = [](const f32x4 h, const f32x4 g) {
      return f32x4{h[0] > 0.0f ? g[0] : 0.0f, h[1] > 0.0f ? g[1] : 0.0f, h[2] > 0.0f ? g[2] : 0.0f, h[3] > 0.0f ? g[3] : 0.0f};
    };
    const float db = a.dbar[row] * S;

    // ---- TRGB: 3 -> 128 (the adjoint of the 3 outputs sits in block 0, lanes q == 0, registers 0..2) ----
    Act<PREC, 2> c3;
    {
      float o[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) o[r] = (q == 0 && r < 3) ? a.cbar[row * 3 + r] * S : 0.0f;
      c3.set_chunk(0, o);
    }
    Act<PREC, 8> zv;
    {
      auto pre = [&](int ch) {
        HPre p;
        p.h0 = ld_stream(rows_ptr(a.save_hv, 0, 128, 2 * ch));
        p.h1 = ld_stream(rows_ptr(a.save_hv, 0, 128, 2 * ch + 1));
        return p;
      };
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1, const HPre& p) {
        const f32x4 z0 = mask(p.h0, acc0), z1 = mask(p.h1, acc1);
        store_rows(a.zvbar, 0, 128, ch, z0, z1);
        zv.set_chunk(ch, z0, z1);
      };
      run_stage<PREC, 2, 4, false, true>(a.wt + ONT_OFF_RGB, a.wt + ONT_OFF_VA, 16, smem, par, c3, nullptr, pre, epi, wave, lane);
    }
    // ---- TVA: fbar;  TVB: vbar ----
    Act<PREC, 16> h;
    {
      auto pre = [&](int) { return 0; };
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1, int) {
        store_rows(a.fbar, 0, 256, ch, acc0, acc1);
        h.set_chunk(ch, acc0, acc1);
      };
      run_stage<PREC, 8, 8, false>(a.wt + ONT_OFF_VA, a.wt + ONT_OFF_VB, 16, smem, par, zv, nullptr, pre, epi, wave, lane);
      auto epv = [&](int ch, f32x4 acc0, f32x4 acc1, int) { store_rows(a.vbar, 0, ON_V, ch, acc0, acc1); };
      run_stage<PREC, 8, 2, false>(a.wt + ONT_OFF_VB, a.wt + ONT_OFF_HF, 32, smem, par, zv, nullptr, pre, epv, wave, lane);
    }
    // ---- THF: hbar_7 = Wf^T fbar + w_alpha dbar -> zbar_7 ----
    {
      struct HWPre { f32x4 h0, h1, w0, w1; };
      auto pre = [&](int ch) {
        HWPre p;
        p.h0 = ld_stream(rows_ptr(a.save_h, 7, 256, 2 * ch));
        p.h1 = ld_stream(rows_ptr(a.save_h, 7, 256, 2 * ch + 1));
        p.w0 = *reinterpret_cast<const f32x4*>(a.walpha + (2 * ch) * 16 + 4 * q);
        p.w1 = *reinterpret_cast<const f32x4*>(a.walpha + (2 * ch + 1) * 16 + 4 * q);
        return p;
      };
      Act<PREC, 16> ho;
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1, const HWPre& p) {
        const f32x4 z0 = mask(p.h0, acc0 + p.w0 * db), z1 = mask(p.h1, acc1 + p.w1 * db);
        store_rows(a.zbar, 7, 256, ch, z0, z1);
        ho.set_chunk(ch, z0, z1);
      };
      run_stage<PREC, 16, 8, false, true>(a.wt + ONT_OFF_HF, a.wt + ONT_OFF_T7, 32, smem, par, h, nullptr, pre, epi, wave, lane);
      h = ho;
    }
    // ---- T7, T6: zbar_6, zbar_5 ----
    for (int l = 7; l >= 6; --l) {
      auto pre = [&](int ch) {
        HPre p;
        p.h0 = ld_stream(rows_ptr(a.save_h, l - 1, 256, 2 * ch));
        p.h1 = ld_stream(rows_ptr(a.save_h, l - 1, 256, 2 * ch + 1));
        return p;
      };
      Act<PREC, 16> ho;
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1, const HPre& p) {
        const f32x4 z0 = mask(p.h0, acc0), z1 = mask(p.h1, acc1);
        store_rows(a.zbar, l - 1, 256, ch, z0, z1);
        ho.set_chunk(ch, z0, z1);
      };
      const float* wn = (l == 7) ? a.wt + ONT_OFF_T6 : a.wt + ONT_OFF_T5X;
      run_stage<PREC, 16, 8, false, true>(a.wt + (l == 7 ? ONT_OFF_T7 : ONT_OFF_T6), wn, 32, smem, par, h, nullptr, pre, epi, wave, lane);
      h = ho;
    }
    // ---- T5x: the skip's share of xbar (parked in the xbar rows until T0 adds its own);  T5h: zbar_4 ----
    {
      auto pre = [&](int) { return 0; };
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1, int) { store_rows(a.xbar, 0, ON_X, ch, acc0, acc1); };
      run_stage<PREC, 16, 3, false>(a.wt + ONT_OFF_T5X, a.wt + ONT_OFF_T5H, 32, smem, par, h, nullptr, pre, epi, wave, lane);
    }
    for (int l = 5; l >= 1; --l) {
      auto pre = [&](int ch) {
        HPre p;
        p.h0 = ld_stream(rows_ptr(a.save_h, l - 1, 256, 2 * ch));
        p.h1 = ld_stream(rows_ptr(a.save_h, l - 1, 256, 2 * ch + 1));
        return p;
      };
      Act<PREC, 16> ho;
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1, const HPre& p) {
        const f32x4 z0 = mask(p.h0, acc0), z1 = mask(p.h1, acc1);
        store_rows(a.zbar, l - 1, 256, ch, z0, z1);
        ho.set_chunk(ch, z0, z1);
      };
      const float* wc = (l == 5) ? a.wt + ONT_OFF_T5H : a.wt + ont_off_T(l);
      const float* wn = (l > 1) ? a.wt + ont_off_T(l - 1) : a.wt + ONT_OFF_T0;
      run_stage<PREC, 16, 8, false, true>(wc, wn, 32, smem, par, h, nullptr, pre, epi, wave, lane);
      h = ho;
    }
    // ---- T0: xbar = W0^T zbar_0 + the skip's share ----
    {
      // (this wave's own stores of T5x, five stages ago: plain loads, same lanes, same addresses)
      auto pre = [&](int ch) {
        HPre p;
        p.h0 = *rows_ptr(a.xbar, 0, ON_X, 2 * ch);
        p.h1 = *rows_ptr(a.xbar, 0, ON_X, 2 * ch + 1);
        return p;
      };
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1, const HPre& p) { store_rows(a.xbar, 0, ON_X, ch, acc0 + p.h0 * S, acc1 + p.h1 * S); };
      run_stage<PREC, 16, 3, false, true>(a.wt + ONT_OFF_T0, a.wt + ONT_OFF_RGB, 4, smem, par, h, nullptr, pre, epi, wave, lane);
    }
  }
}

}  // namespace nrh
