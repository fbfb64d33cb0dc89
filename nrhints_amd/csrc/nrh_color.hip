// Reflectance MLP on gfx950 (reference: fields/reflectance_network.py:68-96, called from
// models/neus_hint_model.py:626): 361 -> 4 x (256, ReLU) -> 3 -> sigmoid, one colour per sample.
//
// The 361-wide input of the reference, cat[pts 3, enc4(view) 27, normal 3, enc4(pl) 27, feat 256, enc4(vis) 9,
// enc4(cue) 36], is never materialised.  Layer 0 is split along K:
//   C0a  feat (256)  - read straight from the D-layout tiles the SDF kernel wrote; they ARE the MFMA B operand
//   C0b  the other 105 inputs (-> 128): 6 per-sample values (point, unit normal) computed here and 99 per-ray
//        values (encodings of view dir, light position, visibility hint, specular cue) read from a per-ray
//        table that one wave per ray filled once (nrh_rays.hip) instead of 128 times
// The columns of W0 are permuted on the host to this order.  C1..C3 are plain 256x256 ReLU layers, C4 has 3 rows.
#include "nrh_mlp.h"

namespace nrh {

struct ColorArgs {
  const float* w;        // packed (COL_PACKED_FLOATS)
  const float* b;        // [4][256] + [16]
  const float* feat;     // [ntiles][16][64][4] D-layout tiles; TRAIN: row-major [npts][256]
  const float* ro;       // [nrays,3]
  const float* rd;       // [nrays,3]
  const float* tmid;     // [nrays,128]
  const float* nhat;     // [npts,3] normal fed to the net: unit normals (NormalizedAnalytic) or raw gradients (Analytic)
  const float* raymisc;  // [nrays,RAYMISC_STRIDE]
  float* color;          // [npts,3]
  long long npts;
  int ntile_groups;
  // TRAIN (forward of a training step): sample points given directly, and what the adjoint + dW GEMMs need, row-major
  const float* pts;      // [npts,3]
  float* save_h;         // [4][npts][256]  ReLU outputs of layers 0..3
  float* save_misc;      // [npts][16*MKB]  the non-feature part of the layer-0 input in kernel order
  void* save_h16;        // optional (PREC 1, TRAIN): fp16 half-tiled [4][npts][256] (csrc/nrh_mlp.h half_ptr) - layers 0..2 of the ReLU
                         // outputs go HERE INSTEAD of save_h: the adjoint sweep reads them as masks, nrh_dw_gemm as half operands
  int misc_shift;        // sample P reads row P >> misc_shift of raymisc: 7 = one row per ray; smaller for the partial shadow
                         // mode (n_shadow_importance_clip: one row per group of 128 / clip consecutive samples)
};
// adjoint sweep of the reflectance net (csrc/nrh_color.hip color_adjoint_kernel)
struct ColorAdjArgs {
  const float* wt;       // packed transposed stages  W4^T | W3^T | W2^T | W1^T | W0feat^T | W0misc^T
  const float* zbar4;    // [npts,3]   adjoint of the pre-sigmoid output
  const float* save_h;   // [4][npts][256]
  float* zbar;           // [4][npts][256]  adjoints of the pre-ReLU outputs of layers 0..3
  float* fbar;           // [npts][256]     adjoint of the feature input (feeds the SDF value sweep)
  float* mbar;           // [npts][16*MKB]  adjoint of the non-feature input part
  const void* save_h16;  // optional (f16x3): layers 0..2 of the ReLU outputs as fp16 half-tiled (ColorArgs.save_h16) - read instead of save_h
  void* zbar16;          // optional (f16x3): layers 1..3 of zbar leave as fp16 half-tiled, times S * half_gain, INSTEAD of zbar
  float half_gain;       // a power of two: the seeds are bounded (|zbar4| <= 1 / (12 rays)), so the stored range needs no measurement
  long long npts;
  int ntile_groups;
  float adj_scale;       // f16x3 only: a power of two S.  The adjoint chain runs on S * (the seeds) and its outputs leave as 1 / S *
                         // (the result): the loss is normalised by the ray count (pipelines/base_pipeline.py:57), so at 1 024 rays
                         // per step the adjoints are ~ 1e-3 of a single ray's and their fp16 halves (absolute floor 3e-11 under
                         // 6e-5) lose up to 6e-3 of a gradient tensor's scale - measured against the reference's 1 024-ray step,
                         // profiles/r05/train1024_diag*.log.  The host passes 2^round(log2(rays)): batch-size independent ranges.
};

struct BiasV {
  f32x4 b0, b1;
};
__device__ __forceinline__ f32x4 relu4(const f32x4 x) {
  return f32x4{fmaxf(x[0], 0.0f), fmaxf(x[1], 0.0f), fmaxf(x[2], 0.0f), fmaxf(x[3], 0.0f)};
}

// FUSED (evaluation with the wide SDF kernels): the tiles in `feat` already hold W0feat * feature - the SDF kernel's feature
// head is linear and feeds this linear block directly, so the two matrices are multiplied at pack time
// (packing32.pack_sdf32_fused) and stage C0a (a quarter of this kernel's MFMA work) disappears.
template <int PREC, int MKB, bool TRAIN = false, bool FUSED = false>
__global__ __launch_bounds__(MLP_THREADS, 2) void color_kernel(const ColorArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  int par = 0;
  if constexpr (FUSED) dma_chunk(a.w + COL_OFF_C0B, smem, 2 * MKB, wave, lane);
  else dma_chunk(a.w + COL_OFF_C0A, smem, 32, wave, lane);
  __syncthreads();

  for (int tg = blockIdx.x; tg < a.ntile_groups; tg += gridDim.x) {
    const long long tile = (long long)tg * WG_WAVES + wave;
    const long long P = tile * TILE_PTS + j;
    const bool valid = P < a.npts;
    const long long Pc = valid ? P : a.npts - 1;
    const long long tilec = Pc / TILE_PTS;
    const long long ray = Pc >> 7;  // 128 samples per ray
    const bool tile_ok = tile * TILE_PTS < a.npts;   // TRAIN: whole tiles only (npts % 16 == 0)
    auto save_rows = [&](float* base, int l, int width, int ch, const f32x4 v0, const f32x4 v1) {
      if (tile_ok) {
        float* p = base + ((size_t)l * (size_t)a.npts + (size_t)Pc) * width + 4 * q;
        st_stream(reinterpret_cast<f32x4*>(p + (2 * ch) * 16), v0);
        st_stream(reinterpret_cast<f32x4*>(p + (2 * ch + 1) * 16), v1);
      }
    };

    auto save_h_rows = [&](int l, int ch, const f32x4 v0, const f32x4 v1) {
      if constexpr (PREC == 1) {
        if (a.save_h16 && l < 3) {          // (wave-uniform: a kernel argument and the layer)
          if (tile_ok) st_stream(half_ptr<true>(a.save_h16, l, a.npts, Pc, ch, q), pack_half8(v0, v1));
          return;
        }
      }
      save_rows(a.save_h, l, 256, ch, v0, v1);
    };

    // ---- C0a: feature part ----
    Act<PREC, 16> h;
    float part[64];
    if constexpr (FUSED) {
      // the tile holds this lane's share of W0feat * feature: rows (2 ch) * 16 + 4 q + r and (2 ch + 1) * 16 + 4 q + r
      const float* ft = a.feat + (size_t)tilec * (16 * 256) + lane * 4;
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) {
        const f32x4 v0 = ld_stream(reinterpret_cast<const f32x4*>(ft + (2 * ch) * 256));
        const f32x4 v1 = ld_stream(reinterpret_cast<const f32x4*>(ft + (2 * ch + 1) * 256));
#pragma unroll
        for (int r = 0; r < 4; ++r) { part[ch * 8 + r] = v0[r]; part[ch * 8 + 4 + r] = v1[r]; }
      }
    } else {
      {
        const float* ft = TRAIN ? a.feat + (size_t)Pc * 256 + 4 * q : a.feat + (size_t)tilec * (16 * 256) + lane * 4;
        const int bs = TRAIN ? 16 : 256;   // floats between consecutive 16-feature blocks
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
          const f32x4 v0 = ld_stream(reinterpret_cast<const f32x4*>(ft + (2 * ch) * bs));
          const f32x4 v1 = ld_stream(reinterpret_cast<const f32x4*>(ft + (2 * ch + 1) * bs));
          const float o[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          h.set_chunk(ch, o);
        }
      }
      auto pre = [&](int) { return 0; };
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1, int) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { part[ch * 8 + r] = acc0[r]; part[ch * 8 + 4 + r] = acc1[r]; }
      };
      run_stage<PREC, 16, 8, false>(a.w + COL_OFF_C0A, a.w + COL_OFF_C0B, 2 * MKB, smem, par, h, nullptr, pre, epi, wave, lane);
    }
    // ---- C0b: per-sample + per-ray part; entry m = 16b + 4q + r of [p, n, raymisc[0..98]] ----
    Act<PREC, MKB> misc;
    {
      const float tt = a.tmid[Pc];
      float pn[6];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        pn[c] = TRAIN ? a.pts[Pc * 3 + c] : a.ro[ray * 3 + c] + a.rd[ray * 3 + c] * tt;
        pn[3 + c] = a.nhat[Pc * 3 + c];
      }
      const float* rm = a.raymisc + (Pc >> a.misc_shift) * RAYMISC_STRIDE;
#pragma unroll
      for (int ch = 0; ch < MKB / 2; ++ch) {
        float o[8];
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) {
          const int b = 2 * ch + (r8 >> 2), r = r8 & 3;
          const int m = b * 16 + 4 * q + r;
          float v;
          if (b == 0) {
            // m in [0,16): the first 6 are per-sample values, selected with compile-time candidates per q
            const float per = sel_q<6>(pn, r, q);
            const float ld = rm[(m >= 6) ? m - 6 : 0];
            v = (m < 6) ? per : ld;
          } else {
            v = (m < col_misc(MKB)) ? rm[(m < col_misc(MKB)) ? m - 6 : 0] : 0.0f;
          }
          o[r8] = v;
        }
        misc.set_chunk(ch, o);
        if constexpr (TRAIN) save_rows(a.save_misc, 0, 16 * MKB, ch, f32x4{o[0], o[1], o[2], o[3]}, f32x4{o[4], o[5], o[6], o[7]});
      }
    }
    {
      auto pre = [&](int ch) {
        BiasV p;
        p.b0 = *reinterpret_cast<const f32x4*>(a.b + (2 * ch) * 16 + 4 * q);
        p.b1 = *reinterpret_cast<const f32x4*>(a.b + (2 * ch + 1) * 16 + 4 * q);
        return p;
      };
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1, const BiasV& p) {
        const f32x4 h0 = relu4(acc0 + p.b0), h1 = relu4(acc1 + p.b1);
        h.set_chunk(ch, h0, h1);
        if constexpr (TRAIN) save_h_rows(0, ch, h0, h1);
      };
      run_stage<PREC, MKB, 8, true, true>(a.w + COL_OFF_C0B, a.w + col_off_C(1, MKB), 32, smem, par, misc, part, pre, epi, wave, lane);
    }
    // ---- C1..C3 ----
    for (int l = 1; l <= 3; ++l) {
      Act<PREC, 16> ho;
      auto pre = [&](int ch) {
        BiasV p;
        p.b0 = *reinterpret_cast<const f32x4*>(a.b + l * 256 + (2 * ch) * 16 + 4 * q);
        p.b1 = *reinterpret_cast<const f32x4*>(a.b + l * 256 + (2 * ch + 1) * 16 + 4 * q);
        return p;
      };
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1, const BiasV& p) {
        const f32x4 h0 = relu4(acc0 + p.b0), h1 = relu4(acc1 + p.b1);
        ho.set_chunk(ch, h0, h1);
        if constexpr (TRAIN) save_h_rows(l, ch, h0, h1);
      };
      const float* wn = (l < 3) ? a.w + col_off_C(l + 1, MKB) : a.w + col_off_C4(MKB);
      run_stage<PREC, 16, 8, false, true>(a.w + col_off_C(l, MKB), wn, 32, smem, par, h, nullptr, pre, epi, wave, lane);
      h = ho;
    }
    // ---- C4: 3 output rows (block 0, lanes q == 0 hold r = 0..2) + sigmoid ----
    {
      auto pre = [&](int) {
        BiasV p;
        p.b0 = *reinterpret_cast<const f32x4*>(a.b + 4 * 256 + 4 * q);
        p.b1 = p.b0;
        return p;
      };
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1, const BiasV& p) {
        (void)acc1;
        if (valid && q == 0) {
#pragma unroll
          for (int r = 0; r < 3; ++r) a.color[P * 3 + r] = sigmoidf_(acc0[r] + p.b0[r]);
        }
      };
      if constexpr (FUSED)
        run_stage<PREC, 16, 1, false>(a.w + col_off_C4(MKB), a.w + COL_OFF_C0B, 2 * MKB, smem, par, h, nullptr, pre, epi, wave, lane);
      else
        run_stage<PREC, 16, 1, false>(a.w + col_off_C4(MKB), a.w + COL_OFF_C0A, 32, smem, par, h, nullptr, pre, epi, wave, lane);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// adjoint sweep:  zbar_3 = (W4^T zbar4) [h4 > 0];  zbar_{l-1} = (W_l^T zbar_l) [h_l > 0];  fbar = W0feat^T zbar_0;
//                 mbar = W0misc^T zbar_0.   Weight gradients are GEMMs over zbar / save_h on the host side.
// ------------------------------------------------------------------------------------------------------------------
constexpr int COLT_C4_FLOATS = 8 * 2 * 2 * 256;   // W4^T: 256 rows, K = 32 (3 used)
__host__ __device__ constexpr int colt_off_T(int l) { return COLT_C4_FLOATS + (3 - l) * SDF_REG_FLOATS; }  // l = 3, 2, 1
constexpr int COLT_OFF_T0A = COLT_C4_FLOATS + 3 * SDF_REG_FLOATS;
constexpr int COLT_OFF_T0B = COLT_C4_FLOATS + 4 * SDF_REG_FLOATS;
__host__ __device__ constexpr int colt_packed_floats(int mkb) { return COLT_OFF_T0B + (mkb / 2) * 2 * 16 * 256; }

template <int PREC, int MKB>
__global__ __launch_bounds__(MLP_THREADS, 2) void color_adjoint_kernel(const ColorAdjArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  int par = 0;
  dma_chunk(a.wt, smem, 4, wave, lane);
  __syncthreads();
  // (ColorAdjArgs.adj_scale) everything between the seeds and the stores is linear in the seeds (ReLU masks come from save_h)
  const float S = (PREC == 1) ? a.adj_scale : 1.0f, IS = 1.0f / S;

  for (int tg = blockIdx.x; tg < a.ntile_groups; tg += gridDim.x) {
    const long long tile = (long long)tg * WG_WAVES + wave;
    const bool tile_ok = tile * TILE_PTS < a.npts;
    const long long row = tile_ok ? tile * TILE_PTS + j : j;
    auto rows_ptr = [&](const float* base, int l, int blk) {
      return reinterpret_cast<const f32x4*>(base + ((size_t)l * (size_t)a.npts + (size_t)row) * 256 + blk * 16 + 4 * q);
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
    // ReLU outputs of layer l as masks: float32 rows, or (16-bit hand-offs, layers 0..2) the fp16 array - only the sign is used
    auto load_h = [&](int l, int ch) {
      HPre p;
      if constexpr (PREC == 1) {
        if (a.save_h16 && l < 3) {
          const f16x8_t hv = ld_stream(half_ptr<true>(const_cast<void*>(a.save_h16), l, a.npts, row, ch, q));
          p.h0 = f32x4{(float)hv[0], (float)hv[1], (float)hv[2], (float)hv[3]};
          p.h1 = f32x4{(float)hv[4], (float)hv[5], (float)hv[6], (float)hv[7]};
          return p;
        }
      }
      p.h0 = ld_stream(rows_ptr(a.save_h, l, 2 * ch));
      p.h1 = ld_stream(rows_ptr(a.save_h, l, 2 * ch + 1));
      return p;
    };
    auto store_zbar = [&](int l, int ch, const f32x4 z0, const f32x4 z1) {
      if constexpr (PREC == 1) {
        if (a.zbar16 && l >= 1) {
          if (tile_ok) st_stream(half_ptr<true>(a.zbar16, l, a.npts, row, ch, q), pack_half8(z0 * a.half_gain, z1 * a.half_gain));
          return;
        }
      }
      store_rows(a.zbar, l, 256, ch, z0, z1);
    };

    // ---- T4: 3 -> 256 (the adjoint of the 3 outputs sits in block 0, lanes q == 0, registers 0..2) ----
    Act<PREC, 2> z4;
    {
      float o[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) o[r] = (q == 0 && r < 3) ? a.zbar4[row * 3 + r] * S : 0.0f;
      z4.set_chunk(0, o);
    }
    Act<PREC, 16> h;
    {
      auto pre = [&](int ch) { return load_h(3, ch); };
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1, const HPre& p) {
        f32x4 z0, z1;
#pragma unroll
        for (int r = 0; r < 4; ++r) { z0[r] = p.h0[r] > 0.0f ? acc0[r] : 0.0f; z1[r] = p.h1[r] > 0.0f ? acc1[r] : 0.0f; }
        store_zbar(3, ch, z0, z1);
        h.set_chunk(ch, z0, z1);
      };
      run_stage<PREC, 2, 8, false, true>(a.wt, a.wt + colt_off_T(3), 32, smem, par, z4, nullptr, pre, epi, wave, lane);
    }
    // ---- T3, T2, T1 ----
    for (int l = 3; l >= 1; --l) {
      Act<PREC, 16> ho;
      auto pre = [&](int ch) { return load_h(l - 1, ch); };
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1, const HPre& p) {
        f32x4 z0, z1;
#pragma unroll
        for (int r = 0; r < 4; ++r) { z0[r] = p.h0[r] > 0.0f ? acc0[r] : 0.0f; z1[r] = p.h1[r] > 0.0f ? acc1[r] : 0.0f; }
        store_zbar(l - 1, ch, z0, z1);
        ho.set_chunk(ch, z0, z1);
      };
      const float* wn = (l > 1) ? a.wt + colt_off_T(l - 1) : a.wt + COLT_OFF_T0A;
      run_stage<PREC, 16, 8, false, true>(a.wt + colt_off_T(l), wn, 32, smem, par, h, nullptr, pre, epi, wave, lane);
      h = ho;
    }
    // ---- T0a: adjoint of the feature input;  T0b: adjoint of the other inputs ----
    {
      auto pre = [&](int) { return 0; };
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1, int) { store_rows(a.fbar, 0, 256, ch, acc0, acc1); };
      run_stage<PREC, 16, 8, false>(a.wt + COLT_OFF_T0A, a.wt + COLT_OFF_T0B, 32, smem, par, h, nullptr, pre, epi, wave, lane);
    }
    {
      auto pre = [&](int) { return 0; };
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1, int) { store_rows(a.mbar, 0, 16 * MKB, ch, acc0, acc1); };
      run_stage<PREC, 16, MKB / 2, false>(a.wt + COLT_OFF_T0B, a.wt, 4, smem, par, h, nullptr, pre, epi, wave, lane);
    }
  }
}

}  // namespace nrh
