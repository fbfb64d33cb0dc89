// SDF network on gfx950: value, appearance feature and analytic d(sdf)/dx in ONE pass per point.
//
// Replaces, per point, the reference's  SDFNetwork.forward  (fields/sdf_field.py:106-123),  .sdf  (:125-126)
// and  .gradient  (:136-148, autograd)  - i.e. up to three full forwards plus an autograd sweep
// (models/neus_hint_model.py:504, :335, :336) - by one forward that keeps sigma'(z) = sigmoid(100 z) of
// every activation and one hand-written reverse chain through W^T.
//
// Layer plan (all GEMMs are 256x256 except the first/last):
//   L0   39(->64) -> 256   softplus100        in = NeRF encoding of 3x, computed in registers
//   L1,L2          256 -> 256
//   L3             256 -> 217(->256, zero rows); outputs 217..255 are REPLACED by the 39 embedding values, which
//                  reproduces cat([h, embed]) of the skip connection with no data movement (fields/sdf_field.py:114)
//   L4..L7         256 -> 256   (W4 pre-scaled by 1/sqrt(2))
//   head           sdf = (w_s . h + b_s) / 3              (VALU dot + 2 shuffles, folded into L7's epilogue)
//   FEAT           256 -> 256, no activation              (MODE 2 only)
//   R7..R1         g <- W_l^T (sigma'_l * g)              (MODE >= 1)
//   R0             39(->64) <- 256, then d/dx of the encoding
// sigma' lives in a per-wave global scratch between the forward and the reverse sweep (2 048 values per point do not
// fit on chip): fp32 in PREC 0, unorm16 in PREC 1 (absolute error <= 7.6e-6 on a factor in [0,1]; halves the only
// non-trivial memory traffic of this kernel so that it stays in the 256 MiB Infinity Cache).
#include "nrh_mlp.h"

namespace nrh {

struct SdfArgs {
  const float* w;      // packed stages (SDF_PACKED_FLOATS floats, or the same number of fp16 hi/lo pairs)
  const float* b;      // [9][256]
  const float* head;   // [257]
  const float* ro;     // [nrays,3]
  const float* rd;     // [nrays,3]
  const float* t;      // ray parameter of point (ray, j): t[ray * t_stride + j]
  float* sdf;          // sdf[ray * sdf_stride + j]
  float* grad;         // [npts,3] (MODE >= 1)
  float* feat;         // [ntiles][16][64][4] D-layout tiles (MODE 2) / row-major [npts][256] (MODE 3)
  float* scratch;      // gridDim.x * WG_WAVES * SDF_SCRATCH_FLOATS_PER_WAVE (MODE 1, 2)
  // MODE 3 (training forward): what the hand-derived backward needs, row-major so that library GEMMs can read it
  float* save_h;       // [8][npts][256]  h_l = softplus(z_l) (layer 3: after the skip substitution, i.e. x_4)
  float* save_s1;      // [8][npts][256]  sigma'_l = sigmoid(100 z_l) (0 on substituted entries); replaces the scratch
  float* save_t;       // [8][npts][256]  t_l = sigma'_l * a_{l+1}, the reverse-chain stage inputs (t_7 = sigma'_7 w_s/3)
  float* save_ge;      // [npts][128]     cols 0..63 a_0 (39 used), cols 64..111 = a_4[208..255] (skip part from col 73)
  void* save_h16;      // optional (PREC 1): fp16 half-tiled [8][npts][256] - layers 0..6 of h go HERE INSTEAD of save_h (nrh_mlp.h half_ptr)
  void* save_t16;      // optional (PREC 1): fp16 half-tiled copy of layers 1..7 of t (save_t keeps all 8: the tangent sweep reads them)
  int t16_only;        // with save_t16: layers 1..6 of t go to save_t16 ONLY (the tangent sweep is told to read them there)
  long long npts;
  int n_per_ray;
  int t_stride;
  int sdf_stride;
  int ntile_groups;    // ceil(npts / (16 * WG_WAVES))
};

__device__ __forceinline__ uint32_t unorm16x2(float a, float b) {
  // v_cvt_pknorm_u16_f32: both values clamped to [0,1], scaled by 65535, rounded to nearest, packed - one VALU op
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  const u16x2 p = __builtin_amdgcn_cvt_pknorm_u16(a, b);
  return __builtin_bit_cast(uint32_t, p);
}

// the 8 sigma' values of chunk ch of layer l (l < 7): two float4 (PREC 0) or one uint4 of unorm16 pairs (PREC 1)
struct PreV {
  f32x4 a0, a1;  // forward: bias of blocks 2ch, 2ch+1;  reverse: sigma' (PREC 0) / a0 = packed unorm16 (PREC 1)
  f32x4 b0, b1;  // layer 7 only: sdf-head weights of the two blocks
};

template <int PREC>
__device__ __forceinline__ void dsig_store(float* scr, int l, int ch, int lane, const f32x4 d0, const f32x4 d1) {
  if constexpr (PREC == 0) {
    st_stream(reinterpret_cast<f32x4*>(scr + ((l * 16 + 2 * ch) * 64 + lane) * 4), d0);
    st_stream(reinterpret_cast<f32x4*>(scr + ((l * 16 + 2 * ch + 1) * 64 + lane) * 4), d1);
  } else {
    const u32x4 v = {unorm16x2(d0[0], d0[1]), unorm16x2(d0[2], d0[3]), unorm16x2(d1[0], d1[1]), unorm16x2(d1[2], d1[3])};
    st_stream(reinterpret_cast<u32x4*>(scr + l * 2048 + (ch * 64 + lane) * 4), v);
  }
}
template <int PREC>
__device__ __forceinline__ void dsig_issue(const float* scr, int l, int ch, int lane, PreV& p) {
  if constexpr (PREC == 0) {
    p.a0 = ld_stream(reinterpret_cast<const f32x4*>(scr + ((l * 16 + 2 * ch) * 64 + lane) * 4));
    p.a1 = ld_stream(reinterpret_cast<const f32x4*>(scr + ((l * 16 + 2 * ch + 1) * 64 + lane) * 4));
  } else {
    p.a0 = __builtin_bit_cast(f32x4, ld_stream(reinterpret_cast<const u32x4*>(scr + l * 2048 + (ch * 64 + lane) * 4)));
  }
}
template <int PREC>
__device__ __forceinline__ void dsig_decode(const PreV& p, f32x4& d0, f32x4& d1) {
  if constexpr (PREC == 0) {
    d0 = p.a0;
    d1 = p.a1;
  } else {
    const u32x4 v = __builtin_bit_cast(u32x4, p.a0);
    d0 = f32x4{(float)(v[0] & 0xffffu), (float)(v[0] >> 16), (float)(v[1] & 0xffffu), (float)(v[1] >> 16)} * (1.0f / 65535.0f);
    d1 = f32x4{(float)(v[2] & 0xffffu), (float)(v[2] >> 16), (float)(v[3] & 0xffffu), (float)(v[3] >> 16)} * (1.0f / 65535.0f);
  }
}
// t_7 = sigma'_7 * w_s / 3 is not confined to [0,1]: always fp32
template <int PREC>
__device__ __forceinline__ float* t7_ptr(float* scr, int blk, int lane) {
  return scr + ((PREC == 0) ? 7 * 16 * 256 : 7 * 2048) + (blk * 64 + lane) * 4;
}

// MODE 0: sdf | 1: sdf + gradient | 2: + feature tiles (inference render) | 3: training forward (= 2 with row-major saves)
template <int MODE, int PREC>
__global__ __launch_bounds__(MLP_THREADS, 2) void sdf_kernel(const SdfArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  int par = 0;
  constexpr bool TRAIN = MODE == 3;
  float* const scr = (MODE == 1 || MODE == 2) ? a.scratch + (size_t)(blockIdx.x * WG_WAVES + wave) * SDF_SCRATCH_FLOATS_PER_WAVE : nullptr;
  constexpr bool WANT_D = MODE >= 1;

  dma_chunk(a.w + SDF_OFF_L0, smem, 8, wave, lane);
  __syncthreads();

  for (int tg = blockIdx.x; tg < a.ntile_groups; tg += gridDim.x) {
    const long long tile = (long long)tg * WG_WAVES + wave;
    const long long P = tile * TILE_PTS + j;
    const bool valid = P < a.npts;
    const long long Pc = valid ? P : a.npts - 1;
    const long long ray = Pc / a.n_per_ray;
    const int jj = (int)(Pc - ray * a.n_per_ray);
    const float tt = a.t[ray * a.t_stride + jj];
    // training saves: whole tiles only (npts % 16 == 0 is checked by the host); tiles past the end write nothing
    const bool tile_ok = tile * TILE_PTS < a.npts;
    const long long row = tile_ok ? tile * TILE_PTS + j : j;
    auto ds_store = [&](int l, int ch, const f32x4 d0, const f32x4 d1) {
      if constexpr (TRAIN) {
        if (tile_ok) {
          st_stream(reinterpret_cast<f32x4*>(arr_ptr<ARR_S1, PREC == 1>(a.save_s1, l, a.npts, row, 2 * ch, q)), d0);
          st_stream(reinterpret_cast<f32x4*>(arr_ptr<ARR_S1, PREC == 1>(a.save_s1, l, a.npts, row, 2 * ch + 1, q)), d1);
        }
      } else {
        dsig_store<PREC>(scr, l, ch, lane, d0, d1);
      }
    };
    auto save_rows = [&](auto AT, float* base, int l, int ch, const f32x4 v0, const f32x4 v1) {
      constexpr int ARR = decltype(AT)::value;
      if constexpr (PREC == 1 && (ARR == ARR_H || ARR == ARR_T)) {
        // 16-bit hand-offs of the weight-gradient operands (wave-uniform branches on kernel arguments)
        if (ARR == ARR_H && a.save_h16 && l < 7) {
          if (tile_ok) st_stream(half_ptr<true>(a.save_h16, l, a.npts, row, ch, q), pack_half8(v0, v1));
          return;
        }
        if (ARR == ARR_T && a.save_t16 && l >= 1) {
          if (tile_ok) st_stream(half_ptr<true>(a.save_t16, l, a.npts, row, ch, q), pack_half8(v0, v1));
          if (a.t16_only && l < 7) return;
        }
      }
      if (tile_ok) {
        st_stream(reinterpret_cast<f32x4*>(arr_ptr<ARR, PREC == 1>(base, l, a.npts, row, 2 * ch, q)), v0);
        st_stream(reinterpret_cast<f32x4*>(arr_ptr<ARR, PREC == 1>(base, l, a.npts, row, 2 * ch + 1, q)), v1);
      }
    };
    float x3[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) x3[c] = (a.ro[ray * 3 + c] + a.rd[ray * 3 + c] * tt) * 3.0f;  // inputs * scale

    // ---- L0: embedding (entry 16b + 4q + r in register b*4 + r; blocks 0..2 hold the 39 entries) -> 256 ----
    Act<PREC, 4> emb;
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
      float o[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int b = 2 * c2 + (r >> 2);
        o[r] = (b < 3) ? nerf_enc_entry_q<3, 6>(x3, b * 16 + (r & 3), q) : 0.0f;
      }
      emb.set_chunk(c2, o);
    }

    Act<PREC, 16> h;
    {
      auto pre = [&](int ch) {
        PreV p;
        p.a0 = *reinterpret_cast<const f32x4*>(a.b + (2 * ch) * 16 + 4 * q);
        p.a1 = *reinterpret_cast<const f32x4*>(a.b + (2 * ch + 1) * 16 + 4 * q);
        return p;
      };
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1, const PreV& p) {
        f32x4 h0, h1, d0, d1;
        softplus100_4<WANT_D>(acc0 + p.a0, h0, d0);
        softplus100_4<WANT_D>(acc1 + p.a1, h1, d1);
        h.set_chunk(ch, h0, h1);
        if (MODE >= 1) ds_store(0, ch, d0, d1);
        if constexpr (TRAIN) save_rows(ArrTag<ARR_H>(), a.save_h, 0, ch, h0, h1);
      };
      run_stage<PREC, 4, 8, false, true>(a.w + SDF_OFF_L0, a.w + sdf_off_L(1), 32, smem, par, emb, nullptr, pre, epi, wave, lane);
    }

    // ---- steps 1..15: L1..L7, FEAT, R7..R1 share one 256x256 body ----
    float skip[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) skip[i] = 0.0f;
    const int last_step = (MODE == 0) ? 7 : 15;
    float head_part = 0.0f;
    for (int s = 1; s <= last_step; ++s) {
      if (MODE == 1 && s == 8) continue;  // no feature head
      if (MODE == 0 && s > 7) break;
      const float* wcur;
      const float* wnxt;
      int npc = 32;
      if (s <= 7) wcur = a.w + sdf_off_L(s);
      else if (s == 8) wcur = a.w + SDF_OFF_FEAT;
      else wcur = a.w + sdf_off_R(16 - s);
      if (s < 7) wnxt = a.w + sdf_off_L(s + 1);
      else if (s == 7) {
        if (MODE == 0) { wnxt = a.w + SDF_OFF_L0; npc = 8; }
        else if (MODE == 1) wnxt = a.w + sdf_off_R(7);
        else wnxt = a.w + SDF_OFF_FEAT;
      } else if (s == 8) wnxt = a.w + sdf_off_R(7);
      else if (s < 15) wnxt = a.w + sdf_off_R(16 - s - 1);
      else wnxt = a.w + SDF_OFF_R0;

      if (MODE >= 1 && s == 9) {
        // start of the reverse chain: t_7 = sigma'_7 * (w_s / 3), written by L7's epilogue
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's own t_7 stores have landed
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
          f32x4 v0, v1;
          if constexpr (TRAIN) {
            v0 = ld_stream(reinterpret_cast<const f32x4*>(arr_ptr<ARR_T, PREC == 1>(a.save_t, 7, a.npts, row, 2 * ch, q)));
            v1 = ld_stream(reinterpret_cast<const f32x4*>(arr_ptr<ARR_T, PREC == 1>(a.save_t, 7, a.npts, row, 2 * ch + 1, q)));
          } else {
            v0 = ld_stream(reinterpret_cast<const f32x4*>(t7_ptr<PREC>(scr, 2 * ch, lane)));
            v1 = ld_stream(reinterpret_cast<const f32x4*>(t7_ptr<PREC>(scr, 2 * ch + 1, lane)));
          }
          h.set_chunk(ch, v0, v1);
        }
      }

      Act<PREC, 16> ho;
      auto pre = [&](int ch) {
        PreV p;
        if (s <= 8) {
          p.a0 = *reinterpret_cast<const f32x4*>(a.b + s * 256 + (2 * ch) * 16 + 4 * q);
          p.a1 = *reinterpret_cast<const f32x4*>(a.b + s * 256 + (2 * ch + 1) * 16 + 4 * q);
          if (s == 7) {
            p.b0 = *reinterpret_cast<const f32x4*>(a.head + (2 * ch) * 16 + 4 * q);
            p.b1 = *reinterpret_cast<const f32x4*>(a.head + (2 * ch + 1) * 16 + 4 * q);
          }
        } else if (MODE >= 1) {
          // sigma' of the layer this stage's output feeds
          if constexpr (TRAIN) {
            p.a0 = ld_stream(reinterpret_cast<const f32x4*>(arr_ptr<ARR_S1, PREC == 1>(a.save_s1, 16 - s - 1, a.npts, row, 2 * ch, q)));
            p.a1 = ld_stream(reinterpret_cast<const f32x4*>(arr_ptr<ARR_S1, PREC == 1>(a.save_s1, 16 - s - 1, a.npts, row, 2 * ch + 1, q)));
          } else {
            dsig_issue<PREC>(scr, 16 - s - 1, ch, lane, p);
          }
        }
        return p;
      };
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1, const PreV& p) {
        if (s <= 7) {
          f32x4 h0, h1, d0, d1;
          softplus100_4<WANT_D>(acc0 + p.a0, h0, d0);
          softplus100_4<WANT_D>(acc1 + p.a1, h1, d1);
          if (ch >= 6 && s == 3) {
            // skip connection: features 217..255 of L4's input are the embedding (fields/sdf_field.py:113-114)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (ch == 7) {  // block 14: always >= 217
                h0[r] = nerf_enc_entry_q<3, 6>(x3, (2 * ch) * 16 + r - 217, q);
                d0[r] = 0.0f;
              }
              if ((2 * ch + 1) * 16 + 4 * q + r - 217 >= 0) {  // block 13 (partly) or 15
                h1[r] = nerf_enc_entry_q<3, 6>(x3, (2 * ch + 1) * 16 + r - 217, q);
                d1[r] = 0.0f;
              }
            }
          }
          if (s == 7) {
            // sdf head folded into L7's epilogue: partial w_s . h8, and t_7 = sigma'_7 * w_s / 3 for the reverse chain
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              head_part += p.b0[r] * h0[r];
              head_part += p.b1[r] * h1[r];
            }
            if constexpr (TRAIN) {
              ds_store(7, ch, d0, d1);
              save_rows(ArrTag<ARR_T>(), a.save_t, 7, ch, d0 * (p.b0 * (1.0f / 3.0f)), d1 * (p.b1 * (1.0f / 3.0f)));
            } else if (MODE >= 1) {
              st_stream(reinterpret_cast<f32x4*>(t7_ptr<PREC>(scr, 2 * ch, lane)), d0 * (p.b0 * (1.0f / 3.0f)));
              st_stream(reinterpret_cast<f32x4*>(t7_ptr<PREC>(scr, 2 * ch + 1, lane)), d1 * (p.b1 * (1.0f / 3.0f)));
            }
          } else if (MODE >= 1) {
            ds_store(s, ch, d0, d1);
          }
          if constexpr (TRAIN) save_rows(ArrTag<ARR_H>(), a.save_h, s, ch, h0, h1);
          ho.set_chunk(ch, h0, h1);
        } else if (s == 8) {
          if constexpr (TRAIN) {
            save_rows(ArrTag<ARR_ROWS>(), a.feat, 0, ch, acc0 + p.a0, acc1 + p.a1);
          } else if (MODE == 2) {
            if (tile * TILE_PTS < a.npts) {
              float* ft = a.feat + (size_t)tile * (16 * 256);
              st_stream(reinterpret_cast<f32x4*>(ft + ((2 * ch) * 64 + lane) * 4), acc0 + p.a0);
              st_stream(reinterpret_cast<f32x4*>(ft + ((2 * ch + 1) * 64 + lane) * 4), acc1 + p.a1);
            }
          }
        } else {
          if (MODE >= 1) {
            const int l = 16 - s;  // this stage multiplied by W_l^T; its output feeds layer l-1's sigma'
            if (l == 4 && ch >= 6) {
              // gradient w.r.t. the embedding through the skip connection (inputs 217..255 of L4)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                if (2 * ch >= 13) skip[(2 * ch - 13) * 4 + r] = acc0[r];
                skip[(2 * ch + 1 - 13) * 4 + r] = acc1[r];
              }
            }
            f32x4 d0, d1;
            if constexpr (TRAIN) {
              d0 = p.a0;
              d1 = p.a1;
              save_rows(ArrTag<ARR_T>(), a.save_t, l - 1, ch, acc0 * d0, acc1 * d1);
            } else {
              dsig_decode<PREC>(p, d0, d1);
            }
            ho.set_chunk(ch, acc0 * d0, acc1 * d1);
          }
        }
      };
      run_stage<PREC, 16, 8, false, true>(wcur, wnxt, npc, smem, par, h, nullptr, pre, epi, wave, lane);

      if (s == 7) {
        // sdf head: (w_s . h8 + b_s) / scale   (fields/sdf_field.py:121)
        float part = head_part;
        part += __shfl_xor(part, 16, 64);
        part += __shfl_xor(part, 32, 64);
        if (valid && q == 0) a.sdf[ray * a.sdf_stride + jj] = (part + a.head[256]) / 3.0f;
      }
      if (s != 8) h = ho;
    }

    if (MODE >= 1) {
      // ---- R0: gradient w.r.t. the 39 embedding entries, then chain through the encoding ----
      float ge[16];
      auto pre = [&](int) { return 0; };
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1, int) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { ge[ch * 8 + r] = acc0[r]; ge[ch * 8 + 4 + r] = acc1[r]; }
      };
      run_stage<PREC, 16, 2, false>(a.w + SDF_OFF_R0, a.w + SDF_OFF_L0, 8, smem, par, h, nullptr, pre, epi, wave, lane);

      if constexpr (TRAIN) {
        if (tile_ok) {
          float* gr = a.save_ge + (size_t)row * 128 + 4 * q;
#pragma unroll
          for (int b = 0; b < 4; ++b) *reinterpret_cast<f32x4*>(gr + b * 16) = f32x4{ge[b * 4], ge[b * 4 + 1], ge[b * 4 + 2], ge[b * 4 + 3]};
#pragma unroll
          for (int b = 0; b < 3; ++b) *reinterpret_cast<f32x4*>(gr + 64 + b * 16) = f32x4{skip[b * 4], skip[b * 4 + 1], skip[b * 4 + 2], skip[b * 4 + 3]};
        }
      }
      float dx[3] = {0.f, 0.f, 0.f};
      {
        float dc[39];
        nerf_enc_dall<3, 6>(x3, dc);
        // entry e = 16b + 4q + r of the R0 output, entry e = 16(b+13) + 4q + r - 217 of the skip part;
        // every candidate q is enumerated statically so nothing is indexed at run time
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
              const int e = b * 16 + 4 * qq + r;
              if (e < 39) dx[nerf_enc_dim<3, 6>(e)] += (q == qq) ? ge[b * 4 + r] * dc[e] : 0.0f;
              const int es = (b + 13) * 16 + 4 * qq + r - 217;
              if (es >= 0 && es < 39) dx[nerf_enc_dim<3, 6>(es)] += (q == qq) ? skip[b * 4 + r] * dc[es] : 0.0f;
            }
      }
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        dx[c] += __shfl_xor(dx[c], 16, 64);
        dx[c] += __shfl_xor(dx[c], 32, 64);
      }
      if (valid && q == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) a.grad[P * 3 + c] = dx[c] * 3.0f;  // d(3x)/dx
      }
    }
  }
}

}  // namespace nrh
