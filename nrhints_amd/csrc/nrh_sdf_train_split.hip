// The SDF network's training forward for SMALL batches (precision f16x3): sdf_kernel<3, 1> (nrh_sdf.hip) with one 16-point tile per
// workgroup and every stage's 256 output channels split over the workgroup's four waves (nrh_mlp_split.h).
//
// A 64-ray training batch (the reference's per-rank share under 8-way DDP, trainer/trainer.py:116-123) is 512 tiles: one round of
// single-tile latencies whatever the kernel, 0.20 ms for the 17 stages of the training forward when one wave does a tile alone.
// Here the tile's MFMA work runs on the CU's four matrix cores at once and the stage costs a quarter of it plus an LDS exchange.
//
// Same contract as nrh_sdf_train_forward (include/nrhints_hip.h): sdf, d sdf / dx, the feature rows and the saved arrays
// save_h / save_s1 / save_t / save_ge, bit for bit (tests/test_gpu_split.py) - the stage arithmetic is run_stage's, the epilogues
// are sdf_kernel<3, 1>'s on the same lane / register positions; what differs is who holds what:
//   head      h_7 goes through LDS as float32 and wave 0 sums w_s . h_7 in the 16-point kernel's order
//   sigma', t_7   written to save_s1 / save_t by the forward stages and read back by the lanes that stored them (after ONE
//             s_waitcnt vmcnt(0) between the two halves); t_7 becomes R7's B rows in the feature stage's epilogue
//   R0        64 output rows = two chunks: waves 0 and 1 take one each; the embedding adjoint's 28 values per lane (16 from R0,
//             12 that wave 3 parks there from R4's skip rows) meet on wave 0 through LDS, which finishes d sdf / dx in the original order
#include "nrh_mlp_split.h"

namespace nrh {

constexpr int SPLT_TAB_FLOATS = 9 * 256 + 272;
constexpr int SPLT_ROW7 = 1056;                                   // one point's 256 float32 h_7 values (+32 bytes)
constexpr int SPLT_OFF_TAB = 2 * SPL_BUF;
constexpr int SPLT_OFF_H7 = SPLT_OFF_TAB + SPLT_TAB_FLOATS * 4;
constexpr int SPLT_OFF_G = SPLT_OFF_H7 + 16 * SPLT_ROW7;
constexpr int SPLT_G_FLOATS = 28;                                 // per (point, q): ge[16] | skip[12]
constexpr int SPLT_LDS_BYTES = SPLT_OFF_G + 16 * 4 * SPLT_G_FLOATS * 4;

struct SplPreF {
  f32x4 a0, a1;   // forward: bias of the two blocks;  reverse: sigma' of the layer the stage's output feeds
  f32x4 b0, b1;   // layer 7: sdf-head weights
};

__global__ __launch_bounds__(256, 2) void sdf_train_split_kernel(const SdfArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  const char* const W = reinterpret_cast<const char*>(a.w);
  char* const buf0 = smem;
  char* const buf1 = smem + SPL_BUF;
  float* const tab = reinterpret_cast<float*>(smem + SPLT_OFF_TAB);
  char* const h7 = smem + SPLT_OFF_H7;
  float* const G = reinterpret_cast<float*>(smem + SPLT_OFF_G);

  // ---- the tile's points; tables -> LDS; then the weight ring starts (see nrh_sdf_split.hip for the order) ----
  const long long row = (long long)blockIdx.x * TILE_PTS + j;     // npts % 16 == 0 (host check): every tile is whole
  const long long ray = row / a.n_per_ray;
  const int jj = (int)(row - ray * a.n_per_ray);
  const float tt = a.t[ray * a.t_stride + jj];
  float x3[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) x3[c] = (a.ro[ray * 3 + c] + a.rd[ray * 3 + c] * tt) * 3.0f;  // inputs * scale
  {
    const f32x4* const bsrc = reinterpret_cast<const f32x4*>(a.b);
    f32x4* const bdst = reinterpret_cast<f32x4*>(tab);
    const f32x4 v0 = bsrc[threadIdx.x], v1 = bsrc[256 + threadIdx.x];
    const f32x4 v2 = (threadIdx.x < 64) ? bsrc[512 + threadIdx.x] : f32x4{0.f, 0.f, 0.f, 0.f};
    const float hv = a.head[threadIdx.x], hl = a.head[256];
    bdst[threadIdx.x] = v0;
    bdst[256 + threadIdx.x] = v1;
    if (threadIdx.x < 64) bdst[512 + threadIdx.x] = v2;
    tab[9 * 256 + threadIdx.x] = hv;
    if (threadIdx.x == 0) tab[9 * 256 + 256] = hl;
  }
  __builtin_amdgcn_sched_barrier(0);

  SplRing<4> ring;      // 4 sets of prefetch (64 VGPRs): with two workgroups per CU that is 128 KiB of weights in flight per CU
  const char* const l0c = W + (size_t)SDF_OFF_L0 * 4 + (size_t)(2 * wave) * 8192;
  auto chunks = [&](int float_off) { return W + (size_t)float_off * 4 + (size_t)(2 * wave) * 32768; };   // this wave's two chunks of a 256 x 256 stage
  spl_prologue<2, 2, 8, 4>(ring, l0c, chunks(sdf_off_L(1)), lane);
  __builtin_amdgcn_sched_barrier(0);

  Act<1, 4> emb;
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
  __syncthreads();     // the tables are in LDS

  auto save_rows = [&](auto AT, float* base, int l, int ch, const f32x4 v0, const f32x4 v1) {
      constexpr int ARR = decltype(AT)::value;
    st_stream(reinterpret_cast<f32x4*>(arr_ptr<ARR>(base, l, a.npts, row, 2 * ch, q)), v0);
    st_stream(reinterpret_cast<f32x4*>(arr_ptr<ARR>(base, l, a.npts, row, 2 * ch + 1, q)), v1);
  };
  float* const Gl = G + (j * 4 + q) * SPLT_G_FLOATS;      // this lane's slots of the embedding-adjoint exchange

  // ---- forward stages L0..L7 ----
  auto fwd = [&](auto SC, const char* cur, const char* nxt, char* out, const auto& bsrc) {
    constexpr int S = decltype(SC)::value;
    auto pre = [&](auto CIC) {
      constexpr int CI = decltype(CIC)::value;
      const int ch = 2 * wave + CI;
      SplPreF p;
      p.a0 = *reinterpret_cast<const f32x4*>(tab + S * 256 + (2 * ch) * 16 + 4 * q);
      p.a1 = *reinterpret_cast<const f32x4*>(tab + S * 256 + (2 * ch + 1) * 16 + 4 * q);
      if constexpr (S == 7) {
        p.b0 = *reinterpret_cast<const f32x4*>(tab + 9 * 256 + (2 * ch) * 16 + 4 * q);
        p.b1 = *reinterpret_cast<const f32x4*>(tab + 9 * 256 + (2 * ch + 1) * 16 + 4 * q);
      }
      return p;
    };
    auto epi = [&](auto CIC, f32x4 acc0, f32x4 acc1, const SplPreF& p) {
      constexpr int CI = decltype(CIC)::value;
      const int ch = 2 * wave + CI;
      f32x4 h0, h1, d0, d1;
      softplus100_4<true>(acc0 + p.a0, h0, d0);
      softplus100_4<true>(acc1 + p.a1, h1, d1);
      if constexpr (S == 3) {
        if (wave == 3) {
          // skip connection: features 217..255 of L4's input are the embedding (fields/sdf_field.py:113-114)
          constexpr int chs = 6 + CI;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (chs == 7) {
              h0[r] = nerf_enc_entry_q<3, 6>(x3, (2 * chs) * 16 + r - 217, q);
              d0[r] = 0.0f;
            }
            if ((2 * chs + 1) * 16 + 4 * q + r - 217 >= 0) {
              h1[r] = nerf_enc_entry_q<3, 6>(x3, (2 * chs + 1) * 16 + r - 217, q);
              d1[r] = 0.0f;
            }
          }
        }
      }
      save_rows(ArrTag<ARR_S1>(), a.save_s1, S, ch, d0, d1);
      if constexpr (S == 7) {
        save_rows(ArrTag<ARR_T>(), a.save_t, 7, ch, d0 * (p.b0 * (1.0f / 3.0f)), d1 * (p.b1 * (1.0f / 3.0f)));
        char* const p7 = h7 + j * SPLT_ROW7 + ((2 * ch) * 16 + 4 * q) * 4;
        *reinterpret_cast<f32x4*>(p7) = h0;
        *reinterpret_cast<f32x4*>(p7 + 64) = h1;
      }
      save_rows(ArrTag<ARR_H>(), a.save_h, S, ch, h0, h1);
      spl_store_act(out, j, q, ch, h0, h1);
    };
    if constexpr (S == 0) spl_stage<2, 2, 0, 8, 2, 4>(ring, cur, nxt, lane, bsrc, pre, epi);
    else spl_stage<8, 2, 0, 8, 2, 4>(ring, cur, nxt, lane, bsrc, pre, epi);
    __syncthreads();
  };
  auto ldsb = [&](const char* in) { return SplLdsB{in + j * SPL_ROW + 16 * q}; };
  fwd(IC<0>(), l0c, chunks(sdf_off_L(1)), buf0, SplRegB{&emb});
  fwd(IC<1>(), chunks(sdf_off_L(1)), chunks(sdf_off_L(2)), buf1, ldsb(buf0));
  fwd(IC<2>(), chunks(sdf_off_L(2)), chunks(sdf_off_L(3)), buf0, ldsb(buf1));
  fwd(IC<3>(), chunks(sdf_off_L(3)), chunks(sdf_off_L(4)), buf1, ldsb(buf0));
  fwd(IC<4>(), chunks(sdf_off_L(4)), chunks(sdf_off_L(5)), buf0, ldsb(buf1));
  fwd(IC<5>(), chunks(sdf_off_L(5)), chunks(sdf_off_L(6)), buf1, ldsb(buf0));
  fwd(IC<6>(), chunks(sdf_off_L(6)), chunks(sdf_off_L(7)), buf0, ldsb(buf1));
  fwd(IC<7>(), chunks(sdf_off_L(7)), chunks(SDF_OFF_FEAT), buf1, ldsb(buf0));

  // this wave's sigma' and t_7 rows are read back below by the lanes that stored them: their stores have landed past this point
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  // ---- feature head (no activation); its epilogue also turns t_7 into R7's B rows (buf0 is free: L7 has read it) ----
  {
    auto pre = [&](auto CIC) {
      constexpr int CI = decltype(CIC)::value;
      const int ch = 2 * wave + CI;
      SplPreF p;
      p.a0 = *reinterpret_cast<const f32x4*>(tab + 8 * 256 + (2 * ch) * 16 + 4 * q);
      p.a1 = *reinterpret_cast<const f32x4*>(tab + 8 * 256 + (2 * ch + 1) * 16 + 4 * q);
      p.b0 = ld_stream(reinterpret_cast<const f32x4*>(arr_ptr<ARR_T>(a.save_t, 7, a.npts, row, 2 * ch, q)));
      p.b1 = ld_stream(reinterpret_cast<const f32x4*>(arr_ptr<ARR_T>(a.save_t, 7, a.npts, row, 2 * ch + 1, q)));
      return p;
    };
    auto epi = [&](auto CIC, f32x4 acc0, f32x4 acc1, const SplPreF& p) {
      constexpr int CI = decltype(CIC)::value;
      const int ch = 2 * wave + CI;
      save_rows(ArrTag<ARR_ROWS>(), a.feat, 0, ch, acc0 + p.a0, acc1 + p.a1);
      spl_store_act(buf0, j, q, ch, p.b0, p.b1);
    };
    spl_stage<8, 2, 0, 8, 2, 4>(ring, chunks(SDF_OFF_FEAT), chunks(sdf_off_R(7)), lane, ldsb(buf1), pre, epi);
    __syncthreads();
  }

  // ---- reverse chain R7..R1: g <- W_l^T (sigma'_l * g) ----
  auto rev = [&](auto LC, const char* cur, const char* nxt, const char* in, char* out, auto NCN) {
    constexpr int L = decltype(LC)::value;          // this stage multiplies by W_L^T; its output feeds layer L-1's sigma'
    auto pre = [&](auto CIC) {
      constexpr int CI = decltype(CIC)::value;
      const int ch = 2 * wave + CI;
      SplPreF p;
      p.a0 = ld_stream(reinterpret_cast<const f32x4*>(arr_ptr<ARR_S1>(a.save_s1, L - 1, a.npts, row, 2 * ch, q)));
      p.a1 = ld_stream(reinterpret_cast<const f32x4*>(arr_ptr<ARR_S1>(a.save_s1, L - 1, a.npts, row, 2 * ch + 1, q)));
      return p;
    };
    auto epi = [&](auto CIC, f32x4 acc0, f32x4 acc1, const SplPreF& p) {
      constexpr int CI = decltype(CIC)::value;
      const int ch = 2 * wave + CI;
      if constexpr (L == 4) {
        if (wave == 3) {
          // gradient w.r.t. the embedding through the skip connection (inputs 217..255 of L4)
          constexpr int chs = 6 + CI;     // skip[(blk - 13) * 4 + r], blk = 13 (chunk 6, second block), 14, 15 (chunk 7)
          if (2 * chs >= 13) *reinterpret_cast<f32x4*>(Gl + 16 + (2 * chs - 13) * 4) = acc0;
          *reinterpret_cast<f32x4*>(Gl + 16 + (2 * chs + 1 - 13) * 4) = acc1;
        }
      }
      const f32x4 o0 = acc0 * p.a0, o1 = acc1 * p.a1;
      save_rows(ArrTag<ARR_T>(), a.save_t, L - 1, ch, o0, o1);
      spl_store_act(out, j, q, ch, o0, o1);
    };
    spl_stage<8, 2, 0, 8, decltype(NCN)::value, 4>(ring, cur, nxt, lane, ldsb(in), pre, epi);
    __syncthreads();
  };
  rev(IC<7>(), chunks(sdf_off_R(7)), chunks(sdf_off_R(6)), buf0, buf1, IC<2>());
  rev(IC<6>(), chunks(sdf_off_R(6)), chunks(sdf_off_R(5)), buf1, buf0, IC<2>());
  rev(IC<5>(), chunks(sdf_off_R(5)), chunks(sdf_off_R(4)), buf0, buf1, IC<2>());
  rev(IC<4>(), chunks(sdf_off_R(4)), chunks(sdf_off_R(3)), buf1, buf0, IC<2>());
  rev(IC<3>(), chunks(sdf_off_R(3)), chunks(sdf_off_R(2)), buf0, buf1, IC<2>());
  rev(IC<2>(), chunks(sdf_off_R(2)), chunks(sdf_off_R(1)), buf1, buf0, IC<2>());
  const char* const r0c = (wave < 2) ? W + (size_t)SDF_OFF_R0 * 4 + (size_t)wave * 32768 : nullptr;   // R0: 64 rows = two chunks, one for each of waves 0 and 1
  rev(IC<1>(), chunks(sdf_off_R(1)), r0c, buf0, buf1, IC<1>());

  // ---- R0: gradient w.r.t. the 39 embedding entries (waves 0, 1); wave 3 contributes the skip rows it kept from R4 ----
  if (wave < 2) {
    auto pre = [&](auto) { return 0; };
    auto epi = [&](auto, f32x4 acc0, f32x4 acc1, int) {
      *reinterpret_cast<f32x4*>(Gl + wave * 8) = acc0;          // ge[ch * 8 + r], ch = wave
      *reinterpret_cast<f32x4*>(Gl + wave * 8 + 4) = acc1;
    };
    spl_stage<8, 1, 0, 0, 0, 4>(ring, r0c, nullptr, lane, ldsb(buf1), pre, epi);
  }
  __syncthreads();

  // ---- wave 0 finishes the tile in sdf_kernel<3>'s order: sdf head, save_ge, chain through the encoding ----
  if (wave == 0) {
    float head_part = 0.0f;
    const char* const r7 = h7 + j * SPLT_ROW7;
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(tab + 9 * 256 + (2 * ch) * 16 + 4 * q);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(tab + 9 * 256 + (2 * ch + 1) * 16 + 4 * q);
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(r7 + ((2 * ch) * 16 + 4 * q) * 4);
      const f32x4 h1 = *reinterpret_cast<const f32x4*>(r7 + ((2 * ch + 1) * 16 + 4 * q) * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        head_part += w0[r] * h0[r];
        head_part += w1[r] * h1[r];
      }
    }
    float part = head_part;
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);
    if (q == 0) a.sdf[ray * a.sdf_stride + jj] = (part + tab[9 * 256 + 256]) / 3.0f;

    float ge[16], sk[12];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(Gl + b * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) ge[b * 4 + r] = v[r];
    }
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(Gl + 16 + b * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) sk[b * 4 + r] = v[r];
    }
    float* gr = a.save_ge + (size_t)row * 128 + 4 * q;
#pragma unroll
    for (int b = 0; b < 4; ++b) *reinterpret_cast<f32x4*>(gr + b * 16) = f32x4{ge[b * 4], ge[b * 4 + 1], ge[b * 4 + 2], ge[b * 4 + 3]};
#pragma unroll
    for (int b = 0; b < 3; ++b) *reinterpret_cast<f32x4*>(gr + 64 + b * 16) = f32x4{sk[b * 4], sk[b * 4 + 1], sk[b * 4 + 2], sk[b * 4 + 3]};
    float dx[3] = {0.f, 0.f, 0.f};
    {
      float dc[39];
      nerf_enc_dall<3, 6>(x3, dc);
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const int e = b * 16 + 4 * qq + r;
            if (e < 39) dx[nerf_enc_dim<3, 6>(e)] += (q == qq) ? ge[b * 4 + r] * dc[e] : 0.0f;
            const int es = (b + 13) * 16 + 4 * qq + r - 217;
            if (es >= 0 && es < 39) dx[nerf_enc_dim<3, 6>(es)] += (q == qq) ? sk[b * 4 + r] * dc[es] : 0.0f;
          }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      dx[c] += __shfl_xor(dx[c], 16, 64);
      dx[c] += __shfl_xor(dx[c], 32, 64);
    }
    if (q == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) a.grad[row * 3 + c] = dx[c] * 3.0f;  // d(3x)/dx
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// the two backward sweeps (sdf_tangent_kernel<1> / sdf_adjoint_kernel<1> of nrh_sdf_train.hip) in the same form
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int SPLB_OFF_G = 2 * SPL_BUF;
constexpr int SPLB_LDS_BYTES = SPLB_OFF_G + 16 * 4 * SPLT_G_FLOATS * 4;

// tangent sweep: FORWARD through L0..L7 on the tangent adjoints; coup_l and abar_{l+1} out
__global__ __launch_bounds__(256, 2) void sdf_tangent_split_kernel(const SdfTrainArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  const char* const W = reinterpret_cast<const char*>(a.w);
  char* const buf0 = smem;
  char* const buf1 = smem + SPL_BUF;

  const long long row = (long long)blockIdx.x * TILE_PTS + j;
  const long long ray = row / a.n_per_ray;
  const int jj = (int)(row - ray * a.n_per_ray);
  const float tpar = a.t[ray * a.t_stride + jj];
  float x3[3], g3[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    x3[c] = (a.ro[ray * 3 + c] + a.rd[ray * 3 + c] * tpar) * 3.0f;
    g3[c] = a.gbar[row * 3 + c] * 3.0f * a.adj_scale;      // (SdfTrainArgs.adj_scale: scaled chain, outputs leave as 1 / S)
  }
  const float IS = 1.0f / a.adj_scale;
  __builtin_amdgcn_sched_barrier(0);

  SplRing<4> ring;
  const char* const l0c = W + (size_t)SDF_OFF_L0 * 4 + (size_t)(2 * wave) * 8192;
  auto chunks = [&](int float_off) { return W + (size_t)float_off * 4 + (size_t)(2 * wave) * 32768; };
  spl_prologue<2, 2, 8, 4>(ring, l0c, chunks(sdf_off_L(1)), lane);
  __builtin_amdgcn_sched_barrier(0);

  // abar_0 (embedding layout of the forward's L0 input); wave 0 writes gebar
  Act<1, 4> emb;
#pragma unroll
  for (int c2 = 0; c2 < 2; ++c2) {
    float o[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int b = 2 * c2 + (r >> 2);
      o[r] = (b < 3) ? enc_dentry_dot_q(x3, g3, b * 16 + (r & 3), q) : 0.0f;
    }
    emb.set_chunk(c2, o);
    if (wave == 0) {
      float* gr = a.gebar + (size_t)row * 64 + 4 * q;
      *reinterpret_cast<f32x4*>(gr + (2 * c2) * 16) = f32x4{o[0], o[1], o[2], o[3]} * IS;
      *reinterpret_cast<f32x4*>(gr + (2 * c2 + 1) * 16) = f32x4{o[4], o[5], o[6], o[7]} * IS;
    }
  }

  auto stage = [&](auto SC, const char* cur, const char* nxt, char* out, const auto& bsrc, auto NCN) {
    constexpr int S = decltype(SC)::value;
    auto pre = [&](auto CIC) {
      constexpr int CI = decltype(CIC)::value;
      const int ch = 2 * wave + CI;
      TrainPre p;
      p.s0 = ld_stream(reinterpret_cast<const f32x4*>(arr_ptr<ARR_S1>(a.s1, S, a.npts, row, 2 * ch, q)));
      p.s1 = ld_stream(reinterpret_cast<const f32x4*>(arr_ptr<ARR_S1>(a.s1, S, a.npts, row, 2 * ch + 1, q)));
      p.t0 = ld_stream(reinterpret_cast<const f32x4*>(arr_ptr<ARR_T>(a.tt, S, a.npts, row, 2 * ch, q)));
      p.t1 = ld_stream(reinterpret_cast<const f32x4*>(arr_ptr<ARR_T>(a.tt, S, a.npts, row, 2 * ch + 1, q)));
      return p;
    };
    auto epi = [&](auto CIC, f32x4 acc0, f32x4 acc1, const TrainPre& p) {
      constexpr int CI = decltype(CIC)::value;
      const int ch = 2 * wave + CI;
      const f32x4 c0 = (1.0f - p.s0) * p.t0 * acc0 * 100.0f;
      const f32x4 c1 = (1.0f - p.s1) * p.t1 * acc1 * 100.0f;
      f32x4 n0 = p.s0 * acc0, n1 = p.s1 * acc1;
      if constexpr (S == 3) {
        if (wave == 3) {
          // abar_4 = [abar_4h (217), abar_0 (39)]: the skip connection (fields/sdf_field.py:113-114)
          constexpr int chs = 6 + CI;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (chs == 7) n0[r] = enc_dentry_dot_q(x3, g3, (2 * chs) * 16 + r - 217, q);
            if ((2 * chs + 1) * 16 + 4 * q + r - 217 >= 0) n1[r] = enc_dentry_dot_q(x3, g3, (2 * chs + 1) * 16 + r - 217, q);
          }
        }
      }
      st_stream(reinterpret_cast<f32x4*>(arr_ptr<ARR_COUP>(a.coup, S, a.npts, row, 2 * ch, q)), c0 * IS);
      st_stream(reinterpret_cast<f32x4*>(arr_ptr<ARR_COUP>(a.coup, S, a.npts, row, 2 * ch + 1, q)), c1 * IS);
      st_stream(reinterpret_cast<f32x4*>(arr_ptr<ARR_ABAR>(a.abar, S, a.npts, row, 2 * ch, q)), n0 * IS);
      st_stream(reinterpret_cast<f32x4*>(arr_ptr<ARR_ABAR>(a.abar, S, a.npts, row, 2 * ch + 1, q)), n1 * IS);
      if constexpr (S < 7) spl_store_act(out, j, q, ch, n0, n1);
    };
    if constexpr (S == 0) spl_stage<2, 2, 0, 8, 2, 4>(ring, cur, nxt, lane, bsrc, pre, epi);
    else spl_stage<8, 2, 0, 8, decltype(NCN)::value, 4>(ring, cur, nxt, lane, bsrc, pre, epi);
    if constexpr (S < 7) __syncthreads();
  };
  auto ldsb = [&](const char* in) { return SplLdsB{in + j * SPL_ROW + 16 * q}; };
  stage(IC<0>(), l0c, chunks(sdf_off_L(1)), buf0, SplRegB{&emb}, IC<2>());
  stage(IC<1>(), chunks(sdf_off_L(1)), chunks(sdf_off_L(2)), buf1, ldsb(buf0), IC<2>());
  stage(IC<2>(), chunks(sdf_off_L(2)), chunks(sdf_off_L(3)), buf0, ldsb(buf1), IC<2>());
  stage(IC<3>(), chunks(sdf_off_L(3)), chunks(sdf_off_L(4)), buf1, ldsb(buf0), IC<2>());
  stage(IC<4>(), chunks(sdf_off_L(4)), chunks(sdf_off_L(5)), buf0, ldsb(buf1), IC<2>());
  stage(IC<5>(), chunks(sdf_off_L(5)), chunks(sdf_off_L(6)), buf1, ldsb(buf0), IC<2>());
  stage(IC<6>(), chunks(sdf_off_L(6)), chunks(sdf_off_L(7)), buf0, ldsb(buf1), IC<2>());
  stage(IC<7>(), chunks(sdf_off_L(7)), nullptr, buf1, ldsb(buf0), IC<0>());
}

// value sweep: REVERSE through FEAT^T, R7..R1, R0 on the value adjoints; zbar_l and pbar out
__global__ __launch_bounds__(256, 2) void sdf_adjoint_split_kernel(const SdfTrainArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  const char* const W = reinterpret_cast<const char*>(a.w);
  char* const buf0 = smem;
  char* const buf1 = smem + SPL_BUF;
  float* const G = reinterpret_cast<float*>(smem + SPLB_OFF_G);
  float* const Gl = G + (j * 4 + q) * SPLT_G_FLOATS;

  const long long row = (long long)blockIdx.x * TILE_PTS + j;
  const long long ray = row / a.n_per_ray;
  const int jj = (int)(row - ray * a.n_per_ray);
  const float tpar = a.t[ray * a.t_stride + jj];
  float x3[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) x3[c] = (a.ro[ray * 3 + c] + a.rd[ray * 3 + c] * tpar) * 3.0f;
  const float AS = a.adj_scale, IS = 1.0f / AS;      // (SdfTrainArgs.adj_scale: scaled chain, outputs leave as 1 / S)
  const float sb3 = a.sbar[row] / 3.0f * AS;
  // fbar of this wave's 64 channels -> B rows of the first stage
  f32x4 fb[2][2];
#pragma unroll
  for (int ci = 0; ci < 2; ++ci) {
    const int ch = 2 * wave + ci;
    fb[ci][0] = ld_stream(reinterpret_cast<const f32x4*>(arr_ptr<ARR_ROWS>(a.fbar, 0, a.npts, row, 2 * ch, q))) * AS;
    fb[ci][1] = ld_stream(reinterpret_cast<const f32x4*>(arr_ptr<ARR_ROWS>(a.fbar, 0, a.npts, row, 2 * ch + 1, q))) * AS;
  }
  __builtin_amdgcn_sched_barrier(0);

  SplRing<4> ring;
  auto chunks = [&](int float_off) { return W + (size_t)float_off * 4 + (size_t)(2 * wave) * 32768; };
  const char* const ftc = reinterpret_cast<const char*>(a.wt_feat) + (size_t)(2 * wave) * 32768;
  spl_prologue<8, 2, 8, 4>(ring, ftc, chunks(sdf_off_R(7)), lane);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int ci = 0; ci < 2; ++ci) spl_store_act(buf0, j, q, 2 * wave + ci, fb[ci][0], fb[ci][1]);
  __syncthreads();

  // S = 8: FEAT^T (-> zbar_7);  S = 7..1: R_S = W_S^T (-> zbar_{S-1})
  auto stage = [&](auto SC, const char* cur, const char* nxt, const char* in, char* out, auto NCN) {
    constexpr int S = decltype(SC)::value;
    constexpr int lz = S - 1;
    auto pre = [&](auto CIC) {
      constexpr int CI = decltype(CIC)::value;
      const int ch = 2 * wave + CI;
      TrainPre p;
      p.s0 = ld_stream(reinterpret_cast<const f32x4*>(arr_ptr<ARR_S1>(a.s1, lz, a.npts, row, 2 * ch, q)));
      p.s1 = ld_stream(reinterpret_cast<const f32x4*>(arr_ptr<ARR_S1>(a.s1, lz, a.npts, row, 2 * ch + 1, q)));
      p.t0 = ld_stream(reinterpret_cast<const f32x4*>(arr_ptr<ARR_COUP>(a.coup, lz, a.npts, row, 2 * ch, q)));
      p.t1 = ld_stream(reinterpret_cast<const f32x4*>(arr_ptr<ARR_COUP>(a.coup, lz, a.npts, row, 2 * ch + 1, q)));
      if constexpr (S == 8) {
        p.w0 = *reinterpret_cast<const f32x4*>(a.head + (2 * ch) * 16 + 4 * q);
        p.w1 = *reinterpret_cast<const f32x4*>(a.head + (2 * ch + 1) * 16 + 4 * q);
      }
      return p;
    };
    auto epi = [&](auto CIC, f32x4 acc0, f32x4 acc1, const TrainPre& p) {
      constexpr int CI = decltype(CIC)::value;
      const int ch = 2 * wave + CI;
      if constexpr (S == 8) {
        acc0 += p.w0 * sb3;  // hbar_7 = Wf^T fbar + sbar w_s / 3
        acc1 += p.w1 * sb3;
      }
      if constexpr (S == 4) {
        if (wave == 3) {
          // adjoint of the embedding through the skip connection (inputs 217..255 of L4): parked for wave 0
          constexpr int chs = 6 + CI;
          if (2 * chs >= 13) *reinterpret_cast<f32x4*>(Gl + 16 + (2 * chs - 13) * 4) = acc0;
          *reinterpret_cast<f32x4*>(Gl + 16 + (2 * chs + 1 - 13) * 4) = acc1;
        }
      }
      const f32x4 z0 = p.s0 * acc0 + p.t0 * AS, z1 = p.s1 * acc1 + p.t1 * AS;
      st_stream(reinterpret_cast<f32x4*>(arr_ptr<ARR_ZBAR>(a.zbar, lz, a.npts, row, 2 * ch, q)), z0 * IS);
      st_stream(reinterpret_cast<f32x4*>(arr_ptr<ARR_ZBAR>(a.zbar, lz, a.npts, row, 2 * ch + 1, q)), z1 * IS);
      spl_store_act(out, j, q, ch, z0, z1);
    };
    spl_stage<8, 2, 0, 8, decltype(NCN)::value, 4>(ring, cur, nxt, lane, SplLdsB{in + j * SPL_ROW + 16 * q}, pre, epi);
    __syncthreads();
  };
  stage(IC<8>(), ftc, chunks(sdf_off_R(7)), buf0, buf1, IC<2>());
  stage(IC<7>(), chunks(sdf_off_R(7)), chunks(sdf_off_R(6)), buf1, buf0, IC<2>());
  stage(IC<6>(), chunks(sdf_off_R(6)), chunks(sdf_off_R(5)), buf0, buf1, IC<2>());
  stage(IC<5>(), chunks(sdf_off_R(5)), chunks(sdf_off_R(4)), buf1, buf0, IC<2>());
  stage(IC<4>(), chunks(sdf_off_R(4)), chunks(sdf_off_R(3)), buf0, buf1, IC<2>());
  stage(IC<3>(), chunks(sdf_off_R(3)), chunks(sdf_off_R(2)), buf1, buf0, IC<2>());
  stage(IC<2>(), chunks(sdf_off_R(2)), chunks(sdf_off_R(1)), buf0, buf1, IC<2>());
  const char* const r0c = (wave < 2) ? W + (size_t)SDF_OFF_R0 * 4 + (size_t)wave * 32768 : nullptr;
  stage(IC<1>(), chunks(sdf_off_R(1)), r0c, buf1, buf0, IC<1>());

  // R0: adjoint of the 39 embedding entries (waves 0, 1), then through the encoding on wave 0 (same tail as the forward's gradient)
  if (wave < 2) {
    auto pre = [&](auto) { return 0; };
    auto epi = [&](auto, f32x4 acc0, f32x4 acc1, int) {
      *reinterpret_cast<f32x4*>(Gl + wave * 8) = acc0;
      *reinterpret_cast<f32x4*>(Gl + wave * 8 + 4) = acc1;
    };
    spl_stage<8, 1, 0, 0, 0, 4>(ring, r0c, nullptr, lane, SplLdsB{buf0 + j * SPL_ROW + 16 * q}, pre, epi);
  }
  __syncthreads();
  if (wave == 0) {
    float ge[16], sk[12];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(Gl + b * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) ge[b * 4 + r] = v[r];
    }
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(Gl + 16 + b * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) sk[b * 4 + r] = v[r];
    }
    float dx[3] = {0.f, 0.f, 0.f};
    {
      float dc[39];
      nerf_enc_dall<3, 6>(x3, dc);
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const int e = b * 16 + 4 * qq + r;
            if (e < 39) dx[nerf_enc_dim<3, 6>(e)] += (q == qq) ? ge[b * 4 + r] * dc[e] : 0.0f;
            const int es = (b + 13) * 16 + 4 * qq + r - 217;
            if (es >= 0 && es < 39) dx[nerf_enc_dim<3, 6>(es)] += (q == qq) ? sk[b * 4 + r] * dc[es] : 0.0f;
          }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      dx[c] += __shfl_xor(dx[c], 16, 64);
      dx[c] += __shfl_xor(dx[c], 32, 64);
    }
    if (q == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) a.pbar[row * 3 + c] = dx[c] * 3.0f * IS;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// sdf + d sdf / dx without saves (sdf_kernel<1, 1>: what the shadow rays' alpha needs, models/neus_hint_model.py:335-336) in the
// same form.  sigma' never leaves the wave that produced it: 8 layers x 2 chunks x 8 values as unorm16 pairs = 64 VGPRs (the
// 16-point kernel parks the same words in a global scratch); t_7 goes straight into R7's B rows (no feature stage in between).
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int SPLG_LDS_BYTES = SPLT_LDS_BYTES;

__global__ __launch_bounds__(256, 2) void sdf_grad_split_kernel(const SdfArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  const char* const W = reinterpret_cast<const char*>(a.w);
  char* const buf0 = smem;
  char* const buf1 = smem + SPL_BUF;
  float* const tab = reinterpret_cast<float*>(smem + SPLT_OFF_TAB);
  char* const h7 = smem + SPLT_OFF_H7;
  float* const G = reinterpret_cast<float*>(smem + SPLT_OFF_G);
  float* const Gl = G + (j * 4 + q) * SPLT_G_FLOATS;

  const long long P = (long long)blockIdx.x * TILE_PTS + j;
  const bool valid = P < a.npts;
  const long long Pc = valid ? P : a.npts - 1;
  const long long ray = Pc / a.n_per_ray;
  const int jj = (int)(Pc - ray * a.n_per_ray);
  const float tt = a.t[ray * a.t_stride + jj];
  float x3[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) x3[c] = (a.ro[ray * 3 + c] + a.rd[ray * 3 + c] * tt) * 3.0f;  // inputs * scale
  {
    const f32x4* const bsrc = reinterpret_cast<const f32x4*>(a.b);
    f32x4* const bdst = reinterpret_cast<f32x4*>(tab);
    const f32x4 v0 = bsrc[threadIdx.x], v1 = bsrc[256 + threadIdx.x];
    const f32x4 v2 = (threadIdx.x < 64) ? bsrc[512 + threadIdx.x] : f32x4{0.f, 0.f, 0.f, 0.f};
    const float hv = a.head[threadIdx.x], hl = a.head[256];
    bdst[threadIdx.x] = v0;
    bdst[256 + threadIdx.x] = v1;
    if (threadIdx.x < 64) bdst[512 + threadIdx.x] = v2;
    tab[9 * 256 + threadIdx.x] = hv;
    if (threadIdx.x == 0) tab[9 * 256 + 256] = hl;
  }
  __builtin_amdgcn_sched_barrier(0);

  SplRing<4> ring;
  const char* const l0c = W + (size_t)SDF_OFF_L0 * 4 + (size_t)(2 * wave) * 8192;
  auto chunks = [&](int float_off) { return W + (size_t)float_off * 4 + (size_t)(2 * wave) * 32768; };
  spl_prologue<2, 2, 8, 4>(ring, l0c, chunks(sdf_off_L(1)), lane);
  __builtin_amdgcn_sched_barrier(0);

  Act<1, 4> emb;
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
  __syncthreads();     // the tables are in LDS

  u32x4 sg[7][2];      // sigma' of layers 0..6, this wave's two chunks: unorm16 pairs exactly as dsig_store<1> packs them

  auto fwd = [&](auto SC, const char* cur, const char* nxt, char* out, const auto& bsrc) {
    constexpr int S = decltype(SC)::value;
    auto pre = [&](auto CIC) {
      constexpr int CI = decltype(CIC)::value;
      const int ch = 2 * wave + CI;
      SplPreF p;
      p.a0 = *reinterpret_cast<const f32x4*>(tab + S * 256 + (2 * ch) * 16 + 4 * q);
      p.a1 = *reinterpret_cast<const f32x4*>(tab + S * 256 + (2 * ch + 1) * 16 + 4 * q);
      if constexpr (S == 7) {
        p.b0 = *reinterpret_cast<const f32x4*>(tab + 9 * 256 + (2 * ch) * 16 + 4 * q);
        p.b1 = *reinterpret_cast<const f32x4*>(tab + 9 * 256 + (2 * ch + 1) * 16 + 4 * q);
      }
      return p;
    };
    auto epi = [&](auto CIC, f32x4 acc0, f32x4 acc1, const SplPreF& p) {
      constexpr int CI = decltype(CIC)::value;
      const int ch = 2 * wave + CI;
      f32x4 h0, h1, d0, d1;
      softplus100_4<true>(acc0 + p.a0, h0, d0);
      softplus100_4<true>(acc1 + p.a1, h1, d1);
      if constexpr (S == 3) {
        if (wave == 3) {
          constexpr int chs = 6 + CI;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (chs == 7) {
              h0[r] = nerf_enc_entry_q<3, 6>(x3, (2 * chs) * 16 + r - 217, q);
              d0[r] = 0.0f;
            }
            if ((2 * chs + 1) * 16 + 4 * q + r - 217 >= 0) {
              h1[r] = nerf_enc_entry_q<3, 6>(x3, (2 * chs + 1) * 16 + r - 217, q);
              d1[r] = 0.0f;
            }
          }
        }
      }
      if constexpr (S == 7) {
        // head operand for wave 0; t_7 = sigma'_7 * w_s / 3 is the reverse chain's first B operand
        char* const p7 = h7 + j * SPLT_ROW7 + ((2 * ch) * 16 + 4 * q) * 4;
        *reinterpret_cast<f32x4*>(p7) = h0;
        *reinterpret_cast<f32x4*>(p7 + 64) = h1;
        spl_store_act(out, j, q, ch, d0 * (p.b0 * (1.0f / 3.0f)), d1 * (p.b1 * (1.0f / 3.0f)));
      } else {
        sg[S][CI] = u32x4{unorm16x2(d0[0], d0[1]), unorm16x2(d0[2], d0[3]), unorm16x2(d1[0], d1[1]), unorm16x2(d1[2], d1[3])};
        spl_store_act(out, j, q, ch, h0, h1);
      }
    };
    if constexpr (S == 0) spl_stage<2, 2, 0, 8, 2, 4>(ring, cur, nxt, lane, bsrc, pre, epi);
    else spl_stage<8, 2, 0, 8, 2, 4>(ring, cur, nxt, lane, bsrc, pre, epi);
    __syncthreads();
  };
  auto ldsb = [&](const char* in) { return SplLdsB{in + j * SPL_ROW + 16 * q}; };
  fwd(IC<0>(), l0c, chunks(sdf_off_L(1)), buf0, SplRegB{&emb});
  fwd(IC<1>(), chunks(sdf_off_L(1)), chunks(sdf_off_L(2)), buf1, ldsb(buf0));
  fwd(IC<2>(), chunks(sdf_off_L(2)), chunks(sdf_off_L(3)), buf0, ldsb(buf1));
  fwd(IC<3>(), chunks(sdf_off_L(3)), chunks(sdf_off_L(4)), buf1, ldsb(buf0));
  fwd(IC<4>(), chunks(sdf_off_L(4)), chunks(sdf_off_L(5)), buf0, ldsb(buf1));
  fwd(IC<5>(), chunks(sdf_off_L(5)), chunks(sdf_off_L(6)), buf1, ldsb(buf0));
  fwd(IC<6>(), chunks(sdf_off_L(6)), chunks(sdf_off_L(7)), buf0, ldsb(buf1));
  fwd(IC<7>(), chunks(sdf_off_L(7)), chunks(sdf_off_R(7)), buf1, ldsb(buf0));

  // ---- reverse chain R7..R1: g <- W_l^T (sigma'_l * g) ----
  auto rev = [&](auto LC, const char* cur, const char* nxt, const char* in, char* out, auto NCN) {
    constexpr int L = decltype(LC)::value;
    auto pre = [&](auto) { return 0; };
    auto epi = [&](auto CIC, f32x4 acc0, f32x4 acc1, int) {
      constexpr int CI = decltype(CIC)::value;
      const int ch = 2 * wave + CI;
      if constexpr (L == 4) {
        if (wave == 3) {
          constexpr int chs = 6 + CI;
          if (2 * chs >= 13) *reinterpret_cast<f32x4*>(Gl + 16 + (2 * chs - 13) * 4) = acc0;
          *reinterpret_cast<f32x4*>(Gl + 16 + (2 * chs + 1 - 13) * 4) = acc1;
        }
      }
      const u32x4 v = sg[L - 1][CI];          // dsig_decode<1>
      const f32x4 d0 = f32x4{(float)(v[0] & 0xffffu), (float)(v[0] >> 16), (float)(v[1] & 0xffffu), (float)(v[1] >> 16)} * (1.0f / 65535.0f);
      const f32x4 d1 = f32x4{(float)(v[2] & 0xffffu), (float)(v[2] >> 16), (float)(v[3] & 0xffffu), (float)(v[3] >> 16)} * (1.0f / 65535.0f);
      spl_store_act(out, j, q, ch, acc0 * d0, acc1 * d1);
    };
    spl_stage<8, 2, 0, 8, decltype(NCN)::value, 4>(ring, cur, nxt, lane, ldsb(in), pre, epi);
    __syncthreads();
  };
  rev(IC<7>(), chunks(sdf_off_R(7)), chunks(sdf_off_R(6)), buf1, buf0, IC<2>());
  rev(IC<6>(), chunks(sdf_off_R(6)), chunks(sdf_off_R(5)), buf0, buf1, IC<2>());
  rev(IC<5>(), chunks(sdf_off_R(5)), chunks(sdf_off_R(4)), buf1, buf0, IC<2>());
  rev(IC<4>(), chunks(sdf_off_R(4)), chunks(sdf_off_R(3)), buf0, buf1, IC<2>());
  rev(IC<3>(), chunks(sdf_off_R(3)), chunks(sdf_off_R(2)), buf1, buf0, IC<2>());
  rev(IC<2>(), chunks(sdf_off_R(2)), chunks(sdf_off_R(1)), buf0, buf1, IC<2>());
  const char* const r0c = (wave < 2) ? W + (size_t)SDF_OFF_R0 * 4 + (size_t)wave * 32768 : nullptr;
  rev(IC<1>(), chunks(sdf_off_R(1)), r0c, buf1, buf0, IC<1>());

  if (wave < 2) {
    auto pre = [&](auto) { return 0; };
    auto epi = [&](auto, f32x4 acc0, f32x4 acc1, int) {
      *reinterpret_cast<f32x4*>(Gl + wave * 8) = acc0;
      *reinterpret_cast<f32x4*>(Gl + wave * 8 + 4) = acc1;
    };
    spl_stage<8, 1, 0, 0, 0, 4>(ring, r0c, nullptr, lane, ldsb(buf0), pre, epi);
  }
  __syncthreads();

  if (wave == 0) {
    float head_part = 0.0f;
    const char* const r7 = h7 + j * SPLT_ROW7;
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(tab + 9 * 256 + (2 * ch) * 16 + 4 * q);
      const f32x4 w1 = *reinterpret_cast<const f32x4*>(tab + 9 * 256 + (2 * ch + 1) * 16 + 4 * q);
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(r7 + ((2 * ch) * 16 + 4 * q) * 4);
      const f32x4 h1 = *reinterpret_cast<const f32x4*>(r7 + ((2 * ch + 1) * 16 + 4 * q) * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        head_part += w0[r] * h0[r];
        head_part += w1[r] * h1[r];
      }
    }
    float part = head_part;
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);
    if (valid && q == 0) a.sdf[ray * a.sdf_stride + jj] = (part + tab[9 * 256 + 256]) / 3.0f;

    float ge[16], sk[12];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(Gl + b * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) ge[b * 4 + r] = v[r];
    }
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(Gl + 16 + b * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) sk[b * 4 + r] = v[r];
    }
    float dx[3] = {0.f, 0.f, 0.f};
    {
      float dc[39];
      nerf_enc_dall<3, 6>(x3, dc);
#pragma unroll
      for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const int e = b * 16 + 4 * qq + r;
            if (e < 39) dx[nerf_enc_dim<3, 6>(e)] += (q == qq) ? ge[b * 4 + r] * dc[e] : 0.0f;
            const int es = (b + 13) * 16 + 4 * qq + r - 217;
            if (es >= 0 && es < 39) dx[nerf_enc_dim<3, 6>(es)] += (q == qq) ? sk[b * 4 + r] * dc[es] : 0.0f;
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

}  // namespace nrh
