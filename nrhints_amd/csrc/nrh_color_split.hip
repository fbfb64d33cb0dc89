// The reflectance network's training forward and adjoint sweep for SMALL batches (precision f16x3): color_kernel<1, MKB, true> and
// color_adjoint_kernel<1, MKB> (nrh_color.hip; reference fields/reflectance_network.py:68-96 and what autograd does through it)
// with one 16-point tile per workgroup and every stage's output channels split over the workgroup's four waves (nrh_mlp_split.h).
//
// A 64-ray training batch (the reference's per-rank share under 8-way DDP, trainer/trainer.py:116-123) is 512 tiles.  The 16-point
// kernels give eight of them to an 8-wave workgroup - 64 workgroups, a quarter of the CUs, each wave walking its tile through the
// five stages alone (60 us forward, 66 us adjoint at 64 rays).  Here a tile's MFMA work runs on the four matrix cores of a CU at
// once, on 512 workgroups.
//
// Same contract as nrh_color_train_forward / nrh_color_train_backward (include/nrhints_hip.h): colour, save_h, save_misc; zbar,
// fbar, mbar - bit for bit (tests/test_gpu_split.py): the stage arithmetic is run_stage's (same packed stages, same MFMA sequence
// per output block, layer 0's K split with the feature part as the second part's start values), the epilogues are the 16-point
// kernels' on the same lane / register positions.  What differs is who holds what: the inputs are staged once into LDS as the
// fp16 hi | lo rows every wave's B operands are read from (wave w converts feature chunks 2w, 2w + 1 and misc chunk w), the
// activations go from stage to stage through LDS, the 3-row output stage (C4) runs on wave 0, the misc-input adjoint (T0b: 128
// or 64 rows) one chunk per wave.  The float32 hand-offs only: the 16-bit hand-offs start above this kernel's batch range.
#include "nrh_mlp_split.h"      // (included by nrh_api.hip after nrh_color.hip, whose argument structs and helpers it uses)

namespace nrh {

constexpr int SPLC_OFF_TAB = 3 * SPL_BUF;                      // three activation buffers, then the bias table
constexpr int SPLC_LDS_BYTES = SPLC_OFF_TAB + COL_BIAS_FLOATS * 4;
constexpr int SPLCA_LDS_BYTES = 2 * SPL_BUF;

struct SplPreC {
  f32x4 a0, a1;     // forward: bias of the chunk's two blocks;  adjoint: the ReLU outputs that mask them
};

template <int MKB>
__global__ __launch_bounds__(256, 2) void color_train_split_kernel(const ColorArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  const char* const W = reinterpret_cast<const char*>(a.w);
  char* const buf0 = smem;
  char* const buf1 = smem + SPL_BUF;
  char* const buf2 = smem + 2 * SPL_BUF;
  float* const tab = reinterpret_cast<float*>(smem + SPLC_OFF_TAB);
  constexpr int KSM = MKB / 2;                                  // K steps of the misc part of layer 0

  const long long P = (long long)blockIdx.x * TILE_PTS + j;     // npts % 16 == 0 (128 samples per ray): every tile is whole
  auto save_rows = [&](float* base, int l, int width, int ch, const f32x4 v0, const f32x4 v1) {
    float* p = base + ((size_t)l * (size_t)a.npts + (size_t)P) * width + 4 * q;
    st_stream(reinterpret_cast<f32x4*>(p + (2 * ch) * 16), v0);
    st_stream(reinterpret_cast<f32x4*>(p + (2 * ch + 1) * 16), v1);
  };

  // ---- inputs and the bias table -> LDS, BEFORE the weight ring starts (vmcnt retires in order) ----
  {
    // feature rows: this wave's two chunks of the 16 points, split exactly as Act<1>::set_chunk splits them
    const float* ft = a.feat + (size_t)P * 256 + 4 * q;
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
      const int ch = 2 * wave + ci;
      const f32x4 v0 = ld_stream(reinterpret_cast<const f32x4*>(ft + (2 * ch) * 16));
      const f32x4 v1 = ld_stream(reinterpret_cast<const f32x4*>(ft + (2 * ch + 1) * 16));
      spl_store_act(buf0, j, q, ch, v0, v1);
    }
    // the non-feature inputs: entry m = 16 b + 4 q + r of [p, n, raymisc[0 .. 98]], chunk `wave` (MKB = 4: waves 0 and 1)
    if (wave < MKB / 2) {
      const int ch = wave;
      float pn[6];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        pn[c] = a.pts[P * 3 + c];
        pn[3 + c] = a.nhat[P * 3 + c];
      }
      const float* rm = a.raymisc + (P >> a.misc_shift) * RAYMISC_STRIDE;
      float o[8];
#pragma unroll
      for (int r8 = 0; r8 < 8; ++r8) {
        const int b = 2 * ch + (r8 >> 2), r = r8 & 3;
        const int m = b * 16 + 4 * q + r;
        float v;
        if (b == 0) {
          const float per = sel_q<6>(pn, r, q);
          const float ld = rm[(m >= 6) ? m - 6 : 0];
          v = (m < 6) ? per : ld;
        } else {
          v = (m < col_misc(MKB)) ? rm[(m < col_misc(MKB)) ? m - 6 : 0] : 0.0f;
        }
        o[r8] = v;
      }
      const f32x4 o0 = {o[0], o[1], o[2], o[3]}, o1 = {o[4], o[5], o[6], o[7]};
      spl_store_act(buf1, j, q, ch, o0, o1);
      save_rows(a.save_misc, 0, 16 * MKB, ch, o0, o1);
    }
    for (int i = threadIdx.x; i < COL_BIAS_FLOATS / 4; i += 256)
      reinterpret_cast<f32x4*>(tab)[i] = reinterpret_cast<const f32x4*>(a.b)[i];
  }
  __builtin_amdgcn_sched_barrier(0);

  SplRing<4> ring;
  auto chunks = [&](int float_off) { return W + (size_t)float_off * 4 + (size_t)(2 * wave) * 32768; };             // a 256 x 256 stage
  const char* const c0b = W + (size_t)COL_OFF_C0B * 4 + (size_t)(2 * wave) * (KSM * 4096);                         // 256 x (16 MKB)
  const char* const c4 = (wave == 0) ? W + (size_t)col_off_C4(MKB) * 4 : nullptr;                                  // one chunk
  spl_prologue<8, 2, KSM, 4>(ring, chunks(COL_OFF_C0A), c0b, lane);
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();     // inputs and table are in LDS

  auto ldsb = [&](const char* in) { return SplLdsB{in + j * SPL_ROW + 16 * q}; };
  auto bias = [&](int l, int ch) {
    SplPreC p;
    p.a0 = *reinterpret_cast<const f32x4*>(tab + l * 256 + (2 * ch) * 16 + 4 * q);
    p.a1 = *reinterpret_cast<const f32x4*>(tab + l * 256 + (2 * ch + 1) * 16 + 4 * q);
    return p;
  };

  // ---- C0a: the feature part of layer 0; its two chunks stay in registers as C0b's start values ----
  f32x4 part[2][2];
  {
    auto pre = [&](auto) { return 0; };
    auto epi = [&](auto CIC, f32x4 acc0, f32x4 acc1, int) {
      constexpr int CI = decltype(CIC)::value;
      part[CI][0] = acc0;
      part[CI][1] = acc1;
    };
    spl_stage<8, 2, 0, KSM, 2, 4>(ring, chunks(COL_OFF_C0A), c0b, lane, ldsb(buf0), pre, epi);
  }
  // ---- C0b: the per-sample + per-ray part, started from C0a's sums; bias, ReLU ----
  {
    auto pre = [&](auto CIC) { return bias(0, 2 * wave + decltype(CIC)::value); };
    auto epi = [&](auto CIC, f32x4 acc0, f32x4 acc1, const SplPreC& p) {
      const int ch = 2 * wave + decltype(CIC)::value;
      const f32x4 h0 = relu4(acc0 + p.a0), h1 = relu4(acc1 + p.a1);
      save_rows(a.save_h, 0, 256, ch, h0, h1);
      spl_store_act(buf2, j, q, ch, h0, h1);
    };
    auto init = [&](auto CIC, f32x4& acc0, f32x4& acc1) {
      constexpr int CI = decltype(CIC)::value;
      acc0 = part[CI][0];
      acc1 = part[CI][1];
    };
    spl_stage<KSM, 2, 0, 8, 2, 4>(ring, c0b, chunks(col_off_C(1, MKB)), lane, ldsb(buf1), pre, epi, init);
    __syncthreads();
  }
  // ---- C1..C3 ----
  auto layer = [&](auto LC, const char* cur, const char* nxt, const char* in, char* out, auto NCN) {
    constexpr int L = decltype(LC)::value;
    auto pre = [&](auto CIC) { return bias(L, 2 * wave + decltype(CIC)::value); };
    auto epi = [&](auto CIC, f32x4 acc0, f32x4 acc1, const SplPreC& p) {
      const int ch = 2 * wave + decltype(CIC)::value;
      const f32x4 h0 = relu4(acc0 + p.a0), h1 = relu4(acc1 + p.a1);
      save_rows(a.save_h, L, 256, ch, h0, h1);
      spl_store_act(out, j, q, ch, h0, h1);
    };
    spl_stage<8, 2, 0, 8, decltype(NCN)::value, 4>(ring, cur, nxt, lane, ldsb(in), pre, epi);
    __syncthreads();
  };
  layer(IC<1>(), chunks(col_off_C(1, MKB)), chunks(col_off_C(2, MKB)), buf2, buf0, IC<2>());
  layer(IC<2>(), chunks(col_off_C(2, MKB)), chunks(col_off_C(3, MKB)), buf0, buf1, IC<2>());
  layer(IC<3>(), chunks(col_off_C(3, MKB)), c4, buf1, buf2, IC<1>());
  // ---- C4: 3 output rows (block 0, lanes q == 0 hold r = 0..2) + sigmoid, on wave 0 ----
  if (wave == 0) {
    auto pre = [&](auto) {
      SplPreC p;
      p.a0 = *reinterpret_cast<const f32x4*>(tab + 4 * 256 + 4 * q);
      p.a1 = p.a0;
      return p;
    };
    auto epi = [&](auto, f32x4 acc0, f32x4 acc1, const SplPreC& p) {
      (void)acc1;
      if (q == 0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) a.color[P * 3 + r] = sigmoidf_(acc0[r] + p.a0[r]);
      }
    };
    spl_stage<8, 1, 0, 0, 0, 4>(ring, c4, nullptr, lane, ldsb(buf2), pre, epi);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// adjoint sweep:  zbar_3 = (W4^T zbar4) [h_3 > 0];  zbar_{l-1} = (W_l^T zbar_l) [h_{l-1} > 0];  fbar = W0feat^T zbar_0;
//                 mbar = W0misc^T zbar_0
// ------------------------------------------------------------------------------------------------------------------
template <int MKB>
__global__ __launch_bounds__(256, 2) void color_adjoint_split_kernel(const ColorAdjArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  const char* const W = reinterpret_cast<const char*>(a.wt);
  char* const buf0 = smem;
  char* const buf1 = smem + SPL_BUF;
  const float S = a.adj_scale, IS = 1.0f / S;     // ColorAdjArgs.adj_scale: linear in the seeds, outputs leave as 1 / S

  const long long row = (long long)blockIdx.x * TILE_PTS + j;
  auto rows_ptr = [&](const float* base, int l, int blk) {
    return reinterpret_cast<const f32x4*>(base + ((size_t)l * (size_t)a.npts + (size_t)row) * 256 + blk * 16 + 4 * q);
  };
  auto store_rows = [&](float* base, int l, int width, int ch, const f32x4 v0, const f32x4 v1) {
    float* p = base + ((size_t)l * (size_t)a.npts + (size_t)row) * width + 4 * q;
    st_stream(reinterpret_cast<f32x4*>(p + (2 * ch) * 16), v0 * IS);
    st_stream(reinterpret_cast<f32x4*>(p + (2 * ch + 1) * 16), v1 * IS);
  };
  auto load_h = [&](int l, int ch) {
    SplPreC p;
    p.a0 = ld_stream(rows_ptr(a.save_h, l, 2 * ch));
    p.a1 = ld_stream(rows_ptr(a.save_h, l, 2 * ch + 1));
    return p;
  };

  // ---- T4's B operand: the adjoint of the 3 outputs (block 0, lanes q == 0, registers 0..2), in every wave ----
  Act<1, 2> z4;
  {
    float o[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) o[r] = (q == 0 && r < 3) ? a.zbar4[row * 3 + r] * S : 0.0f;
    z4.set_chunk(0, o);
  }
  __builtin_amdgcn_sched_barrier(0);

  SplRing<4> ring;
  auto chunks = [&](int float_off) { return W + (size_t)float_off * 4 + (size_t)(2 * wave) * 32768; };
  const char* const t4 = W + (size_t)(2 * wave) * 4096;                                                            // 256 x 32: one K step
  const char* const t0b = (wave < MKB / 2) ? W + (size_t)COLT_OFF_T0B * 4 + (size_t)wave * 32768 : nullptr;        // (16 MKB) x 256: a chunk per wave
  spl_prologue<1, 2, 8, 4>(ring, t4, chunks(colt_off_T(3)), lane);
  __builtin_amdgcn_sched_barrier(0);

  auto ldsb = [&](const char* in) { return SplLdsB{in + j * SPL_ROW + 16 * q}; };
  auto masked = [&](int l, char* out) {
    return [&, l, out](auto CIC, f32x4 acc0, f32x4 acc1, const SplPreC& p) {
      const int ch = 2 * wave + decltype(CIC)::value;
      f32x4 z0, z1;
#pragma unroll
      for (int r = 0; r < 4; ++r) { z0[r] = p.a0[r] > 0.0f ? acc0[r] : 0.0f; z1[r] = p.a1[r] > 0.0f ? acc1[r] : 0.0f; }
      store_rows(a.zbar, l, 256, ch, z0, z1);
      spl_store_act(out, j, q, ch, z0, z1);
    };
  };
  // ---- T4: 3 -> 256, masked by h_3 ----
  {
    auto pre = [&](auto CIC) { return load_h(3, 2 * wave + decltype(CIC)::value); };
    spl_stage<1, 2, 0, 8, 2, 4>(ring, t4, chunks(colt_off_T(3)), lane, SplRegBk<2>{&z4}, pre, masked(3, buf0));
    __syncthreads();
  }
  // ---- T3, T2, T1: zbar_{l-1} = (W_l^T zbar_l) [h_{l-1} > 0] ----
  auto sweep = [&](auto LC, const char* cur, const char* nxt, const char* in, char* out) {
    constexpr int L = decltype(LC)::value;
    auto pre = [&](auto CIC) { return load_h(L - 1, 2 * wave + decltype(CIC)::value); };
    spl_stage<8, 2, 2, 8, 2, 4>(ring, cur, nxt, lane, ldsb(in), pre, masked(L - 1, out));
    __syncthreads();
  };
  sweep(IC<3>(), chunks(colt_off_T(3)), chunks(colt_off_T(2)), buf0, buf1);
  sweep(IC<2>(), chunks(colt_off_T(2)), chunks(colt_off_T(1)), buf1, buf0);
  sweep(IC<1>(), chunks(colt_off_T(1)), chunks(COLT_OFF_T0A), buf0, buf1);
  // ---- T0a: adjoint of the feature input;  T0b: adjoint of the other inputs (one chunk per wave) ----
  {
    auto pre = [&](auto) { return 0; };
    auto epi = [&](auto CIC, f32x4 acc0, f32x4 acc1, int) { store_rows(a.fbar, 0, 256, 2 * wave + decltype(CIC)::value, acc0, acc1); };
    spl_stage<8, 2, 2, 8, 1, 4>(ring, chunks(COLT_OFF_T0A), t0b, lane, ldsb(buf1), pre, epi);
  }
  if (wave < MKB / 2) {
    auto pre = [&](auto) { return 0; };
    auto epi = [&](auto, f32x4 acc0, f32x4 acc1, int) { store_rows(a.mbar, 0, 16 * MKB, wave, acc0, acc1); };
    spl_stage<8, 1, 2, 0, 0, 4>(ring, t0b, nullptr, lane, ldsb(buf1), pre, epi);
  }
}

}  // namespace nrh
