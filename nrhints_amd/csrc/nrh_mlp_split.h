// The channel-split GEMM stage for SMALL batches (f16x3 only): one 16-point tile per workgroup, the 256 output channels of a stage
// split over the workgroup's four waves - the generic form of what nrh_sdf_split.hip does for the sdf value, for the training
// kernels (nrh_sdf_train_split.hip).  Same arithmetic as run_stage<1, ...> of nrh_mlp.h (same packed stages, same six MFMAs per
// K step and output-block pair in the same order, cross terms joined the same way): a wave's chunk 2w + ci produces exactly the bits
// chunk 2w + ci of the 16-point kernels produces, in the same lane / register positions (D-layout), so their epilogues carry over.
//
//   weights      each wave streams only its own chunks, L2 -> registers, through a ring of A-operand sets (one K step of
//                one chunk = 4 x 16 bytes per lane); the ring runs ACROSS stages: element n of the wave's sequence of sets lives
//                in slot n % RD and is requested when element n - RD has been multiplied
//   activations  the previous stage's outputs as fp16 hi | lo rows in LDS (SPL_ROW bytes per point and term), read two K steps
//                ahead of their MFMAs; written by the epilogues through spl_store_act; one __syncthreads() per stage
//   epilogue loads  issued by pre(ci) BEFORE the chunk's K loop, i.e. before that loop's ring requests: vmcnt retires in order, a
//                load issued behind ring requests would drain the ring when waited for
#pragma once
#include <type_traits>
#include "nrh_mlp.h"

namespace nrh {

template <int V>
using IC = std::integral_constant<int, V>;

constexpr int SPL_ROW = 544;            // bytes of one point's 256 fp16 values (+32: the ds_read_b128 lane groups hit 16 distinct 16-byte slots)
constexpr int SPL_TERM = 16 * SPL_ROW;  // one term (hi or lo) of one 16-point tile
constexpr int SPL_BUF = 2 * SPL_TERM;   // one activation buffer: hi rows, then lo rows

template <int RD>   // ring depth in sets: the prefetch distance (4 sets = 64 VGPRs, 8 = 128)
struct SplRing {
  static constexpr int DEPTH = RD;
  u32x4 v[RD][4];   // {hi, lo of output block 2ch; hi, lo of block 2ch + 1} of one K step
};

// request set (chunk, K step s) of a stage with KS K steps into `slot` (chunk: this wave's chunk base, lane-uniform)
template <int RD>
__device__ __forceinline__ void spl_issue(SplRing<RD>& r, int slot, const char* chunk, int KS, int s, int lane) {
  const u32x4* p = reinterpret_cast<const u32x4*>(chunk) + lane;
  r.v[slot][0] = p[((0 * KS + s) * 2 + 0) * 64];
  r.v[slot][1] = p[((0 * KS + s) * 2 + 1) * 64];
  r.v[slot][2] = p[((1 * KS + s) * 2 + 0) * 64];
  r.v[slot][3] = p[((1 * KS + s) * 2 + 1) * 64];
}

// the first RD elements of a wave's sequence: NC0 chunks x KS0 K steps of the first stage, then the second stage's first sets
template <int KS0, int NC0, int KS1, int RD>
__device__ __forceinline__ void spl_prologue(SplRing<RD>& r, const char* first, const char* second, int lane) {
#pragma unroll
  for (int n = 0; n < RD; ++n) {
    if (n < NC0 * KS0) spl_issue(r, n, first + (n / KS0) * (KS0 * 4096), KS0, n % KS0, lane);
    else spl_issue(r, n, second + ((n - NC0 * KS0) / KS1) * (KS1 * 4096), KS1, (n - NC0 * KS0) % KS1, lane);
  }
}

// the two blocks (2ch, 2ch + 1) of one point's chunk ch as the next stage's B rows: hi | lo split exactly as Act<1>::set_chunk
__device__ __forceinline__ void spl_store_act(char* out, int j, int q, int ch, const f32x4 o0, const f32x4 o1) {
  uint32_t hw[4], lw[4];
  split_pack2(o0[0], o0[1], hw[0], lw[0]);
  split_pack2(o0[2], o0[3], hw[1], lw[1]);
  split_pack2(o1[0], o1[1], hw[2], lw[2]);
  split_pack2(o1[2], o1[3], hw[3], lw[3]);
  char* const p = out + j * SPL_ROW + (32 * ch + 8 * q) * 2;
  *reinterpret_cast<u32x4*>(p) = u32x4{hw[0], hw[1], hw[2], hw[3]};
  *reinterpret_cast<u32x4*>(p + SPL_TERM) = u32x4{lw[0], lw[1], lw[2], lw[3]};
}

// B operands of K step s from an activation buffer / from the embedding registers (KB = 4: two K steps)
struct SplLdsB {
  const char* row;    // in + j * SPL_ROW + 16 * q
  __device__ __forceinline__ void operator()(int s, u32x4& bh, u32x4& bl) const {
    bh = *reinterpret_cast<const u32x4*>(row + 64 * s);
    bl = *reinterpret_cast<const u32x4*>(row + 64 * s + SPL_TERM);
  }
};
// accumulator start values of a chunk (run_stage's HAS_INIT: one layer's K split in two stages); default: zeros
struct SplNoInit {
  template <typename C>
  __device__ __forceinline__ void operator()(C, f32x4&, f32x4&) const {}
};
template <int KB>
struct SplRegBk {       // B operands from an Act<1, KB> in registers (KB / 2 K steps)
  const Act<1, KB>* e;
  __device__ __forceinline__ void operator()(int s, u32x4& bh, u32x4& bl) const {
    bh = u32x4{e->h[s * 4 + 0], e->h[s * 4 + 1], e->h[s * 4 + 2], e->h[s * 4 + 3]};
    bl = u32x4{e->l[s * 4 + 0], e->l[s * 4 + 1], e->l[s * 4 + 2], e->l[s * 4 + 3]};
  }
};
struct SplRegB {
  const Act<1, 4>* e;
  __device__ __forceinline__ void operator()(int s, u32x4& bh, u32x4& bl) const {
    bh = u32x4{e->h[s * 4 + 0], e->h[s * 4 + 1], e->h[s * 4 + 2], e->h[s * 4 + 3]};
    bl = u32x4{e->l[s * 4 + 0], e->l[s * 4 + 1], e->l[s * 4 + 2], e->l[s * 4 + 3]};
  }
};

// One stage for this wave: NC chunks (consecutive, KS * 4 KiB each, from `cur`) x KS K steps.  BASE = the stage's first element's
// slot; on entry the wave's next RD elements are in flight, and so they are on exit (the tail of the K loops requests the first
// sets of the next stage: NC_N chunks x KS_N K steps at `nxt`; nxt == nullptr: nothing follows for this wave).
//   pre(IC<ci>) -> P   issues the epilogue's global loads;   epi(IC<ci>, acc0, acc1, P)   consumes the chunk's two blocks
//   init(IC<ci>, acc0, acc1)   optional: the chunk's accumulator start values (the first MFMA adds to them, as run_stage's HAS_INIT)
template <int KS, int NC, int BASE, int KS_N, int NC_N, int RD, typename BFn, typename Pre, typename Epi, typename Init = SplNoInit>
__device__ __forceinline__ void spl_stage(SplRing<RD>& ring, const char* cur, const char* nxt, int lane, const BFn& bsrc, Pre&& pre,
                                          Epi&& epi, const Init& init = Init()) {
  constexpr int S = NC * KS, SN = NC_N * KS_N;
  u32x4 bh[3], bl[3];
  if (S > 0) bsrc(0, bh[0], bl[0]);
  if (S > 1) bsrc(1 % KS, bh[1], bl[1]);
  auto chunk = [&](auto CIC) {
    constexpr int CI = decltype(CIC)::value;
    const auto pv = pre(CIC);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f}, c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
    init(CIC, acc0, acc1);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int i = CI * KS + s;
      const int slot = (BASE + i) % RD;
      if (i + 2 < S) bsrc((i + 2) % KS, bh[(i + 2) % 3], bl[(i + 2) % 3]);
      const f16x8 h = __builtin_bit_cast(f16x8, bh[i % 3]), l = __builtin_bit_cast(f16x8, bl[i % 3]);
      const f16x8 ah0 = __builtin_bit_cast(f16x8, ring.v[slot][0]), al0 = __builtin_bit_cast(f16x8, ring.v[slot][1]);
      const f16x8 ah1 = __builtin_bit_cast(f16x8, ring.v[slot][2]), al1 = __builtin_bit_cast(f16x8, ring.v[slot][3]);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, h, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, h, acc1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, l, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, l, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, h, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, h, c1, 0, 0, 0);
      const int n = i + RD;
      if (n < S) {
        spl_issue(ring, slot, cur + (n / KS) * (KS * 4096), KS, n % KS, lane);
      } else if (SN > 0 && n - S < SN) {
        if (nxt != nullptr) spl_issue(ring, slot, nxt + ((n - S) / (KS_N > 0 ? KS_N : 1)) * (KS_N * 4096), KS_N, (n - S) % (KS_N > 0 ? KS_N : 1), lane);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    acc0 += c0 * LO_UNSCALE;
    acc1 += c1 * LO_UNSCALE;
    epi(CIC, acc0, acc1, pv);
  };
  if constexpr (NC > 0) chunk(IC<0>());
  if constexpr (NC > 1) chunk(IC<1>());
}

}  // namespace nrh
