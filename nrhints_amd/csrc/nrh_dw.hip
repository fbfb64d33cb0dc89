// Weight gradients of the training step on gfx950: every  dW = X^T Y  of the SDF and reflectance networks' backward
// (dW_l = zbar_l^T x_l + t_l^T abar_l, the feature head, the reflectance layers) as hand-written split-K MFMA GEMMs in ONE
// launch, plus the column sums that are the bias gradients - the replacement of the rocBLAS `Cijk_*` calls and the torch
// reductions that made up 36 % of a training step in round 2 (profiles/r02/rocprof_train_stats_v6.csv).
// Reference: what loss.backward() computes for nn.Linear under autograd (fields/sdf_field.py:81-101,
// fields/reflectance_network.py:52-66; trainer/trainer.py:278-279).
//
// Shape of the problem: X, Y are the row-major [P, 256] float32 arrays the sweep kernels saved (P = rays x 128 = 131 072 points
// for a 1024-ray batch), the product is a tiny [256, 256] with a huge reduction dimension, so the reduction is what gets split:
//   * one workgroup (4 waves, one per SIMD) owns a SLAB of points of one job and the whole 256 x N output of it; wave w owns
//     rows 64 w .. 64 w + 63 as 2 x NF accumulator tiles of v_mfma_f32_32x32x16_bf16 (NF = N / 32: 8, 4, 2 or 1);
//   * K step = 16 points.  Their rows arrive by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction straight into an LDS
//     raw stage, two steps ahead, no VGPR in between).  The MFMA wants K (points) contiguous per lane while the rows have the
//     channels contiguous, so a thread reads two adjacent channels of the 16 raw rows (ds_read_b64, conflict-free), splits every
//     value into bf16 hi + lo (v_cvt_pk_bf16_f32, round to nearest even) and writes 16 B per half per 8 points into the fragment
//     layout [row block][k half][row][8 k] (ds_write_b128 / ds_read_b128) - the transposition is the addressing, no value is
//     converted twice;
//   * a product is three MFMAs, hi*hi + hi*lo + lo*hi (the 3-term split of the f16x3 kernels, in bf16 because adjoints have no
//     a-priori range: bf16 keeps float32's exponent, so nothing is scaled and nothing can overflow); operands carry 16 mantissa
//     bits, products are exact in the fp32 accumulator: |error| <= 2^-16 per term, measured 4e-6 of an entry's own magnitude
//     against 2.5e-7 for an fp32 GEMM in another summation order (CHANGELOG.md section 7a; the parity bound is 1e-4);
//   * the kernel is bound by HBM latency x bytes in flight: with the rows loaded into registers one step ahead it ran at
//     1.5-2.3 TB/s (hipcc drains vmcnt to 0 before the first use of any loaded register, and copies the destination registers of
//     asm loads while they are in flight), hence the DMA: 64 KB per CU stay in flight across MFMAs, conversion and barriers;
//   * partial sums go to a workspace [item][256][256] (+ column sums [item][2][256]); dw_reduce_kernel adds the slabs of a job
//     in a FIXED order (deterministic, unlike float atomics), applies the job's scale / row limit / column map / transposition
//     and writes the gradient tensors.
// Algorithmic work per job: 2 * P * M * N flop (x npairs); bytes: P * (lda + ldb) * 4 read once.  A 1024-ray step: 20.1 full
// 256 x 256 pair-equivalents = 0.69 TFLOP (x 3 MFMA passes), 5.7 GB: HBM-bound (0.7 ms at 8 TB/s, 0.41 ms of MFMA at peak).
#pragma once
#include "nrh_common.h"

namespace nrhdw {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#ifndef NRH_DW_ABL
#define NRH_DW_ABL 0    // timing ablations (WRONG RESULTS): 1 no global loads, 2 no MFMAs, 4 no conversion / LDS writes
#endif
#ifdef NRH_DW_TIMING
// diagnosis builds (make variant DEFS=-DNRH_DW_TIMING): shader cycles per phase, summed over wave 0 of every workgroup
__device__ unsigned long long g_dw_cycles[8];
#define DW_STAMP(k) do { if (lane == 0 && wave == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[k] += t_ - tlast; tlast = t_; } } while (0)
#else
#define DW_STAMP(k) do { } while (0)
#endif
constexpr int MAX_JOBS = 24;
constexpr int THREADS = 256;
constexpr int KSTEP = 16;                       // points per K step
constexpr int KPTS = KSTEP;
constexpr int FRAG_STAGE = 32768;               // bf16 fragments of one K step: A_hi | A_lo | B_hi | B_lo, 8 KiB each
constexpr int RAW_STAGE = 32768;                // raw float32 rows of one K step: A 16 x 1 KiB | B 16 x (up to) 1 KiB
constexpr int LDS_RAW = 2 * FRAG_STAGE;
constexpr int LDS_BYTES = 2 * FRAG_STAGE + 2 * RAW_STAGE;      // 128 KiB
constexpr int SLOT_FLOATS = 256 * 256;          // one item's partial product
constexpr int CSUM_FLOATS = 512;                // one item's column sums: A (256) | B (256)

struct JobDev {
  const float* a[2];     // pair k: A_k [npts, lda_k], channels contiguous
  const float* b[2];     // pair k: B_k [npts, ldb_k]
  int lda[2], ldb[2];
  int npairs;            // 1 or 2 (both pairs accumulate into the same product)
  int m, n;              // channels of A / B that exist (<= 256); the rest reads as zero
  int slab0, slabs;      // this job's work items: [slab0, slab0 + slabs)
  int colsum;            // bit 0: column sums of A_0, bit 1: of B_0
  int half;              // all four operands are fp16 in the half-tiled layout (dw_item_half); full 256 x 256 products only
  int ta[2], tb[2];      // pair k: operand is TILED - [tile of 16 points][block 16][point 16][16 channels] (csrc/nrh_mlp.h, the layout
                         // the sweep kernels write h, t, abar, zbar in) instead of row-major; 256 channels only
};

struct DwArgs {
  JobDev job[MAX_JOBS];
  int njobs;
  int nsteps;            // npts / 16 (K steps)
  float* partial;        // [items][256][256]
  float* csum;           // [items][2][256]
};

__host__ __device__ constexpr int frow(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }

// bf16 hi/lo of 8 consecutive points of one channel -> two 16 B LDS words
__device__ __forceinline__ void split8(const float* v, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h = (__bf16)v[i];
    hi[i] = h;
    lo[i] = (__bf16)(v[i] - (float)h);
  }
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- LDS-DMA of the operand rows (global_load_lds_dwordx4: 16 B per lane, 1 KiB per wave instruction, no VGPR in between) ----
// Global address = gp + 16 * lane, LDS address = m0 + 16 * lane (csrc/nrh_mlp32.h dma_piece measured the addressing).  M0 cannot be
// named as a clobber (reserved register for hipcc); it is rewritten in front of every piece.
// lane16: the lane's byte offset in the SOURCE KiB (16 * lane for a row-major operand; for a tiled operand the lanes of block b
// fetch 16 * ((lane - 4 b) & 63), which lands point i of the block in LDS row (i + b) & 15 - see tiled_lane16)
__device__ __forceinline__ void dma_piece(const char* gp, uint32_t m0v, uint32_t lane16, uint64_t lanes) {
  // `lanes`: which lanes take part (the last piece of a narrow operand is partial, pieces past its end are empty).  EXEC is set
  // inside the statement - no branch for hipcc to build, so a piece can sit between two MFMAs of a straight-line block - and every
  // piece counts in vmcnt, empty or not: a wave ALWAYS issues 8 pieces per step, and one `s_waitcnt vmcnt(8)` serves all jobs.
  uint64_t save;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %4 nt\n\ts_mov_b64 exec, %0"
               : "=&s"(save) : "s"(lanes), "s"(m0v), "v"(lane16), "s"(gp) : "memory");
}
// Tiled operand, block b (= piece b of the tile's 16 KiB): global slot of LDS slot `lane` (slots are 16 bytes: [point 16][quarter 4])
// with the block's rows rotated by b - the conversion then reads channel pair (b, cc) of point i at row (i + b) & 15, and the four
// blocks a 32-lane LDS pass touches sit in four different 64-byte bank groups (conflict-free ds_read_b64)
__device__ __forceinline__ uint32_t tiled_lane16(int lane, int b) { return (uint32_t)((lane - 4 * b) & 63) * 16u; }
__device__ __forceinline__ uint32_t lds_off(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ const char* uni(const char* p) {
  const uint64_t g = (uint64_t)p;
  return (const char*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(g >> 32)) << 32) |
                       (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)g));
}
// wait until at most `keep` of this wave's DMA pieces (the youngest) are outstanding, then meet the other waves.  NOT
// __syncthreads(): that drains vmcnt to 0, i.e. also the pieces of the step requested last.
__device__ __forceinline__ void wait_pieces_and_barrier(bool newer_in_flight) {
  if (newer_in_flight) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NF>
__device__ __forceinline__ void dw_item(const JobDev& J, const int slab, const int nsteps_all, float* __restrict__ part,
                                        float* __restrict__ csum, char* smem) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lane16 = lane * 16;
  // this item's share of the points, in K steps of 16
  const int s0 = (int)((long long)nsteps_all * slab / J.slabs), s1 = (int)((long long)nsteps_all * (slab + 1) / J.slabs);
  const int nst = s1 - s0;
  const int total = nst * J.npairs;
  // Converter roles: waves 0, 1 split the A operand, waves 2, 3 the B operand; a thread owns two adjacent channels (8 bytes of
  // every raw row) when the operand is a full 256-channel array, one channel otherwise (the narrow B operands: 39-, 105-, 3- and
  // 1-column products, whose row strides need not be even).
  const bool is_b = wave >= 2;
  const int t2 = tid & 127;
  const bool wide = !is_b || NF == 8;
  const int lim = is_b ? J.n : J.m;
  const int c0 = wide ? 2 * t2 : t2;
  const bool ok0 = c0 < lim, ok1 = wide && (c0 + 1 < lim);
  // fragment homes of the thread's channel(s) in a K step's image: [row block][k half][row][8 x bf16]
  const int frag0 = (is_b ? 16384 : 0) + (c0 >> 5) * 1024 + (c0 & 31) * 16;
  const bool writes = wide || (t2 < 32 * NF);          // narrow B: only channels < 32 NF have a home
  const uint32_t raw_lds = lds_off(smem) + LDS_RAW;

  f32x16 acc[2][NF];
#pragma unroll
  for (int mf = 0; mf < 2; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mf][nf][r] = 0.0f;
  float sum0 = 0.0f, sum1 = 0.0f;

  // raw rows of step idx -> LDS raw stage idx & 1.  16 points of an operand are ONE contiguous block of 64 * ld bytes (rows are
  // dense), moved in 1 KiB pieces; piece p goes out from wave p & 3.  Returns the number of pieces this wave issued.
  // 8 pieces per wave: k = 0..3 operand A, k = 4..7 operand B; piece p = wave + 4 (k & 3) of the operand's block
  const char* pg[8];
  uint32_t pm[8];
  uint64_t pl[8];
  uint32_t pv[8];        // per-lane source offsets (tiled operands: rotated per block)
  auto issue_setup = [&](int idx_raw) {
    // past the end of the item the eight pieces are still issued, with no lane taking part: one code path, one vmcnt pattern.
    // Everything here is wave-uniform integer arithmetic on kernel arguments and loop counters (SALU): lane masks included.
    const bool live = idx_raw < total;
    const int idx = live ? idx_raw : 0;
    const int pair = (idx >= nst) ? 1 : 0;
    const int s = s0 + (pair ? idx - nst : idx);
    const uint32_t stage = raw_lds + (idx & 1) * RAW_STAGE;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int op = k >> 2, p = wave + 4 * (k & 3);
      const int bytes = 64 * (op ? J.ldb[pair] : J.lda[pair]);
      const char* g = reinterpret_cast<const char*>(op ? J.b[pair] : J.a[pair]) + (long long)s * bytes;
      const int left = bytes - p * 1024;                  // bytes of this piece that exist (<= 0: none)
      const int nl = left <= 0 ? 0 : (left >= 1024 ? 64 : (left + 15) >> 4);          // lanes that take part (16 B each)
      const uint64_t mk = !live ? 0ull : (nl >= 64 ? ~0ull : ((1ull << nl) - 1ull));
      pl[k] = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(mk >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)mk);
      pg[k] = uni(g + (left > 0 ? p * 1024 : 0));
      pm[k] = __builtin_amdgcn_readfirstlane(stage + op * 16384 + p * 1024);
      pv[k] = (op ? J.tb[pair] : J.ta[pair]) ? tiled_lane16(lane, p) : lane16;
    }
  };
  auto piece = [&](int k) { dma_piece(pg[k], pm[k], pv[k], pl[k]); };
  auto issue = [&](int idx) {     // all eight at once (prologue)
    issue_setup(idx);
#pragma unroll
    for (int k = 0; k < 8; ++k) piece(k);
  };
  // (zeroing of the lanes past the channel count and the column sums are wave-uniform branches: most jobs need neither)
  const bool all_ok = __builtin_amdgcn_ballot_w64(ok0 && (ok1 || !wide)) == ~0ull;
  const bool want_sum = (J.colsum & (is_b ? 2 : 1)) != 0;
  // raw stage idx & 1 -> bf16 hi / lo fragments in frag stage idx & 1; column sums of the first pair's operands on the way
  auto convert = [&](int idx) {
    if (!writes) return;
    const int pair = (idx >= nst) ? 1 : 0;
    const int ld = is_b ? J.ldb[pair] : J.lda[pair];
    const char* raw = smem + LDS_RAW + (idx & 1) * RAW_STAGE + (is_b ? 16384 : 0) + (ok0 ? c0 : 0) * 4;
    char* base = smem + (idx & 1) * FRAG_STAGE + frag0;
    const bool first = want_sum && idx < nst;
    const bool tiled = (is_b ? J.tb[pair] : J.ta[pair]) != 0;       // (256-channel operands only: wide is true)
    const int tb_ = c0 >> 4;
    const char* rawt = smem + LDS_RAW + (idx & 1) * RAW_STAGE + (is_b ? 16384 : 0) + tb_ * 1024 + (c0 & 15) * 4;
    float x[KSTEP], y[KSTEP];
#pragma unroll
    for (int i = 0; i < KSTEP; ++i) {
      if (wide) {
        const f32x2 t = tiled ? *reinterpret_cast<const f32x2*>(rawt + (((i + tb_) & 15) << 6))
                              : *reinterpret_cast<const f32x2*>(raw + i * ld * 4);
        x[i] = t[0];
        y[i] = t[1];
      } else {
        x[i] = *reinterpret_cast<const float*>(raw + i * ld * 4);
        y[i] = 0.0f;
      }
    }
    if (!all_ok) {
#pragma unroll
      for (int i = 0; i < KSTEP; ++i) { x[i] = ok0 ? x[i] : 0.0f; y[i] = ok1 ? y[i] : 0.0f; }
    }
    bf16x8 h0, l0, h1, l1;
    split8(x, h0, l0);
    split8(x + 8, h1, l1);
    *reinterpret_cast<bf16x8*>(base) = h0;
    *reinterpret_cast<bf16x8*>(base + 512) = h1;
    *reinterpret_cast<bf16x8*>(base + 8192) = l0;
    *reinterpret_cast<bf16x8*>(base + 8192 + 512) = l1;
    if (wide) {
      split8(y, h0, l0);
      split8(y + 8, h1, l1);
      *reinterpret_cast<bf16x8*>(base + 16) = h0;
      *reinterpret_cast<bf16x8*>(base + 16 + 512) = h1;
      *reinterpret_cast<bf16x8*>(base + 16 + 8192) = l0;
      *reinterpret_cast<bf16x8*>(base + 16 + 8192 + 512) = l1;
    }
    if (first) {
#pragma unroll
      for (int i = 0; i < KSTEP; ++i) { sum0 += x[i]; sum1 += y[i]; }
    }
  };
  auto compute = [&](int idx) {
    const char* base = smem + (idx & 1) * FRAG_STAGE + lane * 16;
    bf16x8 ah[2], al[2];
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      ah[mf] = *reinterpret_cast<const bf16x8*>(base + (2 * wave + mf) * 1024);
      al[mf] = *reinterpret_cast<const bf16x8*>(base + 8192 + (2 * wave + mf) * 1024);
    }
    // the B fragments of column block nf + 1 are requested before the six MFMAs of block nf go out: an LDS round trip is longer
    // than the issue time of the MFMAs that precede it (with the reads behind the MFMAs the bare K loop ran at 55 % of issue rate)
    bf16x8 bh = *reinterpret_cast<const bf16x8*>(base + 16384), bl = *reinterpret_cast<const bf16x8*>(base + 24576);
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      bf16x8 nh = bh, nl = bl;
      if (nf + 1 < NF) {
        nh = *reinterpret_cast<const bf16x8*>(base + 16384 + (nf + 1) * 1024);
        nl = *reinterpret_cast<const bf16x8*>(base + 24576 + (nf + 1) * 1024);
      }
#pragma unroll
      for (int mf = 0; mf < 2; ++mf) {
        acc[mf][nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mf], bh, acc[mf][nf], 0, 0, 0);
        acc[mf][nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mf], bh, acc[mf][nf], 0, 0, 0);
        acc[mf][nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mf], bl, acc[mf][nf], 0, 0, 0);
      }
      bh = nh;
      bl = nl;
      // one DMA piece of step idx + 2 behind each column block's MFMAs (NF = 8; fewer blocks: the rest goes out after the loop)
      piece(nf);
    }
#pragma unroll
    for (int k = NF; k < 8; ++k) piece(k);
    // pin that order (hipcc otherwise sinks the next block's reads behind the current MFMAs to save eight registers):
    // 6 fragment reads, then per column block [2 reads of the next block, 6 MFMAs]
    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      if (nf + 1 < NF) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
    }
  };

  // Pipeline: the raw rows of steps idx + 1 and idx + 2 are on their way from HBM into LDS (no registers involved, so they stay in
  // flight across the MFMAs, the conversion and the barriers).  An iteration multiplies step idx out of its fragment stage while
  // requesting step idx + 2, makes sure step idx + 1 (requested an iteration and a half ago) has landed, and converts it into the
  // other fragment stage.
  if (total > 0) {
    issue(0);
    issue(1);
    wait_pieces_and_barrier(true);
    convert(0);
    lds_barrier();
#ifdef NRH_DW_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    for (int idx = 0; idx < total; ++idx) {
      const bool n1 = idx + 1 < total;
      // the 48 MFMAs of step idx, with the eight DMA pieces of step idx + 2 between them (raw stage idx & 1 is free: step idx
      // was converted out of it in the previous iteration)
      issue_setup(idx + 2);
      DW_STAMP(0);
      if (!(NRH_DW_ABL & 2)) compute(idx);
      DW_STAMP(2);
      if (n1) {
        // step idx + 1 has landed (this wave's 8 OLDEST pieces; the 8 just issued stay in flight), for all four waves
        wait_pieces_and_barrier(true);
        DW_STAMP(1);
        if (!(NRH_DW_ABL & 4)) convert(idx + 1);
        DW_STAMP(3);
      }
      lds_barrier();
      DW_STAMP(4);
    }
#ifdef NRH_DW_TIMING
    if (lane == 0 && wave == 0) {
      for (int k = 0; k < 5; ++k) atomicAdd(&g_dw_cycles[k], tacc[k]);
      atomicAdd(&g_dw_cycles[5], (unsigned long long)total);
    }
#endif
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // nothing of this item is in flight into LDS any more

  // partial product: row-major [256][256] slot (columns >= 32 NF are not written and not read back)
  const int col = lane & 31, hf = lane >> 5;
#pragma unroll
  for (int mf = 0; mf < 2; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        __builtin_nontemporal_store(acc[mf][nf][r], part + (size_t)(64 * wave + 32 * mf + frow(r, hf)) * 256 + 32 * nf + col);
  // column sums: [A channels 256 | B channels 256]
  if (writes) {
    float* cs = csum + (is_b ? 256 : 0);
    cs[c0] = sum0;
    if (wide) cs[c0 + 1] = sum1;
  }
}

// ---- the fast path: full 256 x 256 products (every lane converts two channels, nothing to mask) ---------------------------------
// Same data flow as dw_item, re-arranged so that ONE straight-line block per K step holds the 48 MFMAs of step idx, the eight DMA
// pieces of step idx + 3 and the whole conversion of step idx + 1: a SIMD hides about five single-issue instructions in every
// 32-cycle MFMA gap, which is what the conversion needs.  For that the raw rows of step idx + 1 must be in LDS BEFORE the block
// starts, so there are three raw stages (steps idx + 2 and idx + 3 in flight) and one barrier per step:
//     [vmcnt(8), barrier]   step idx + 1 landed (everyone's pieces); fragments of step idx written (previous block)
//     block                 MFMAs(idx) out of frag[idx & 1]  |  DMA(idx + 3) -> raw[idx % 3]  |  raw[(idx + 1) % 3] -> frag[(idx + 1) & 1]
constexpr int FAST_RAW_STAGES = 3;
constexpr int FAST_LDS_BYTES = 2 * FRAG_STAGE + FAST_RAW_STAGES * RAW_STAGE;      // 160 KiB
static_assert(FAST_LDS_BYTES <= 163840, "LDS per workgroup");

__device__ __forceinline__ void dma_piece_fast(const char* gp, uint32_t m0v, uint32_t lane16, uint64_t lanes) {
  uint64_t save;    // (no memory clobber: the block's own LDS traffic may be scheduled around it; volatile keeps it between the barriers)
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %4 nt\n\ts_mov_b64 exec, %0"
               : "=&s"(save) : "s"(lanes), "s"(m0v), "v"(lane16), "s"(gp));
}

__device__ __forceinline__ void dw_item_fast(const JobDev& J, const int slab, const int nsteps_all, float* __restrict__ part,
                                             float* __restrict__ csum, char* smem) {
  constexpr int NF = 8;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lane16 = lane * 16;
  const int s0 = (int)((long long)nsteps_all * slab / J.slabs), s1 = (int)((long long)nsteps_all * (slab + 1) / J.slabs);
  const int nst = s1 - s0;
  const int total = nst * J.npairs;
  // converter role: waves 0, 1 the A operand, waves 2, 3 the B operand; channels c0, c0 + 1
  const int is_b = wave >> 1;
  const int c0 = 2 * (tid & 127);
  const int frag0 = is_b * 16384 + (c0 >> 5) * 1024 + (c0 & 31) * 16;
  const int raw0 = LDS_RAW + is_b * 16384 + c0 * 4;
  const int cb = c0 >> 4;                                           // the thread's channel block
  const int raw0t = LDS_RAW + is_b * 16384 + cb * 1024 + (c0 & 15) * 4;   // ... and its column in a TILED operand's block
  const uint32_t raw_lds = lds_off(smem) + LDS_RAW;
  const float sum_flag = (J.colsum & (is_b ? 2 : 1)) ? 1.0f : 0.0f;

  f32x16 acc[2][NF];
#pragma unroll
  for (int mf = 0; mf < 2; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mf][nf][r] = 0.0f;
  float sum0 = 0.0f, sum1 = 0.0f;

  // DMA descriptors of the step to be requested next, advanced incrementally (a few SALU instructions per step): this wave's 4
  // pieces of the A block and 4 of the B block (both 16 KiB: piece p = wave + 4 k)
  int fidx = 0;                      // step the descriptors point at
  int fstage = 0;                    // its raw stage (fidx % 3)
  const char* ga = nullptr;
  const char* gb = nullptr;
  // per-lane source offsets of this wave's pieces (blocks wave, wave + 4, wave + 8, wave + 12): rotated for a tiled operand.  The
  // layout of an operand may differ between the two pairs of a job (it does not in the training step's table), hence per pair.
  auto lane_off = [&](int tiled, int kk) { return tiled ? tiled_lane16(lane, wave + 4 * kk) : lane16; };
  uint32_t va[4], vb[4];
  auto set_offsets = [&](int pair) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { va[kk] = lane_off(J.ta[pair], kk); vb[kk] = lane_off(J.tb[pair], kk); }
  };
  set_offsets(0);
  const bool same_layout = J.npairs == 1 || (J.ta[0] == J.ta[1] && J.tb[0] == J.tb[1]);
  auto point_at = [&](int idx) {
    const int pair = (idx >= nst) ? 1 : 0;
    const int s = s0 + (pair ? idx - nst : idx);
    ga = reinterpret_cast<const char*>(J.a[pair]) + (long long)s * 16384 + wave * 1024;
    gb = reinterpret_cast<const char*>(J.b[pair]) + (long long)s * 16384 + wave * 1024;
  };
  point_at(0);
  auto piece = [&](int k) {          // k = 0..3 A, 4..7 B
    const uint32_t m = __builtin_amdgcn_readfirstlane(fidx < total ? 0xffffffffu : 0u);     // (wave-uniform, said so to hipcc)
    const uint64_t lanes = ((uint64_t)m << 32) | m;
    const uint32_t m0v = __builtin_amdgcn_readfirstlane(raw_lds + fstage * RAW_STAGE + (k >> 2) * 16384 + (wave + 4 * (k & 3)) * 1024);
    dma_piece_fast(uni(((k >> 2) ? gb : ga) + (k & 3) * 4096), m0v, (k >> 2) ? vb[k & 3] : va[k & 3], lanes);
  };
  // where the second pair starts (branch-free advance: the block below must stay one basic block)
  const char* const ga1 = reinterpret_cast<const char*>(J.a[J.npairs - 1]) + (long long)s0 * 16384 + wave * 1024;
  const char* const gb1 = reinterpret_cast<const char*>(J.b[J.npairs - 1]) + (long long)s0 * 16384 + wave * 1024;
  auto advance = [&]() {
    ++fidx;
    fstage = (fstage == FAST_RAW_STAGES - 1) ? 0 : fstage + 1;
    const bool second = fidx == nst;
    ga = second ? ga1 : ga + 16384;
    gb = second ? gb1 : gb + 16384;
    if (second && !same_layout) set_offsets(1);       // (wave-uniform, rare: a job whose pairs differ in layout)
  };
  auto issue_all = [&]() {
#pragma unroll
    for (int k = 0; k < 8; ++k) piece(k);
    advance();
  };

  // Conversion of a raw stage into a fragment stage (this thread's two channels, 16 points), cut into eight parts that ride in
  // the eight column-block groups of the MFMA block: part 2g converts group g (8 values: channel c0 points 0..7, c0 points 8..15,
  // c0 + 1 points 0..7, c0 + 1 points 8..15) to its hi halves and residuals, part 2g + 1 rounds the residuals, writes both 16-byte
  // words and adds the group to the column sum.  (Every DMA piece is an asm statement and therefore a scheduling boundary for
  // hipcc: what shall run beside the six MFMAs of a group has to stand between the same two pieces in program order.)
  float xv[4][8];                    // the 32 raw values, group-major
  float rs_[8];                      // residuals of the group in flight
  bf16x8 hi_;
  // raw_read(stage, tiled): the thread's two channels of the 16 points.  Row-major: point i at i KiB + 4 c0; tiled: block cb, LDS row
  // (i + cb) & 15 (64 bytes each).  One form for both: offset_i = ((i * stride + rot) & mask), base chosen per layout - a few SALU /
  // VALU operations per step instead of a second code path in the scheduled block.
  auto raw_read = [&](int rs, int tiled) {
    const char* raw = smem + (tiled ? raw0t : raw0) + rs * RAW_STAGE;
    const int sh = tiled ? 6 : 10, rot = tiled ? cb * 64 : 0, mask = tiled ? 1023 : 16383;
#pragma unroll
    for (int i = 0; i < KSTEP; ++i) {
      const f32x2 t = *reinterpret_cast<const f32x2*>(raw + (((i << sh) + rot) & mask));
      xv[i >> 3][i & 7] = t[0];
      xv[2 + (i >> 3)][i & 7] = t[1];
    }
  };
  // layout of this thread's operand in the step that is converted next (the pair changes at idx = nst)
  auto tiled_at = [&](int idx) { const int pair = (idx >= nst) ? 1 : 0; return is_b ? J.tb[pair] : J.ta[pair]; };
  auto convert_part = [&](int part, int fs, float flag) {
    const int g = part >> 1;
    // home of group g: channel (g >> 1), k half (g & 1)
    char* base = smem + fs * FRAG_STAGE + frag0 + (g >> 1) * 16 + (g & 1) * 512;
    if ((part & 1) == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const __bf16 h = (__bf16)xv[g][i];
        hi_[i] = h;
        rs_[i] = xv[g][i] - (float)h;
      }
    } else {
      bf16x8 lo;
#pragma unroll
      for (int i = 0; i < 8; ++i) lo[i] = (__bf16)rs_[i];
      *reinterpret_cast<bf16x8*>(base) = hi_;
      *reinterpret_cast<bf16x8*>(base + 8192) = lo;
      float a = 0.0f;
#pragma unroll
      for (int i = 0; i < 8; ++i) a += xv[g][i];
      // (a select, not a multiplication by 0: past the end of the item the rows are stale LDS content, possibly NaN)
      a = flag != 0.0f ? a : 0.0f;
      if (g >> 1) sum1 += a;
      else sum0 += a;
    }
  };

  if (total > 0) {
    issue_all();                       // step 0 -> raw 0
    issue_all();                       // step 1 -> raw 1   (empty pieces past the end of the item)
    issue_all();                       // step 2 -> raw 2
    asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // step 0 landed
    raw_read(0, tiled_at(0));
#pragma unroll
    for (int part = 0; part < 8; ++part) convert_part(part, 0, sum_flag);
    int rs_next = 1;                   // raw stage of step idx + 1
#ifdef NRH_DW_TIMING
    unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    for (int idx = 0; idx < total; ++idx) {
      // step idx + 1 landed (this wave's 8 oldest pieces; the 8 of step idx + 2 stay in flight), fragments of step idx written
      DW_STAMP(0);
      asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      DW_STAMP(1);
      const char* base = smem + (idx & 1) * FRAG_STAGE + lane * 16;
      const int fs = (idx + 1) & 1;
      // (past the end of the item this converts stale rows into a stage nobody reads, with flag 0)
      const float flag = (idx + 1 < nst) ? sum_flag : 0.0f;
      raw_read(rs_next, tiled_at(idx + 1));
      bf16x8 ah[2], al[2];
#pragma unroll
      for (int mf = 0; mf < 2; ++mf) {
        ah[mf] = *reinterpret_cast<const bf16x8*>(base + (2 * wave + mf) * 1024);
        al[mf] = *reinterpret_cast<const bf16x8*>(base + 8192 + (2 * wave + mf) * 1024);
      }
      bf16x8 bh = *reinterpret_cast<const bf16x8*>(base + 16384), bl = *reinterpret_cast<const bf16x8*>(base + 24576);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        bf16x8 nh = bh, nl = bl;
        if (nf + 1 < NF) {
          nh = *reinterpret_cast<const bf16x8*>(base + 16384 + (nf + 1) * 1024);
          nl = *reinterpret_cast<const bf16x8*>(base + 24576 + (nf + 1) * 1024);
        }
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) {
          acc[mf][nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mf], bh, acc[mf][nf], 0, 0, 0);
          acc[mf][nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mf], bh, acc[mf][nf], 0, 0, 0);
          acc[mf][nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mf], bl, acc[mf][nf], 0, 0, 0);
        }
        bh = nh;
        bl = nl;
        convert_part(nf, fs, flag);
        // this group's shape for the scheduler: the next block's two fragment reads, then MFMA / VALU alternating
        if (nf + 1 < NF) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
        }
        piece(nf);                     // step idx + 3 -> raw stage idx % 3 (its previous content, step idx, was converted a block ago)
      }
      advance();
      rs_next = (rs_next == FAST_RAW_STAGES - 1) ? 0 : rs_next + 1;
    }
#ifdef NRH_DW_TIMING
    if (lane == 0 && wave == 0) {
      atomicAdd(&g_dw_cycles[6], tacc[0]);            // fast path: block
      atomicAdd(&g_dw_cycles[7], tacc[1]);            // fast path: wait + barrier
      atomicAdd(&g_dw_cycles[5], (unsigned long long)total << 32);    // fast-path steps in the high half
    }
#endif
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // nothing of this item is in flight into LDS any more

  const int col = lane & 31, hf = lane >> 5;
#pragma unroll
  for (int mf = 0; mf < 2; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        __builtin_nontemporal_store(acc[mf][nf][r], part + (size_t)(64 * wave + 32 * mf + frow(r, hf)) * 256 + 32 * nf + col);
  float* cs = csum + is_b * 256;
  cs[c0] = sum0;
  cs[c0 + 1] = sum1;
}

// ---- 16-bit operands (round 5): full 256 x 256 products whose four arrays are fp16 in the HALF-TILED layout -----------------------
// The sweep kernels of the f16x3 training step can write h, t, abar, zbar - arrays nothing but this kernel reads - as fp16
// (csrc/nrh_mlp.h, half_ptr: [tile of 16 points][block pair 8][point 16][quarter 4][block of the pair 2][4 channels], 8 KiB per
// tile, one contiguous KiB per wave instruction), the adjoints multiplied by the step's power-of-two adjoint scale so that they sit in
// fp16's range.  Then a product is ONE v_mfma_f32_32x32x16_f16 per K step instead of three bf16 ones, the operand bytes halve, and
// nothing is converted: the LDS-DMA image of a KiB piece IS a [16 points][32 channels] block (64 bytes per point), and
// ds_read_b64_tr_b16 - gfx950's transpose read: a 16-lane group fetches a 4 points x 16 channels matrix, lane c receives the four
// points of channel c - turns it into MFMA fragments (K = points contiguous per lane) directly.  Lane l (group G = l >> 4, i = l & 15)
// supplies point 8 (G >> 1) + (i >> 2), quarter i & 3, block G & 1 of the pair; the 32 lanes of a half-wave cover 256 contiguous
// bytes: no bank conflict.  Accuracy: operands carry 11 bits, products are exact in the fp32 accumulator; priced against the
// reference's float64 gradients before it was built (profiles/dw16_emulation.py -> profiles/r05/dw16_emulation.log: every tensor
// inside the 1 024-ray test's bounds at the three anneal steps; bf16 operands are not - 4-5x outside on the reflectance net).
// Pipeline: a stage = 32 points of both operands (32 KiB), five stages, four in flight (128 KiB per CU), one barrier per stage.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short i16x4 __attribute__((ext_vector_type(4)));
constexpr int H16_STAGE = 32768;         // A: 2 tiles x 8 KiB | B: 2 tiles x 8 KiB
constexpr int H16_STAGES = 5;
constexpr int H16_DIST = 4;
static_assert(H16_STAGES * H16_STAGE <= FAST_LDS_BYTES, "LDS per workgroup");

__device__ __forceinline__ f16x8 tr_frag(const char* lds) {
  typedef __attribute__((address_space(3))) i16x4 lds_v;
  const lds_v* p = (const lds_v*)(const __attribute__((address_space(3))) char*)lds;
  const i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(const_cast<lds_v*>(p));            // points +0..3 of the lane's k group
  const i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(const_cast<lds_v*>(p + 32));       // points +4..7 (256 bytes further)
  const f16x4 l4 = __builtin_bit_cast(f16x4, lo), h4 = __builtin_bit_cast(f16x4, hi);
  return f16x8{l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]};
}

__device__ __forceinline__ void dw_item_half(const JobDev& J, const int slab, const int nsteps_all, float* __restrict__ part,
                                             float* __restrict__ csum, char* smem) {
  constexpr int NF = 8;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lane16 = lane * 16;
  const int n32 = nsteps_all >> 1;                      // stages of 32 points
  const int s0 = (int)((long long)n32 * slab / J.slabs), s1 = (int)((long long)n32 * (slab + 1) / J.slabs);
  const int nst = s1 - s0;
  const int total = nst * J.npairs;
  const uint32_t lds0 = lds_off(smem);
  const bool want_sum = (J.colsum & 1) != 0;

  f32x16 acc[2][NF];
#pragma unroll
  for (int mf = 0; mf < 2; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mf][nf][r] = 0.0f;
  float sum[2] = {0.0f, 0.0f};

  // this wave's 4 + 4 pieces of a stage: piece p = wave + 4 k of the operand's 16 KiB (a stage is contiguous in both operands)
  int fidx = 0, fstage = 0;
  auto issue = [&]() {
    const bool live = fidx < total;
    const int pair = (fidx >= nst) ? 1 : 0;
    const long long s = s0 + (pair ? fidx - nst : fidx);
    const char* ga = reinterpret_cast<const char*>(J.a[pair]) + s * 16384 + wave * 1024;
    const char* gb = reinterpret_cast<const char*>(J.b[pair]) + s * 16384 + wave * 1024;
    const uint32_t m = __builtin_amdgcn_readfirstlane(live ? 0xffffffffu : 0u);
    const uint64_t lanes = ((uint64_t)m << 32) | m;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds0 + fstage * H16_STAGE + (k >> 2) * 16384 + (wave + 4 * (k & 3)) * 1024);
      dma_piece_fast(uni(((k >> 2) ? gb : ga) + (k & 3) * 4096), m0v, lane16, lanes);
    }
    ++fidx;
    fstage = (fstage == H16_STAGES - 1) ? 0 : fstage + 1;
  };
  // the lane's corner of a [16 points][32 channels] piece for the transpose read
  const int lb = (((lane >> 5) * 8 + ((lane & 15) >> 2)) * 64) + (lane & 3) * 16 + ((lane >> 4) & 1) * 8;

  if (total > 0) {
#pragma unroll
    for (int d = 0; d < H16_DIST; ++d) issue();
    int cstage = 0;
    for (int idx = 0; idx < total; ++idx) {
      // stage idx landed (this wave's 8 oldest pieces; everybody's after the barrier); everybody is done reading stage idx - 1
      asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      issue();                            // stage idx + 4 -> the buffer stage idx - 1 was read from
      const char* st = smem + cstage * H16_STAGE + lb;
      const bool sum_now = want_sum && idx < nst;
      // The stage's two tiles of 16 points = 16 B fragments, each feeding two MFMAs.  Fragment reads run TWO fragments (128 MFMA
      // cycles ~ one LDS round trip) ahead of the MFMAs that consume them - a SIMD has one wave, an LDS wait that is not under
      // MFMAs is lost time - and the A fragments of the second tile are fetched under the first tile's last MFMAs.  hipcc sinks
      // loads to their uses; sched_group_barrier pins the order read, read, MFMA, MFMA per fragment.
      f16x8 af[2], an[2], b0 = tr_frag(st + 16384), b1 = tr_frag(st + 16384 + 1024);
#pragma unroll
      for (int mf = 0; mf < 2; ++mf) an[mf] = af[mf] = tr_frag(st + (2 * wave + mf) * 1024);
      if (sum_now) {
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          // (the second tile's A fragments are read once more for the column sums: two more of 42 reads, off the MFMA path)
          const f16x8 a0 = ss ? tr_frag(st + (8 + 2 * wave) * 1024) : af[0], a1 = ss ? tr_frag(st + (8 + 2 * wave + 1) * 1024) : af[1];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            sum[0] = __builtin_amdgcn_fdot2(f16x2{a0[2 * i], a0[2 * i + 1]}, f16x2{(_Float16)1.0f, (_Float16)1.0f}, sum[0], false);
            sum[1] = __builtin_amdgcn_fdot2(f16x2{a1[2 * i], a1[2 * i + 1]}, f16x2{(_Float16)1.0f, (_Float16)1.0f}, sum[1], false);
          }
        }
      }
#pragma unroll
      for (int f = 0; f < 16; ++f) {           // fragment f = tile f >> 3, column block f & 7
        f16x8 b2 = b1;
        if (f + 2 < 16) b2 = tr_frag(st + 16384 + (f + 2) * 1024);
        if (f == 5) an[0] = tr_frag(st + (8 + 2 * wave) * 1024);
        if (f == 6) an[1] = tr_frag(st + (8 + 2 * wave + 1) * 1024);
        if (f == 8) { af[0] = an[0]; af[1] = an[1]; }
#pragma unroll
        for (int mf = 0; mf < 2; ++mf) acc[mf][f & 7] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[mf], b0, acc[mf][f & 7], 0, 0, 0);
        b0 = b1;
        b1 = b2;
        if (f == 5 || f == 6) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        else if (f + 2 < 16) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
      }
      cstage = (cstage == H16_STAGES - 1) ? 0 : cstage + 1;
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // nothing of this item is in flight into LDS any more

  const int col = lane & 31, hf = lane >> 5;
#pragma unroll
  for (int mf = 0; mf < 2; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        __builtin_nontemporal_store(acc[mf][nf][r], part + (size_t)(64 * wave + 32 * mf + frow(r, hf)) * 256 + 32 * nf + col);
  // column sums of A_0: lane l and l ^ 32 hold the two K halves of channel 64 wave + 32 mf + (l & 31)
#pragma unroll
  for (int mf = 0; mf < 2; ++mf) {
    const float other = __shfl_xor(sum[mf], 32);
    if (hf == 0) csum[64 * wave + 32 * mf + col] = sum[mf] + other;
  }
  if (tid < 256) csum[256 + tid] = 0.0f;
}

// ---- the thin path: products with at most four columns (the SDF head: h_7^T sbar + abar_7^T 1; the reflectance net's output layer
// transposed: save_h_3^T zbar4) are matrix-VECTOR work - 256 x n multiply-adds per point against 1 KiB of A to read - so they run as
// plain float32 FMAs straight from global memory, thread t = channel t of A, no LDS, no conversion (exact fp32 products).
__device__ __forceinline__ void dw_item_thin(const JobDev& J, const int slab, const int nsteps_all, float* __restrict__ part,
                                             float* __restrict__ csum) {
  const int tid = threadIdx.x;
  const int s0 = (int)((long long)nsteps_all * slab / J.slabs), s1 = (int)((long long)nsteps_all * (slab + 1) / J.slabs);
  const int c = tid < J.m ? tid : J.m - 1;
  float acc[4] = {0.f, 0.f, 0.f, 0.f}, sa = 0.0f, sb[4] = {0.f, 0.f, 0.f, 0.f};
  for (int pair = 0; pair < J.npairs; ++pair) {
    const int lda = J.lda[pair], ldb = J.ldb[pair], n = J.n;
    // channel c of point i of the step: row-major i * lda + c; tiled (256 channels): block c >> 4, row i, column c & 15
    const int astride = J.ta[pair] ? 16 : lda;
    const float* A = J.a[pair] + (size_t)s0 * KSTEP * lda + (J.ta[pair] ? (c >> 4) * 256 + (c & 15) : c);
    const float* B = J.b[pair] + (size_t)s0 * KSTEP * ldb;
    for (int s = s0; s < s1; ++s) {
      float a[KSTEP];
#pragma unroll
      for (int i = 0; i < KSTEP; ++i) a[i] = __builtin_nontemporal_load(A + (size_t)i * astride);
#pragma unroll
      for (int i = 0; i < KSTEP; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float b = j < n ? B[i * ldb + j] : 0.0f;          // wave-uniform address: scalar loads
          acc[j] = __builtin_fmaf(a[i], b, acc[j]);
          if (pair == 0 && tid == 0) sb[j] += b;
        }
        if (pair == 0) sa += a[i];
      }
      A += (size_t)KSTEP * lda;
      B += (size_t)KSTEP * ldb;
    }
  }
  if (tid < J.m) {
#pragma unroll
    for (int j = 0; j < 4; ++j) part[(size_t)tid * 256 + j] = acc[j];
    csum[tid] = sa;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) part[(size_t)tid * 256 + j] = 0.0f;
    csum[tid] = 0.0f;
  }
  if (tid == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) csum[256 + j] = sb[j];
  }
}

__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void dw_kernel(const DwArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int item = blockIdx.x;
  int j = 0;
#pragma unroll 1
  for (int k = 1; k < a.njobs; ++k)
    if (item >= a.job[k].slab0) j = k;
  const JobDev& J = a.job[j];
  const int slab = item - J.slab0;
  float* part = a.partial + (size_t)item * SLOT_FLOATS;
  float* cs = a.csum + (size_t)item * CSUM_FLOATS;
  const bool full = J.m == 256 && J.n == 256 && J.lda[0] == 256 && J.ldb[0] == 256 && (J.npairs == 1 || (J.lda[1] == 256 && J.ldb[1] == 256));
  if (J.half) dw_item_half(J, slab, a.nsteps, part, cs, smem);
  else if (full) dw_item_fast(J, slab, a.nsteps, part, cs, smem);
  else if (J.n <= 4) dw_item_thin(J, slab, a.nsteps, part, cs);
  else if (J.n <= 32) dw_item<1>(J, slab, a.nsteps, part, cs, smem);
  else if (J.n <= 64) dw_item<2>(J, slab, a.nsteps, part, cs, smem);
  else if (J.n <= 128) dw_item<4>(J, slab, a.nsteps, part, cs, smem);
  else dw_item<8>(J, slab, a.nsteps, part, cs, smem);
}

// ---- slabs -> gradient tensors -------------------------------------------------------------------------------------------
struct OutDev {
  float* out;            // destination matrix (or null)
  const int* col_map;    // optional [n]: destination column of product column j
  float* colsum_a;       // optional [m]: scale_a * sum over points of A_0[:, i]
  float* colsum_b;       // optional [n]
  int ldo, transpose;    // out[i * ldo + j], or out[j * ldo + i] when transposed
  int rows, cols;        // only i < rows, j < cols are written
  float scale, scale_a, scale_b;
  int slab0, slabs;
  const float* dyn;      // optional device pointer {S, 1 / S}: the product and the column sums of A are multiplied by dyn[1]
};
struct ReduceArgs {
  OutDev job[MAX_JOBS];
  int njobs;
  const float* partial;
  const float* csum;
};

__global__ __launch_bounds__(256) void dw_reduce_kernel(const ReduceArgs a) {
  const OutDev& J = a.job[blockIdx.y];
  const int e = blockIdx.x * 256 + threadIdx.x;     // element of the [256][256] product
  const int i = e >> 8, j = e & 255;
  const float dyn = J.dyn ? J.dyn[1] : 1.0f;
  if (J.out && i < J.rows && j < J.cols) {
    // fixed summation order; the loads of eight slabs go out together (the adds alone serialised them: 96 us for 67 MB)
    float s = 0.0f;
    int k = 0;
    for (; k + 8 <= J.slabs; k += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(a.partial + (size_t)(J.slab0 + k + u) * SLOT_FLOATS + e);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < J.slabs; ++k) s += a.partial[(size_t)(J.slab0 + k) * SLOT_FLOATS + e];
    const int jj = J.col_map ? J.col_map[j] : j;
    J.out[J.transpose ? (size_t)jj * J.ldo + i : (size_t)i * J.ldo + jj] = s * J.scale * dyn;
  }
  if (blockIdx.x == 0) {
    const int c = threadIdx.x;
    if (J.colsum_a && c < J.rows) {
      float s = 0.0f;
      for (int k = 0; k < J.slabs; ++k) s += a.csum[(size_t)(J.slab0 + k) * CSUM_FLOATS + c];
      J.colsum_a[c] = s * J.scale_a * dyn;
    }
    if (J.colsum_b && c < J.cols) {
      float s = 0.0f;
      for (int k = 0; k < J.slabs; ++k) s += a.csum[(size_t)(J.slab0 + k) * CSUM_FLOATS + 256 + c];
      J.colsum_b[c] = s * J.scale_b;
    }
  }
}

}  // namespace nrhdw
