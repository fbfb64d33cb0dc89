// Fused per-point MLP machinery for gfx950: the "transposed register chain".
//
// A wavefront owns a tile of 16 points.  Every layer is computed as  H_out^T = W * H_in^T  with
// v_mfma_f32_16x16x4_f32 (exact fp32, bit-identical to an fmaf chain): the WEIGHTS are the A operand
// (16 output features x 4 inputs per instruction) and the ACTIVATIONS the B operand (4 inputs x 16
// points).  The C/D fragment of that instruction (lane l, register r  <->  feature 4*(l>>4)+r of the
// 16-row block, point l&15) is exactly the B fragment the next layer needs when its K index is
// enumerated as k = 16*kb + 4*(l>>4) + c.  So activations never leave registers between layers: no
// LDS round trip, no cross-lane shuffle, no transposition.  "D-layout" below means that mapping:
//
//        value[b*4 + r] on lane (j = l & 15, q = l >> 4)   <->   feature 16*b + 4*q + r of point j
//
// Weights are streamed once per workgroup (8 waves = 128 points) through a double-buffered 2 x 32 KiB
// LDS ring by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction), in "chunks" of two
// 16-row output blocks x all K blocks, pre-packed on the host so that the 64 lanes' float4 operands
// of one (output block, K block) pair are one contiguous, conflict-free 1 KiB line.
//
// The same machinery runs every per-point chain of the path: SDF value / gradient / feature (nrh_sdf.hip), the
// reflectance net (nrh_color.hip), and for training the tangent and adjoint sweeps of both networks (nrh_sdf_train.hip,
// nrh_color.hip) - they differ only in the packed stages they stream and in their epilogues.
#pragma once
#include "nrh_common.h"

namespace nrh {

constexpr int WBUF_BYTES = 32768;          // one LDS weight buffer (2 ob x 16 kb x 1 KiB)
constexpr int MLP_LDS_BYTES = 2 * WBUF_BYTES;
// ---- tuning knobs (compile-time; profiles/README.md records what each was measured to do) ----
#ifndef NRH_WG_WAVES
#define NRH_WG_WAVES 8        // waves per workgroup = 16-point tiles sharing one weight stream (8: +6..24 % vs 4)
#endif
#ifndef NRH_KPREFETCH
#define NRH_KPREFETCH 1       // K steps of A operands in flight ahead of the MFMAs (f16x3 path)
#endif
#ifndef NRH_SCHED_BARRIER
#define NRH_SCHED_BARRIER 1   // pin the ds_read / MFMA interleave with sched_barrier per K step
#endif
#ifndef NRH_NT_SCRATCH
#define NRH_NT_SCRATCH 1      // non-temporal loads/stores for the stream-once sigma' scratch and feature tiles
#endif
#ifndef NRH_PKRTZ
#define NRH_PKRTZ 1           // v_cvt_pkrtz_f16_f32 for the hi/lo split (round-toward-zero hi, still exact hi+lo)
#endif
constexpr int WG_WAVES = NRH_WG_WAVES;
constexpr int MLP_THREADS = 64 * WG_WAVES;  // one 16-point tile per wave
constexpr int TILE_PTS = 16;


typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Asynchronously copy `npieces` KiB (global, contiguous) into an LDS buffer; the waves of the workgroup interleave.
__device__ __forceinline__ void dma_chunk(const float* __restrict__ src, char* dst_lds, int npieces, int wave,
                                          int lane) {
  for (int k = wave; k < npieces; k += WG_WAVES) {
    __builtin_amdgcn_global_load_lds((gptr_t)(src + k * 256 + lane * 4), (lptr_t)(dst_lds + k * 1024), 16, 0, 0);
  }
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---- precision modes -------------------------------------------------------------------------------------------
// PREC 0  "f32"    v_mfma_f32_16x16x4_f32: exact fp32 products and accumulation (157 TFLOP/s peak).
// PREC 1  "f16x3"  every fp32 value x is carried as two fp16 numbers, hi = fp16(x) and lo = fp16((x - hi) * 2^11)
//                  (the 2^11 keeps lo out of the fp16 subnormal range); a product x*w is evaluated as
//                  hi_x*hi_w + 2^-11 (hi_x*lo_w + lo_x*hi_w) with three v_mfma_f32_16x16x32_f16 into two fp32
//                  accumulators.  The dropped lo*lo term and the fp16 rounding of lo are both ~2^-22 relative, i.e.
//                  fp32 round-off class (measured on the SDF net: same error against fp64 as the fp32 chain,
//                  CHANGELOG.md section 5), at 16/3 of the fp32 matrix rate.
constexpr float LO_SCALE = 2048.0f;
constexpr float LO_UNSCALE = 1.0f / 2048.0f;

#ifndef NRH_SPLIT_FMA
#define NRH_SPLIT_FMA 1       // residual a - hi as one fma with the fp16 operand read in place (v_fma_mix_f32), no cvt back
#endif
__device__ __forceinline__ void split_pack2(float a, float b, uint32_t& hi, uint32_t& lo) {
#if NRH_PKRTZ
  typedef __fp16 h16x2 __attribute__((ext_vector_type(2)));
  const h16x2 h = __builtin_amdgcn_cvt_pkrtz(a, b);
#if NRH_SPLIT_FMA
  const float ra = __builtin_fmaf((float)h.x, -LO_SCALE, a * LO_SCALE);
  const float rb = __builtin_fmaf((float)h.y, -LO_SCALE, b * LO_SCALE);
  const h16x2 l = __builtin_amdgcn_cvt_pkrtz(ra, rb);
#else
  const h16x2 l = __builtin_amdgcn_cvt_pkrtz((a - (float)h.x) * LO_SCALE, (b - (float)h.y) * LO_SCALE);
#endif
#else
  const f16x2 h = {(_Float16)a, (_Float16)b};
  const f16x2 l = {(_Float16)((a - (float)h.x) * LO_SCALE), (_Float16)((b - (float)h.y) * LO_SCALE)};
#endif
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
}

// stream-once global traffic (sigma' scratch, feature tiles, the training sweeps' saved arrays); loads and stores can be switched
// separately for A/B runs (make variant DEFS=-DNRH_NT_LOAD=0)
#ifndef NRH_NT_LOAD
#define NRH_NT_LOAD NRH_NT_SCRATCH
#endif
#ifndef NRH_NT_STORE
#define NRH_NT_STORE NRH_NT_SCRATCH
#endif
template <typename V>
__device__ __forceinline__ void st_stream(V* p, V v) {
#if NRH_NT_STORE
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
template <typename V>
__device__ __forceinline__ V ld_stream(const V* p) {
#if NRH_NT_LOAD
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}

// Activations of one layer for this wave's 16 points, in the form the MFMA B operand wants them.
//   PREC 0: KB*4 floats in D-layout.   PREC 1: per 32-wide K step, 8 fp16 hi + 8 fp16 lo (4 + 4 packed VGPRs):
//   elements 0..3 = features 16*(2s)+4q+0..3, elements 4..7 = features 16*(2s+1)+4q+0..3.
template <int PREC, int KB>
struct Act;
template <int KB>
struct Act<0, KB> {
  float v[KB * 4];
  // the 8 outputs of chunk ch (blocks 2ch and 2ch+1, registers 0..3 each)
  __device__ __forceinline__ void set_chunk(int ch, const float (&o)[8]) {
#pragma unroll
    for (int r = 0; r < 8; ++r) v[ch * 8 + r] = o[r];
  }
  __device__ __forceinline__ void set_chunk(int ch, const f32x4 o0, const f32x4 o1) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { v[ch * 8 + r] = o0[r]; v[ch * 8 + 4 + r] = o1[r]; }
  }
};
template <int KB>
struct Act<1, KB> {
  static_assert(KB % 2 == 0, "f16x3 K steps are 32 wide");
  uint32_t h[KB * 2], l[KB * 2];
  __device__ __forceinline__ void set_chunk(int ch, const float (&o)[8]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) split_pack2(o[2 * r], o[2 * r + 1], h[ch * 4 + r], l[ch * 4 + r]);
  }
  __device__ __forceinline__ void set_chunk(int ch, const f32x4 o0, const f32x4 o1) {
    split_pack2(o0[0], o0[1], h[ch * 4 + 0], l[ch * 4 + 0]);
    split_pack2(o0[2], o0[3], h[ch * 4 + 1], l[ch * 4 + 1]);
    split_pack2(o1[0], o1[1], h[ch * 4 + 2], l[ch * 4 + 2]);
    split_pack2(o1[2], o1[3], h[ch * 4 + 3], l[ch * 4 + 3]);
  }
};

// One GEMM stage: out[NCH*32 features] = A[NCH*32 x KB*16] * in[KB*16 features], all for this wave's 16 points.
//   wsrc   packed weights of this stage: NCH chunks of 2*KB KiB, chunk ch resident in LDS buffer `par` on entry
//   wnext  first chunk of whatever runs next (prefetched during the last chunk), next_pieces its size in KiB
//   init   optional accumulator start values (D-layout, NCH*8 floats) - used to split one layer's K in two
//   pre    pre(ch) -> P: issues the epilogue's global loads (bias, sigma', head weights) BEFORE the K loop so that
//          their L2 latency hides under the chunk's MFMAs (left inside the epilogue they cost ~700 exposed cycles
//          per chunk: the K loop is fenced by sched_barrier, nothing can be hoisted across it)
//   epi    epi(ch, acc0, acc1, P): consumes the two finished 16-feature blocks 2*ch and 2*ch+1
// LDS image of a chunk (both precisions 2*KB KiB):
//   PREC 0: [obi 2][kb KB][lane 64] float4           - A of 4 consecutive 16x16x4 MFMAs
//   PREC 1: [obi 2][s KB/2][hi|lo][lane 64] 8 x fp16 - A of one 16x16x32 MFMA (hi) / its low part
// PRE_LOADS documents that pre() issues global loads (kept for the call sites' readability; no effect on the code).
// Variants of this loop that were measured and dropped (code in the commits named in profiles/README.md): chunk barrier
// without the vmcnt drain, barrier before the epilogue, LDS-DMA issue inside the K loop / as straight-line code, a
// 4-slot ring with one barrier per two chunks, deeper / pinned A-operand prefetch, the epilogue of chunk c-1 interleaved
// with the MFMAs of chunk c by sched_group_barrier.
template <int PREC, int KB, int NCH, bool HAS_INIT, bool PRE_LOADS = false, typename Pre, typename Epi>
__device__ __forceinline__ void run_stage(const float* __restrict__ wsrc, const float* __restrict__ wnext,
                                          int next_pieces, char* smem, int& par, const Act<PREC, KB>& in,
                                          const float* init, Pre&& pre, Epi&& epi, int wave, int lane) {
  constexpr int PIECES = 2 * KB;
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    // the other buffer receives the stage's next chunk, or the first chunk of whatever runs next, during this chunk
    char* nxt = smem + (par ^ 1) * WBUF_BYTES;
    if (ch + 1 < NCH) {
      dma_chunk(wsrc + (ch + 1) * PIECES * 256, nxt, PIECES, wave, lane);
    } else if (wnext != nullptr) {
      dma_chunk(wnext, nxt, next_pieces, wave, lane);
    }
    asm volatile("" ::: "memory");  // the weight stream goes out first, then the epilogue's loads
    const auto pv = pre(ch);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (HAS_INIT) {
      acc0 = f32x4{init[ch * 8 + 0], init[ch * 8 + 1], init[ch * 8 + 2], init[ch * 8 + 3]};
      acc1 = f32x4{init[ch * 8 + 4], init[ch * 8 + 5], init[ch * 8 + 6], init[ch * 8 + 7]};
    }
    if constexpr (PREC == 0) {
      const f32x4* A = reinterpret_cast<const f32x4*>(smem + par * WBUF_BYTES);
      // software-pipelined by one K block: the A operands of block kb+1 are in flight while block kb's
      // 8 MFMAs (2 independent accumulators, 256 cycles) run; sched_barrier keeps hipcc from hoisting all
      // 2*KB ds_read_b128 to the top of the chunk (128 live VGPRs -> spills at 2 waves/SIMD).
      f32x4 a0 = A[(0 * KB + 0) * 64 + lane];
      f32x4 a1 = A[(1 * KB + 0) * 64 + lane];
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        f32x4 n0 = a0, n1 = a1;
        if (kb + 1 < KB) {
          n0 = A[(0 * KB + kb + 1) * 64 + lane];
          n1 = A[(1 * KB + kb + 1) * 64 + lane];
        }
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, in.v[kb * 4 + 0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, in.v[kb * 4 + 0], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, in.v[kb * 4 + 1], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, in.v[kb * 4 + 1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, in.v[kb * 4 + 2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, in.v[kb * 4 + 2], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, in.v[kb * 4 + 3], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, in.v[kb * 4 + 3], acc1, 0, 0, 0);
        a0 = n0;
        a1 = n1;
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      constexpr int KS = KB / 2;
      const u32x4* A = reinterpret_cast<const u32x4*>(smem + par * WBUF_BYTES);
      auto ld = [&](int obi, int s, int part) {
        return A[((obi * KS + s) * 2 + part) * 64 + lane];
      };
      f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};  // cross terms, scaled by 2^11
      constexpr int PF = NRH_KPREFETCH;          // K steps in flight ahead of the one being multiplied
      u32x4 ra[PF + 1][4];                       // ring of A operand sets {hi0, lo0, hi1, lo1}
#pragma unroll
      for (int p = 0; p < PF; ++p) {
        if (p < KS) { ra[p][0] = ld(0, p, 0); ra[p][1] = ld(0, p, 1); ra[p][2] = ld(1, p, 0); ra[p][3] = ld(1, p, 1); }
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        if (s + PF < KS) {
          const int w = (s + PF) % (PF + 1);
          ra[w][0] = ld(0, s + PF, 0);
          ra[w][1] = ld(0, s + PF, 1);
          ra[w][2] = ld(1, s + PF, 0);
          ra[w][3] = ld(1, s + PF, 1);
        }
        const int c = s % (PF + 1);
        const u32x4 bhu = {in.h[s * 4 + 0], in.h[s * 4 + 1], in.h[s * 4 + 2], in.h[s * 4 + 3]};
        const u32x4 blu = {in.l[s * 4 + 0], in.l[s * 4 + 1], in.l[s * 4 + 2], in.l[s * 4 + 3]};
        const f16x8 bh = __builtin_bit_cast(f16x8, bhu), bl = __builtin_bit_cast(f16x8, blu);
        const f16x8 ah0 = __builtin_bit_cast(f16x8, ra[c][0]), al0 = __builtin_bit_cast(f16x8, ra[c][1]);
        const f16x8 ah1 = __builtin_bit_cast(f16x8, ra[c][2]), al1 = __builtin_bit_cast(f16x8, ra[c][3]);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bh, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bh, acc1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bl, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bl, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, bh, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, bh, c1, 0, 0, 0);
#if NRH_SCHED_BARRIER
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
      acc0 += c0 * LO_UNSCALE;
      acc1 += c1 * LO_UNSCALE;
    }
    epi(ch, acc0, acc1, pv);
    __syncthreads();  // the next chunk's weights are in LDS for every wave past this point
    par ^= 1;
  }
}

// ---------------- layout of the [layer][npts][256] float32 arrays of the training step ----------------
// What the SDF training forward saves and the two backward sweeps hand on (csrc/nrh_sdf.hip MODE 3, nrh_sdf_train.hip, their
// channel-split and 4-wave builds) - h, sigma', t, abar, coup, zbar - is written once and read once or twice, 1 KiB per point and
// layer: the step's HBM traffic.  Two layouts, both with the 16 points of a tile in the same contiguous 16 KiB:
//   ROWS   row-major [point][256]: a wave instruction (lane (j, q): 16 bytes of block blk of point j) touches 16 rows x 64 bytes;
//          measured 5.6 TB/s for stores and 4.2 TB/s for loads (profiles/r04/rowstore.log);
//   TILED  [tile][block 16][point 16][16 channels]: the same instruction is ONE contiguous KiB (6.3 / 6.4-7.1 TB/s); a block is
//          a 16 x 16 sub-matrix, row-major.
// sigma' and coup are private to these kernels; h, t, abar, zbar are operands of nrh_dw_gemm, which is told per operand
// (NrhDwJob.tiled_a / tiled_b) and rotates a block's rows by the block index on their way into LDS (the lanes of its LDS-DMA
// fetch permuted addresses), so that its conversion reads two adjacent channels of 16 points without a bank conflict.
// What round 5 measured (CHANGELOG.md section 7g): written as one 64-bit sum per access, a tiled address needs its own register pair
// per 16-channel block (the blocks lie 1 KiB apart, beyond the instruction's immediate offset, where row-major blocks are 64 bytes
// apart), which the 8-wave kernels - already at 256 registers - paid in 40-90 spilled registers, and the gain was gone; split
// into a wave-uniform and a 32-bit lane part (arr_ptr below) nothing spills in the sweeps and sigma' tiled is worth +3 %.
enum TrainArr { ARR_ROWS = 0, ARR_H, ARR_S1, ARR_T, ARR_ABAR, ARR_COUP, ARR_ZBAR };
#ifndef NRH_TILE_S1
#define NRH_TILE_S1 1         // sigma' tiled (round 5): +3 % on the 1 024-ray step once its addressing stopped spilling
#endif                        // (profiles/r05/train_layout_ab2.log; as first built: no gain, profiles/r05/train_s1_tiled_ab.log)
#ifndef NRH_COUP_TILE
#define NRH_COUP_TILE 1       // coup tiled (round 4: profiles/r04/coup_ab.log)
#endif
#ifndef NRH_TILE_DW
#define NRH_TILE_DW 1         // h, t, abar, zbar tiled: nrh_dw_gemm takes them through NrhDwJob.tiled_a / tiled_b (csrc/nrh_dw.hip)
#endif
__host__ __device__ constexpr bool arr_tiled(int arr) {
  return arr == ARR_ROWS ? false : (arr == ARR_S1 ? (NRH_TILE_S1 != 0) : (arr == ARR_COUP ? (NRH_COUP_TILE != 0) : (NRH_TILE_DW != 0)));
}
// Address of the lane's 4 consecutive channels (block blk, quarter q) of point `row` in layer l of such an array, written as
//   (array + WAVE-UNIFORM part: layer, block)  +  (32-bit LANE part: the point's row / tile slot)
// so that hipcc keeps the first in SGPRs and the second in ONE register per tile for every array, layer and block
// (global_load / global_store with a scalar base and a 32-bit vector offset).  As one 64-bit sum per access the tiled form cost an
// address register pair per block and 40-90 spilled registers in the 8-wave kernels (profiles/r05/spill_counts.log).
// (npts * 1 KiB < 4 GiB: at most 4 194 303 points per launch, checked by the host entry points.)
#ifndef NRH_ARR_SGPR
#define NRH_ARR_SGPR 1        // pin the wave-uniform part of a training-array address in SGPRs (scalar-base addressing)
#endif
template <bool PIN, typename F>
__device__ __forceinline__ F* arr_join(F* u, uint32_t v) {
  if constexpr (!PIN) return u + v;
#if NRH_ARR_SGPR
  // wave-uniform by construction (kernel arguments, layer and block indices): said so to hipcc with readfirstlane on the
  // address as an integer, re-typed as a GLOBAL pointer afterwards (through an opaque generic pointer the accesses became flat_load /
  // flat_store).  (An `asm("" : "+s"(address))` constraint did the same for the f16x3 kernels but broke the 4-wave float32 builds -
  // wrong gradients at 40 rays, profiles/r05/run7_tests.log - readfirstlane is the defined way to say it.)
  unsigned long long ub = (unsigned long long)u;
  ub = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(ub >> 32)) << 32) |
       (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)ub);
  typedef __attribute__((address_space(1))) F* gptr;
  return (F*)((gptr)ub + v);
#else
  return u + v;
#endif
}
// PIN: the f16x3 kernels only.  With it the 4-WAVE FLOAT32 builds (csrc/nrh_small.hip, batches of 8 193 .. 16 384 points) computed
// wrong gradients (tests at 40 rays, profiles/r05/run7_tests.log / run8_tests.log: both with an SGPR asm constraint and with
// readfirstlane) while the same source as 8-wave float32, 4-wave f16x3 and channel-split builds is bit-identical to the unpinned
// form - not understood (the address IS uniform); the exact-fp32 mode keeps the plain 64-bit sum.
template <int ARR, bool PIN = true, typename F>
__device__ __forceinline__ F* arr_ptr(F* base, int l, long long npts, long long row, int blk, int q) {
  if constexpr (arr_tiled(ARR)) {
    const int j = (int)(row & 15);
    F* const u = base + ((size_t)l * (size_t)npts * 256 + (size_t)blk * 256);
    const uint32_t v = (uint32_t)(row - j) * 256u + (uint32_t)(16 * j + 4 * q);       // block = a 16 x 16 sub-matrix, row-major
    return arr_join<PIN>(u, v);
  } else {
    F* const u = base + ((size_t)l * (size_t)npts * 256 + (size_t)blk * 16);
    const uint32_t v = (uint32_t)row * 256u + (uint32_t)(4 * q);
    return arr_join<PIN>(u, v);
  }
}
template <int V>
struct ArrTag { static constexpr int value = V; };

// ---- 16-bit hand-offs to nrh_dw_gemm (round 5, f16x3 step only) -----------------------------------------------------------------
// h, abar, zbar (and a copy of t) are read by nothing but the weight-gradient kernel, so the sweeps can write them as fp16 in the
// HALF-TILED layout [tile of 16 points][block pair 8][point 16][quarter 4][block of the pair 2][4 channels] (8 KiB per tile): a
// chunk's two blocks of a lane (8 values) become ONE 16-byte store, a wave instruction one contiguous KiB, and the KiB is, as it
// lands in LDS, the [16 points][32 channels] image ds_read_b64_tr_b16 turns into MFMA fragments (csrc/nrh_dw.hip, dw_item_half).
// Adjoints are written as the chain carries them (S x the true value, S = the step's power-of-two adjoint scale); round to nearest.
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
template <bool PIN = true>
__device__ __forceinline__ f16x8_t* half_ptr(void* base, int l, long long npts, long long row, int ch, int q) {
  const int j = (int)(row & 15);
  _Float16* const u = reinterpret_cast<_Float16*>(base) + ((size_t)l * (size_t)npts * 256 + (size_t)ch * 512);
  const uint32_t v = (uint32_t)(row - j) * 256u + (uint32_t)(32 * j + 8 * q);
  return reinterpret_cast<f16x8_t*>(arr_join<PIN>(u, v));
}
__device__ __forceinline__ f16x8_t pack_half8(const f32x4 v0, const f32x4 v1) {
  typedef float f32x8_t __attribute__((ext_vector_type(8)));
  const f32x8_t v = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
  return __builtin_convertvector(v, f16x8_t);           // v_cvt_pk_f16_f32 x 4 (round to nearest even)
}


// ---------------- packed-buffer geometry of the SDF net (floats) ----------------
// execution order: L0 | L1..L7 | FEAT | R7..R1 | R0        (R_l = W_l^T, the reverse chain)
constexpr int SDF_L0_FLOATS = 8 * 2 * 4 * 256;     // 8 chunks x (2 ob x 4 kb) KiB (39 inputs -> 64)
constexpr int SDF_REG_FLOATS = 8 * 2 * 16 * 256;   // a 256x256 stage
constexpr int SDF_R0_FLOATS = 2 * 2 * 16 * 256;    // 64 (39 used) x 256
constexpr int SDF_OFF_L0 = 0;
__host__ __device__ constexpr int sdf_off_L(int l) { return SDF_L0_FLOATS + (l - 1) * SDF_REG_FLOATS; }  // l=1..7
constexpr int SDF_OFF_FEAT = SDF_L0_FLOATS + 7 * SDF_REG_FLOATS;
__host__ __device__ constexpr int sdf_off_R(int l) { return SDF_L0_FLOATS + 8 * SDF_REG_FLOATS + (7 - l) * SDF_REG_FLOATS; }  // l=7..1
constexpr int SDF_OFF_R0 = SDF_L0_FLOATS + 15 * SDF_REG_FLOATS;
constexpr int SDF_PACKED_FLOATS = SDF_OFF_R0 + SDF_R0_FLOATS;
constexpr int SDF_BIAS_FLOATS = 9 * 256;            // L0..L7, FEAT
constexpr int SDF_HEAD_FLOATS = 257;                // w_s[256], b_s
constexpr int SDF_SCRATCH_FLOATS_PER_WAVE = 8 * 16 * 256;  // sigma' of 8 layers x 256 features x 16 points

// ---------------- packed-buffer geometry of the reflectance net ----------------
// C0a (feature part of the input, K=256) | C0b (105 per-sample/per-ray inputs, K=128) | C1..C3 | C4 (3 rows in a 32-row chunk)
// MKB = K blocks of the non-feature part of layer 0: 8 with the shadow/specular hints (105 inputs -> 128), 4 without
// (the reference's `pl-naive` preset, configs/main_config.py:67-76: 60 inputs -> 64)
constexpr int COL_OFF_C0A = 0;
constexpr int COL_OFF_C0B = SDF_REG_FLOATS;
__host__ __device__ constexpr int col_c0b_floats(int mkb) { return 8 * 2 * mkb * 256; }
__host__ __device__ constexpr int col_off_C(int l, int mkb) { return SDF_REG_FLOATS + col_c0b_floats(mkb) + (l - 1) * SDF_REG_FLOATS; }  // l=1..3
__host__ __device__ constexpr int col_off_C4(int mkb) { return SDF_REG_FLOATS + col_c0b_floats(mkb) + 3 * SDF_REG_FLOATS; }
__host__ __device__ constexpr int col_packed_floats(int mkb) { return col_off_C4(mkb) + 2 * 16 * 256; }
constexpr int COL_PACKED_FLOATS = col_packed_floats(8);
constexpr int COL_BIAS_FLOATS = 4 * 256 + 16;
__host__ __device__ constexpr int col_misc(int mkb) { return mkb == 8 ? 105 : 60; }  // [p 3, n 3, enc4(view) 27, enc4(pl) 27 (, enc4(vis) 9, enc4(cue) 36)]
constexpr int RAYMISC_STRIDE = 100;  // per-ray part of the above: 27 + 27 + 9 + 36 = 99 (+1 pad)
constexpr int MAX_SHADOW_CLIP = 16;   // largest n_shadow_importance_clip the workspace carve-up provides rows for

}  // namespace nrh
