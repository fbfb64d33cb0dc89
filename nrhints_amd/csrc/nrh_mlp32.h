// "Wide" per-point MLP machinery for gfx950, precision mode f16x3: one wavefront per SIMD, 512 registers, 32-point tiles.
//
// Same transposed register chain as nrh_mlp.h (H_out^T = W * H_in^T, activations never leave registers between layers),
// re-shaped around what limited the 16-point / 2-waves-per-SIMD form (CHANGELOG.md section 4.1):
//   * v_mfma_f32_32x32x16_f16: a wave owns 32 points, so every weight fragment read from LDS feeds twice the FLOPs
//     (half the ds_read_b128 and half the LDS-DMA pieces per FLOP) and MFMAs are 32 cycles apart: room for VALU fillers.
//   * 4 waves per workgroup, ONE per SIMD (amdgpu_waves_per_eu(1,1)): 256 arch VGPRs + 256 AGPRs per wave.  The B operands
//     (activations as packed fp16 hi/lo pairs: 128 registers for the layer input, 128 for the layer being produced) live in
//     AGPRs - MFMA reads them there directly - so accumulators, weight fragments and all epilogue temporaries fit the
//     arch VGPRs with no spill.  (Built with -mllvm -amdgpu-mfma-vgpr-form: accumulators stay in VGPRs, the VALU epilogue
//     reads them without v_accvgpr_read.)
//   * software pipeline inside the wave: the epilogue (activation, hi/lo split) of output chunk c-1 is issued between the
//     MFMAs of chunk c - there is no second wave on the SIMD to overlap with, and none is needed.
//   * scalar (non-packed) f32 VALU only: v_pk_*_f32 beside MFMAs is an anti-lever on gfx950 (MI355X_MICROARCH.md).
//   * the softplus(beta=100) layers run in a scaled domain u = h * 100/ln2: the accumulator IS t = 100 z / ln2, so
//     u = log2(1 + 2^t) needs no multiply on either side, and 1 - sigma'(z) = 1 / (1 + 2^t) is one v_rcp_f32.
//
// Layouts (lane = 32*hf + j: point j of the tile, half hf):
//   D32:  register r (0..15) of a 32-feature block  <->  feature (r & 3) + 8 * (r >> 2) + 4 * hf      (MFMA C/D fragment)
//   B:    K step s = 2 * block + t uses registers 8t..8t+7 of that block as its 8 fp16 elements, i.e. element i of
//         half hf is feature col32(s, hf, i) = 16 s + (i & 3) + 8 * (i >> 2) + 4 * hf - the weights absorb the permutation.
//   A:    lane (row = lane & 31, hf) holds W[32 * chunk + row][col32(s, hf, 0..7)] as 8 fp16 (16 B): one ds_read_b128.
//   chunk image in LDS / in the packed stream:  [s][hi | lo][lane 64] x 16 B   = KS * 2 KiB  (32 KiB for K = 256)
#pragma once
#include "nrh_common.h"

namespace nrh32 {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __fp16 h16x2 __attribute__((ext_vector_type(2)));

#ifndef NRH32_ABL
#define NRH32_ABL 0   // timing ablations for profiles/ubench (WRONG RESULTS): 1 no LDS-DMA, 2 no chunk barrier
#endif
constexpr int WAVES = 4;                  // one per SIMD
constexpr int THREADS = 64 * WAVES;
constexpr int TILE = 32;                  // points per wave
constexpr int GROUP = TILE * WAVES;       // points per workgroup pass
constexpr int SLOT_BYTES = 32768;         // one weight block: 32 output rows x K = 256, hi + lo
constexpr int RING = 3;                   // block n is consumed while n + 1 and n + 2 are in flight / landed
constexpr int RESIDENT_BYTES = 49152;     // weights that stay in LDS for the whole launch (the skip part E4 of the SDF net)
constexpr int NTAB = 11;                  // bias / constant tables, 256 floats each (see nrh_sdf32.hip)
constexpr int LDS_RING = 0;
constexpr int LDS_RESIDENT = RING * SLOT_BYTES;
constexpr int LDS_TAB = LDS_RESIDENT + RESIDENT_BYTES;
constexpr int LDS_BYTES = LDS_TAB + NTAB * 1024;
static_assert(LDS_BYTES <= 163840, "LDS per workgroup");
constexpr int WAVE_PIECES = SLOT_BYTES / 1024 / WAVES;   // LDS-DMA pieces per wave per block (8 KiB contiguous per wave)

constexpr float LO_SCALE = 2048.0f;
constexpr float LO_UNSCALE = 1.0f / 2048.0f;
constexpr float IK = 144.26950408889634074f;   // 100 / ln 2: scaled domain t = z * IK, u = h * IK
constexpr float KK = 6.9314718055994530942e-3f;  // ln 2 / 100

__host__ __device__ constexpr int frow(int r, int hf) { return (r & 3) + 8 * (r >> 2) + 4 * hf; }
__host__ __device__ constexpr int col32(int s, int hf, int i) { return 16 * s + (i & 3) + 8 * (i >> 2) + 4 * hf; }

// Activations as MFMA B operands: x = hi + lo with the residual lo UNSCALED (fp16 subnormals are honoured by the MFMA on
// gfx950 and so does v_cvt_pkrtz_f16_f32: profiles/r02/denorm32.log, ubench/denorm32.hip; |x - hi - lo| <= max(2^-22 |x|, 2^-25)), two values per register.  Weights (A operands)
// keep lo scaled by 2^11 in the packed stream, so a product is hh += A_hi B_hi + A_hi B_lo and cc += A_lo B_hi with cc in
// units of 2^-11 (one fused multiply-add joins them in the epilogue).
// two floats -> packed fp16, ROUND TO NEAREST EVEN (one v_cvt_pk_f16_f32 on gfx950).  The one-term builds (W32_ONE_TERM: the high half
// is all there is) need it: v_cvt_pkrtz truncates, a bias of half an ulp per operand that 256-term dot products and eight layers
// add up coherently - measured 4e-3 on the sdf and 42-57 dB PSNR with truncation against 2-6e-4 / 73-85 dB emulated with rounding.
// (In the three-term split the low half carries whatever the high half dropped, truncated or rounded.)
__device__ __forceinline__ h16x2 cvt_rn2(float a, float b) {
  typedef float f32x2_ __attribute__((ext_vector_type(2)));
  typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(h16x2, __builtin_convertvector((f32x2_{a, b}), f16x2_));
}

__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
#if defined(W32_ONE_TERM) && W32_ONE_TERM
  hi = __builtin_bit_cast(uint32_t, cvt_rn2(a, b));
  lo = 0u;
  return;
#endif
  const h16x2 h = __builtin_amdgcn_cvt_pkrtz(a, b);
  const float ra = __builtin_fmaf((float)h.x, -1.0f, a);   // v_fma_mix_f32 on the packed fp16; exact
  const float rb = __builtin_fmaf((float)h.y, -1.0f, b);
  const h16x2 l = __builtin_amdgcn_cvt_pkrtz(ra, rb);
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
}

// the same with the residual scaled by 2^11 (kept out of the fp16 subnormal range; the reflectance net's convention)
__device__ __forceinline__ void split2_scaled(float a, float b, uint32_t& hi, uint32_t& lo) {
  const h16x2 h = __builtin_amdgcn_cvt_pkrtz(a, b);
  const float ra = __builtin_fmaf((float)h.x, -LO_SCALE, a * LO_SCALE);
  const float rb = __builtin_fmaf((float)h.y, -LO_SCALE, b * LO_SCALE);
  const h16x2 l = __builtin_amdgcn_cvt_pkrtz(ra, rb);
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
}

// two values in [0,1] -> unorm16 pair (v_cvt_pknorm_u16_f32: clamp, scale by 65535, round to nearest)
__device__ __forceinline__ uint32_t unorm16x2(float a, float b) {
  typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pknorm_u16(a, b));
}

// ---- LDS-DMA of the weight stream ----
// One 1 KiB piece: lane-linear 16 B per lane.  Global address = gp + OFF + 16 * lane, LDS address = m0 + OFF + 16 * lane: the
// instruction offset moves BOTH (measured, profiles/ubench/dma_offset_test.hip), so four pieces share one (gp, m0) pair.
// Issue cost is what matters here (one wave per SIMD: every issue slot is the MFMA stream's): M0 is rewritten each time
// because nothing reserves it for us between asm statements.  M0 cannot be named as a clobber (it is a RESERVED register for
// hipcc: "inline asm clobber list contains reserved registers: m0", the entry is ignored) - the compiler's own M0 users in
// these kernels (v_readlane / LDS-DMA it never emits itself) set it right before use; the "memory" clobber keeps ordinary
// LDS reads of the ring from moving across the DMA issue.
template <int OFF>
__device__ __forceinline__ void dma_piece(const char* gp, uint32_t m0v, uint32_t lane16) {
  if (NRH32_ABL & 1) return;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:%3" ::"s"(m0v), "v"(lane16), "s"(gp), "n"(OFF) : "memory");
}
// wave-uniform values the compiler may not believe to be uniform
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ const char* uni(const char* p) {
  const uint64_t g = (uint64_t)p;   // (readfirstlane returns int: no sign extension)
  return (const char*)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(g >> 32)) << 32) |
                       (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)g));
}
// this wave's 8 KiB share of a 32 KiB block, all at once (prologue only; in the windows the pieces are spread over MFMA slots)
__device__ __forceinline__ void dma_block(const char* gsrc_w, uint32_t lds_w, uint32_t lane16) {
  dma_piece<0>(gsrc_w, lds_w, lane16); dma_piece<1024>(gsrc_w, lds_w, lane16);
  dma_piece<2048>(gsrc_w, lds_w, lane16); dma_piece<3072>(gsrc_w, lds_w, lane16);
  dma_piece<0>(gsrc_w + 4096, lds_w + 4096, lane16); dma_piece<1024>(gsrc_w + 4096, lds_w + 4096, lane16);
  dma_piece<2048>(gsrc_w + 4096, lds_w + 4096, lane16); dma_piece<3072>(gsrc_w + 4096, lds_w + 4096, lane16);
}

// wait until at most `keep` of this wave's VMEM operations (the youngest) are outstanding, then meet the other waves
template <int KEEP>
__device__ __forceinline__ void chunk_sync() {
  if (NRH32_ABL & 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(KEEP) : "memory");
  else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(KEEP) : "memory");
}

__device__ __forceinline__ uint32_t lds_off(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}

// 16 accumulator start values (D32 layout) from LDS: 4 x ds_read_b128 at base + g * gstride
__device__ __forceinline__ f32x16 ld_init(const char* base, int gstride) {
  f32x16 v;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 x = *reinterpret_cast<const f32x4*>(base + g * gstride);
    v[4 * g + 0] = x[0]; v[4 * g + 1] = x[1]; v[4 * g + 2] = x[2]; v[4 * g + 3] = x[3];
  }
  return v;
}

}  // namespace nrh32
