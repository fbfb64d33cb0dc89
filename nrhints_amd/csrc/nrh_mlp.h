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
// Weights are streamed once per workgroup (4 waves = 64 points) through a double-buffered 2 x 32 KiB
// LDS ring by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction), in "chunks" of two
// 16-row output blocks x all K blocks, pre-packed on the host so that the 64 lanes' float4 operands
// of one (output block, K block) pair are one contiguous, conflict-free 1 KiB line.
#pragma once
#include "nrh_common.h"

namespace nrh {

constexpr int WBUF_BYTES = 32768;          // one LDS weight buffer (2 ob x 16 kb x 1 KiB)
constexpr int MLP_LDS_BYTES = 2 * WBUF_BYTES;
constexpr int MLP_THREADS = 256;           // 4 waves, one 16-point tile each
constexpr int TILE_PTS = 16;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Asynchronously copy `npieces` KiB (global, contiguous) into an LDS buffer; the 4 waves interleave.
__device__ __forceinline__ void dma_chunk(const float* __restrict__ src, char* dst_lds, int npieces, int wave,
                                          int lane) {
  for (int k = wave; k < npieces; k += 4) {
    __builtin_amdgcn_global_load_lds((gptr_t)(src + k * 256 + lane * 4), (lptr_t)(dst_lds + k * 1024), 16, 0, 0);
  }
}

#define NRH_MFMA4(ACC, AV, B0, B1, B2, B3)                                   \
  ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((AV).x, (B0), ACC, 0, 0, 0);    \
  ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((AV).y, (B1), ACC, 0, 0, 0);    \
  ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((AV).z, (B2), ACC, 0, 0, 0);    \
  ACC = __builtin_amdgcn_mfma_f32_16x16x4f32((AV).w, (B3), ACC, 0, 0, 0);

// One GEMM stage: out[NCH*32 features] = A[NCH*32 x KB*16] * in[KB*16 features], all for this wave's 16 points.
//   wsrc   packed weights of this stage: NCH chunks of 2*KB KiB, chunk ch resident in LDS buffer `par` on entry
//   wnext  first chunk of whatever runs next (prefetched during the last chunk), next_pieces its size in KiB
//   init   optional accumulator start values (D-layout, NCH*8 floats) - used to split one layer's K in two
//   epi    epi(ch, acc0, acc1): consumes the two finished 16-feature blocks 2*ch and 2*ch+1
template <int KB, int NCH, bool HAS_INIT, typename Epi>
__device__ __forceinline__ void run_stage(const float* __restrict__ wsrc, const float* __restrict__ wnext,
                                          int next_pieces, char* smem, int& par, const float (&in)[KB * 4],
                                          const float* init, Epi&& epi, int wave, int lane) {
  constexpr int PIECES = 2 * KB;
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    const f32x4* A = reinterpret_cast<const f32x4*>(smem + par * WBUF_BYTES);
    char* nxt = smem + (par ^ 1) * WBUF_BYTES;
    if (ch + 1 < NCH) {
      dma_chunk(wsrc + (ch + 1) * PIECES * 256, nxt, PIECES, wave, lane);
    } else if (wnext != nullptr) {
      dma_chunk(wnext, nxt, next_pieces, wave, lane);
    }
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (HAS_INIT) {
      acc0 = f32x4{init[ch * 8 + 0], init[ch * 8 + 1], init[ch * 8 + 2], init[ch * 8 + 3]};
      acc1 = f32x4{init[ch * 8 + 4], init[ch * 8 + 5], init[ch * 8 + 6], init[ch * 8 + 7]};
    }
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
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, in[kb * 4 + 0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, in[kb * 4 + 0], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, in[kb * 4 + 1], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, in[kb * 4 + 1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, in[kb * 4 + 2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, in[kb * 4 + 2], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, in[kb * 4 + 3], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, in[kb * 4 + 3], acc1, 0, 0, 0);
      a0 = n0;
      a1 = n1;
      __builtin_amdgcn_sched_barrier(0);
    }
    epi(ch, acc0, acc1);
    __syncthreads();  // waits this wave's LDS-DMA (vmcnt(0)) and orders the buffer swap
    par ^= 1;
  }
}

// ---------------- packed-buffer geometry of the SDF net (floats) ----------------
// execution order: L0 | L1..L7 | FEAT | R7..R1 | R0        (R_l = W_l^T, the reverse chain)
constexpr int SDF_L0_FLOATS = 8 * 2 * 3 * 256;     // 8 chunks x (2 ob x 3 kb) KiB
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
// C0a (feature part of the input, K=256) | C0b (105 per-sample/per-ray inputs, K=112) | C1..C3 | C4 (3 rows in a 32-row chunk)
constexpr int COL_OFF_C0A = 0;
constexpr int COL_C0B_FLOATS = 8 * 2 * 7 * 256;
constexpr int COL_OFF_C0B = SDF_REG_FLOATS;
__host__ __device__ constexpr int col_off_C(int l) { return SDF_REG_FLOATS + COL_C0B_FLOATS + (l - 1) * SDF_REG_FLOATS; }  // l=1..3
constexpr int COL_OFF_C4 = SDF_REG_FLOATS + COL_C0B_FLOATS + 3 * SDF_REG_FLOATS;
constexpr int COL_PACKED_FLOATS = COL_OFF_C4 + 2 * 16 * 256;
constexpr int COL_BIAS_FLOATS = 4 * 256 + 16;
constexpr int COL_MISC = 105;        // [p 3, n 3, enc4(view) 27, enc4(pl) 27, enc4(vis) 9, enc4(cue) 36]
constexpr int RAYMISC_STRIDE = 100;  // per-ray part of the above: 27 + 27 + 9 + 36 = 99 (+1 pad)

}  // namespace nrh
