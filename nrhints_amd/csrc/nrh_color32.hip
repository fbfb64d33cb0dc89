// The reflectance net on the wide f16x3 machinery (nrh_mlp32.h, gen_mlp32.py): one wave per SIMD, 32-point tiles,
// v_mfma_f32_32x32x16_f16, activations as MFMA B operands in AGPRs, weights streamed through the 3 x 32 KiB LDS ring.
// Evaluation only, and only behind the fused feature head (NrhNet.feat_fused): the feature block of layer 0
// (fields/reflectance_network.py:77-84, columns 60:316) arrives as `part` = W0feat * feature from the SDF kernel's FEAT stage
// (16-point D-layout tiles, csrc/nrh_sdf32.hip), so this kernel runs
//   C0  the other inputs: 6 per-sample values (point, normal) + up to 99 per-ray encodings (`raymisc`), K = 128, + part, ReLU
//   C1..C3  256 x 256, ReLU      C4  3 x 256, sigmoid   (fields/reflectance_network.py:85-96)
// A workgroup pass is one ray: 4 waves x 32 samples = 128 samples (models/neus_hint_model.py:626-637 evaluates the net at all of them).
//
// Layer-0 column order (the packer permutes W0 accordingly, packing32.pack_color32): K index e = 16 s + (i & 3) + 8 (i >> 2) + 4 hf
//   e 0..2 point, 4..6 normal, 3 / 7..15 unused (zero weights);  e = 16 + m  <->  raymisc[m], m < 99 (shadow + specular hints only:
//   the pl-naive model keeps the 16-point kernel)
#pragma once
#include "nrh_mlp32.h"

namespace nrh32 {

struct Color32Args {
  const char* w;          // block stream: C0 (8 blocks; only K steps 0..7 of each are read), C1, C2, C3 (8 each), C4 (1)
  const float* tab;       // [5][256]: rows 0..3 packed fp16 bias pairs of C0..C3, row 4 b4 (3 floats)
  const float* part;      // [ntiles16][16][64][4]  W0feat * feature (+ W0feat * b_feat), D-layout tiles of 16 points
  const float* ro;        // [nrays,3]
  const float* rd;        // [nrays,3]
  const float* tmid;      // [nrays,128]
  const float* nhat;      // [nrays*128,3] the normal fed to the net (unit normal, or the raw gradient)
  const float* raymisc;   // [nrays, raymisc_stride] per-ray encodings (enc(view) 27, enc(pl) 27, enc(vis) 9, enc(cue) 36)
  float* color;           // [nrays*128,3]
  int nrays;
  int raymisc_stride;
};

constexpr int COL32_NTAB = 5;
constexpr int COL32_BLOCKS = 33;
constexpr int COL32_LDS_TAB = RING * SLOT_BYTES;                 // 98304
constexpr int COL32_LDS_BYTES = COL32_LDS_TAB + COL32_NTAB * 1024;  // 103424
__host__ __device__ constexpr long long color32_stream_bytes() { return (long long)COL32_BLOCKS * SLOT_BYTES; }

// NRH32_COL_UNSCALED (experiment builds only, with gen_mlp32.py run under NRH32_COL_UNSCALED=1): the unscaled residual here too
#ifdef NRH32_COL_UNSCALED
#define COL32_SPLIT split2
#else
#define COL32_SPLIT split2_scaled
#endif
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void color32_kernel(const Color32Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, hf = lane >> 5;
  const uint32_t lane16 = lane * 16;
  char* const ring = smem + LDS_RING;
  const char* const tabs = smem + COL32_LDS_TAB;
  const uint32_t ring_lds = lds_off(ring);
  const uint32_t wlane = ring_lds + lane16;

  for (int i = threadIdx.x; i < COL32_NTAB * 64; i += THREADS)
    reinterpret_cast<f32x4*>(smem + COL32_LDS_TAB)[i] = reinterpret_cast<const f32x4*>(a.tab)[i];

  // the block stream (as in sdf32_kernel): block n in ring slot n % 3, n + 1 landed or landing, the pieces of n + 2 go out
  const char* const wblocks = a.w + wave * 8192;
  const char* wfetch = wblocks;
  int bfetch = 0;
  uint32_t cur_off = 0, fetch_off = 0;
  const char* fg0 = nullptr; const char* fg1 = nullptr;
  uint32_t fm0 = 0, fm1 = 0;
  auto fetch_setup = [&]() {
    fg0 = uni(wfetch);
    fg1 = uni(wfetch + 4096);
    fm0 = uni(ring_lds + fetch_off + wave * 8192);
    fm1 = fm0 + 4096;
    wfetch += SLOT_BYTES;
    if (++bfetch == COL32_BLOCKS) { bfetch = 0; wfetch = wblocks; }
    fetch_off = (fetch_off == 2 * SLOT_BYTES) ? 0 : fetch_off + SLOT_BYTES;
  };
#define W32_DMA(i) dma_piece<((i) & 3) * 1024>(((i) < 4) ? fg0 : fg1, ((i) < 4) ? fm0 : fm1, lane16)
  for (int b = 0; b < 2; ++b) {
    fetch_setup();
    W32_DMA(0); W32_DMA(1); W32_DMA(2); W32_DMA(3); W32_DMA(4); W32_DMA(5); W32_DMA(6); W32_DMA(7);
  }
#define W32_SYNC() chunk_sync<8>()
#define W32_FETCH_SETUP() fetch_setup()
#define W32_WADDR() (wlane + cur_off)
#define W32_NEXT() (cur_off = (cur_off == 2 * SLOT_BYTES) ? 0 : cur_off + SLOT_BYTES)
// cross-window prefetch inside a stage (gen_mlp32.py Window.emit): the next block's barrier behind K step 14 of this window, then
// the first fragment reads of the next window from the slot W32_NEXT() is about to select
#define W32_SYNC_MID() W32_SYNC()
#define W32_WADDR_NEXT() (wlane + ((cur_off == 2 * SLOT_BYTES) ? 0 : cur_off + SLOT_BYTES))
#define W32_BCONST bconst
  const u32x4 bconst = {hf ? 0u : 0x10003c00u, 0u, 0u, 0u};
  const char* const brow = tabs + (lane & 31) * 4;
  // this lane's 16 B inside a 16-point tile pair of `part`: sub-tile j >> 4, quarter hf (+ 2 for the odd half-blocks), point j & 15
  const uint32_t plane = (uint32_t)((j >> 4) * 16384 + hf * 256 + (j & 15) * 16);
  auto tab_row = [&](int table, int c) { return ld_init(tabs + table * 1024 + (32 * c + 4 * hf) * 4, 32); };

  for (int ray = blockIdx.x; ray < a.nrays; ray += gridDim.x) {
    const long long tile = (long long)ray * WAVES + wave;      // 32 consecutive samples of this ray
    const long long P = tile * TILE + j;
    // ---- B operands of C0 -> AGPR set 0, K steps 0..7 ----
    {
      const float tt = a.tmid[P];
      float v[4];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float p = a.ro[ray * 3 + c] + a.rd[ray * 3 + c] * tt;       // the sample point, as the SDF kernel forms it
        const float n = a.nhat[P * 3 + c];
        v[c] = hf ? n : p;
      }
      v[3] = 0.0f;
      uint32_t h0, l0, h1, l1;
      COL32_SPLIT(v[0], v[1], h0, l0);
      COL32_SPLIT(v[2], v[3], h1, l1);
      asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %1\n\tv_accvgpr_write_b32 a2, %4\n\tv_accvgpr_write_b32 a3, %4\n\t"
                   "v_accvgpr_write_b32 a64, %2\n\tv_accvgpr_write_b32 a65, %3\n\tv_accvgpr_write_b32 a66, %4\n\tv_accvgpr_write_b32 a67, %4"
                   ::"v"(h0), "v"(h1), "v"(l0), "v"(l1), "v"(0u) : "a0", "a1", "a2", "a3", "a64", "a65", "a66", "a67");
      const float* rm = a.raymisc + (long long)ray * a.raymisc_stride + 4 * hf;
#define COL32_MISC(OFF, A0, A1, A2, A3)                                                                                    \
      {                                                                                                                    \
        const f32x4 m = *reinterpret_cast<const f32x4*>(rm + (OFF));                                                       \
        uint32_t mh0, ml0, mh1, ml1;                                                                                       \
        COL32_SPLIT(m[0], m[1], mh0, ml0);                                                                                      \
        COL32_SPLIT(m[2], m[3], mh1, ml1);                                                                                      \
        asm volatile("v_accvgpr_write_b32 a" #A0 ", %0\n\tv_accvgpr_write_b32 a" #A1 ", %1\n\tv_accvgpr_write_b32 a" #A2 ", %2\n\t"    \
                     "v_accvgpr_write_b32 a" #A3 ", %3" ::"v"(mh0), "v"(mh1), "v"(ml0), "v"(ml1) : "a" #A0, "a" #A1, "a" #A2, "a" #A3); \
      }
      // K step s = 1..7, half h2: raymisc[16 (s - 1) + 8 h2 + 4 hf + 0..3] -> a[4 s + 2 h2 + {0, 1}] (hi), a[64 + ...] (lo)
      COL32_MISC(0, 4, 5, 68, 69)
      COL32_MISC(8, 6, 7, 70, 71)
      COL32_MISC(16, 8, 9, 72, 73)
      COL32_MISC(24, 10, 11, 74, 75)
      COL32_MISC(32, 12, 13, 76, 77)
      COL32_MISC(40, 14, 15, 78, 79)
      COL32_MISC(48, 16, 17, 80, 81)
      COL32_MISC(56, 18, 19, 82, 83)
      COL32_MISC(64, 20, 21, 84, 85)
      COL32_MISC(72, 22, 23, 86, 87)
      COL32_MISC(80, 24, 25, 88, 89)
      COL32_MISC(88, 26, 27, 90, 91)
      {   // K step 7: only raymisc[96..98] exist (hf = 0, first half); index 99 is padding and what follows is the next ray's row
        const f32x4 m = *reinterpret_cast<const f32x4*>(a.raymisc + (long long)ray * a.raymisc_stride + 96);
        uint32_t mh0, ml0, mh1, ml1;
        COL32_SPLIT(hf ? 0.0f : m[0], hf ? 0.0f : m[1], mh0, ml0);
        COL32_SPLIT(hf ? 0.0f : m[2], 0.0f, mh1, ml1);
        asm volatile("v_accvgpr_write_b32 a28, %0\n\tv_accvgpr_write_b32 a29, %1\n\tv_accvgpr_write_b32 a92, %2\n\tv_accvgpr_write_b32 a93, %3\n\t"
                     "v_accvgpr_write_b32 a30, %4\n\tv_accvgpr_write_b32 a31, %4\n\tv_accvgpr_write_b32 a94, %4\n\tv_accvgpr_write_b32 a95, %4"
                     ::"v"(mh0), "v"(mh1), "v"(ml0), "v"(ml1), "v"(0u) : "a28", "a29", "a92", "a93", "a30", "a31", "a94", "a95");
      }
#undef COL32_MISC
      __builtin_amdgcn_sched_barrier(0);
    }
    f32x16 hp, cp;
    f32x4 ppa, ppb, ppc, ppd;
    const char* const pbase = uni(reinterpret_cast<const char*>(a.part) + tile * 32768);
    // feature-block share of chunk c, quad g (rows 32 c + 8 g + 4 hf + 0..3): asm loads (invisible to hipcc's vmcnt bookkeeping,
    // written straight into their registers), consumed one window later
#define W32_PLOAD_ASM(dst, c, g) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(plane), "s"(pbase + (c) * 2048), "n"(((g) >> 1) * 1024 + ((g) & 1) * 512))
#define W32_BIAS(c) (*reinterpret_cast<const uint32_t*>(brow + qlayer * 1024 + (c) * 128))
    { const int qlayer = 0;
#include "gen32/col_c0.inc"
    }
    { const int qlayer = 1;
#include "gen32/col_c1.inc"
    }
    { const int qlayer = 2;
#include "gen32/col_c2.inc"
    }
    { const int qlayer = 3;
#include "gen32/col_c3.inc"
    }
#include "gen32/col_fin.inc"
#undef W32_BIAS
#undef W32_PLOAD_ASM
    // ---- C4: the 3 output rows (chunk 0, registers 0..2 of the hf = 0 lanes) + sigmoid ----
    {
      W32_SYNC();
      W32_FETCH_SETUP();
      f32x16 hh = tab_row(4, 0), cc;
      const uint32_t wa = W32_WADDR();
#include "gen32/kloop16_a0.inc"
      if (hf == 0) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const float x = __builtin_fmaf(cc[r], LO_UNSCALE, hh[r]);
          a.color[P * 3 + r] = 1.0f / (1.0f + expf(-x));                 // nrh::sigmoidf_ of the 16-point kernel
        }
      }
      W32_NEXT();
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // LDS-DMA pieces still in flight must not outlive the workgroup's LDS
#undef W32_DMA
#undef W32_SYNC
#undef W32_FETCH_SETUP
#undef W32_WADDR
#undef W32_NEXT
#undef W32_SYNC_MID
#undef W32_WADDR_NEXT
#undef W32_BCONST
}

}  // namespace nrh32
