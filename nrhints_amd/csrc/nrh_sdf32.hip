// SDF network, precision mode f16x3, on the wide machinery of nrh_mlp32.h: value, appearance feature and analytic
// d(sdf)/dx in one pass per point.  Same contract as nrh_sdf.hip (reference: fields/sdf_field.py:106-148, called from
// models/neus_hint_model.py:325, :335-336, :504), different execution plan:
//
//   L0     emb -> 256                          softplus100 in the scaled domain (nrh_mlp32.h)
//   L1..L7 256 -> 256                          L3 has 217 real rows.  The skip connection (the reference concatenates
//          [h, emb] / sqrt2 in front of layer 4, fields/sdf_field.py:113-114) is a K split: W4's columns 217.. are zero in the
//          main stage and every chunk of layer 4 carries 8 KiB more, E4 = W4[:, 217:] / sqrt2 in the embedding's K order,
//          multiplied with the embedding (still in registers) into the accumulator start values - nothing is substituted
//   FEAT   256 -> 256, no activation           MODE 2; written as the 16-point D-layout tiles the colour kernel reads
//   HEAD   256 -> 1 (row 0 of a 32-row chunk)  sdf = (w_s . h8 + b_s) / 3
//   T7     t_7 = sigma'_7 * w_s / 3            MODE >= 1 (no GEMM)
//   R7..R1 t_{l-1} = sigma'_{l-1} * (W_l^T t_l)
//   R4e    g_emb += W4[:, 217:]^T t_4 / sqrt2  (2 chunks, after R4: both read t_4)
//   R0     g_emb += W0^T t_0, then the chain rule through the positional encoding and the input scale 3
// 1 - sigma' = 1 / (1 + 2^t) travels from the forward to the reverse sweep as unorm16 through a per-wave scratch
// (128 KiB per wave, L2 / Infinity-Cache resident).
#include "nrh_mlp32.h"

// The generated schedules (gen_mlp32.py) come from W32_GENDIR: gen32 (three product terms per K step: precision f16x3) or, in the
// one-term translation unit (nrh_wide1.hip: W32_ONE_TERM 1, namespace nrh32t), gen32_1t - precision "f16", a single fp16 MFMA pass.
#ifndef W32_GENDIR
#define W32_GENDIR gen32
#endif
#ifndef W32_ONE_TERM
#define W32_ONE_TERM 0
#endif
#define W32_STR2(x) #x
#define W32_STR(x) W32_STR2(x)
#define W32_INC(name) W32_STR(W32_GENDIR/name.inc)
#define W32_ZERO16 (f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f})

namespace nrh32 {

struct Sdf32Args {
  const char* w;        // weight stream of this MODE (sdf32_stream_bytes(MODE) bytes), chunks in execution order
  const float* tab;     // [NTAB][256]: 0..7 b_l * IK (l = 3: rows >= 217 zero), 8 b_feat, 9 {b_s / 3, 0...}, 10 w_s / 3
  const float* ro;      // [nrays,3]
  const float* rd;      // [nrays,3]
  const float* t;       // t[ray * t_stride + j]
  float* sdf;           // sdf[ray * sdf_stride + j]
  float* grad;          // [npts,3]                     (MODE >= 1)
  float* feat;          // [ceil(npts/16)][16][64][4]   (MODE 2)
  uint32_t* scratch;    // gridDim.x * WAVES * SCRATCH_WORDS_PER_WAVE (MODE >= 1)
  long long npts;
  int n_per_ray;
  int t_stride;
  int sdf_stride;
  int ngroups;          // ceil(npts / GROUP)
  uint32_t* dbg;        // diagnosis builds only (-DNRH32_TIMING): per-wave cycle totals, see profiles/ubench/sdf32_bench.hip
  int dbg_stage;        // unused
#if defined(NRH32_ARGPAD) && NRH32_ARGPAD
  void* argpad[4];      // A/B aid: the kernarg size of rounds 3-4 (hipcc's scalar register allocation of modes 1 / 2 depends on it)
#endif
};

constexpr int SCRATCH_WORDS_PER_WAVE = 8 * 8 * 2 * 64 * 4;   // [layer][chunk][half][lane] uint4
// The stream of one MODE: a 48 KiB preamble that stays resident in LDS (E4: 8 chunks x 3 K steps), then 32 KiB blocks in
// execution order: L0 x2 (four 8 KiB chunks each) | L1..L7 x56 | [FEAT x8] | HEAD | [R7..R4 x32 | R4e x2 | R3..R1 x24 | R0 x2]
__host__ __device__ constexpr int sdf32_stream_blocks(int mode) { return 2 + 56 + 1 + (mode == 2 ? 8 : 0) + (mode >= 1 ? 60 : 0); }
__host__ __device__ constexpr long long sdf32_stream_bytes(int mode) {
  return RESIDENT_BYTES + (long long)sdf32_stream_blocks(mode) * SLOT_BYTES;
}

// entry e of enc_6(x3) for this lane: e = hf ? e1 : e0 (both static); one sine per entry
__device__ __forceinline__ float emb_entry(const float (&x)[3], int e0, int e1, int hf) {
  float arg = 0.0f, raw = 0.0f;
  int kind = 0;  // 0 zero, 1 raw input, 2 sine
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int e = c ? e1 : e0;
    if (e < 0 || e >= 39) continue;
    const bool mine = (hf == c);
    if (e < 3) {
      raw = mine ? x[e] : raw;
      kind = mine ? 1 : kind;
    } else {
      int idx = e - 3;
      const float ph = (idx >= 18) ? NRH_HALF_PI : 0.0f;
      idx = idx % 18;
      const float cand = x[idx / 6] * (float)(1 << (idx % 6)) + ph;
      arg = mine ? cand : arg;
      kind = mine ? 2 : kind;
    }
  }
  const float s = nrh::sin_cw(arg);
  return (kind == 2) ? s : ((kind == 1) ? raw : 0.0f);
}
// d(entry e)/d(x_dim(e)) for this lane's entry (what autograd of the encoding yields; 0 outside the 39 entries)
__device__ __forceinline__ float emb_dentry(const float (&x)[3], int e0, int e1, int hf) {
  float arg = 0.0f, fr = 0.0f;
  int kind = 0;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int e = c ? e1 : e0;
    if (e < 0 || e >= 39) continue;
    const bool mine = (hf == c);
    if (e < 3) {
      kind = mine ? 1 : kind;
    } else {
      int idx = e - 3;
      const float ph = (idx >= 18) ? NRH_HALF_PI : 0.0f;
      idx = idx % 18;
      const float f = (float)(1 << (idx % 6));
      const float cand = x[idx / 6] * f + ph;
      arg = mine ? cand : arg;
      fr = mine ? f : fr;
      kind = mine ? 2 : kind;
    }
  }
  const float c = nrh::cos_cw(arg) * fr;
  return (kind == 2) ? c : ((kind == 1) ? 1.0f : 0.0f);
}
__host__ __device__ constexpr int emb_dim(int e) { return e < 3 ? e : ((e - 3) % 18) / 6; }

// lanes 16..31 / 48..63 <- the value of lanes 0..15 / 32..47 (v_permlane16_swap_b32: odd 16-lane rows of the first operand
// trade places with the even rows of the second); the even rows keep their own value
__device__ __forceinline__ float swap_rows16(float x) {
  const unsigned u = __builtin_bit_cast(unsigned, x);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned v = r[0];
  return __builtin_bit_cast(float, v);
}

// MODE 0: sdf | 1: sdf + gradient | 2: + feature tiles | 3: sdf + the derivative ALONG the ray, forward mode: a tile is 16 points
// (columns 0..15) and their 16 tangents (columns 16..31) through the same forward chain - no sigma' scratch, no reverse chain.
// `grad` receives rd * (d sdf / dt) / |rd|^2, so that <rd, grad> is the directional derivative the alpha formula of a shadow ray
// needs (models/neus_hint_model.py:343, true_cos = (dirs * gradients).sum(-1); only that product is used, :379-432).
template <int MODE>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void sdf32_kernel(const Sdf32Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, hf = lane >> 5;
  const uint32_t lane16 = lane * 16;
  constexpr bool JVP = MODE == 3;
  constexpr bool WANT_D = MODE == 1 || MODE == 2;
  constexpr int SMODE = JVP ? 0 : MODE;   // which stream this mode consumes (3: the forward-only one)
  constexpr long long STREAM = sdf32_stream_bytes(SMODE);
  constexpr int TPTS = JVP ? 16 : TILE;              // points per wave tile
  const bool is_pt = !JVP || j < 16;                 // JVP: point column (else: the tangent column of point j - 16)

  char* const ring = smem + LDS_RING;
  const char* const tabs = smem + LDS_TAB;
  const uint32_t ring_lds = lds_off(ring);
  const uint32_t wlane = ring_lds + lane16;   // this lane's 16 B of every 1 KiB weight line, ring slot 0
  // per-wave sigma' scratch: wave-uniform base (SGPRs) + this lane's 16 B, so every access is saddr + voffset (no 64-bit VGPR addresses)
  char* const scr = WANT_D ? reinterpret_cast<char*>(a.scratch + (size_t)(blockIdx.x * WAVES + wave) * SCRATCH_WORDS_PER_WAVE) : nullptr;
  auto scr_at = [&](int layer, int c, int half) {
    typedef __attribute__((address_space(1))) char* gchar_p;     // explicitly global: a pointer that went through asm is
    typedef __attribute__((address_space(1))) u32x4* gvec_p;     // generic otherwise, and flat_* accesses also count in lgkmcnt
    gchar_p b = (gchar_p)scr;
    asm volatile("" : "+s"(b));   // keeps the address arithmetic scalar and local (hipcc otherwise hoists 64-bit VGPR addresses)
    return (gvec_p)(b + ((layer * 8 + c) * 2 + half) * 1024 + lane16);
  };

  // constant tables -> LDS (once per workgroup)
  for (int i = threadIdx.x; i < NTAB * 64; i += THREADS)
    reinterpret_cast<f32x4*>(smem + LDS_TAB)[i] = reinterpret_cast<const f32x4*>(a.tab)[i];

  // resident weights (E4) -> LDS once per launch: 48 pieces, 12 per wave
  for (int p = wave; p < RESIDENT_BYTES / 1024; p += WAVES)
    dma_piece<0>(uni(a.w + p * 1024), uni(ring_lds + LDS_RESIDENT + p * 1024), lane16);

  // the block stream: block n sits in ring slot n % 3; while it is consumed, n + 1 has landed or is landing and the pieces
  // of n + 2 are issued (8 per wave: this wave owns bytes [8192 wave, +8192) of every block, in global memory and in LDS)
  const char* const wblocks = a.w + RESIDENT_BYTES + wave * 8192;
  const char* wfetch = wblocks;     // this wave's share of the next block to fetch
  int bfetch = 0;                   // its index in the pass
  uint32_t cur_off = 0, fetch_off = 0;   // ring offsets (bytes) of the block being consumed / being fetched
  const char* fg0 = nullptr; const char* fg1 = nullptr;
  uint32_t fm0 = 0, fm1 = 0;
  auto fetch_setup = [&]() {        // addresses for the 8 pieces of the next block; the pieces go out in later MFMA slots
    fg0 = uni(wfetch);
    fg1 = uni(wfetch + 4096);
    fm0 = uni(ring_lds + fetch_off + wave * 8192);
    fm1 = fm0 + 4096;
    wfetch += SLOT_BYTES;
    if (++bfetch == sdf32_stream_blocks(SMODE)) { bfetch = 0; wfetch = wblocks; }
    fetch_off = (fetch_off == 2 * SLOT_BYTES) ? 0 : fetch_off + SLOT_BYTES;
  };
#define W32_DMA(i) dma_piece<((i) & 3) * 1024>(((i) < 4) ? fg0 : fg1, ((i) < 4) ? fm0 : fm1, lane16)
  // blocks 0 and 1 up front
  for (int b = 0; b < 2; ++b) {
    fetch_setup();
    W32_DMA(0); W32_DMA(1); W32_DMA(2); W32_DMA(3); W32_DMA(4); W32_DMA(5); W32_DMA(6); W32_DMA(7);
  }

#ifdef NRH32_TIMING
  // diagnosis: shader cycles per stage family, summed over this wave's passes -> dbg[wave-global][8] (uint64)
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_readcyclecounter();
// (s_memtime is an SMEM read: it returns out of order with LDS reads, so nothing of it may be in flight when a K loop starts
// counting its LDS reads with lgkmcnt(N) - hence the drain after every stamp)
#define NRH32_STAMP(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); tacc[k] += t_ - tlast; tlast = t_; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } while (0)
#else
#define NRH32_STAMP(k) do { } while (0)
#endif
  for (int tg = blockIdx.x; tg < a.ngroups; tg += gridDim.x) {
    const long long tile = (long long)tg * WAVES + wave;
    const long long P = tile * TPTS + (JVP ? (j & 15) : j);
    const bool valid = P < a.npts;
    const long long Pc = valid ? P : a.npts - 1;
    const long long ray = Pc / a.n_per_ray;
    const int jj = (int)(Pc - ray * a.n_per_ray);
    const float tt = a.t[ray * a.t_stride + jj];
    float x3[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) x3[c] = (a.ro[ray * 3 + c] + a.rd[ray * 3 + c] * tt) * 3.0f;  // inputs * scale
    float xd[3] = {0.f, 0.f, 0.f};       // JVP: d x3 / dt
    if constexpr (JVP) {
#pragma unroll
      for (int c = 0; c < 3; ++c) xd[c] = a.rd[ray * 3 + c] * 3.0f;
    }

    // ---- embedding as the B operand of E4 / L0: K step s, element i <-> entry col32(s, hf, i) ----
    u32x4 ebh0, ebh1, ebh2, ebl0, ebl1, ebl2;
    {
      uint32_t eh[12], el[12];
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          float v0 = emb_entry(x3, col32(s, 0, 2 * p), col32(s, 1, 2 * p), hf);
          float v1 = emb_entry(x3, col32(s, 0, 2 * p + 1), col32(s, 1, 2 * p + 1), hf);
          if constexpr (JVP) {   // tangent columns: d entry / dt = entry'(x3) * d x3 / dt
            auto dsel = [&](int e0, int e1) {
              const float d0 = (e0 < 39) ? xd[emb_dim(e0 < 39 ? e0 : 0)] : 0.0f, d1 = (e1 < 39) ? xd[emb_dim(e1 < 39 ? e1 : 0)] : 0.0f;
              return hf ? d1 : d0;
            };
            const float t0 = emb_dentry(x3, col32(s, 0, 2 * p), col32(s, 1, 2 * p), hf) * dsel(col32(s, 0, 2 * p), col32(s, 1, 2 * p));
            const float t1 = emb_dentry(x3, col32(s, 0, 2 * p + 1), col32(s, 1, 2 * p + 1), hf) * dsel(col32(s, 0, 2 * p + 1), col32(s, 1, 2 * p + 1));
            v0 = is_pt ? v0 : t0;
            v1 = is_pt ? v1 : t1;
          }
          split2(v0, v1, eh[4 * s + p], el[4 * s + p]);
          __builtin_amdgcn_sched_barrier(0);   // two sines at a time: hipcc otherwise runs all 24 side by side and spills
        }
      ebh0 = u32x4{eh[0], eh[1], eh[2], eh[3]}; ebl0 = u32x4{el[0], el[1], el[2], el[3]};
      ebh1 = u32x4{eh[4], eh[5], eh[6], eh[7]}; ebl1 = u32x4{el[4], el[5], el[6], el[7]};
      ebh2 = u32x4{eh[8], eh[9], eh[10], eh[11]}; ebl2 = u32x4{el[8], el[9], el[10], el[11]};
    }
    auto tab_init = [&](int table, int c) { return ld_init(tabs + table * 1024 + (32 * c + 4 * hf) * 4, 32); };
    // every streamed block is older than the 8 youngest VMEM operations of this wave when it is needed (the pieces of the
    // block after it), so one static wait serves every window; the barrier makes the other waves' pieces visible and tells
    // them that this wave is done with the previous block (whose slot the next pieces overwrite)
#ifdef NRH32_TIMING
#define W32_SYNC() do { const unsigned long long s0_ = __builtin_readcyclecounter(); chunk_sync<8>(); tacc[7] += __builtin_readcyclecounter() - s0_; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); } while (0)
#else
#define W32_SYNC() chunk_sync<8>()
#endif
#define W32_FETCH_SETUP() fetch_setup()
#define W32_WADDR() (wlane + cur_off)
#define W32_NEXT() (cur_off = (cur_off == 2 * SLOT_BYTES) ? 0 : cur_off + SLOT_BYTES)
// cross-window prefetch inside a stage (gen_mlp32.py Window.emit): the next block's barrier behind K step 14 of this window, then
// the first fragment reads of the next window from the slot W32_NEXT() is about to select
#define W32_SYNC_MID() W32_SYNC()
#define W32_WADDR_NEXT() (wlane + ((cur_off == 2 * SLOT_BYTES) ? 0 : cur_off + SLOT_BYTES))

    NRH32_STAMP(0);   // setup: rays, embedding
    // The layers ping-pong between the two AGPR sets (a[0:127], a[128:255]): L0 writes set 0, L1 reads it and writes set 1,
    // ... no copies.  The last chunk of every stage stays PENDING in (hp, cp): its epilogue runs in the MFMA shadows of the
    // next stage's first window (its outputs are that stage's K steps 14 and 15, needed last).
    f32x16 hp, cp;
#if W32_ONE_TERM
    cp = W32_ZERO16;      // (one-term build: no cross-term accumulator; the generated epilogues do not read it)
#endif
    // Every layer's windows start from zero and take their bias through one extra MFMA (gen_mlp32.py gen_stage, bias_mfma):
    // table rows 0..7 hold one packed fp16 pair per output row, (b_hi | b_lo * 2^11 << 16), and the B operand is the constant
    // [1, 2^-11, 0, ...] of the hf = 0 lanes.
    const u32x4 bconst = {(hf || !is_pt) ? 0u : 0x10003c00u, 0u, 0u, 0u};     // (tangent columns take no bias)
    const char* const brow = tabs + (lane & 31) * 4;
    // JVP: a tangent lane reads its point lane's value 16 lanes below (v_permlane16_swap: odd rows of 16 <- even rows)
#define W32_SWAP(x) swap_rows16(x)
#define W32_ISPT is_pt
    // layer 4's skip part: E4 * emb on top of the window's sums, 9 MFMAs over the resident block (B operands in VGPRs)
    auto skip_e4 = [&](int c, f32x16& hh, f32x16& cc) {
      const uint32_t wa = ring_lds + LDS_RESIDENT + c * 6144 + lane16;   // resident E4 chunk c: 3 K steps
#include W32_INC(kloop3v_acc)
    };
#define W32_BIAS(c) (*reinterpret_cast<const uint32_t*>(brow + qlayer * 1024 + (c) * 128))
#define W32_BCONST bconst
#define W32_SKIP(c, HH, CC) do { if (qlayer == 4) skip_e4(c, HH, CC); } while (0)
// cache policy of the sigma' scratch (128 KiB per wave, rewritten every pass): NRH32_QPOL bit 0 = plain (temporal) stores,
// bit 1 = plain loads; default: plain stores, non-temporal loads (measured -2 % against non-temporal stores, profiles/r02/qpol_ab.log)
#ifndef NRH32_QPOL
#define NRH32_QPOL 1
#endif
#if NRH32_QPOL & 1
#define W32_QST_(val, p) (*(p) = (val))
#else
#define W32_QST_(val, p) __builtin_nontemporal_store((val), (p))
#endif
#if NRH32_QPOL & 2
#define W32_QLD_NT ""
#else
#define W32_QLD_NT " nt"
#endif
#define W32_QSTORE(c, half, val) W32_QST_((val), scr_at(qlayer, c, half))
#define W32_QSTORE_P(c, half, val) W32_QST_((val), scr_at(qlayer - 1, c, half))
#ifndef NRH32_Q7REG
#define NRH32_Q7REG 1      // layer 7's sigma' words stay in registers from its epilogues to the T7 pass: +0.6 % frame rate, 16 KiB of scratch round trip per tile less; 0 = the scratch path
#endif
#if NRH32_Q7REG
    u32x4 q0a, q0b, q1a, q1b, q2a, q2b, q3a, q3b, q4a, q4b, q5a, q5b, q6a, q6b, qpa, qpb;
#define W32_Q7_0_0 q0a
#define W32_Q7_0_1 q0b
#define W32_Q7_1_0 q1a
#define W32_Q7_1_1 q1b
#define W32_Q7_2_0 q2a
#define W32_Q7_2_1 q2b
#define W32_Q7_3_0 q3a
#define W32_Q7_3_1 q3b
#define W32_Q7_4_0 q4a
#define W32_Q7_4_1 q4b
#define W32_Q7_5_0 q5a
#define W32_Q7_5_1 q5b
#define W32_Q7_6_0 q6a
#define W32_Q7_6_1 q6b
#define W32_Q7_7_0 qpa
#define W32_Q7_7_1 qpb
#endif
    // ---- L0 ----
    {
      const int qlayer = 0;
      if constexpr (JVP) {
#include W32_INC(l0_j)
      } else if constexpr (WANT_D) {
#include W32_INC(l0_d1)
      } else {
#include W32_INC(l0_d0)
      }
    }
    NRH32_STAMP(1);   // L0
    // ---- L1..L7: odd layers read set 0 / write set 1, even layers the other way round ----
    constexpr bool Q7REG = NRH32_Q7REG && (MODE == 1 || MODE == 2);
    for (int l = 1; l <= (Q7REG ? 5 : 7); l += 2) {
      {
        const int qlayer = l;
        if constexpr (JVP) {
#include W32_INC(fwd_j_p0)
        } else if constexpr (WANT_D) {
#include W32_INC(fwd_d1_p0)
        } else {
#include W32_INC(fwd_d0_p0)
        }
      }
      if (l == 7) break;
      {
        const int qlayer = l + 1;
        if constexpr (JVP) {
#include W32_INC(fwd_j_p1)
        } else if constexpr (WANT_D) {
#include W32_INC(fwd_d1_p1)
        } else {
#include W32_INC(fwd_d0_p1)
        }
      }
    }
#if NRH32_Q7REG
    if constexpr (Q7REG) {
      // layer 7 on its own: its sigma' words go to the registers the T7 pass reads instead of through the scratch
      // (its first window still finishes layer 6's last chunk: W32_QSTORE_P stays the scratch store)
      const int qlayer = 7;
#pragma push_macro("W32_QSTORE")
#undef W32_QSTORE
#define W32_QSTORE(c, half, val) W32_Q7_##c##_##half = (val)
#include W32_INC(fwd_d1_p0)
#pragma pop_macro("W32_QSTORE")
    }
#endif
    {
      const int qlayer = 8;   // the pending chunk 7 of layer 7 -> set 1, where FEAT / HEAD read their input
      if constexpr (JVP) {
#include W32_INC(fwd_fin_j)
      } else if constexpr (WANT_D) {
#if NRH32_Q7REG
#pragma push_macro("W32_QSTORE_P")
#undef W32_QSTORE_P
#define W32_QSTORE_P(c, half, val) W32_Q7_##c##_##half = (val)
#include W32_INC(fwd_fin_d1)
#pragma pop_macro("W32_QSTORE_P")
#else
#include W32_INC(fwd_fin_d1)
#endif
      } else {
#include W32_INC(fwd_fin_d0)
      }
      (void)qlayer;
    }
#undef W32_BIAS
#undef W32_BCONST
#undef W32_SKIP
#undef W32_QSTORE
#undef W32_QSTORE_P
#undef W32_SWAP
#undef W32_ISPT

    NRH32_STAMP(2);   // L1..L7
    // ---- FEAT (MODE 2) and HEAD ----
    if (MODE == 2) {
      const long long t16 = 2 * tile + (j >> 4);
      const bool t16_ok = t16 * 16 < a.npts;
      float* const ft = a.feat + (size_t)t16 * 4096 + ((j & 15) + 16 * hf) * 4;
      // the feature head as a pipelined stage like the layers (epilogue of chunk c - 1 under the K loop of chunk c, bias through
      // an MFMA): features 32c + 8g + 4hf + r  ->  16-row block 2c + (g >> 1), quarter 2 (g & 1) + hf of the 16-point tile
#define W32_FSTORE(c, g, val) do { if (t16_ok) __builtin_nontemporal_store((val), reinterpret_cast<f32x4*>(ft + ((2 * (c) + ((g) >> 1)) * 64 + 32 * ((g) & 1)) * 4)); } while (0)
#define W32_BIAS(c) (*reinterpret_cast<const uint32_t*>(brow + 8 * 1024 + (c) * 128))
#define W32_BCONST bconst
      {
#include W32_INC(feat)
      }
#include W32_INC(feat_fin)
#undef W32_FSTORE
#undef W32_BIAS
#undef W32_BCONST
    }
    // q_7 words for the T7 pass: requested inside the HEAD window, consumed after it (their L2 latency passes under HEAD's K loop)
#if !NRH32_Q7REG
    u32x4 q0a, q0b, q1a, q1b, q2a, q2b, q3a, q3b, q4a, q4b, q5a, q5b, q6a, q6b, qpa, qpb;
#endif
    const char* const q7base = uni(scr + 7 * 16384);
    // nt: served by L2; asm: the destination registers are written straight by the load (no compiler copy of a value still in
    // flight - checked in the build's ISA, csrc/check_wide_isa.py, run by the Makefile), the wait is in t7.inc
#define W32_QLOAD7_ASM(dst, c, half) asm volatile("global_load_dwordx4 %0, %1, %2" W32_QLD_NT : "=v"(dst) : "v"(lane16), "s"(q7base + ((c) * 2 + (half)) * 1024))
    {
      W32_SYNC();
      W32_FETCH_SETUP();
      if constexpr (WANT_D && !Q7REG) {
#include W32_INC(t7_loads)
        W32_QLOAD7_ASM(qpa, 7, 0);
        W32_QLOAD7_ASM(qpb, 7, 1);
      }
      f32x16 hh = tab_init(9, 0), cc;
#if W32_ONE_TERM
      cc = W32_ZERO16;
#endif
      if constexpr (JVP) {     // tangent columns: the head is linear, its bias does not differentiate
#pragma unroll
        for (int r = 0; r < 16; ++r) hh[r] = is_pt ? hh[r] : 0.0f;
      }
      const uint32_t wa = W32_WADDR();
#include W32_INC(kloop16)
      const float head = __builtin_fmaf(cc[0], LO_UNSCALE, hh[0]);
      if constexpr (JVP) {
        if (valid && hf == 0) {
          if (is_pt) {
            a.sdf[ray * a.sdf_stride + jj] = head;
          } else {     // head = d sdf / dt of point j - 16
            const float rx = a.rd[ray * 3 + 0], ry = a.rd[ray * 3 + 1], rz = a.rd[ray * 3 + 2];
            const float k = head / (rx * rx + ry * ry + rz * rz);
            a.grad[P * 3 + 0] = rx * k; a.grad[P * 3 + 1] = ry * k; a.grad[P * 3 + 2] = rz * k;
          }
        }
      } else {
        if (valid && hf == 0) a.sdf[ray * a.sdf_stride + jj] = head;
      }
      W32_NEXT();
    }

    NRH32_STAMP(3);   // FEAT + HEAD
    if constexpr (WANT_D) {
      // ---- T7: t_7 = (1 - q_7) * w_s / 3: chunks 0..6 straight into set 0 (R7's input), chunk 7 as R7's pending pair ----
#define W32_A8(c) tab_init(10, c)
#include W32_INC(t7)
      hp = W32_A8(7);
      cp = f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#undef W32_A8
#undef W32_QLOAD7_ASM

      NRH32_STAMP(4);   // T7
      // g_emb chunk c (register r <-> entry 32c + frow(r, hf)) contracted with d enc / dx.  The 20 derivative values are
      // recomputed for each of the two stages that need them (R4e, R0): keeping them live across R4..R1 costs more
      // registers than 20 cosines cost time.
      float dx[3] = {0.f, 0.f, 0.f};
      auto epi_emb = [&](int c, const f32x16& hh, const f32x16& cc) {
        float xx[3] = {x3[0], x3[1], x3[2]};
        asm volatile("" : "+v"(xx[0]), "+v"(xx[1]), "+v"(xx[2]));   // not hoisted, not shared between the two stages
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (c == 1 && r >= 4) continue;
          const int e0 = 32 * c + frow(r, 0), e1 = 32 * c + frow(r, 1);
          if (e0 >= 39) continue;
          const float ge = __builtin_fmaf(cc[r], LO_UNSCALE, hh[r]);
          const float v = ge * emb_dentry(xx, e0, e1, hf);
          const int d0 = emb_dim(e0), d1 = (e1 < 39) ? emb_dim(e1) : d0;
          if (d0 == d1) {
            dx[d0] += v;
          } else {
            dx[d0] += hf ? 0.0f : v;
            dx[d1] += hf ? v : 0.0f;
          }
          if (r & 1) __builtin_amdgcn_sched_barrier(0);
        }
      };
      auto emb_stage = [&]() {   // two 32-row chunks (R4e or R0) over AGPR set 1
        for (int c = 0; c < 2; ++c) {
          W32_SYNC();
          W32_FETCH_SETUP();
          f32x16 hh, cc;
#if W32_ONE_TERM
          cc = W32_ZERO16;
#endif
          const uint32_t wa = W32_WADDR();
#include W32_INC(kloop16z)
          if (c == 0) epi_emb(0, hh, cc); else epi_emb(1, hh, cc);
          W32_NEXT();
        }
      };

      // ---- R7..R1 ----
      // q words of chunk c of layer l - 1: asm loads (invisible to hipcc's vmcnt bookkeeping), nt: served by L2
#define W32_QLOAD_ASM(dst, c, half) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" W32_QLD_NT : "=v"(dst) : "v"(lane16), "s"(qbase + (c) * 2048), "n"((half) * 1024))
      // R7 R6 | R5 R4 R4e | R3 R2 | R1 finish R0: odd layers read set 0 / write set 1, even layers the other way round, so both
      // embedding-gradient stages (after R4: t_4 is R4's input; after R1: t_0) read set 1 - one copy of emb_stage in the code
      for (int k = 0; k < 4; ++k) {
        {
          const int l = 7 - 2 * k;
          const char* const qbase = uni(scr + (l - 1) * 16384);
#include W32_INC(rev_p0)
        }
        if (k == 3) {
          const int l = 0;
          (void)l;
#include W32_INC(rev_fin)
        } else {
          const int l = 6 - 2 * k;
          const char* const qbase = uni(scr + (l - 1) * 16384);
#include W32_INC(rev_p1)
        }
        if (k == 1 || k == 3) emb_stage();
      }
#undef W32_QLOAD_ASM

      NRH32_STAMP(5);   // R7..R1 (+ R4e)
      // ---- the chain rule through the encoding ----
#pragma unroll
      for (int c = 0; c < 3; ++c) dx[c] += __shfl_xor(dx[c], 32, 64);
      if (valid && hf == 0) {
#pragma unroll
        for (int c = 0; c < 3; ++c) a.grad[P * 3 + c] = dx[c] * 3.0f;  // d(3x)/dx
      }
      NRH32_STAMP(6);   // R0 + gradient out
    }
  }
#ifdef NRH32_TIMING
  if (a.dbg != nullptr && lane == 0) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(a.dbg) + (size_t)(blockIdx.x * WAVES + wave) * 8;
    for (int k = 0; k < 8; ++k) o[k] = tacc[k];
  }
#endif
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the last prefetch (never consumed) has landed before the LDS is released
}

}  // namespace nrh32
