// SDF value for SMALL point sets: one 16-point tile's output channels split over the four waves of a workgroup.
//
// The sampler of a training step evaluates the SDF network (reference SDFNetwork.sdf, fields/sdf_field.py:106-126, called by
// the hierarchical sampler models/neus_hint_model.py:175-246 through :504 / :335) on 64 and then 3 x 16 points per ray, one pass
// after the other.  With the reference's split batch (trainer/trainer.py:116-123: 512 / 8 = 64 rays per rank) a pass has 1 024 -
// 4 096 points: the evaluation kernels (nrh_sdf32.hip: one wave = one 32-point tile through all eight layers, ~40 us of that
// wave's MFMA stream whatever the batch) keep 8 - 32 of the 256 CUs busy and the pass costs their single-tile latency.
//
// Here a workgroup owns T 16-point tiles and each of its four waves (one per SIMD) computes 64 of every layer's 256 output channels
// for all of them: the MFMA work of a tile is spread over the CU's four matrix cores, the layer's activations are exchanged
// through LDS as the fp16 hi | lo pairs the next layer's B operands are made of (one barrier per layer), and the weights go
// straight from L2 into registers (each wave streams only its own quarter, 64 KiB per layer, eight K steps ahead of the MFMAs).
// A layer costs max(its MFMAs / 4, the CU's 256 KiB of weight loads) instead of its MFMAs: ~4x less latency per pass.
//
// Arithmetic: the f16x3 stages of nrh_mlp.h (run_stage, PREC 1) on the SAME packed weights (packing.pack_sdf, precision 1), the
// same MFMA sequence per output block, the same softplus, skip substitution and head - every sdf value is bit-identical to
// sdf_kernel<0, 1>'s (tests/test_gpu_split.py), which is the kernel the oracle comparisons of tests/test_gpu_parity.py pin.
#include <type_traits>
#include "nrh_mlp.h"
#include "nrh_step.h"

namespace nrh {

struct SdfSplitArgs {
  const float* w;      // packed f16x3 stages (SDF_PACKED_FLOATS * 4 bytes; only L0..L7 are read)
  const float* b;      // [9][256]
  const float* head;   // [257]
  const float* ro;     // [nrays,3]
  const float* rd;     // [nrays,3]
  const float* t;      // t[ray * t_stride + j]
  float* sdf;          // sdf[ray * sdf_stride + j]
  long long npts;
  int n_per_ray;
  int t_stride;
  int sdf_stride;
  // the per-ray sampler step that consumes this pass (nrh_step.h), run in the launch's tail by the wave that summed the tile's
  // head: only for passes of 16 samples per ray, where tile i is exactly ray i's new samples (models/neus_hint_model.py:325-331:
  // the sdf of the new samples, then cat_z_vals + the next up_sample).  Saves a launch per sampler step of a small training batch.
  int fused_step;
  StepArgs step;
};

constexpr int SPLIT_ROW = 544;     // bytes of one point's 256 fp16 activations (+32: the ds_read_b128 lane groups of the B operand hit 16 distinct 16-byte slots)
constexpr int SPLIT_ROW7 = 1056;   // bytes of one point's 256 float32 h_7 values for the head (same slot argument)
constexpr int SPLIT_TAB_FLOATS = 9 * 256 + 272;   // the bias table and the head (257 -> 272) staged once per workgroup
__host__ __device__ constexpr int split_lds_bytes(int T) { return 2 * 2 * 16 * T * SPLIT_ROW + SPLIT_TAB_FLOATS * 4; }   // two buffers x (hi, lo) x points, + tables

template <int T>
__global__ __launch_bounds__(256, T == 1 ? 2 : 1) void sdf_split_kernel(const SdfSplitArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TERM = 16 * T * SPLIT_ROW, BUF = 2 * TERM;
  static_assert(16 * T * SPLIT_ROW7 <= BUF, "h_7 rows fit one activation buffer");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  const char* const W = reinterpret_cast<const char*>(a.w);

  // ---- this workgroup's points (every wave computes the embedding of all of them: it is the B operand of its L0 MFMAs).  Their
  // loads and the table staging go out BEFORE the weight ring starts: vmcnt retires in order, so a load issued behind ring
  // requests would drain the ring when it is waited for (which is why the biases are read from LDS in the epilogues) ----
  bool valid[T];
  long long ray[T];
  int jj[T];
  float x3[T][3];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const long long P = ((long long)blockIdx.x * T + t) * TILE_PTS + j;
    valid[t] = P < a.npts;
    const long long Pc = valid[t] ? P : a.npts - 1;
    ray[t] = Pc / a.n_per_ray;
    jj[t] = (int)(Pc - ray[t] * a.n_per_ray);
    const float tt = a.t[ray[t] * a.t_stride + jj[t]];
#pragma unroll
    for (int c = 0; c < 3; ++c) x3[t][c] = (a.ro[ray[t] * 3 + c] + a.rd[ray[t] * 3 + c] * tt) * 3.0f;  // inputs * scale
  }
  float* const tab = reinterpret_cast<float*>(smem + 2 * BUF);      // [9][256] biases, then the head's 257 values
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

  // ---- the weight ring: 8 A-operand sets {hi, lo of output block 2ch; hi, lo of block 2ch + 1} of one K step, each 4 x 16 bytes
  // per lane.  Set n of the wave's sequence (L0: 2 chunks x 2 K steps; L1..L7: 2 chunks x 8 K steps) lives in slot n % 8 and is
  // requested when set n - 8 has been multiplied.
  u32x4 ring[8][4];
  auto issue = [&](int slot, const char* chunk, int KS, int s) {
    const u32x4* p = reinterpret_cast<const u32x4*>(chunk) + lane;
    ring[slot][0] = p[((0 * KS + s) * 2 + 0) * 64];
    ring[slot][1] = p[((0 * KS + s) * 2 + 1) * 64];
    ring[slot][2] = p[((1 * KS + s) * 2 + 0) * 64];
    ring[slot][3] = p[((1 * KS + s) * 2 + 1) * 64];
  };
  const char* const l0c = W + (size_t)SDF_OFF_L0 * 4 + (size_t)(2 * wave) * 8192;          // this wave's two L0 chunks (8 KiB each)
  auto lchunk = [&](int l, int ci) { return W + (size_t)sdf_off_L(l) * 4 + (size_t)(2 * wave + ci) * 32768; };
  issue(0, l0c, 2, 0);
  issue(1, l0c, 2, 1);
  issue(2, l0c + 8192, 2, 0);
  issue(3, l0c + 8192, 2, 1);
#pragma unroll
  for (int s = 0; s < 4; ++s) issue(4 + s, lchunk(1, 0), 8, s);
  __builtin_amdgcn_sched_barrier(0);

  Act<1, 4> emb[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
      float o[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int b = 2 * c2 + (r >> 2);
        o[r] = (b < 3) ? nerf_enc_entry_q<3, 6>(x3[t], b * 16 + (r & 3), q) : 0.0f;
      }
      emb[t].set_chunk(c2, o);
    }
  }
  __syncthreads();     // the tables are in LDS

  f32x4 acc0[T], acc1[T], c0[T], c1[T];
  auto zero_acc = [&]() {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      acc0[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      c0[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      c1[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  // the six MFMAs of one K step and output-block pair, per tile, in run_stage's order (nrh_mlp.h, PREC 1)
  auto mfma_set = [&](int slot, int t, const u32x4 bhu, const u32x4 blu) {
    const f16x8 bh = __builtin_bit_cast(f16x8, bhu), bl = __builtin_bit_cast(f16x8, blu);
    const f16x8 ah0 = __builtin_bit_cast(f16x8, ring[slot][0]), al0 = __builtin_bit_cast(f16x8, ring[slot][1]);
    const f16x8 ah1 = __builtin_bit_cast(f16x8, ring[slot][2]), al1 = __builtin_bit_cast(f16x8, ring[slot][3]);
    acc0[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bh, acc0[t], 0, 0, 0);
    acc1[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bh, acc1[t], 0, 0, 0);
    c0[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bl, c0[t], 0, 0, 0);
    c1[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bl, c1[t], 0, 0, 0);
    c0[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, bh, c0[t], 0, 0, 0);
    c1[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, bh, c1[t], 0, 0, 0);
  };
  // epilogue of layer L, chunk 2 * wave + CI: bias, softplus, (layer 3) the skip substitution; layers 0..6 hand the fp16 hi | lo
  // pairs to the next layer through `out`, layer 7 its float32 values to the head
  auto finish = [&](auto LC, auto CIC, char* out) {
    constexpr int L = decltype(LC)::value, CI = decltype(CIC)::value;
    const int ch = 2 * wave + CI;
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(tab + L * 256 + (2 * ch) * 16 + 4 * q);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(tab + L * 256 + (2 * ch + 1) * 16 + 4 * q);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      acc0[t] += c0[t] * LO_UNSCALE;
      acc1[t] += c1[t] * LO_UNSCALE;
      f32x4 h0, h1, d0, d1;
      softplus100_4<false>(acc0[t] + b0, h0, d0);
      softplus100_4<false>(acc1[t] + b1, h1, d1);
      if constexpr (L == 3) {
        if (wave == 3) {
          // skip connection: features 217..255 of L4's input are the embedding (fields/sdf_field.py:113-114)
          constexpr int chs = 6 + CI;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (chs == 7) h0[r] = nerf_enc_entry_q<3, 6>(x3[t], (2 * chs) * 16 + r - 217, q);            // block 14: always >= 217
            if ((2 * chs + 1) * 16 + 4 * q + r - 217 >= 0) h1[r] = nerf_enc_entry_q<3, 6>(x3[t], (2 * chs + 1) * 16 + r - 217, q);
          }
        }
      }
      if constexpr (L < 7) {
        uint32_t hw[4], lw[4];
        split_pack2(h0[0], h0[1], hw[0], lw[0]);
        split_pack2(h0[2], h0[3], hw[1], lw[1]);
        split_pack2(h1[0], h1[1], hw[2], lw[2]);
        split_pack2(h1[2], h1[3], hw[3], lw[3]);
        const u32x4 hi = {hw[0], hw[1], hw[2], hw[3]}, lo = {lw[0], lw[1], lw[2], lw[3]};
        char* const p = out + (t * 16 + j) * SPLIT_ROW + (32 * ch + 8 * q) * 2;
        *reinterpret_cast<u32x4*>(p) = hi;
        *reinterpret_cast<u32x4*>(p + TERM) = lo;
      } else {
        char* const p = out + (t * 16 + j) * SPLIT_ROW7 + ((2 * ch) * 16 + 4 * q) * 4;
        *reinterpret_cast<f32x4*>(p) = h0;
        *reinterpret_cast<f32x4*>(p + 64) = h1;
      }
    }
  };

  // ---- L0: 39 (-> 64) -> 256: two K steps per chunk, the B operands are the embedding registers ----
  {
    auto chunk0 = [&](auto CIC) {
      constexpr int CI = decltype(CIC)::value;
      zero_acc();
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int slot = CI * 2 + s;
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const u32x4 bhu = {emb[t].h[s * 4 + 0], emb[t].h[s * 4 + 1], emb[t].h[s * 4 + 2], emb[t].h[s * 4 + 3]};
          const u32x4 blu = {emb[t].l[s * 4 + 0], emb[t].l[s * 4 + 1], emb[t].l[s * 4 + 2], emb[t].l[s * 4 + 3]};
          mfma_set(slot, t, bhu, blu);
        }
        issue(slot, lchunk(1, 0), 8, 4 + slot);     // sets 8..11 of the sequence: L1, first chunk, K steps 4..7
        __builtin_amdgcn_sched_barrier(0);
      }
      finish(std::integral_constant<int, 0>(), CIC, smem);
    };
    chunk0(std::integral_constant<int, 0>());
    chunk0(std::integral_constant<int, 1>());
    __syncthreads();
  }

  // ---- L1..L7: 256 -> 256 ----
  auto layer = [&](auto LC) {
    constexpr int L = decltype(LC)::value;
    const char* const in = smem + ((L - 1) & 1) * BUF;
    char* const out = smem + (L & 1) * BUF;
    const char* const brow = in + j * SPLIT_ROW + 16 * q;          // + t * 16 rows, + 64 s, + TERM for lo
    // B operands are read just in time (two K steps ahead), once per chunk: 2 x 8 x T x 2 ds_read_b128 per layer and wave
    u32x4 bh[3][T], bl[3][T];
    auto bread = [&](int m) {                                     // m = CI * 8 + s
      const int s = m & 7;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        bh[m % 3][t] = *reinterpret_cast<const u32x4*>(brow + t * 16 * SPLIT_ROW + 64 * s);
        bl[m % 3][t] = *reinterpret_cast<const u32x4*>(brow + t * 16 * SPLIT_ROW + 64 * s + TERM);
      }
    };
    bread(0);
    bread(1);
    auto chunk = [&](auto CIC) {
      constexpr int CI = decltype(CIC)::value;
      zero_acc();
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int m = CI * 8 + s;
        const int slot = (4 + s) & 7;
        if (m + 2 < 16) bread(m + 2);
#pragma unroll
        for (int t = 0; t < T; ++t) mfma_set(slot, t, bh[m % 3][t], bl[m % 3][t]);
        if (CI == 0) issue(slot, lchunk(L, 1), 8, s);
        else if (L < 7) issue(slot, lchunk(L + 1, 0), 8, s);
        __builtin_amdgcn_sched_barrier(0);
      }
      finish(LC, CIC, out);
    };
    chunk(std::integral_constant<int, 0>());
    chunk(std::integral_constant<int, 1>());
    __syncthreads();
  };
  layer(std::integral_constant<int, 1>());
  layer(std::integral_constant<int, 2>());
  layer(std::integral_constant<int, 3>());
  layer(std::integral_constant<int, 4>());
  layer(std::integral_constant<int, 5>());
  layer(std::integral_constant<int, 6>());
  layer(std::integral_constant<int, 7>());

  // ---- sdf head: (w_s . h_7 + b_s) / scale (fields/sdf_field.py:121), summed in sdf_kernel's order by wave t for tile t ----
  const char* const h7 = smem + BUF;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    if (wave == t) {
      float head_part = 0.0f;
      const char* const row = h7 + (t * 16 + j) * SPLIT_ROW7;
#pragma unroll
      for (int ch = 0; ch < 8; ++ch) {
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(tab + 9 * 256 + (2 * ch) * 16 + 4 * q);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(tab + 9 * 256 + (2 * ch + 1) * 16 + 4 * q);
        const f32x4 h0 = *reinterpret_cast<const f32x4*>(row + ((2 * ch) * 16 + 4 * q) * 4);
        const f32x4 h1 = *reinterpret_cast<const f32x4*>(row + ((2 * ch + 1) * 16 + 4 * q) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          head_part += w0[r] * h0[r];
          head_part += w1[r] * h1[r];
        }
      }
      float part = head_part;
      part += __shfl_xor(part, 16, 64);
      part += __shfl_xor(part, 32, 64);
      const float sdfv = (part + tab[9 * 256 + 256]) / 3.0f;
      if (valid[t] && q == 0) a.sdf[ray[t] * a.sdf_stride + jj[t]] = sdfv;
      if (a.fused_step) {
        // ---- fused sampler step: this wave = this tile = ray `r`; lanes 0..15 hold the sdf of its 16 new samples ----
        const long long r = (long long)blockIdx.x * T + t;
        if (r < a.step.nrays) {
          // activation buffer 0 is free (layer 7 read it, every wave passed the layer's barrier): four 144-float rows per tile
          float* const Zs = reinterpret_cast<float*>(smem) + t * 4 * 144;
          const int n = a.step.n, j0 = lane, j1 = lane + 64;
          RayState st;
          st.z0 = (j0 < n) ? a.step.z[r * 128 + j0] : 0.0f;
          st.z1 = (j1 < n) ? a.step.z[r * 128 + j1] : 0.0f;
          st.s0 = (j0 < n) ? a.step.s[r * 128 + j0] : 0.0f;
          st.s1 = (j1 < n) ? a.step.s[r * 128 + j1] : 0.0f;
          st.n = n;
          const float zn = (lane < 16) ? a.step.znew_in[r * 16 + lane] : 0.0f;
          const float sn = (lane < 16) ? sdfv : 0.0f;       // (lane < 16: q == 0, j == lane)
          sampler_step_phases<false>(a.step, r, true, Zs, Zs + 144, Zs + 288, Zs + 432, st, zn, sn);
        }
      }
    }
  }
}

}  // namespace nrh
