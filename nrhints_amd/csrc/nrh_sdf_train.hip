// Backward of (sdf, feature, d sdf/dx) for training on gfx950: the two sweeps of the hand-derived second-order backward
// (nrhints_amd/sdf_function.py states the maths and is the torch reference of these kernels).
//
// The reference gets d sdf/dx with autograd (create_graph=True, fields/sdf_field.py:136-148) and lets autograd
// differentiate that graph again in loss.backward() (pipelines/base_pipeline.py:59-62).  Written out, the backward is
//
//   tangent sweep (runs FORWARD through the layers, same packed weights L0..L7 as the value chain):
//       abar_0 = 3 * dc * gbar[dim]                              (adjoint of the embedding gradient ge)
//       tbar_l = W_l abar_l
//       coup_l = 100 (1 - s'_l) t_l tbar_l                       (= s''_l a_{l+1} tbar_l, the coupling into zbar_l)
//       abar_{l+1} = s'_l tbar_l                                 (layer 3: entries 217.. are abar_0 again - the skip)
//   value sweep (runs in REVERSE, packed weights FEAT^T, R7..R0):
//       hbar_7 = Wf^T fbar + sbar w_s / 3
//       zbar_l = s'_l hbar_l + coup_l ;   hbar_{l-1} = W_l^T zbar_l
//       pbar   = 3 * sum_e (xbar_0[e] + xbar_4[217 + e]) dc[e]   (value path only; the host adds the d2c term)
//
// and the weight gradients are plain GEMMs over the saved row-major arrays (host side, rocBLAS):
//       dW_l = zbar_l^T x_l + t_l^T abar_l,   db_l = sum_P zbar_l,   ...
// Both sweeps are the transposed register chain of nrh_mlp.h with different epilogues; s'_l, t_l come from the training
// forward (sdf_kernel<3>), abar / coup / zbar are written row-major [layer][npts][256].
#include "nrh_mlp.h"

namespace nrh {

struct SdfTrainArgs {
  const float* w;        // packed SDF stages (same buffer as the forward)
  const float* wt_feat;  // packed Wf^T as one regular 256x256 stage (value sweep)
  const float* head;     // [257]
  const float* ro;       // points as rays: p = ro[ray] + rd[ray] * t[ray * t_stride + j]
  const float* rd;
  const float* t;
  const float* s1;       // [8][npts][256] from the forward
  const float* tt;       // [8][npts][256] from the forward
  const float* gbar;     // [npts][3]   tangent sweep in
  float* abar;           // [8][npts][256] tangent sweep out: abar_{l+1} at index l (index 7: s'_7 tbar_7, for d w_s)
  float* coup;           // [8][npts][256] floats, tangent sweep out / value sweep in: tiled (nrh_mlp.h arr_ptr<ARR_COUP>), not row-major
  float* gebar;          // [npts][64]  tangent sweep out: abar_0 (39 used)
  const float* fbar;     // [npts][256] value sweep in
  const float* sbar;     // [npts]      value sweep in
  float* zbar;           // [8][npts][256] value sweep out
  float* pbar;           // [npts][3]   value sweep out
  long long npts;
  int n_per_ray;
  int t_stride;
  int ntile_groups;
  void* abar16;          // optional (f16x3): fp16 half-tiled [8][npts][256]; layers 0..6 of abar go HERE, as S x the value, INSTEAD of abar
  void* zbar16;          // optional (f16x3): the same for layers 1..7 of zbar
  int coup16;            // with abar16: coup - private to the two sweeps - is fp16 half-tiled as well (S x the value, in the same buffer)
  const void* t16;       // optional (f16x3): layers 1..6 of t are read from this fp16 half-tiled array instead of tt (SdfArgs.t16_only)
  const float* dyn;      // optional (f16x3) device {S, 1 / S}: the step's adjoint scale from the seeds' range (adjoint_range_kernel)
  float adj_scale;       // f16x3 only: a power of two S.  The adjoint chain runs on S * (the seeds) and its outputs leave as 1 / S *
                         // (the result): the loss is normalised by the ray count (pipelines/base_pipeline.py:57), so at 1 024 rays
                         // per step the adjoints are ~ 1e-3 of a single ray's and their fp16 halves (absolute floor 3e-11 under
                         // 6e-5) lose up to 6e-3 of a gradient tensor's scale - measured against the reference's 1 024-ray step,
                         // profiles/r05/train1024_diag*.log.  The host passes 2^round(log2(rays)): batch-size independent ranges.
};

struct TrainPre {
  f32x4 s0, s1, t0, t1;  // sigma' and t (tangent) / sigma' and coup (value) of the two blocks of a chunk
  f32x4 w0, w1;          // value sweep, first stage only: sdf-head weights
};

// 3 * d(entry e)/dx_{dim e} * g[dim e]  for the lane's entry e = base + 4q (0 outside the 39 entries); g already holds 3*gbar
__device__ __forceinline__ float enc_dentry_dot_q(const float (&x)[3], const float (&g)[3], int base, int q) {
  constexpr int D = 3, F = 6, N = D * (2 * F + 1);
  float arg = 0.0f, mul = 0.0f;
  int kind = 0;  // 0 -> zero, 1 -> raw input (derivative 1), 2 -> sine (derivative cos * 2^k)
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {
    const int e = base + 4 * qq;
    if (e < 0 || e >= N) continue;
    const bool mine = (q == qq);
    if (e < D) {
      mul = mine ? g[e] : mul;
      kind = mine ? 1 : kind;
    } else {
      int idx = e - D;
      const float ph = (idx >= D * F) ? NRH_HALF_PI : 0.0f;
      idx = idx % (D * F);
      const int d = idx / F, k = idx % F;
      arg = mine ? x[d] * (float)(1 << k) + ph : arg;
      mul = mine ? g[d] * (float)(1 << k) : mul;
      kind = mine ? 2 : kind;
    }
  }
  const float c = cos_cw(arg);
  return (kind == 2) ? c * mul : ((kind == 1) ? mul : 0.0f);
}

// (layouts of s1 / t / abar / coup / zbar: nrh_mlp.h arr_ptr)

// ------------------------------------------------------------------------------------------------------------------
// tangent sweep
// ------------------------------------------------------------------------------------------------------------------
template <int PREC>
__global__ __launch_bounds__(MLP_THREADS, 2) void sdf_tangent_kernel(const SdfTrainArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  int par = 0;

  dma_chunk(a.w + SDF_OFF_L0, smem, 8, wave, lane);
  __syncthreads();
  const float S = (PREC == 1) ? (a.dyn ? a.dyn[0] : a.adj_scale) : 1.0f, IS = 1.0f / S;   // SdfTrainArgs.adj_scale: the sweep is linear in gbar

  for (int tg = blockIdx.x; tg < a.ntile_groups; tg += gridDim.x) {
    const long long tile = (long long)tg * WG_WAVES + wave;
    const bool tile_ok = tile * TILE_PTS < a.npts;
    const long long row = tile_ok ? tile * TILE_PTS + j : j;
    const long long ray = row / a.n_per_ray;
    const int jj = (int)(row - ray * a.n_per_ray);
    const float tpar = a.t[ray * a.t_stride + jj];
    float x3[3], g3[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      x3[c] = (a.ro[ray * 3 + c] + a.rd[ray * 3 + c] * tpar) * 3.0f;
      g3[c] = a.gbar[row * 3 + c] * 3.0f * S;
    }

    // abar_0 (embedding layout of the forward's L0 input)
    Act<PREC, 4> emb;
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
      float o[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int b = 2 * c2 + (r >> 2);
        o[r] = (b < 3) ? enc_dentry_dot_q(x3, g3, b * 16 + (r & 3), q) : 0.0f;
      }
      emb.set_chunk(c2, o);
      if (tile_ok) {
        float* gr = a.gebar + (size_t)row * 64 + 4 * q;
        *reinterpret_cast<f32x4*>(gr + (2 * c2) * 16) = f32x4{o[0], o[1], o[2], o[3]} * IS;
        *reinterpret_cast<f32x4*>(gr + (2 * c2 + 1) * 16) = f32x4{o[4], o[5], o[6], o[7]} * IS;
      }
    }

    Act<PREC, 16> h;
    for (int s = 0; s <= 7; ++s) {
      auto pre = [&](int ch) {
        TrainPre p;
        p.s0 = ld_stream(reinterpret_cast<const f32x4*>(arr_ptr<ARR_S1, PREC == 1>(a.s1, s, a.npts, row, 2 * ch, q)));
        p.s1 = ld_stream(reinterpret_cast<const f32x4*>(arr_ptr<ARR_S1, PREC == 1>(a.s1, s, a.npts, row, 2 * ch + 1, q)));
        if (PREC == 1 && a.t16 && s >= 1 && s < 7) {
          const f16x8_t tv = ld_stream(half_ptr<true>(const_cast<void*>(a.t16), s, a.npts, row, ch, q));
          p.t0 = f32x4{(float)tv[0], (float)tv[1], (float)tv[2], (float)tv[3]};
          p.t1 = f32x4{(float)tv[4], (float)tv[5], (float)tv[6], (float)tv[7]};
          return p;
        }
        p.t0 = ld_stream(reinterpret_cast<const f32x4*>(arr_ptr<ARR_T, PREC == 1>(a.tt, s, a.npts, row, 2 * ch, q)));
        p.t1 = ld_stream(reinterpret_cast<const f32x4*>(arr_ptr<ARR_T, PREC == 1>(a.tt, s, a.npts, row, 2 * ch + 1, q)));
        return p;
      };
      Act<PREC, 16> ho;
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1, const TrainPre& p) {
        const f32x4 c0 = (1.0f - p.s0) * p.t0 * acc0 * 100.0f;
        const f32x4 c1 = (1.0f - p.s1) * p.t1 * acc1 * 100.0f;
        f32x4 n0 = p.s0 * acc0, n1 = p.s1 * acc1;
        if (ch >= 6 && s == 3) {
          // abar_4 = [abar_4h (217), abar_0 (39)]: the skip connection (fields/sdf_field.py:113-114)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (ch == 7) n0[r] = enc_dentry_dot_q(x3, g3, (2 * ch) * 16 + r - 217, q);
            if ((2 * ch + 1) * 16 + 4 * q + r - 217 >= 0) n1[r] = enc_dentry_dot_q(x3, g3, (2 * ch + 1) * 16 + r - 217, q);
          }
        }
        if (tile_ok) {
          if (PREC == 1 && a.coup16) {
            st_stream(half_ptr<true>(a.coup, s, a.npts, row, ch, q), pack_half8(c0, c1));
          } else {
            st_stream(reinterpret_cast<f32x4*>(arr_ptr<ARR_COUP, PREC == 1>(a.coup, s, a.npts, row, 2 * ch, q)), c0 * IS);
            st_stream(reinterpret_cast<f32x4*>(arr_ptr<ARR_COUP, PREC == 1>(a.coup, s, a.npts, row, 2 * ch + 1, q)), c1 * IS);
          }
          if (PREC == 1 && a.abar16 && s < 7) {
            st_stream(half_ptr<true>(a.abar16, s, a.npts, row, ch, q), pack_half8(n0, n1));
          } else {
            st_stream(reinterpret_cast<f32x4*>(arr_ptr<ARR_ABAR, PREC == 1>(a.abar, s, a.npts, row, 2 * ch, q)), n0 * IS);
            st_stream(reinterpret_cast<f32x4*>(arr_ptr<ARR_ABAR, PREC == 1>(a.abar, s, a.npts, row, 2 * ch + 1, q)), n1 * IS);
          }
        }
        ho.set_chunk(ch, n0, n1);
      };
      if (s == 0) {
        run_stage<PREC, 4, 8, false, true>(a.w + SDF_OFF_L0, a.w + sdf_off_L(1), 32, smem, par, emb, nullptr, pre, epi, wave, lane);
      } else {
        const float* wnxt = (s < 7) ? a.w + sdf_off_L(s + 1) : a.w + SDF_OFF_L0;
        run_stage<PREC, 16, 8, false, true>(a.w + sdf_off_L(s), wnxt, s < 7 ? 32 : 8, smem, par, h, nullptr, pre, epi, wave, lane);
      }
      h = ho;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// value (adjoint) sweep
// ------------------------------------------------------------------------------------------------------------------
template <int PREC>
__global__ __launch_bounds__(MLP_THREADS, 2) void sdf_adjoint_kernel(const SdfTrainArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  int par = 0;

  dma_chunk(a.wt_feat, smem, 32, wave, lane);
  __syncthreads();
  const float S = (PREC == 1) ? (a.dyn ? a.dyn[0] : a.adj_scale) : 1.0f, IS = 1.0f / S;   // SdfTrainArgs.adj_scale: linear in (fbar, sbar, coup)

  for (int tg = blockIdx.x; tg < a.ntile_groups; tg += gridDim.x) {
    const long long tile = (long long)tg * WG_WAVES + wave;
    const bool tile_ok = tile * TILE_PTS < a.npts;
    const long long row = tile_ok ? tile * TILE_PTS + j : j;
    const long long ray = row / a.n_per_ray;
    const int jj = (int)(row - ray * a.n_per_ray);
    const float tpar = a.t[ray * a.t_stride + jj];
    float x3[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) x3[c] = (a.ro[ray * 3 + c] + a.rd[ray * 3 + c] * tpar) * 3.0f;
    const float sb3 = a.sbar[row] / 3.0f * S;

    Act<PREC, 16> h;
#pragma unroll
    for (int ch = 0; ch < 8; ++ch) {
      const f32x4 v0 = ld_stream(reinterpret_cast<const f32x4*>(arr_ptr<ARR_ROWS, PREC == 1>(a.fbar, 0, a.npts, row, 2 * ch, q)));
      const f32x4 v1 = ld_stream(reinterpret_cast<const f32x4*>(arr_ptr<ARR_ROWS, PREC == 1>(a.fbar, 0, a.npts, row, 2 * ch + 1, q)));
      h.set_chunk(ch, v0 * S, v1 * S);
    }

    float skip[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) skip[i] = 0.0f;
    // s = 8: FEAT^T (-> zbar_7);  s = 7..1: R_s = W_s^T (-> zbar_{s-1})
    for (int s = 8; s >= 1; --s) {
      const int lz = s - 1;  // layer whose zbar this stage's epilogue produces
      auto pre = [&](int ch) {
        TrainPre p;
        p.s0 = ld_stream(reinterpret_cast<const f32x4*>(arr_ptr<ARR_S1, PREC == 1>(a.s1, lz, a.npts, row, 2 * ch, q)));
        p.s1 = ld_stream(reinterpret_cast<const f32x4*>(arr_ptr<ARR_S1, PREC == 1>(a.s1, lz, a.npts, row, 2 * ch + 1, q)));
        if (PREC == 1 && a.coup16) {
          // (stored as S x the value: IS here cancels the S of the epilogue - exact, powers of two)
          const f16x8_t cv = ld_stream(half_ptr<true>(a.coup, lz, a.npts, row, ch, q));
          p.t0 = f32x4{(float)cv[0], (float)cv[1], (float)cv[2], (float)cv[3]} * IS;
          p.t1 = f32x4{(float)cv[4], (float)cv[5], (float)cv[6], (float)cv[7]} * IS;
        } else {
          p.t0 = ld_stream(reinterpret_cast<const f32x4*>(arr_ptr<ARR_COUP, PREC == 1>(a.coup, lz, a.npts, row, 2 * ch, q)));
          p.t1 = ld_stream(reinterpret_cast<const f32x4*>(arr_ptr<ARR_COUP, PREC == 1>(a.coup, lz, a.npts, row, 2 * ch + 1, q)));
        }
        if (s == 8) {
          p.w0 = *reinterpret_cast<const f32x4*>(a.head + (2 * ch) * 16 + 4 * q);
          p.w1 = *reinterpret_cast<const f32x4*>(a.head + (2 * ch + 1) * 16 + 4 * q);
        }
        return p;
      };
      Act<PREC, 16> ho;
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1, const TrainPre& p) {
        if (s == 8) {
          acc0 += p.w0 * sb3;  // hbar_7 = Wf^T fbar + sbar w_s / 3
          acc1 += p.w1 * sb3;
        }
        if (s == 4 && ch >= 6) {
          // adjoint of the embedding through the skip connection (inputs 217..255 of L4)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if (2 * ch >= 13) skip[(2 * ch - 13) * 4 + r] = acc0[r];
            skip[(2 * ch + 1 - 13) * 4 + r] = acc1[r];
          }
        }
        const f32x4 z0 = p.s0 * acc0 + p.t0 * S, z1 = p.s1 * acc1 + p.t1 * S;
        if (tile_ok) {
          if (PREC == 1 && a.zbar16 && lz >= 1) {
            st_stream(half_ptr<true>(a.zbar16, lz, a.npts, row, ch, q), pack_half8(z0, z1));
          } else {
            st_stream(reinterpret_cast<f32x4*>(arr_ptr<ARR_ZBAR, PREC == 1>(a.zbar, lz, a.npts, row, 2 * ch, q)), z0 * IS);
            st_stream(reinterpret_cast<f32x4*>(arr_ptr<ARR_ZBAR, PREC == 1>(a.zbar, lz, a.npts, row, 2 * ch + 1, q)), z1 * IS);
          }
        }
        ho.set_chunk(ch, z0, z1);
      };
      const float* wcur = (s == 8) ? a.wt_feat : a.w + sdf_off_R(s);
      const float* wnxt = (s == 8) ? a.w + sdf_off_R(7) : (s > 1 ? a.w + sdf_off_R(s - 1) : a.w + SDF_OFF_R0);
      run_stage<PREC, 16, 8, false, true>(wcur, wnxt, 32, smem, par, h, nullptr, pre, epi, wave, lane);
      h = ho;
    }

    // R0: adjoint of the 39 embedding entries, then through the encoding (same tail as the forward's gradient)
    float ge[16];
    auto pre0 = [&](int) { return 0; };
    auto epi0 = [&](int ch, f32x4 acc0, f32x4 acc1, int) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { ge[ch * 8 + r] = acc0[r]; ge[ch * 8 + 4 + r] = acc1[r]; }
    };
    run_stage<PREC, 16, 2, false>(a.w + SDF_OFF_R0, a.wt_feat, 32, smem, par, h, nullptr, pre0, epi0, wave, lane);

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
            if (es >= 0 && es < 39) dx[nerf_enc_dim<3, 6>(es)] += (q == qq) ? skip[b * 4 + r] * dc[es] : 0.0f;
          }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      dx[c] += __shfl_xor(dx[c], 16, 64);
      dx[c] += __shfl_xor(dx[c], 32, 64);
    }
    if (tile_ok && q == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) a.pbar[row * 3 + c] = dx[c] * 3.0f * IS;
    }
  }
}

}  // namespace nrh
