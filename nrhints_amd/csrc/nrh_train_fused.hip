// Small per-ray / per-point kernels of the fused training step (nrhints_amd/train_fused.py): what round 2 left to ~300 torch
// launches per step - the compositing sum, the loss and its adjoint seeds, the positional encoding as a GEMM operand, the
// scalar reductions - as a handful of HIP launches.  Reference: models/neus_hint_model.py:635-637 (composite),
// pipelines/base_pipeline.py:57-62 (L1 colour loss + eikonal loss), :64-69 (s_val, psnr), fields/encodings.py:168-174.
#pragma once
#include "nrh_common.h"

namespace nrh {

// ---- e = enc_6(3 p) as rows [P][64] (39 used, the rest zero): the B operand of layer 0's weight gradient ----------------
struct EmbRowsArgs {
  const float* ro; const float* rd; const float* t;   // p = ro[ray] + rd[ray] * t[ray * t_stride + j]
  float* out;                                          // [npts][64]
  long long npts;
  int n_per_ray, t_stride;
};
__global__ __launch_bounds__(256) void emb_rows_kernel(const EmbRowsArgs a) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long P = i >> 6;
  const int e = (int)(i & 63);
  if (P >= a.npts) return;
  const long long ray = P / a.n_per_ray;
  const int jj = (int)(P - ray * a.n_per_ray);
  const float tt = a.t[ray * a.t_stride + jj];
  float v = 0.0f;
  if (e < 39) {
    const int d = e < 3 ? e : ((e - 3) % 18) / 6;
    const float x = (a.ro[ray * 3 + d] + a.rd[ray * 3 + d] * tt) * 3.0f;
    if (e < 3) v = x;
    else {
      const int idx = (e - 3) % 18;
      v = sin_cw(x * (float)(1 << (idx % 6)) + ((e - 3) >= 18 ? NRH_HALF_PI : 0.0f));
    }
  }
  a.out[i] = v;
}

// ---- adjoints of the rays (pose / light refinement: nr-hints-cam-opt, register_view) ------------------------------------------
// What autograd derives for ray_bundle.origins / directions / pl_positions in the reference (the samplers, depth / hit point and
// both hints are outside its graph: models/neus_hint_model.py:697, :531, :379, :589), collected from the adjoints the sweeps left:
//   pbar_j = pbar_sdf_j                                  value path of the SDF network through the encoding (sdf_adjoint_kernel)
//          + 9 gbar_j[dim] sum_e ge_j[e] enc''_e(3 p_j)  the spatial gradient's own dependence on the point (second derivative of
//                                                        the positional encoding; ge = d sdf / d embedding via layer 0 and the skip)
//          + mbar_j[0:3]                                 the reflectance net's point input
//   obar = sum_j pbar_j;   dbar = sum_j mid_j pbar_j + rd_bar (alpha stage: cos = <d, g>) + enc4'(d)^T sum_j mbar_j[6:33]
//   plbar = enc4'(pl)^T sum_j mbar_j[33:60]              (the reflectance net sees enc4 of the view direction and the light position)
// One wavefront per ray; 128 samples = 2 per lane for the per-sample part, lanes 0..53 = the 54 encoding columns for the column sums.
struct RayAdjArgs {
  const float* ro; const float* rd; const float* pl;   // [N,3]
  const float* mid;       // [N,128] section mid-points
  const float* pbar;      // [N*128,3]
  const float* gbar;      // [N*128,3]  adjoint of d sdf / dx (after the alpha adjoint incl. the eikonal seed)
  const float* ge;        // [N*128,128] save_ge: columns e and 73 + e
  const float* mbar;      // [N*128,mw]
  const float* rd_bar;    // [N,3]
  float* obar; float* dbar; float* plbar;   // [N,3]
  int mw;
  int nrays;
};
__global__ __launch_bounds__(256) void ray_adjoint_kernel(const RayAdjArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long ray = (long long)blockIdx.x * 4 + wave;
  if (ray >= a.nrays) return;
  float o[3], d[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) { o[c] = a.ro[ray * 3 + c]; d[c] = a.rd[ray * 3 + c]; }
  float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};     // obar, sum mid pbar
#pragma unroll
  for (int e2 = 0; e2 < 2; ++e2) {
    const long long P = ray * 128 + lane + 64 * e2;
    const float t = a.mid[P];
    const float* ge = a.ge + P * 128;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float x3 = (o[c] + d[c] * t) * 3.0f;
      float s2 = 0.0f;     // sum_e ge[e] enc''_e over the 12 sine entries of coordinate c (the raw entry has no second derivative)
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const float f = (float)(1 << k), arg = x3 * f;
        const float g0 = ge[3 + 6 * c + k] + ge[73 + 3 + 6 * c + k], g1 = ge[21 + 6 * c + k] + ge[73 + 21 + 6 * c + k];
        s2 -= (g0 * sin_cw(arg) + g1 * sin_cw(arg + NRH_HALF_PI)) * (f * f);
      }
      const float pb = a.pbar[P * 3 + c] + 9.0f * a.gbar[P * 3 + c] * s2 + a.mbar[P * a.mw + c];
      acc[c] += pb;
      acc[3 + c] += t * pb;
    }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc[i] += __shfl_xor(acc[i], off, 64);
  }
  // column sums of the per-ray encodings' adjoints: lane L < 54 owns column 6 + L (27 of the view direction, 27 of the light)
  float cs = 0.0f;
  if (lane < 54) {
    const float* mb = a.mbar + ray * 128 * a.mw + 6 + lane;
    for (int j = 0; j < 128; ++j) cs += mb[(long long)j * a.mw];
  }
  // enc4(v) = [v (3) | sin(v_c 2^k), c-major (12) | sin(v_c 2^k + pi/2) (12)]: column m of it differentiates against v[dim(m)]
  const int m = lane < 27 ? lane : lane - 27;
  const int dim = m < 3 ? m : ((m - 3) % 12) / 4;
  float coef = 0.0f;
  if (lane < 54) {
    const float v = lane < 27 ? d[dim] : a.pl[ray * 3 + dim];
    if (m < 3) coef = 1.0f;
    else {
      const int k = (m - 3) % 4;
      const float f = (float)(1 << k);
      coef = cos_cw(v * f + ((m - 3) >= 12 ? NRH_HALF_PI : 0.0f)) * f;
    }
  }
  const float contrib = cs * coef;
  float vs[6];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    vs[c] = (lane < 27 && dim == c) ? contrib : 0.0f;
    vs[3 + c] = (lane >= 27 && lane < 54 && dim == c) ? contrib : 0.0f;
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) vs[i] += __shfl_xor(vs[i], off, 64);
  }
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      a.obar[ray * 3 + c] = acc[c];
      a.dbar[ray * 3 + c] = acc[3 + c] + a.rd_bar[ray * 3 + c] + vs[c];
      a.plbar[ray * 3 + c] = vs[3 + c];
    }
  }
}

// ---- composite + loss terms + the adjoint seeds of both (one wavefront per ray) ----------------------------------------------
//   rgb = sum_j c_j w_j + bg (1 - sum_j w_j)                                     (models/neus_hint_model.py:635-637)
//   rgb_loss = sum |rgb - gt| / (N + 1e-5);  eik = sum inside (|g| - 1)^2 / (sum inside + 1e-5);  loss = rgb_loss + igr * eik
// d loss / d rgb = sign(rgb - gt) / (N + 1e-5) is known without any downstream information, so the same pass writes
//   cbar_j = w_j dl/drgb  ->  zbar4 = cbar c (1 - c)   (through the reflectance net's output sigmoid)
//   wbar_j = sum_c (c_jc - bg_c) dl/drgb_c
// and the per-ray partial sums of the four scalars; the eikonal seed needs sum(inside) first and is added by the alpha adjoint.
struct CompositeLossArgs {
  const float* color;    // [N*128,3] sigmoid outputs
  const float* weights;  // [N,128]
  const float* gt;       // [N,3]
  const float* bg;       // [3] or null
  const float* grad;     // [N*128,3] analytic normals
  const float* inside;   // [N,128]
  float* rgb;            // [N,3]
  float* zbar4;          // [N*128,3]
  float* wbar;           // [N,128]
  float* partial;        // [N,4]: sum_c |rgb - gt|, sum_j inside (|g| - 1)^2, sum_j inside, sum_c (rgb - gt)^2
  float inv_n;           // 1 / (N + 1e-5)
  int nrays;
};
__global__ __launch_bounds__(256) void composite_loss_kernel(const CompositeLossArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ray_raw = blockIdx.x * 4 + wave;
  const bool active = ray_raw < a.nrays;
  const long long ray = active ? ray_raw : a.nrays - 1;
  float c[2][3], w[2], acc[3] = {0.f, 0.f, 0.f}, ws = 0.0f, eik = 0.0f, cnt = 0.0f;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const long long P = ray * 128 + lane + 64 * e;
    w[e] = a.weights[P];
    ws += w[e];
#pragma unroll
    for (int k = 0; k < 3; ++k) { c[e][k] = a.color[P * 3 + k]; acc[k] += c[e][k] * w[e]; }
    const float gx = a.grad[P * 3 + 0], gy = a.grad[P * 3 + 1], gz = a.grad[P * 3 + 2];
    const float ins = a.inside[P];
    const float ge = sqrtf(gx * gx + gy * gy + gz * gz) - 1.0f;
    eik += ins * (ge * ge);
    cnt += ins;
  }
  ws = wave_sum(ws); eik = wave_sum(eik); cnt = wave_sum(cnt);
  float dl[3], l1 = 0.0f, l2 = 0.0f, bgc[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    bgc[k] = a.bg ? a.bg[k] : 0.0f;
    const float rgb = wave_sum(acc[k]) + (a.bg ? bgc[k] * (1.0f - ws) : 0.0f);
    const float diff = rgb - a.gt[ray * 3 + k];
    l1 += fabsf(diff);
    l2 += diff * diff;
    dl[k] = (diff > 0.0f ? 1.0f : (diff < 0.0f ? -1.0f : 0.0f)) * a.inv_n;     // torch's abs backward: sign(x), sign(0) = 0
    if (active && lane == 0) a.rgb[ray * 3 + k] = rgb;
  }
  if (!active) return;
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const long long P = ray * 128 + lane + 64 * e;
    float wb = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      wb += (c[e][k] - bgc[k]) * dl[k];
      a.zbar4[P * 3 + k] = (w[e] * dl[k]) * c[e][k] * (1.0f - c[e][k]);
    }
    a.wbar[P] = wb;
  }
  if (lane == 0) {
    a.partial[ray * 4 + 0] = l1; a.partial[ray * 4 + 1] = eik; a.partial[ray * 4 + 2] = cnt; a.partial[ray * 4 + 3] = l2;
  }
}

// one block: per-ray partials -> the loss dict (pipelines/base_pipeline.py:57-69) and the eikonal seed coefficient.
// out[0] loss, [1] rgb_loss, [2] eikonal_loss, [3] s_val = 1 / inv_s, [4] psnr, [5] igr_weight / (sum inside + 1e-5)
struct LossFinishArgs {
  const float* partial;  // [N,4]
  float* out;            // [8]
  const float* dyn;      // optional device [inv_s, cos_anneal]
  float inv_s, igr_weight;
  int nrays;
};
__global__ __launch_bounds__(256) void loss_finish_kernel(const LossFinishArgs a) {
  __shared__ float red[4][4];
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  for (int r = threadIdx.x; r < a.nrays; r += 256)       // fixed order per thread, fixed tree below: deterministic
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] += a.partial[(size_t)r * 4 + k];
#pragma unroll
  for (int k = 0; k < 4; ++k) s[k] = wave_sum(s[k]);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < 4; ++k) red[wave][k] = s[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    float t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
    const float n = (float)a.nrays;
    const float rgb_loss = t[0] / (n + 1e-5f);
    const float eik = t[1] / (t[2] + 1e-5f);
    const float S = a.dyn ? a.dyn[0] : a.inv_s;
    a.out[0] = rgb_loss + eik * a.igr_weight;
    a.out[1] = rgb_loss;
    a.out[2] = eik;
    a.out[3] = 1.0f / S;
    a.out[4] = 10.0f * log10f(1.0f / (t[3] / (3.0f * n)));
    a.out[5] = a.igr_weight / (t[2] + 1e-5f);
    a.out[6] = 0.0f; a.out[7] = 0.0f;
  }
}

// d loss / d variance = sum_rays invs_bar * d inv_s / d variance, inv_s = clip(exp(10 variance), 1e-6, 1e6)
// (models/neus_hint_model.py:104-110, :337): 10 inv_s inside the clip range, 0 outside.  One block.
// ---- the step's adjoint scale from the range of the seeds ------------------------------------------------------------------------
// The f16x3 sweeps of the SDF net are linear in their seeds (sbar, gbar, fbar) and run on S x the seeds, S a power of two
// (SdfTrainArgs.adj_scale).  With 16-bit hand-offs to nrh_dw_gemm the chain's S-scaled adjoints are also what is STORED (fp16:
// abar, zbar), so S has to follow the data: over scenes, anneal steps and batch sizes the seeds' maximum times the ray count spans
// 2^-8 .. 2^8 (events: one sample next to the surface at inv_s ~ 1000; profiles/r05/dw16_ranges.log), the ratio of the stored
// arrays' maxima to the seeds' maximum only 2^-7 .. 2^-2.  S = 2^(4 - ceil(log2 max|seed|)) puts the seeds' maximum in [8, 16]:
// stored maxima around 2^-3 .. 2^2 with 2^14 of head-room to fp16's largest number and 2^10 .. 2^15 above the level (2^-13)
// where the format's absolute floor would start to cost accuracy relative to the array's largest entries.
// dyn = {S, 1 / S, max bits (uint), block counter}: the last block to arrive writes S and resets the two work words - no memset
// between steps, so the kernel is a plain node of the captured step; max is order-independent: deterministic.
struct AdjRangeArgs {
  const float* sbar;     // [npts]
  const float* gbar;     // [npts][3]
  const float* fbar;     // [npts][256] (every 8th row is looked at: its share of the range is small and it is 64x the bytes)
  float* dyn;            // [4]
  long long npts;
};
__global__ __launch_bounds__(256) void adjoint_range_kernel(const AdjRangeArgs a) {
  const long long tid = (long long)blockIdx.x * 256 + threadIdx.x, nth = (long long)gridDim.x * 256;
  // |x| as its bit pattern: non-negative floats order like their bits, and a NaN or inf seed (bits >= 0x7f800000) wins the maximum
  // instead of being dropped by fmaxf - it then fails the range test below (S = 1) and propagates through the chain into the
  // gradients, which is the loud outcome.  EVERY row of fbar is scanned (134 MB at 1 024 rays, ~30 us): an outlier
  // in an unsampled row would have over-estimated S against fp16's 65 504.
  unsigned int mb = 0u;
  auto take = [&](float x) { const unsigned int b = __float_as_uint(x) & 0x7fffffffu; mb = b > mb ? b : mb; };
  for (long long i = tid; i < a.npts; i += nth) take(a.sbar[i]);
  for (long long i = tid; i < a.npts * 3; i += nth) take(a.gbar[i]);
  const long long nf = a.npts * 64;                     // float4 words of fbar [npts, 256]
  for (long long i = tid; i < nf; i += nth) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(a.fbar + i * 4);
    take(v[0]); take(v[1]); take(v[2]); take(v[3]);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { const unsigned int t = (unsigned int)__shfl_xor((int)mb, o); mb = t > mb ? t : mb; }
  const float m = __uint_as_float(mb);
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  unsigned int* const w = reinterpret_cast<unsigned int*>(a.dyn);
  if (threadIdx.x == 0) {
    // one pair of atomics per block (per wave they serialised to 50 us on 2 048 waves); bit patterns again (NaN / inf on top)
    unsigned int bm = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const unsigned int b = __float_as_uint(wm[k]); bm = b > bm ? b : bm; }
    atomicMax(w + 2, bm);
    __threadfence();
    if (atomicAdd(w + 3, 1u) == gridDim.x - 1) {
      const float mx = __uint_as_float(atomicMax(w + 2, 0u));
      float S = 1.0f;
      if (mx > 0.0f && mx < 3.0e38f) {
        int e;
        (void)frexpf(mx, &e);                              // mx = f 2^e, f in [0.5, 1): ceil(log2 mx) <= e
        e = 4 - e;
        e = e < -60 ? -60 : (e > 60 ? 60 : e);
        S = ldexpf(1.0f, e);
      }
      a.dyn[0] = S;
      a.dyn[1] = 1.0f / S;
      w[2] = 0u;
      w[3] = 0u;
    }
  }
}

struct VarGradArgs {
  const float* invs_bar;  // [N]
  const float* dyn;
  float inv_s;
  float* out;             // [1]
  int nrays;
};
__global__ __launch_bounds__(256) void variance_grad_kernel(const VarGradArgs a) {
  __shared__ float red[4];
  float s = 0.0f;
  for (int r = threadIdx.x; r < a.nrays; r += 256) s += a.invs_bar[r];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float S = a.dyn ? a.dyn[0] : a.inv_s;
    const float tot = (red[0] + red[1]) + (red[2] + red[3]);
    a.out[0] = (S > 1e-6f && S < 1e6f) ? tot * 10.0f * S : 0.0f;
  }
}

// The handful of scalars a training step needs on the device, in ONE launch instead of one torch fill / pointwise kernel each
// (a 64-ray step is ~50 launches of 5-20 us: every removed launch is 0.5 % of it): up to four host floats written to four device
// addresses (cos-anneal ratio, learning rates), and 1 / s = clip(exp(10 variance), 1e-6, 1e6) (models/neus_hint_model.py:104-110,
// :337-338) from the variance parameter.  NaN passes through the clip as in torch.clip.
struct StepScalarsArgs {
  float* dst[4];
  float val[4];
  int n;
  const float* variance;   // [1] or null
  float* inv_s_out;        // [1]
};
__global__ __launch_bounds__(64) void step_scalars_kernel(const StepScalarsArgs a) {
  const int t = threadIdx.x;
  if (t < a.n && a.dst[t]) *a.dst[t] = a.val[t];
  if (t == 0 && a.variance) {
    const float s = expf(a.variance[0] * 10.0f);
    *a.inv_s_out = (s < 1e-6f) ? 1e-6f : ((s > 1e6f) ? 1e6f : s);
  }
}

}  // namespace nrh
