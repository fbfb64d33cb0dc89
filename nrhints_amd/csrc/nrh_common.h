// Shared device helpers for the NRHints MI355X (gfx950 / CDNA4) hot path.
// Wave = 64 lanes everywhere; no CUDA compatibility layer, no dual paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NRH_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- error codes of the C ABI (include/nrhints_hip.h) ----
#define NRH_OK 0
#define NRH_E_INVALID (-1)
#define NRH_E_LAUNCH (-2)
#define NRH_E_WORKSPACE (-3)
#define NRH_E_UNSUPPORTED (-4)

namespace nrh {

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// ---- wave-wide reductions / scans (64 lanes, DPP/ds_bpermute via __shfl) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// inclusive scan over the 64 lanes, op = + or *
__device__ __forceinline__ float wave_scan_add(float v) {
  const int l = lane_id();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float t = __shfl_up(v, o, 64);
    if (l >= o) v += t;
  }
  return v;
}
__device__ __forceinline__ float wave_scan_mul(float v) {
  const int l = lane_id();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    float t = __shfl_up(v, o, 64);
    if (l >= o) v *= t;
  }
  return v;
}

// Exclusive running product over a 128-long per-ray sequence held two per lane:
// element j0 = lane (first half), j1 = lane + 64 (second half).
// in: f0, f1 = factors; out: e0, e1 = prod_{k<j} f_k.
__device__ __forceinline__ void excl_prod_128(float f0, float f1, float& e0, float& e1) {
  const int l = lane_id();
  float i0 = wave_scan_mul(f0);
  float tot0 = __shfl(i0, 63, 64);
  float i1 = wave_scan_mul(f1) * tot0;
  float p0 = __shfl_up(i0, 1, 64);
  float p1 = __shfl_up(i1, 1, 64);
  e0 = (l == 0) ? 1.0f : p0;
  e1 = (l == 0) ? tot0 : p1;
}
// Inclusive running sum, same layout.
__device__ __forceinline__ void incl_sum_128(float f0, float f1, float& s0, float& s1) {
  s0 = wave_scan_add(f0);
  float tot0 = __shfl(s0, 63, 64);
  s1 = wave_scan_add(f1) + tot0;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// nn.Softplus(beta=100, threshold=20) and its derivative sigmoid(100 z)
// (reference: fields/sdf_field.py:104; aten softplus / softplus_backward), four adjacent registers at a time so
// that the mul/add parts compile to v_pk_*_f32 without register shuffles.
// Built on the hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32, <= 1 ulp each): the libm
// expf/log1pf pair costs ~300 VALU instructions per activation.  log2(1+e) instead of log1p(e) costs at most
// 6e-8 * ln2/100 = 4e-10 ABSOLUTE error in h - below half an ulp of the O(1e-2..1) activations it feeds.
// The "linear above 20" switch is taken on t*log2(e) > 20*log2(e); at the switch both branches agree to 2e-11.
#ifndef NRH_SOFTPLUS_V2
#define NRH_SOFTPLUS_V2 1     // branch-free form  max(z,0) + log2(1 + 2^-|t|) ln2/100  (no compare/select on the value path)
#endif
template <bool WANT_D>
__device__ __forceinline__ void softplus100_4(const f32x4 z, f32x4& h, f32x4& d) {
  const f32x4 t = z * 144.26950408889634074f;  // 100 * log2(e)
  f32x4 e, l;
#if NRH_SOFTPLUS_V2
  // softplus(z) = max(z, 0) + log1p(exp(-|100 z|)) / 100: mathematically the reference's log1p(exp(100 z))/100 below its
  // threshold and z (+ < 2e-11) above it, with no overflow for any z, so the "linear above 20" switch
  // (fields/sdf_field.py:104, threshold 20) needs no compare/select.  sigma' = (z >= 0 ? 1 : e) / (1 + e).
#pragma unroll
  for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(-__builtin_fabsf(t[r]));
  const f32x4 ope = e + 1.0f;
#pragma unroll
  for (int r = 0; r < 4; ++r) l[r] = __builtin_amdgcn_logf(ope[r]);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    h[r] = __builtin_fmaf(l[r], 6.9314718055994530942e-3f, __builtin_fmaxf(z[r], 0.0f));
    if (WANT_D) d[r] = (z[r] >= 0.0f ? 1.0f : e[r]) * __builtin_amdgcn_rcpf(ope[r]);
  }
#else
#pragma unroll
  for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(t[r]);
  const f32x4 ope = e + 1.0f;
#pragma unroll
  for (int r = 0; r < 4; ++r) l[r] = __builtin_amdgcn_logf(ope[r]);
  l = l * 6.9314718055994530942e-3f;  // ln2 / 100
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const bool lin = t[r] > 28.853900817779268147f;
    h[r] = lin ? z[r] : l[r];
    if (WANT_D) d[r] = lin ? 1.0f : e[r] * __builtin_amdgcn_rcpf(ope[r]);
  }
#endif
}

// sin for |x| up to a few 1e3: 3-term Cody-Waite reduction by pi/2 (fma) + cephes minimax kernels on
// [-pi/4, pi/4]; <= 2 ulp.  libm sinf drags a Payne-Hanek slow path (scratch memory, ~150 instructions
// inlined per call site) into every encoding entry.
__device__ __forceinline__ float sin_cw(float x) {
  const float kf = rintf(x * 0.63661977236758134308f);
  float r = fmaf(-kf, 1.5703125f, x);
  r = fmaf(-kf, 4.837512969970703125e-4f, r);
  r = fmaf(-kf, 7.54978995489188216e-8f, r);
  const int k = (int)kf;
  const float z = r * r;
  const float sp = fmaf(r * z, fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
  const float cp = fmaf(z * z, fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f),
                        fmaf(z, -0.5f, 1.0f));
  const float v = (k & 1) ? cp : sp;
  return (k & 2) ? -v : v;
}
__device__ __forceinline__ float cos_cw(float x) {
  const float kf = rintf(x * 0.63661977236758134308f);
  float r = fmaf(-kf, 1.5703125f, x);
  r = fmaf(-kf, 4.837512969970703125e-4f, r);
  r = fmaf(-kf, 7.54978995489188216e-8f, r);
  const int k = (int)kf + 1;
  const float z = r * r;
  const float sp = fmaf(r * z, fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), r);
  const float cp = fmaf(z * z, fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f),
                        fmaf(z, -0.5f, 1.0f));
  const float v = (k & 1) ? cp : sp;
  return (k & 2) ? -v : v;
}

// NeRFEncoding, include_input=True (reference: fields/encodings.py:168-174):
//   enc_F(x) = [x, sin(x_d 2^k) (d-major, k-minor), sin(x_d 2^k + pi/2)]
// The cosine half is the sine of the float32-rounded sum, exactly as the reference evaluates it.
#define NRH_HALF_PI 1.57079637050628662109375f /* float(pi/2) */

// All D*(2F+1) entries with compile-time indices only (no scratch-resident arrays).
template <int D, int F>
__device__ __forceinline__ void nerf_enc_all(const float (&x)[D], float (&v)[D * (2 * F + 1)]) {
#pragma unroll
  for (int d = 0; d < D; ++d) v[d] = x[d];
#pragma unroll
  for (int d = 0; d < D; ++d)
#pragma unroll
    for (int k = 0; k < F; ++k) {
      const float s = x[d] * (float)(1 << k);
      v[D + d * F + k] = sin_cw(s);
      v[D + D * F + d * F + k] = sin_cw(s + NRH_HALF_PI);
    }
}
// d(entry e)/d(x_{dim(e)}) for every entry (what autograd of the encoding yields).
template <int D, int F>
__device__ __forceinline__ void nerf_enc_dall(const float (&x)[D], float (&c)[D * (2 * F + 1)]) {
#pragma unroll
  for (int d = 0; d < D; ++d) c[d] = 1.0f;
#pragma unroll
  for (int d = 0; d < D; ++d)
#pragma unroll
    for (int k = 0; k < F; ++k) {
      const float fr = (float)(1 << k);
      const float s = x[d] * fr;
      c[D + d * F + k] = cos_cw(s) * fr;
      c[D + D * F + d * F + k] = cos_cw(s + NRH_HALF_PI) * fr;
    }
}
// which input dimension entry e differentiates against
template <int D, int F>
__host__ __device__ constexpr int nerf_enc_dim(int e) {
  return e < D ? e : ((e - D) % (D * F)) / F;
}
// Entry e = base + 4*q of enc_F(x) for THIS lane's q (0 outside [0, D(2F+1))): the four candidates are enumerated at
// compile time, the sine argument is selected, and ONE sin is evaluated - 36 sines per point would otherwise be
// computed three times per tile to use 28 of them.
template <int D, int F>
__device__ __forceinline__ float nerf_enc_entry_q(const float (&x)[D], int base, int q) {
  constexpr int N = D * (2 * F + 1);
  float arg = 0.0f, raw = 0.0f;
  int kind = 0;  // 0 -> zero, 1 -> raw input, 2 -> sine
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {
    const int e = base + 4 * qq;
    if (e < 0 || e >= N) continue;
    const bool mine = (q == qq);
    if (e < D) {
      raw = mine ? x[e] : raw;
      kind = mine ? 1 : kind;
    } else {
      int idx = e - D;
      const float ph = (idx >= D * F) ? NRH_HALF_PI : 0.0f;
      idx = idx % (D * F);
      const int d = idx / F, k = idx % F;
      const float cand = x[d] * (float)(1 << k) + ph;
      arg = mine ? cand : arg;
      kind = mine ? 2 : kind;
    }
  }
  const float s = sin_cw(arg);
  return (kind == 2) ? s : ((kind == 1) ? raw : 0.0f);
}

// select v[base + 4*q] for the lane's q = lane>>4 with compile-time candidates (entries >= N read as 0)
template <int N>
__device__ __forceinline__ float sel_q(const float (&v)[N], int base, int q) {
  float r = 0.0f;
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {
    const int e = base + 4 * qq;
    if (e >= 0 && e < N) r = (q == qq) ? v[e] : r;
  }
  return r;
}

}  // namespace nrh
