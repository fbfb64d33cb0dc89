// The per-ray step of the hierarchical sampler as device functions: merge the samples of the previous up-sample, up-sample the
// next ones, finalise the sections (reference: cat_z_vals / up_sample / sample_pdf, models/neus_hint_model.py:317-331, :270-315,
// :21-65).  One wave = one ray, a 128-long sequence two elements per lane.  Two callers: sampler_step_kernel (nrh_rays.hip: four
// rays per block) and the tail of sdf_split_kernel (nrh_sdf_split.hip: the SDF pass of a small training batch and the step that
// consumes it in ONE launch).
#pragma once
#include "nrh_common.h"

namespace nrh {

// -------------------------------------------------------------------------------------------------
// one launch = [merge the 16 samples of the previous step] + [up-sample 16 new ones | finalise sections]
// -------------------------------------------------------------------------------------------------
struct StepArgs {
  const float* ro;        // [N,3]
  const float* rd;        // [N,3]
  float* z;               // [N,128] sorted ray parameters (n valid)
  float* s;               // [N,128] sdf at those parameters
  const float* znew_in;   // [N,16] samples to merge
  const float* snew_in;   // [N,16] their sdf (merge_sdf)
  float* znew_out;        // [N,16] new samples
  const float* lin16;     // torch.linspace(0,1,16)
  const float* last_dist_ray;  // [N] per-ray last section length, or null -> last_dist
  float* tmid;            // [N,128] section mid-points (finalize)
  float* dists;           // [N,128] section lengths (finalize)
  float inv_s;
  float last_dist;
  int nrays;
  int n;                  // valid entries before the merge
  int do_merge, merge_sdf, do_upsample, do_finalize;
  int then_finalize;      // after this phase (an up-sample): merge the samples just drawn, without sdf, and finalise - the last two
                          // launches of a sampler in one
  int n_new;              // samples per up-sampling step (n_importance_samples / up_sample_steps; 16 by default, at most 16): what is
                          // merged and what is drawn; lin16 holds linspace(0, 1, n_new).  A finalised ray with fewer than 128 samples
                          // is padded as finalize64_kernel pads: the last mid-point repeated, length 0
};

// Ordering of a wave's OWN LDS traffic between the phases of a step.  BLOCK: __syncthreads (the stand-alone kernel, where all four
// waves of a block run the same phases); otherwise wave-level only (the fused tail of sdf_split_kernel, where only some waves of
// the workgroup run a step): a wave's LDS instructions execute in issue order, so a scheduling barrier is all that is needed.
template <bool BLOCK>
__device__ __forceinline__ void step_sync() {
  if constexpr (BLOCK) {
    __syncthreads();
  } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// the ray's sorted samples two per lane (j0 = lane, j1 = lane + 64) and their count
struct RayState {
  float z0, z1, s0, s1;
  int n;
};

// ONE phase of the per-ray step for one wave = one ray: [merge the n_new samples (zn, sn in lanes < n_new)] + [up-sample n_new new
// ones -> zn_out (and a.znew_out) | finalise sections].  Z, S, X, C: 144-float LDS rows private to the wave.  The arithmetic is
// the reference's cat_z_vals / up_sample / sample_pdf (models/neus_hint_model.py:317-331, :270-315, :21-65) statement by statement.
template <bool BLOCK>
__device__ __forceinline__ void sampler_step_phase(const StepArgs& a, long long ray, bool active, float* Z, float* S, float* X, float* C,
                                                   RayState& st, int do_merge, int merge_sdf, int do_upsample, int do_finalize,
                                                   float inv_s, float zn, float sn, float& zn_out) {
  const int lane = threadIdx.x & 63;
  const int j0 = lane, j1 = lane + 64;
  float z0 = st.z0, z1 = st.z1, s0 = st.s0, s1 = st.s1;
  int n = st.n;
  const int nn = a.n_new;
  if (do_merge) {
    // stable merge of two sorted lists by rank counting (old entries win ties, as a stable sort of cat[z, z_new])
    int c0 = 0, c1 = 0, cn = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (k < nn) {       // (wave-uniform; the loop stays unrolled)
        const float zk = __shfl(zn, k, 64);
        c0 += (zk < z0) ? 1 : 0;
        c1 += (zk < z1) ? 1 : 0;
        const unsigned long long b0 = __ballot(j0 < n && z0 <= zk);
        const unsigned long long b1 = __ballot(j1 < n && z1 <= zk);
        const int cnt = __popcll(b0) + __popcll(b1);
        cn = (lane == k) ? cnt : cn;
      }
    }
    if (j0 < n) { Z[j0 + c0] = z0; S[j0 + c0] = s0; }
    if (j1 < n) { Z[j1 + c1] = z1; S[j1 + c1] = s1; }
    if (lane < nn) { Z[lane + cn] = zn; S[lane + cn] = merge_sdf ? sn : 0.0f; }
    step_sync<BLOCK>();
    n += nn;
    z0 = (j0 < n) ? Z[j0] : 0.0f;
    z1 = (j1 < n) ? Z[j1] : 0.0f;
    s0 = (j0 < n) ? S[j0] : 0.0f;
    s1 = (j1 < n) ? S[j1] : 0.0f;
    if (active) {
      if (j0 < n) { a.z[ray * 128 + j0] = z0; if (merge_sdf) a.s[ray * 128 + j0] = s0; }
      if (j1 < n) { a.z[ray * 128 + j1] = z1; if (merge_sdf) a.s[ray * 128 + j1] = s1; }
    }
  } else {
    if (j0 < n) { Z[j0] = z0; S[j0] = s0; }
    if (j1 < n) { Z[j1] = z1; S[j1] = s1; }
    step_sync<BLOCK>();
  }
  st.z0 = z0; st.z1 = z1; st.s0 = s0; st.s1 = s1; st.n = n;

  if (do_upsample) {
    const float ox = a.ro[ray * 3 + 0], oy = a.ro[ray * 3 + 1], oz = a.ro[ray * 3 + 2];
    const float dx = a.rd[ray * 3 + 0], dy = a.rd[ray * 3 + 1], dz = a.rd[ray * 3 + 2];
    auto radius = [&](float z) {
      const float px = ox + dx * z, py = oy + dy * z, pz = oz + dz * z;
      return sqrtf(px * px + py * py + pz * pz);
    };
    const float r0 = radius(z0), r1 = radius(z1);
    if (j0 < n) X[j0] = r0;
    if (j1 < n) X[j1] = r1;
    step_sync<BLOCK>();
    const bool v0 = j0 < n - 1, v1 = j1 < n - 1;  // section j = [z_j, z_{j+1}]
    const float zn0 = v0 ? Z[j0 + 1] : z0, zn1 = v1 ? Z[j1 + 1] : z1;
    const float sn0 = v0 ? S[j0 + 1] : s0, sn1 = v1 ? S[j1 + 1] : s1;
    const float rn0 = v0 ? X[j0 + 1] : r0, rn1 = v1 ? X[j1 + 1] : r1;
    const float cos0 = (sn0 - s0) / (zn0 - z0 + 1e-5f);
    const float cos1 = (sn1 - s1) / (zn1 - z1 + 1e-5f);
    C[j0] = cos0;
    C[j1] = cos1;
    step_sync<BLOCK>();
    const float pc0 = (j0 == 0) ? 0.0f : C[j0 - 1];
    const float pc1 = C[j1 - 1];
    auto section_alpha = [&](float s_a, float s_b, float dist, float cosv, float pcos, float ra, float rb) {
      const float inside = ((ra < 1.0f) || (rb < 1.0f)) ? 1.0f : 0.0f;
      float c = fminf(pcos, cosv);
      c = fminf(fmaxf(c, -1e3f), 0.0f) * inside;
      const float mid = (s_a + s_b) * 0.5f;
      const float pe = mid - c * dist * 0.5f;
      const float ne = mid + c * dist * 0.5f;
      const float pcdf = sigmoidf_(pe * inv_s);
      const float ncdf = sigmoidf_(ne * inv_s);
      return (pcdf - ncdf + 1e-5f) / (pcdf + 1e-5f);
    };
    const float al0 = section_alpha(s0, sn0, zn0 - z0, cos0, pc0, r0, rn0);
    const float al1 = section_alpha(s1, sn1, zn1 - z1, cos1, pc1, r1, rn1);
    float T0, T1;
    excl_prod_128(v0 ? (1.0f - al0 + 1e-7f) : 1.0f, v1 ? (1.0f - al1 + 1e-7f) : 1.0f, T0, T1);
    const float w0 = v0 ? (al0 * T0 + 1e-5f) : 0.0f;  // weights + 1e-5 (sample_pdf)
    const float w1 = v1 ? (al1 * T1 + 1e-5f) : 0.0f;
    const float tot = wave_sum(w0 + w1);
    float cs0, cs1;
    incl_sum_128(w0 / tot, w1 / tot, cs0, cs1);
    step_sync<BLOCK>();  // everyone done with X (radius) before it becomes the cdf
    if (lane == 0) X[0] = 0.0f;
    if (v0) X[j0 + 1] = cs0;
    if (v1) X[j1 + 1] = cs1;
    step_sync<BLOCK>();
    const float cd0 = (j0 < n) ? X[j0] : 2.0f;
    const float cd1 = (j1 < n) ? X[j1] : 2.0f;
    int ind = 0;
    float u = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (k < nn) {
        const float uk = a.lin16[k];
        const unsigned long long b0 = __ballot(j0 < n && cd0 <= uk);  // searchsorted(right=True)
        const unsigned long long b1 = __ballot(j1 < n && cd1 <= uk);
        const int cnt = __popcll(b0) + __popcll(b1);
        if (lane == k) { ind = cnt; u = uk; }
      }
    }
    zn_out = 0.0f;
    if (lane < nn) {
      const int below = max(ind - 1, 0), above = min(ind, n - 1);
      const float cb = X[below], ca = X[above];
      const float bb = Z[below], ba = Z[above];
      float den = ca - cb;
      den = (den < 1e-5f) ? 1.0f : den;
      const float t = (u - cb) / den;
      zn_out = bb + t * (ba - bb);
      if (active) a.znew_out[ray * 16 + lane] = zn_out;
    }
    step_sync<BLOCK>();   // (a following phase rewrites Z / S / X)
  }

  if (do_finalize) {
    // section lengths and mid-points of the final n samples (128 by default; fewer: padded, see StepArgs.n_new)
    const float last = a.last_dist_ray ? a.last_dist_ray[ray] : a.last_dist;
    const int r0 = min(j0, n - 1), r1 = min(j1, n - 1);
    const float y0 = Z[r0], y1 = Z[r1];
    const float d0 = (r0 < n - 1) ? (Z[r0 + 1] - y0) : last;
    const float d1 = (r1 < n - 1) ? (Z[r1 + 1] - y1) : last;
    if (active) {
      a.dists[ray * 128 + j0] = (j0 < n) ? d0 : 0.0f;
      a.dists[ray * 128 + j1] = (j1 < n) ? d1 : 0.0f;
      a.tmid[ray * 128 + j0] = y0 + d0 * 0.5f;
      a.tmid[ray * 128 + j1] = y1 + d1 * 0.5f;
    }
  }
}

// The phases one launch runs for a ray (StepArgs): the phase described by do_merge / merge_sdf / do_upsample / do_finalize at inv_s,
// then - with then_finalize - the LAST phase of a sampler right behind it: merge the samples just drawn (without sdf, :328-329) and
// finalise.  (The reference's last up-sample is followed by no SDF pass, so its two step launches were always back to back.)
template <bool BLOCK>
__device__ __forceinline__ void sampler_step_phases(const StepArgs& a, long long ray, bool active, float* Z, float* S, float* X, float* C,
                                                    RayState& st, float zn, float sn) {
  float zn_out = 0.0f;
  sampler_step_phase<BLOCK>(a, ray, active, Z, S, X, C, st, a.do_merge, a.merge_sdf, a.do_upsample, a.do_finalize, a.inv_s, zn, sn, zn_out);
  if (a.then_finalize) {
    float unused = 0.0f;
    sampler_step_phase<BLOCK>(a, ray, active, Z, S, X, C, st, 1, 0, 0, 1, 0.0f, zn_out, 0.0f, unused);
  }
}


}  // namespace nrh
