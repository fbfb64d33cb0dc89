// Per-ray stages of the NeuS-with-hints renderer on gfx950: one 64-lane wavefront per ray, every 128-sample
// sequence held two elements per lane (j0 = lane, j1 = lane + 64), scans and reductions by wave shuffles.
//
// Restates (reference, models/neus_hint_model.py): coarse z (:673-683), up_sample (:270-315), sample_pdf (:21-65),
// cat_z_vals (:317-331), section mid-points (:491-496 / :416-418), get_alpha's alpha formula (:339-356), weights /
// depth / hit point (:512-533), shadow-ray set-up and transmittance (:380-395, :429-432), hit normal and the
// Cook-Torrance cue (:583-616), composite (:635-637).
#include "nrh_mlp.h"
#include "nrh_step.h"

namespace nrh {

constexpr int RAYS_PER_BLOCK = 4;

// -------------------------------------------------------------------------------------------------
// One iteration of the reference's sphere tracer (models/neus_hint_model.py:359-372) after the SDF has been evaluated at the
// current points: rays with |sdf| < threshold or depth > far stay, the others advance by sdf along the ray.  A ray that has
// stopped never moves again (its point, hence its sdf, no longer changes), so iterating past "all converged" changes nothing
// and the host may test the flag only every few iterations.
struct TraceArgs {
  const float* rd;    // [N,3]
  const float* sdf;   // [N] at pts
  float* pts;         // [N,3] in/out
  float* depth;       // [N] in/out
  int* moved;         // device flag, set to 1 when any ray advanced
  float threshold, far_;
  int nrays;
};

__global__ __launch_bounds__(256) void sphere_trace_step_kernel(const TraceArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  bool adv = false;
  if (i < a.nrays) {
    const float s = a.sdf[i], dep = a.depth[i];
    adv = !(fabsf(s) < a.threshold || dep > a.far_);
    if (adv) {
#pragma unroll
      for (int c = 0; c < 3; ++c) a.pts[i * 3LL + c] = __fadd_rn(a.pts[i * 3LL + c], __fmul_rn(s, a.rd[i * 3LL + c]));  // pts + sdf * d, unfused
      a.depth[i] = __fadd_rn(dep, s);
    }
  }
  if (__any(adv) && lane_id() == 0) atomicOr(a.moved, 1);
}

// -------------------------------------------------------------------------------------------------
// coarse z:  z_j = near + (far - near) * lin64[j]   (+ one jitter per ray in training)
// -------------------------------------------------------------------------------------------------
struct CoarseArgs {
  const float* near_;
  const float* far_;
  const float* lin64;   // torch.linspace(0,1,nc) as float32 (bit-exact table from the host; nc = 64 by default)
  const float* t_rand;  // [N] or null
  float* z;             // [N,128], first nc written
  int nrays;
  int nc;               // renderer.n_samples (models/neus_hint_model.py:673-683)
};
__global__ void coarse_z_kernel(const CoarseArgs a) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)a.nrays * a.nc) return;
  const int ray = (int)(i / a.nc), j = (int)(i - (long long)ray * a.nc);
  const float nr = a.near_[ray], fr = a.far_[ray];
  float z = nr + (fr - nr) * a.lin64[j];
  if (a.t_rand) z = z + (a.t_rand[ray] - 0.5f) * 2.0f / (float)a.nc;
  a.z[(long long)ray * 128 + j] = z;
}

// -------------------------------------------------------------------------------------------------
// renderer.n_importance_samples = 0 (models/neus_hint_model.py:696: no hierarchical sampling): the 64 coarse samples are the
// final ones.  Sections (:491-496) for them; entries 64..127 of the 128-wide per-ray arrays repeat the last mid-point with
// length 0 - the alpha kernels give those padded samples alpha = 0 (CoreArgs.nreal), so they carry no weight anywhere.
// -------------------------------------------------------------------------------------------------
// (nt = the number of coarse samples: 64 by default, renderer.n_samples in general)
__global__ void finalize64_kernel(const float* z, float last_dist, float* tmid, float* dists, int nrays, int nt) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)nrays * 128) return;
  const long long ray = i >> 7;
  const int j = (int)(i & 127), jr = j < nt ? j : nt - 1;
  const float zj = z[ray * 128 + jr];
  const float d = (jr < nt - 1) ? z[ray * 128 + jr + 1] - zj : last_dist;
  tmid[i] = zj + d * 0.5f;
  dists[i] = j < nt ? d : 0.0f;
}

// (StepArgs and the phases of the per-ray step live in nrh_step.h: sdf_split_kernel runs them in its tail as well)
__global__ __launch_bounds__(256) void sampler_step_kernel(const StepArgs a) {
  __shared__ float sz[RAYS_PER_BLOCK][144];
  __shared__ float ss[RAYS_PER_BLOCK][144];
  __shared__ float sx[RAYS_PER_BLOCK][144];  // radius, then cdf
  __shared__ float sc[RAYS_PER_BLOCK][144];  // section cos
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ray_raw = blockIdx.x * RAYS_PER_BLOCK + wave;
  const bool active = ray_raw < a.nrays;
  const long long ray = active ? ray_raw : a.nrays - 1;
  const int n = a.n;
  const int j0 = lane, j1 = lane + 64;
  RayState st;
  st.z0 = (j0 < n) ? a.z[ray * 128 + j0] : 0.0f;
  st.z1 = (j1 < n) ? a.z[ray * 128 + j1] : 0.0f;
  st.s0 = (j0 < n) ? a.s[ray * 128 + j0] : 0.0f;
  st.s1 = (j1 < n) ? a.s[ray * 128 + j1] : 0.0f;
  st.n = n;
  const float zn = (a.do_merge && lane < a.n_new) ? a.znew_in[ray * 16 + lane] : 0.0f;
  const float sn = (a.do_merge && a.merge_sdf && lane < a.n_new) ? a.snew_in[ray * 16 + lane] : 0.0f;
  sampler_step_phases<true>(a, ray, active, sz[wave], ss[wave], sx[wave], sc[wave], st, zn, sn);
}

// -------------------------------------------------------------------------------------------------
// SDF -> alpha -> weights on the primary ray; depth, hit point, hit normal, specular cue; shadow-ray set-up
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ float neus_alpha(float sdf, float gx, float gy, float gz, float dx, float dy, float dz,
                                            float dist, float inv_s, float ca) {
  const float true_cos = dx * gx + dy * gy + dz * gz;
  const float iter_cos = -(fmaxf(-true_cos * 0.5f + 0.5f, 0.0f) * (1.0f - ca) + fmaxf(-true_cos, 0.0f) * ca);
  const float en = sdf + iter_cos * dist * 0.5f;
  const float ep = sdf - iter_cos * dist * 0.5f;
  const float pc = sigmoidf_(ep * inv_s);
  const float nc = sigmoidf_(en * inv_s);
  return fminf(fmaxf((pc - nc + 1e-5f) / (pc + 1e-5f), 0.0f), 1.0f);
}

struct CoreArgs {
  const float* ro;
  const float* rd;
  const float* pl;
  const float* sdf;     // [N,128]
  const float* grad;    // [N*128,3]
  const float* dists;   // [N,128]
  const float* tmid;    // [N,128]
  const float* lin64;
  const float* t_rand_shadow;  // [N,64] or null
  float* weights;       // [N,128]
  float* inside;        // [N,128]
  float* nhat;          // [N*128,3]
  float* depth;         // [N]
  float* wsum;          // [N]
  float* cue;           // [N,4]
  float* cue_b;         // [N,128,4] broadcast copy (RenderOutput.specular_cue) or null
  float* hit;           // [N,3] hit point o + d * depth (unit entry nrh_alpha_composite only) or null
  float* hit_n;         // [N,3] unit hit normal normalize(sum_j n_j w_j) (unit entry only) or null
  float* pts;           // [N*128,3] p = o + d * mid (the reflectance net's point input in a training step) or null
  const float* depth_in;  // [N] DepthComputationType.SphereTracing: depth and hit point come from the tracer (:527-528) ...
  const float* hit_in;    // [N,3] ... instead of the compositing weights; both null otherwise
  float* srd;           // [N,3] shadow ray direction
  float* slast;         // [N] light distance / 64
  float* zs;            // [N,128] coarse shadow z (first 64)
  float inv_s, cos_anneal, shadow_om;   // shadow_om = 1 - renderer.shadow_ray_offset (host double -> float)
  const float* dyn;     // optional device [inv_s, cos_anneal] overriding the two values above (hipGraph-captured training steps)
  // per-roughness constants evaluated in double on the host, as Python does for the reference's scalars
  // (models/neus_hint_model.py:604, 609): k, 1 - k, a^2, a^2 - 1
  float kk[4], omk[4], a2[4], a2m1[4];
  int zero_hints;  // geometry warm-up: cue = 0 (:617-619)
  int depth_max_weight;  // DepthComputationType.MaximalWeightPoint (:534-538) instead of alpha blending
  int nreal;             // samples per ray that exist (0 = all 128; 64 for n_importance_samples = 0): the rest get alpha = 0
  int s_coarse;          // coarse samples of the shadow ray (renderer.n_shadow_samples, :377; 0 = 64, at most 64); lin64 holds
                         // linspace(0, 1, s_coarse) and t_rand_shadow is [N, s_coarse]
  // renderer.use_outside_nerf (:516-519): outside the unit sphere a sample's alpha is the background NeRF's.  bg_alpha [N,160]
  // (the caller's render_outside at the merged positions; the first 128 are this ray's samples), tail_t [N] out = transmittance
  // behind sample 127, from which the caller composites the 32 samples beyond the sphere; both null otherwise
  const float* bg_alpha;
  float* tail_t;
  int nrays;
};

__global__ __launch_bounds__(256) void core_alpha_kernel(const CoreArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ray_raw = blockIdx.x * RAYS_PER_BLOCK + wave;
  const bool active = ray_raw < a.nrays;
  const long long ray = active ? ray_raw : a.nrays - 1;
  const float ox = a.ro[ray * 3 + 0], oy = a.ro[ray * 3 + 1], oz = a.ro[ray * 3 + 2];
  const float dx = a.rd[ray * 3 + 0], dy = a.rd[ray * 3 + 1], dz = a.rd[ray * 3 + 2];
  float al[2], mid[2], nh[2][3], ins[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const long long P = ray * 128 + lane + 64 * e;
    const float gx = a.grad[P * 3 + 0], gy = a.grad[P * 3 + 1], gz = a.grad[P * 3 + 2];
    mid[e] = a.tmid[P];
    al[e] = neus_alpha(a.sdf[P], gx, gy, gz, dx, dy, dz, a.dists[P], a.dyn ? a.dyn[0] : a.inv_s, a.dyn ? a.dyn[1] : a.cos_anneal);
    const float px = ox + dx * mid[e], py = oy + dy * mid[e], pz = oz + dz * mid[e];
    ins[e] = (sqrtf(px * px + py * py + pz * pz) < 1.0f) ? 1.0f : 0.0f;
    if (a.pts && active) { a.pts[P * 3 + 0] = px; a.pts[P * 3 + 1] = py; a.pts[P * 3 + 2] = pz; }   // fl(o + fl(d t)), as the SDF kernels form it
    if (a.nreal && lane + 64 * e >= a.nreal) { al[e] = 0.0f; ins[e] = 0.0f; }   // padded sample: no weight, not counted
    if (a.bg_alpha) al[e] = al[e] * ins[e] + a.bg_alpha[ray * 160 + lane + 64 * e] * (1.0f - ins[e]);
    const float gn = fmaxf(sqrtf(gx * gx + gy * gy + gz * gz), 1e-12f);  // F.normalize eps
    nh[e][0] = gx / gn;
    nh[e][1] = gy / gn;
    nh[e][2] = gz / gn;
  }
  float T0, T1;
  excl_prod_128(1.0f - al[0] + 1e-7f, 1.0f - al[1] + 1e-7f, T0, T1);
  const float w0 = al[0] * T0, w1 = al[1] * T1;
  const float wsum = wave_sum(w0 + w1);
  float depth = wave_sum(mid[0] * w0 + mid[1] * w1);
  if (a.depth_max_weight) {
    // depth = mid_z at argmax_j w_j, first index on ties (torch.argmax)
    float wmax = fmaxf(w0, w1);
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o2, 64));
    int idx = (w0 == wmax) ? lane : ((w1 == wmax) ? lane + 64 : 1 << 20);
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) idx = min(idx, __shfl_xor(idx, o2, 64));
    const float cand = (idx & 64) ? mid[1] : mid[0];
    depth = __shfl(cand, idx & 63, 64);
  }
  const float hnx = wave_sum(nh[0][0] * w0 + nh[1][0] * w1);
  const float hny = wave_sum(nh[0][1] * w0 + nh[1][1] * w1);
  const float hnz = wave_sum(nh[0][2] * w0 + nh[1][2] * w1);

  // ---- per-ray quantities (computed redundantly on all lanes; cheap) ----
  float hx = ox + dx * depth, hy = oy + dy * depth, hz = oz + dz * depth;  // hit point
  if (a.depth_in) {
    depth = a.depth_in[ray];
    hx = a.hit_in[ray * 3 + 0]; hy = a.hit_in[ray * 3 + 1]; hz = a.hit_in[ray * 3 + 2];
  }
  const float plx = a.pl[ray * 3 + 0], ply = a.pl[ray * 3 + 1], plz = a.pl[ray * 3 + 2];
  const float hnn = fmaxf(sqrtf(hnx * hnx + hny * hny + hnz * hnz), 1e-12f);
  const float nx = hnx / hnn, ny = hny / hnn, nz = hnz / hnn;
  // l, v, h
  float lx = plx - hx, ly = ply - hy, lz = plz - hz;
  const float ln = fmaxf(sqrtf(lx * lx + ly * ly + lz * lz), 1e-12f);
  lx /= ln; ly /= ln; lz /= ln;
  float vx = -dx, vy = -dy, vz = -dz;
  const float vn = fmaxf(sqrtf(vx * vx + vy * vy + vz * vz), 1e-12f);
  vx /= vn; vy /= vn; vz /= vn;
  float hvx = lx + vx, hvy = ly + vy, hvz = lz + vz;
  const float hvn = fmaxf(sqrtf(hvx * hvx + hvy * hvy + hvz * hvz), 1e-12f);
  hvx /= hvn; hvy /= hvn; hvz /= hvn;
  auto clip01 = [](float x) { return fminf(fmaxf(x, 0.0f), 1.0f); };
  const float ndl = clip01(nx * lx + ny * ly + nz * lz);
  const float ndv = clip01(nx * vx + ny * vy + nz * vz);
  const float ndh = clip01(nx * hvx + ny * hvy + nz * hvz);
  const float hdv = clip01(hvx * vx + hvy * vy + hvz * vz);
  const float ndh2 = ndh * ndh;
  float cue[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float g1 = ndv / (ndv * a.omk[i] + a.kk[i]);
    const float g2 = ndl / (ndl * a.omk[i] + a.kk[i]);
    const float dn = ndh2 * a.a2m1[i] + 1.0f;
    const float ndf = a.a2[i] / (3.14159265358979323846f * (dn * dn));
    const float om = 1.0f - hdv;
    const float f = 0.04f + 0.96f * (om * om * om * om * om);
    cue[i] = a.zero_hints ? 0.0f : ndf * (g1 * g2) * f / (4.0f * ndv + 1e-3f);
  }
  // shadow ray: light -> hit point (get_visibility :380-386)
  const float svx = hx - plx, svy = hy - ply, svz = hz - plz;
  const float L = sqrtf(svx * svx + svy * svy + svz * svz);

  if (active) {
    a.weights[ray * 128 + lane] = w0;
    a.weights[ray * 128 + lane + 64] = w1;
    a.inside[ray * 128 + lane] = ins[0];
    a.inside[ray * 128 + lane + 64] = ins[1];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int c = 0; c < 3; ++c) a.nhat[(ray * 128 + lane + 64 * e) * 3 + c] = nh[e][c];
    if (a.cue_b) {
      const f32x4 cv = {cue[0], cue[1], cue[2], cue[3]};
      reinterpret_cast<f32x4*>(a.cue_b)[ray * 128 + lane] = cv;
      reinterpret_cast<f32x4*>(a.cue_b)[ray * 128 + lane + 64] = cv;
    }
    // coarse shadow samples z_j = lin[j] * L * (1 - offset)  (+ stratified jitter in training, :388-395)
    const float om = a.shadow_om;
    const int snc = a.s_coarse ? a.s_coarse : 64;
    const int sl = min(lane, snc - 1);
    float zj = a.lin64[sl] * L * om;
    if (a.t_rand_shadow) {
      const float zp = (sl > 0) ? a.lin64[sl - 1] * L * om : zj;
      const float zn = (sl < snc - 1) ? a.lin64[sl + 1] * L * om : zj;
      const float lower = (sl > 0) ? 0.5f * (zj + zp) : zj;
      const float upper = (sl < snc - 1) ? 0.5f * (zn + zj) : zj;
      zj = lower + (upper - lower) * a.t_rand_shadow[ray * snc + sl];
    }
    if (lane < snc) a.zs[ray * 128 + lane] = zj;
    if (a.tail_t && lane == 63) a.tail_t[ray] = T1 * (1.0f - al[1] + 1e-7f);
    if (lane == 0) {
      a.depth[ray] = depth;
      a.wsum[ray] = wsum;
#pragma unroll
      for (int i = 0; i < 4; ++i) a.cue[ray * 4 + i] = cue[i];
      a.srd[ray * 3 + 0] = svx / L;
      a.srd[ray * 3 + 1] = svy / L;
      a.srd[ray * 3 + 2] = svz / L;
      a.slast[ray] = L / (float)snc;
      if (a.hit) { a.hit[ray * 3 + 0] = hx; a.hit[ray * 3 + 1] = hy; a.hit[ray * 3 + 2] = hz; }
      if (a.hit_n) { a.hit_n[ray * 3 + 0] = nx; a.hit_n[ray * 3 + 1] = ny; a.hit_n[ray * 3 + 2] = nz; }
    }
  }
}

// -------------------------------------------------------------------------------------------------
// shadow ray: alpha -> transmittance in front of the last sample; then the per-ray encodings for the colour net
// -------------------------------------------------------------------------------------------------
struct ShadowArgs {
  const float* rd;      // primary ray direction (view)
  const float* pl;      // light position
  const float* srd;     // shadow ray direction
  const float* sdf;     // [N,128] along the shadow ray
  const float* grad;    // [N*128,3]
  const float* dists;   // [N,128]
  const float* cue;     // [N,4]
  float* vis;           // [N]
  float* raymisc;       // [N,RAYMISC_STRIDE]
  float inv_s, cos_anneal;
  const float* dyn;     // optional device [inv_s, cos_anneal], see CoreArgs
  int nrays;
  int zero_hints;       // geometry warm-up: hints are zero (models/neus_hint_model.py:577-579, 617-619)
  int row_mul, row_off; // vis / raymisc row of ray r: r * row_mul + row_off (0, 0 = r; the partial shadow mode writes group g of
                        // clip groups per ray: row_mul = clip, row_off = g)
  int nreal;            // samples of the shadow ray that exist (0 = 128): visibility = transmittance in front of sample nreal - 1
};

__device__ __forceinline__ float enc4_entry_dyn(const float* x, int D, int e) {
  if (e < D) return x[e];
  int idx = e - D;
  float ph = 0.0f;
  if (idx >= D * 4) { idx -= D * 4; ph = NRH_HALF_PI; }
  const int d = idx >> 2, k = idx & 3;
  return sin_cw(x[d] * (float)(1 << k) + ph);
}

__global__ __launch_bounds__(256) void shadow_finish_kernel(const ShadowArgs a) {
  __shared__ float sv[RAYS_PER_BLOCK][12];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ray_raw = blockIdx.x * RAYS_PER_BLOCK + wave;
  const bool active = ray_raw < a.nrays;
  const long long ray = active ? ray_raw : a.nrays - 1;
  float vis = 0.0f;
  if (!a.zero_hints) {
    const float dx = a.srd[ray * 3 + 0], dy = a.srd[ray * 3 + 1], dz = a.srd[ray * 3 + 2];
    float al[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const long long P = ray * 128 + lane + 64 * e;
      al[e] = neus_alpha(a.sdf[P], a.grad[P * 3 + 0], a.grad[P * 3 + 1], a.grad[P * 3 + 2], dx, dy, dz, a.dists[P],
                         a.dyn ? a.dyn[0] : a.inv_s, a.dyn ? a.dyn[1] : a.cos_anneal);
    }
    float T0, T1;
    excl_prod_128(1.0f - al[0] + 1e-7f, 1.0f - al[1] + 1e-7f, T0, T1);
    const int jl = (a.nreal ? a.nreal : 128) - 1;      // the last sample that exists; padded ones behind it do not enter
    vis = __shfl(jl >= 64 ? T1 : T0, jl & 63, 64);  // exclusive product at j = 127 (models/neus_hint_model.py:429-432)
  }
  float* V = sv[wave];
  if (lane < 3) V[lane] = a.rd[ray * 3 + lane];
  if (lane >= 3 && lane < 6) V[lane] = a.pl[ray * 3 + lane - 3];
  if (lane == 6) V[6] = vis;
  if (lane >= 7 && lane < 11) V[lane] = a.zero_hints ? 0.0f : a.cue[ray * 4 + lane - 7];
  __syncthreads();
  if (active) {
    const long long row = a.row_mul ? ray * a.row_mul + a.row_off : ray;
    if (lane == 0) a.vis[row] = vis;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int i = lane + 64 * e;
      if (i < 99) {
        float v;
        if (i < 27) v = enc4_entry_dyn(V + 0, 3, i);        // enc4(view dir)
        else if (i < 54) v = enc4_entry_dyn(V + 3, 3, i - 27);  // enc4(light position, un-normalised)
        else if (i < 63) v = enc4_entry_dyn(V + 6, 1, i - 54);  // enc4(visibility)
        else v = enc4_entry_dyn(V + 7, 4, i - 63);              // enc4(specular cue)
        a.raymisc[row * RAYMISC_STRIDE + i] = v;
      }
    }
  }
}

// -------------------------------------------------------------------------------------------------
// Partial visibility hint (n_shadow_importance_clip > 0, models/neus_hint_model.py:553-575): a shadow ray per group of
// 128 / clip consecutive samples, aimed at the group's first sample position z[ray, g * ratio] (z_vals, not the mid-points).
// One wave per ray sets up group g's shadow ray exactly as core_alpha_kernel does for the hit point (get_visibility :380-395).
// -------------------------------------------------------------------------------------------------
struct PartialSetupArgs {
  const float* ro;
  const float* rd;
  const float* pl;
  const float* z;              // [N,128] sorted sample positions of the primary ray
  const float* lin64;
  const float* t_rand_shadow;  // [N*clip,64] or null; row ray * clip + g
  float* srd;                  // [N,3]
  float* slast;                // [N]
  float* zs;                   // [N,128] (first 64: coarse shadow samples)
  float shadow_om;             // 1 - renderer.shadow_ray_offset
  int z_index;                 // g * ratio
  int clip, group;
  int nrays;
};

__global__ __launch_bounds__(256) void partial_shadow_setup_kernel(const PartialSetupArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long ray = (long long)blockIdx.x * RAYS_PER_BLOCK + wave;
  if (ray >= a.nrays) return;
  const float zt = a.z[ray * 128 + a.z_index];
  const float hx = a.ro[ray * 3 + 0] + a.rd[ray * 3 + 0] * zt, hy = a.ro[ray * 3 + 1] + a.rd[ray * 3 + 1] * zt,
              hz = a.ro[ray * 3 + 2] + a.rd[ray * 3 + 2] * zt;
  const float svx = hx - a.pl[ray * 3 + 0], svy = hy - a.pl[ray * 3 + 1], svz = hz - a.pl[ray * 3 + 2];
  const float L = sqrtf(svx * svx + svy * svy + svz * svz);
  const float om = a.shadow_om;
  float zj = a.lin64[lane] * L * om;
  if (a.t_rand_shadow) {
    const float zp = (lane > 0) ? a.lin64[lane - 1] * L * om : zj;
    const float zn = (lane < 63) ? a.lin64[lane + 1] * L * om : zj;
    const float lower = (lane > 0) ? 0.5f * (zj + zp) : zj;
    const float upper = (lane < 63) ? 0.5f * (zn + zj) : zj;
    zj = lower + (upper - lower) * a.t_rand_shadow[(ray * a.clip + a.group) * 64 + lane];
  }
  a.zs[ray * 128 + lane] = zj;
  if (lane == 0) {
    a.srd[ray * 3 + 0] = svx / L;
    a.srd[ray * 3 + 1] = svy / L;
    a.srd[ray * 3 + 2] = svz / L;
    a.slast[ray] = L / 64.0f;
  }
}

// shadow_map = the group visibility at the maximal-weight sample (:573-574; first index on ties, torch.argmax)
__global__ __launch_bounds__(256) void partial_shadow_map_kernel(const float* weights, const float* vis_groups, float* shadow_map,
                                                                 int clip, int nrays) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long ray = (long long)blockIdx.x * RAYS_PER_BLOCK + wave;
  if (ray >= nrays) return;
  const float w0 = weights[ray * 128 + lane], w1 = weights[ray * 128 + lane + 64];
  float wmax = fmaxf(w0, w1);
#pragma unroll
  for (int o2 = 32; o2 > 0; o2 >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, o2, 64));
  int idx = (w0 == wmax) ? lane : ((w1 == wmax) ? lane + 64 : 1 << 20);
#pragma unroll
  for (int o2 = 32; o2 > 0; o2 >>= 1) idx = min(idx, __shfl_xor(idx, o2, 64));
  if (lane == 0) shadow_map[ray] = vis_groups[ray * clip + idx / (128 / clip)];
}

// -------------------------------------------------------------------------------------------------
// rgb = sum_j c_j w_j + background * (1 - sum_j w_j)
// -------------------------------------------------------------------------------------------------
struct CompositeArgs {
  const float* color;    // [N*128,3]
  const float* weights;  // [N,128]
  const float* wsum;     // [N]
  const float* bg;       // [3] or null
  float* rgb;            // [N,3]
  // optional per-pixel normal maps  sum_j n_j w_j inside_j  (pipelines/base_pipeline.py:125-131, einsum on the CPU there)
  const float* inside;   // [N,128]
  const float* grad;     // [N*128,3] analytic normals
  const float* nhat;     // [N*128,3] normalised
  float* nmap;           // [N,3] or null
  float* nnmap;          // [N,3] or null
  int nrays;
};
__global__ __launch_bounds__(256) void composite_kernel(const CompositeArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ray_raw = blockIdx.x * RAYS_PER_BLOCK + wave;
  const bool active = ray_raw < a.nrays;
  const long long ray = active ? ray_raw : a.nrays - 1;
  float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const long long P = ray * 128 + lane + 64 * e;
    const float w = a.weights[P];
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] += a.color[P * 3 + c] * w;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) acc[c] = wave_sum(acc[c]);
  if (active && lane == 0) {
    const float ws = a.wsum[ray];
#pragma unroll
    for (int c = 0; c < 3; ++c) a.rgb[ray * 3 + c] = acc[c] + (a.bg ? a.bg[c] * (1.0f - ws) : 0.0f);
  }
  if (a.nmap || a.nnmap) {
    float m0[3] = {0.f, 0.f, 0.f}, m1[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const long long P = ray * 128 + lane + 64 * e;
      const float w = a.weights[P] * a.inside[P];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        m0[c] += a.grad[P * 3 + c] * w;
        m1[c] += a.nhat[P * 3 + c] * w;
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { m0[c] = wave_sum(m0[c]); m1[c] = wave_sum(m1[c]); }
    if (active && lane == 0) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (a.nmap) a.nmap[ray * 3 + c] = m0[c];
        if (a.nnmap) a.nnmap[ray * 3 + c] = m1[c];
      }
    }
  }
}

// -------------------------------------------------------------------------------------------------
// pixel -> ray for one pinhole view (camera/ray_generator.py:79-139, no pose/light deltas):
//   dir_cam = ((x + .5 - cx)/fx, -(y + .5 - cy)/fy, -1);  d = normalize(R dir_cam);  o = t;  near/far = mid -+ 1
// -------------------------------------------------------------------------------------------------
struct RayGenArgs {
  float pose[12];   // row-major [3,4] camera-to-world
  float pl[3];      // point light position of the view
  float cx, cy, fx, fy;
  int width, row0, nrows;
  float* origins;   // [nrows*width,3]
  float* dirs;
  float* pls;
  float* nears;     // [nrows*width]
  float* fars;
};
__global__ void raygen_kernel(const RayGenArgs a) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)a.nrows * a.width) return;
  const int y = a.row0 + (int)(i / a.width), x = (int)(i % a.width);
  const float dcx = ((float)x + 0.5f - a.cx) / a.fx;
  const float dcy = -((float)y + 0.5f - a.cy) / a.fy;
  const float dcz = -1.0f;
  float d[3], o[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    d[r] = dcx * a.pose[r * 4 + 0] + dcy * a.pose[r * 4 + 1] + dcz * a.pose[r * 4 + 2];  // sum(dirs[None,:] * R, -1)
    o[r] = a.pose[r * 4 + 3];
  }
  const float nrm = fmaxf(sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), 1e-12f);
#pragma unroll
  for (int r = 0; r < 3; ++r) d[r] /= nrm;
  const float aa = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
  const float bb = 2.0f * (o[0] * d[0] + o[1] * d[1] + o[2] * d[2]);
  const float mid = 0.5f * (-bb) / aa;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    a.origins[i * 3 + r] = o[r];
    a.dirs[i * 3 + r] = d[r];
    a.pls[i * 3 + r] = a.pl[r];
  }
  a.nears[i] = mid - 1.0f;
  a.fars[i] = mid + 1.0f;
}

// -------------------------------------------------------------------------------------------------
// pixel bundle -> rays with per-view refinement (camera/ray_generator.py:75-150, the training form): every ray carries its
// own pixel (h, w), camera-to-world pose and light; `delta` [ncam,3,4] is the per-view left delta (noise then SO3xR3/SE3
// adjustment, composed per VIEW on the host side of the boundary - ncam x 12 floats) and `pl_delta` [ncam,3] the light one.
//   R = dR R0,  t = dt + dR t0,  d = normalize(R dir_cam),  o = t,  pl = pl0 + dpl,  near/far = chord mid-point -+ 1 | zn/zf
// The adjoint kernel scatters d(loss)/d(delta), d(loss)/d(pl_delta) with atomics (a batch holds few distinct views).
// -------------------------------------------------------------------------------------------------
struct RayGenIdxArgs {
  const long long* img;   // [n] view index or nullptr (novel views: no refinement, :103-105)
  const float* hidx;      // [n] pixel row (float, as the reference's bundle stores them)
  const float* widx;      // [n]
  const float* poses;     // [n, pose_stride] row-major camera-to-world, first 12 floats used
  const float* pls;       // [n,3]
  const float* delta;     // [ncam,3,4] or nullptr
  const float* pl_delta;  // [ncam,3] or nullptr
  int pose_stride, ncam, sphere;
  float cx, cy, fx, fy, zn, zf;
  long long n;
  float* origins; float* dirs; float* pl_out; float* nears; float* fars;
  // adjoint
  const float* g_o; const float* g_d; const float* g_pl; const float* g_near; const float* g_far;
  float* g_delta; float* g_pl_delta;
};

__device__ __forceinline__ void raygen_idx_point(const RayGenIdxArgs& a, long long i, float R[9], float t[3], float c[3], float R0[9],
                                                 float t0[3], int& cam) {
  c[0] = (a.widx[i] + 0.5f - a.cx) / a.fx;
  c[1] = -(a.hidx[i] + 0.5f - a.cy) / a.fy;
  c[2] = -1.0f;
  const float* P = a.poses + i * a.pose_stride;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int k = 0; k < 3; ++k) R0[r * 3 + k] = P[r * 4 + k];
    t0[r] = P[r * 4 + 3];
  }
  // a view index outside [0, ncam) refines nothing (and scatters nothing in the adjoint): no out-of-range access for a bad index
  const long long cam_raw = a.img ? a.img[i] : -1;
  cam = (cam_raw >= 0 && cam_raw < (long long)a.ncam) ? (int)cam_raw : -1;
  if (cam >= 0 && a.delta) {
    const float* D = a.delta + (long long)cam * 12;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
      for (int k = 0; k < 3; ++k) R[r * 3 + k] = D[r * 4 + 0] * R0[0 * 3 + k] + D[r * 4 + 1] * R0[1 * 3 + k] + D[r * 4 + 2] * R0[2 * 3 + k];
      t[r] = D[r * 4 + 3] + (D[r * 4 + 0] * t0[0] + D[r * 4 + 1] * t0[1] + D[r * 4 + 2] * t0[2]);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = R0[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) t[k] = t0[k];
  }
}

__global__ void raygen_indexed_kernel(const RayGenIdxArgs a) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  float R[9], t[3], c[3], R0[9], t0[3];
  int cam;
  raygen_idx_point(a, i, R, t, c, R0, t0, cam);
  float d[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) d[r] = c[0] * R[r * 3 + 0] + c[1] * R[r * 3 + 1] + c[2] * R[r * 3 + 2];
  const float nrm = fmaxf(sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]), 1e-12f);
#pragma unroll
  for (int r = 0; r < 3; ++r) d[r] /= nrm;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    a.origins[i * 3 + r] = t[r];
    a.dirs[i * 3 + r] = d[r];
    float p = a.pls[i * 3 + r];
    if (cam >= 0 && a.pl_delta) p += a.pl_delta[(long long)cam * 3 + r];
    a.pl_out[i * 3 + r] = p;
  }
  if (a.sphere) {
    const float aa = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    const float bb = 2.0f * (t[0] * d[0] + t[1] * d[1] + t[2] * d[2]);
    const float mid = 0.5f * (-bb) / aa;
    a.nears[i] = mid - 1.0f;
    a.fars[i] = mid + 1.0f;
  } else {
    a.nears[i] = a.zn;
    a.fars[i] = a.zf;
  }
}

__global__ void raygen_indexed_adjoint_kernel(const RayGenIdxArgs a) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  float R[9], t[3], c[3], R0[9], t0[3];
  int cam;
  raygen_idx_point(a, i, R, t, c, R0, t0, cam);
  if (cam < 0) return;
  if (a.g_pl_delta && a.g_pl) {
#pragma unroll
    for (int r = 0; r < 3; ++r) atomicAdd(a.g_pl_delta + (long long)cam * 3 + r, a.g_pl[i * 3 + r]);
  }
  if (!a.g_delta) return;
  float v[3], d[3], go[3], gd[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) v[r] = c[0] * R[r * 3 + 0] + c[1] * R[r * 3 + 1] + c[2] * R[r * 3 + 2];
  const float len = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  const float nrm = fmaxf(len, 1e-12f);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    d[r] = v[r] / nrm;
    go[r] = a.g_o ? a.g_o[i * 3 + r] : 0.0f;
    gd[r] = a.g_d ? a.g_d[i * 3 + r] : 0.0f;
  }
  if (a.sphere && (a.g_near || a.g_far)) {
    const float gm = (a.g_near ? a.g_near[i] : 0.0f) + (a.g_far ? a.g_far[i] : 0.0f);
    const float aa = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    const float od = t[0] * d[0] + t[1] * d[1] + t[2] * d[2];
#pragma unroll
    for (int r = 0; r < 3; ++r) {             // mid = -(o.d) / (d.d)
      go[r] += gm * (-d[r] / aa);
      gd[r] += gm * (-t[r] / aa + 2.0f * od * d[r] / (aa * aa));
    }
  }
  // d = v / max(|v|, eps): projector (the clamp branch passes g / eps straight through)
  float gv[3];
  if (len >= 1e-12f) {
    const float dg = d[0] * gd[0] + d[1] * gd[1] + d[2] * gd[2];
#pragma unroll
    for (int r = 0; r < 3; ++r) gv[r] = (gd[r] - d[r] * dg) / nrm;
  } else {
#pragma unroll
    for (int r = 0; r < 3; ++r) gv[r] = gd[r] / nrm;
  }
  // v = dR (R0 c),  t = dt + dR t0
  float r0c[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) r0c[k] = R0[k * 3 + 0] * c[0] + R0[k * 3 + 1] * c[1] + R0[k * 3 + 2] * c[2];
  float* G = a.g_delta + (long long)cam * 12;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int k = 0; k < 3; ++k) atomicAdd(G + r * 4 + k, gv[r] * r0c[k] + go[r] * t0[k]);
    atomicAdd(G + r * 4 + 3, go[r]);
  }
}

}  // namespace nrh
