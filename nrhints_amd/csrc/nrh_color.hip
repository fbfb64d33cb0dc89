// Reflectance MLP on gfx950 (reference: fields/reflectance_network.py:68-96, called from
// models/neus_hint_model.py:626): 361 -> 4 x (256, ReLU) -> 3 -> sigmoid, one colour per sample.
//
// The 361-wide input of the reference, cat[pts 3, enc4(view) 27, normal 3, enc4(pl) 27, feat 256, enc4(vis) 9,
// enc4(cue) 36], is never materialised.  Layer 0 is split along K:
//   C0a  feat (256)  - read straight from the D-layout tiles the SDF kernel wrote; they ARE the MFMA B operand
//   C0b  the other 105 inputs (-> 112): 6 per-sample values (point, unit normal) computed here and 99 per-ray
//        values (encodings of view dir, light position, visibility hint, specular cue) read from a per-ray
//        table that one wave per ray filled once (nrh_rays.hip) instead of 128 times
// The columns of W0 are permuted on the host to this order.  C1..C3 are plain 256x256 ReLU layers, C4 has 3 rows.
#include "nrh_mlp.h"

namespace nrh {

struct ColorArgs {
  const float* w;        // packed (COL_PACKED_FLOATS)
  const float* b;        // [4][256] + [16]
  const float* feat;     // [ntiles][16][64][4]
  const float* ro;       // [nrays,3]
  const float* rd;       // [nrays,3]
  const float* tmid;     // [nrays,128]
  const float* nhat;     // [npts,3] unit normals
  const float* raymisc;  // [nrays,RAYMISC_STRIDE]
  float* color;          // [npts,3]
  long long npts;
  int ntile_groups;
};

__global__ __launch_bounds__(MLP_THREADS, 2) void color_kernel(const ColorArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, q = lane >> 4;
  int par = 0;
  dma_chunk(a.w + COL_OFF_C0A, smem, 32, wave, lane);
  __syncthreads();

  for (int tg = blockIdx.x; tg < a.ntile_groups; tg += gridDim.x) {
    const long long tile = (long long)tg * 4 + wave;
    const long long P = tile * TILE_PTS + j;
    const bool valid = P < a.npts;
    const long long Pc = valid ? P : a.npts - 1;
    const long long tilec = Pc / TILE_PTS;
    const long long ray = Pc >> 7;  // 128 samples per ray

    // ---- C0a: feature part ----
    float h[64];
    {
      const float* ft = a.feat + (size_t)tilec * (16 * 256);
#pragma unroll
      for (int b = 0; b < 16; ++b) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(ft + (b * 64 + lane) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) h[b * 4 + r] = v[r];
      }
    }
    float part[64];
    {
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { part[ch * 8 + r] = acc0[r]; part[ch * 8 + 4 + r] = acc1[r]; }
      };
      run_stage<16, 8, false>(a.w + COL_OFF_C0A, a.w + COL_OFF_C0B, 14, smem, par, h, nullptr, epi, wave, lane);
    }
    // ---- C0b: per-sample + per-ray part; entry m = 16b + 4q + r of [p, n, raymisc[0..98]] ----
    float misc[28];
    {
      const float tt = a.tmid[Pc];
      float pn[6];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        pn[c] = a.ro[ray * 3 + c] + a.rd[ray * 3 + c] * tt;
        pn[3 + c] = a.nhat[Pc * 3 + c];
      }
      const float* rm = a.raymisc + ray * RAYMISC_STRIDE;
#pragma unroll
      for (int b = 0; b < 7; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int m = b * 16 + 4 * q + r;
          float v;
          if (b == 0) {
            // m in [0,16): first 6 are per-sample, with compile-time candidates per q
            const float per = sel_q<6>(pn, r, q);
            const int mm = (m >= 6) ? m - 6 : 0;
            const float ld = rm[mm];
            v = (m < 6) ? per : ld;
          } else {
            v = (m < COL_MISC) ? rm[(m < COL_MISC) ? m - 6 : 0] : 0.0f;
          }
          misc[b * 4 + r] = v;
        }
    }
    {
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.b + (2 * ch) * 16 + 4 * q);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(a.b + (2 * ch + 1) * 16 + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          h[ch * 8 + r] = fmaxf(acc0[r] + b0[r], 0.0f);
          h[ch * 8 + 4 + r] = fmaxf(acc1[r] + b1[r], 0.0f);
        }
      };
      run_stage<7, 8, true>(a.w + COL_OFF_C0B, a.w + col_off_C(1), 32, smem, par, misc, part, epi, wave, lane);
    }
    // ---- C1..C3 ----
    for (int l = 1; l <= 3; ++l) {
      float ho[64];
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.b + l * 256 + (2 * ch) * 16 + 4 * q);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(a.b + l * 256 + (2 * ch + 1) * 16 + 4 * q);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          ho[ch * 8 + r] = fmaxf(acc0[r] + b0[r], 0.0f);
          ho[ch * 8 + 4 + r] = fmaxf(acc1[r] + b1[r], 0.0f);
        }
      };
      const float* wn = (l < 3) ? a.w + col_off_C(l + 1) : a.w + COL_OFF_C4;
      run_stage<16, 8, false>(a.w + col_off_C(l), wn, 32, smem, par, h, nullptr, epi, wave, lane);
#pragma unroll
      for (int i = 0; i < 64; ++i) h[i] = ho[i];
    }
    // ---- C4: 3 output rows (block 0, lanes q == 0 hold r = 0..2) + sigmoid ----
    {
      auto epi = [&](int ch, f32x4 acc0, f32x4 acc1) {
        (void)acc1;
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.b + 4 * 256 + 4 * q);
        if (valid && q == 0) {
#pragma unroll
          for (int r = 0; r < 3; ++r) a.color[P * 3 + r] = sigmoidf_(acc0[r] + b0[r]);
        }
      };
      run_stage<16, 1, false>(a.w + COL_OFF_C4, a.w + COL_OFF_C0A, 32, smem, par, h, nullptr, epi, wave, lane);
    }
  }
}

}  // namespace nrh
