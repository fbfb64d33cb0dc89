// End-to-end check of the activation hand-off of the wide kernels on gfx950:
//   x -> v_cvt_pkrtz (hi) -> v_fma_mix_f32 (x - hi) -> v_cvt_pkrtz (lo) -> v_accvgpr_write -> v_mfma_f32_32x32x16_f16 B operand
// exactly as gen_mlp32.py split_ops emits it (nrh32::split2), for values whose residuals (and, further down, hi halves) are
// fp16 subnormals.  A is a 0/1 selection matrix (A[row][k] = (row == k), rows 0..15), so D[row][col] = B[k = row][col]: the
// MFMA hands back what it SAW in the AGPRs.  Both halves are compared bit for bit with the host's round-toward-zero fp16 split.
// Second pass: the same B operand from arch VGPRs instead of AGPRs.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __fp16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  const h16x2 h = __builtin_amdgcn_cvt_pkrtz(a, b);
  const float ra = __builtin_fmaf((float)h.x, -1.0f, a);
  const float rb = __builtin_fmaf((float)h.y, -1.0f, b);
  const h16x2 l = __builtin_amdgcn_cvt_pkrtz(ra, rb);
  hi = __builtin_bit_cast(uint32_t, h);
  lo = __builtin_bit_cast(uint32_t, l);
}

template <bool FROM_AGPR>
__global__ __launch_bounds__(64) void k(const float* x /*[64][8]*/, float* dhi /*[16][64]*/, float* dlo, uint32_t* raw /*[64][8] hi4 lo4*/) {
  const int lane = threadIdx.x, row = lane & 31, hf = lane >> 5;
  uint32_t h[4], l[4];
  for (int p = 0; p < 4; ++p) split2(x[lane * 8 + 2 * p], x[lane * 8 + 2 * p + 1], h[p], l[p]);
  for (int p = 0; p < 4; ++p) { raw[lane * 8 + p] = h[p]; raw[lane * 8 + 4 + p] = l[p]; }
  // A[row][k = 8 hf + i] = (row == 8 hf + i)
  f16x8 a;
  for (int i = 0; i < 8; ++i) a[i] = (row == 8 * hf + i) ? (_Float16)1.0f : (_Float16)0.0f;
  u32x4 av = __builtin_bit_cast(u32x4, a);
  f32x16 dh, dl;
  if (FROM_AGPR) {
    asm volatile("v_accvgpr_write_b32 a0, %0\n\tv_accvgpr_write_b32 a1, %1\n\tv_accvgpr_write_b32 a2, %2\n\tv_accvgpr_write_b32 a3, %3\n\t"
                 "v_accvgpr_write_b32 a4, %4\n\tv_accvgpr_write_b32 a5, %5\n\tv_accvgpr_write_b32 a6, %6\n\tv_accvgpr_write_b32 a7, %7\n\ts_nop 4"
                 ::"v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]), "v"(l[0]), "v"(l[1]), "v"(l[2]), "v"(l[3]) : "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[0:3], 0" : "=&v"(dh) : "v"(av));
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, a[4:7], 0\n\ts_nop 7\n\ts_nop 7" : "=&v"(dl) : "v"(av));
  } else {
    const u32x4 bh = {h[0], h[1], h[2], h[3]}, bl = {l[0], l[1], l[2], l[3]};
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(dh) : "v"(av), "v"(bh));
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0\n\ts_nop 7\n\ts_nop 7" : "=&v"(dl) : "v"(av), "v"(bl));
  }
  for (int r = 0; r < 16; ++r) { dhi[r * 64 + lane] = dh[r]; dlo[r * 64 + lane] = dl[r]; }
}

static float half_rtz(float v) {      // float32 -> fp16 round toward zero (subnormals kept), back to float
  if (v == 0.0f || !isfinite(v)) return v;
  int e; frexpf(fabsf(v), &e);         // |v| = m 2^e, m in [0.5, 1)
  int q = (e - 1 < -14 ? -14 : e - 1) - 10;   // quantum exponent: 2^(E-10), E >= -14
  const float s = ldexpf(1.0f, q);
  const float t = truncf(v / s) * s;
  return fabsf(t) > 65504.0f ? copysignf(65504.0f, v) : t;
}

int main() {
  float hx[512];
  // lanes: magnitudes from 8 down to 2^-22 in the scaled domain, both signs, awkward mantissas
  for (int i = 0; i < 512; ++i) {
    const int k = i % 32;
    const float mant = 1.0f + (float)((i * 2654435761u) >> 9 & 0x7fffff) / 8388608.0f;
    hx[i] = ((i / 32) & 1 ? -1.0f : 1.0f) * mant * ldexpf(1.0f, 3 - k);
  }
  float *dx, *dh, *dl; uint32_t* dr;
  hipMalloc(&dx, sizeof(hx)); hipMalloc(&dh, 1024 * 4); hipMalloc(&dl, 1024 * 4); hipMalloc(&dr, 512 * 4);
  hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 0) hipLaunchKernelGGL(k<true>, dim3(1), dim3(64), 0, 0, dx, dh, dl, dr);
    else hipLaunchKernelGGL(k<false>, dim3(1), dim3(64), 0, 0, dx, dh, dl, dr);
    float oh[1024], ol[1024]; uint32_t raw[512];
    hipMemcpy(oh, dh, sizeof(oh), hipMemcpyDeviceToHost); hipMemcpy(ol, dl, sizeof(ol), hipMemcpyDeviceToHost);
    hipMemcpy(raw, dr, sizeof(raw), hipMemcpyDeviceToHost);
    int bad_split = 0, bad_mfma = 0, nsub_lo = 0, nsub_hi = 0;
    for (int lane = 0; lane < 64; ++lane)
      for (int i = 0; i < 8; ++i) {
        const float x = hx[lane * 8 + i];
        const float ehi = half_rtz(x), elo = half_rtz(x - ehi);
        if (fabsf(elo) > 0 && fabsf(elo) < 6.103515625e-05f) ++nsub_lo;
        if (fabsf(ehi) > 0 && fabsf(ehi) < 6.103515625e-05f) ++nsub_hi;
        // what the split produced (raw fp16 bits)
        const uint32_t wh = raw[lane * 8 + i / 2], wl = raw[lane * 8 + 4 + i / 2];
        const uint16_t bh = (i & 1) ? wh >> 16 : wh & 0xffff, bl = (i & 1) ? wl >> 16 : wl & 0xffff;
        _Float16 fh, fl; memcpy(&fh, &bh, 2); memcpy(&fl, &bl, 2);
        if ((float)fh != ehi || (float)fl != elo) { if (bad_split++ < 5) printf("  split: x=%.9e hi %.9e (want %.9e) lo %.9e (want %.9e)\n", x, (float)fh, ehi, (float)fl, elo); }
        // what the MFMA saw: D[row = 8 hf + i][col = lane & 31]; register r of lane (hf2, col): row = (r&3) + 8 (r>>2) + 4 hf2
        const int row = 8 * (lane >> 5) + i, col = lane & 31;
        const int hf2 = (row >> 2) & 1, r = (row & 3) + 4 * (row >> 3);
        const float mh = oh[r * 64 + 32 * hf2 + col], ml = ol[r * 64 + 32 * hf2 + col];
        if (mh != ehi || ml != elo) { if (bad_mfma++ < 5) printf("  mfma: x=%.9e saw hi %.9e (want %.9e) lo %.9e (want %.9e)\n", x, mh, ehi, ml, elo); }
      }
    printf("B from %s: 512 values, %d subnormal residuals, %d subnormal hi halves; split mismatches %d, MFMA mismatches %d\n",
           pass == 0 ? "AGPRs" : "VGPRs", nsub_lo, nsub_hi, bad_split, bad_mfma);
  }
  return 0;
}
