// Stand-alone check + timing of the wide f16x3 SDF kernels (csrc/nrh_sdf32.hip), no Python / torch:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -mllvm -amdgpu-mfma-vgpr-form \
//         -I nrhints_amd/csrc profiles/ubench/sdf32_bench.hip -o gpurun_out/sdf32_bench
//   gpurun_out/sdf32_bench profiles/ubench/data/sdf32_case.bin [reps]
// Input file: profiles/ubench/make_sdf32_data.py (packed streams, rays, fp64 oracle values at the first points).
#include "nrh_sdf32.hip"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

template <typename T>
static T* upload(const void* src, size_t bytes) {
  T* p;
  CK(hipMalloc(&p, bytes));
  CK(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
  return p;
}

static long long stream_bytes(int mode) { return nrh32::sdf32_stream_bytes(mode); }

int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "profiles/ubench/data/sdf32_case.bin";
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  FILE* f = fopen(path, "rb");
  if (!f) { printf("cannot open %s\n", path); return 2; }
  long long hdr[5];
  if (fread(hdr, 8, 5, f) != 5) return 2;
  const long long nrays = hdr[0], nper = hdr[1], ncheck = hdr[2], nstream = hdr[3], ntab = hdr[4];
  const long long npts = nrays * nper;
  std::vector<uint16_t> streams(nstream);
  std::vector<float> tab(ntab), ro(nrays * 3), rd(nrays * 3), t(npts);
  std::vector<double> e_sdf(ncheck), e_grad(ncheck * 3), e_feat(ncheck * 256);
  bool ok = fread(streams.data(), 2, nstream, f) == (size_t)nstream && fread(tab.data(), 4, ntab, f) == (size_t)ntab &&
            fread(ro.data(), 4, nrays * 3, f) == (size_t)nrays * 3 && fread(rd.data(), 4, nrays * 3, f) == (size_t)nrays * 3 &&
            fread(t.data(), 4, npts, f) == (size_t)npts && fread(e_sdf.data(), 8, ncheck, f) == (size_t)ncheck &&
            fread(e_grad.data(), 8, ncheck * 3, f) == (size_t)ncheck * 3 && fread(e_feat.data(), 8, ncheck * 256, f) == (size_t)ncheck * 256;
  fclose(f);
  if (!ok) { printf("short read\n"); return 2; }
  if (nstream * 2 != stream_bytes(0) + stream_bytes(1) + stream_bytes(2)) { printf("stream size mismatch\n"); return 2; }

  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, LDS per workgroup %d B, npts %lld\n", prop.gcnArchName, cus, nrh32::LDS_BYTES, npts);

  char* d_w = upload<char>(streams.data(), nstream * 2);
  float* d_tab = upload<float>(tab.data(), ntab * 4);
  float* d_ro = upload<float>(ro.data(), nrays * 12);
  float* d_rd = upload<float>(rd.data(), nrays * 12);
  float* d_t = upload<float>(t.data(), npts * 4);
  float *d_sdf, *d_grad, *d_feat;
  uint32_t* d_scr;
  CK(hipMalloc(&d_sdf, npts * 4));
  CK(hipMalloc(&d_grad, npts * 12));
  CK(hipMalloc(&d_feat, ((npts + 15) / 16) * 4096 * 4));
  const int grid = cus;
  CK(hipMalloc(&d_scr, (size_t)grid * nrh32::WAVES * nrh32::SCRATCH_WORDS_PER_WAVE * 4));

  const void* fns[3] = {(const void*)nrh32::sdf32_kernel<0>, (const void*)nrh32::sdf32_kernel<1>, (const void*)nrh32::sdf32_kernel<2>};
  for (int m = 0; m < 3; ++m) CK(hipFuncSetAttribute(fns[m], hipFuncAttributeMaxDynamicSharedMemorySize, nrh32::LDS_BYTES));

  const double flop_pt[3] = {2.0 * 459008, 2.0 * (459008 + 459008), 2.0 * (524544 + 459008)};
  std::vector<float> h_sdf(npts), h_grad(npts * 3), h_feat((size_t)((npts + 15) / 16) * 4096);
  int bad = 0;
  for (int mode = 0; mode < 3; ++mode) {
    nrh32::Sdf32Args a;
    long long off = 0;
    for (int m = 0; m < mode; ++m) off += stream_bytes(m);
    a.w = d_w + off; a.tab = d_tab; a.ro = d_ro; a.rd = d_rd; a.t = d_t; a.sdf = d_sdf; a.grad = d_grad; a.feat = d_feat;
    a.scratch = d_scr; a.npts = npts; a.n_per_ray = (int)nper; a.t_stride = (int)nper; a.sdf_stride = (int)nper;
    a.ngroups = (int)((npts + nrh32::GROUP - 1) / nrh32::GROUP);
    a.dbg = nullptr; a.dbg_stage = 99;
    CK(hipMemset(d_sdf, 0xff, npts * 4));
    CK(hipMemset(d_grad, 0xff, npts * 12));
    auto launch = [&]() {
      if (mode == 0) hipLaunchKernelGGL(nrh32::sdf32_kernel<0>, dim3(grid), dim3(nrh32::THREADS), nrh32::LDS_BYTES, 0, a);
      if (mode == 1) hipLaunchKernelGGL(nrh32::sdf32_kernel<1>, dim3(grid), dim3(nrh32::THREADS), nrh32::LDS_BYTES, 0, a);
      if (mode == 2) hipLaunchKernelGGL(nrh32::sdf32_kernel<2>, dim3(grid), dim3(nrh32::THREADS), nrh32::LDS_BYTES, 0, a);
    };
    launch();
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h_sdf.data(), d_sdf, npts * 4, hipMemcpyDeviceToHost));
    double esdf = 0, egrad = 0, efeat = 0, chk = 0, gsum = 0, fsum = 0;     // checksums over ALL points: sdf, gradient, features
    long long nan = 0;
    for (long long i = 0; i < npts; ++i) { if (!(h_sdf[i] == h_sdf[i])) ++nan; else chk += h_sdf[i]; }
    for (long long i = 0; i < ncheck; ++i) esdf = fmax(esdf, fabs(h_sdf[i] - e_sdf[i]));
    if (mode >= 1) {
      CK(hipMemcpy(h_grad.data(), d_grad, npts * 12, hipMemcpyDeviceToHost));
      for (long long i = 0; i < ncheck * 3; ++i) egrad = fmax(egrad, fabs(h_grad[i] - e_grad[i]));
      for (long long i = 0; i < npts * 3; ++i) { if (!(h_grad[i] == h_grad[i])) ++nan; else gsum += h_grad[i]; }
    }
    if (mode == 2) {
      CK(hipMemcpy(h_feat.data(), d_feat, h_feat.size() * 4, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < (size_t)(npts / 16) * 4096; ++i) fsum += h_feat[i];
      for (long long p = 0; p < ncheck; ++p)
        for (int ft = 0; ft < 256; ++ft) {
          const long long tile = p / 16;
          const int j = (int)(p % 16), b = ft / 16, q = (ft % 16) / 4, r = ft % 4;
          const float v = h_feat[(size_t)tile * 4096 + (b * 64 + q * 16 + j) * 4 + r];
          efeat = fmax(efeat, fabs(v - e_feat[p * 256 + ft]));
        }
    }
#ifdef NRH32_TIMING
    {
      unsigned long long* d_t;
      const int nw = grid * nrh32::WAVES;
      CK(hipMalloc(&d_t, nw * 64));
      CK(hipMemset(d_t, 0, nw * 64));
      a.dbg = reinterpret_cast<uint32_t*>(d_t);
      launch();
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> ht(nw * 8);
      CK(hipMemcpy(ht.data(), d_t, nw * 64, hipMemcpyDeviceToHost));
      double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tot = 0;
      for (int w = 0; w < nw; ++w) for (int k = 0; k < 8; ++k) { acc[k] += (double)ht[w * 8 + k] / nw; }
      for (int k = 0; k < 7; ++k) tot += acc[k];
      const double passes = (double)a.ngroups / grid;
      printf("  cycles per pass (mean over waves, %.1f passes): setup %.0f | L0 %.0f | L1..L7 %.0f | FEAT+HEAD %.0f | T7 %.0f | R7..R1+R4e %.0f | R0+out %.0f | total %.0f  (of which waiting in chunk_sync %.0f)\n",
             passes, acc[0] / passes, acc[1] / passes, acc[2] / passes, acc[3] / passes, acc[4] / passes, acc[5] / passes, acc[6] / passes, tot / passes, acc[7] / passes);
      a.dbg = nullptr;
      CK(hipFree(d_t));
    }
#endif
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    const bool pass = nan == 0 && esdf < 5e-6 && egrad < 5e-4 && efeat < 5e-5;
    if (!pass) ++bad;
    printf("mode %d: %8.3f ms  %7.1f TFLOP/s (algorithmic)  max|sdf-ref| %.3e  max|grad-ref| %.3e  max|feat-ref| %.3e  nan %lld  sum(sdf) %.6f  sum(grad) %.6f  sum(feat) %.4f  %s\n",
           mode, ms, flop_pt[mode] * npts / ms / 1e9, esdf, egrad, efeat, nan, chk, gsum, fsum, pass ? "PASS" : "FAIL");
  }
  return bad ? 1 : 0;
}
