// Stand-alone host program over the C ABI of libnrhints_hip.so - no Python, no torch: what a maintainer of a compiled
// host (or any FFI) links against.  It reads a flat binary scene file written by examples/dump_scene.py (packed
// network buffers + a batch of rays), renders the rays with nrh_render_forward and writes rgb / depth / visibility as
// raw float32.  tests/test_gpu_parity.py::test_c_abi_standalone_program runs it and compares with the Python host.
//
//   hipcc -O2 --offload-arch=gfx950 examples/c_abi_render.cpp -Lnrhints_amd/lib -lnrhints_hip -o examples/c_abi_render
//   LD_LIBRARY_PATH=nrhints_amd/lib examples/c_abi_render scene.bin out.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "../include/nrhints_hip.h"

#define HIP_OK(x)                                                                      \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));                   \
      return 2;                                                                        \
    }                                                                                  \
  } while (0)
#define NRH_CHECK(x)                                                                   \
  do {                                                                                 \
    int rc_ = (x);                                                                     \
    if (rc_ != 0) {                                                                    \
      fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, nrh_last_error_string());       \
      return 3;                                                                        \
    }                                                                                  \
  } while (0)

// file layout (little endian): int64 header[10] = {magic, precision, hints | feat_fused << 1, nrays, bytes of sdf_w, sdf_b, sdf_head,
// col_w, col_b, bytes of the wide block}; float inv_s; float cos_anneal; then the five buffers, then the wide block (0 bytes, or
// the streams of the wide f16x3 SDF kernels followed by their [11][256] float32 tables: NrhNet.sdf_w32 / sdf_tab32, optionally
// followed by the reflectance net's block stream and its [5][256] tables: NrhNet.col_w32 / col_tab32), then
// o, d, pl [n,3], near, far [n], background [3], lin64 [64], lin16 [16] as float32
static const int64_t MAGIC = 0x4e52483031;  // "NRH01"

static bool read_exact(FILE* f, void* dst, size_t bytes) { return fread(dst, 1, bytes, f) == bytes; }

template <typename T>
static T* to_device(const std::vector<T>& h) {
  T* d = nullptr;
  if (hipMalloc((void**)&d, h.size() * sizeof(T)) != hipSuccess) return nullptr;
  if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}

int main(int argc, char** argv) {
  if (argc != 3) {
    fprintf(stderr, "usage: %s scene.bin out.bin\n", argv[0]);
    return 1;
  }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  int64_t hdr[10];
  float inv_s = 0.f, cos_anneal = 1.f;
  if (!read_exact(f, hdr, sizeof(hdr)) || hdr[0] != MAGIC || !read_exact(f, &inv_s, 4) || !read_exact(f, &cos_anneal, 4)) {
    fprintf(stderr, "bad scene file\n");
    return 1;
  }
  const int precision = (int)hdr[1], hints = (int)(hdr[2] & 1), feat_fused = (int)((hdr[2] >> 1) & 1);
  const long long n = hdr[3];
  std::vector<std::vector<char>> blobs(5);
  for (int i = 0; i < 5; ++i) {
    blobs[i].resize((size_t)hdr[4 + i]);
    if (!read_exact(f, blobs[i].data(), blobs[i].size())) { fprintf(stderr, "truncated scene file\n"); return 1; }
  }
  std::vector<char> wide((size_t)hdr[9]);
  if (!wide.empty() && !read_exact(f, wide.data(), wide.size())) { fprintf(stderr, "truncated scene file\n"); return 1; }
  auto read_f = [&](size_t count) {
    std::vector<float> v(count);
    if (!read_exact(f, v.data(), count * 4)) v.clear();
    return v;
  };
  std::vector<float> o = read_f(n * 3), d = read_f(n * 3), pl = read_f(n * 3), nearv = read_f(n), farv = read_f(n), bg = read_f(3),
                     lin64 = read_f(64), lin16 = read_f(16);
  fclose(f);
  if (lin16.empty()) { fprintf(stderr, "truncated scene file\n"); return 1; }

  printf("%s | abi %d | %lld rays, precision %d, hints %d\n", nrh_build_info(), nrh_version(), n, precision, hints);
  NrhNet net = {};
  void* dev[5];
  for (int i = 0; i < 5; ++i) {
    dev[i] = to_device(blobs[i]);
    if (!dev[i]) { fprintf(stderr, "device upload failed\n"); return 2; }
  }
  net.sdf_w = (const float*)dev[0]; net.sdf_b = (const float*)dev[1]; net.sdf_head = (const float*)dev[2];
  net.col_w = (const float*)dev[3]; net.col_b = (const float*)dev[4];
  if (!wide.empty()) {
    const long long stream_bytes = nrh_sdf_wide_stream_bytes(), sdf_part = stream_bytes + 11 * 256 * 4;
    const long long col_bytes = nrh_color_wide_stream_bytes(), col_part = col_bytes + 5 * 256 * 4;
    if ((long long)wide.size() != sdf_part && (long long)wide.size() != sdf_part + col_part) {
      fprintf(stderr, "wide block has the wrong size for this library\n");
      return 1;
    }
    char* d_wide = to_device(wide);
    if (!d_wide) { fprintf(stderr, "device upload failed\n"); return 2; }
    net.sdf_w32 = d_wide;
    net.sdf_tab32 = (const float*)(d_wide + stream_bytes);
    net.feat_fused = feat_fused;   // the FEAT block of the streams already holds W0feat * W_feat (include/nrhints_hip.h)
    if ((long long)wide.size() == sdf_part + col_part) {   // + the reflectance net's block stream and tables
      net.col_w32 = d_wide + sdf_part;
      net.col_tab32 = (const float*)(d_wide + sdf_part + col_bytes);
    }
  }
  net.inv_s = inv_s; net.precision = precision; net.hints = hints; net.normal_type = 0; net.depth_type = 0;
  float *d_o = to_device(o), *d_d = to_device(d), *d_pl = to_device(pl), *d_near = to_device(nearv), *d_far = to_device(farv),
        *d_bg = to_device(bg), *d_l64 = to_device(lin64), *d_l16 = to_device(lin16);
  float *rgb, *depth, *vis, *ws;
  HIP_OK(hipMalloc((void**)&rgb, n * 3 * 4));
  HIP_OK(hipMalloc((void**)&depth, n * 4));
  HIP_OK(hipMalloc((void**)&vis, n * 4));
  const long long ws_floats = nrh_render_workspace_floats(n);
  HIP_OK(hipMalloc((void**)&ws, (size_t)ws_floats * 4));
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  NRH_CHECK(nrh_render_forward(&net, d_o, d_d, d_pl, d_near, d_far, n, d_bg, cos_anneal, nullptr, nullptr, 0, d_l64, d_l16, rgb, depth,
                               nullptr, nullptr, nullptr, nullptr, vis, nullptr, nullptr, nullptr, nullptr, nullptr, ws, ws_floats, st));
  HIP_OK(hipStreamSynchronize(st));
  std::vector<float> h_rgb(n * 3), h_depth(n), h_vis(n);
  HIP_OK(hipMemcpy(h_rgb.data(), rgb, n * 3 * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(h_depth.data(), depth, n * 4, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(h_vis.data(), vis, n * 4, hipMemcpyDeviceToHost));
  FILE* g = fopen(argv[2], "wb");
  if (!g) { perror(argv[2]); return 1; }
  fwrite(h_rgb.data(), 4, h_rgb.size(), g);
  fwrite(h_depth.data(), 4, h_depth.size(), g);
  fwrite(h_vis.data(), 4, h_vis.size(), g);
  fclose(g);
  printf("rendered %lld rays: rgb[0] = %.6f %.6f %.6f\n", n, h_rgb[0], h_rgb[1], h_rgb[2]);
  return 0;
}
