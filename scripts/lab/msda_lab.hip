// Stand-alone lab for the MSDA backward (encoder shape of BASELINE configs[1]): times the whole tile-accumulation backward,
// the sample kernel with and without the bin words / masks and the forward kernel with HIP events, and checks grad_value
// against the atomic scatter.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I rscotr_amd/csrc -I include scripts/lab/msda_lab.hip \
//         rscotr_amd/csrc/abi.hip -o scripts/lab/msda_lab
#include <cstdio>
#include <random>
#include <vector>
#include "../../rscotr_amd/csrc/msda.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const float spread = argc > 1 ? atof(argv[1]) : 0.05f;
  const int B = 2, H = 8, D = 32, L = 4, P = 4;
  const int64_t shp[8] = {64, 64, 32, 32, 16, 16, 8, 8};
  int64_t lsi_h[4];
  int Nk = 0;
  for (int l = 0; l < L; ++l) { lsi_h[l] = Nk; Nk += (int)(shp[2 * l] * shp[2 * l + 1]); }
  const int Nq = Nk;
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> value((size_t)B * Nk * H * D), go((size_t)B * Nq * H * D), loc((size_t)B * Nq * H * L * P * 2), attn((size_t)B * Nq * H * L * P);
  for (auto& v : value) v = nd(rng);
  for (auto& v : go) v = nd(rng);
  for (int b = 0; b < B; ++b) {
    int q = 0;
    for (int l0 = 0; l0 < L; ++l0)
      for (int y = 0; y < shp[2 * l0]; ++y)
        for (int x = 0; x < shp[2 * l0 + 1]; ++x, ++q) {
          const float rx = (x + 0.5f) / shp[2 * l0 + 1], ry = (y + 0.5f) / shp[2 * l0];
          for (int h = 0; h < H; ++h) {
            float s = 0.f;
            float* a = &attn[(((size_t)b * Nq + q) * H + h) * L * P];
            for (int i = 0; i < L * P; ++i) { a[i] = expf(nd(rng)); s += a[i]; }
            for (int i = 0; i < L * P; ++i) {
              a[i] /= s;
              float* xy = &loc[((((size_t)b * Nq + q) * H + h) * L * P + i) * 2];
              xy[0] = spread < 0 ? (float)(rng() % 100000) / 100000.f : rx + spread * nd(rng);
              xy[1] = spread < 0 ? (float)(rng() % 100000) / 100000.f : ry + spread * nd(rng);
            }
          }
        }
  }
  float *dv, *dgo, *dloc, *dattn, *gv1, *gv2, *gl, *ga;
  int64_t *dshp, *dlsi;
  CK(hipMalloc(&dv, value.size() * 4)); CK(hipMalloc(&dgo, go.size() * 4)); CK(hipMalloc(&dloc, loc.size() * 4));
  CK(hipMalloc(&dattn, attn.size() * 4)); CK(hipMalloc(&gv1, value.size() * 4)); CK(hipMalloc(&gv2, value.size() * 4));
  CK(hipMalloc(&gl, loc.size() * 4)); CK(hipMalloc(&ga, attn.size() * 4)); CK(hipMalloc(&dshp, 64)); CK(hipMalloc(&dlsi, 32));
  CK(hipMemcpy(dv, value.data(), value.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dgo, go.data(), go.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dloc, loc.data(), loc.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dattn, attn.data(), attn.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dshp, shp, 64, hipMemcpyHostToDevice)); CK(hipMemcpy(dlsi, lsi_h, 32, hipMemcpyHostToDevice));
  MsdaTiles T1;
  if (!msda_tiles_build(&T1, shp, L, Nk, (long)Nq * P, D)) { printf("geometry failed\n"); return 1; }
  printf("partial tiles per (b,h): %d\n", T1.NW);
  const MsdaTileWs W = msda_tile_ws(T1, B * H, Nq, P, D);
  char* w1;
  CK(hipMalloc(&w1, W.total + 4096));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](auto fn, const char* name) {
    for (int i = 0; i < 3; ++i) fn();
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int i = 0; i < 20; ++i) fn();
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-46s %8.1f us\n", name, ms * 1000.f / 20);
  };
  timeit([&] { launch_bwd_tiled<32, 4>(dv, dshp, dlsi, dloc, dattn, dgo, gv1, gl, ga, B, Nk, Nq, H, L, T1, w1, s); }, "tiled: sample + tile + combine");
  {
    constexpr int QB = 32;
    const int ntiles = (Nq + QB - 1) / QB;
    MsdaMaskGeom MG;
    for (int l = 0; l < 8; ++l) { MG.itx[l] = 1.f / T1.tsx[l]; MG.ity[l] = 1.f / T1.tsy[l]; MG.ntx[l] = T1.ntx[l]; }
    timeit([&] { msda_bwd_kernel<32, 4, 0, true><<<dim3(B * ntiles * H), 256, QB * L * P * 7 * 4 + 64, s>>>(dv, dshp, dlsi, dloc, dattn, dgo, gv2, gl, ga, (int*)(w1 + W.binw), (unsigned long long*)(w1 + W.mask), MG, Nk, Nq, H, L, ntiles, 0); }, "  sample kernel, bin words + masks");
    timeit([&] { msda_bwd_kernel<32, 4, 0, false><<<dim3(B * ntiles * H), 256, QB * L * P * 6 * 4, s>>>(dv, dshp, dlsi, dloc, dattn, dgo, gv2, gl, ga, nullptr, nullptr, MG, Nk, Nq, H, L, ntiles, 0); }, "  sample kernel, grad_loc / grad_attn only");
    timeit([&] { launch_fwd<32, 4>(dv, dshp, dlsi, dloc, dattn, gv2, B, Nk, Nq, H, L, s); }, "  forward kernel");
  }
  // reference: the atomic scatter into a zeroed buffer
  CK(hipMemsetAsync(gv2, 0, value.size() * 4, s));
  launch_bwd<32, 4>(dv, dshp, dlsi, dloc, dattn, dgo, gv2, gl, ga, B, Nk, Nq, H, L, s);
  launch_bwd_tiled<32, 4>(dv, dshp, dlsi, dloc, dattn, dgo, gv1, gl, ga, B, Nk, Nq, H, L, T1, w1, s);
  CK(hipStreamSynchronize(s));
  std::vector<float> a(value.size()), b2(value.size());
  CK(hipMemcpy(a.data(), gv1, a.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b2.data(), gv2, a.size() * 4, hipMemcpyDeviceToHost));
  double md = 0, mx = 0;
  for (size_t i = 0; i < a.size(); ++i) { md = std::max(md, (double)fabsf(a[i] - b2[i])); mx = std::max(mx, (double)fabsf(a[i])); }
  printf("tiled vs atomic scatter: max |diff| %.3g of max %.3g\n", md, mx);
  return 0;
}
