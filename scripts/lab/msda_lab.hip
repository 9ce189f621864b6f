// Stand-alone lab for the MSDA backward (encoder shape of BASELINE configs[1]): times the whole tile-accumulation backward,
// the sample kernel with and without the bin words / masks and the forward kernel with HIP events, and checks grad_value
// against the atomic scatter.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I rscotr_amd/csrc -I include scripts/lab/msda_lab.hip \
//         rscotr_amd/csrc/abi.hip -o scripts/lab/msda_lab
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#ifndef MSDA_LAB_OLD
#define MSDA_T_PROFILE 1
#include "../../rscotr_amd/csrc/msda.hip"
#else  // -DMSDA_LAB_OLD -include <older msda.hip>: A/B against an earlier kernel (no phase table)
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// usage: msda_lab enc <spread>          encoder call: queries = the value tokens, locations = own position + spread * N(0,1) (spread < 0: uniform)
//        msda_lab dec <Nq> [wh_max]     decoder call: Nq queries per image with reference BOXES (centre uniform, w / h uniform in [0.05, wh_max]);
//                                       sample p of head h at centre + direction(h) * (p + 1) / P * 0.5 * wh (+ 2 % noise): DINO's box-scaled offsets
int main(int argc, char** argv) {
  const bool dec = argc > 1 && !strcmp(argv[1], "dec");
  const float spread = (!dec && argc > 2) ? atof(argv[2]) : 0.05f;
  const int B = 2, H = 8, D = 32, L = 4, P = 4;
  const int64_t shp[8] = {64, 64, 32, 32, 16, 16, 8, 8};
  int64_t lsi_h[4];
  int Nk = 0;
  for (int l = 0; l < L; ++l) { lsi_h[l] = Nk; Nk += (int)(shp[2 * l] * shp[2 * l + 1]); }
  const int Nq = dec ? (argc > 2 ? atoi(argv[2]) : 800) : Nk;
  const float wh_max = dec && argc > 3 ? atof(argv[3]) : 0.5f;
  const int npad = dec && argc > 4 ? atoi(argv[4]) : 0;  // the LAST npad queries of every image are identical (padded denoising slots: box (0.5, 0.5, 0.5, 0.5), same offsets)
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::uniform_real_distribution<float> ud(0.f, 1.f);
  std::vector<float> value((size_t)B * Nk * H * D), go((size_t)B * Nq * H * D), loc((size_t)B * Nq * H * L * P * 2), attn((size_t)B * Nq * H * L * P);
  for (auto& v : value) v = nd(rng);
  for (auto& v : go) v = nd(rng);
  for (int b = 0; b < B; ++b) {
    for (int q = 0; q < Nq; ++q) {
      float rx, ry, bw = 0.f, bh = 0.f;
      const bool pad = dec && q >= Nq - npad;
      if (pad) { rx = ry = bw = bh = 0.5f; }
      else if (dec) { rx = ud(rng); ry = ud(rng); bw = 0.05f + (wh_max - 0.05f) * ud(rng); bh = 0.05f + (wh_max - 0.05f) * ud(rng); }
      else {
        int l0 = 0, r = q;
        while (r >= shp[2 * l0] * shp[2 * l0 + 1]) { r -= (int)(shp[2 * l0] * shp[2 * l0 + 1]); ++l0; }
        const int y = r / (int)shp[2 * l0 + 1], x = r % (int)shp[2 * l0 + 1];
        rx = (x + 0.5f) / shp[2 * l0 + 1]; ry = (y + 0.5f) / shp[2 * l0];
      }
      for (int h = 0; h < H; ++h) {
        float s = 0.f;
        float* a = &attn[(((size_t)b * Nq + q) * H + h) * L * P];
        for (int i = 0; i < L * P; ++i) { a[i] = expf(nd(rng)); s += a[i]; }
        const float th = 2.f * 3.14159265f * h / H, dx = cosf(th), dy = sinf(th), dm = fmaxf(fabsf(dx), fabsf(dy));
        for (int i = 0; i < L * P; ++i) {
          a[i] /= s;
          float* xy = &loc[((((size_t)b * Nq + q) * H + h) * L * P + i) * 2];
          if (dec) {
            const int pp = i % P;
            xy[0] = rx + dx / dm * (pp + 1) / P * 0.5f * bw + (pad ? 0.f : 0.02f * nd(rng));
            xy[1] = ry + dy / dm * (pp + 1) / P * 0.5f * bh + (pad ? 0.f : 0.02f * nd(rng));
          } else {
            xy[0] = spread < 0 ? ud(rng) : rx + spread * nd(rng);
            xy[1] = spread < 0 ? ud(rng) : ry + spread * nd(rng);
          }
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
  printf("== %s Nq %d (%d identical): partial tiles per (b,h): %d;", dec ? "dec" : "enc", Nq, npad, T1.NW);
  for (int l = 0; l < L; ++l) printf("  L%d: %dx%d bins/tile, %dx%d tiles, %d chunks", l, T1.tsx[l], T1.tsy[l], T1.ntx[l], T1.nty[l], T1.nch[l]);
  printf("\n");
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
    // the tile kernel and the combine alone (bin words / masks are in place from the launches above)
    const int BH = B * H;
    const unsigned bh8 = (unsigned)((BH + 7) / 8) * 8;
    int bshift = 0;
    while ((1 << bshift) < QB * P) ++bshift;
    constexpr size_t lds = MsdaTileGeom<32>::lds_bytes();
    float* part = reinterpret_cast<float*>(w1 + W.part);
    timeit([&] { msda_tile_kernel<32, 4><<<dim3(bh8 * (unsigned)T1.NW), 256, lds, s>>>(dgo, dloc, dattn, (int*)(w1 + W.binw), (unsigned long long*)(w1 + W.mask), part, T1, Nq, bshift, ntiles, H, BH); }, "  tile kernel");
    const int bpb = (Nk + 256 / (32 / 4) - 1) / (256 / (32 / 4));
    timeit([&] { msda_tile_combine_kernel<32><<<dim3(bh8 * (unsigned)bpb), 256, 0, s>>>(part, gv1, T1, Nk, H, BH, bpb); }, "  combine kernel");
#ifndef MSDA_LAB_OLD
    // per-level phase cycles of the tile workgroups (last launch)
    static long long prof[1 << 16][8];
    CK(hipMemcpyFromSymbol(prof, HIP_SYMBOL(g_msda_tprof), sizeof(prof)));
    const unsigned nwg = bh8 * (unsigned)T1.NW;
    printf("  tile kernel: %u workgroups, %zu bytes of LDS each; kilocycles per workgroup, mean (max): prep | scan | sort | walk | final | total | samples\n", nwg, lds);
    for (int l = 0; l < L; ++l) {
      double sum[6] = {0}, mx[6] = {0}; long cnt = 0; double ns = 0, nmx = 0;
      for (unsigned i = 0; i < nwg && i < (1u << 16); ++i) {
        if (!prof[i][7] || prof[i][5] != l) continue;
        double tot = 0;
        for (int k = 0; k < 5; ++k) { sum[k] += prof[i][k]; mx[k] = std::max(mx[k], (double)prof[i][k]); tot += prof[i][k]; }
        sum[5] += tot; mx[5] = std::max(mx[5], tot); ns += prof[i][6]; nmx = std::max(nmx, (double)prof[i][6]); ++cnt;
      }
      if (!cnt) continue;
      printf("    level %d (%4ld wgs):", l, cnt);
      for (int k = 0; k < 6; ++k) printf(" %7.1f (%7.1f)", sum[k] / cnt / 1e3, mx[k] / 1e3);
      printf("  %7.0f (%6.0f)\n", ns / cnt, nmx);
    }
#endif
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
