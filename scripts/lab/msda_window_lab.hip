// Lab for VERDICT r5 item 3 / north_star "K/V tiles staged in LDS": the MSDA FORWARD with the value windows of a block of queries
// staged in LDS, against the shipped kernel (rscotr_msda_fwd: every tap a 128-byte L1/L2 gather), at the encoder shape of BASELINE
// configs[1] (B = 2, pyramid 64^2 + 32^2 + 16^2 + 8^2 = 5440 tokens per image, 8 heads x 32 channels, 4 levels x 4 points).
//
// Window kernel: one workgroup (512 threads) = an 8 x 8 block of queries of one level (8 x 8 spatial neighbours) x one head.
//   1. every sample of the block (64 queries x 16) is set up ONCE by one thread (as the shipped kernel does: 32-byte records in LDS)
//      and its taps extend the per-level bounding box (LDS integer min / max);
//   2. the boxes are cut to the LDS budget (caps below, 1 000 pixels x 128 bytes = 125 KB: one workgroup per CU) around their centre;
//   3. the windows are loaded cooperatively (one 16-byte load per lane, coalesced 128-byte pixels);
//   4. the gather loop of a lane group (8 lanes x float4 = the 32 channels of a head) takes a tap from the window when it lies inside
//      and from global memory otherwise.
// Prints: max |difference| to the shipped kernel, the share of taps served from LDS, kernel times (HIP events, back-to-back
// launches; run under rocprofv3 --kernel-trace --stats for the per-kernel durations).
//   bash scripts/lab/build_msda_window_lab.sh   (links the tree's librscotr.so for the shipped kernel)
//   scripts/lab/msda_window_lab init [noise_px]     sampling offsets of mmcv's init (head direction x (p + 1) pixels of the level) + noise
//   scripts/lab/msda_window_lab spread <s>          reference point + s * N(0, 1) (normalised units)
//   ... [cap0 cap1 cap2 cap3]                      window capacities in pixels (default 400 256 196 144 = 125 KB + 32 KB of records: one workgroup per CU;
//                                                  160 96 64 36 = 45 KB + 32 KB: two per CU)
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include <hip/hip_runtime.h>
#include <cmath>
#include "rscotr.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

namespace lab {

constexpr int HD = 32, NL = 4, NP = 4, NS = NL * NP, QB = 64;  // channels per head, levels, points, samples per (query, head), queries per block
__constant__ int c_cap[NL];                                      // window capacity per level, pixels

struct Geom {
  int W[NL], H[NL], start[NL];  // level width / height / first token
  int blk_first[NL + 1];        // first block index of a level (blocks of 8 x 8 queries, row-major)
  int blk_w[NL];                // blocks per row
  int Nk, Nq, heads;
};

struct Rec {        // one sample: four taps
  float w[4];       // bilinear weight x attention weight (0: outside the map)
  int a[4];         // >= 0: float offset of the pixel in the workgroup's window buffer; < 0: -(token index + 1), from global memory
};

__global__ __launch_bounds__(512) void msda_fwd_window_kernel(const float* __restrict__ value, const float* __restrict__ loc,
                                                              const float* __restrict__ attn, float* __restrict__ out, Geom g,
                                                              unsigned long long* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int box[NL][4];    // x min, x max, y min, y max of the taps of a level
  __shared__ int win[NL][5];    // x0, y0, w, h, float offset of the level's window in wbuf
  Rec* recs = reinterpret_cast<Rec*>(smem);                               // [QB][NS]
  float* wbuf = reinterpret_cast<float*>(smem + QB * NS * sizeof(Rec));   // the windows, 32 floats per pixel
  const int tid = threadIdx.x;
  const int head = blockIdx.x % g.heads;
  int blk = blockIdx.x / g.heads;
  const int nblk = g.blk_first[NL];
  const int b = blk / nblk;
  blk -= b * nblk;
  int lq = 0;
  while (blk >= g.blk_first[lq + 1]) ++lq;
  const int bl = blk - g.blk_first[lq];
  const int by0 = (bl / g.blk_w[lq]) * 8, bx0 = (bl % g.blk_w[lq]) * 8;
  if (tid < NL * 4) box[tid >> 2][tid & 3] = (tid & 1) ? -1 : (1 << 30);
  __syncthreads();
  // ---- 1. sample set-up (two samples per thread), bounding boxes
  int tx0[2], ty0[2], tl[2];
  float fx[2], fy[2], aw[2];
  bool qok[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int s = tid * 2 + u, qi = s / NS, smp = s % NS, l = smp / NP;
    const int qy = by0 + (qi >> 3), qx = bx0 + (qi & 7);
    qok[u] = qy < g.H[lq] && qx < g.W[lq];
    tl[u] = l;
    tx0[u] = ty0[u] = 0; fx[u] = fy[u] = aw[u] = 0.f;
    if (qok[u]) {
      const long q = (long)b * g.Nq + g.start[lq] + qy * g.W[lq] + qx;
      const float2 xy = *reinterpret_cast<const float2*>(loc + ((q * g.heads + head) * NS + smp) * 2);
      aw[u] = attn[(q * g.heads + head) * NS + smp];
      const float px = xy.x * g.W[l] - 0.5f, py = xy.y * g.H[l] - 0.5f;
      const float flx = floorf(px), fly = floorf(py);
      tx0[u] = (int)flx; ty0[u] = (int)fly; fx[u] = px - flx; fy[u] = py - fly;
      const int xa = max(tx0[u], 0), xb = min(tx0[u] + 1, g.W[l] - 1), ya = max(ty0[u], 0), yb = min(ty0[u] + 1, g.H[l] - 1);
      if (xa <= xb && ya <= yb) {
        atomicMin(&box[l][0], xa); atomicMax(&box[l][1], xb); atomicMin(&box[l][2], ya); atomicMax(&box[l][3], yb);
      }
    }
  }
  __syncthreads();
  // ---- 2. windows: the box cut to the level's capacity around its centre
  if (tid < NL) {
    const int l = tid;
    int x0 = box[l][0], x1 = box[l][1], y0 = box[l][2], y1 = box[l][3];
    int w = x1 - x0 + 1, h = y1 - y0 + 1;
    if (w <= 0 || h <= 0) { w = h = 0; x0 = y0 = 0; }
    const int cap = c_cap[l];
    if (w * h > cap) {
      int side = 1;
      while ((side + 1) * (side + 1) <= cap) ++side;
      const int nw = min(w, max(side, cap / max(min(h, side), 1)));
      const int nh = min(h, cap / nw);
      x0 += (w - nw) / 2; y0 += (h - nh) / 2; w = nw; h = nh;
    }
    win[l][0] = x0; win[l][1] = y0; win[l][2] = w; win[l][3] = h;
  }
  __syncthreads();
  if (tid == 0) {
    int o = 0;
    for (int l = 0; l < NL; ++l) { win[l][4] = o; o += win[l][2] * win[l][3] * HD; }
  }
  __syncthreads();
  // ---- records
  unsigned in_lds = 0, in_glob = 0;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int s = tid * 2 + u, l = tl[u];
    Rec r;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int x = tx0[u] + (t & 1), y = ty0[u] + (t >> 1);
      const float wt = ((t & 1) ? fx[u] : 1.f - fx[u]) * ((t >> 1) ? fy[u] : 1.f - fy[u]) * aw[u];
      const bool ok = qok[u] && x >= 0 && x < g.W[l] && y >= 0 && y < g.H[l];
      r.w[t] = ok ? wt : 0.f;
      int a = -1;  // (token 0 with weight 0: harmless)
      if (ok) {
        const int wx = x - win[l][0], wy = y - win[l][1];
        if (wx >= 0 && wx < win[l][2] && wy >= 0 && wy < win[l][3]) { a = win[l][4] + (wy * win[l][2] + wx) * HD; ++in_lds; }
        else { a = -(g.start[l] + y * g.W[l] + x + 1); ++in_glob; }
      }
      r.a[t] = a;
    }
    recs[s] = r;
  }
  // ---- 3. the windows -> LDS
  const float* vb = value + ((long)b * g.Nk * g.heads + head) * HD;
  for (int l = 0; l < NL; ++l) {
    const int w = win[l][2], n = w * win[l][3] * 8;
    float* dst = wbuf + win[l][4];
    for (int i = tid; i < n; i += 512) {
      const int pix = i >> 3, c4 = i & 7;
      const int tok = g.start[l] + (win[l][1] + pix / w) * g.W[l] + win[l][0] + pix % w;
      *reinterpret_cast<float4*>(dst + pix * HD + c4 * 4) = *reinterpret_cast<const float4*>(vb + (long)tok * g.heads * HD + c4 * 4);
    }
  }
  __syncthreads();
  // ---- 4. gather: lane group = query, lane = four channels
  const int qi = tid >> 3, c4 = tid & 7;
  const int qy = by0 + (qi >> 3), qx = bx0 + (qi & 7);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
  for (int smp = 0; smp < NS; ++smp) {
    const Rec r = recs[qi * NS + smp];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float4 v;
      if (r.a[t] >= 0) v = *reinterpret_cast<const float4*>(wbuf + r.a[t] + c4 * 4);
      else v = *reinterpret_cast<const float4*>(vb + (long)(-r.a[t] - 1) * g.heads * HD + c4 * 4);
      acc.x = fmaf(r.w[t], v.x, acc.x); acc.y = fmaf(r.w[t], v.y, acc.y); acc.z = fmaf(r.w[t], v.z, acc.z); acc.w = fmaf(r.w[t], v.w, acc.w);
    }
  }
  if (qy < g.H[lq] && qx < g.W[lq]) {
    const long q = (long)b * g.Nq + g.start[lq] + qy * g.W[lq] + qx;
    *reinterpret_cast<float4*>(out + (q * g.heads + head) * HD + c4 * 4) = acc;
  }
  if (stats) {
    atomicAdd(&stats[0], (unsigned long long)in_lds);
    atomicAdd(&stats[1], (unsigned long long)in_glob);
  }
}

}  // namespace lab

int main(int argc, char** argv) {
  const bool init = argc < 2 || !strcmp(argv[1], "init");
  const float par = argc > 2 ? atof(argv[2]) : (init ? 0.3f : 0.05f);
  const int B = 2, H = 8, D = 32, L = 4, P = 4;
  const int64_t shp[8] = {64, 64, 32, 32, 16, 16, 8, 8};
  int64_t lsi_h[4];
  int Nk = 0;
  for (int l = 0; l < L; ++l) { lsi_h[l] = Nk; Nk += (int)(shp[2 * l] * shp[2 * l + 1]); }
  const int Nq = Nk;
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> value((size_t)B * Nk * H * D), loc((size_t)B * Nq * H * L * P * 2), attn((size_t)B * Nq * H * L * P);
  for (auto& v : value) v = nd(rng);
  for (int b = 0; b < B; ++b)
    for (int q = 0; q < Nq; ++q) {
      int l0 = 0, r = q;
      while (r >= shp[2 * l0] * shp[2 * l0 + 1]) { r -= (int)(shp[2 * l0] * shp[2 * l0 + 1]); ++l0; }
      const int y = r / (int)shp[2 * l0 + 1], x = r % (int)shp[2 * l0 + 1];
      const float rx = (x + 0.5f) / shp[2 * l0 + 1], ry = (y + 0.5f) / shp[2 * l0];
      for (int h = 0; h < H; ++h) {
        float s = 0.f;
        float* a = &attn[(((size_t)b * Nq + q) * H + h) * L * P];
        for (int i = 0; i < L * P; ++i) { a[i] = expf(nd(rng)); s += a[i]; }
        const float th = 2.f * 3.14159265f * h / H, dx = cosf(th), dy = sinf(th), dm = fmaxf(fabsf(dx), fabsf(dy));
        for (int i = 0; i < L * P; ++i) {
          a[i] /= s;
          float* xy = &loc[((((size_t)b * Nq + q) * H + h) * L * P + i) * 2];
          const int l = i / P, p = i % P;
          if (init) {  // mmcv MultiScaleDeformableAttention.init_weights: grid_init[h] * (p + 1) pixels of level l, plus noise (pixels)
            xy[0] = rx + (dx / dm * (p + 1) + par * nd(rng)) / shp[2 * l + 1];
            xy[1] = ry + (dy / dm * (p + 1) + par * nd(rng)) / shp[2 * l];
          } else {
            xy[0] = rx + par * nd(rng);
            xy[1] = ry + par * nd(rng);
          }
        }
      }
    }
  float *d_value, *d_loc, *d_attn, *d_ref, *d_win;
  int64_t *d_shp, *d_lsi;
  unsigned long long* d_stats;
  const size_t nout = (size_t)B * Nq * H * D;
  CK(hipMalloc(&d_value, value.size() * 4)); CK(hipMalloc(&d_loc, loc.size() * 4)); CK(hipMalloc(&d_attn, attn.size() * 4));
  CK(hipMalloc(&d_ref, nout * 4)); CK(hipMalloc(&d_win, nout * 4)); CK(hipMalloc(&d_shp, 64)); CK(hipMalloc(&d_lsi, 32)); CK(hipMalloc(&d_stats, 16));
  CK(hipMemcpy(d_value, value.data(), value.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_loc, loc.data(), loc.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_attn, attn.data(), attn.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_shp, shp, 64, hipMemcpyHostToDevice)); CK(hipMemcpy(d_lsi, lsi_h, 32, hipMemcpyHostToDevice));
  CK(hipMemset(d_stats, 0, 16)); CK(hipMemset(d_win, 0, nout * 4));

  lab::Geom g{};
  g.Nk = Nk; g.Nq = Nq; g.heads = H;
  int nb = 0;
  for (int l = 0; l < L; ++l) {
    g.H[l] = (int)shp[2 * l]; g.W[l] = (int)shp[2 * l + 1]; g.start[l] = (int)lsi_h[l];
    g.blk_first[l] = nb; g.blk_w[l] = (g.W[l] + 7) / 8;
    nb += g.blk_w[l] * ((g.H[l] + 7) / 8);
  }
  g.blk_first[L] = nb;
  // capacity: windows of a level-0 block at the init pattern are 18^2 / 14^2 / 12^2 / 11^2 = 785 pixels; a level-1 block wants 26^2 at level 0
  const int caps[4] = {argc > 3 ? atoi(argv[3]) : 400, argc > 4 ? atoi(argv[4]) : 256, argc > 5 ? atoi(argv[5]) : 196, argc > 6 ? atoi(argv[6]) : 144};
  CK(hipMemcpyToSymbol(HIP_SYMBOL(lab::c_cap), caps, sizeof(caps)));
  const size_t lds = lab::QB * lab::NS * sizeof(lab::Rec) + (size_t)(caps[0] + caps[1] + caps[2] + caps[3]) * lab::HD * 4;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(lab::msda_fwd_window_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int grid = B * nb * H;
  printf("encoder call: B = %d, %d tokens, %d workgroups of 512 (%d blocks x %d heads), %zu bytes of LDS each; locations: %s %.3f\n", B, Nk, grid,
         B * nb, H, lds, init ? "mmcv init pattern + noise (pixels)" : "reference + N(0, 1) x", par);

  if (rscotr_msda_fwd(d_value, d_shp, d_lsi, d_loc, d_attn, d_ref, B, Nk, Nq, H, D, L, P, nullptr)) { printf("rscotr_msda_fwd: %s\n", rscotr_last_error()); return 1; }
  hipLaunchKernelGGL(lab::msda_fwd_window_kernel, dim3(grid), dim3(512), lds, 0, d_value, d_loc, d_attn, d_win, g, d_stats);
  CK(hipDeviceSynchronize());
  std::vector<float> o_ref(nout), o_win(nout);
  unsigned long long st[2];
  CK(hipMemcpy(o_ref.data(), d_ref, nout * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(o_win.data(), d_win, nout * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(st, d_stats, 16, hipMemcpyDeviceToHost));
  double md = 0, mx = 0;
  for (size_t i = 0; i < nout; ++i) { md = fmax(md, fabs((double)o_ref[i] - o_win[i])); mx = fmax(mx, fabs((double)o_ref[i])); }
  printf("max |window - shipped| = %.3g (max |out| %.3g); taps from LDS %.1f %%, from global memory %.1f %%\n", md, mx,
         100.0 * st[0] / (double)(st[0] + st[1]), 100.0 * st[1] / (double)(st[0] + st[1]));

  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 50;
  float ms;
  for (int w = 0; w < 2; ++w) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) rscotr_msda_fwd(d_value, d_shp, d_lsi, d_loc, d_attn, d_ref, B, Nk, Nq, H, D, L, P, nullptr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    if (w) printf("shipped kernel   %.1f us per launch (back to back)\n", ms / reps * 1e3);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(lab::msda_fwd_window_kernel, dim3(grid), dim3(512), lds, 0, d_value, d_loc, d_attn, d_win, g, (unsigned long long*)nullptr);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    if (w) printf("window kernel    %.1f us per launch (back to back)\n", ms / reps * 1e3);
  }
  return 0;
}
