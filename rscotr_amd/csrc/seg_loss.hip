// Fused bilinear-upsample + cross-entropy (+ top-1 accuracy) for the seg head (gfx950).
//
// Replaces mmseg BaseDecodeHead.losses as reached from models/multi/seg_head/mask2former_head.py:204:
//   seg_logit = resize(seg_logit (B,C,h,w), size=label.shape, mode='bilinear', align_corners=False)
//   loss_ce   = F.cross_entropy(seg_logit, label, ignore_index=255, reduction='none').mean()   (all pixels)
//   acc_seg   = accuracy(seg_logit, label)   (top-1 over the non-ignored pixels, in percent)
// The reference materialises the upsampled logits — (2,100,512,512) fp32 = 210 MB written, read by
// log-softmax, NLL, arg-max, and again in backward.  Here a pixel's C interpolated logits only ever
// exist in registers: forward reads the (B,C,h,w) logits (3.3 MB) and the labels, writes one
// log-sum-exp per pixel (for backward) and three scalars; backward is a GATHER per low-resolution
// cell (no atomics): lane = class, loop over the <= (2*sy)x(2*sx) output pixels the cell touches.
//
// PyTorch's upsample_bilinear2d (align_corners=False) source index: s = max((d + 0.5) * in/out - 0.5, 0),
// i0 = floor(s), i1 = min(i0 + 1, in - 1), lambda1 = s - i0.
#include "common.h"

namespace rscotr {

struct Interp {
  int i0, i1;
  float l0, l1;
};

__device__ __forceinline__ Interp src_index(int d, float scale, int in) {
  Interp r;
  const float s = fmaxf(((float)d + 0.5f) * scale - 0.5f, 0.f);
  r.i0 = min((int)s, in - 1);
  r.i1 = min(r.i0 + 1, in - 1);
  r.l1 = s - (float)r.i0;
  r.l0 = 1.f - r.l1;
  return r;
}

// one thread per output pixel; classes streamed (online log-sum-exp + running arg-max)
__global__ __launch_bounds__(256) void upsample_ce_fwd_kernel(const float* __restrict__ logit,
                                                              const int64_t* __restrict__ label,
                                                              float* __restrict__ lse, float* __restrict__ sums, int B,
                                                              int C, int h, int w, int H, int W, int ignore) {
  const long npix = (long)B * H * W;
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  float loss = 0.f, correct = 0.f, valid = 0.f;
  for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < npix; p += (long)gridDim.x * 256) {
    const int x = (int)(p % W), y = (int)((p / W) % H), b = (int)(p / ((long)W * H));
    const Interp iy = src_index(y, sy, h), ix = src_index(x, sx, w);
    const float w00 = iy.l0 * ix.l0, w01 = iy.l0 * ix.l1, w10 = iy.l1 * ix.l0, w11 = iy.l1 * ix.l1;
    const float* base = logit + (long)b * C * h * w;
    const int o00 = iy.i0 * w + ix.i0, o01 = iy.i0 * w + ix.i1, o10 = iy.i1 * w + ix.i0, o11 = iy.i1 * w + ix.i1;
    const long lab = label[p];
    float m = -3.0e38f, s = 0.f, best = -3.0e38f, at_label = 0.f;
    int arg = 0;
    for (int c = 0; c < C; ++c) {
      const float* pc = base + (long)c * h * w;
      const float v = w00 * pc[o00] + w01 * pc[o01] + w10 * pc[o10] + w11 * pc[o11];
      if (v > best) { best = v; arg = c; }
      if (c == lab) at_label = v;
      if (v > m) { s = s * __expf(m - v) + 1.f; m = v; }
      else s += __expf(v - m);
    }
    const float l = m + __logf(s);
    lse[p] = l;
    if (lab != ignore) {
      loss += l - at_label;
      valid += 1.f;
      correct += (arg == (int)lab) ? 1.f : 0.f;
    }
  }
  loss = wave_sum(loss);
  correct = wave_sum(correct);
  valid = wave_sum(valid);
  __shared__ float red[4][3];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { red[wv][0] = loss; red[wv][1] = correct; red[wv][2] = valid; }
  __syncthreads();
  if (threadIdx.x < 3)  // one partial row per workgroup, folded in fixed order by upsample_ce_sums_kernel (no atomics)
    sums[(long)blockIdx.x * 3 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ __launch_bounds__(256) void upsample_ce_sums_kernel(const float* __restrict__ part, float* __restrict__ sums,
                                                               int nparts) {
  float a[3] = {0.f, 0.f, 0.f};
  for (int i = threadIdx.x; i < nparts; i += 256)
#pragma unroll
    for (int k = 0; k < 3; ++k) a[k] += part[(long)i * 3 + k];
  __shared__ float red[4][3];
#pragma unroll
  for (int k = 0; k < 3; ++k) a[k] = wave_sum(a[k]);
  if ((threadIdx.x & 63) == 0)
#pragma unroll
    for (int k = 0; k < 3; ++k) red[threadIdx.x >> 6][k] = a[k];
  __syncthreads();
  if (threadIdx.x < 3) sums[threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// Backward: dlogit[b,c,cy,cx] = scale * sum over output pixels p touching the cell of  wt(p->cell) * (softmax_c(p) - [c == label_p]).
// Round 5: one workgroup per BLOCK of UCE_TB x UCE_TB low-resolution cells instead of one per cell.  The output pixels whose
// 2 x 2 tap footprint meets the block — (UCE_TB + 1) * 8 = 40 per axis at the step's x8 resize, 1.56 x the pixels the block
// owns — are visited ONCE per class (the per-cell kernel visited every pixel from each of the four cells it touches and
// re-interpolated its logits each time: 245 us at 2 x 100 x 64 x 64 -> 512^2).  512 threads = 4 row groups x 128 class lanes:
// group g takes the footprint rows y_lo + g, + 4, ...; a lane keeps the four cell sums of the current (row pair, column
// pair) in registers and folds them into its private LDS column when the column pair changes (every ~8 pixels); the
// (UCE_TB + 2)^2 cell neighbourhood of all classes, the labels and log-sum-exps of the footprint and the row / column
// interpolation tables are staged in LDS once.  The four groups meet in fixed order: no atomics, bit-reproducible.
constexpr int UCE_TB = 4;    // cells per block edge
constexpr int UCE_NB = UCE_TB + 2;
constexpr int UCE_FP = 48;   // staged footprint rows / columns (40 at x8); larger footprints read labels / lse from global memory

__global__ __launch_bounds__(512) void upsample_ce_bwd_kernel(const float* __restrict__ logit,
                                                              const int64_t* __restrict__ label,
                                                              const float* __restrict__ lse,
                                                              const float* __restrict__ gscale, float* __restrict__ dlogit,
                                                              int B, int C, int h, int w, int H, int W, int ignore) {
  extern __shared__ __attribute__((aligned(16))) float uce_smem[];
  float* sN = uce_smem;                                  // [UCE_NB * UCE_NB][128]   neighbourhood logits of the 128 classes of a pass
  float* sAcc = sN + UCE_NB * UCE_NB * 128;              // [4 groups][UCE_TB * UCE_TB][128]
  int2* sLL = reinterpret_cast<int2*>(sAcc + 4 * UCE_TB * UCE_TB * 128);  // [UCE_FP * UCE_FP] {label, bits of lse * log2 e} of a footprint pixel
  float* sYl = reinterpret_cast<float*>(sLL + UCE_FP * UCE_FP);    // [UCE_FP][2]  l0, l1 of a footprint row
  int* sYi = reinterpret_cast<int*>(sYl + 2 * UCE_FP);   // [UCE_FP][2]  i0, i1 relative to the block's first cell - 1
  int4* sXT = reinterpret_cast<int4*>(sYi + 2 * UCE_FP);  // [UCE_FP] {i0, i1 (same origin), bits of l0, l1} of a footprint column
  const int bw = (w + UCE_TB - 1) / UCE_TB, bh = (h + UCE_TB - 1) / UCE_TB;
  const int blk = blockIdx.x;
  const int cx0 = (blk % bw) * UCE_TB, cy0 = ((blk / bw) % bh) * UCE_TB, b = blk / (bw * bh);
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  // output rows / columns whose taps can include a cell of the block: source coordinate in (c0 - 1, c0 + TB)
  const int y_lo = max(0, (int)floorf(((float)cy0 - 1.f + 0.5f) / sy - 0.5f));
  const int y_hi = min(H - 1, (int)ceilf(((float)(cy0 + UCE_TB) + 0.5f) / sy - 0.5f));
  const int x_lo = max(0, (int)floorf(((float)cx0 - 1.f + 0.5f) / sx - 0.5f));
  const int x_hi = min(W - 1, (int)ceilf(((float)(cx0 + UCE_TB) + 0.5f) / sx - 0.5f));
  const int ny = y_hi - y_lo + 1, nx = x_hi - x_lo + 1;
  const bool staged = ny <= UCE_FP && nx <= UCE_FP;  // uniform
  const int tid = threadIdx.x, lane = tid & 127, grp = tid >> 7;
  const float* base = logit + (long)b * C * h * w;
  const float scale = gscale[0];
  if (staged) {
    for (int i = tid; i < ny * nx; i += 512) {
      const long p = ((long)b * H + y_lo + i / nx) * W + x_lo + i % nx;
      sLL[i] = make_int2((int)label[p], __float_as_int(lse[p] * 1.4426950408889634f));  // (the staged path works in the log2 domain: one v_exp_f32 per pixel and class)
    }
  }
  for (int i = tid; i < min(ny, UCE_FP); i += 512) {
    const Interp iy = src_index(y_lo + i, sy, h);
    sYl[2 * i] = iy.l0; sYl[2 * i + 1] = iy.l1;
    sYi[2 * i] = iy.i0 - cy0 + 1; sYi[2 * i + 1] = iy.i1 - cy0 + 1;
  }
  for (int i = tid; i < min(nx, UCE_FP); i += 512) {
    const Interp ix = src_index(x_lo + i, sx, w);
    sXT[i] = make_int4(ix.i0 - cx0 + 1, ix.i1 - cx0 + 1, __float_as_int(ix.l0), __float_as_int(ix.l1));
  }
  for (int c0 = 0; c0 < C; c0 += 128) {
    const int c = c0 + lane;
    __syncthreads();
    // neighbourhood (cells cy0 - 1 .. cy0 + TB, cx0 - 1 .. cx0 + TB; zeros outside the map: never weighted) of this pass's classes
    for (int k = grp; k < UCE_NB * UCE_NB; k += 4) {
      const int ny_ = cy0 + k / UCE_NB - 1, nx_ = cx0 + k % UCE_NB - 1;
      sN[k * 128 + lane] = (c < C && ny_ >= 0 && ny_ < h && nx_ >= 0 && nx_ < w) ? base[(long)c * h * w + ny_ * w + nx_] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < UCE_TB * UCE_TB; ++k) sAcc[(grp * UCE_TB * UCE_TB + k) * 128 + lane] = 0.f;
    __syncthreads();
    if (c < C && staged) {
      // Staged footprints (the step's shapes).  Everything that depends on the pixel alone — tap indices and weights, label,
      // log-sum-exp — is the same for the 64 lanes of a wavefront (a row group is two whole wavefronts): read to scalar
      // registers.  The footprint columns are walked one COLUMN PAIR (first tap on block column k - 1, k = 0 .. TB) at a time,
      // the pair loop unrolled: the vector work per pixel and class is the interpolation (2 FMAs on the row-combined cell
      // logits A0 / A1 of the pair), one exp2, the one-hot and two FMAs into the pair's sums s0 / s1, which land in the row's
      // column sums cs[] by static register index; the row weights multiply cs once per row and only then touch the thread's
      // LDS cells (8 read-add-writes per footprint row; the first cut folded 4 per column pair: 151 -> 118 us at
      // 2 x 100 x 64 x 64 -> 512^2, this form: see profiles/README.md)
      float* acc = sAcc + grp * UCE_TB * UCE_TB * 128 + lane;
      auto sc = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
      auto scf = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
      int xs[UCE_TB + 2];  // xs[k] = first footprint column whose first tap is block column >= k - 1 (columns ascend with xx)
#pragma unroll
      for (int k = 0; k < UCE_TB + 2; ++k) xs[k] = 0;
      for (int xx = 0; xx < nx; ++xx) {
        const int kx0 = sc(sXT[xx].x);
#pragma unroll
        for (int k = 0; k < UCE_TB + 2; ++k) xs[k] += kx0 < k ? 1 : 0;
      }
      for (int r = grp; r < ny; r += 4) {
        const int ky0 = sc(sYi[2 * r]), ky1 = sc(sYi[2 * r + 1]);
        const float yl0 = scf(sYl[2 * r]), yl1 = scf(sYl[2 * r + 1]);
        const bool in0 = ky0 >= 1 && ky0 <= UCE_TB && cy0 + ky0 - 1 < h, in1 = ky1 >= 1 && ky1 <= UCE_TB && cy0 + ky1 - 1 < h;
        if (!in0 && !in1) continue;
        const int b0 = min(max(ky0, 0), UCE_NB - 1), b1 = min(max(ky1, 0), UCE_NB - 1);
        const int2* llr = sLL + r * nx;
        float cs[UCE_TB + 2];  // column sums of this row by block column + 1 (0 and TB + 1: the halo, dropped)
#pragma unroll
        for (int k = 0; k < UCE_TB + 2; ++k) cs[k] = 0.f;
#pragma unroll
        for (int k = 0; k <= UCE_TB; ++k) {  // pairs (k, k + 1); at the far edge of the map (k, k)
          const int x0 = xs[k], x1 = xs[k + 1];
          if (x0 == x1) continue;  // (uniform)
          const int kx1 = sc(sXT[x0].y);
          const int a1 = min(max(kx1, 0), UCE_NB - 1);
          const float n00 = sN[(b0 * UCE_NB + k) * 128 + lane], n01 = sN[(b0 * UCE_NB + a1) * 128 + lane];
          const float n10 = sN[(b1 * UCE_NB + k) * 128 + lane], n11 = sN[(b1 * UCE_NB + a1) * 128 + lane];
          const float A0 = (yl0 * n00 + yl1 * n10) * 1.4426950408889634f;
          const float A1 = (yl0 * n01 + yl1 * n11) * 1.4426950408889634f;
          float s0 = 0.f, s1 = 0.f;
          // one 16-byte column record + one 8-byte pixel record per step, the next step's pair requested before this one is used
          int4 xt_n = sXT[x0];
          int2 ll_n = llr[x0];
          for (int xx = x0; xx < x1; ++xx) {
            const int4 xt = xt_n;
            const int2 ll = ll_n;
            const int xn = min(xx + 1, nx - 1);
            xt_n = sXT[xn];
            ll_n = llr[xn];
            const int lab = sc(ll.x);
            if (lab == ignore) continue;
            const float xl0 = __int_as_float(sc(xt.z)), xl1 = __int_as_float(sc(xt.w));
            const float ls = __int_as_float(sc(ll.y));
            const float gq = __builtin_amdgcn_exp2f(fmaf(xl1, A1, fmaf(xl0, A0, -ls))) - (c == lab ? 1.f : 0.f);
            s0 = fmaf(xl0, gq, s0);
            s1 = fmaf(xl1, gq, s1);
          }
          cs[k] += s0;
          if (kx1 == k) cs[k] += s1; else cs[k + 1] += s1;  // (uniform)
        }
#pragma unroll
        for (int k = 1; k <= UCE_TB; ++k) {
          if (cx0 + k - 1 < w) {
            if (in0) acc[((ky0 - 1) * UCE_TB + k - 1) * 128] += yl0 * cs[k];
            if (in1) acc[((ky1 - 1) * UCE_TB + k - 1) * 128] += yl1 * cs[k];
          }
        }
      }
    } else if (c < C) {
      float* acc = sAcc + grp * UCE_TB * UCE_TB * 128 + lane;
      for (int r = grp; r < ny; r += 4) {
        int ky0, ky1;
        float yl0, yl1;
        if (r < UCE_FP) { ky0 = sYi[2 * r]; ky1 = sYi[2 * r + 1]; yl0 = sYl[2 * r]; yl1 = sYl[2 * r + 1]; }
        else { const Interp iy = src_index(y_lo + r, sy, h); ky0 = iy.i0 - cy0 + 1; ky1 = iy.i1 - cy0 + 1; yl0 = iy.l0; yl1 = iy.l1; }
        // (a row whose both taps lie outside the block contributes nothing; the footprint bounds are conservative)
        const bool in0 = ky0 >= 1 && ky0 <= UCE_TB && cy0 + ky0 - 1 < h, in1 = ky1 >= 1 && ky1 <= UCE_TB && cy0 + ky1 - 1 < h;
        if (!in0 && !in1) continue;
        float t00 = 0.f, t01 = 0.f, t10 = 0.f, t11 = 0.f;
        int cur0 = -100, cur1 = -100;
        auto flush = [&]() {
          if (cur0 == -100) return;
          const bool jx0 = cur0 >= 1 && cur0 <= UCE_TB && cx0 + cur0 - 1 < w, jx1 = cur1 >= 1 && cur1 <= UCE_TB && cx0 + cur1 - 1 < w;
          if (in0 && jx0) acc[((ky0 - 1) * UCE_TB + cur0 - 1) * 128] += t00;
          if (in0 && jx1) acc[((ky0 - 1) * UCE_TB + cur1 - 1) * 128] += t01;
          if (in1 && jx0) acc[((ky1 - 1) * UCE_TB + cur0 - 1) * 128] += t10;
          if (in1 && jx1) acc[((ky1 - 1) * UCE_TB + cur1 - 1) * 128] += t11;
          t00 = t01 = t10 = t11 = 0.f;
        };
        float n00 = 0.f, n01 = 0.f, n10 = 0.f, n11 = 0.f;
        for (int xx = 0; xx < nx; ++xx) {
          int kx0, kx1;
          float xl0, xl1;
          if (xx < UCE_FP) { const int4 xt = sXT[xx]; kx0 = xt.x; kx1 = xt.y; xl0 = __int_as_float(xt.z); xl1 = __int_as_float(xt.w); }
          else { const Interp ix = src_index(x_lo + xx, sx, w); kx0 = ix.i0 - cx0 + 1; kx1 = ix.i1 - cx0 + 1; xl0 = ix.l0; xl1 = ix.l1; }
          if (kx0 != cur0 || kx1 != cur1) {  // a new column pair: fold the finished one, fetch this one's four cell logits
            flush();
            cur0 = kx0; cur1 = kx1;
            const int a0 = min(max(kx0, 0), UCE_NB - 1), a1 = min(max(kx1, 0), UCE_NB - 1);
            const int b0 = min(max(ky0, 0), UCE_NB - 1), b1 = min(max(ky1, 0), UCE_NB - 1);
            n00 = sN[(b0 * UCE_NB + a0) * 128 + lane]; n01 = sN[(b0 * UCE_NB + a1) * 128 + lane];
            n10 = sN[(b1 * UCE_NB + a0) * 128 + lane]; n11 = sN[(b1 * UCE_NB + a1) * 128 + lane];
          }
          int lab;
          float ls;
          { const long p = ((long)b * H + y_lo + r) * W + x_lo + xx; lab = (int)label[p]; ls = lse[p]; }  // (footprints past the staged size)
          if (lab == ignore) continue;
          const float v = yl0 * (xl0 * n00 + xl1 * n01) + yl1 * (xl0 * n10 + xl1 * n11);
          const float gq = __expf(v - ls) - (c == lab ? 1.f : 0.f);
          t00 += yl0 * xl0 * gq; t01 += yl0 * xl1 * gq; t10 += yl1 * xl0 * gq; t11 += yl1 * xl1 * gq;
        }
        flush();
      }
    }
    __syncthreads();
    // the four row groups meet in fixed order; thread (grp, lane) writes block rows grp of class lane
    if (c < C && cy0 + grp < h) {
      float* drow = dlogit + ((long)b * C + c) * h * w + (long)(cy0 + grp) * w + cx0;
#pragma unroll
      for (int k = 0; k < UCE_TB; ++k) {
        if (cx0 + k < w) {
          float v = 0.f;
#pragma unroll
          for (int g2 = 0; g2 < 4; ++g2) v += sAcc[((g2 * UCE_TB + grp) * UCE_TB + k) * 128 + lane];
          drow[k] = v * scale;
        }
      }
    }
  }
}

constexpr size_t uce_bwd_lds_bytes() {
  return (size_t)(UCE_NB * UCE_NB * 128 + 4 * UCE_TB * UCE_TB * 128 + 2 * UCE_FP * UCE_FP + 8 * UCE_FP) * 4;
}

// Masked-attention mask of the Mask2Former-style decoder (models/multi/seg_head/mask2former_head.py:126-136 and
// :177-178): mask logits (rows, h, w) -> bilinear resize to (th, tw) (align_corners=False) -> sigmoid < 0.5 ->
// rows that came out all-True are reset to all-False -> bool (rows, th*tw).  One workgroup per (image, query) row;
// replaces interpolate + sigmoid + compare + all + and-not (5 launches, 10 times per seg step).
__global__ __launch_bounds__(256) void seg_attn_mask_kernel(const float* __restrict__ pred, unsigned char* __restrict__ out,
                                                            int h, int w, int th, int tw) {
  __shared__ int s_any_false;
  const float* src = pred + (long)blockIdx.x * h * w;
  unsigned char* dst = out + (long)blockIdx.x * th * tw;
  const float sy = (float)h / (float)th, sx = (float)w / (float)tw;
  if (threadIdx.x == 0) s_any_false = 0;
  __syncthreads();
  bool any_false = false;
  for (int i = threadIdx.x; i < th * tw; i += 256) {
    const Interp iy = src_index(i / tw, sy, h), ix = src_index(i % tw, sx, w);
    const float v = iy.l0 * (ix.l0 * src[iy.i0 * w + ix.i0] + ix.l1 * src[iy.i0 * w + ix.i1]) +
                    iy.l1 * (ix.l0 * src[iy.i1 * w + ix.i0] + ix.l1 * src[iy.i1 * w + ix.i1]);
    const bool blocked = 1.f / (1.f + expf(-v)) < 0.5f;
    dst[i] = blocked ? 1 : 0;
    any_false |= !blocked;
  }
  if (any_false) s_any_false = 1;  // benign race: every writer stores 1
  __syncthreads();
  if (!s_any_false)
    for (int i = threadIdx.x; i < th * tw; i += 256) dst[i] = 0;
}

}  // namespace rscotr

using namespace rscotr;

extern "C" int rscotr_seg_attn_mask(const float* mask_pred, unsigned char* out, int rows, int h, int w, int th, int tw,
                                    void* stream) {
  if (rows < 0 || h <= 0 || w <= 0 || th <= 0 || tw <= 0) return fail(RSCOTR_E_SHAPE, "rscotr_seg_attn_mask: bad shape");
  if (rows == 0) return RSCOTR_OK;
  if (!mask_pred || !out) return fail(RSCOTR_E_ARG, "rscotr_seg_attn_mask: null pointer");
  seg_attn_mask_kernel<<<rows, 256, 0, (hipStream_t)stream>>>(mask_pred, out, h, w, th, tw);
  return check_launch("rscotr_seg_attn_mask");
}


// sums[3] = {sum of per-pixel CE over non-ignored pixels, #correct, #non-ignored}; lse (B,H,W) saved.
constexpr int UCE_MAX_WG = 4096;
extern "C" int64_t rscotr_upsample_ce_workspace(void) { return (int64_t)UCE_MAX_WG * 3 * 4; }

extern "C" int rscotr_upsample_ce_fwd(const float* logit, const int64_t* label, float* lse, float* sums, int B,
                                      int C, int h, int w, int H, int W, int ignore_index, float* workspace,
                                      int64_t workspace_bytes, void* stream) {
  if (B < 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0)
    return fail(RSCOTR_E_SHAPE, "rscotr_upsample_ce_fwd: bad shape");
  if (!sums) return fail(RSCOTR_E_ARG, "rscotr_upsample_ce_fwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (B == 0) {
    hipMemsetAsync(sums, 0, 3 * sizeof(float), s);
    return RSCOTR_OK;
  }
  if (!logit || !label || !lse) return fail(RSCOTR_E_ARG, "rscotr_upsample_ce_fwd: null pointer");
  if (!workspace || workspace_bytes < rscotr_upsample_ce_workspace())
    return fail(RSCOTR_E_ARG, "rscotr_upsample_ce_fwd: workspace of rscotr_upsample_ce_workspace() bytes required");
  const long npix = (long)B * H * W;
  const int nwg = (int)std::min<long>((npix + 255) / 256, UCE_MAX_WG);
  upsample_ce_fwd_kernel<<<nwg, 256, 0, s>>>(logit, label, lse, workspace, B, C, h, w, H, W, ignore_index);
  upsample_ce_sums_kernel<<<1, 256, 0, s>>>(workspace, sums, nwg);
  return check_launch("rscotr_upsample_ce_fwd");
}

// dlogit (B,C,h,w) = grad_scale[0] * d(sum of per-pixel CE)/d(logit); grad_scale is a DEVICE scalar
// (upstream gradient / number of pixels), so no host sync is needed.
extern "C" int rscotr_upsample_ce_bwd(const float* logit, const int64_t* label, const float* lse,
                                      const float* grad_scale, float* dlogit, int B, int C, int h, int w, int H,
                                      int W, int ignore_index, void* stream) {
  if (B < 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0)
    return fail(RSCOTR_E_SHAPE, "rscotr_upsample_ce_bwd: bad shape");
  if (B == 0) return RSCOTR_OK;
  if (!logit || !label || !lse || !grad_scale || !dlogit) return fail(RSCOTR_E_ARG, "rscotr_upsample_ce_bwd: null pointer");
  static_assert(UCE_TB == 4, "four row groups write the four cell rows of a block");
  static const bool attr_set = [] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(upsample_ce_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)uce_bwd_lds_bytes());
    return true;
  }();
  (void)attr_set;
  const int nblk = B * ((h + UCE_TB - 1) / UCE_TB) * ((w + UCE_TB - 1) / UCE_TB);
  upsample_ce_bwd_kernel<<<nblk, 512, uce_bwd_lds_bytes(), (hipStream_t)stream>>>(logit, label, lse, grad_scale, dlogit, B, C, h,
                                                                                 w, H, W, ignore_index);
  return check_launch("rscotr_upsample_ce_bwd");
}
