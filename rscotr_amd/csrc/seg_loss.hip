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

// grid = B*h*w low-resolution cells, 128 threads: lane = class; gathers
//   dlogit[b,c,cy,cx] = scale * sum over output pixels p touching the cell of  wt(p->cell) * (softmax_c(p) - [c == label_p])
// Every tap of such a pixel lies in the 3x3 cell neighbourhood of (cy, cx): the neighbourhood of all classes is
// staged once in LDS ([9][class], conflict-free) — read straight from the (C, h, w) planes each lane would touch a
// different 16 KB-strided plane on every one of the ~1000 taps of the pixel loop (1.85 ms at 2x100x64x64 -> 512^2).
constexpr int UCE_F = 24;  // staged footprint (output pixels per axis that can touch one cell); larger -> direct loads

__global__ __launch_bounds__(128) void upsample_ce_bwd_kernel(const float* __restrict__ logit,
                                                              const int64_t* __restrict__ label,
                                                              const float* __restrict__ lse,
                                                              const float* __restrict__ gscale, float* __restrict__ dlogit,
                                                              int B, int C, int h, int w, int H, int W, int ignore) {
  __shared__ float sN[9][128];
  // per-cell tables shared by all classes: labels / lse of the footprint pixels and the column interpolation
  // (the pixel loop below runs ~256 times per class lane: without them every iteration redoes the index arithmetic and
  // two global loads that are the same for all 100 lanes — 261 us at 2x100x64x64 -> 512^2)
  __shared__ int sLab[UCE_F * UCE_F];
  __shared__ float sLse[UCE_F * UCE_F];
  __shared__ float sWx[UCE_F], sXl0[UCE_F], sXl1[UCE_F];
  __shared__ int sKx0[UCE_F], sKx1[UCE_F];
  const int cell = blockIdx.x;
  const int cx = cell % w, cy = (cell / w) % h, b = cell / (w * h);
  const float sy = (float)h / (float)H, sx = (float)w / (float)W;
  // output rows / columns whose 2-tap footprint can include this cell: source coord in (c-1, c+1)
  const int y_lo = max(0, (int)floorf(((float)cy - 1.f + 0.5f) / sy - 0.5f));
  const int y_hi = min(H - 1, (int)ceilf(((float)cy + 1.f + 0.5f) / sy - 0.5f));
  const int x_lo = max(0, (int)floorf(((float)cx - 1.f + 0.5f) / sx - 0.5f));
  const int x_hi = min(W - 1, (int)ceilf(((float)cx + 1.f + 0.5f) / sx - 0.5f));
  const int ny = y_hi - y_lo + 1, nx = x_hi - x_lo + 1;
  const bool staged = ny <= UCE_F && nx <= UCE_F;  // uniform
  const float* base = logit + (long)b * C * h * w;
  const float scale = gscale[0];
  if (staged) {
    for (int i = threadIdx.x; i < ny * nx; i += 128) {
      const long p = ((long)b * H + y_lo + i / nx) * W + x_lo + i % nx;
      sLab[i] = (int)label[p];
      sLse[i] = lse[p];
    }
    for (int i = threadIdx.x; i < nx; i += 128) {
      const Interp ix = src_index(x_lo + i, sx, w);
      sWx[i] = (ix.i0 == cx ? ix.l0 : 0.f) + (ix.i1 == cx ? ix.l1 : 0.f);
      sXl0[i] = ix.l0; sXl1[i] = ix.l1;
      sKx0[i] = ix.i0 - cx + 1; sKx1[i] = ix.i1 - cx + 1;
    }
  }
  for (int c0 = 0; c0 < C; c0 += 128) {
    const int c = c0 + threadIdx.x;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      const int ny_ = cy + k / 3 - 1, nx_ = cx + k % 3 - 1;
      sN[k][threadIdx.x] = (c < C && ny_ >= 0 && ny_ < h && nx_ >= 0 && nx_ < w) ? base[(long)c * h * w + ny_ * w + nx_] : 0.f;
    }
    __syncthreads();
    if (c >= C) continue;
    float acc = 0.f;
    for (int y = y_lo; y <= y_hi; ++y) {
      const Interp iy = src_index(y, sy, h);
      const float wy = (iy.i0 == cy ? iy.l0 : 0.f) + (iy.i1 == cy ? iy.l1 : 0.f);
      if (wy == 0.f) continue;
      const int ky0 = (iy.i0 - cy + 1) * 3, ky1 = (iy.i1 - cy + 1) * 3;
      if (staged) {
        const int row = (y - y_lo) * nx;
        for (int xx = 0; xx < nx; ++xx) {
          const float wx = sWx[xx];
          if (wx == 0.f) continue;
          const int lab = sLab[row + xx];
          if (lab == ignore) continue;
          const int kx0 = sKx0[xx], kx1 = sKx1[xx];
          const float l0 = sXl0[xx], l1 = sXl1[xx];
          const float v = iy.l0 * (l0 * sN[ky0 + kx0][threadIdx.x] + l1 * sN[ky0 + kx1][threadIdx.x]) +
                          iy.l1 * (l0 * sN[ky1 + kx0][threadIdx.x] + l1 * sN[ky1 + kx1][threadIdx.x]);
          const float prob = __expf(v - sLse[row + xx]);
          acc += wy * wx * (prob - (c == lab ? 1.f : 0.f));
        }
      } else {
        for (int x = x_lo; x <= x_hi; ++x) {
          const Interp ix = src_index(x, sx, w);
          const float wx = (ix.i0 == cx ? ix.l0 : 0.f) + (ix.i1 == cx ? ix.l1 : 0.f);
          if (wx == 0.f) continue;
          const long p = ((long)b * H + y) * W + x;
          const long lab = label[p];
          if (lab == ignore) continue;
          const int kx0 = ix.i0 - cx + 1, kx1 = ix.i1 - cx + 1;
          const float v = iy.l0 * (ix.l0 * sN[ky0 + kx0][threadIdx.x] + ix.l1 * sN[ky0 + kx1][threadIdx.x]) +
                          iy.l1 * (ix.l0 * sN[ky1 + kx0][threadIdx.x] + ix.l1 * sN[ky1 + kx1][threadIdx.x]);
          const float prob = __expf(v - lse[p]);
          acc += wy * wx * (prob - (c == lab ? 1.f : 0.f));
        }
      }
    }
    dlogit[((long)b * C + c) * h * w + cy * w + cx] = acc * scale;
  }
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
  upsample_ce_bwd_kernel<<<B * h * w, 128, 0, (hipStream_t)stream>>>(logit, label, lse, grad_scale, dlogit, B, C, h, w, H,
                                                                   W, ignore_index);
  return check_launch("rscotr_upsample_ce_bwd");
}
