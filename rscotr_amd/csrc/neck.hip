// ChannelMapper pieces in TOKEN layout (B, H*W, C) for gfx950: GroupNorm forward/backward and the
// im2col / col2im gathers that turn the 3x3 stride-2 "extra" convolution into an MFMA GEMM.
//
// Replaces, for mmdet's ChannelMapper neck (configs/multi/MTL_slvlcls_...potsdam.py:26-33; called from
// models/multi/multitask_learner.py:84 on backbone_feature[-3:]):
//   convs[i]       = Conv2d(in_i, 256, 1, bias=False) -> GroupNorm(32, 256)      (i = 0, 1, 2)
//   extra_convs[0] = Conv2d(768, 256, 3, stride=2, padding=1, bias=False) -> GroupNorm(32, 256)
// The reference runs them through cuDNN/MIOpen on NCHW maps; the backbone here produces tokens and
// every consumer of the neck flattens the maps back to tokens, so the 1x1 convolutions ARE the
// fp32 MFMA GEMM (rscotr_gemm_f32) on the token matrix, the 3x3/s2 convolution is the same GEMM on
// a (B*Ho*Wo, 9*C) gathered matrix with K ordered (c, ky, kx) like the Conv2d weight, and
// GroupNorm normalises 8-channel groups over all tokens of an image without leaving token layout.
//
// GroupNorm mapping: one wavefront per token row per step (C/4 lanes hold the row as float4, a
// group of 8 channels = 2 adjacent lanes); per-(image, group) sums are lane-pair + LDS reductions
// into one partial row per workgroup (caller workspace), folded in fixed order by the finalize
// kernel — no atomics: the statistics, hence the whole forward pass, are bit-reproducible; a second
// pass applies the affine transform.  All HBM-streaming: forward reads x twice and writes y once.
#include "common.h"

namespace rscotr {

// ---- GroupNorm statistics: sums[b][g] = {sum f, sum f*h} over the tokens and channels of group g -----
// MODE 0: f = x,        h = x      (forward: mean / variance)
// MODE 1: f = gamma*dy, h = xhat   (backward: the two projections) and per-channel dgamma/dbeta
template <int MODE>
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                       const float* __restrict__ gamma,
                                                       const float* __restrict__ mean_rstd, float* __restrict__ part,
                                                       int L, int C, int G, int tokens_per_block, long dy_bstride = 0) {
  // part: [B][chunks][2 G + 2 C] = {sum f, sum f h} per group, then (MODE 1) dgamma | dbeta partials per channel
  const int LT = C >> 2;            // lanes per token row
  const int TPW = kWave / LT;       // token rows per wavefront step
  const int gs4 = (C / G) >> 2;     // lanes per group
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % LT, slot = lane / LT;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * tokens_per_block, t1 = min(L, t0 + tokens_per_block);
  const int g = sub / gs4;
  float4 gm = make_float4(1.f, 1.f, 1.f, 1.f);
  float mu = 0.f, rs = 0.f;
  if (MODE == 1) {
    if (gamma) gm = reinterpret_cast<const float4*>(gamma)[sub];
    mu = mean_rstd[((long)b * G + g) * 2];
    rs = mean_rstd[((long)b * G + g) * 2 + 1];
  }
  float s1 = 0.f, s2 = 0.f;
  float4 ag = make_float4(0.f, 0.f, 0.f, 0.f), ab = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int t = t0 + wave * TPW + slot; t < t1; t += 4 * TPW) {
    const long off = ((long)b * L + t) * C;
    const float4 xv = reinterpret_cast<const float4*>(x + off)[sub];
    if (MODE == 0) {
      s1 += xv.x + xv.y + xv.z + xv.w;
      s2 += xv.x * xv.x + xv.y * xv.y + xv.z * xv.z + xv.w * xv.w;
    } else {
      const float4 d = reinterpret_cast<const float4*>(dy + (long)b * dy_bstride + (long)t * C)[sub];
      const float4 xh = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
      s1 += gm.x * d.x + gm.y * d.y + gm.z * d.z + gm.w * d.w;
      s2 += gm.x * d.x * xh.x + gm.y * d.y * xh.y + gm.z * d.z * xh.z + gm.w * d.w * xh.w;
      ag.x += d.x * xh.x; ag.y += d.y * xh.y; ag.z += d.z * xh.z; ag.w += d.w * xh.w;
      ab.x += d.x; ab.y += d.y; ab.z += d.z; ab.w += d.w;
    }
  }
  // lanes of one group (gs4 adjacent lanes), then token slots, then wavefronts
  for (int o = 1; o < gs4; o <<= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  for (int o = LT; o < kWave; o <<= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
    if (MODE == 1) {
      ag.x += __shfl_xor(ag.x, o, 64); ag.y += __shfl_xor(ag.y, o, 64);
      ag.z += __shfl_xor(ag.z, o, 64); ag.w += __shfl_xor(ag.w, o, 64);
      ab.x += __shfl_xor(ab.x, o, 64); ab.y += __shfl_xor(ab.y, o, 64);
      ab.z += __shfl_xor(ab.z, o, 64); ab.w += __shfl_xor(ab.w, o, 64);
    }
  }
  __shared__ float red[4][64][10];
  if (slot == 0) {
    float* r = red[wave][sub];
    r[0] = s1; r[1] = s2;
    r[2] = ag.x; r[3] = ag.y; r[4] = ag.z; r[5] = ag.w;
    r[6] = ab.x; r[7] = ab.y; r[8] = ab.z; r[9] = ab.w;
  }
  __syncthreads();
  if (wave == 0 && slot == 0) {
    float v[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) v[k] = red[0][sub][k] + red[1][sub][k] + red[2][sub][k] + red[3][sub][k];
    float* row = part + ((long)b * gridDim.x + blockIdx.x) * (2 * G + 2 * C);
    if (sub % gs4 == 0) {
      row[2 * g] = v[0];
      row[2 * g + 1] = v[1];
    }
    if (MODE == 1) {
      for (int k = 0; k < 4; ++k) row[2 * G + sub * 4 + k] = v[2 + k];
      for (int k = 0; k < 4; ++k) row[2 * G + C + sub * 4 + k] = v[6 + k];
    }
  }
}

// partial rows -> (mean, rstd): fixed summation order over the chunks
__global__ void gn_finalize_kernel(const float* __restrict__ part, float* __restrict__ mean_rstd, int B, int G, int C,
                                   int chunks, float inv_count, float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * G) return;
  const int b = i / G, g = i - b * G;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
  for (int c = 0; c < chunks; ++c) {
    const float* row = part + ((long)b * chunks + c) * (2 * G + 2 * C);
    s1 += row[2 * g];
    s2 += row[2 * g + 1];
  }
  const float mu = s1 * inv_count;
  const float var = fmaxf(s2 * inv_count - mu * mu, 0.f);
  mean_rstd[2 * i] = mu;
  mean_rstd[2 * i + 1] = rsqrtf(var + eps);
}

// backward: proj[b][g] = the two projections; dgamma / dbeta (+)= their per-channel sums over images and chunks (fixed order)
__global__ void gn_finalize_bwd_kernel(const float* __restrict__ part, float* __restrict__ proj, float* __restrict__ dgamma,
                                       float* __restrict__ dbeta, int B, int G, int C, int chunks) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * G) {
    const int b = i / G, g = i - b * G;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 8
    for (int c = 0; c < chunks; ++c) {
      const float* row = part + ((long)b * chunks + c) * (2 * G + 2 * C);
      s1 += row[2 * g];
      s2 += row[2 * g + 1];
    }
    proj[2 * i] = s1;
    proj[2 * i + 1] = s2;
  }
  if (i < 2 * C) {  // i < C: dgamma[i], else dbeta[i - C]
    float* dst = i < C ? dgamma : dbeta;
    if (dst) {
      float s = 0.f;
#pragma unroll 8
      for (int r = 0; r < B * chunks; ++r) s += part[(long)r * (2 * G + 2 * C) + 2 * G + i];
      dst[i < C ? i : i - C] += s;
    }
  }
}

// MODE 0: y = (x - mean) * rstd * gamma + beta
// MODE 1: dx = rstd * (gamma*dy - p1/n - xhat * p2/n),  proj[b][g] = {p1, p2}
template <int MODE>
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ mean_rstd,
                                                       const float* __restrict__ proj, float* __restrict__ out, int L,
                                                       int C, int G, float inv_count, long dy_bstride4 = 0) {
  const int C4 = C >> 2, gs4 = (C / G) >> 2;
  const long total = (long)gridDim.y * L * C4;  // gridDim.y = B
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)L * C4; i += (long)gridDim.x * 256) {
    const int b = blockIdx.y;
    const int c4 = (int)(i % C4);
    const int g = c4 / gs4;
    const long off = (long)b * L * C4 + i;
    const float mu = mean_rstd[((long)b * G + g) * 2], rs = mean_rstd[((long)b * G + g) * 2 + 1];
    const float4 xv = reinterpret_cast<const float4*>(x)[off];
    const float4 gm = gamma ? reinterpret_cast<const float4*>(gamma)[c4] : make_float4(1.f, 1.f, 1.f, 1.f);
    float4 o;
    if (MODE == 0) {
      const float4 bt = beta ? reinterpret_cast<const float4*>(beta)[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
      o.x = (xv.x - mu) * rs * gm.x + bt.x; o.y = (xv.y - mu) * rs * gm.y + bt.y;
      o.z = (xv.z - mu) * rs * gm.z + bt.z; o.w = (xv.w - mu) * rs * gm.w + bt.w;
    } else {
      const float4 d = reinterpret_cast<const float4*>(dy)[(long)b * dy_bstride4 + i];
      const float p1 = proj[((long)b * G + g) * 2] * inv_count, p2 = proj[((long)b * G + g) * 2 + 1] * inv_count;
      o.x = rs * (gm.x * d.x - p1 - (xv.x - mu) * rs * p2); o.y = rs * (gm.y * d.y - p1 - (xv.y - mu) * rs * p2);
      o.z = rs * (gm.z * d.z - p1 - (xv.z - mu) * rs * p2); o.w = rs * (gm.w * d.w - p1 - (xv.w - mu) * rs * p2);
    }
    reinterpret_cast<float4*>(out)[off] = o;
  }
  (void)total;
}

// ---- 3x3 stride-2 padding-1 gathers --------------------------------------------------------------
// col[(b*Ho + oy)*Wo + ox][c*9 + ky*3 + kx] = x[b][(2*oy-1+ky)*W + (2*ox-1+kx)][c]  (0 outside)
__global__ __launch_bounds__(256) void im2col3x3s2_kernel(const float* __restrict__ x, float* __restrict__ col, int B,
                                                          int H, int W, int C, int Ho, int Wo) {
  const long total = (long)B * Ho * Wo * C * 9;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int k = (int)(i % 9);
    long r = i / 9;
    const int c = (int)(r % C);
    r /= C;
    const int ox = (int)(r % Wo), oy = (int)((r / Wo) % Ho), b = (int)(r / ((long)Wo * Ho));
    const int y = 2 * oy - 1 + k / 3, xx = 2 * ox - 1 + k % 3;
    col[i] = (y >= 0 && y < H && xx >= 0 && xx < W) ? x[(((long)b * H + y) * W + xx) * C + c] : 0.f;
  }
}

// adjoint: dx[b][y*W + x][c] = sum over the (<= 4) output positions whose window covers (y, x)
__global__ __launch_bounds__(256) void col2im3x3s2_kernel(const float* __restrict__ dcol, float* __restrict__ dx, int B,
                                                          int H, int W, int C, int Ho, int Wo) {
  const long total = (long)B * H * W * C;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c = (int)(i % C);
    long r = i / C;
    const int xx = (int)(r % W), y = (int)((r / W) % H), b = (int)(r / ((long)W * H));
    float acc = 0.f;
    for (int ky = 0; ky < 3; ++ky) {
      const int ty = y + 1 - ky;
      if (ty < 0 || (ty & 1)) continue;
      const int oy = ty >> 1;
      if (oy >= Ho) continue;
      for (int kx = 0; kx < 3; ++kx) {
        const int tx = xx + 1 - kx;
        if (tx < 0 || (tx & 1)) continue;
        const int ox = tx >> 1;
        if (ox >= Wo) continue;
        acc += dcol[((((long)b * Ho + oy) * Wo + ox) * C + c) * 9 + ky * 3 + kx];
      }
    }
    dx[i] = acc;
  }
}

static int gn_check(const char* fn, int B, int L, int C, int G) {
  if (B < 0 || L < 0 || C <= 0 || G <= 0 || C % G != 0)
    return fail(RSCOTR_E_SHAPE, "%s: bad shape B=%d L=%d C=%d G=%d", fn, B, L, C, G);
  const int lt = C / 4, gs = C / G;
  if (C % 4 != 0 || (lt != 16 && lt != 32 && lt != 64) || gs % 4 != 0 || ((gs / 4) & (gs / 4 - 1)) != 0)
    return fail(RSCOTR_E_SHAPE, "%s: C=%d must be 64, 128 or 256 with a power-of-two multiple of 4 channels per group", fn, C);
  return RSCOTR_OK;
}

static dim3 gn_stats_grid(int B, int L, int* tpb) {
  // ~32 tokens per workgroup, at most 128 chunks per image: 2 x 128 workgroups on the largest level (with 32 chunks of
  // 128 tokens the 64 workgroups of a (2, 4096, 256) level ran 16-28 us, latency-bound on a quarter of the chip)
  int chunks = std::max(1, std::min((L + 31) / 32, 128));
  *tpb = (L + chunks - 1) / chunks;
  chunks = (L + *tpb - 1) / *tpb;
  return dim3(chunks, B);
}

}  // namespace rscotr

using namespace rscotr;

extern "C" int64_t rscotr_groupnorm_tokens_workspace(int B, int L, int C, int G) {
  if (B <= 0 || L <= 0 || C <= 0 || G <= 0) return 0;
  int tpb;
  const dim3 grid = gn_stats_grid(B, L, &tpb);
  return (int64_t)B * grid.x * (2 * G + 2 * C) * 4;
}

extern "C" int rscotr_groupnorm_tokens_fwd(const float* x, const float* weight, const float* bias, float* y,
                                           float* mean_rstd, int B, int L, int C, int G, float eps, float* workspace,
                                           int64_t workspace_bytes, void* stream) {
  if (int e = gn_check("rscotr_groupnorm_tokens_fwd", B, L, C, G)) return e;
  if (B == 0 || L == 0) return RSCOTR_OK;
  if (!x || !y || !mean_rstd) return fail(RSCOTR_E_ARG, "rscotr_groupnorm_tokens_fwd: null pointer");
  if (!aligned16(x) || !aligned16(y) || (weight && !aligned16(weight)) || (bias && !aligned16(bias)))
    return fail(RSCOTR_E_ALIGN, "rscotr_groupnorm_tokens_fwd: pointers must be 16-byte aligned");
  if (!workspace || workspace_bytes < rscotr_groupnorm_tokens_workspace(B, L, C, G))
    return fail(RSCOTR_E_ARG, "rscotr_groupnorm_tokens_fwd: workspace of rscotr_groupnorm_tokens_workspace() bytes required");
  hipStream_t s = (hipStream_t)stream;
  int tpb;
  const dim3 grid = gn_stats_grid(B, L, &tpb);
  gn_stats_kernel<0><<<grid, 256, 0, s>>>(x, nullptr, nullptr, nullptr, workspace, L, C, G, tpb);
  const float inv = 1.f / ((float)L * (float)(C / G));
  gn_finalize_kernel<<<(B * G + 255) / 256, 256, 0, s>>>(workspace, mean_rstd, B, G, C, (int)grid.x, inv, eps);
  const int ax = (int)std::min<long>(((long)L * (C / 4) + 255) / 256, 1024);
  gn_apply_kernel<0><<<dim3(ax, B), 256, 0, s>>>(x, nullptr, weight, bias, mean_rstd, nullptr, y, L, C, G, inv);
  return check_launch("rscotr_groupnorm_tokens_fwd");
}

extern "C" int rscotr_groupnorm_tokens_bwd(const float* dy, const float* x, const float* weight,
                                           const float* mean_rstd, float* dx, float* dweight, float* dbias,
                                           float* proj_ws, int B, int L, int C, int G, int64_t dy_batch_stride,
                                           float* workspace, int64_t workspace_bytes, void* stream) {
  if (int e = gn_check("rscotr_groupnorm_tokens_bwd", B, L, C, G)) return e;
  if (B == 0 || L == 0) return RSCOTR_OK;
  if (dy_batch_stride < (int64_t)L * C || (dy_batch_stride & 3))
    return fail(RSCOTR_E_SHAPE, "rscotr_groupnorm_tokens_bwd: dy batch stride %lld must be >= L * C and a multiple of 4",
                (long long)dy_batch_stride);
  if (!dy || !x || !mean_rstd || !dx || !proj_ws) return fail(RSCOTR_E_ARG, "rscotr_groupnorm_tokens_bwd: null pointer");
  if (!aligned16(dy) || !aligned16(x) || !aligned16(dx) || (weight && !aligned16(weight)))
    return fail(RSCOTR_E_ALIGN, "rscotr_groupnorm_tokens_bwd: pointers must be 16-byte aligned");
  if (!workspace || workspace_bytes < rscotr_groupnorm_tokens_workspace(B, L, C, G))
    return fail(RSCOTR_E_ARG, "rscotr_groupnorm_tokens_bwd: workspace of rscotr_groupnorm_tokens_workspace() bytes required");
  hipStream_t s = (hipStream_t)stream;
  int tpb;
  const dim3 grid = gn_stats_grid(B, L, &tpb);
  gn_stats_kernel<1><<<grid, 256, 0, s>>>(x, dy, weight, mean_rstd, workspace, L, C, G, tpb, (long)dy_batch_stride);
  gn_finalize_bwd_kernel<<<(std::max(B * G, 2 * C) + 255) / 256, 256, 0, s>>>(workspace, proj_ws, dweight, dbias, B, G, C,
                                                                              (int)grid.x);
  const float inv = 1.f / ((float)L * (float)(C / G));
  const int ax = (int)std::min<long>(((long)L * (C / 4) + 255) / 256, 1024);
  gn_apply_kernel<1><<<dim3(ax, B), 256, 0, s>>>(x, dy, weight, nullptr, mean_rstd, proj_ws, dx, L, C, G, inv,
                                                 (long)(dy_batch_stride / 4));
  return check_launch("rscotr_groupnorm_tokens_bwd");
}

extern "C" int rscotr_im2col3x3s2_tokens(const float* x, float* col, int B, int H, int W, int C, void* stream) {
  if (B < 0 || H <= 0 || W <= 0 || C <= 0) return fail(RSCOTR_E_SHAPE, "rscotr_im2col3x3s2_tokens: bad shape");
  if (B == 0) return RSCOTR_OK;
  if (!x || !col) return fail(RSCOTR_E_ARG, "rscotr_im2col3x3s2_tokens: null pointer");
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long total = (long)B * Ho * Wo * C * 9;
  im2col3x3s2_kernel<<<(int)std::min<long>((total + 255) / 256, 4096), 256, 0, (hipStream_t)stream>>>(x, col, B, H, W, C,
                                                                                                 Ho, Wo);
  return check_launch("rscotr_im2col3x3s2_tokens");
}

extern "C" int rscotr_col2im3x3s2_tokens(const float* dcol, float* dx, int B, int H, int W, int C, void* stream) {
  if (B < 0 || H <= 0 || W <= 0 || C <= 0) return fail(RSCOTR_E_SHAPE, "rscotr_col2im3x3s2_tokens: bad shape");
  if (B == 0) return RSCOTR_OK;
  if (!dcol || !dx) return fail(RSCOTR_E_ARG, "rscotr_col2im3x3s2_tokens: null pointer");
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long total = (long)B * H * W * C;
  col2im3x3s2_kernel<<<(int)std::min<long>((total + 255) / 256, 4096), 256, 0, (hipStream_t)stream>>>(dcol, dx, B, H, W, C,
                                                                                                 Ho, Wo);
  return check_launch("rscotr_col2im3x3s2_tokens");
}
