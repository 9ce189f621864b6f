// Detection loss arithmetic of the DINO head as three kernels (each replaces a few dozen element-wise launches):
//   match_cost_kernel   mmdet FocalLossCost + BBoxL1Cost(xywh) + IoUCost(giou) of HungarianAssigner.assign
//                       (cfg ...potsdam.py:169-174; reached from models/multi/bbox_head/mmdet_detr_head/detr_head.py:513-515)
//   focal_sum_kernel    mmcv sigmoid_focal_loss summed per prediction set (detr_head.py:384-385, dino_head.py:272-273)
//   box_loss_kernel     L1 (cxcywh, normalised) and GIoU (xyxy, pixels) sums per set (detr_head.py:392-415)
// The two loss kernels also write the gradient of their sums with respect to the predictions (the upstream
// gradient of a set's sum is one scalar, applied afterwards), so backward needs no second pass over the formulas.
// One workgroup per prediction set and a fixed reduction order: deterministic sums.
#include "common.h"

namespace rscotr {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float powg(float x, float gamma) { return gamma == 2.f ? x * x : powf(x, gamma); }

// deterministic block sum (256 threads): returns the total in every thread
__device__ __forceinline__ float block_sum256(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

struct Giou {
  float giou;
  float d[4];  // d giou / d (x1, y1, x2, y2) of the first box
};

// GIoU of p against t (both xyxy), eps as mmdet bbox_overlaps; optionally its gradient with respect to p
template <bool GRAD>
__device__ __forceinline__ Giou giou_xyxy(const float* p, const float* t, float eps) {
  Giou r;
  const float pw = p[2] - p[0], ph = p[3] - p[1];
  const float a1 = pw * ph, a2 = (t[2] - t[0]) * (t[3] - t[1]);
  const float ix1 = fmaxf(p[0], t[0]), iy1 = fmaxf(p[1], t[1]), ix2 = fminf(p[2], t[2]), iy2 = fminf(p[3], t[3]);
  const float iw = fmaxf(ix2 - ix1, 0.f), ih = fmaxf(iy2 - iy1, 0.f);
  const float ov = iw * ih;
  const float uraw = a1 + a2 - ov;
  const float u = fmaxf(uraw, eps);
  const float ex1 = fminf(p[0], t[0]), ey1 = fminf(p[1], t[1]), ex2 = fmaxf(p[2], t[2]), ey2 = fmaxf(p[3], t[3]);
  const float ew = fmaxf(ex2 - ex1, 0.f), eh = fmaxf(ey2 - ey1, 0.f);
  const float earaw = ew * eh;
  const float ea = fmaxf(earaw, eps);
  r.giou = ov / u - (ea - u) / ea;
  if (GRAD) {
    // d(ov), d(a1), d(earaw) with respect to (x1, y1, x2, y2)
    const float wpos = ix2 - ix1 > 0.f ? 1.f : 0.f, hpos = iy2 - iy1 > 0.f ? 1.f : 0.f;
    const float diw[4] = {p[0] > t[0] ? -wpos : 0.f, 0.f, p[2] < t[2] ? wpos : 0.f, 0.f};
    const float dih[4] = {0.f, p[1] > t[1] ? -hpos : 0.f, 0.f, p[3] < t[3] ? hpos : 0.f};
    const float ewpos = ex2 - ex1 > 0.f ? 1.f : 0.f, ehpos = ey2 - ey1 > 0.f ? 1.f : 0.f;
    const float dew[4] = {p[0] < t[0] ? -ewpos : 0.f, 0.f, p[2] > t[2] ? ewpos : 0.f, 0.f};
    const float deh[4] = {0.f, p[1] < t[1] ? -ehpos : 0.f, 0.f, p[3] > t[3] ? ehpos : 0.f};
    const float da1[4] = {-ph, -pw, ph, pw};
    const float ulive = uraw > eps ? 1.f : 0.f, elive = earaw > eps ? 1.f : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float dov = diw[k] * ih + iw * dih[k];
      const float du = ulive * (da1[k] - dov);
      const float dea = elive * (dew[k] * eh + ew * deh[k]);
      // giou = ov/u - 1 + u/ea
      r.d[k] = (dov * u - ov * du) / (u * u) + (du * ea - u * dea) / (ea * ea);
    }
  }
  return r;
}

// one thread per (s, b, q, g)
__global__ __launch_bounds__(256) void match_cost_kernel(const float* __restrict__ cls, const float* __restrict__ box,
                                                         const float* __restrict__ gt_box, const long* __restrict__ gt_lab,
                                                         const float* __restrict__ factors, float* __restrict__ cost,
                                                         int S, int B, int Q, int C, int G, float w_cls, float w_l1,
                                                         float w_iou, float alpha, float gamma, float eps) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)S * B * Q * G;
  if (i >= total) return;
  const int g = (int)(i % G);
  const long sbq = i / G;
  const int b = (int)((sbq / Q) % B);
  const float* f = factors + b * 4;
  const float* gb = gt_box + ((long)b * G + g) * 4;
  const long lab = gt_lab[(long)b * G + g];
  const float p = sigmoidf_(cls[sbq * C + lab]);
  const float neg = -logf(1.f - p + eps) * (1.f - alpha) * powg(p, gamma);
  const float pos = -logf(p + eps) * alpha * powg(1.f - p, gamma);
  const float c_cls = (pos - neg) * w_cls;
  const float* pb = box + sbq * 4;
  const float n0 = gb[0] / f[0], n1 = gb[1] / f[1], n2 = gb[2] / f[2], n3 = gb[3] / f[3];
  const float gc[4] = {(n0 + n2) / 2.f, (n1 + n3) / 2.f, n2 - n0, n3 - n1};
  const float c_l1 = (fabsf(pb[0] - gc[0]) + fabsf(pb[1] - gc[1]) + fabsf(pb[2] - gc[2]) + fabsf(pb[3] - gc[3])) * w_l1;
  const float px[4] = {(pb[0] - 0.5f * pb[2]) * f[0], (pb[1] - 0.5f * pb[3]) * f[1], (pb[0] + 0.5f * pb[2]) * f[2],
                       (pb[1] + 0.5f * pb[3]) * f[3]};
  const float c_iou = -giou_xyxy<false>(px, gb, 1e-6f).giou * w_iou;
  cost[i] = c_cls + c_l1 + c_iou;
}

// grid = S sets; sums[s] = sum over (n, c) of weight[s,n] * focal(pred[s,n,c], target[s,n]); dpred = d sums / d pred
// (1024 threads per set: a set is 36 000 logits with an exp and a log each — 256 threads took 51 us for the 7 sets of a det
// iteration, a launch that runs alone on 7 CUs; the 16 wavefront sums meet in fixed order)
constexpr int FOCAL_THREADS = 1024;
__global__ __launch_bounds__(FOCAL_THREADS) void focal_sum_kernel(const float* __restrict__ pred, const long* __restrict__ target,
                                                                  const float* __restrict__ weight, float* __restrict__ sums,
                                                                  float* __restrict__ dpred, int N, int C, float gamma, float alpha) {
  __shared__ float red[FOCAL_THREADS / 64];
  const int s = blockIdx.x;
  const float tiny = 1.17549435e-38f;
  float acc = 0.f;
  const long base = (long)s * N * C;
  for (long e = threadIdx.x; e < (long)N * C; e += FOCAL_THREADS) {
    const long n = e / C;
    const int c = (int)(e - n * C);
    const float w = weight ? weight[(long)s * N + n] : 1.f;
    const float x = pred[base + e];
    const float p = sigmoidf_(x);
    const bool is_pos = target[(long)s * N + n] == c;
    float l, d;
    if (is_pos) {
      const float lp = logf(fmaxf(p, tiny)), q = powg(1.f - p, gamma);
      l = -alpha * q * lp;
      d = alpha * q * (gamma * p * lp - (1.f - p));
    } else {
      const float lq = logf(fmaxf(1.f - p, tiny)), q = powg(p, gamma);
      l = -(1.f - alpha) * q * lq;
      d = (1.f - alpha) * q * (p - gamma * (1.f - p) * lq);
    }
    acc += w * l;
    dpred[base + e] = w * d;
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < FOCAL_THREADS / 64; ++w) t += red[w];
    sums[s] = t;
  }
}

// grid = S sets over the B*Q boxes of a set: sums[0][s] = sum |pred - target| * weight (cxcywh, normalised);
// sums[1][s] = sum (1 - GIoU(pred_xyxy * f, target_xyxy * f)) * mean(weight); d_l1 / d_giou = their gradients wrt pred
__global__ __launch_bounds__(256) void box_loss_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                       const float* __restrict__ weight, const float* __restrict__ factors,
                                                       float* __restrict__ sums, float* __restrict__ d_l1,
                                                       float* __restrict__ d_giou, int S, int B, int Q, float eps) {
  __shared__ float red[4];
  const int s = blockIdx.x;
  float a_l1 = 0.f, a_gi = 0.f;
  for (int e = threadIdx.x; e < B * Q; e += 256) {
    const int b = e / Q;
    const long o = ((long)s * B * Q + e) * 4;
    const float* f = factors + b * 4;
    const float* p = pred + o;
    const float* t = target + o;
    const float* w = weight + o;
    float dl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float df = p[k] - t[k];
      a_l1 += fabsf(df) * w[k];
      dl[k] = (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) * w[k];
    }
    const float px[4] = {(p[0] - 0.5f * p[2]) * f[0], (p[1] - 0.5f * p[3]) * f[1], (p[0] + 0.5f * p[2]) * f[2],
                         (p[1] + 0.5f * p[3]) * f[3]};
    const float tx[4] = {(t[0] - 0.5f * t[2]) * f[0], (t[1] - 0.5f * t[3]) * f[1], (t[0] + 0.5f * t[2]) * f[2],
                         (t[1] + 0.5f * t[3]) * f[3]};
    const float wm = (w[0] + w[1] + w[2] + w[3]) * 0.25f;
    const Giou g = giou_xyxy<true>(px, tx, eps);
    a_gi += (1.f - g.giou) * wm;
    // d(1 - giou)/d(cx, cy, w, h) through x1 = (cx - w/2) f0, y1 = (cy - h/2) f1, x2 = (cx + w/2) f2, y2 = (cy + h/2) f3
    const float gx1 = -g.d[0] * wm * f[0], gy1 = -g.d[1] * wm * f[1], gx2 = -g.d[2] * wm * f[2], gy2 = -g.d[3] * wm * f[3];
    float4 dg = make_float4(gx1 + gx2, gy1 + gy2, 0.5f * (gx2 - gx1), 0.5f * (gy2 - gy1));
    *reinterpret_cast<float4*>(d_l1 + o) = make_float4(dl[0], dl[1], dl[2], dl[3]);
    *reinterpret_cast<float4*>(d_giou + o) = dg;
  }
  const float t1 = block_sum256(a_l1, red);
  const float t2 = block_sum256(a_gi, red);
  if (threadIdx.x == 0) {
    sums[s] = t1;
    sums[S + s] = t2;
  }
}

// Iterative box refinement of Deformable-DETR / DINO: out = sigmoid(delta + inverse_sigmoid(ref, eps))
// (models/multi/bbox_head/transformer.py:112-118 in the decoder, dino_head.py:133-137 in the head), with
// inverse_sigmoid(x) = log(max(clamp(x,0,1), eps) / max(1 - clamp(x,0,1), eps)).  One launch instead of eight.
__global__ __launch_bounds__(256) void refine_box_fwd_kernel(const float* __restrict__ delta, const float* __restrict__ ref,
                                                             float* __restrict__ out, long n, float eps) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = fminf(fmaxf(ref[i], 0.f), 1.f);
  const float inv = logf(fmaxf(x, eps) / fmaxf(1.f - x, eps));
  out[i] = sigmoidf_(delta[i] + inv);
}

// d_delta = g * out * (1 - out); d_ref = d_delta * d inverse_sigmoid / d ref (zero where a clamp is active)
__global__ __launch_bounds__(256) void refine_box_bwd_kernel(const float* __restrict__ g, const float* __restrict__ out,
                                                             const float* __restrict__ ref, float* __restrict__ d_delta,
                                                             float* __restrict__ d_ref, long n, float eps) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float o = out[i];
  const float d = g[i] * o * (1.f - o);
  if (d_delta) d_delta[i] = d;
  if (d_ref) {
    const float r = ref[i];
    float di = 0.f;
    if (r >= 0.f && r <= 1.f) {  // inside the outer clamp
      if (r >= eps) di += 1.f / r;                 // d log(max(x, eps))
      if (1.f - r >= eps) di += 1.f / (1.f - r);   // -d log(max(1 - x, eps))
    }
    d_ref[i] = d * di;
  }
}

}  // namespace rscotr

using namespace rscotr;

extern "C" int rscotr_refine_box_fwd(const float* delta, const float* ref, float* out, int64_t n, float eps, void* stream) {
  if (n < 0) return fail(RSCOTR_E_SHAPE, "rscotr_refine_box_fwd: negative size");
  if (n == 0) return RSCOTR_OK;
  if (!delta || !ref || !out) return fail(RSCOTR_E_ARG, "rscotr_refine_box_fwd: null pointer");
  refine_box_fwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(delta, ref, out, n, eps);
  return check_launch("rscotr_refine_box_fwd");
}

extern "C" int rscotr_refine_box_bwd(const float* grad_out, const float* out, const float* ref, float* d_delta, float* d_ref,
                                     int64_t n, float eps, void* stream) {
  if (n < 0) return fail(RSCOTR_E_SHAPE, "rscotr_refine_box_bwd: negative size");
  if (n == 0) return RSCOTR_OK;
  if (!grad_out || !out || !ref) return fail(RSCOTR_E_ARG, "rscotr_refine_box_bwd: null pointer");
  refine_box_bwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(grad_out, out, ref, d_delta, d_ref, n, eps);
  return check_launch("rscotr_refine_box_bwd");
}

extern "C" int rscotr_match_cost(const float* cls, const float* box, const float* gt_box, const int64_t* gt_lab,
                                 const float* factors, float* cost, int S, int B, int Q, int C, int G, float w_cls,
                                 float w_l1, float w_iou, float alpha, float gamma, float eps, void* stream) {
  if (S < 0 || B < 0 || Q < 0 || C <= 0 || G < 0) return fail(RSCOTR_E_SHAPE, "rscotr_match_cost: bad shape");
  const long total = (long)S * B * Q * G;
  if (total == 0) return RSCOTR_OK;
  if (!cls || !box || !gt_box || !gt_lab || !factors || !cost) return fail(RSCOTR_E_ARG, "rscotr_match_cost: null pointer");
  match_cost_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      cls, box, gt_box, (const long*)gt_lab, factors, cost, S, B, Q, C, G, w_cls, w_l1, w_iou, alpha, gamma, eps);
  return check_launch("rscotr_match_cost");
}

extern "C" int rscotr_focal_sum(const float* pred, const int64_t* target, const float* weight, float* sums, float* dpred,
                                int S, int N, int C, float gamma, float alpha, void* stream) {
  if (S < 0 || N < 0 || C <= 0) return fail(RSCOTR_E_SHAPE, "rscotr_focal_sum: bad shape");
  if (S == 0) return RSCOTR_OK;
  if (!pred || !target || !sums || !dpred) return fail(RSCOTR_E_ARG, "rscotr_focal_sum: null pointer");
  focal_sum_kernel<<<S, FOCAL_THREADS, 0, (hipStream_t)stream>>>(pred, (const long*)target, weight, sums, dpred, N, C, gamma, alpha);
  return check_launch("rscotr_focal_sum");
}

extern "C" int rscotr_box_loss(const float* pred, const float* target, const float* weight, const float* factors,
                               float* sums, float* d_l1, float* d_giou, int S, int B, int Q, float eps, void* stream) {
  if (S < 0 || B < 0 || Q < 0) return fail(RSCOTR_E_SHAPE, "rscotr_box_loss: bad shape");
  if (S == 0) return RSCOTR_OK;
  if (!pred || !target || !weight || !factors || !sums || !d_l1 || !d_giou)
    return fail(RSCOTR_E_ARG, "rscotr_box_loss: null pointer");
  if (!aligned16(d_l1) || !aligned16(d_giou)) return fail(RSCOTR_E_ALIGN, "rscotr_box_loss: gradient buffers must be 16-byte aligned");
  box_loss_kernel<<<S, 256, 0, (hipStream_t)stream>>>(pred, target, weight, factors, sums, d_l1, d_giou, S, B, Q, eps);
  return check_launch("rscotr_box_loss");
}
