// Two-stage proposal selection of the DINO transformer (gfx950): models/multi/bbox_head/transformer.py:226-241 —
//   topk_proposals = torch.topk(enc_outputs_class.max(-1)[0], topk, dim=1)[1]
//   topk_coords_unact = torch.gather(enc_outputs_coord_unact, 1, topk_proposals[..., None].repeat(1, 1, 4))
//   topk_anchor = topk_coords_unact.sigmoid();  topk_score = torch.gather(enc_outputs_class, 1, ...)
// with enc_outputs_coord_unact = reg_branch(output_memory) + output_proposals — a row maximum, a top-k (two library launches:
// radix select + sort), an add, two gathers and a sigmoid forward; two zero-fills, two scatter-adds and a sigmoid backward in
// backward: ~14 launches, ~150 us of a det iteration.  Here: ONE workgroup of 1024 threads per image forward, one launch over
// the (B, N) rows backward.
//
// Forward, per image: (1) the row maxima become order-preserving 32-bit keys in LDS (N <= 36864: every pyramid of the configs,
// up to 1024 x 1024 inputs); (2) MSD radix select, four 8-bit passes of LDS histograms, finds the K-th largest key T, the
// number of keys above it and how many keys EQUAL to T are still needed; (3) one pass in index order collects the winners
// (keys > T, then the first `need` keys == T: ties go to the lower index — torch.topk leaves that order unspecified);
// (4) bitonic sort of the (key, ~index) pairs, descending: torch.topk(sorted=True) order; (5) the gathers, the proposal add
// and the sigmoid for the K winners, plus inv[b, n] = rank of token n or -1 for the backward.
// Backward: d(class)[b, n, :] = d(score)[b, inv, :] or 0, d(reg)[b, n, :] = d(anchor)[b, inv, :] * a * (1 - a) or 0 — every
// row written exactly once (no zero-fill, no atomics: bit-reproducible).
#include "common.h"

namespace rscotr {

constexpr int SEL_THREADS = 1024;
constexpr int SEL_MAX_N = 36864;  // keys in LDS: 144 KB
constexpr int SEL_MAX_K = 1024;

__device__ __forceinline__ unsigned order_key(float v) {  // larger float <-> larger unsigned; -0 < +0; NaN sorts high
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(SEL_THREADS) void det_proposals_kernel(
    const float* __restrict__ cls, const float* __restrict__ raw, const float* __restrict__ prop, long prop_bstride,
    long long* __restrict__ out_idx, float* __restrict__ out_score, float* __restrict__ out_unact,
    float* __restrict__ out_anchor, int* __restrict__ inv, int N, int C, int K) {
  extern __shared__ __attribute__((aligned(16))) unsigned sel_lds[];
  unsigned* keys = sel_lds;                                                       // [N]
  unsigned long long* cand = reinterpret_cast<unsigned long long*>(sel_lds + ((N + 3) & ~3));  // [1024] (key << 32 | ~idx)
  __shared__ int hist[256];
  __shared__ int wsum[16][2];
  __shared__ unsigned s_prefix;
  __shared__ int s_need, s_above;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x;
  const float* cb = cls + (long)b * N * C;

  for (int n = tid; n < N; n += SEL_THREADS) {  // (1) row maxima (enc_outputs_class.max(-1)[0])
    const float* r = cb + (long)n * C;
    // torch.max propagates NaN and torch.topk ranks it above everything (fmaxf would drop it: a row holding a NaN logit
    // would be ranked by its finite entries and the eager fallback for N > 36 864 would disagree): a NaN row keeps NaN, and
    // its key is the canonical quiet NaN, which order_key puts above +inf whatever sign the payload had
    float m = r[0];
    bool nan = m != m;
    for (int c = 1; c < C; ++c) { nan = nan || r[c] != r[c]; m = fmaxf(m, r[c]); }
    keys[n] = order_key(nan ? __uint_as_float(0x7fc00000u) : m);
    inv[(long)b * N + n] = -1;
  }
  if (tid == 0) { s_prefix = 0u; s_need = K; s_above = 0; }
  __syncthreads();

  // (2) radix select from the most significant byte down: after pass p the K-th largest key is known to start with
  // s_prefix (its top 8 (p + 1) bits), `s_above` keys are larger than anything with that prefix, `s_need` = K - s_above
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned prefix = s_prefix, hmask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    for (int n = tid; n < N; n += SEL_THREADS) {
      const unsigned k = keys[n];
      if ((k & hmask) == prefix) atomicAdd(&hist[(k >> shift) & 255], 1);
    }
    __syncthreads();
    if (tid == 0) {  // (256 bins: a serial walk from the top costs nothing next to the passes over N)
      int need = s_need, bin = 255;
      for (; bin > 0; --bin) {
        if (hist[bin] >= need) break;
        need -= hist[bin];
      }
      s_above += s_need - need;
      s_need = need;
      s_prefix = prefix | ((unsigned)bin << shift);
    }
    __syncthreads();
  }
  const unsigned T = s_prefix;     // the K-th largest key
  const int need_eq = s_need;      // keys == T still to take (lowest indices first)
  const int n_above = s_above;     // keys > T

  // (3) winners in index order: positions [0, n_above) for keys > T, [n_above, K) for the first need_eq keys == T
  for (int i = tid; i < SEL_MAX_K; i += SEL_THREADS) cand[i] = 0ull;  // padding sorts last (key 0 < every real key)
  __syncthreads();
  int base_gt = 0, base_eq = 0;
  for (int n0 = 0; n0 < N; n0 += SEL_THREADS) {
    const int n = n0 + tid;
    const unsigned k = n < N ? keys[n] : 0u;
    const bool gt = n < N && k > T, eq = n < N && k == T;
    const unsigned long long mg = __ballot(gt), me = __ballot(eq);
    if (lane == 0) { wsum[wave][0] = __popcll(mg); wsum[wave][1] = __popcll(me); }
    __syncthreads();
    int og = 0, oe = 0, tg = 0, te = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const int a = wsum[w][0], c = wsum[w][1];
      if (w < wave) { og += a; oe += c; }
      tg += a; te += c;
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    if (gt) cand[base_gt + og + __popcll(mg & below)] = ((unsigned long long)k << 32) | (unsigned)(0xffffffffu - (unsigned)n);
    if (eq) {
      const int r = base_eq + oe + __popcll(me & below);
      if (r < need_eq) cand[n_above + r] = ((unsigned long long)k << 32) | (unsigned)(0xffffffffu - (unsigned)n);
    }
    base_gt += tg; base_eq += te;
    __syncthreads();
  }

  // (4) bitonic sort of 1024 pairs, descending (one element per thread)
  for (int size = 2; size <= SEL_MAX_K; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const int partner = tid ^ stride;
      if (partner > tid) {
        const unsigned long long a = cand[tid], c = cand[partner];
        const bool desc = (tid & size) == 0;
        if (desc ? a < c : a > c) { cand[tid] = c; cand[partner] = a; }
      }
      __syncthreads();
    }
  }

  // (5) outputs
  for (int k = tid; k < K; k += SEL_THREADS) {
    const int n = (int)(0xffffffffu - (unsigned)(cand[k] & 0xffffffffull));
    out_idx[(long)b * K + k] = n;
    inv[(long)b * N + n] = k;
    const float4 r = *reinterpret_cast<const float4*>(raw + ((long)b * N + n) * 4);
    const float4 p = *reinterpret_cast<const float4*>(prop + b * prop_bstride + (long)n * 4);
    const float4 u = make_float4(r.x + p.x, r.y + p.y, r.z + p.z, r.w + p.w);
    *reinterpret_cast<float4*>(out_unact + ((long)b * K + k) * 4) = u;
    *reinterpret_cast<float4*>(out_anchor + ((long)b * K + k) * 4) =
        make_float4(1.f / (1.f + expf(-u.x)), 1.f / (1.f + expf(-u.y)), 1.f / (1.f + expf(-u.z)), 1.f / (1.f + expf(-u.w)));
  }
  for (int i = tid; i < K * C; i += SEL_THREADS) {
    const int k = i / C, c = i - k * C;
    const int n = (int)(0xffffffffu - (unsigned)(cand[k] & 0xffffffffull));
    out_score[((long)b * K + k) * C + c] = cb[(long)n * C + c];
  }
}

// one thread per (b, n) row
__global__ __launch_bounds__(256) void det_proposals_bwd_kernel(const float* __restrict__ d_score, const float* __restrict__ d_anchor,
                                                                const float* __restrict__ anchor, const int* __restrict__ inv,
                                                                float* __restrict__ d_cls, float* __restrict__ d_raw, long rows,
                                                                int N, int C, int K) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows) return;
  const int b = (int)(i / N);
  const int k = inv[i];
  if (d_cls) {
    float* o = d_cls + i * C;
    if (k >= 0 && d_score) {
      const float* g = d_score + ((long)b * K + k) * C;
      for (int c = 0; c < C; ++c) o[c] = g[c];
    } else {
      for (int c = 0; c < C; ++c) o[c] = 0.f;
    }
  }
  if (d_raw) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k >= 0 && d_anchor) {
      const float4 g = *reinterpret_cast<const float4*>(d_anchor + ((long)b * K + k) * 4);
      const float4 a = *reinterpret_cast<const float4*>(anchor + ((long)b * K + k) * 4);
      v = make_float4(g.x * (1.f - a.x) * a.x, g.y * (1.f - a.y) * a.y, g.z * (1.f - a.z) * a.z, g.w * (1.f - a.w) * a.w);
    }
    *reinterpret_cast<float4*>(d_raw + i * 4) = v;
  }
}

// Targets of the Hungarian matching for S x B (prediction set, image) pairs (detr_head.py:475-543 `_get_target_single`): query
// q of (s, b) gets label gt_lab[b, g] / box gt_boxn[b, g] / weight 1 if ground truth g was assigned to it (qfg[s, b, g] == q),
// else the background class / a zero box / weight 0.  One workgroup per (s, b): the assignment is inverted in LDS, then every
// query row is written once — instead of three fills, three scatters, the index arithmetic and the slices around them.
__global__ __launch_bounds__(256) void det_targets_kernel(const int* __restrict__ qfg, const long long* __restrict__ gt_lab,
                                                          const float* __restrict__ gt_boxn, long long* __restrict__ labels,
                                                          float* __restrict__ bbox_t, float* __restrict__ bbox_w, int B, int Q, int G,
                                                          int bg) {
  extern __shared__ int tg_map[];  // [Q] ground truth of a query or -1
  const int sb = blockIdx.x, b = sb % B;
  for (int q = threadIdx.x; q < Q; q += 256) tg_map[q] = -1;
  __syncthreads();
  for (int g = threadIdx.x; g < G; g += 256) {
    const int q = qfg[(long)sb * G + g];
    if (q >= 0 && q < Q) tg_map[q] = g;  // (an assignment: every query appears at most once)
  }
  __syncthreads();
  for (int q = threadIdx.x; q < Q; q += 256) {
    const int g = tg_map[q];
    const long o = (long)sb * Q + q;
    labels[o] = g >= 0 ? gt_lab[(long)b * G + g] : (long long)bg;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f), one = make_float4(1.f, 1.f, 1.f, 1.f);
    reinterpret_cast<float4*>(bbox_t)[o] = g >= 0 ? reinterpret_cast<const float4*>(gt_boxn)[(long)b * G + g] : z;
    reinterpret_cast<float4*>(bbox_w)[o] = g >= 0 ? one : z;
  }
}

}  // namespace rscotr

using namespace rscotr;

extern "C" int rscotr_det_targets(const int32_t* q_for_gt, const int64_t* gt_lab, const float* gt_boxn, int64_t* labels,
                                  float* bbox_targets, float* bbox_weights, int S, int B, int Q, int G, int num_classes,
                                  void* stream) {
  if (S < 0 || B < 0 || Q <= 0 || G < 0 || Q > 16384) return fail(RSCOTR_E_SHAPE, "rscotr_det_targets: bad shape S=%d B=%d Q=%d G=%d", S, B, Q, G);
  if (S == 0 || B == 0) return RSCOTR_OK;
  if ((G > 0 && (!q_for_gt || !gt_lab || !gt_boxn)) || !labels || !bbox_targets || !bbox_weights)
    return fail(RSCOTR_E_ARG, "rscotr_det_targets: null pointer");
  if ((gt_boxn && !aligned16(gt_boxn)) || !aligned16(bbox_targets) || !aligned16(bbox_weights))
    return fail(RSCOTR_E_ALIGN, "rscotr_det_targets: box tensors must be 16-byte aligned");
  det_targets_kernel<<<dim3((unsigned)(S * B)), 256, (size_t)Q * 4, (hipStream_t)stream>>>(
      q_for_gt, reinterpret_cast<const long long*>(gt_lab), gt_boxn, reinterpret_cast<long long*>(labels), bbox_targets,
      bbox_weights, B, Q, G, num_classes);
  return check_launch("rscotr_det_targets");
}

extern "C" int rscotr_det_proposals(const float* enc_cls, const float* enc_reg, const float* proposals, int proposals_batched,
                                    int64_t* topk_idx, float* topk_score, float* topk_unact, float* topk_anchor, int32_t* inv,
                                    int B, int N, int C, int K, void* stream) {
  if (B < 0 || N <= 0 || C <= 0 || K <= 0) return fail(RSCOTR_E_SHAPE, "rscotr_det_proposals: bad shape B=%d N=%d C=%d K=%d", B, N, C, K);
  if (K > N || K > SEL_MAX_K || N > SEL_MAX_N)
    return fail(RSCOTR_E_SHAPE, "rscotr_det_proposals: needs K <= min(N, %d) and N <= %d (K=%d N=%d)", SEL_MAX_K, SEL_MAX_N, K, N);
  if (B == 0) return RSCOTR_OK;
  if (!enc_cls || !enc_reg || !proposals || !topk_idx || !topk_score || !topk_unact || !topk_anchor || !inv)
    return fail(RSCOTR_E_ARG, "rscotr_det_proposals: null pointer");
  if (!aligned16(enc_reg) || !aligned16(proposals) || !aligned16(topk_unact) || !aligned16(topk_anchor))
    return fail(RSCOTR_E_ALIGN, "rscotr_det_proposals: box tensors must be 16-byte aligned");
  const size_t lds = (size_t)((N + 3) & ~3) * 4 + (size_t)SEL_MAX_K * 8;
  static const bool attr_set = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(det_proposals_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              SEL_MAX_N * 4 + SEL_MAX_K * 8);
    return true;
  }();
  (void)attr_set;
  det_proposals_kernel<<<dim3((unsigned)B), SEL_THREADS, lds, (hipStream_t)stream>>>(
      enc_cls, enc_reg, proposals, proposals_batched ? (long)N * 4 : 0, reinterpret_cast<long long*>(topk_idx), topk_score,
      topk_unact, topk_anchor, inv, N, C, K);
  return check_launch("rscotr_det_proposals");
}

extern "C" int rscotr_det_proposals_bwd(const float* d_score, const float* d_anchor, const float* topk_anchor, const int32_t* inv,
                                        float* d_cls, float* d_reg, int B, int N, int C, int K, void* stream) {
  if (B < 0 || N <= 0 || C <= 0 || K <= 0) return fail(RSCOTR_E_SHAPE, "rscotr_det_proposals_bwd: bad shape");
  if (B == 0) return RSCOTR_OK;
  if (!inv || (d_anchor && !topk_anchor)) return fail(RSCOTR_E_ARG, "rscotr_det_proposals_bwd: null pointer");
  if ((d_anchor && !aligned16(d_anchor)) || (topk_anchor && !aligned16(topk_anchor)) || (d_reg && !aligned16(d_reg)))
    return fail(RSCOTR_E_ALIGN, "rscotr_det_proposals_bwd: box tensors must be 16-byte aligned");
  const long rows = (long)B * N;
  det_proposals_bwd_kernel<<<dim3((unsigned)((rows + 255) / 256)), 256, 0, (hipStream_t)stream>>>(
      d_score, d_anchor, topk_anchor, inv, d_cls, d_reg, rows, N, C, K);
  return check_launch("rscotr_det_proposals_bwd");
}
