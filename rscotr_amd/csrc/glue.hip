// Small fused kernels for the glue between the attention blocks (what the reference leaves to chains of element-wise ATen
// ops and autograd's generic backward nodes): level-embedding adds with their segment-sum gradient, ...
#include "common.h"
#include "rscotr.h"

namespace rscotr {

struct LevelStarts {
  int n;
  int start[9];  // start[n] = N (tokens of all levels)
};

__device__ __forceinline__ int level_of(const LevelStarts& ls, int tok) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < 8; ++i) l += (i < ls.n && tok >= ls.start[i]) ? 1 : 0;
  return l;
}

// out[b, t, :] = (x ? x[b * x_bstride + t * C ...] : 0) + (cst ? cst[b * cst_bstride + t * C ...] : 0) + w[level(t), :]; one thread per float4
__global__ __launch_bounds__(256) void level_embed_fwd_kernel(const float4* __restrict__ x, long x_bstride4,
                                                              const float4* __restrict__ cst, long cst_bstride4,
                                                              const float4* __restrict__ w,
                                                              float4* __restrict__ out, LevelStarts ls, int B, int N, int C4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long per = (long)N * C4;
  if (i >= per * B) return;
  const int b = (int)(i / per);
  const long r = i - (long)b * per;
  const int t = (int)(r / C4), c = (int)(r - (long)t * C4);
  float4 v = w[level_of(ls, t) * C4 + c];
  if (x) { const float4 a = x[(long)b * x_bstride4 + r]; v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
  if (cst) { const float4 a = cst[(long)b * cst_bstride4 + r]; v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
  out[i] = v;
}

// dw[l, c] (+)= sum over b and the tokens t of level l of g[b, t, c], in a FIXED order (bit-reproducible): workgroup
// (chunk k, level l) sums its rows of the level (rows = b * n_l + position) — four row lanes x 64 float4 columns, every lane a
// strided chain with eight loads in flight, the lanes folded in LDS in lane order — leaves a partial, and the LAST workgroup of
// a level to finish folds the partials k = 0.. in order (counter[l] returns to 0: self-resetting).  (Round 4: the first
// version walked 256 rows per workgroup as ONE dependent load-add chain per column: 27 - 60 us per call, 0.4 ms per round.)
__global__ __launch_bounds__(256) void level_embed_bwd_kernel(const float* __restrict__ g, float* __restrict__ part,
                                                              int* __restrict__ counter, float* __restrict__ dw,
                                                              LevelStarts ls, int B, int N, int C, int accumulate) {
  const int l = blockIdx.y, k = blockIdx.x, chunks = gridDim.x;
  const int s0 = ls.start[l], nl = ls.start[l + 1] - s0;
  const long rows = (long)B * nl;
  const long r0 = rows * k / chunks, r1 = rows * (k + 1) / chunks;
  const int C4 = C >> 2, rl = threadIdx.x >> 6, cq = threadIdx.x & 63;
  __shared__ float4 sh[4][256];
  __shared__ int last;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  float4* part4 = reinterpret_cast<float4*>(part);
  for (int c4 = cq; c4 < C4; c4 += 64) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
    for (long r = r0 + rl; r < r1; r += 4) {
      const float4 v = g4[((r / nl) * N + s0 + r % nl) * C4 + c4];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    sh[rl][c4] = a;
  }
  __syncthreads();
  if (rl == 0)
    for (int c4 = cq; c4 < C4; c4 += 64) {
      float4 a = sh[0][c4];
#pragma unroll
      for (int j = 1; j < 4; ++j) { const float4 v = sh[j][c4]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
      part4[((long)l * chunks + k) * C4 + c4] = a;
    }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = (atomicAdd(&counter[l], 1) == chunks - 1);
  __syncthreads();
  if (!last) return;
  __threadfence();
  for (int c = threadIdx.x; c < C; c += 256) {
    float s = 0.f;
#pragma unroll 8
    for (int j = 0; j < chunks; ++j) s += __builtin_nontemporal_load(&part[((long)l * chunks + j) * C + c]);
    dw[(long)l * C + c] = accumulate ? dw[(long)l * C + c] + s : s;
  }
  if (threadIdx.x == 0) counter[l] = 0;
}

// out = [a | b | c | d] (flat concatenation of up to four arrays), one thread per element
__global__ __launch_bounds__(256) void pack4_kernel(const float* __restrict__ a, long na, const float* __restrict__ b, long nb,
                                                    const float* __restrict__ c, long nc, const float* __restrict__ d, long nd,
                                                    float* __restrict__ out, unsigned* __restrict__ amax_out) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long o = i;
  float v = 0.f;
  bool ok = true;
  if (i < na) v = a[i];
  else if ((i -= na) < nb) v = b[i];
  else if ((i -= nb) < nc) v = c[i];
  else if ((i -= nc) < nd) v = d[i];
  else ok = false;
  if (ok) out[o] = v;
  amax_commit(amax_out, fabsf(v));  // (the range word of the packed tensor: common.h)
}


// ---- contrastive denoising queries in slot layout (models/multi/bbox_head/query_denoising.py:104-178) ---------------------
// One wavefront per denoising slot: lane 0 derives the noised label and box of the slot from its ground truth and its ten
// random numbers u = [label_p, new_label, sign x 4, part x 4] (uniform != 0: raw uniforms — new_label = floor(u * classes),
// sign = u >= 0.5; else already the reference's draws, integer-valued), all lanes then gather the label-embedding row.
__global__ __launch_bounds__(256) void cdn_queries_kernel(const int64_t* __restrict__ gt_lab, const float* __restrict__ gt_boxn,
                                                          const int64_t* __restrict__ slot_src, const float* __restrict__ slot_valid,
                                                          const float* __restrict__ slot_neg, const float* __restrict__ u,
                                                          int uniform, const float* __restrict__ embed, float label_thr,
                                                          float box_scale, int num_classes, int64_t* __restrict__ kl_out,
                                                          float* __restrict__ q_label, float* __restrict__ q_bbox, int n, int C) {
  const int slot = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (slot >= n) return;
  const long src = slot_src[slot];
  const bool valid = slot_valid[slot] > 0.f;
  long kl = gt_lab[src];
  const float* r = u + (long)slot * 10;
  if (label_thr > 0.f && r[0] < label_thr) {
    kl = uniform ? min((long)(r[1] * (float)num_classes), (long)num_classes - 1) : (long)r[1];
  }
  if (lane == 0) {
    const float cx = gt_boxn[src * 4], cy = gt_boxn[src * 4 + 1], w = gt_boxn[src * 4 + 2], h = gt_boxn[src * 4 + 3];
    float kb[4] = {cx, cy, w, h};
    if (box_scale > 0.f) {
      const float hw = w / 2.f, hh = h / 2.f;
      float xy[4] = {cx - hw, cy - hh, cx + hw, cy + hh};
      const float df[4] = {hw, hh, hw, hh};
      const float neg = slot_neg[slot];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float sg = uniform ? (r[2 + k] >= 0.5f ? 1.f : 0.f) : r[2 + k];
        const float part = (r[6 + k] + neg) * (sg * 2.0f - 1.0f);
        xy[k] = fminf(fmaxf(xy[k] + part * df[k] * box_scale, 0.f), 1.f);
      }
      kb[0] = (xy[0] + xy[2]) / 2.f; kb[1] = (xy[1] + xy[3]) / 2.f; kb[2] = xy[2] - xy[0]; kb[3] = xy[3] - xy[1];
    }
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) {  // inverse_sigmoid(kb, eps = 1e-3)
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float x = fminf(fmaxf(kb[k], 0.f), 1.f);
        v[k] = logf(fmaxf(x, 1e-3f) / fmaxf(1.f - x, 1e-3f));
      }
      o = make_float4(v[0], v[1], v[2], v[3]);
    }
    reinterpret_cast<float4*>(q_bbox)[slot] = o;
    kl_out[slot] = kl;
  }
  const float4* row = reinterpret_cast<const float4*>(embed + kl * C);
  float4* dst = reinterpret_cast<float4*>(q_label + (long)slot * C);
  for (int c = lane; c < (C >> 2); c += 64) dst[c] = valid ? row[c] : make_float4(0.f, 0.f, 0.f, 0.f);
}

// d(embedding)[r, :] (+)= sum over the valid slots with label r of g[slot, :], slots in order (bit-reproducible)
__global__ __launch_bounds__(256) void cdn_embed_grad_kernel(const float* __restrict__ g, const int64_t* __restrict__ kl,
                                                             const float* __restrict__ slot_valid, float* __restrict__ dw,
                                                             int n, int C, int accumulate) {
  // the slots' labels are staged in LDS (one cooperative load per 4096 slots) so that the ordered scan reads LDS, not
  // a dependent chain of global loads (38 us for 400 slots before)
  __shared__ int lab[4096];
  const int r = blockIdx.x;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};  // columns threadIdx.x + 256 j, C <= 1024
  for (int s0 = 0; s0 < n; s0 += 4096) {
    const int m = min(4096, n - s0);
    __syncthreads();
    for (int i = threadIdx.x; i < m; i += 256) lab[i] = slot_valid[s0 + i] > 0.f ? (int)kl[s0 + i] : -1;
    __syncthreads();
    for (int i = 0; i < m; ++i) {
      if (lab[i] != r) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = threadIdx.x + 256 * j;
        if (c < C) acc[j] += g[(long)(s0 + i) * C + c];
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = threadIdx.x + 256 * j;
    if (c < C) dw[(long)r * C + c] = accumulate ? dw[(long)r * C + c] + acc[j] : acc[j];
  }
}

// ---- classification head pieces (models/multi/cls_head/slvl_cls_head.py:14-23: GlobalAveragePooling + LabelSmoothLoss) --------
// mean over the T tokens of a (B, T, C) map: one workgroup per (b, 16 float4 channel groups), its 16 token slices (tokens
// t = slice mod 16, in order) each sum their share and meet in fixed order — one thread per channel group walking all T tokens
// was a 128-deep dependent chain on 2 workgroups: 37 us for 1.5 MB
constexpr int GAP_SLICES = 16;
__global__ __launch_bounds__(256) void gap_tokens_fwd_kernel(const float4* __restrict__ x, float4* __restrict__ out, int B, int T,
                                                             int C4) {
  __shared__ float4 part[GAP_SLICES][16];
  const int groups = (C4 + 15) / 16;
  const int b = blockIdx.x / groups, c = (blockIdx.x % groups) * 16 + (threadIdx.x & 15), sl = threadIdx.x >> 4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < C4) {
    const float4* p = x + (long)b * T * C4 + c;
    for (int t = sl; t < T; t += GAP_SLICES) {
      const float4 u = p[(long)t * C4];
      a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
    }
  }
  part[sl][threadIdx.x & 15] = a;
  __syncthreads();
  if (sl == 0 && c < C4) {
#pragma unroll
    for (int k = 1; k < GAP_SLICES; ++k) {
      const float4 u = part[k][threadIdx.x & 15];
      a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
    }
    const float inv = 1.f / (float)T;
    out[(long)b * C4 + c] = make_float4(a.x * inv, a.y * inv, a.z * inv, a.w * inv);
  }
}

// dx[b, t, :] = g[b, :] / T (dense, so that the consumer reads it without a copy)
__global__ __launch_bounds__(256) void gap_tokens_bwd_kernel(const float4* __restrict__ g, float4* __restrict__ dx, int B, int T,
                                                             int C4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long per = (long)T * C4;
  if (i >= per * B) return;
  const int b = (int)(i / per), c = (int)(i % C4);
  const float inv = 1.f / (float)T;
  const float4 v = g[(long)b * C4 + c];
  dx[i] = make_float4(v.x * inv, v.y * inv, v.z * inv, v.w * inv);
}

// loss = sum_b sum_c -t[b,c] * log_softmax(score[b])[c] / avg_factor with t = label * (1 - smooth) + smooth / C
// (mmcls LabelSmoothLoss 'original' + soft cross-entropy); dscore[b,c] = (softmax[b,c] * sum_c t[b,c] - t[b,c]) / avg_factor.
// One workgroup (rows are a few, C a few dozen): one wavefront per row, rows folded in order.
__global__ __launch_bounds__(256) void soft_ce_kernel(const float* __restrict__ score, const float* __restrict__ label,
                                                      float* __restrict__ loss, float* __restrict__ dscore, int B, int C,
                                                      float smooth, float inv_avg) {
  __shared__ float rowloss[1024];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int b = w; b < B; b += 4) {
    const float* s = score + (long)b * C;
    const float* l = label + (long)b * C;
    float mx = -3.0e38f;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, s[c]);
    mx = wave_max(mx);
    float se = 0.f, st = 0.f, dot = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float t = l[c] * (1.f - smooth) + smooth / (float)C;
      se += expf(s[c] - mx);
      st += t;
      dot += t * (s[c] - mx);
    }
    se = wave_sum(se); st = wave_sum(st); dot = wave_sum(dot);
    const float lse = logf(se);
    if (lane == 0 && b < 1024) rowloss[b] = (st * lse - dot) * inv_avg;  // -sum t (s - mx - lse)
    for (int c = lane; c < C; c += 64) {
      const float t = l[c] * (1.f - smooth) + smooth / (float)C;
      dscore[(long)b * C + c] = (expf(s[c] - mx) / se * st - t) * inv_avg;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += rowloss[b];
    *loss = a;
  }
}

// out = p0 + p1 + ... + p(n-1), n <= 8, left to right (fixed order); out may be p0.  One thread per float4.
struct SumPtrs { const float4* p[8]; };
__global__ __launch_bounds__(256) void sum8_kernel(SumPtrs P, int n, float4* out, long n4) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 a = P.p[0][i];
#pragma unroll
  for (int k = 1; k < 8; ++k) {
    if (k < n) {
      const float4 v = P.p[k][i];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  out[i] = a;
}

static int level_starts(const char* who, const int* sizes, int L, int N, LevelStarts* ls) {
  if (L < 1 || L > 8) return fail(RSCOTR_E_SHAPE, "%s: 1..8 levels supported, got %d", who, L);
  if (!sizes) return fail(RSCOTR_E_ARG, "%s: null sizes", who);
  ls->n = L;
  int s = 0;
  for (int i = 0; i < L; ++i) {
    if (sizes[i] < 0) return fail(RSCOTR_E_SHAPE, "%s: negative level size", who);
    ls->start[i] = s;
    s += sizes[i];
  }
  ls->start[L] = s;
  for (int i = L + 1; i < 9; ++i) ls->start[i] = s;
  if (s != N) return fail(RSCOTR_E_SHAPE, "%s: level sizes sum to %d, N = %d", who, s, N);
  return RSCOTR_OK;
}

}  // namespace rscotr

using namespace rscotr;

extern "C" int rscotr_level_embed_fwd(const float* x, int64_t x_bstride, const float* cst, int cst_batched, const float* w,
                                      float* out, const int* sizes, int L, int B, int N, int C, void* stream) {
  if (B < 0 || N < 0 || C <= 0 || (C & 3)) return fail(RSCOTR_E_SHAPE, "rscotr_level_embed_fwd: C must be a positive multiple of 4");
  LevelStarts ls;
  if (int e = level_starts("rscotr_level_embed_fwd", sizes, L, N, &ls)) return e;
  const long total = (long)B * N * (C / 4);
  if (total == 0) return RSCOTR_OK;
  if (!w || !out) return fail(RSCOTR_E_ARG, "rscotr_level_embed_fwd: null pointer");
  if (x && (x_bstride & 3)) return fail(RSCOTR_E_ALIGN, "rscotr_level_embed_fwd: x batch stride must be a multiple of 4");
  if (!aligned16(w) || !aligned16(out) || (x && !aligned16(x)) || (cst && !aligned16(cst)))
    return fail(RSCOTR_E_ALIGN, "rscotr_level_embed_fwd: 16-byte aligned tensors required");
  level_embed_fwd_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      reinterpret_cast<const float4*>(x), (long)(x_bstride / 4), reinterpret_cast<const float4*>(cst), cst_batched ? (long)N * (C / 4) : 0,
      reinterpret_cast<const float4*>(w), reinterpret_cast<float4*>(out), ls, B, N, C / 4);
  return check_launch("rscotr_level_embed_fwd");
}

extern "C" int64_t rscotr_level_embed_bwd_workspace(int L, int C) { return (int64_t)L * 32 * C * 4; }

extern "C" int rscotr_level_embed_bwd(const float* g, float* dw, const int* sizes, int L, int B, int N, int C,
                                      int accumulate, float* workspace, int* counters, void* stream) {
  if (B < 0 || N < 0 || C <= 0) return fail(RSCOTR_E_SHAPE, "rscotr_level_embed_bwd: bad shape");
  LevelStarts ls;
  if (int e = level_starts("rscotr_level_embed_bwd", sizes, L, N, &ls)) return e;
  if (!g || !dw || !workspace || !counters) return fail(RSCOTR_E_ARG, "rscotr_level_embed_bwd: null pointer");
  if ((C & 3) || C > 1024 || !aligned16(g) || !aligned16(workspace))
    return fail(RSCOTR_E_SHAPE, "rscotr_level_embed_bwd: C = %d (a multiple of 4, at most 1024) and 16-byte aligned g / workspace required", C);
  level_embed_bwd_kernel<<<dim3(32, (unsigned)L), 256, 0, (hipStream_t)stream>>>(g, workspace, counters, dw, ls, B, N, C, accumulate);
  return check_launch("rscotr_level_embed_bwd");
}

extern "C" int rscotr_pack4(const float* a, int64_t na, const float* b, int64_t nb, const float* c, int64_t nc, const float* d,
                            int64_t nd, float* out, uint32_t* amax_out, void* stream) {
  if (na < 0 || nb < 0 || nc < 0 || nd < 0) return fail(RSCOTR_E_SHAPE, "rscotr_pack4: negative length");
  const long total = na + nb + nc + nd;
  if (total == 0) return RSCOTR_OK;
  if (!out || (na && !a) || (nb && !b) || (nc && !c) || (nd && !d)) return fail(RSCOTR_E_ARG, "rscotr_pack4: null pointer");
  pack4_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(a, na, b, nb, c, nc, d, nd, out, amax_out);
  return check_launch("rscotr_pack4");
}

extern "C" int rscotr_cdn_queries(const int64_t* gt_lab, const float* gt_boxn, const int64_t* slot_src, const float* slot_valid,
                                  const float* slot_neg, const float* u, int uniform, const float* embed, float label_thr,
                                  float box_scale, int num_classes, int64_t* kl_out, float* q_label, float* q_bbox,
                                  int n_slots, int C, void* stream) {
  if (n_slots < 0 || C <= 0 || (C & 3) || num_classes <= 0) return fail(RSCOTR_E_SHAPE, "rscotr_cdn_queries: bad shape");
  if (n_slots == 0) return RSCOTR_OK;
  if (!gt_lab || !gt_boxn || !slot_src || !slot_valid || !slot_neg || !u || !embed || !kl_out || !q_label || !q_bbox)
    return fail(RSCOTR_E_ARG, "rscotr_cdn_queries: null pointer");
  if (!aligned16(embed) || !aligned16(q_label) || !aligned16(q_bbox))
    return fail(RSCOTR_E_ALIGN, "rscotr_cdn_queries: 16-byte aligned embedding / outputs required");
  cdn_queries_kernel<<<(unsigned)((n_slots + 3) / 4), 256, 0, (hipStream_t)stream>>>(
      gt_lab, gt_boxn, slot_src, slot_valid, slot_neg, u, uniform, embed, label_thr, box_scale, num_classes, kl_out, q_label,
      q_bbox, n_slots, C);
  return check_launch("rscotr_cdn_queries");
}

extern "C" int rscotr_cdn_embed_grad(const float* g, const int64_t* kl, const float* slot_valid, float* dw, int rows,
                                     int n_slots, int C, int accumulate, void* stream) {
  if (rows < 0 || n_slots < 0 || C <= 0 || C > 1024) return fail(RSCOTR_E_SHAPE, "rscotr_cdn_embed_grad: bad shape (C <= 1024)");
  if (rows == 0) return RSCOTR_OK;
  if (!g || !kl || !slot_valid || !dw) return fail(RSCOTR_E_ARG, "rscotr_cdn_embed_grad: null pointer");
  cdn_embed_grad_kernel<<<(unsigned)rows, 256, 0, (hipStream_t)stream>>>(g, kl, slot_valid, dw, n_slots, C, accumulate);
  return check_launch("rscotr_cdn_embed_grad");
}

extern "C" int rscotr_gap_tokens_fwd(const float* x, float* out, int B, int T, int C, void* stream) {
  if (B < 0 || T <= 0 || C <= 0 || (C & 3)) return fail(RSCOTR_E_SHAPE, "rscotr_gap_tokens_fwd: bad shape (C a multiple of 4)");
  if (B == 0) return RSCOTR_OK;
  if (!x || !out) return fail(RSCOTR_E_ARG, "rscotr_gap_tokens_fwd: null pointer");
  if (!aligned16(x) || !aligned16(out)) return fail(RSCOTR_E_ALIGN, "rscotr_gap_tokens_fwd: 16-byte aligned tensors required");
  gap_tokens_fwd_kernel<<<(unsigned)(B * ((C / 4 + 15) / 16)), 256, 0, (hipStream_t)stream>>>(
      reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(out), B, T, C / 4);
  return check_launch("rscotr_gap_tokens_fwd");
}

extern "C" int rscotr_gap_tokens_bwd(const float* g, float* dx, int B, int T, int C, void* stream) {
  if (B < 0 || T <= 0 || C <= 0 || (C & 3)) return fail(RSCOTR_E_SHAPE, "rscotr_gap_tokens_bwd: bad shape (C a multiple of 4)");
  if (B == 0) return RSCOTR_OK;
  if (!g || !dx) return fail(RSCOTR_E_ARG, "rscotr_gap_tokens_bwd: null pointer");
  if (!aligned16(g) || !aligned16(dx)) return fail(RSCOTR_E_ALIGN, "rscotr_gap_tokens_bwd: 16-byte aligned tensors required");
  const long total = (long)B * T * (C / 4);
  gap_tokens_bwd_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(
      reinterpret_cast<const float4*>(g), reinterpret_cast<float4*>(dx), B, T, C / 4);
  return check_launch("rscotr_gap_tokens_bwd");
}

extern "C" int rscotr_soft_ce(const float* score, const float* label, float* loss, float* dscore, int B, int C, float smooth,
                              float avg_factor, void* stream) {
  if (B <= 0 || B > 1024 || C <= 0) return fail(RSCOTR_E_SHAPE, "rscotr_soft_ce: 1 <= B <= 1024 rows, C >= 1");
  if (!(avg_factor > 0.f)) return fail(RSCOTR_E_ARG, "rscotr_soft_ce: avg_factor must be positive");
  if (!score || !label || !loss || !dscore) return fail(RSCOTR_E_ARG, "rscotr_soft_ce: null pointer");
  soft_ce_kernel<<<1, 256, 0, (hipStream_t)stream>>>(score, label, loss, dscore, B, C, smooth, 1.f / avg_factor);
  return check_launch("rscotr_soft_ce");
}

extern "C" int rscotr_sum8(const float* p0, const float* p1, const float* p2, const float* p3, const float* p4, const float* p5,
                           const float* p6, const float* p7, int n, float* out, int64_t count, void* stream) {
  if (n < 1 || n > 8 || count < 0 || (count & 3)) return fail(RSCOTR_E_SHAPE, "rscotr_sum8: 1..8 inputs, count a multiple of 4");
  if (count == 0) return RSCOTR_OK;
  const float* ps[8] = {p0, p1, p2, p3, p4, p5, p6, p7};
  SumPtrs P;
  for (int k = 0; k < 8; ++k) {
    if (k < n && (!ps[k] || !aligned16(ps[k]))) return fail(RSCOTR_E_ARG, "rscotr_sum8: null or unaligned input %d", k);
    P.p[k] = reinterpret_cast<const float4*>(k < n ? ps[k] : ps[0]);
  }
  if (!out || !aligned16(out)) return fail(RSCOTR_E_ARG, "rscotr_sum8: null or unaligned output");
  sum8_kernel<<<(unsigned)((count / 4 + 255) / 256), 256, 0, (hipStream_t)stream>>>(P, n, reinterpret_cast<float4*>(out), count / 4);
  return check_launch("rscotr_sum8");
}
