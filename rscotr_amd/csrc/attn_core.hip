// Fused dense-attention core for the det / seg decoders (head dim 32): O = softmax(scale * q k^T + mask) v and its backward in
// ONE pass over the keys each, scores never leaving registers.  Replaces, inside torch.nn.MultiheadAttention as
// wrapped by mmcv (reached from models/multi/bbox_head/transformer.py:103-108 and models/multi/seg_head/mask2former_head.py:
// 183-192), the chain  batched q k^T -> softmax_mask -> batched P v  (three launches and a (B, heads, Lq, Lk) score tensor that
// made a round trip through HBM: 41 MB per DINO decoder layer) and the five launches of its backward.
//
// Layout of the work (all products on v_mfma_f32_32x32x2_f32: fp32 in, fp32 accumulate):
//   * a wavefront owns a 32 x 32 tile of the score matrix at a time.  The reduction index of a 32x32x2 MFMA is spread over the
//     two lane halves; the head dimension is mapped as d = 16 * half + j (step j = 0..15), so the operand of a lane is a
//     CONTIGUOUS half row (64 bytes) of q / k / v / dO — four 16-byte loads, no LDS staging, no transposes.
//   * forward and the dQ pass compute the TRANSPOSED tile S^T = K Q^T: a lane then holds 16 keys of ONE query (its column), so
//     the row maximum / row sum of the online softmax are register reductions plus one cross-half shuffle, and the
//     probabilities sit exactly where the B operand of the next product (O^T = V^T P^T, dQ^T = K^T dS^T) wants them: register
//     j of lane (q, half) is key 4 * half + (j & 3) + 8 * (j >> 2) of the tile, and the A operand of step j is that key's row
//     of V (K) read across the lanes (128-byte segments).
//   * the dK / dV pass owns 32 keys and walks the queries with S = Q K^T (keys across the lanes) for the same reason.
//   * the four wavefronts of a workgroup split the keys (queries) of the walk and meet in LDS in a fixed order; when a launch
//     would have too few workgroups (the seg decoder: 100 queries against 4096 keys) the keys are also cut into chunks whose
//     partial results (running maximum, sum, unnormalised rows) are merged by a second small launch.  No atomics: every sum
//     has an order fixed by the shapes alone.
//   * backward recomputes the probabilities from the saved log-sum-exp of every row (B * heads * Lq floats) instead of
//     reading a stored P.
//   * scores live in log2 units (log2(e) rides in the scale of q: v_exp_f32 is 2^x); a tile whose 32 x 32 mask block is all
//     True is skipped without a product (the DINO decoder's denoising groups against the matching queries, regions outside a
//     predicted mask), a tile without any True bit runs without mask arithmetic; interior tiles address their operands from
//     wave-uniform bases (no clamps).  With few query tiles against many keys (seg cross-attention) the key-side pass gives
//     every wavefront a key block of its own and walks all queries: nothing to combine.
// Mask: bool, True = blocked, indexed by mask_mode as in attn.hip (0 none, 1 (Lq, Lk), 2 (B, Lq, Lk), 3 (B * heads, Lq, Lk)); a
// fully blocked row gives zeros (never produced on this path), as rscotr_softmax_mask_fwd.
#include "common.h"
#include <math.h>
#include <algorithm>

// register prefetch of the next tile's operands, per kernel.  OFF: a set costs 36 / 52 / 70 registers, the passes then fit one
// wavefront per SIMD instead of two, and two resident workgroups hide each other's load latency better than the prefetch does
// (in the step: 35.19 ms per round without any, 35.35 with all three; profiles/r4_attn_core.txt)
#ifndef ATTN_PF_FWD
#define ATTN_PF_FWD 0
#endif
#ifndef ATTN_PF_DQ
#define ATTN_PF_DQ 0
#endif
#ifndef ATTN_PF_DKV
#define ATTN_PF_DKV 0
#endif

namespace rscotr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr float kNegBig = -3.0e38f;
constexpr float kLog2e = 1.4426950408889634f, kLn2 = 0.6931471805599453f;

__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.f;
  return z;
}

// 16 consecutive floats at p (16-byte aligned) -> x[0..15]
__device__ __forceinline__ void load_half_row(const float* __restrict__ p, float (&x)[16]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4 t = reinterpret_cast<const float4*>(p)[i];
    x[4 * i] = t.x; x[4 * i + 1] = t.y; x[4 * i + 2] = t.z; x[4 * i + 3] = t.w;
  }
}

// row (0..31) of the tile that register r of a lane in half `half` holds
__device__ __forceinline__ int tile_row(int r, int half) { return 4 * half + (r & 3) + 8 * (r >> 2); }

__device__ __forceinline__ long mask_block(int mode, int b, int bh) { return mode == 1 ? 0 : (mode == 2 ? b : bh); }

// blocked flags of the 16 (row r of this lane, column `col`) pairs of a TRANSPOSED tile: rows = keys key0 + tile_row(r), the
// lane's query row `mrow` of the mask -> bit r
__device__ __forceinline__ unsigned blocked_keys(const unsigned char* __restrict__ mrow, int key0, int half, int Lk, bool vec) {
  unsigned bits = 0u;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int kb = key0 + 4 * half + 8 * g;
    if (vec && kb + 3 < Lk) {
      if (mrow) {
        const uchar4 m = *reinterpret_cast<const uchar4*>(mrow + kb);
        bits |= ((m.x ? 1u : 0u) | (m.y ? 2u : 0u) | (m.z ? 4u : 0u) | (m.w ? 8u : 0u)) << (4 * g);
      }
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int key = kb + u;
        const bool bl = key >= Lk || (mrow && mrow[key]);
        bits |= (bl ? 1u : 0u) << (4 * g + u);
      }
    }
  }
  return bits;
}

struct AttnGeom {
  int H, Lq, Lk, ldq, ldk, ldv, ldo, mode, nch, tiles_per_chunk;
  float scale;
};

struct AttnBwdGeom {
  int H, Lq, Lk, ldq, ldk, ldv, ldo, lddq, lddk, lddv, mode, nch, tiles_per_chunk;
  float scale;
};

// operands of one key tile as the forward / dQ walks read them: half rows of K (and V) for the score products, columns of V
// (K) for the product that follows, and the lane's 16 mask bytes (loaded WITH the operands: a load issued later would, on the
// in-order return path, wait for the whole prefetch in front of it)
template <bool BWD>
struct KeyTile {
  float krow[16], vrow[BWD ? 16 : 1], col[16];
  uint4 mraw;  // vec: the four uchar4 words of keys key0 + 4 * half + 8 * g + 0..3; ragged: mraw[0] = blocked bits
  __device__ __forceinline__ void load(const float* __restrict__ kb, const float* __restrict__ vb, int ldk, int ldv, int key0,
                                       int Lk, int half, int l32, const unsigned char* __restrict__ mrow, bool vec) {
    if (key0 + 32 <= Lk) {  // (uniform) interior tile: no clamps
      const float* kr = kb + (long)key0 * ldk;
      const float* vr = vb + (long)key0 * ldv;
      load_half_row(kr + l32 * ldk + 16 * half, krow);
      if constexpr (BWD) load_half_row(vr + l32 * ldv + 16 * half, vrow);
      const float* cb = BWD ? kr + 4 * half * ldk + l32 : vr + 4 * half * ldv + l32;
      const int ldc = BWD ? ldk : ldv;
#pragma unroll
      for (int j = 0; j < 16; ++j) col[j] = cb[((j & 3) + 8 * (j >> 2)) * ldc];
    } else {
      const long r = min(key0 + l32, Lk - 1);
      load_half_row(kb + r * ldk + 16 * half, krow);
      if constexpr (BWD) load_half_row(vb + r * ldv + 16 * half, vrow);
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const long rj = min(key0 + tile_row(j, half), Lk - 1);
        col[j] = BWD ? kb[rj * ldk + l32] : vb[rj * ldv + l32];
      }
    }
    if (vec && key0 + 32 <= Lk) {
      const unsigned char* mp = mrow + key0 + 4 * half;
      mraw = mrow ? make_uint4(*reinterpret_cast<const unsigned*>(mp), *reinterpret_cast<const unsigned*>(mp + 8),
                               *reinterpret_cast<const unsigned*>(mp + 16), *reinterpret_cast<const unsigned*>(mp + 24))
                  : make_uint4(0u, 0u, 0u, 0u);
    } else {
      mraw = make_uint4(blocked_keys(mrow, key0, half, Lk, false), 0u, 0u, 0u);
    }
  }
  // bit r: key tile_row(r) of the tile is blocked for the lane's query
  __device__ __forceinline__ unsigned blocked(int key0, int Lk, bool vec) const {
    if (!(vec && key0 + 32 <= Lk)) return mraw.x;
    auto nib = [](unsigned w) { return ((w & 0xffu) ? 1u : 0u) | ((w & 0xff00u) ? 2u : 0u) | ((w & 0xff0000u) ? 4u : 0u) | ((w & 0xff000000u) ? 8u : 0u); };
    return nib(mraw.x) | (nib(mraw.y) << 4) | (nib(mraw.z) << 8) | (nib(mraw.w) << 12);
  }
};

}  // namespace

// ------------------------------------------------------------------------------------------------------------------
// forward: grid (nqb * nch, B * H), 64 * NW threads.  PART: leave (max, sum, unnormalised rows) of the chunk for attn_merge_kernel
template <bool PART, int NW>
__global__ __launch_bounds__(64 * NW) void attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, const unsigned char* __restrict__ mask,
                                                           float* __restrict__ out, float* __restrict__ lse,
                                                           float* __restrict__ part, const AttnGeom G) {
  __shared__ float sO[NW][32][33];
  __shared__ float sM[NW][32], sL[NW][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5, l32 = lane & 31;
  const int bh = blockIdx.y, b = bh / G.H, h = bh % G.H;
  const int qb = blockIdx.x / G.nch, c = blockIdx.x % G.nch;
  const int q0 = qb * 32, qi = min(q0 + l32, G.Lq - 1);
  const int ntiles = (G.Lk + 31) >> 5;
  const int t_beg = c * G.tiles_per_chunk, t_end = min(ntiles, t_beg + G.tiles_per_chunk);
  float qreg[16];
  load_half_row(q + ((long)b * G.Lq + qi) * G.ldq + h * 32 + 16 * half, qreg);
#pragma unroll
  for (int j = 0; j < 16; ++j) qreg[j] *= G.scale * kLog2e;
  const unsigned char* mrow = nullptr;
  if (mask != nullptr && G.mode != 0) mrow = mask + (mask_block(G.mode, b, bh) * G.Lq + qi) * (long)G.Lk;
  const bool vec = (G.Lk & 3) == 0 && ((reinterpret_cast<uintptr_t>(mask) & 3u) == 0);
  const float* kb = k + (long)b * G.Lk * G.ldk + h * 32;
  const float* vb = v + (long)b * G.Lk * G.ldv + h * 32;

  float m_run = kNegBig, l_run = 0.f;
  f32x16 oacc = zero16();
  int t = t_beg + wave;
  KeyTile<false> cur;
  if (t < t_end) cur.load(kb, vb, G.ldk, G.ldv, t * 32, G.Lk, half, l32, mrow, vec);
  for (; t < t_end; t += NW) {
    const int key0 = t * 32;
    KeyTile<false> nxt;  // the next tile of this wavefront is on its way while this one is multiplied (past the end: the last again)
    if (ATTN_PF_FWD) nxt.load(kb, vb, G.ldk, G.ldv, min(t + NW, t_end - 1) * 32, G.Lk, half, l32, mrow, vec);
    const unsigned bl = cur.blocked(key0, G.Lk, vec);
    if (__all(bl == 0xffffu)) {  // nothing of this tile is visible to these 32 queries (DINO's denoising groups, whole
      if (ATTN_PF_FWD) cur = nxt; else cur.load(kb, vb, G.ldk, G.ldv, min(t + NW, t_end - 1) * 32, G.Lk, half, l32, mrow, vec);                 // regions outside a predicted mask): no products at all
      continue;
    }
    f32x16 s = zero16();
#pragma unroll
    for (int j = 0; j < 16; ++j) s = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.krow[j], qreg[j], s, 0, 0, 0);
    if (__any(bl != 0u)) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if ((bl >> r) & 1u) s[r] = -INFINITY;
    }
    float tmax = kNegBig;
#pragma unroll
    for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __builtin_amdgcn_exp2f(s[r] - m_new);  // (scores in log2 units: log2(e) rides in the scale of q)
      psum += s[r];
    }
    if (__any(m_new != m_run)) {
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[r] *= alpha;
    }
    l_run += psum;
    m_run = m_new;
#pragma unroll
    for (int j = 0; j < 16; ++j) oacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.col[j], s[j], oacc, 0, 0, 0);
    if (ATTN_PF_FWD) cur = nxt; else cur.load(kb, vb, G.ldk, G.ldv, min(t + NW, t_end - 1) * 32, G.Lk, half, l32, mrow, vec);
  }
  l_run += __shfl_xor(l_run, 32, 64);
#pragma unroll
  for (int r = 0; r < 16; ++r) sO[wave][l32][tile_row(r, half)] = oacc[r];
  if (half == 0) {
    sM[wave][l32] = m_run;
    sL[wave][l32] = l_run;
  }
  __syncthreads();
  if (tid >= 256) return;
  const int qq = tid >> 3, d4 = (tid & 7) * 4;
  float M = kNegBig;
#pragma unroll
  for (int w = 0; w < NW; ++w) M = fmaxf(M, sM[w][qq]);
  float L = 0.f, o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const float f = __builtin_amdgcn_exp2f(sM[w][qq] - M);
    L += f * sL[w][qq];
#pragma unroll
    for (int u = 0; u < 4; ++u) o[u] += f * sO[w][qq][d4 + u];
  }
  if (PART) {
    const long base = ((long)bh * gridDim.x + blockIdx.x);  // (gridDim.x = nqb * nch: chunk c of query block qb)
    float* po = part + base * 1088;
    *reinterpret_cast<float4*>(po + qq * 32 + d4) = make_float4(o[0], o[1], o[2], o[3]);
    if (d4 == 0) {
      po[1024 + qq] = M;
      po[1056 + qq] = L;
    }
  } else if (q0 + qq < G.Lq) {
    const float inv = L > 0.f ? 1.f / L : 0.f;
    *reinterpret_cast<float4*>(out + ((long)b * G.Lq + q0 + qq) * G.ldo + h * 32 + d4) =
        make_float4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
    if (d4 == 0) lse[(long)bh * G.Lq + q0 + qq] = L > 0.f ? (M + log2f(L)) * kLn2 : INFINITY;
  }
}

// merge of the chunk partials of a query block: grid (nqb, B * H).  SOFTMAX: (max, sum, rows) triples -> out, lse; else plain
// sums of the chunk rows (dQ) in chunk order
template <bool SOFTMAX>
__global__ __launch_bounds__(256) void attn_merge_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                         float* __restrict__ lse, int H, int Lq, int ldo, int nch) {
  const int tid = threadIdx.x, qq = tid >> 3, d4 = (tid & 7) * 4;
  const int bh = blockIdx.y, b = bh / H, h = bh % H, qb = blockIdx.x;
  const float* p0 = part + ((long)bh * gridDim.x + qb) * nch * 1088;
  float M = kNegBig;
  if (SOFTMAX)
    for (int c = 0; c < nch; ++c) M = fmaxf(M, p0[c * 1088 + 1024 + qq]);
  float L = 0.f, o[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c = 0; c < nch; ++c) {
    const float* pc = p0 + c * 1088;
    const float f = SOFTMAX ? __builtin_amdgcn_exp2f(pc[1024 + qq] - M) : 1.f;
    if (SOFTMAX) L += f * pc[1056 + qq];
    const float4 t = *reinterpret_cast<const float4*>(pc + qq * 32 + d4);
    o[0] += f * t.x; o[1] += f * t.y; o[2] += f * t.z; o[3] += f * t.w;
  }
  const int qi = qb * 32 + qq;
  if (qi >= Lq) return;
  const float inv = SOFTMAX ? (L > 0.f ? 1.f / L : 0.f) : 1.f;
  *reinterpret_cast<float4*>(out + ((long)b * Lq + qi) * ldo + h * 32 + d4) =
      make_float4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv);
  if (SOFTMAX && d4 == 0) lse[(long)bh * Lq + qi] = L > 0.f ? (M + log2f(L)) * kLn2 : INFINITY;
}

// ------------------------------------------------------------------------------------------------------------------
// backward, query side: dQ = scale * sum_k dS K with dS = P * (dP - D), D = rowsum(dO * O) (written to `dsum` for the key
// side by the workgroups of chunk 0).  grid (nqb * nch, B * H)
template <bool PART, int NW>
__global__ __launch_bounds__(64 * NW) void attn_bwd_dq_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                              const float* __restrict__ v, const unsigned char* __restrict__ mask,
                                                              const float* __restrict__ out, const float* __restrict__ dout,
                                                              const float* __restrict__ lse, float* __restrict__ dsum,
                                                              float* __restrict__ dq, float* __restrict__ part,
                                                              const AttnBwdGeom G) {
  __shared__ float sQ[NW][32][33];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5, l32 = lane & 31;
  const int bh = blockIdx.y, b = bh / G.H, h = bh % G.H;
  const int qb = blockIdx.x / G.nch, c = blockIdx.x % G.nch;
  const int q0 = qb * 32, qi = min(q0 + l32, G.Lq - 1);
  const int ntiles = (G.Lk + 31) >> 5;
  const int t_beg = c * G.tiles_per_chunk, t_end = min(ntiles, t_beg + G.tiles_per_chunk);
  float qreg[16], doreg[16];
  load_half_row(q + ((long)b * G.Lq + qi) * G.ldq + h * 32 + 16 * half, qreg);
  load_half_row(dout + ((long)b * G.Lq + qi) * G.ldo + h * 32 + 16 * half, doreg);
  float D;
  {
    float oreg[16];
    load_half_row(out + ((long)b * G.Lq + qi) * G.ldo + h * 32 + 16 * half, oreg);
    float a = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) a = fmaf(doreg[j], oreg[j], a);
    D = a + __shfl_xor(a, 32, 64);
  }
  if (c == 0 && wave == 0 && half == 0 && q0 + l32 < G.Lq) dsum[(long)bh * G.Lq + q0 + l32] = D;
#pragma unroll
  for (int j = 0; j < 16; ++j) qreg[j] *= G.scale * kLog2e;
  const float lse_q = lse[(long)bh * G.Lq + qi] * kLog2e;
  const unsigned char* mrow = nullptr;
  if (mask != nullptr && G.mode != 0) mrow = mask + (mask_block(G.mode, b, bh) * G.Lq + qi) * (long)G.Lk;
  const bool vec = (G.Lk & 3) == 0 && ((reinterpret_cast<uintptr_t>(mask) & 3u) == 0);
  const float* kb = k + (long)b * G.Lk * G.ldk + h * 32;
  const float* vb = v + (long)b * G.Lk * G.ldv + h * 32;

  f32x16 acc = zero16();  // dQ^T: rows d, columns = this wavefront's queries
  int t = t_beg + wave;
  KeyTile<true> cur;
  if (t < t_end) cur.load(kb, vb, G.ldk, G.ldv, t * 32, G.Lk, half, l32, mrow, vec);
  for (; t < t_end; t += NW) {
    const int key0 = t * 32;
    KeyTile<true> nxt;
    if (ATTN_PF_DQ) nxt.load(kb, vb, G.ldk, G.ldv, min(t + NW, t_end - 1) * 32, G.Lk, half, l32, mrow, vec);
    const unsigned bl = cur.blocked(key0, G.Lk, vec);
    if (__all(bl == 0xffffu)) {
      if (ATTN_PF_DQ) cur = nxt; else cur.load(kb, vb, G.ldk, G.ldv, min(t + NW, t_end - 1) * 32, G.Lk, half, l32, mrow, vec);
      continue;
    }
    f32x16 s = zero16(), dp = zero16();
#pragma unroll
    for (int j = 0; j < 16; ++j) s = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.krow[j], qreg[j], s, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 16; ++j) dp = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.vrow[j], doreg[j], dp, 0, 0, 0);
    if (__any(bl != 0u)) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if ((bl >> r) & 1u) s[r] = -INFINITY;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = __builtin_amdgcn_exp2f(s[r] - lse_q) * (dp[r] - D) * G.scale;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.col[j], s[j], acc, 0, 0, 0);
    if (ATTN_PF_DQ) cur = nxt; else cur.load(kb, vb, G.ldk, G.ldv, min(t + NW, t_end - 1) * 32, G.Lk, half, l32, mrow, vec);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) sQ[wave][l32][tile_row(r, half)] = acc[r];
  __syncthreads();
  if (tid >= 256) return;
  const int qq = tid >> 3, d4 = (tid & 7) * 4;
  float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < NW; ++w)
#pragma unroll
    for (int u = 0; u < 4; ++u) o[u] += sQ[w][qq][d4 + u];
  if (PART) {
    float* po = part + ((long)bh * gridDim.x + blockIdx.x) * 1088;
    *reinterpret_cast<float4*>(po + qq * 32 + d4) = make_float4(o[0], o[1], o[2], o[3]);
  } else if (q0 + qq < G.Lq) {
    *reinterpret_cast<float4*>(dq + ((long)b * G.Lq + q0 + qq) * G.lddq + h * 32 + d4) = make_float4(o[0], o[1], o[2], o[3]);
  }
}

// operands of one query tile as the key-side walk reads them
struct QueryTile {
  float qrow[16], drow[16], qcol[16], dcol[16], nlse, nD;
  unsigned mb[4];  // mask bytes of (query tile_row(r), this lane's key), four per word
  __device__ __forceinline__ void load(const float* __restrict__ qbp, const float* __restrict__ dob, const float* __restrict__ lseb,
                                       const float* __restrict__ dsb, const unsigned char* __restrict__ mcol, int ldq, int ldo,
                                       int q0, int Lq, int Lk, int half, int l32) {
    if (q0 + 32 <= Lq) {  // (uniform) interior tile
      const float* qr = qbp + (long)q0 * ldq;
      const float* dr = dob + (long)q0 * ldo;
      load_half_row(qr + l32 * ldq + 16 * half, qrow);
      load_half_row(dr + l32 * ldo + 16 * half, drow);
      const float* qc = qr + 4 * half * ldq + l32;
      const float* dc = dr + 4 * half * ldo + l32;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        qcol[j] = qc[((j & 3) + 8 * (j >> 2)) * ldq];
        dcol[j] = dc[((j & 3) + 8 * (j >> 2)) * ldo];
      }
      nlse = lseb[q0 + l32];
      nD = dsb[q0 + l32];
      if (mcol) {
        const unsigned char* mc = mcol + (long)(q0 + 4 * half) * Lk;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const unsigned char* mg = mc + (long)(8 * g) * Lk;
          mb[g] = (unsigned)mg[0] | ((unsigned)mg[Lk] << 8) | ((unsigned)mg[2 * (long)Lk] << 16) | ((unsigned)mg[3 * (long)Lk] << 24);
        }
      } else {
        mb[0] = mb[1] = mb[2] = mb[3] = 0u;
      }
    } else {
      const long qrw = min(q0 + l32, Lq - 1);
      load_half_row(qbp + qrw * ldq + 16 * half, qrow);
      load_half_row(dob + qrw * ldo + 16 * half, drow);
      nlse = lseb[qrw];
      nD = dsb[qrw];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const long qr = min(q0 + tile_row(j, half), Lq - 1);
        qcol[j] = qbp[qr * ldq + l32];
        dcol[j] = dob[qr * ldo + l32];
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        unsigned w = 0u;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int qi = q0 + 4 * half + 8 * g + u;
          const bool bl = qi >= Lq || (mcol && mcol[(long)min(qi, Lq - 1) * Lk]);
          w |= (bl ? 1u : 0u) << (8 * u);
        }
        mb[g] = w;
      }
    }
  }
  __device__ __forceinline__ unsigned blocked() const {
    auto nib = [](unsigned w) { return ((w & 0xffu) ? 1u : 0u) | ((w & 0xff00u) ? 2u : 0u) | ((w & 0xff0000u) ? 4u : 0u) | ((w & 0xff000000u) ? 8u : 0u); };
    return nib(mb[0]) | (nib(mb[1]) << 4) | (nib(mb[2]) << 8) | (nib(mb[3]) << 12);
  }
};

// backward, key side: dV = P^T dO, dK = scale * dS^T Q for 32 keys, walking the queries.  grid (nkb, B * H).  The per-query
// terms of a tile — the row's log-sum-exp and D — enter the score / dP accumulators through ONE extra MFMA step each (A = -lse
// or -D of the lane's query in the first half of the reduction pair, 0 in the second; B = 1): the keys sit across the lanes
// here, so subtracting them element-wise would take a broadcast load per register instead
// OWN (few query tiles against many keys — the seg decoder's cross-attention): every wavefront owns a key block of its own and
// walks ALL query tiles; nothing to combine, no LDS, a quarter of the workgroups
template <int NW, bool OWN>
__global__ __launch_bounds__(64 * NW) void attn_bwd_dkv_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                               const float* __restrict__ v, const unsigned char* __restrict__ mask,
                                                               const float* __restrict__ dout, const float* __restrict__ lse,
                                                               const float* __restrict__ dsum, float* __restrict__ dk,
                                                               float* __restrict__ dv, const AttnBwdGeom G) {
  __shared__ float sK[OWN ? 1 : NW][32][33];
  __shared__ float sV[OWN ? 1 : NW][32][33];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), half = lane >> 5, l32 = lane & 31;
  const int bh = blockIdx.y, b = bh / G.H, h = bh % G.H;
  const int k0 = (OWN ? blockIdx.x * NW + wave : blockIdx.x) * 32, key = k0 + l32, kcl = min(key, G.Lk - 1);
  if (OWN && k0 >= G.Lk) return;
  float kreg[16], vreg[16];
  load_half_row(k + ((long)b * G.Lk + kcl) * G.ldk + h * 32 + 16 * half, kreg);
  load_half_row(v + ((long)b * G.Lk + kcl) * G.ldv + h * 32 + 16 * half, vreg);
#pragma unroll
  for (int j = 0; j < 16; ++j) kreg[j] *= G.scale * kLog2e;
  const unsigned char* mcol = nullptr;  // + qi * Lk per query row
  if (mask != nullptr && G.mode != 0) mcol = mask + mask_block(G.mode, b, bh) * G.Lq * (long)G.Lk + kcl;
  const float* qbp = q + (long)b * G.Lq * G.ldq + h * 32;
  const float* dob = dout + (long)b * G.Lq * G.ldo + h * 32;
  const float* lseb = lse + (long)bh * G.Lq;
  const float* dsb = dsum + (long)bh * G.Lq;
  const int nqt = (G.Lq + 31) >> 5;
  const unsigned kdead = key >= G.Lk ? 0xffffu : 0u;
  f32x16 dvacc = zero16(), dkacc = zero16();  // rows d, columns = this workgroup's keys
  constexpr int TS = OWN ? 1 : NW;  // tile stride of this wavefront's walk
  int t = OWN ? 0 : wave;
  QueryTile cur;
  if (t < nqt) cur.load(qbp, dob, lseb, dsb, mcol, G.ldq, G.ldo, t * 32, G.Lq, G.Lk, half, l32);
  for (; t < nqt; t += TS) {
    QueryTile nxt;
    if (ATTN_PF_DKV) nxt.load(qbp, dob, lseb, dsb, mcol, G.ldq, G.ldo, min(t + TS, nqt - 1) * 32, G.Lq, G.Lk, half, l32);
    const unsigned bl = cur.blocked() | kdead;
    if (__all(bl == 0xffffu)) {
      if (ATTN_PF_DKV) cur = nxt; else cur.load(qbp, dob, lseb, dsb, mcol, G.ldq, G.ldo, min(t + TS, nqt - 1) * 32, G.Lq, G.Lk, half, l32);
      continue;
    }
    f32x16 s = zero16(), dp = zero16();
    s = __builtin_amdgcn_mfma_f32_32x32x2f32(half == 0 ? -cur.nlse * kLog2e : 0.f, 1.f, s, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 16; ++j) s = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.qrow[j], kreg[j], s, 0, 0, 0);
    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(half == 0 ? -cur.nD : 0.f, 1.f, dp, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 16; ++j) dp = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.drow[j], vreg[j], dp, 0, 0, 0);
    if (__any(bl != 0u)) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if ((bl >> r) & 1u) s[r] = -INFINITY;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = __builtin_amdgcn_exp2f(s[r]);
      dp[r] = s[r] * dp[r] * G.scale;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) dvacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.dcol[j], s[j], dvacc, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 16; ++j) dkacc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.qcol[j], dp[j], dkacc, 0, 0, 0);
    if (ATTN_PF_DKV) cur = nxt; else cur.load(qbp, dob, lseb, dsb, mcol, G.ldq, G.ldo, min(t + TS, nqt - 1) * 32, G.Lq, G.Lk, half, l32);
  }
  if (OWN) {  // lane (key, half) holds d = 4 * half + 8 * g + 0..3 of its key's rows
    if (key < G.Lk) {
      float* pv = dv + ((long)b * G.Lk + key) * G.lddv + h * 32 + 4 * half;
      float* pk = dk + ((long)b * G.Lk + key) * G.lddk + h * 32 + 4 * half;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        *reinterpret_cast<float4*>(pv + 8 * g) = make_float4(dvacc[4 * g], dvacc[4 * g + 1], dvacc[4 * g + 2], dvacc[4 * g + 3]);
        *reinterpret_cast<float4*>(pk + 8 * g) = make_float4(dkacc[4 * g], dkacc[4 * g + 1], dkacc[4 * g + 2], dkacc[4 * g + 3]);
      }
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    sV[wave][l32][tile_row(r, half)] = dvacc[r];
    sK[wave][l32][tile_row(r, half)] = dkacc[r];
  }
  __syncthreads();
  if (tid >= 256) return;
  const int kk = tid >> 3, d4 = (tid & 7) * 4;
  if (k0 + kk >= G.Lk) return;
  float a[4] = {0.f, 0.f, 0.f, 0.f}, c[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < NW; ++w)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] += sV[w][kk][d4 + u];
      c[u] += sK[w][kk][d4 + u];
    }
  *reinterpret_cast<float4*>(dv + ((long)b * G.Lk + k0 + kk) * G.lddv + h * 32 + d4) = make_float4(a[0], a[1], a[2], a[3]);
  *reinterpret_cast<float4*>(dk + ((long)b * G.Lk + k0 + kk) * G.lddk + h * 32 + d4) = make_float4(c[0], c[1], c[2], c[3]);
}

namespace {

// key chunks of a launch: enough workgroups for the chip when the query blocks alone are few, at least 8 key tiles (two per
// wavefront) per chunk
void attn_chunks(int BH, int Lq, int Lk, int* nch, int* tiles_per_chunk) {
  const int nqb = (Lq + 31) / 32, ntiles = (Lk + 31) / 32;
  const long wgs = (long)nqb * BH;
  int n = 1;
  if (wgs < 256) n = (int)std::min((long)(512 + wgs - 1) / wgs, (long)std::max(ntiles / 8, 1));
  const int per = (ntiles + n - 1) / n;
  *tiles_per_chunk = per;
  *nch = (ntiles + per - 1) / per;
}

int check_attn(const char* what, const void* const* ptrs, int nptr, int B, int heads, int Lq, int Lk, int hd, const int* lds,
               int nld) {
  if (hd != 32) return fail(RSCOTR_E_SHAPE, "%s: head dim %d (this kernel is built for 32)", what, hd);
  if (B <= 0 || heads <= 0 || Lq <= 0 || Lk <= 0) return fail(RSCOTR_E_SHAPE, "%s: B=%d heads=%d Lq=%d Lk=%d", what, B, heads, Lq, Lk);
  if ((long)B * heads > 65535) return fail(RSCOTR_E_SHAPE, "%s: B * heads = %ld exceeds the grid", what, (long)B * heads);
  for (int i = 0; i < nptr; ++i)
    if (ptrs[i] == nullptr || !aligned16(ptrs[i])) return fail(RSCOTR_E_ALIGN, "%s: operand %d is NULL or not 16-byte aligned", what, i);
  for (int i = 0; i < nld; ++i)
    if (lds[i] < heads * 32 || (lds[i] & 3)) return fail(RSCOTR_E_SHAPE, "%s: row stride %d (>= heads * 32, multiple of 4)", what, lds[i]);
  return RSCOTR_OK;
}

}  // namespace
}  // namespace rscotr

using namespace rscotr;

#ifndef ATTN_NW
#define ATTN_NW 4  // wavefronts per workgroup: they split the walk over the key (query) tiles (8: 35.2 ms per round, 4: 35.0)
#endif
static_assert(ATTN_NW >= 4, "the combine stages use 256 threads");

extern "C" int64_t rscotr_attn_core_workspace(int B, int heads, int Lq, int Lk) {
  int nch, per;
  attn_chunks(B * heads, Lq, Lk, &nch, &per);
  const int64_t nqb = (Lq + 31) / 32;
  int64_t bytes = ((int64_t)B * heads * Lq * 4 + 255) / 256 * 256;  // D = rowsum(dO * O)
  if (nch > 1) bytes += (int64_t)B * heads * nqb * nch * 1088 * 4;
  return bytes;
}

extern "C" int rscotr_attn_core_fwd(const float* q, const float* k, const float* v, const unsigned char* mask, int mask_mode,
                                    float* out, float* lse, int B, int heads, int Lq, int Lk, int hd, int ldq, int ldk, int ldv,
                                    int ldo, float scale, void* workspace, int64_t workspace_bytes, void* stream) {
  const void* ptrs[5] = {q, k, v, out, lse};
  const int lds[4] = {ldq, ldk, ldv, ldo};
  if (int e = check_attn("rscotr_attn_core_fwd", ptrs, 4, B, heads, Lq, Lk, hd, lds, 4)) return e;
  if (lse == nullptr) return fail(RSCOTR_E_ARG, "rscotr_attn_core_fwd: lse is NULL");
  if (mask_mode < 0 || mask_mode > 3 || (mask_mode != 0 && mask == nullptr))
    return fail(RSCOTR_E_ARG, "rscotr_attn_core_fwd: mask_mode %d with mask %p", mask_mode, (const void*)mask);
  AttnGeom G;
  G.H = heads; G.Lq = Lq; G.Lk = Lk; G.ldq = ldq; G.ldk = ldk; G.ldv = ldv; G.ldo = ldo; G.mode = mask ? mask_mode : 0;
  G.scale = scale;
  attn_chunks(B * heads, Lq, Lk, &G.nch, &G.tiles_per_chunk);
  const int nqb = (Lq + 31) / 32;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(nqb * G.nch, B * heads);
  // (launch-site profiler, bench.py: a matrix-pipe kernel like the GEMMs — 2 products of 2 Lq Lk 32 flop per (image, head))
  ProfScope prof(PROF_GEMM, 4.0 * B * heads * (double)Lq * Lk * 32, s, "rscotr::attn_fwd_kernel");
  if (G.nch == 1) {
    attn_fwd_kernel<false, ATTN_NW><<<grid, 64 * ATTN_NW, 0, s>>>(q, k, v, mask, out, lse, nullptr, G);
  } else {
    if (workspace == nullptr || workspace_bytes < rscotr_attn_core_workspace(B, heads, Lq, Lk))
      return fail(RSCOTR_E_ARG, "rscotr_attn_core_fwd: workspace of %lld bytes needed", (long long)rscotr_attn_core_workspace(B, heads, Lq, Lk));
    float* part = reinterpret_cast<float*>(static_cast<char*>(workspace) + ((int64_t)B * heads * Lq * 4 + 255) / 256 * 256);
    attn_fwd_kernel<true, ATTN_NW><<<grid, 64 * ATTN_NW, 0, s>>>(q, k, v, mask, out, lse, part, G);
    attn_merge_kernel<true><<<dim3(nqb, B * heads), 256, 0, s>>>(part, out, lse, heads, Lq, ldo, G.nch);
  }
  return check_launch("rscotr_attn_core_fwd");
}

extern "C" int rscotr_attn_core_bwd(const float* q, const float* k, const float* v, const unsigned char* mask, int mask_mode,
                                    const float* out, const float* dout, const float* lse, float* dq, float* dk, float* dv, int B,
                                    int heads, int Lq, int Lk, int hd, int ldq, int ldk, int ldv, int ldo, int lddq, int lddk,
                                    int lddv, float scale, void* workspace, int64_t workspace_bytes, void* stream) {
  const void* ptrs[8] = {q, k, v, out, dout, dq, dk, dv};
  const int lds[7] = {ldq, ldk, ldv, ldo, lddq, lddk, lddv};
  if (int e = check_attn("rscotr_attn_core_bwd", ptrs, 8, B, heads, Lq, Lk, hd, lds, 7)) return e;
  if (lse == nullptr) return fail(RSCOTR_E_ARG, "rscotr_attn_core_bwd: lse is NULL");
  if (mask_mode < 0 || mask_mode > 3 || (mask_mode != 0 && mask == nullptr))
    return fail(RSCOTR_E_ARG, "rscotr_attn_core_bwd: mask_mode %d with mask %p", mask_mode, (const void*)mask);
  if (workspace == nullptr || workspace_bytes < rscotr_attn_core_workspace(B, heads, Lq, Lk))
    return fail(RSCOTR_E_ARG, "rscotr_attn_core_bwd: workspace of %lld bytes needed", (long long)rscotr_attn_core_workspace(B, heads, Lq, Lk));
  AttnBwdGeom G;
  G.H = heads; G.Lq = Lq; G.Lk = Lk; G.ldq = ldq; G.ldk = ldk; G.ldv = ldv; G.ldo = ldo; G.lddq = lddq; G.lddk = lddk;
  G.lddv = lddv; G.mode = mask ? mask_mode : 0; G.scale = scale;
  attn_chunks(B * heads, Lq, Lk, &G.nch, &G.tiles_per_chunk);
  const int nqb = (Lq + 31) / 32, nkb = (Lk + 31) / 32;
  hipStream_t s = static_cast<hipStream_t>(stream);
  float* dsum = static_cast<float*>(workspace);
  float* part = reinterpret_cast<float*>(static_cast<char*>(workspace) + ((int64_t)B * heads * Lq * 4 + 255) / 256 * 256);
  const dim3 grid(nqb * G.nch, B * heads);
  // (five gradient products + the two recomputed in each pass: 14 Lq Lk 32 flop issued per (image, head), 10 of them algorithmic)
  ProfScope prof(PROF_GEMM, 10.0 * B * heads * (double)Lq * Lk * 32, s, "rscotr::attn_bwd (dq + dkv kernels)");
  if (G.nch == 1) {
    attn_bwd_dq_kernel<false, ATTN_NW><<<grid, 64 * ATTN_NW, 0, s>>>(q, k, v, mask, out, dout, lse, dsum, dq, nullptr, G);
  } else {
    attn_bwd_dq_kernel<true, ATTN_NW><<<grid, 64 * ATTN_NW, 0, s>>>(q, k, v, mask, out, dout, lse, dsum, dq, part, G);
    attn_merge_kernel<false><<<dim3(nqb, B * heads), 256, 0, s>>>(part, dq, nullptr, heads, Lq, lddq, G.nch);
  }
  if (nqb <= 8 && (long)((nkb + ATTN_NW - 1) / ATTN_NW) * B * heads >= 128)  // few queries, many keys: a key block per wavefront
    attn_bwd_dkv_kernel<ATTN_NW, true><<<dim3((nkb + ATTN_NW - 1) / ATTN_NW, B * heads), 64 * ATTN_NW, 0, s>>>(q, k, v, mask, dout, lse, dsum, dk, dv, G);
  else
    attn_bwd_dkv_kernel<ATTN_NW, false><<<dim3(nkb, B * heads), 64 * ATTN_NW, 0, s>>>(q, k, v, mask, dout, lse, dsum, dk, dv, G);
  return check_launch("rscotr_attn_core_bwd");
}
