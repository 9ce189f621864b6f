// Shared pieces of the GEMM family (csrc/gemm.hip, csrc/gemm_pp.hip): the parameter block of a product, the fused
// epilogues, the XCD-aware tile order and the three-plane bf16 split.  See gemm.hip for the conventions.
#pragma once
#include "common.h"

namespace rscotr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_RELU_GRAD = 3, ACT_GELU_GRAD = 4,
       // the ReLU pair of the wide FFN products with the gate as ONE BIT per element (interior 128 x 128 split-product tiles only;
       // rscotr_gemm_relu_bits_ok): ACT_RELU_BITS = ACT_RELU that also leaves [y > 0] through `pre`, ACT_RELU_GRAD_BITS =
       // ACT_RELU_GRAD that reads those words through `aux` instead of the M x N activation.  Word layout: uint64
       // [tile (row-major over the 128 x 128 tiles)][thread of the workgroup], bit (i * 2 + j) * 16 + r = accumulator element r of the
       // thread's 32 x 32 tile (i, j) — opaque to callers, the same in both kernels.
       ACT_RELU_BITS = 5, ACT_RELU_GRAD_BITS = 6 };

constexpr int GEMM_BK = 16;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;

struct GemmParams {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* aux;
  float* pre;
  const float* resid;
  float* C2;          // optional second output (same ldc): C2 = C + resid while C itself is stored WITHOUT the residual
                      // (a gradient that is wanted both alone and merged with another one: no element-wise add launch)
  int M, N, K;
  int lda, ldb, ldc;
  int act, accumulate;
  int vecA, vecB;     // 16-byte vector loads legal for this operand
  int vecC;           // C / aux / pre / resid / bias: 16-byte accesses legal at (m, n % 4 == 0) (split-K combine)
  int ksplit_len;     // k elements per split (multiple of GEMM_BK)
  int splits;         // > 1: split-K through slabs, combined by gemm_splitk_reduce_kernel
  int tiles;          // output tiles (tiles_m * tiles_n)
  float* slabs;       // [splits][M][N] when splits > 1
  float* rs_slabs;    // [splits][M] row-sum partials when splits > 1 and rowsum
  float* rowsum;      // a_kmajor only: rowsum[m] (+)= sum_k Aop[m,k]  (bias gradient riding the dW contraction)
  int rowsum_acc;
  // batched mode (nb1 > 0): blockIdx.y = (b0 * nb1 + b1) * nb2 + b2 selects the problem; element offsets per
  // level (level 2 is used to cut a long reduction into slices that write separate slabs)
  int nb1, nb2;
  long sA0, sA1, sA2, sB0, sB1, sB2, sC0, sC1, sC2;
  // per-sample scaling (stochastic depth folded into the Linear around it): the epilogue multiplies row m by
  // rowscale[m / rows_per]; a k-major A operand is multiplied by kscale[k / krows_per] while it is staged
  const float* rowscale;
  const float* kscale;
  int rows_per, krows_per;
  // value ranges (round 5, the fp16 split product below): amax_a / amax_b -> the bit pattern of max |x| over (a superset
  // of) the operand, written by whoever produced the tensor (rscotr_amax_f32, an epilogue, the optimizer); both non-null
  // = the product may run on fp16 planes scaled by powers of two taken from them.  amax_out: the epilogue folds max |C|
  // into this slot (atomicMax on the bit pattern: order-independent, hence deterministic).
  const unsigned* amax_a = nullptr;
  const unsigned* amax_b = nullptr;
  unsigned* amax_out = nullptr;
};

// Power-of-two scale of an operand of the fp16 split product from the bit pattern of its amax: amax * 2^s lands in
// [2^12, 2^13) — 8 x below fp16's largest finite value (per-sample k factors of DropPath / Mixup ride on top), 26 binades
// above the point where the planes start to lose relative precision (then the error is 2^-36 absolute in the scaled
// domain = 2^-48 of amax).  Returns the exponent field of 2^s, clamped so that 2^(s+11) and 2^-s are finite normals; a zero
// tensor (amax 0) takes the clamp, which multiplies zeros.
__device__ __forceinline__ int h3_scale_exp(unsigned amax_bits) {
  const int e = (int)((amax_bits >> 23) & 0xffu);
  return max(13, min(266 - e, 243));
}

__device__ __forceinline__ float gelu_f(float x) {
  return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// amx (all epilogue forms): running max |stored value| of this thread (C and, when present, C2) — folded into
// p.amax_out by amax_commit at the end of the kernel.
__device__ __forceinline__ float epilogue_one(const GemmParams& p, float v, int m, int n, float& amx) {
  if (p.bias) v += p.bias[n];
  const long o = (long)m * p.ldc + n;
  if (p.pre) p.pre[o] = v;
  switch (p.act) {
    case ACT_RELU: v = fmaxf(v, 0.f); break;
    case ACT_GELU: v = gelu_f(v); break;
    case ACT_RELU_GRAD: v = p.aux[o] > 0.f ? v : 0.f; break;
    case ACT_GELU_GRAD: v *= gelu_grad_f(p.aux[o]); break;
    default: break;
  }
  if (p.rowscale) v *= p.rowscale[m / p.rows_per];
  if (p.C2) {
    if (p.accumulate) v += p.C[o];
    const float v2 = p.resid ? v + p.resid[o] : v;
    p.C2[o] = v2;
    amx = fmaxf(amx, fmaxf(fabsf(v), fabsf(v2)));
    return v;
  }
  if (p.resid) v += p.resid[o];
  if (p.accumulate) v += p.C[o];
  amx = fmaxf(amx, fabsf(v));
  return v;
}


// End of a kernel: the wavefront's max goes to p.amax_out with one atomicMax on the bit pattern (non-negative floats order
// like unsigned integers; a maximum does not depend on the order of its operands: deterministic).  Every lane of the
// wavefront must reach this call.

// Four consecutive columns of one row (n % 4 == 0, p.vecC): every load is issued before the first store — the
// element-wise form chains load -> store four times, and each wait also drains the store queued before it.
__device__ __forceinline__ void epilogue_store4(const GemmParams& p, float4 v, int m, int n, float& amx) {
  const long o = (long)m * p.ldc + n;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), r = a, c = a;
  const bool need_aux = p.act == ACT_RELU_GRAD || p.act == ACT_GELU_GRAD;
  if (need_aux) a = *reinterpret_cast<const float4*>(p.aux + o);
  if (p.resid) r = *reinterpret_cast<const float4*>(p.resid + o);
  if (p.accumulate) c = *reinterpret_cast<const float4*>(p.C + o);
  if (p.bias) {
    const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
  }
  if (p.pre) *reinterpret_cast<float4*>(p.pre + o) = v;
  switch (p.act) {
    case ACT_RELU: v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); break;
    case ACT_GELU: v.x = gelu_f(v.x); v.y = gelu_f(v.y); v.z = gelu_f(v.z); v.w = gelu_f(v.w); break;
    case ACT_RELU_GRAD:
      v.x = a.x > 0.f ? v.x : 0.f; v.y = a.y > 0.f ? v.y : 0.f; v.z = a.z > 0.f ? v.z : 0.f; v.w = a.w > 0.f ? v.w : 0.f;
      break;
    case ACT_GELU_GRAD:
      v.x *= gelu_grad_f(a.x); v.y *= gelu_grad_f(a.y); v.z *= gelu_grad_f(a.z); v.w *= gelu_grad_f(a.w);
      break;
    default: break;
  }
  if (p.rowscale) {
    const float f = p.rowscale[m / p.rows_per];
    v.x *= f; v.y *= f; v.z *= f; v.w *= f;
  }
  if (p.C2) {
    if (p.accumulate) { v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w; }
    *reinterpret_cast<float4*>(p.C + o) = v;
    const float4 v2 = make_float4(v.x + r.x, v.y + r.y, v.z + r.z, v.w + r.w);
    *reinterpret_cast<float4*>(p.C2 + o) = v2;
    amx = amax4(amax4(amx, v), v2);
    return;
  }
  if (p.resid) { v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
  if (p.accumulate) { v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w; }
  *reinterpret_cast<float4*>(p.C + o) = v;
  amx = amax4(amx, v);
}

// Staged epilogue of four consecutive rows m..m+3 of one column n (v already holds accumulator + bias): the loads of
// the four rows are issued together, phase by phase (aux -> row scale -> residual -> old C), ahead of the C stores.
// epilogue_one in a loop costs one dependent load latency per element because the stores in between may alias (~40 %
// of a K = 256 tile); four rows at a time keep the 64x64 kernels at 55-76 VGPRs (8 rows: 75-96, 16 rows: 110-170;
// measured on the step: 41.3 / 42.0 / 43.0 ms of GEMM per round against 44.3 element-wise).
template <bool EDGE>
__device__ __forceinline__ void epilogue_rows4(const GemmParams& p, float (&v)[4], int m, int n, float& amx) {
  float x[4];
  int o[4];  // element offsets from row m
  bool ok[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    ok[u] = !EDGE || m + u < p.M;
    o[u] = u * p.ldc;
  }
  const long base = (long)m * p.ldc + n;
  float* crow = p.C + base;
  if (p.pre) {  // (stores do not hold back the loads issued after them; only load -> use -> store chains hurt)
    float* pp = p.pre + base;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (ok[u]) pp[o[u]] = v[u];
  }
  if (p.act == ACT_RELU_GRAD || p.act == ACT_GELU_GRAD) {
    const float* auxp = p.aux + base;
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = ok[u] ? auxp[o[u]] : 0.f;
  }
  switch (p.act) {
    case ACT_RELU:
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = fmaxf(v[u], 0.f);
      break;
    case ACT_GELU:
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = gelu_f(v[u]);
      break;
    case ACT_RELU_GRAD:
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = x[u] > 0.f ? v[u] : 0.f;
      break;
    case ACT_GELU_GRAD:
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] *= gelu_grad_f(x[u]);
      break;
    default: break;
  }
  if (p.rowscale) {
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = ok[u] ? p.rowscale[(m + u) / p.rows_per] : 1.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] *= x[u];
  }
  if (p.C2) {  // C = value (+ old C), C2 = C + resid
    float r[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.resid) {
      const float* rp = p.resid + base;
#pragma unroll
      for (int u = 0; u < 4; ++u) r[u] = ok[u] ? rp[o[u]] : 0.f;
    }
    if (p.accumulate) {
#pragma unroll
      for (int u = 0; u < 4; ++u) x[u] = ok[u] ? crow[o[u]] : 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] += x[u];
    }
    float* c2 = p.C2 + base;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (ok[u]) {
        crow[o[u]] = v[u]; c2[o[u]] = v[u] + r[u];
        amx = fmaxf(amx, fmaxf(fabsf(v[u]), fabsf(v[u] + r[u])));
      }
    return;
  }
  if (p.resid) {
    const float* rp = p.resid + base;
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = ok[u] ? rp[o[u]] : 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] += x[u];
  }
  if (p.accumulate) {
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = ok[u] ? crow[o[u]] : 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] += x[u];
  }
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (ok[u]) { crow[o[u]] = v[u]; amx = fmaxf(amx, fabsf(v[u])); }
}

// One 32 x 32 accumulator tile of a lane (16 values: rows m0 + (r & 3) + 8 (r >> 2), column n) whose epilogue reads exactly ONE
// extra tensor (aux of an act' epilogue, or the residual, or the old C of an accumulate; no second output): all 16 loads are
// issued before the first use.  epilogue_rows4 walks the tile in four groups of rows, each paying one memory latency with
// cold operands — measured with cold operands, 10880 x 256 x 256 + residual: 23.0 us against 17.1 without the residual; the
// single-tensor case is what the step's large products carry (identity of an attention / FFN block, ReLU' of a dX, a merged
// gradient), and one 16-register buffer does not cost the 64 x 64 kernels a resident workgroup.
template <bool EDGE>
__device__ __forceinline__ void epilogue_tile16(const GemmParams& p, const f32x16& acc, float bv, int m0, int n, float& amx) {
  const bool need_aux = p.act == ACT_RELU_GRAD || p.act == ACT_GELU_GRAD;
  const float* ep = need_aux ? p.aux : (p.resid ? p.resid : p.C);
  const long base = (long)m0 * p.ldc + n;
  float e[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int dm = (r & 3) + 8 * (r >> 2);
    e[r] = (!EDGE || m0 + dm < p.M) ? ep[base + (long)dm * p.ldc] : 0.f;
  }
#ifndef RSCOTR_NO_UNSWITCH  // (A/B builds: scripts/build_variant.sh)
  // the two forms the step's large products carry, without the per-element tests of the general loop below (the compiler keeps
  // them inside the unrolled body: ~15 scalar instructions per element): a residual / old C on a plain product, and ReLU'
  if (!p.pre && !p.rowscale && (p.act == ACT_NONE || p.act == ACT_RELU_GRAD)) {
    const bool gate = p.act == ACT_RELU_GRAD;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int dm = (r & 3) + 8 * (r >> 2);
      if (EDGE && m0 + dm >= p.M) continue;
      float v = acc[r] + bv;
      v = gate ? (e[r] > 0.f ? v : 0.f) : v + e[r];
      p.C[base + (long)dm * p.ldc] = v;
      amx = fmaxf(amx, fabsf(v));
    }
    return;
  }
#endif
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int dm = (r & 3) + 8 * (r >> 2);
    if (EDGE && m0 + dm >= p.M) continue;
    const long o = base + (long)dm * p.ldc;
    float v = acc[r] + bv;
    if (p.pre) p.pre[o] = v;
    switch (p.act) {
      case ACT_RELU: v = fmaxf(v, 0.f); break;
      case ACT_GELU: v = gelu_f(v); break;
      case ACT_RELU_GRAD: v = e[r] > 0.f ? v : 0.f; break;
      case ACT_GELU_GRAD: v *= gelu_grad_f(e[r]); break;
      default: break;
    }
    if (p.rowscale) v *= p.rowscale[(m0 + dm) / p.rows_per];
    if (!need_aux) v += e[r];  // the residual, or the old C of an accumulate
    p.C[o] = v;
    amx = fmaxf(amx, fabsf(v));
  }
}

// One 32 x 32 accumulator tile of a lane whose epilogue reads NO tensor (bias + ReLU / GELU, optionally the pre-activation
// stored as well: the first Linear of an FFN / MLP): nothing to batch, so none of epilogue_rows4's staging and scheduling
// barriers — the encoder's 10880 x 2048 x 256 FFN product ran 82-92 us through epilogue_rows4 and 70-74 us through a loop like
// this one (profiles/r5_relu_bits_ab.txt).
template <bool EDGE>
__device__ __forceinline__ void epilogue_noload16(const GemmParams& p, const f32x16& acc, float bv, int m0, int n, float& amx) {
  const long base = (long)m0 * p.ldc + n;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int dm = (r & 3) + 8 * (r >> 2);
    if (EDGE && m0 + dm >= p.M) continue;
    const long o = base + (long)dm * p.ldc;
    float v = acc[r] + bv;
    if (p.pre) p.pre[o] = v;
    if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
    else if (p.act == ACT_GELU) v = gelu_f(v);
    p.C[o] = v;
    amx = fmaxf(amx, fabsf(v));
  }
}

// XCD-aware tile order: the dispatcher places workgroup b on XCD b % 8 (observed, used for speed
// only); give each XCD a contiguous run of tiles in row-major (tile_m, tile_n) order so that the
// A row-panel a run shares is fetched into ONE private L2 (bijective for any tile count).
__device__ __forceinline__ int xcd_swizzle(int id, int n) {
  const int q = n >> 3, r = n & 7, x = id & 7, j = id >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
}

template <int NPL>
__device__ __forceinline__ void split_planes(float x, __bf16 (&p)[3]) {
  p[0] = (__bf16)x;
  const float r1 = x - (float)p[0];
  p[1] = (__bf16)r1;
  if (NPL == 3) p[2] = (__bf16)(r1 - (float)p[1]);
}

__device__ __forceinline__ unsigned pack_bf16(__bf16 a, __bf16 b) {
  return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}

// ---- the fp16 split ("h3") of one value pair, shared by csrc/gemm.hip and csrc/ffn.hip (the comment block in front of
// split_rows4_h in gemm.hip describes the arithmetic)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
struct H3Scale {
  float sc, sc2;  // 2^s, 2^(s + 11)
};
__device__ __forceinline__ void split_pair_h(float a, float b, const H3Scale& k, unsigned (&out)[3]) {
  const f32x2_t x = {a, b};
  const f32x2_t y = x * k.sc, y2 = x * k.sc2;
  const f16x2_t h = __builtin_convertvector(y, f16x2_t);
  out[0] = __builtin_bit_cast(unsigned, h);
  const f32x2_t r = {__builtin_fmaf((float)h.x, -2048.f, y2.x), __builtin_fmaf((float)h.y, -2048.f, y2.y)};
  out[1] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2_t));
}

// combine kernel for the split-K slabs of p (p.splits > 1): gemm.hip
void splitk_reduce_launch(const GemmParams& p, const float* workspace, hipStream_t s);

}  // namespace rscotr
