// Swin (shifted-)window attention core for gfx950: everything between the qkv Linear and the
// proj Linear of mmdet's ShiftWindowMSA / WindowMSA (the Swin-T backbone the reference builds from
// configs/multi/MTL_slvlcls_swin-t-p4-w7_1x1_resisc&dior&potsdam.py:9-25 and runs at
// models/multi/multitask_learner.py:83) in ONE kernel per direction:
//   zero-pad to a multiple of 7 -> cyclic shift -> 7x7 window partition -> per (window, head)
//   softmax(q k^T / sqrt(32) + relative-position bias [+ -100 shift mask]) v -> window reverse ->
//   un-shift -> crop.
// The reference does this with F.pad, torch.roll, three reshape/permute copies, two batched
// matmuls, a gather of the bias table, two adds, a softmax and the inverse copies (~20 launches
// forward, ~40 backward per block).  Here pad/shift/partition are index arithmetic on the token
// grid, so the kernel reads the (B, H*W, 3C) output of the qkv GEMM in place and writes
// (B, H*W, C) for the proj GEMM.
//
// Upstream semantics kept: the qkv Linear acts on the zero padding too, so a pad
// token has q = k = v = qkv bias and takes part in the softmax; pad tokens are only masked from
// other regions by the shift mask; relative-position index (dy+6)*13 + (dx+6); shift is applied
// even when the map is not larger than one window.
//
// CDNA4 mapping: one wavefront per (batch, window, head), persistent over windows of a fixed head;
// lane i < 49 owns query row i (scores, softmax and the row's output stay in its registers — no
// cross-lane traffic); K/V (and Q/dO in backward) tiles of the window live in LDS as 49 x 32 fp32
// and are read as wave-wide broadcasts (ds_read_b128, conflict-free).  f32 MFMA runs at the VALU
// rate on gfx950, so the 49x49x32 products stay on the VALU.  Backward recomputes the
// probabilities from q, k (nothing but qkv is saved), exchanges P / dS through one 49x49 LDS
// matrix for the column-wise products (dV, dK), and accumulates the bias-table and pad-token
// (qkv-bias) gradients in LDS, leaving one atomic per entry per workgroup.
#include "common.h"
#include "rscotr.h"
#include <stdlib.h>

namespace rscotr {

constexpr int WS = 7, WN = 49, HD = 32, TBL = 169;
// per-workgroup partial row of the backward kernels: [0, TBL) bias-table gradient of the workgroup's head,
// [WATTN_PBIAS, WATTN_PBIAS + 3 HD) pad-token gradient of the head's q | k | v bias slice
constexpr int WATTN_PBIAS = 172, WATTN_PROW = WATTN_PBIAS + 3 * HD;

struct WinGeom {
  int H, W, Hp, Wp, nWw, nW, C, heads, shift;
};

struct TokPos {
  bool pad;
  long tok;   // token index inside the image (y*W + x) when !pad
  int label;  // shift-mask region label
};

__device__ __forceinline__ TokPos win_token(const WinGeom& g, int wy, int wx, int t) {
  TokPos r;
  const int ys = wy * WS + t / WS, xs = wx * WS + t % WS;  // position on the shifted, padded canvas
  int yo = ys + g.shift, xo = xs + g.shift;                // roll(-shift): shifted[y] = padded[(y+shift) % Hp]
  if (yo >= g.Hp) yo -= g.Hp;
  if (xo >= g.Wp) xo -= g.Wp;
  r.pad = (yo >= g.H) || (xo >= g.W);
  r.tok = (long)yo * g.W + xo;
  const int ry = ys < g.Hp - WS ? 0 : (ys < g.Hp - g.shift ? 1 : 2);
  const int rx = xs < g.Wp - WS ? 0 : (xs < g.Wp - g.shift ? 1 : 2);
  r.label = ry * 3 + rx;
  return r;
}

// One 32-float LDS row held in registers.  The loops below prefetch row j+1 (8 x ds_read_b128,
// wave-wide broadcast) before the FMAs of row j and pin that order with sched_barrier: with one
// wavefront per SIMD nothing else hides the LDS latency, and left alone the compiler waits
// lgkmcnt(0) after every single read.
struct Row {
  float4 v[HD / 4];
};

__device__ __forceinline__ void ld_row(Row& r, const float4* p) {
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) r.v[c] = p[c];
}

__device__ __forceinline__ float dot_row(const float* q, const Row& r) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;  // four independent chains
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) {
    s0 += q[4 * c] * r.v[c].x;
    s1 += q[4 * c + 1] * r.v[c].y;
    s2 += q[4 * c + 2] * r.v[c].z;
    s3 += q[4 * c + 3] * r.v[c].w;
  }
  return (s0 + s1) + (s2 + s3);
}

__device__ __forceinline__ void axpy_row(float* acc, float a, const Row& r) {
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) {
    acc[4 * c] += a * r.v[c].x; acc[4 * c + 1] += a * r.v[c].y;
    acc[4 * c + 2] += a * r.v[c].z; acc[4 * c + 3] += a * r.v[c].w;
  }
}

#define WATTN_PIN() __builtin_amdgcn_sched_barrier(0)

// for j in [0, 49): body(j, row j of `base`, pre(j)) with row j+1 and pre(j+1) already in flight.
// Hand-unrolled by two (ping-pong register sets) and otherwise kept ROLLED: fully unrolled, the
// kernel is ~20k instructions and spills.
template <typename Pre, typename Body>
__device__ __forceinline__ void for_rows(const float4* base, Pre&& pre, Body&& body) {
  Row A, B;
  ld_row(A, base);
  float a = pre(0), b;
#pragma clang loop unroll(disable)
  for (int j = 0; j < WN - 1; j += 2) {
    ld_row(B, base + (j + 1) * (HD / 4));
    b = pre(j + 1);
    WATTN_PIN();
    body(j, A, a);
    WATTN_PIN();
    ld_row(A, base + (j + 2) * (HD / 4));
    a = pre(j + 2);
    WATTN_PIN();
    body(j + 1, B, b);
    WATTN_PIN();
  }
  body(WN - 1, A, a);
}

// same with two row sources
template <typename Pre, typename Body>
__device__ __forceinline__ void for_rows2(const float4* base0, const float4* base1, Pre&& pre, Body&& body) {
  Row A0, A1, B0, B1;
  ld_row(A0, base0);
  ld_row(A1, base1);
  float a = pre(0), b;
#pragma clang loop unroll(disable)
  for (int j = 0; j < WN - 1; j += 2) {
    ld_row(B0, base0 + (j + 1) * (HD / 4));
    ld_row(B1, base1 + (j + 1) * (HD / 4));
    b = pre(j + 1);
    WATTN_PIN();
    body(j, A0, A1, a);
    WATTN_PIN();
    ld_row(A0, base0 + (j + 2) * (HD / 4));
    ld_row(A1, base1 + (j + 2) * (HD / 4));
    a = pre(j + 2);
    WATTN_PIN();
    body(j + 1, B0, B1, b);
    WATTN_PIN();
  }
  body(WN - 1, A0, A1, a);
}

// Stage one 49 x 32 operand of the window into LDS ([t][32] floats); `which` = 0 q, 1 k, 2 v selects
// the slice of the (B, L, 3C) qkv tensor; pad tokens read the qkv bias.  The 7 loads of a lane are
// issued back to back (one wavefront per SIMD has nothing else to hide their latency behind).
__device__ __forceinline__ void stage_qkv(float4* dst, const float* __restrict__ qkv,
                                          const float* __restrict__ qkv_b, const WinGeom& g, int b, int wy,
                                          int wx, int head, int which, int lane) {
  const long L = (long)g.H * g.W;
  constexpr int NI = (WN * (HD / 4) + 63) / 64;  // 7
  float4 v[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int idx = lane + i * 64;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < WN * (HD / 4)) {
      const int t = idx >> 3, c4 = idx & 7;
      const TokPos p = win_token(g, wy, wx, t);
      const int ch = which * g.C + head * HD + c4 * 4;
      if (!p.pad)
        v[i] = *reinterpret_cast<const float4*>(qkv + ((long)b * L + p.tok) * 3 * g.C + ch);
      else if (qkv_b)
        v[i] = *reinterpret_cast<const float4*>(qkv_b + ch);
    }
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int idx = lane + i * 64;
    if (idx < WN * (HD / 4)) dst[idx] = v[i];
  }
}

// dO of the window (zero for pad tokens)
__device__ __forceinline__ void stage_dout(float4* dst, const float* __restrict__ dout, const WinGeom& g, int b,
                                           int wy, int wx, int head, int lane) {
  const long L = (long)g.H * g.W;
  constexpr int NI = (WN * (HD / 4) + 63) / 64;
  float4 v[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int idx = lane + i * 64;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < WN * (HD / 4)) {
      const int t = idx >> 3, c4 = idx & 7;
      const TokPos p = win_token(g, wy, wx, t);
      if (!p.pad) v[i] = *reinterpret_cast<const float4*>(dout + ((long)b * L + p.tok) * g.C + head * HD + c4 * 4);
    }
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int idx = lane + i * 64;
    if (idx < WN * (HD / 4)) dst[idx] = v[i];
  }
}

__device__ __forceinline__ void load_row(float* r, const float4* src, float scale) {
#pragma unroll
  for (int c = 0; c < HD / 4; ++c) {
    const float4 v = src[c];
    r[4 * c] = v.x * scale; r[4 * c + 1] = v.y * scale; r[4 * c + 2] = v.z * scale; r[4 * c + 3] = v.w * scale;
  }
}

// Sum acc[0..32) over the pad-token lanes of the wavefront into sdB[0..32) (the gradient those
// tokens send to the qkv bias).  Butterfly reduction + one plain LDS update per channel: same-address
// LDS atomics from up to 45 lanes serialise for thousands of cycles per phase.
__device__ __forceinline__ void pad_bias_grad(float* sdB, const float* acc, bool pad_lane, float scale, int lane) {
  if (!__any(pad_lane)) return;  // wave-uniform
#pragma unroll
  for (int c = 0; c < HD; ++c) {
    const float v = wave_sum(pad_lane ? acc[c] * scale : 0.f);
    if (lane == 0) sdB[c] += v;
  }
}

__device__ __forceinline__ void store_row(float* dst, const float* acc, float scale) {
#pragma unroll
  for (int c = 0; c < HD / 4; ++c)
    reinterpret_cast<float4*>(dst)[c] =
        make_float4(acc[4 * c] * scale, acc[4 * c + 1] * scale, acc[4 * c + 2] * scale, acc[4 * c + 3] * scale);
}

// bias + shift-mask term of score (row i = this lane, column j)
#define WATTN_BIAS(j) \
  (sT[tb + (6 - (j) / WS) * 13 + (6 - (j) % WS)] + ((g.shift > 0 && sLab[j] != me.label) ? -100.0f : 0.f))

__global__ __launch_bounds__(64) void swin_wattn_fwd_kernel(const float* __restrict__ qkv,
                                                            const float* __restrict__ qkv_b,
                                                            const float* __restrict__ table,
                                                            float* __restrict__ out, WinGeom g, int B) {
  __shared__ float4 sK[WN * HD / 4], sV[WN * HD / 4];
  __shared__ float sP[WN * WN];  // row i = lane i's scores
  __shared__ float sT[TBL];
  __shared__ int sLab[WN];
  const int lane = threadIdx.x;
  const int head = blockIdx.x % g.heads;
  const int stride = gridDim.x / g.heads;
  const float scale = 0.17677669529663687f;  // 32^-0.5
  const long L = (long)g.H * g.W;
  for (int t = lane; t < TBL; t += 64) sT[t] = table[t * g.heads + head];
  for (int bw = blockIdx.x / g.heads; bw < B * g.nW; bw += stride) {
    const int b = bw / g.nW, win = bw % g.nW, wy = win / g.nWw, wx = win % g.nWw;
    __syncthreads();  // previous window's readers are done with sK/sV/sLab
    stage_qkv(sK, qkv, qkv_b, g, b, wy, wx, head, 1, lane);
    stage_qkv(sV, qkv, qkv_b, g, b, wy, wx, head, 2, lane);
    if (lane < WN) sLab[lane] = win_token(g, wy, wx, lane).label;
    __syncthreads();
    if (lane < WN) {
      const TokPos me = win_token(g, wy, wx, lane);
      float q[HD];
      if (!me.pad)
        load_row(q, reinterpret_cast<const float4*>(qkv + ((long)b * L + me.tok) * 3 * g.C + head * HD), scale);
      else if (qkv_b)
        load_row(q, reinterpret_cast<const float4*>(qkv_b + head * HD), scale);
      else
        for (int c = 0; c < HD; ++c) q[c] = 0.f;
      const int tb = (lane / WS) * 13 + lane % WS;  // table index = tb + (6 - yj) * 13 + (6 - xj)
      float* row = sP + lane * WN;
      float m = -3.0e38f;
      for_rows(sK, [&](int j) { return WATTN_BIAS(j); }, [&](int j, const Row& k, float bj) {
        const float v = dot_row(q, k) + bj;
        row[j] = v;
        m = fmaxf(m, v);
      });
      float sum = 0.f;
      float o[HD];
#pragma unroll
      for (int c = 0; c < HD; ++c) o[c] = 0.f;
      for_rows(sV, [&](int j) { return row[j]; }, [&](int j, const Row& v, float sj) {
        const float pj = __expf(sj - m);
        sum += pj;
        axpy_row(o, pj, v);
      });
      if (!me.pad) store_row(out + ((long)b * L + me.tok) * g.C + head * HD, o, 1.f / sum);
    }
  }
}

__global__ __launch_bounds__(64) void swin_wattn_bwd_kernel(const float* __restrict__ qkv,
                                                            const float* __restrict__ qkv_b,
                                                            const float* __restrict__ table,
                                                            const float* __restrict__ dout,
                                                            float* __restrict__ dqkv, float* __restrict__ part,
                                                            WinGeom g, int B, int phases) {
  __shared__ float4 sQ[WN * HD / 4], sK[WN * HD / 4], sV[WN * HD / 4], sG[WN * HD / 4];
  __shared__ float sP[WN * WN];
  __shared__ float sT[TBL], sdT[TBL], sdB[3 * HD];
  __shared__ int sLab[WN];
  const int lane = threadIdx.x;
  const int head = blockIdx.x % g.heads;
  const int stride = gridDim.x / g.heads;
  const float scale = 0.17677669529663687f;
  const long L = (long)g.H * g.W;
  for (int t = lane; t < TBL; t += 64) {
    sT[t] = table[t * g.heads + head];
    sdT[t] = 0.f;
  }
  for (int t = lane; t < 3 * HD; t += 64) sdB[t] = 0.f;
  for (int bw = blockIdx.x / g.heads; bw < B * g.nW; bw += stride) {
    const int b = bw / g.nW, win = bw % g.nW, wy = win / g.nWw, wx = win % g.nWw;
    __syncthreads();
    stage_qkv(sQ, qkv, qkv_b, g, b, wy, wx, head, 0, lane);
    stage_qkv(sK, qkv, qkv_b, g, b, wy, wx, head, 1, lane);
    stage_qkv(sV, qkv, qkv_b, g, b, wy, wx, head, 2, lane);
    stage_dout(sG, dout, g, b, wy, wx, head, lane);
    if (lane < WN) sLab[lane] = win_token(g, wy, wx, lane).label;
    __syncthreads();
    if (phases < 1) continue;  // timing ablation (RSCOTR_WATTN_PHASES); never set in production
    const TokPos me = win_token(g, wy, wx, lane < WN ? lane : 0);
    const int tb = (lane / WS) * 13 + lane % WS;
    float* row = sP + (lane < WN ? lane : 0) * WN;
    float* tok_q = dqkv + ((long)b * L + me.tok) * 3 * g.C + head * HD;  // + C: k, + 2C: v
    // ---- 1. probabilities of row i -> sP row i -----------------------------------------------
    if (lane < WN) {
      float q[HD];
      load_row(q, sQ + lane * (HD / 4), scale);
      float m = -3.0e38f;
      for_rows(sK, [&](int j) { return WATTN_BIAS(j); }, [&](int j, const Row& k, float bj) {
        const float v = dot_row(q, k) + bj;
        row[j] = v;
        m = fmaxf(m, v);
      });
      float sum = 0.f;
#pragma unroll 7
      for (int j = 0; j < WN; ++j) {
        const float pj = __expf(row[j] - m);
        row[j] = pj;
        sum += pj;
      }
      const float inv = 1.f / sum;
#pragma unroll 7
      for (int j = 0; j < WN; ++j) row[j] *= inv;
    }
    __syncthreads();
    if (phases < 2) continue;
    // ---- 2. dV_j = sum_i P_ij dO_i  (lane = column j) ------------------------------------------
    float acc[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) acc[c] = 0.f;
    if (lane < WN) {
      for_rows(sG, [&](int i) { return sP[i * WN + lane]; },
               [&](int i, const Row& dO, float pij) { axpy_row(acc, pij, dO); });
      if (!me.pad) store_row(tok_q + 2 * g.C, acc, 1.f);
    }
    pad_bias_grad(sdB + 2 * HD, acc, lane < WN && me.pad, 1.f, lane);
    __syncthreads();  // all columns read sP before the rows overwrite it with dS
    if (phases < 3) continue;
    // ---- 3. dS row i (over P in place), dQ_i, bias-table gradient -----------------------------
    if (lane < WN) {
      float q[HD];  // here: dO_i
      load_row(q, sG + lane * (HD / 4), 1.f);
      float delta = 0.f;
      for_rows(sV, [&](int j) { return row[j]; },
               [&](int j, const Row& v, float pj) { delta += pj * dot_row(q, v); });
#pragma unroll
      for (int c = 0; c < HD; ++c) acc[c] = 0.f;
      for_rows2(sV, sK, [&](int j) { return row[j]; }, [&](int j, const Row& v, const Row& k, float pj) {
        const float ds = pj * (dot_row(q, v) - delta);  // dP_ij recomputed: cheaper than a 2nd matrix
        row[j] = ds;
        // LDS atomic: lanes hit distinct entries for one j, but the same entry across different j
        atomicAdd(&sdT[tb + (6 - j / WS) * 13 + (6 - j % WS)], ds);
        axpy_row(acc, ds, k);
      });
      if (!me.pad) store_row(tok_q, acc, scale);
    } else {
#pragma unroll
      for (int c = 0; c < HD; ++c) acc[c] = 0.f;
    }
    pad_bias_grad(sdB, acc, lane < WN && me.pad, scale, lane);
    __syncthreads();
    if (phases < 4) continue;
    // ---- 4. dK_j = scale * sum_i dS_ij q_i  (lane = column j) ----------------------------------
#pragma unroll
    for (int c = 0; c < HD; ++c) acc[c] = 0.f;
    if (lane < WN) {
      for_rows(sQ, [&](int i) { return sP[i * WN + lane]; },
               [&](int i, const Row& qi, float dsij) { axpy_row(acc, dsij, qi); });
      if (!me.pad) store_row(tok_q + g.C, acc, scale);
    }
    pad_bias_grad(sdB + HD, acc, lane < WN && me.pad, scale, lane);
  }
  __syncthreads();
  // this workgroup's share of the bias-table / pad-token (qkv-bias) gradients: one partial row, folded over the
  // workgroups of the head in fixed order by wattn_param_fold_kernel (no global atomics: bit-reproducible)
  float* prow = part + (long)blockIdx.x * WATTN_PROW;
  for (int t = lane; t < TBL; t += 64) prow[t] = sdT[t];
  for (int t = lane; t < 3 * HD; t += 64) prow[WATTN_PBIAS + t] = sdB[t];
}


// =====================================================================================================
// Matrix-core version.  The 49x49x32 products of a (window, head) are small, but on the VALU they are
// bound by wave-wide LDS broadcasts (8 ds_read_b128 per 32 FMAs, one wavefront per SIMD: measured
// ~590 cycles per key row against ~150 of FMA issue).  v_mfma_f32_32x32x2_f32 runs at the same FLOP rate
// but takes its operands as ONE dword per lane and step, so the same products cost 50-64 MFMAs each:
//   S  = Q K^T          (64 x 64 x 32, both operands row-per-lane reads of [token][33] tiles)
//   O  = P V            (64 x 32 x 50)
//   dP = dO V^T, dV = P^T dO, dQ = dS K, dK = dS^T Q   in backward.
// Operand tiles are 49 rows with an odd stride (33): a lane reading row (l & 31) at a fixed channel is
// conflict-free; rows >= 49 of the 64-row MFMA tiles are clamped reads whose results are discarded or
// multiplied by the zero padding of the score matrix SP (50 x 50, row / column 49 are zeros, stride 51).
// Softmax / dS stay row-per-lane on the VALU (49 lanes x 49 steps), between the MFMA phases.
constexpr int LDT = HD + 1;   // 33
constexpr int NPD = WN + 1;   // 50: padded score dimension (index 49 = zero row / column)
constexpr int LDP = NPD + 1;  // 51

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct TileRegs {
  float4 v[(WN * (HD / 4) + 63) / 64];  // 7
};

__device__ __forceinline__ void tile_load(TileRegs& r, const float* __restrict__ src_tok0, long tok_stride,
                                          const float* __restrict__ pad_vals, const int* sTok, int lane) {
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int idx = lane + i * 64;
    r.v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < WN * (HD / 4)) {
      const int t = idx >> 3, c4 = idx & 7;
      const int tok = sTok[t];
      if (tok >= 0) r.v[i] = *reinterpret_cast<const float4*>(src_tok0 + (long)tok * tok_stride + c4 * 4);
      else if (pad_vals) r.v[i] = *reinterpret_cast<const float4*>(pad_vals + c4 * 4);
    }
  }
}

__device__ __forceinline__ void tile_store(const TileRegs& r, float* dst, float scale, int lane) {
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int idx = lane + i * 64;
    if (idx < WN * (HD / 4)) {
      float* d = dst + (idx >> 3) * LDT + (idx & 7) * 4;
      d[0] = r.v[i].x * scale; d[1] = r.v[i].y * scale; d[2] = r.v[i].z * scale; d[3] = r.v[i].w * scale;
    }
  }
}

// dst[t][33] <- scale * (token t's 32 channels at src_tok0 + tok*tok_stride, or pad_vals for pad tokens)
__device__ __forceinline__ void stage_tile33(float* dst, const float* __restrict__ src_tok0, long tok_stride,
                                             const float* __restrict__ pad_vals, const int* sTok, float scale, int lane) {
  TileRegs r;
  tile_load(r, src_tok0, tok_stride, pad_vals, sTok, lane);
  tile_store(r, dst, scale, lane);
}

// acc[mi][ni] += sum_k A[i][k] * B[j][k]; i = mi*32 + (l&31), j = ni*32 + (l&31), rows clamped to 48 (results of
// clamped rows are never used); k = 0..31
__device__ __forceinline__ void mma_rr(f32x16 (&acc)[2][2], const float* A, const float* B, int lane) {
  const int fr = lane & 31, fk = lane >> 5;
  const float* a0 = A + fr * LDT + fk;
  const float* a1 = A + min(32 + fr, WN - 1) * LDT + fk;
  const float* b0 = B + fr * LDT + fk;
  const float* b1 = B + min(32 + fr, WN - 1) * LDT + fk;
#pragma unroll
  for (int kk = 0; kk < HD; kk += 2) {
    const float x0 = a0[kk], x1 = a1[kk], y0 = b0[kk], y1 = b1[kk];
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y1, acc[1][1], 0, 0, 0);
  }
}

// acc[mi] (rows i, 32 channels) += sum_{k<50} SP[i][k] * T[k][c]; SP rows >= 49 read the zero row, T row 49 is a
// clamped read multiplied by the zero column of SP
__device__ __forceinline__ void mma_pt(f32x16 (&acc)[2], const float* SP, const float* T, int lane) {
  const int fr = lane & 31, fk = lane >> 5;
  const float* a0 = SP + fr * LDP + fk;
  const float* a1 = SP + min(32 + fr, WN) * LDP + fk;
#pragma unroll 5
  for (int kk = 0; kk < NPD; kk += 2) {
    const float x0 = a0[kk], x1 = a1[kk];
    const float y = T[min(kk + fk, WN - 1) * LDT + fr];
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y, acc[1], 0, 0, 0);
  }
}

// acc[mi] (rows j, 32 channels) += sum_{k<50} SP[k][j] * T[k][c]  (SP^T T); columns j >= 49 read the zero column
__device__ __forceinline__ void mma_ptT(f32x16 (&acc)[2], const float* SP, const float* T, int lane) {
  const int fr = lane & 31, fk = lane >> 5;
  const int j0 = fr, j1 = min(32 + fr, WN);
#pragma unroll 5
  for (int kk = 0; kk < NPD; kk += 2) {
    const float* row = SP + (kk + fk) * LDP;
    const float x0 = row[j0], x1 = row[j1];
    const float y = T[min(kk + fk, WN - 1) * LDT + fr];
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, y, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, y, acc[1], 0, 0, 0);
  }
}

__device__ __forceinline__ void zero16(f32x16& a) {
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
}

// row of C element r held by this lane inside a 32-row tile
__device__ __forceinline__ int crow(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// S accumulators (+ relative-position bias, + shift mask) -> sP[i][j]
__device__ __forceinline__ void store_scores(const f32x16 (&acc)[2][2], float* sP, const float* sT, const int* sLab,
                                             int shift, int lane) {
  const int fr = lane & 31;
  int jb[2], jl[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int j = min(ni * 32 + fr, WN - 1);
    jb[ni] = (6 - j / WS) * 13 + (6 - j % WS);
    jl[ni] = sLab[j];
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = mi * 32 + crow(r, lane);
      if (i >= WN) continue;
      const int ti = (i / WS) * 13 + i % WS;
      const int li = sLab[i];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int j = ni * 32 + fr;
        if (j < WN)
          sP[i * LDP + j] = acc[mi][ni][r] + sT[ti + jb[ni]] + ((shift > 0 && jl[ni] != li) ? -100.0f : 0.f);
      }
    }
}

// softmax of row `lane` (< 49) of sP in place: three passes of 7 x 7 independent LDS accesses
__device__ __forceinline__ void row_softmax(float* row) {
  float m = -3.0e38f;
#pragma unroll
  for (int j0 = 0; j0 < WN; j0 += WS) {
    float v[WS];
#pragma unroll
    for (int u = 0; u < WS; ++u) v[u] = row[j0 + u];
#pragma unroll
    for (int u = 0; u < WS; ++u) m = fmaxf(m, v[u]);
  }
  float sum = 0.f;
#pragma unroll
  for (int j0 = 0; j0 < WN; j0 += WS) {
    float v[WS];
#pragma unroll
    for (int u = 0; u < WS; ++u) v[u] = __expf(row[j0 + u] - m);
#pragma unroll
    for (int u = 0; u < WS; ++u) { row[j0 + u] = v[u]; sum += v[u]; }
  }
  const float inv = 1.f / sum;
#pragma unroll
  for (int j0 = 0; j0 < WN; j0 += WS) {
    float v[WS];
#pragma unroll
    for (int u = 0; u < WS; ++u) v[u] = row[j0 + u] * inv;
#pragma unroll
    for (int u = 0; u < WS; ++u) row[j0 + u] = v[u];
  }
}

__global__ __launch_bounds__(64) void swin_wattn_fwd_mfma_kernel(const float* __restrict__ qkv,
                                                                 const float* __restrict__ qkv_b,
                                                                 const float* __restrict__ table,
                                                                 float* __restrict__ out, WinGeom g, int B) {
  __shared__ float sQ[WN * LDT], sK[WN * LDT], sV[WN * LDT];
  __shared__ float sP[NPD * LDP];
  __shared__ float sT[TBL];
  __shared__ int sLab[WN], sTok[WN];
  const int lane = threadIdx.x;
  const int head = blockIdx.x % g.heads;
  const int stride = gridDim.x / g.heads;
  const float scale = 0.17677669529663687f;
  const long L = (long)g.H * g.W;
  for (int t = lane; t < TBL; t += 64) sT[t] = table[t * g.heads + head];
  for (int t = lane; t < NPD * LDP; t += 64) sP[t] = 0.f;  // the padding row / column stay zero
  for (int bw = blockIdx.x / g.heads; bw < B * g.nW; bw += stride) {
    const int b = bw / g.nW, win = bw % g.nW, wy = win / g.nWw, wx = win % g.nWw;
    __syncthreads();
    TokPos me = win_token(g, wy, wx, lane < WN ? lane : 0);
    if (lane < WN) {
      sLab[lane] = me.label;
      sTok[lane] = me.pad ? -1 : (int)me.tok;
    }
    __syncthreads();
    const float* base = qkv + (long)b * L * 3 * g.C + head * HD;
    {
      TileRegs rq, rk, rv;
      tile_load(rq, base, 3 * g.C, qkv_b ? qkv_b + head * HD : nullptr, sTok, lane);
      tile_load(rk, base + g.C, 3 * g.C, qkv_b ? qkv_b + g.C + head * HD : nullptr, sTok, lane);
      tile_load(rv, base + 2 * g.C, 3 * g.C, qkv_b ? qkv_b + 2 * g.C + head * HD : nullptr, sTok, lane);
      tile_store(rq, sQ, scale, lane);
      tile_store(rk, sK, 1.f, lane);
      tile_store(rv, sV, 1.f, lane);
    }
    __syncthreads();
    {  // S = (q * scale) k^T -> sP[i][j]
      f32x16 acc[2][2];
      zero16(acc[0][0]); zero16(acc[0][1]); zero16(acc[1][0]); zero16(acc[1][1]);
      mma_rr(acc, sQ, sK, lane);
      store_scores(acc, sP, sT, sLab, g.shift, lane);
    }
    __syncthreads();
    if (lane < WN) row_softmax(sP + lane * LDP);
    __syncthreads();
    {  // O = P v
      f32x16 o[2];
      zero16(o[0]); zero16(o[1]);
      mma_pt(o, sP, sV, lane);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = mi * 32 + crow(r, lane);
          if (i < WN) {
            const int tok = sTok[i];
            if (tok >= 0) out[((long)b * L + tok) * g.C + head * HD + (lane & 31)] = o[mi][r];
          }
        }
    }
  }
}


// sum of v over the 32 lanes of a half-wave (lanes l and l^32 never mix)
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(64) void swin_wattn_bwd_mfma_kernel(const float* __restrict__ qkv,
                                                                 const float* __restrict__ qkv_b,
                                                                 const float* __restrict__ table,
                                                                 const float* __restrict__ dout,
                                                                 float* __restrict__ dqkv, float* __restrict__ part,
                                                                 WinGeom g, int B) {
  __shared__ float sQ[WN * LDT], sK[WN * LDT], sV[WN * LDT], sG[WN * LDT];
  __shared__ float sP[NPD * LDP];
  __shared__ float sT[TBL], sdT[TBL], sdB[3 * HD];
  __shared__ int sLab[WN], sTok[WN];
  const int lane = threadIdx.x, fr = lane & 31;
  const int head = blockIdx.x % g.heads;
  const int stride = gridDim.x / g.heads;
  const float scale = 0.17677669529663687f;
  const long L = (long)g.H * g.W;
  for (int t = lane; t < TBL; t += 64) {
    sT[t] = table[t * g.heads + head];
    sdT[t] = 0.f;
  }
  for (int t = lane; t < 3 * HD; t += 64) sdB[t] = 0.f;
  for (int t = lane; t < NPD * LDP; t += 64) sP[t] = 0.f;
  for (int bw = blockIdx.x / g.heads; bw < B * g.nW; bw += stride) {
    const int b = bw / g.nW, win = bw % g.nW, wy = win / g.nWw, wx = win % g.nWw;
    __syncthreads();
    const TokPos me = win_token(g, wy, wx, lane < WN ? lane : 0);
    if (lane < WN) {
      sLab[lane] = me.label;
      sTok[lane] = me.pad ? -1 : (int)me.tok;
    }
    __syncthreads();
    const float* base = qkv + (long)b * L * 3 * g.C + head * HD;
    {  // the 28 loads of the four tiles are issued before the first LDS write (one wavefront: nothing else
       // hides their latency)
      TileRegs rq, rk, rv, rg;
      tile_load(rq, base, 3 * g.C, qkv_b ? qkv_b + head * HD : nullptr, sTok, lane);
      tile_load(rk, base + g.C, 3 * g.C, qkv_b ? qkv_b + g.C + head * HD : nullptr, sTok, lane);
      tile_load(rv, base + 2 * g.C, 3 * g.C, qkv_b ? qkv_b + 2 * g.C + head * HD : nullptr, sTok, lane);
      tile_load(rg, dout + (long)b * L * g.C + head * HD, g.C, nullptr, sTok, lane);  // dO, zero on pads
      tile_store(rq, sQ, scale, lane);  // q * scale
      tile_store(rk, sK, 1.f, lane);
      tile_store(rv, sV, 1.f, lane);
      tile_store(rg, sG, 1.f, lane);
    }
    __syncthreads();
    bool any_pad = false;
    for (int t = 0; t < WN; ++t) any_pad |= sTok[t] < 0;  // wave-uniform
    float* dq_base = dqkv + (long)b * L * 3 * g.C + head * HD;
    {  // ---- 1. S -> sP, row softmax -> P
      f32x16 acc[2][2];
      zero16(acc[0][0]); zero16(acc[0][1]); zero16(acc[1][0]); zero16(acc[1][1]);
      mma_rr(acc, sQ, sK, lane);
      store_scores(acc, sP, sT, sLab, g.shift, lane);
    }
    __syncthreads();
    if (lane < WN) row_softmax(sP + lane * LDP);
    __syncthreads();
    {  // ---- 2. dV = P^T dO  (rows = key j)
      f32x16 dv[2];
      zero16(dv[0]); zero16(dv[1]);
      mma_ptT(dv, sP, sG, lane);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = mi * 32 + crow(r, lane);
          if (j < WN) {
            const int tok = sTok[j];
            if (tok >= 0) dq_base[(long)tok * 3 * g.C + 2 * g.C + fr] = dv[mi][r];
            else atomicAdd(&sdB[2 * HD + fr], dv[mi][r]);
          }
        }
    }
    // ---- 3. dP = dO V^T (registers), delta_i = sum_j P_ij dP_ij, dS = P (dP - delta) -> sP, bias-table gradient
    {
      f32x16 dp[2][2];
      zero16(dp[0][0]); zero16(dp[0][1]); zero16(dp[1][0]); zero16(dp[1][1]);
      mma_rr(dp, sG, sV, lane);
      __syncthreads();  // every lane is done reading P as an MFMA operand (step 2) before it is overwritten
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = mi * 32 + crow(r, lane);  // same row for the 32 lanes of a half-wave
          const bool iv = i < WN;
          const int j0 = fr, j1 = 32 + fr;
          const float p0 = iv ? sP[i * LDP + j0] : 0.f;
          const float p1 = (iv && j1 < WN) ? sP[i * LDP + j1] : 0.f;
          const float delta = half_sum(p0 * dp[mi][0][r] + p1 * dp[mi][1][r]);
          if (iv) {
            const int ti = (i / WS) * 13 + i % WS;
            const float ds0 = p0 * (dp[mi][0][r] - delta);
            sP[i * LDP + j0] = ds0;
            atomicAdd(&sdT[ti + (6 - j0 / WS) * 13 + (6 - j0 % WS)], ds0);
            if (j1 < WN) {
              const float ds1 = p1 * (dp[mi][1][r] - delta);
              sP[i * LDP + j1] = ds1;
              atomicAdd(&sdT[ti + (6 - j1 / WS) * 13 + (6 - j1 % WS)], ds1);
            }
          }
        }
    }
    __syncthreads();
    {  // ---- 4. dQ = scale * dS K   (rows = query i)
      f32x16 dq[2];
      zero16(dq[0]); zero16(dq[1]);
      mma_pt(dq, sP, sK, lane);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = mi * 32 + crow(r, lane);
          if (i < WN) {
            const int tok = sTok[i];
            if (tok >= 0) dq_base[(long)tok * 3 * g.C + fr] = dq[mi][r] * scale;
            else atomicAdd(&sdB[fr], dq[mi][r] * scale);
          }
        }
    }
    {  // ---- 5. dK = dS^T (q * scale)   (rows = key j; sQ already holds q * scale)
      f32x16 dk[2];
      zero16(dk[0]); zero16(dk[1]);
      mma_ptT(dk, sP, sQ, lane);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = mi * 32 + crow(r, lane);
          if (j < WN) {
            const int tok = sTok[j];
            if (tok >= 0) dq_base[(long)tok * 3 * g.C + g.C + fr] = dk[mi][r];
            else atomicAdd(&sdB[HD + fr], dk[mi][r]);
          }
        }
    }
    (void)any_pad;
  }
  __syncthreads();
  // this workgroup's share of the bias-table / pad-token (qkv-bias) gradients: one partial row, folded over the
  // workgroups of the head in fixed order by wattn_param_fold_kernel (no global atomics: bit-reproducible)
  float* prow = part + (long)blockIdx.x * WATTN_PROW;
  for (int t = lane; t < TBL; t += 64) prow[t] = sdT[t];
  for (int t = lane; t < 3 * HD; t += 64) prow[WATTN_PBIAS + t] = sdB[t];
}

// =====================================================================================================
// Four wavefronts per (window, head) item.  One wavefront per item (above) is a chain of ~280 dependent-latency MFMAs,
// five LDS phases and a 49-step row softmax: ~50 us per item however few items there are (stages 3-4 have 600 items for
// 1024 SIMDs).  Here the 256 threads of a workgroup share one item: each wavefront stages one of the four operand tiles,
// owns one 32 x 32 tile of S, the softmax runs four lanes per row, and the backward products are spread as
//   waves 0,1: dP rows 0-31 / 32-63 (kept in registers) -> dS for those rows -> dQ rows 0-31 / 32-63
//   waves 2,3: dV rows 0-31 / 32-63                                          -> dK rows 0-31 / 32-63
// so the MFMA chain of an item is 16 + 32 + 25 instead of 278 issues.  The relative-position-bias gradient is a GATHER
// (thread t < 169 sums dS over the <= 49 (i, j) pairs of its table entry in fixed order, in a register across the items
// of the workgroup) instead of ~64 LDS float atomics per lane and item, and the pad-token (qkv-bias) sums are per-lane
// registers folded wave by wave at the end: no atomics at all, bit-reproducible.
__device__ __forceinline__ void mma_rr1(f32x16& acc, const float* A, const float* B, int mi, int ni, int lane) {
  const int fr = lane & 31, fk = lane >> 5;
  const float* a = A + min(mi * 32 + fr, WN - 1) * LDT + fk;
  const float* b = B + min(ni * 32 + fr, WN - 1) * LDT + fk;
#pragma unroll
  for (int kk = 0; kk < HD; kk += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], b[kk], acc, 0, 0, 0);
}

__device__ __forceinline__ void mma_pt1(f32x16& acc, const float* SP, const float* T, int mi, int lane) {
  const int fr = lane & 31, fk = lane >> 5;
  const float* a = SP + min(mi * 32 + fr, WN) * LDP + fk;
#pragma unroll 5
  for (int kk = 0; kk < NPD; kk += 2)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kk], T[min(kk + fk, WN - 1) * LDT + fr], acc, 0, 0, 0);
}

__device__ __forceinline__ void mma_ptT1(f32x16& acc, const float* SP, const float* T, int mi, int lane) {
  const int fr = lane & 31, fk = lane >> 5;
  const int j = min(mi * 32 + fr, WN);
#pragma unroll 5
  for (int kk = 0; kk < NPD; kk += 2)
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(SP[(kk + fk) * LDP + j], T[min(kk + fk, WN - 1) * LDT + fr], acc, 0, 0, 0);
}

__device__ __forceinline__ void store_scores1(const f32x16& acc, float* sP, const float* sT, const int* sLab, int shift,
                                              int mi, int ni, int lane) {
  const int fr = lane & 31;
  const int j = ni * 32 + fr, jc = min(j, WN - 1);
  const int jb = (6 - jc / WS) * 13 + (6 - jc % WS), jl = sLab[jc];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = mi * 32 + crow(r, lane);
    if (i >= WN || j >= WN) continue;
    const int ti = (i / WS) * 13 + i % WS;
    sP[i * LDP + j] = acc[r] + sT[ti + jb] + ((shift > 0 && jl != sLab[i]) ? -100.0f : 0.f);
  }
}

// softmax of the 49 rows of sP in place, four lanes per row (all 256 threads call it)
__device__ __forceinline__ void softmax_rows4(float* sP, int tid) {
  const int row = tid >> 2, q = tid & 3;
  float* rowp = sP + min(row, WN - 1) * LDP;
  float v[13];
  float m = -3.0e38f;
#pragma unroll
  for (int u = 0; u < 13; ++u) {
    const int j = q + 4 * u;
    v[u] = j < WN ? rowp[j] : -3.0e38f;
    m = fmaxf(m, v[u]);
  }
  m = fmaxf(m, __shfl_xor(m, 1, 64));
  m = fmaxf(m, __shfl_xor(m, 2, 64));
  float sum = 0.f;
#pragma unroll
  for (int u = 0; u < 13; ++u) {
    v[u] = (q + 4 * u < WN) ? __expf(v[u] - m) : 0.f;
    sum += v[u];
  }
  sum += __shfl_xor(sum, 1, 64);
  sum += __shfl_xor(sum, 2, 64);
  const float inv = 1.f / sum;
  if (row < WN) {
#pragma unroll
    for (int u = 0; u < 13; ++u)
      if (q + 4 * u < WN) rowp[q + 4 * u] = v[u] * inv;
  }
}

// wave w stages operand tile w of the item (0: q * scale, 1: k, 2: v, 3: dO when `dout` is given)
__device__ __forceinline__ void stage_item_tiles(float* sQ, float* sK, float* sV, float* sG, const float* base,
                                                 const float* __restrict__ qkv_b, const float* dout_b, const WinGeom& g,
                                                 int head, const int* sTok, float scale, int w, int lane) {
  if (w == 3 && !dout_b) return;
  const float* src = w == 3 ? dout_b : base + w * g.C;
  const long tstride = w == 3 ? g.C : 3 * g.C;
  const float* padv = (w == 3 || !qkv_b) ? nullptr : qkv_b + w * g.C + head * HD;
  float* dst = w == 0 ? sQ : (w == 1 ? sK : (w == 2 ? sV : sG));
  TileRegs r;
  tile_load(r, src, tstride, padv, sTok, lane);
  tile_store(r, dst, w == 0 ? scale : 1.f, lane);
}

__global__ __launch_bounds__(256) void swin_wattn_fwd_mfma4_kernel(const float* __restrict__ qkv,
                                                                   const float* __restrict__ qkv_b,
                                                                   const float* __restrict__ table,
                                                                   float* __restrict__ out, WinGeom g, int B,
                                                                   unsigned* __restrict__ amax_out) {
  __shared__ float sQ[WN * LDT], sK[WN * LDT], sV[WN * LDT];
  __shared__ float sP[NPD * LDP];
  __shared__ float sT[TBL];
  __shared__ int sLab[WN], sTok[WN];
  float amx = 0.f;  // max |out| -> the output's range word (the proj Linear multiplies with it)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int head = blockIdx.x % g.heads;
  const int stride = gridDim.x / g.heads;
  const float scale = 0.17677669529663687f;
  const long L = (long)g.H * g.W;
  for (int t = tid; t < TBL; t += 256) sT[t] = table[t * g.heads + head];
  for (int t = tid; t < NPD * LDP; t += 256) sP[t] = 0.f;  // the padding row / column stay zero
  for (int bw = blockIdx.x / g.heads; bw < B * g.nW; bw += stride) {
    const int b = bw / g.nW, win = bw % g.nW, wy = win / g.nWw, wx = win % g.nWw;
    __syncthreads();
    if (tid < WN) {
      const TokPos me = win_token(g, wy, wx, tid);
      sLab[tid] = me.label;
      sTok[tid] = me.pad ? -1 : (int)me.tok;
    }
    __syncthreads();
    stage_item_tiles(sQ, sK, sV, nullptr, qkv + (long)b * L * 3 * g.C + head * HD, qkv_b, nullptr, g, head, sTok, scale, w, lane);
    __syncthreads();
    {
      f32x16 a;
      zero16(a);
      mma_rr1(a, sQ, sK, w >> 1, w & 1, lane);
      store_scores1(a, sP, sT, sLab, g.shift, w >> 1, w & 1, lane);
    }
    __syncthreads();
    softmax_rows4(sP, tid);
    __syncthreads();
    if (w < 2) {  // O rows 32 w .. 32 w + 31
      f32x16 o;
      zero16(o);
      mma_pt1(o, sP, sV, w, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = w * 32 + crow(r, lane);
        if (i < WN) {
          const int tok = sTok[i];
          if (tok >= 0) { out[((long)b * L + tok) * g.C + head * HD + (lane & 31)] = o[r]; amx = fmaxf(amx, fabsf(o[r])); }
        }
      }
    }
  }
  amax_commit(amax_out, amx);
}

__device__ __forceinline__ void swin_wattn_bwd_mfma4_body(const float* __restrict__ qkv,
                                                                   const float* __restrict__ qkv_b,
                                                                   const float* __restrict__ table,
                                                                   const float* __restrict__ dout,
                                                                   const float* __restrict__ outp,
                                                                   float* __restrict__ dqkv, float* __restrict__ part,
                                                                   const WinGeom& g, int B, unsigned* __restrict__ amax_out) {
  __shared__ float sQ[WN * LDT], sK[WN * LDT], sV[WN * LDT], sG[WN * LDT];
  __shared__ float sP[NPD * LDP];
  __shared__ float sT[TBL];
  __shared__ float sBw[4][3 * HD];
  float amx = 0.f;  // max |dqkv| of what this lane stores -> the gradient's range word (common.h: amax_commit)
  __shared__ float sDelta[64];
  __shared__ int sLab[WN], sTok[WN];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fr = lane & 31;
  const int head = blockIdx.x % g.heads;
  const int stride = gridDim.x / g.heads;
  const float scale = 0.17677669529663687f;
  const long L = (long)g.H * g.W;
  for (int t = tid; t < TBL; t += 256) sT[t] = table[t * g.heads + head];
  for (int t = tid; t < NPD * LDP; t += 256) sP[t] = 0.f;
  float dT = 0.f;                        // thread t < 169: gradient of table entry t over this workgroup's items
  float accB[3] = {0.f, 0.f, 0.f};       // pad-token sums of this lane's rows: dq (waves 0,1), dk, dv (waves 2,3), column fr
  const int tdy = tid / 13 - 6, tdx = tid % 13 - 6;
  const int iy0 = max(0, tdy), iy1 = min(6, 6 + tdy), ix0 = max(0, tdx), ix1 = min(6, 6 + tdx);
  for (int bw = blockIdx.x / g.heads; bw < B * g.nW; bw += stride) {
    const int b = bw / g.nW, win = bw % g.nW, wy = win / g.nWw, wx = win % g.nWw;
    __syncthreads();
    if (tid < WN) {
      const TokPos me = win_token(g, wy, wx, tid);
      sLab[tid] = me.label;
      sTok[tid] = me.pad ? -1 : (int)me.tok;
    }
    __syncthreads();
    // delta_i = sum_j P_ij dP_ij = dO_i . O_i (O = the forward output of this head): with the forward output at hand
    // it is a 32-channel dot product per row (four lanes per row) instead of a cross-lane reduction per dS row
    float4 o_a = make_float4(0.f, 0.f, 0.f, 0.f), o_b = o_a;
    if (outp && (tid >> 2) < WN) {
      const int tok = sTok[tid >> 2];
      if (tok >= 0) {
        const float4* o4 = reinterpret_cast<const float4*>(outp + ((long)b * L + tok) * g.C + head * HD + (tid & 3) * 8);
        o_a = o4[0];
        o_b = o4[1];
      }
    }
    stage_item_tiles(sQ, sK, sV, sG, qkv + (long)b * L * 3 * g.C + head * HD, qkv_b, dout + (long)b * L * g.C + head * HD, g,
                     head, sTok, scale, w, lane);
    __syncthreads();
    float* dq_base = dqkv + (long)b * L * 3 * g.C + head * HD;
    if (outp) {
      const float* gq = sG + min(tid >> 2, WN - 1) * LDT + (tid & 3) * 8;
      float d = ((o_a.x * gq[0] + o_a.y * gq[1]) + (o_a.z * gq[2] + o_a.w * gq[3])) +
                ((o_b.x * gq[4] + o_b.y * gq[5]) + (o_b.z * gq[6] + o_b.w * gq[7]));
      d += __shfl_xor(d, 1, 64);
      d += __shfl_xor(d, 2, 64);
      if ((tid & 3) == 0) sDelta[tid >> 2] = d;  // (read after the barriers below)
    }
    {  // ---- 1. S tile (w >> 1, w & 1) -> sP
      f32x16 a;
      zero16(a);
      mma_rr1(a, sQ, sK, w >> 1, w & 1, lane);
      store_scores1(a, sP, sT, sLab, g.shift, w >> 1, w & 1, lane);
    }
    __syncthreads();
    softmax_rows4(sP, tid);
    __syncthreads();
    // ---- 2. waves 0,1: dP rows of tile w (both column tiles, registers); waves 2,3: dV rows of tile w - 2
    f32x16 dp[2];
    if (w < 2) {
      zero16(dp[0]); zero16(dp[1]);
      mma_rr1(dp[0], sG, sV, w, 0, lane);
      mma_rr1(dp[1], sG, sV, w, 1, lane);
    } else {
      f32x16 dv;
      zero16(dv);
      mma_ptT1(dv, sP, sG, w - 2, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = (w - 2) * 32 + crow(r, lane);
        if (j < WN) {
          const int tok = sTok[j];
          if (tok >= 0) { dq_base[(long)tok * 3 * g.C + 2 * g.C + fr] = dv[r]; amx = fmaxf(amx, fabsf(dv[r])); }
          else accB[2] += dv[r];
        }
      }
    }
    __syncthreads();  // P has been consumed as an MFMA operand before the rows overwrite it with dS
    if (w < 2) {      // ---- 3. delta_i = sum_j P_ij dP_ij, dS = P (dP - delta) -> sP
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = w * 32 + crow(r, lane);  // same row for the 32 lanes of a half-wave
        const bool iv = i < WN;
        const int j0 = fr, j1 = 32 + fr;
        const float p0 = iv ? sP[i * LDP + j0] : 0.f;
        const float p1 = (iv && j1 < WN) ? sP[i * LDP + j1] : 0.f;
        const float delta = outp ? sDelta[min(i, 63)] : half_sum(p0 * dp[0][r] + p1 * dp[1][r]);
        if (iv) {
          sP[i * LDP + j0] = p0 * (dp[0][r] - delta);
          if (j1 < WN) sP[i * LDP + j1] = p1 * (dp[1][r] - delta);
        }
      }
    }
    __syncthreads();
    if (w < 2) {  // ---- 4. dQ = scale * dS K   (rows = query i)
      f32x16 dq;
      zero16(dq);
      mma_pt1(dq, sP, sK, w, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = w * 32 + crow(r, lane);
        if (i < WN) {
          const int tok = sTok[i];
          if (tok >= 0) { dq_base[(long)tok * 3 * g.C + fr] = dq[r] * scale; amx = fmaxf(amx, fabsf(dq[r] * scale)); }
          else accB[0] += dq[r] * scale;
        }
      }
    } else {      // ---- 5. dK = dS^T (q * scale)   (rows = key j; sQ already holds q * scale)
      f32x16 dk;
      zero16(dk);
      mma_ptT1(dk, sP, sQ, w - 2, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = (w - 2) * 32 + crow(r, lane);
        if (j < WN) {
          const int tok = sTok[j];
          if (tok >= 0) { dq_base[(long)tok * 3 * g.C + g.C + fr] = dk[r]; amx = fmaxf(amx, fabsf(dk[r])); }
          else accB[1] += dk[r];
        }
      }
    }
    if (tid < TBL) {  // bias-table gradient: entry (dy, dx) = sum of dS over the pairs i - j = (dy, dx), fixed order
      // (constant trip counts + predicates: the 49 LDS reads are issued together instead of one dependent read per add)
      float acc7 = 0.f;
#pragma unroll 1
      for (int iy = iy0; iy <= iy1; ++iy) {  // (seven reads in flight per row of the window: registers)
        float v[WS];
#pragma unroll
        for (int ix = 0; ix < WS; ++ix) {
          const bool ok = ix >= ix0 && ix <= ix1;
          v[ix] = ok ? sP[(iy * WS + ix) * LDP + (iy - tdy) * WS + (ix - tdx)] : 0.f;
        }
        acc7 += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + v[6]);
      }
      dT += acc7;
    }
  }
  // this workgroup's share of the bias-table / pad-token (qkv-bias) gradients: one partial row, folded over the
  // workgroups of the head in fixed order by wattn_param_fold_kernel
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float other = __shfl_xor(accB[k], 32, 64);
    if (lane < 32) sBw[w][k * HD + fr] = accB[k] + other;
  }
  __syncthreads();
  float* prow = part + (long)blockIdx.x * WATTN_PROW;
  if (tid < TBL) prow[tid] = dT;
  if (tid < 3 * HD) prow[WATTN_PBIAS + tid] = ((sBw[0][tid] + sBw[1][tid]) + sBw[2][tid]) + sBw[3][tid];
  amax_commit(amax_out, amx);
}

// Resident workgroups per CU = wavefronts per SIMD: LDS allows four (39 KB each); the registers decide — 2: 202 VGPRs, no
// scratch; 3: 168 + 92 B of scratch per lane; 4: 128 + 252 B.  OCC picks the budget (rscotr_swin_wattn_bwd measures which wins).
template <int OCC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OCC, OCC))) void swin_wattn_bwd_mfma4_kernel(
    const float* __restrict__ qkv, const float* __restrict__ qkv_b, const float* __restrict__ table,
    const float* __restrict__ dout, const float* __restrict__ outp, float* __restrict__ dqkv, float* __restrict__ part,
    WinGeom g, int B, unsigned* __restrict__ amax_out) {
  swin_wattn_bwd_mfma4_body(qkv, qkv_b, table, dout, outp, dqkv, part, g, B, amax_out);
}

static int wattn_geom(const char* fn, WinGeom* g, int B, int H, int W, int C, int heads, int ws, int shift) {
  if (B < 0 || H <= 0 || W <= 0 || C <= 0 || heads <= 0)
    return fail(RSCOTR_E_SHAPE, "%s: bad shape B=%d H=%d W=%d C=%d heads=%d", fn, B, H, W, C, heads);
  if (ws != WS) return fail(RSCOTR_E_SHAPE, "%s: window size %d (only 7 is built)", fn, ws);
  if (C != heads * HD) return fail(RSCOTR_E_SHAPE, "%s: C=%d must be heads*32 (heads=%d)", fn, C, heads);
  if (shift < 0 || shift >= WS) return fail(RSCOTR_E_SHAPE, "%s: shift %d outside [0,7)", fn, shift);
  g->H = H; g->W = W; g->C = C; g->heads = heads; g->shift = shift;
  g->Hp = (H + WS - 1) / WS * WS;
  g->Wp = (W + WS - 1) / WS * WS;
  g->nWw = g->Wp / WS;
  g->nW = (g->Hp / WS) * g->nWw;
  return RSCOTR_OK;
}

static int wattn_grid(const WinGeom& g, int B, int per_cu) {
  const long items = (long)B * g.nW;                       // windows per head
  long per_head = std::min<long>(items, std::max<long>(1, (256L * per_cu) / g.heads));
  return (int)(per_head * g.heads);
}

}  // namespace rscotr

using namespace rscotr;

extern "C" int rscotr_swin_wattn_fwd(const float* qkv, const float* qkv_bias, const float* bias_table,
                                     float* out, int B, int H, int W, int C, int heads, int ws, int shift,
                                     uint32_t* amax_out, void* stream) {
  WinGeom g;
  if (int e = wattn_geom("rscotr_swin_wattn_fwd", &g, B, H, W, C, heads, ws, shift)) return e;
  if (B == 0) return RSCOTR_OK;
  if (!qkv || !bias_table || !out) return fail(RSCOTR_E_ARG, "rscotr_swin_wattn_fwd: null pointer");
  if (!aligned16(qkv) || !aligned16(out) || (qkv_bias && !aligned16(qkv_bias)))
    return fail(RSCOTR_E_ALIGN, "rscotr_swin_wattn_fwd: pointers must be 16-byte aligned");
  // forward: latency-bound per (window, head) item either way.  The VALU kernel needs less LDS (7 instead of 4-5
  // resident workgroups per CU) and wins when there are more items than resident wavefronts (stages 1-2: 45 vs 47 us,
  // 25 vs 31 us); the matrix-core kernel has the shorter per-item chain (stages 3-4: 16.5 vs 21 us).
  static const int force = getenv("RSCOTR_WATTN_IMPL") ? atoi(getenv("RSCOTR_WATTN_IMPL")) : -1;  // 0 VALU, 1 MFMA
  const int impl = force >= 0 ? force : 2;  // 2: four wavefronts per item
  // algorithmic work: q k^T and P v of every (image, window, head) on the 49 real tokens: 4 * 49 * 49 * 32 flop
  const double items = (double)B * ((H + ws - 1) / ws) * ((W + ws - 1) / ws) * heads;
  ProfScope prof(PROF_MFMA, items * 4.0 * 49 * 49 * 32, (hipStream_t)stream, "rscotr::swin_wattn_fwd_kernel");
  if (impl == 0)
    swin_wattn_fwd_kernel<<<wattn_grid(g, B, 8), 64, 0, (hipStream_t)stream>>>(qkv, qkv_bias, bias_table, out, g, B);
  else if (impl == 1)
    swin_wattn_fwd_mfma_kernel<<<wattn_grid(g, B, 4), 64, 0, (hipStream_t)stream>>>(qkv, qkv_bias, bias_table, out, g, B);
  else
    swin_wattn_fwd_mfma4_kernel<<<wattn_grid(g, B, 4), 256, 0, (hipStream_t)stream>>>(qkv, qkv_bias, bias_table, out, g, B, amax_out);
  if (int e = check_launch("rscotr_swin_wattn_fwd")) return e;
  // (only the four-wavefront kernel folds the output's range itself)
  return (amax_out && impl != 2) ? rscotr_amax_f32(out, (int64_t)B * H * W, C, C, amax_out, stream) : RSCOTR_OK;
}

// workgroups per head of the backward launch (the partial rows of a head are rows head, head + heads, ...)
static int wattn_bwd_grid(const WinGeom& g, int B) { return wattn_grid(g, B, 4); }

namespace rscotr {
// grid (ceil(WATTN_PROW / 64), heads): column block x head; the 4 wavefronts take the head's partial rows round-robin,
// fold through LDS in fixed order, and ADD into dtable[t][head] / dqkv_b[k * C + head * HD + c].
__global__ __launch_bounds__(256) void wattn_param_fold_kernel(const float* __restrict__ part, float* __restrict__ dtable,
                                                               float* __restrict__ dqkv_b, int heads, int C, int rows) {
  __shared__ float red[3][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane, head = blockIdx.y;
  float a = 0.f;
  if (col < WATTN_PROW) {
#pragma unroll 8
    for (int r = w; r < rows; r += 4) a += part[((long)r * heads + head) * WATTN_PROW + col];
  }
  if (w > 0) red[w - 1][lane] = a;
  __syncthreads();
  if (w != 0 || col >= WATTN_PROW) return;
  a += red[0][lane];
  a += red[1][lane];
  a += red[2][lane];
  if (col < TBL) {
    if (dtable) dtable[col * heads + head] += a;
  } else if (col >= WATTN_PBIAS && dqkv_b) {
    const int t = col - WATTN_PBIAS;
    dqkv_b[(t / HD) * C + head * HD + (t % HD)] += a;
  }
}

// The same fold for ALL pending window-attention backward passes of a backward pass in one launch
// (rscotr_swin_wattn_flush): grid (ceil(WATTN_PROW / 64), total heads); table rows {partial rows, dtable | 0,
// dqkv_bias | 0, heads, C, rows per head, first head of this entry in the grid}.
__global__ __launch_bounds__(256) void wattn_param_flush_kernel(const int64_t* __restrict__ table, int n) {
  __shared__ float red[3][64];
  int e = 0;
  while (e + 1 < n && (int)table[(long)(e + 1) * 16 + 6] <= (int)blockIdx.y) ++e;
  const int64_t* t = table + (long)e * 16;
  const float* part = reinterpret_cast<const float*>(t[0]);
  float* dtable = reinterpret_cast<float*>(t[1]);
  float* dqkv_b = reinterpret_cast<float*>(t[2]);
  const int heads = (int)t[3], C = (int)t[4], rows = (int)t[5], head = (int)blockIdx.y - (int)t[6];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  float a = 0.f;
  if (col < WATTN_PROW) {
#pragma unroll 8
    for (int r = w; r < rows; r += 4) a += part[((long)r * heads + head) * WATTN_PROW + col];
  }
  if (w > 0) red[w - 1][lane] = a;
  __syncthreads();
  if (w != 0 || col >= WATTN_PROW) return;
  a += red[0][lane];
  a += red[1][lane];
  a += red[2][lane];
  if (col < TBL) {
    if (dtable) dtable[col * heads + head] += a;
  } else if (col >= WATTN_PBIAS && dqkv_b) {
    const int c = col - WATTN_PBIAS;
    dqkv_b[(c / HD) * C + head * HD + (c % HD)] += a;
  }
}
}  // namespace rscotr

// table: device (n, 16) int64 rows {partial rows left by rscotr_swin_wattn_bwd(dqkv_bias = dbias_table = NULL), dbias_table | 0,
// dqkv_bias | 0, heads, C, rows per head (= workspace bytes / (heads * WATTN_PROW * 4)), first grid row of the entry (running
// sum of heads), 0 ...}; total_heads = sum of heads.  The destinations are ADDED to.
extern "C" int rscotr_swin_wattn_flush(const int64_t* table, int n, int total_heads, void* stream) {
  if (n < 0 || total_heads < 0) return fail(RSCOTR_E_SHAPE, "rscotr_swin_wattn_flush: negative count");
  if (n == 0 || total_heads == 0) return RSCOTR_OK;
  if (!table) return fail(RSCOTR_E_ARG, "rscotr_swin_wattn_flush: null table");
  wattn_param_flush_kernel<<<dim3((WATTN_PROW + 63) / 64, (unsigned)total_heads), 256, 0, (hipStream_t)stream>>>(table, n);
  return check_launch("rscotr_swin_wattn_flush");
}

extern "C" int64_t rscotr_swin_wattn_bwd_workspace(int B, int H, int W, int C, int heads) {
  WinGeom g;
  if (B <= 0 || wattn_geom("rscotr_swin_wattn_bwd_workspace", &g, B, H, W, C, heads, WS, 0)) return 0;
  return (int64_t)wattn_bwd_grid(g, B) * WATTN_PROW * 4;
}

extern "C" int rscotr_swin_wattn_bwd(const float* qkv, const float* qkv_bias, const float* bias_table,
                                     const float* dout, float* dqkv, float* dqkv_bias, float* dbias_table,
                                     int B, int H, int W, int C, int heads, int ws, int shift, const float* out,
                                     float* workspace, int64_t workspace_bytes, uint32_t* amax_out, void* stream) {
  WinGeom g;
  if (int e = wattn_geom("rscotr_swin_wattn_bwd", &g, B, H, W, C, heads, ws, shift)) return e;
  if (B == 0) return RSCOTR_OK;
  if (!qkv || !bias_table || !dout || !dqkv) return fail(RSCOTR_E_ARG, "rscotr_swin_wattn_bwd: null pointer");
  if (!aligned16(qkv) || !aligned16(dout) || !aligned16(dqkv) || (qkv_bias && !aligned16(qkv_bias)))
    return fail(RSCOTR_E_ALIGN, "rscotr_swin_wattn_bwd: pointers must be 16-byte aligned");
  const int nwg = wattn_bwd_grid(g, B);
  if (!workspace || workspace_bytes < (int64_t)nwg * WATTN_PROW * 4)
    return fail(RSCOTR_E_ARG, "rscotr_swin_wattn_bwd: workspace of rscotr_swin_wattn_bwd_workspace() bytes required");
  static const int phases = getenv("RSCOTR_WATTN_PHASES") ? atoi(getenv("RSCOTR_WATTN_PHASES")) : 4;
  static const int impl = getenv("RSCOTR_WATTN_BWD_IMPL") ? atoi(getenv("RSCOTR_WATTN_BWD_IMPL")) : 2;  // 0 VALU, 1 MFMA, 2 MFMA x 4 waves
  hipStream_t s = (hipStream_t)stream;
  // algorithmic work: S = q k^T (recomputed), dP = dO v^T, dV = P^T dO, dQ = dS k, dK = dS^T q: 10 * 49 * 49 * 32 flop per item
  const double items = (double)B * ((H + ws - 1) / ws) * ((W + ws - 1) / ws) * heads;
  ProfScope prof(PROF_MFMA, items * 10.0 * 49 * 49 * 32, s, "rscotr::swin_wattn_bwd_kernel");
  if (impl == 0)
    swin_wattn_bwd_kernel<<<nwg, 64, 0, s>>>(qkv, qkv_bias, bias_table, dout, dqkv, workspace, g, B, phases);
  else if (impl == 1)
    swin_wattn_bwd_mfma_kernel<<<nwg, 64, 0, s>>>(qkv, qkv_bias, bias_table, dout, dqkv, workspace, g, B);
  else
  {
    // measured (scripts/bench_wattn.py, B = 2 at 512^2): 2 resident workgroups per CU win except where the items fit the
    // chip at 3 per CU but not at 2 (stage 3: 600 items, 35.7 -> 28.9 us); 4 (with scratch) always loses
    static const int occ_env = getenv("RSCOTR_WATTN_OCC") ? atoi(getenv("RSCOTR_WATTN_OCC")) : 0;
    const int occ = occ_env ? occ_env : ((nwg > 512 && nwg <= 768) ? 3 : 2);
    static const int use_out = getenv("RSCOTR_WATTN_DELTA_OUT") ? atoi(getenv("RSCOTR_WATTN_DELTA_OUT")) : 1;
    const float* o = (use_out && out && aligned16(out)) ? out : nullptr;
    if (occ >= 4) swin_wattn_bwd_mfma4_kernel<4><<<nwg, 256, 0, s>>>(qkv, qkv_bias, bias_table, dout, o, dqkv, workspace, g, B, amax_out);
    else if (occ == 3) swin_wattn_bwd_mfma4_kernel<3><<<nwg, 256, 0, s>>>(qkv, qkv_bias, bias_table, dout, o, dqkv, workspace, g, B, amax_out);
    else swin_wattn_bwd_mfma4_kernel<2><<<nwg, 256, 0, s>>>(qkv, qkv_bias, bias_table, dout, o, dqkv, workspace, g, B, amax_out);
  }
  if (dbias_table || dqkv_bias)
    wattn_param_fold_kernel<<<dim3((WATTN_PROW + 63) / 64, heads), 256, 0, s>>>(workspace, dbias_table, dqkv_bias, heads, C,
                                                                              nwg / heads);
  if (int e = check_launch("rscotr_swin_wattn_bwd")) return e;
  return (amax_out && impl != 2) ? rscotr_amax_f32(dqkv, (int64_t)B * H * W, 3 * C, 3 * C, amax_out, stream) : RSCOTR_OK;
}
