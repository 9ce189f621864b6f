// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32 accumulate, exact
// fp32 = an fmaf chain; gfx950 has no TF32/xf32) with fused epilogues.
//
// This is the substrate under every dense contraction of the co-training step: what the
// reference reaches as nn.Linear / F.linear / 1x1 and patchify Conv2d through mmcv, mmdet, mmcls
// and torch (QKV/proj/MLP of mmdet SwinTransformer — configs/multi/MTL_slvlcls_...potsdam.py:9-25;
// FFN 256->2048->256 and the MSDeformAttn projections of the shared encoder — :34-50; the DINO and
// Mask2Former decoder/branch Linears — models/multi/bbox_head/dino_head.py:40-47,
// models/multi/seg_head/mask2former_head.py:60-83; ChannelMapper 1x1 convs — :26-33), together with
// the two backward contractions autograd derives from each of them.
//
//   C[m,n] = epilogue( sum_k Aop[m,k] * Bop[n,k] )
//   Aop[m,k] = a_kmajor ? A[k*lda + m] : A[m*lda + k]     (same for B with ldb, over n)
// so that   y  = x W^T + b      is (A=x,  B=W,  a_kmajor=0, b_kmajor=0)   [F.linear]
//           dx = dy W           is (A=dy, B=W,  a_kmajor=0, b_kmajor=1)
//           dW = dy^T x         is (A=dy, B=x,  a_kmajor=1, b_kmajor=1)
// epilogue(v): v += bias[n]; if (pre) pre[m,n] = v; v = act(v) or v *= act'(aux[m,n]);
//              v += resid[m,n]; if (accumulate) v += C[m,n].
//
// Kernel shape (wave64, not a warp-shaped CUDA tiling): 256 threads = 4 wavefronts, one per SIMD;
// block tile BM x BN x 16; each wavefront owns a (BM/WM) x (BN/WN) sub-tile as MT x NT
// accumulators of 32x32 (16 VGPRs each).  Operand tiles are staged global -> VGPR -> LDS in
// K-MAJOR order ([k][m], [k][n]) so that an MFMA operand fragment (lane l: row l&31, k = l>>5) is
// one conflict-free ds_read_b32 of 32 consecutive floats per half-wave; LDS is double-buffered
// with one barrier per k-tile, the next tile's global loads are issued before the MFMAs of the
// current one.  Small grids on long reductions (the dW contractions) are split along K into
// fp32 slabs in a caller-provided workspace and combined in fixed order by a second kernel that applies
// the epilogue (deterministic: no atomics).  (Finishing a tile in the same launch by its last-arriving
// workgroup was tried: the device-scope fences it needs write back / invalidate the whole per-XCD L2 on
// every workgroup and made the step 2.5x slower.)
// A k-major A operand can also deliver its row sums over k (the bias gradient of the dW contraction).
#include "gemm_common.h"
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <mutex>
#include <set>
#include <type_traits>

#ifndef RSCOTR_GEMM_PREC_DEFAULT
#define RSCOTR_GEMM_PREC_DEFAULT 3
#endif

namespace rscotr {

// Load the (R rows x 16 k) operand tile at (row0, k0) into registers: NV float4 per thread.
template <int R, bool KMAJOR>
struct TileLoader {
  static constexpr int NV = (R * 4 + 255) / 256;
  float4 v[NV];

  __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int rows, int kend, int row0,
                                       int k0, int vec, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * 256;
      float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < R * 4) {
        if (!KMAJOR) {
          const int row = row0 + (idx >> 2), k = k0 + (idx & 3) * 4;
          if (row < rows) {
            const float* src = P + (long)row * ld + k;
            if (vec && k + 3 < kend) {
              r = *reinterpret_cast<const float4*>(src);
            } else {
              if (k + 0 < kend) r.x = src[0];
              if (k + 1 < kend) r.y = src[1];
              if (k + 2 < kend) r.z = src[2];
              if (k + 3 < kend) r.w = src[3];
            }
          }
        } else {
          const int k = k0 + idx / (R / 4), row = row0 + (idx % (R / 4)) * 4;
          if (k < kend) {
            const float* src = P + (long)k * ld + row;
            if (vec && row + 3 < rows) {
              r = *reinterpret_cast<const float4*>(src);
            } else {
              if (row + 0 < rows) r.x = src[0];
              if (row + 1 < rows) r.y = src[1];
              if (row + 2 < rows) r.z = src[2];
              if (row + 3 < rows) r.w = src[3];
            }
          }
        }
      }
      v[i] = r;
    }
  }

  // whole tile in bounds, 16-byte loads legal
  __device__ __forceinline__ void load_fast(const float* __restrict__ P, int ld, int row0, int k0, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * 256;
      if (R * 4 % 256 == 0 || idx < R * 4) {
        if (!KMAJOR)
          v[i] = *reinterpret_cast<const float4*>(P + (long)(row0 + (idx >> 2)) * ld + k0 + (idx & 3) * 4);
        else
          v[i] = *reinterpret_cast<const float4*>(P + (long)(k0 + idx / (R / 4)) * ld + row0 + (idx % (R / 4)) * 4);
      }
    }
  }

  // k-major tile: row k of the staged tile times ks[k / per]
  __device__ __forceinline__ void scale_k(const float* __restrict__ ks, int per, int k0, int kend, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * 256;
      const int k = k0 + idx / (R / 4);
      if ((R * 4 % 256 == 0 || idx < R * 4) && k < kend) {
        const float f = ks[k / per];
        v[i].x *= f; v[i].y *= f; v[i].z *= f; v[i].w *= f;
      }
    }
  }

  // running sums over k of the columns this thread stages (k-major tiles: the thread's columns are fixed)
  __device__ __forceinline__ void accum(float4& a) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      a.x += v[i].x; a.y += v[i].y; a.z += v[i].z; a.w += v[i].w;
    }
  }

  // LDS image is always k-major: S[k][LD] with LD = R + 4.
  __device__ __forceinline__ void store(float* S, int tid) const {
    constexpr int LD = R + 4;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * 256;
      if (idx < R * 4) {
        if (!KMAJOR) {
          const int row = idx >> 2, kq = (idx & 3) * 4;
          S[(kq + 0) * LD + row] = v[i].x;
          S[(kq + 1) * LD + row] = v[i].y;
          S[(kq + 2) * LD + row] = v[i].z;
          S[(kq + 3) * LD + row] = v[i].w;
        } else {
          const int k = idx / (R / 4), c = (idx % (R / 4)) * 4;
          *reinterpret_cast<float4*>(S + k * LD + c) = v[i];
        }
      }
    }
  }

};


// EDGE = false: the host guarantees M % BM == 0, N % BN == 0, every k range a whole number of k-tiles and
// 16-byte vector loads legal on both operands — no bounds logic is compiled in (10-30 % faster on the
// step's forward shapes than the general kernel, which keeps both load paths and per-row store guards).
//
// KG > 1: KG groups of 4 wavefronts share one output tile and take the k-tiles round-robin (group g: k-tiles g,
// g+KG, ...), each with its own LDS double buffer; the partial accumulators meet in LDS at the end (fixed order).
// For launches with fewer workgroups than CUs the single-group loop runs at ~0.4 us per k-tile (LDS refill,
// barrier and fragment latency sit on the critical path with nothing to hide them): KG groups on the same CU
// interleave their chains.  No extra launch, no slabs in HBM.
//
// SLAB: always leave the result as split-K slabs / row-sum partials, also for a single k-slice (the grouped launch of
// deferred weight gradients: several problems may share a destination, the combine launch orders them).
template <int BM, int BN, int WM, int WN, bool AK, bool BK_, bool EDGE, int KG, bool SLAB>
__device__ __forceinline__ void gemm_f32_body(GemmParams& p, const int bx, const int gx, const int by) {
  static_assert(WM * WN == 4, "4 wavefronts per group");
  if (p.nb1 > 0) {  // batched: (b0, b1) = e.g. (image, head) of an attention product
    const int b01 = by / p.nb2, b2 = by - b01 * p.nb2;
    const int b0 = b01 / p.nb1, b1 = b01 - b0 * p.nb1;
    p.A += b0 * p.sA0 + b1 * p.sA1 + b2 * p.sA2;
    p.B += b0 * p.sB0 + b1 * p.sB1 + b2 * p.sB2;
    p.C += b0 * p.sC0 + b1 * p.sC1 + b2 * p.sC2;
  }
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MT = TM / 32, NT = TN / 32;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  extern __shared__ __attribute__((aligned(16))) float gemm_smem[];
  const int grp = KG > 1 ? (int)(threadIdx.x >> 8) : 0;
  // (offsets, not a pointer array: a runtime-indexed array of pointers loses the LDS address space and the
  // accesses degrade to flat loads)
  // floats of LDS per k-group: two operand-tile pairs (k-major [16][LD])
  constexpr int GROUP_FLOATS = 2 * GEMM_BK * (LDA + LDB);
  constexpr int SA_FLOATS = GEMM_BK * LDA, SB_FLOATS = GEMM_BK * LDB;
  const int lds0 = grp * GROUP_FLOATS;
  float* const sA0 = gemm_smem + lds0;
  float* const sA1 = sA0 + SA_FLOATS;
  float* const sB0 = sA1 + SA_FLOATS;
  float* const sB1 = sB0 + SB_FLOATS;

  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (p.N + BN - 1) / BN;
  int tile, split = 0;
  if (p.splits == 1) {
    tile = xcd_swizzle(bx, gx);
  } else {
    // split-K: every XCD (workgroup id % 8) owns a contiguous run of tiles with ALL their splits, so the
    // slabs of a tile are written and summed through one L2; inside the run the order is split-major
    // (neighbouring workgroups = neighbouring tiles on the same k-slice share operand panels).
    const int x = bx & 7, j = bx >> 3;
    const int q = p.tiles >> 3, r = p.tiles & 7, run = q + (r ? 1 : 0);
    const int nt = q + (x < r ? 1 : 0);
    split = j / run;
    const int tl = j - split * run;
    if (tl >= nt) return;
    tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + tl;
  }
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int kbeg = split * p.ksplit_len;
  const int kend = min(p.K, kbeg + p.ksplit_len);
  const int nk = (kend - kbeg + GEMM_BK - 1) / GEMM_BK;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  TileLoader<BM, AK> la;
  TileLoader<BN, BK_> lb;
  // interior tiles (the common case) skip the per-element bounds logic
  const bool fastA = !EDGE || (p.vecA && (m0 + BM <= p.M)), fastB = !EDGE || (p.vecB && (n0 + BN <= p.N));
  auto load_tiles = [&](int k0) {
    if (!EDGE) {
      la.load_fast(p.A, p.lda, m0, k0, tid);
      lb.load_fast(p.B, p.ldb, n0, k0, tid);
    } else {
      const bool kfull = k0 + GEMM_BK <= kend;
      if (fastA && kfull) la.load_fast(p.A, p.lda, m0, k0, tid);
      else la.load(p.A, p.lda, p.M, kend, m0, k0, p.vecA, tid);
      if (fastB && kfull) lb.load_fast(p.B, p.ldb, n0, k0, tid);
      else lb.load(p.B, p.ldb, p.N, kend, n0, k0, p.vecB, tid);
    }
    if (AK && p.kscale) la.scale_k(p.kscale, p.krows_per, k0, kend, tid);
  };
  // bias gradient riding the dW contraction: workgroups of tile column 0 also sum their A tile over k
  const bool do_rs = AK && p.rowsum && n0 == 0;
  float4 rs = make_float4(0.f, 0.f, 0.f, 0.f);
  if (grp < nk) {
    load_tiles(kbeg + grp * GEMM_BK);
    if (AK && do_rs) la.accum(rs);
    la.store(sA0, tid);
    lb.store(sB0, tid);
  }
  __syncthreads();

  const int fr = lane & 31, fk = lane >> 5;
  const int nit = (nk + KG - 1) / KG;
  for (int it = 0; it < nit; ++it) {
    const int kt = it * KG + grp;
    const int cur = it & 1;
    const bool more = kt + KG < nk;
    if (more) load_tiles(kbeg + (kt + KG) * GEMM_BK);
    if (KG > 1 && kt >= nk) {  // this group has run out of k-tiles (wave-uniform)
      __syncthreads();
      continue;
    }
    {
      const float* a = (cur ? sA1 : sA0) + fk * LDA + wm * TM + fr;
      const float* b = (cur ? sB1 : sB0) + fk * LDB + wn * TN + fr;
      // operand fragments double-buffered in registers: the ds_reads of step kk+2 are in flight
      // while the MFMAs of step kk execute
      float af[2][MT], bf[2][NT];
  #pragma unroll
      for (int i = 0; i < MT; ++i) af[0][i] = a[i * 32];
  #pragma unroll
      for (int j = 0; j < NT; ++j) bf[0][j] = b[j * 32];
  #pragma unroll
      for (int kk = 0; kk < GEMM_BK; kk += 2) {
        const int c = (kk >> 1) & 1;
        if (kk + 2 < GEMM_BK) {
  #pragma unroll
          for (int i = 0; i < MT; ++i) af[c ^ 1][i] = a[(kk + 2) * LDA + i * 32];
  #pragma unroll
          for (int j = 0; j < NT; ++j) bf[c ^ 1][j] = b[(kk + 2) * LDB + j * 32];
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch reads ahead of this step's MFMAs
  #pragma unroll
        for (int i = 0; i < MT; ++i)
  #pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][i], bf[c][j], acc[i][j], 0, 0, 0);
      }
    }
    if (more) {
      if (AK && do_rs) la.accum(rs);
      la.store(cur ? sA0 : sA1, tid);
      lb.store(cur ? sB0 : sB1, tid);
    }
    __syncthreads();
  }

  if (AK && do_rs) {
    // thread t staged columns (t % (BM/4))*4.. of every k row it touched: fold the 256/(BM/4) k-lanes
    static_assert(!AK || 256 % (BM / 4) == 0, "row-sum fold needs fixed columns per thread");
    constexpr int CG = BM / 4, KL = 256 / CG;
    float4* red = reinterpret_cast<float4*>(sA0);  // KL x CG float4 = 4 KB <= one sA buffer (one per k-group)
    red[(tid / CG) * CG + (tid % CG)] = rs;
    __syncthreads();
    if (grp == 0 && tid < BM) {
      float v = 0.f;
#pragma unroll
      for (int g2 = 0; g2 < KG; ++g2) {
        const float* rf = gemm_smem + g2 * GROUP_FLOATS;
#pragma unroll
        for (int k = 0; k < KL; ++k) v += rf[k * BM + tid];
      }
      const int m = m0 + tid;
      if (m < p.M) {
        if (p.splits > 1 || SLAB) p.rs_slabs[(long)split * p.M + m] = v;
        else p.rowsum[m] = p.rowsum_acc ? p.rowsum[m] + v : v;
      }
    }
  }

  if (KG > 1) {
    if (AK && do_rs) __syncthreads();  // the row-sum fold above has finished reading the groups' LDS (uniform)
    // groups 1..KG-1 hand their accumulators to group 0 through LDS ([group][register][thread]: conflict-free)
    static_assert(KG == 1 || (MT == 1 && NT == 1), "in-workgroup k-groups are built for one 32x32 tile per wavefront");
    float* red = gemm_smem;
    if (grp > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((grp - 1) * 16 + r) * 256 + tid] = acc[0][0][r];
    }
    __syncthreads();
    if (grp > 0) return;
#pragma unroll
    for (int g2 = 0; g2 < KG - 1; ++g2)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][0][r] += red[(g2 * 16 + r) * 256 + tid];
  }

  // C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
  if (p.splits > 1 || SLAB) {
    float* slab = p.slabs + (long)split * p.M * p.N;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = n0 + wn * TN + j * 32 + fr;
        if (EDGE && n >= p.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * TM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
          if (!EDGE || m < p.M) slab[(long)m * p.N + n] = acc[i][j][r];
        }
      }
    return;
  }
  const bool plain = !p.pre && p.act == ACT_NONE && !p.resid && !p.accumulate && !p.rowscale && !p.C2;
  // exactly one extra tensor read by the epilogue: its 16 values per 32 x 32 tile in one batch (epilogue_tile16; one-tile-per-
  // wavefront configurations only: the 128-row tiles keep their registers)
  const bool one_extra = MT == 1 && NT == 1 && !p.C2 &&
                         ((p.act == ACT_RELU_GRAD || p.act == ACT_GELU_GRAD) ? 1 : 0) + (p.resid ? 1 : 0) + (p.accumulate ? 1 : 0) == 1;
  // an activation (and / or the stored pre-activation) but no tensor to read: epilogue_noload16
#ifdef RSCOTR_NO_NOLOAD  // (A/B builds: scripts/build_variant.sh)
  const bool noload = false;
#else
  const bool noload = !plain && (p.act == ACT_NONE || p.act == ACT_RELU || p.act == ACT_GELU) && !p.resid && !p.accumulate &&
                      !p.rowscale && !p.C2;
#endif
  float amx = 0.f;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + wn * TN + j * 32 + fr;
      if (EDGE && n >= p.N) continue;
      const float bv = p.bias ? p.bias[n] : 0.f;
      const int mb = m0 + wm * TM + i * 32 + 4 * fk;
      float* crow = p.C + (long)mb * p.ldc + n;
      if (one_extra) {
        epilogue_tile16<EDGE>(p, acc[i][j], bv, mb, n, amx);
      } else if (noload) {
        epilogue_noload16<EDGE>(p, acc[i][j], bv, mb, n, amx);
      } else if (plain) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int dm = (r & 3) + 8 * (r >> 2);
          if (!EDGE || mb + dm < p.M) {
            const float v = acc[i][j][r] + bv;
            crow[(long)dm * p.ldc] = v;
            amx = fmaxf(amx, fabsf(v));
          }
        }
      } else {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {  // rows 8 * g4 + {0..3} of this lane's 16 (C/D layout of the 32x32 MFMA)
          float v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = acc[i][j][4 * g4 + u] + bv;
          epilogue_rows4<EDGE>(p, v, mb + 8 * g4, n, amx);
          __builtin_amdgcn_sched_barrier(0);  // keep the next group's loads from being hoisted across (registers)
        }
      }
    }
  amax_commit(p.amax_out, amx);
}

// (holding the k-group instantiations — 512 / 1024 threads, 68-72 registers — to 64 registers so that two 1024-thread workgroups
// fit a CU measured nothing: 33.80 against 33.79 ms per round)
template <int BM, int BN, int WM, int WN, bool AK, bool BK_, bool EDGE, int KG>
__global__ __launch_bounds__(256 * KG) void gemm_f32_kernel(GemmParams p) {
  gemm_f32_body<BM, BN, WM, WN, AK, BK_, EDGE, KG, false>(p, blockIdx.x, gridDim.x, blockIdx.y);
}

// ---------------------------------------------------------------------------------------------------------------
// bf16x6: the fp32-ACCURATE split product (precision mode 3; scripts/lab/bf16x6_lab.hip is the stand-alone version).
// x = h + m + l with h = rne_bf16(x), m = rne_bf16(x - h), l = rne_bf16(x - h - m) — both subtractions exact in fp32, so the
// three bf16 planes carry all 24 significand bits and bf16 keeps the fp32 exponent (no range problem).  Per k-step of 16
// the product keeps the six plane pairs of order <= 2^-16 — l*h + h*l + m*m + m*h + h*m + h*h, small terms first, fp32
// accumulate (v_mfma_f32_32x32x16_bf16); what is dropped (m*l + l*m + l*l) is <= 2^-23 |a||b| per product, the rounding
// class of an fp32 FMA.  Measured against fp64 (lab, MI355X): 3.0e-7 of max|C| on M = 10880, N = 2048, K = 256 where the
// fp32 FMA chain has 4.4e-7 and the two-plane bf16x3 product 4.4e-6.  Six MFMAs of 32 cycles per 32x32x16 block against
// eight of 64 on the fp32 pipe: 2500 / 6 = 417 TFLOP/s-equivalent peak against 157.3.
// Structure: BK = 16 per stage; operands staged global -> VGPR -> (split, pack) -> LDS with the three planes of a row
// side by side (row-major source: 112-byte rows, one conflict-free 16-byte read per fragment and plane) or as k-pair
// dwords (k-major source: written as 16-byte rows, four dword reads per fragment); 128 x 128 tiles on one LDS stage with
// two barriers per k-tile (1-3 resident workgroups cover each other), 64 x 64 tiles double-buffered with one barrier.
// Interior shapes only (host-checked); split-K slabs, deferred combine, bias-gradient row sums, per-sample k scaling and
// the staged epilogue are shared with the kernels above.
// The planes of TWO adjacent values as packed dwords (low half = a, high half = b): the same conversions and exact
// subtractions as split_planes, written on 2-vectors so that they compile to v_cvt_pk_bf16_f32 (two conversions and the
// pack in one instruction), one mask + one shift for the way back and v_pk_add_f32 for the two subtractions: 9 VALU
// instructions per pair against ~19 for two scalar splits + two packs (a VALU instruction occupies its SIMD's issue port
// for 4 cycles: the conversion was 41 % of the 4096^3 kernel's SIMD time next to 43 % of MFMA, profiles/r3_bf16x6_pmc.txt).
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
template <int NPL>
__device__ __forceinline__ void split_pair(float a, float b, unsigned (&out)[3]) {
  const f32x2_t x = {a, b};
  const bf16x2_t h = __builtin_convertvector(x, bf16x2_t);
  out[0] = __builtin_bit_cast(unsigned, h);
  const f32x2_t r1 = x - __builtin_convertvector(h, f32x2_t);
  const bf16x2_t m = __builtin_convertvector(r1, bf16x2_t);
  out[1] = __builtin_bit_cast(unsigned, m);
  if (NPL == 3) {
    const f32x2_t r2 = r1 - __builtin_convertvector(m, f32x2_t);
    out[2] = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2_t));
  } else {
    out[2] = 0u;
  }
}

__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {
  const f32x2_t x = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2_t));
}

// Planes of the pairs (e.x, o.x), (e.y, o.y), (e.z, o.z), (e.w, o.w) (low half = e): out[plane] = the four packed dwords.
template <int NPL>
__device__ __forceinline__ void split_rows4(const float4& e, const float4& o, uint4 (&out)[3]) {
  f32x2_t e01 = {e.x, e.y}, e23 = {e.z, e.w}, o01 = {o.x, o.y}, o23 = {o.z, o.w};
#pragma unroll
  for (int pl = 0; pl < NPL; ++pl) {
    const unsigned h0 = cvt_pk_bf16(e01.x, o01.x), h1 = cvt_pk_bf16(e01.y, o01.y);
    const unsigned h2 = cvt_pk_bf16(e23.x, o23.x), h3 = cvt_pk_bf16(e23.y, o23.y);
    out[pl] = make_uint4(h0, h1, h2, h3);
    if (pl + 1 < NPL) {
      const f32x2_t fe01 = {__uint_as_float(h0 << 16), __uint_as_float(h1 << 16)};
      const f32x2_t fe23 = {__uint_as_float(h2 << 16), __uint_as_float(h3 << 16)};
      const f32x2_t fo01 = {__uint_as_float(h0 & 0xffff0000u), __uint_as_float(h1 & 0xffff0000u)};
      const f32x2_t fo23 = {__uint_as_float(h2 & 0xffff0000u), __uint_as_float(h3 & 0xffff0000u)};
      e01 -= fe01; e23 -= fe23; o01 -= fo01; o23 -= fo23;
    }
  }
}

// ---- fp16 split product ("h3", round 5): x 2^s = h + l 2^-11 with h = rne_f16(x 2^s), l = rne_f16((x 2^s - h) 2^11).  The
// subtraction is exact in fp32 and |x 2^s - h| <= 2^-12 |x 2^s|, so l keeps 11 of the remaining 13 bits: the two planes
// carry x to 2^-24 relative (the rounding class of fp32 itself) wherever fp16 is normal, i.e. down to 2^-26 of the tensor's
// amax with the scale of h3_scale_exp; below that the error is 2^-48 of amax absolute.  The product keeps three terms,
// h h into one accumulator and l h + h l into a second one that enters with 2^-11 at the end (fp32 accumulate; the dropped
// l l term is <= 2^-24 |a||b|): THREE v_mfma_f32_32x32x16_f16 per 16 k instead of six bf16 ones, two planes instead of three
// through the conversion and LDS.  7 VALU instructions per value pair (pk_mul, cvt_pk, 2 cvt, pk_mul, pk_fma, cvt_pk).
// (Measured and not kept, profiles/r5_h3_one_acc.txt: l UNSCALED and all three MFMAs into ONE accumulator set — 16 / 64 accumulator
// registers and one VALU instruction per value pair less, the 64 x 64 kernel at four workgroups per CU: 34.55 -> 33.86 ms per
// round with every product on it, 34.0 -> 33.7 with the forward-layout products only.  Its error is that of an fp32 FMA chain
// (4.3e-7 of max|C| against 2.5e-7 here), and elements more than 2^16 below the tensor's amax keep 11 bits only (fp16's 5-bit
// exponent; 2^27 with the scaled l).  The det step at 256^2, seed 4, then takes a ReLU gate of a decoder FFN on the other side
// — a coin toss for any fp32-class product, but outside the band tests/parity.py flips (3e-6 of the mean |pre-activation|)
// — and leaves the 1e-3 tier by 4e-3 on decoder layer 5.  Parity first: the two accumulator sets stay.)
// (f16x2_t / f16x8 / H3Scale / split_pair_h: gemm_common.h — shared with csrc/ffn.hip)
__device__ __forceinline__ void split_rows4_h(const float4& e, const float4& o, const H3Scale& k, uint4 (&out)[3]) {
  unsigned a[3], b[3], c[3], d[3];
  split_pair_h(e.x, o.x, k, a);
  split_pair_h(e.y, o.y, k, b);
  split_pair_h(e.z, o.z, k, c);
  split_pair_h(e.w, o.w, k, d);
  out[0] = make_uint4(a[0], b[0], c[0], d[0]);
  out[1] = make_uint4(a[1], b[1], c[1], d[1]);
}

template <int R, bool KM, int NPL, int SBK = 16, bool H16 = false>
struct SplitOperand {
  static_assert(!H16 || NPL == 2, "the fp16 split has two planes");
  static constexpr int LDR = SBK * NPL + 8;                         // bf16 per LDS row (row-major source): 112 / 208 bytes
  static constexpr int KP = SBK / 2;                                // k pairs per stage
  static constexpr int Q = SBK / 4;                                 // float4 per row per stage (row-major source)
  static constexpr int WORDS = KM ? NPL * KP * R : R * LDR / 2;     // dwords per stage
  static constexpr int ITEMS = KM ? KP * R / 4 : R * Q;             // float4 (pairs) per tile
  static constexpr int NV = (ITEMS + 255) / 256;
  float4 v[NV], w[NV];  // row-major: v; k-major: v = even k row, w = odd k row of a pair

  // EDGE instantiations (ragged M / N / K).  rlast: the last row a load may touch — rows - 1 of a row-major operand, rows - 4
  // of a k-major one (whose rows are read four at a time; rows % 4 == 0, host-checked): rows past it are CLAMPED reads, and
  // what they bring is multiplied into accumulator rows / columns that are never stored.  klim: the end of the reduction
  // (K % 4 == 0, host-checked): k positions past it are clamped reads replaced by ZEROS (they do enter the sums).
  template <bool EDGE = false>
  __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int row0, int k0, int tid, int rlast = 0, int klim = 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * 256;
      if (ITEMS % 256 == 0 || idx < ITEMS) {
        if (!KM) {
          const int row = EDGE ? min(row0 + idx / Q, rlast) : row0 + idx / Q;
          const int kk = k0 + (idx % Q) * 4;
          v[i] = *reinterpret_cast<const float4*>(P + (long)row * ld + (EDGE ? min(kk, klim - 4) : kk));
          if (EDGE && kk >= klim) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
          const int kp = idx / (R / 4), r4 = (idx % (R / 4)) * 4;
          const int ka = k0 + 2 * kp, col = EDGE ? min(row0 + r4, rlast) : row0 + r4;
          if (!EDGE) {
            const float* src = P + (long)ka * ld + col;
            v[i] = *reinterpret_cast<const float4*>(src);
            w[i] = *reinterpret_cast<const float4*>(src + ld);
          } else {
            v[i] = *reinterpret_cast<const float4*>(P + (long)min(ka, klim - 1) * ld + col);
            w[i] = *reinterpret_cast<const float4*>(P + (long)min(ka + 1, klim - 1) * ld + col);
            if (ka >= klim) v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ka + 1 >= klim) w[i] = make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
      }
    }
  }
  // k-major only: row k of the staged tile times ks[k / per]
  __device__ __forceinline__ void scale_k(const float* __restrict__ ks, int per, int k0, int tid, int klast = 0x7ffffffe) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * 256;
      if (ITEMS % 256 == 0 || idx < ITEMS) {
        const int k = k0 + 2 * (idx / (R / 4));
        const float f0 = ks[min(k, klast) / per], f1 = ks[min(k + 1, klast) / per];  // (rows past K hold zeros: any factor)
        v[i].x *= f0; v[i].y *= f0; v[i].z *= f0; v[i].w *= f0;
        w[i].x *= f1; w[i].y *= f1; w[i].z *= f1; w[i].w *= f1;
      }
    }
  }
  // k-major only: running sums over k of the four rows this thread stages (r4 is the same for all its items: 256 is a
  // multiple of R / 4), times `f` (0 for a tile staged a second time at the end of the pipelined loop)
  __device__ __forceinline__ void accum(float4& a, int tid, float f = 1.f) const {
    static_assert(!KM || 256 % (R / 4) == 0, "row sums assume one row group per thread");
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (ITEMS % 256 == 0 || tid + i * 256 < ITEMS) {
        a.x = fmaf(f, v[i].x + w[i].x, a.x); a.y = fmaf(f, v[i].y + w[i].y, a.y);
        a.z = fmaf(f, v[i].z + w[i].z, a.z); a.w = fmaf(f, v[i].w + w[i].w, a.w);
      }
  }
  __device__ __forceinline__ void store(unsigned* S, int tid, const H3Scale& hs = H3Scale{1.f, 2048.f}) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * 256;
      if (ITEMS % 256 == 0 || idx < ITEMS) {
        if (!KM) {
          const int row = idx / Q, kq = (idx % Q) * 4;
          unsigned ab[3], cd[3];
          if constexpr (H16) {
            split_pair_h(v[i].x, v[i].y, hs, ab);
            split_pair_h(v[i].z, v[i].w, hs, cd);
          } else {
            split_pair<NPL>(v[i].x, v[i].y, ab);
            split_pair<NPL>(v[i].z, v[i].w, cd);
          }
          unsigned* dst = S + (row * LDR + kq) / 2;
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) {
            uint2 q;
            q.x = ab[pl];
            q.y = cd[pl];
            *reinterpret_cast<uint2*>(dst + pl * (SBK / 2)) = q;
          }
        } else {
          const int kp = idx / (R / 4), r4 = (idx % (R / 4)) * 4;
          // (even k, odd k) pairs of the four rows: the conversions take one value of each row vector (v_cvt_pk_bf16_f32 has
          // two independent sources), the exact subtractions run on the rows' own register pairs (v_pk_add_f32)
          uint4 q[3];
          if constexpr (H16) split_rows4_h(v[i], w[i], hs, q);
          else split_rows4<NPL>(v[i], w[i], q);
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<uint4*>(S + (pl * KP + kp) * R + r4) = q[pl];
        }
      }
    }
  }
  // fragment of k-substep ks (16 k) of the stage
  static __device__ __forceinline__ void frag(const unsigned* S, int row, int g, int ks, bf16x8 (&f)[3]) {
    if (!KM) {
      const unsigned* q = S + (row * LDR + 16 * ks + 8 * g) / 2;
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) f[pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(q + pl * (SBK / 2)));
    } else {
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        const unsigned* q = S + (pl * KP + 8 * ks + 4 * g) * R + row;
        uint4 t;
        t.x = q[0]; t.y = q[R]; t.z = q[2 * R]; t.w = q[3 * R];
        f[pl] = __builtin_bit_cast(bf16x8, t);
      }
    }
  }
};

// B operand of the fp16 split product from PRE-SPLIT planes (round 5, rscotr_gemm_split_weights_h3): in y = x W^T and dx = dy W the
// B tile of a workgroup is a weight, which changes once per optimizer step, yet every one of the M / 64 row tiles of every launch
// converts it again — and the k loop of the 64 x 64 kernel is bound by exactly that conversion issue (78 VALU instructions per 6
// MFMAs and k-step, half of them B's: profiles/r5_h3_64_pmc.txt).  Plane layout [K / 32][rows padded to 64][h | l][32 k] fp16 (the
// planes of W for y = x W^T, of W^T for dx = dy W, so the kernel never sees a k-major B): the 64-row stage of a workgroup is ONE
// contiguous 8 KB run, 32 bytes per thread, written to the LDS rows of SplitOperand<R, false, 2, 32, true> as they are — no VALU
// work.  The planes carry the scale of the weight's range word at the time of the split; the consumer takes its 2^-s from the
// same word (the word only changes in the optimizer step, after which the planes are re-split).  (The same operand on the
// 128 x 128 one-stage kernels — four pieces per thread — measured nothing, 33.68 against 33.64 ms per round: those launches wait on
// memory, not on conversion issue.)
template <int R, int SBK>
struct PlaneOperandH {
  using Lay = SplitOperand<R, false, 2, SBK, true>;
  static_assert(R == 64 && SBK == 32, "one 32-byte piece per thread");
  static constexpr int WORDS = Lay::WORDS, KP = Lay::KP;
  typedef float vec4 __attribute__((ext_vector_type(4)));  // (a native vector: whole-struct copies of HIP's float4 / uint4 between
  vec4 v0, v1;                                             //  address spaces stay memcpys, and the register sets stayed in scratch)
  // P: the plane set (as const float* for the shared body), ld: its padded row count
  template <bool EDGE = false>
  __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int row0, int k0, int tid, int = 0, int = 0) {
    const vec4* src = reinterpret_cast<const vec4*>(P) + ((long)(k0 / SBK) * ld + row0) * 8 + tid * 2;
    v0 = src[0];
    v1 = src[1];
  }
  __device__ __forceinline__ void store(unsigned* S, int tid, const H3Scale& = H3Scale{1.f, 2048.f}) const {
    vec4* dst = reinterpret_cast<vec4*>(S + (tid >> 2) * (Lay::LDR / 2) + (tid & 3) * 8);
    dst[0] = v0;
    dst[1] = v1;
  }
  static __device__ __forceinline__ void frag(const unsigned* S, int row, int g, int ks, bf16x8 (&f)[3]) { Lay::frag(S, row, g, ks, f); }
};

// PIPE: 0 = one LDS stage, two barriers per k-tile of 16; 1 = two LDS stages, one barrier, next tile's loads one step ahead;
// 2 / 3 = the software-pipelined loop (two LDS stages, one barrier): the loads of tile t + D are issued at the top of step t
// into the register set step t - 1 freed (D = 2 / 3 sets), and the split / pack / LDS writes of tile t + 1 are interleaved
// with the MFMAs of tile t inside the wavefront (sched_group_barrier: 1 MFMA : 4 VALU : 1 DS write) — the conversion runs in
// the shadow of the matrix pipe instead of in a phase of its own.  PIPE 2 stages 32 k per step (64 x 64 tiles: a row-major
// operand row is one whole 128-byte line per step; half as many barriers), PIPE 3 stages 16 (128 x 128 tiles: LDS).
__device__ const float bf16x6_one = 1.f;
#ifndef RSCOTR_X6_BK0
#define RSCOTR_X6_BK0 32  // k per barrier pair of the one-stage loop (PIPE 0: the 128 x 128 kernels, the grouped launch's bodies).  Round 4 measured 32 at +0.2 ms per round on the six-term bf16 product (MFMA + conversion issue bound); on the fp16 product, which waits on memory for half of its wave life (profiles/r5_h3_64_pmc.txt), 32 is -0.6 ms: 33.73 against 34.32
#endif
template <int PIPE> constexpr int bf16x6_bk() { return PIPE == 2 ? 32 : PIPE == 0 ? RSCOTR_X6_BK0 : 16; }
#ifndef RSCOTR_X6_D2
#define RSCOTR_X6_D2 2
#endif
#ifndef RSCOTR_H3_VPM
#define RSCOTR_H3_VPM 8
#endif
#ifndef RSCOTR_H3_DPM
#define RSCOTR_H3_DPM 2
#endif
template <int PIPE> constexpr int bf16x6_depth() { return PIPE == 2 ? RSCOTR_X6_D2 : PIPE == 3 ? 3 : 1; }

template <int BM, int BN, bool AKM, bool BKM, int PIPE, bool H16 = false>
constexpr int bf16x6_lds_words() {
  constexpr int NPL = H16 ? 2 : 3;
  return (PIPE ? 2 : 1) * (SplitOperand<BM, AKM, NPL, bf16x6_bk<PIPE>(), H16>::WORDS + SplitOperand<BN, BKM, NPL, bf16x6_bk<PIPE>(), H16>::WORDS);
}

// SLAB: leave the result as split-K slabs / row-sum partials also for a single k-slice (grouped launch, see below).
// lds: bf16x6_lds_words() dwords, 16-byte aligned.
// H16: the fp16 split product (split_pair_h above): operands scaled by powers of two from p.amax_a / p.amax_b (both
// required), three MFMAs per 16 k into two accumulator sets.
template <int BM, int BN, bool AKM, bool BKM, int PIPE, bool SLAB, bool EDGE = false, bool H16 = false, bool BPL = false>
__device__ __forceinline__ void gemm_bf16x6_body(GemmParams& p, const int bx, const int gx, unsigned* lds) {
  constexpr int NPL = H16 ? 2 : 3, SBK = bf16x6_bk<PIPE>(), D = bf16x6_depth<PIPE>();
  constexpr int MT = BM / 64, NT = BN / 64;
  using OA = SplitOperand<BM, AKM, NPL, SBK, H16>;
  static_assert(!BPL || (H16 && !BKM && !EDGE && PIPE == 2), "B from planes: the interior pipelined fp16 kernel");
  using OB = typename std::conditional<BPL, PlaneOperandH<BN, SBK>, SplitOperand<BN, BKM, NPL, SBK, H16>>::type;
  H3Scale ha{1.f, 2048.f}, hb{1.f, 2048.f};
  float inva = 1.f, invb = 1.f;
  // The range words are REQUESTED here and reduced (h3_scales) only after the first operand tiles have been requested too:
  // the words are cold lines for this XCD's L2 — waiting for them first would put a full memory latency in front of every
  // workgroup's first tile (measured in the step: the split product's gain over the six-term one was gone).
  unsigned ra = 0u, rb = 0u;
  if constexpr (H16) {
    ra = p.amax_a[(long)(threadIdx.x & (kAmaxPlanes - 1)) * kAmaxStride];
    rb = p.amax_b[(long)(threadIdx.x & (kAmaxPlanes - 1)) * kAmaxStride];
  }
  auto h3_scales = [&]() {
    if constexpr (H16) {
      const int ea = h3_scale_exp(amax_fold(ra)), eb = h3_scale_exp(amax_fold(rb));
      ha.sc = __uint_as_float((unsigned)ea << 23); ha.sc2 = __uint_as_float((unsigned)(ea + 11) << 23);
      hb.sc = __uint_as_float((unsigned)eb << 23); hb.sc2 = __uint_as_float((unsigned)(eb + 11) << 23);
      inva = __uint_as_float((unsigned)(254 - ea) << 23); invb = __uint_as_float((unsigned)(254 - eb) << 23);
    }
  };
  constexpr int NBUF = PIPE ? 2 : 1;
  unsigned* sA[2] = {lds, lds + (NBUF - 1) * OA::WORDS};
  unsigned* sB[2] = {lds + NBUF * OA::WORDS, lds + NBUF * OA::WORDS + (NBUF - 1) * OB::WORDS};
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = EDGE ? (p.N + BN - 1) / BN : p.N / BN;
  const int alast = EDGE ? (AKM ? p.M - 4 : p.M - 1) : 0, blast = EDGE ? (BKM ? p.N - 4 : p.N - 1) : 0;
  int tile, split = 0;
  if (p.splits == 1) {
    tile = xcd_swizzle(bx, gx);
  } else {  // an XCD owns a run of tiles with all their splits (as gemm_f32_kernel)
    const int x = bx & 7, j = bx >> 3;
    const int q = p.tiles >> 3, r = p.tiles & 7, run = q + (r ? 1 : 0);
    const int nt = q + (x < r ? 1 : 0);
    split = j / run;
    const int tl = j - split * run;
    if (tl >= nt) return;
    tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + tl;
  }
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int kbeg = split * p.ksplit_len;
  const int kend = min(p.K, kbeg + p.ksplit_len);
  const int nk = EDGE ? (kend - kbeg + SBK - 1) / SBK : (kend - kbeg) / SBK;
  const int klast = EDGE ? p.K - 1 : 0x7ffffffe;

  f32x16 acc[MT][NT];
  f32x16 acc2[H16 ? MT : 1][H16 ? NT : 1];  // (fp16 split: the l h + h l terms, scaled by 2^11)
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[i][j][r] = 0.f;
        if constexpr (H16) acc2[i][j][r] = 0.f;
      }
  OA las[D];
  OB lbs[D];
  OA& la = las[0];
  OB& lb = lbs[0];
  const bool do_rs = AKM && p.rowsum && n0 == 0;  // bias gradient riding the dW contraction (tile column 0)
  float4 rs = make_float4(0.f, 0.f, 0.f, 0.f);
  const int fr = lane & 31, g = lane >> 5;
  auto fetch = [&](int t) {
    la.template load<EDGE>(p.A, p.lda, m0, kbeg + t * SBK, tid, alast, p.K);
    lb.template load<EDGE>(p.B, p.ldb, n0, kbeg + t * SBK, tid, blast, p.K);
    if (AKM && p.kscale) la.scale_k(p.kscale, p.krows_per, kbeg + t * SBK, tid, klast);
  };
  auto stage = [&](unsigned* a_s, unsigned* b_s) {
    if (AKM && do_rs) la.accum(rs, tid);
    la.store(a_s, tid, ha);
    lb.store(b_s, tid, hb);
  };
  auto mma = [&](const unsigned* a_s, const unsigned* b_s) {
#pragma unroll
    for (int ks = 0; ks < SBK / 16; ++ks) {
      bf16x8 af[MT][3], bf[NT][3];
#pragma unroll
      for (int i = 0; i < MT; ++i) OA::frag(a_s, wm * (BM / 2) + i * 32 + fr, g, ks, af[i]);
#pragma unroll
      for (int j = 0; j < NT; ++j) OB::frag(b_s, wn * (BN / 2) + j * 32 + fr, g, ks, bf[j]);
      if constexpr (H16) {
        // l h, h l into the second accumulator set, h h into the first; term-major as below
#pragma unroll
        for (int tm = 0; tm < 3; ++tm)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
              const f16x8 a = __builtin_bit_cast(f16x8, af[i][tm == 0 ? 1 : 0]), b = __builtin_bit_cast(f16x8, bf[j][tm == 1 ? 1 : 0]);
              if (tm < 2) acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc2[i][j], 0, 0, 0);
              else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i][j], 0, 0, 0);
            }
        continue;
      }
      // small terms first; term-major over the MT x NT accumulators (same sums, bit for bit): consecutive MFMAs write
      // DIFFERENT accumulators, so none waits for its predecessor's result (six back-to-back MFMAs on one accumulator are a
      // dependent chain: RSCOTR_X6_CHAIN below restores that order for A/B builds)
      constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#ifdef RSCOTR_X6_CHAIN
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int tm = 0; tm < 6; ++tm)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA[tm]], bf[j][PB[tm]], acc[i][j], 0, 0, 0);
#else
#pragma unroll
      for (int tm = 0; tm < 6; ++tm)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA[tm]], bf[j][PB[tm]], acc[i][j], 0, 0, 0);
#endif
    }
  };
  if (PIPE >= 2) {
    // Steady state without branches inside a step (the scheduler interleaves within one basic block): loads past the end
    // re-read the last tile, the last step stages it a second time into the idle LDS stage (its row sums times 0).
    constexpr int U = (D % 2 == 0) ? D : 2 * D;  // steps per unrolled round: register set and LDS stage indices static
    constexpr int NMFMA = MT * NT * (H16 ? 3 : 6) * (SBK / 16);
    constexpr int VPM = H16 ? RSCOTR_H3_VPM : 4, DPM = H16 ? RSCOTR_H3_DPM : 1;  // VALU / DS writes the scheduler places behind each MFMA
    // per-sample k scaling of a k-major A (weight gradients under DropPath / Mixup): always applied, so that a step stays
    // one basic block — without a scale vector every k reads the constant 1
    const float* ksp = (AKM && p.kscale) ? p.kscale : &bf16x6_one;
    const int ksper = (AKM && p.kscale) ? p.krows_per : 0x7fffffff;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const int tt = min(d, nk - 1);
      las[d].template load<EDGE>(p.A, p.lda, m0, kbeg + tt * SBK, tid, alast, p.K);
      lbs[d].template load<EDGE>(p.B, p.ldb, n0, kbeg + tt * SBK, tid, blast, p.K);
    }
    h3_scales();
    if (AKM) las[0].scale_k(ksp, ksper, kbeg, tid, klast);
    if (AKM) las[0].accum(rs, tid);
    las[0].store(sA[0], tid, ha);
    lbs[0].store(sB[0], tid, hb);
    __syncthreads();
    for (int t0 = 0; t0 < nk; t0 += U) {
#pragma unroll
      for (int s = 0; s < U; ++s) {
        const int t = t0 + s;
        if (t < nk) {
          {  // tile t + D into the set tile t left (staged during step t - 1 / the prologue)
            const int tt = min(t + D, nk - 1);
            las[s % D].template load<EDGE>(p.A, p.lda, m0, kbeg + tt * SBK, tid, alast, p.K);
            lbs[s % D].template load<EDGE>(p.B, p.ldb, n0, kbeg + tt * SBK, tid, blast, p.K);
          }
          mma(sA[s & 1], sB[s & 1]);
          if (AKM) las[(s + 1) % D].scale_k(ksp, ksper, kbeg + min(t + 1, nk - 1) * SBK, tid, klast);
          if (AKM) las[(s + 1) % D].accum(rs, tid, t + 1 < nk ? 1.f : 0.f);
          las[(s + 1) % D].store(sA[(s + 1) & 1], tid, ha);
          lbs[(s + 1) % D].store(sB[(s + 1) & 1], tid, hb);
#pragma unroll
          for (int i = 0; i < NMFMA; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);    // MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);  // VALU
            __builtin_amdgcn_sched_group_barrier(0x200, DPM, 0);  // DS write
          }
          __syncthreads();
        }
      }
    }
  } else if (PIPE) {
    fetch(0);
    h3_scales();
    stage(sA[0], sB[0]);
    if (nk > 1) fetch(1);
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
      const bool odd = t & 1;  // (selects, not a runtime-indexed pointer array: the accesses must stay LDS accesses)
      mma(odd ? sA[1] : sA[0], odd ? sB[1] : sB[0]);
      if (t + 1 < nk) {  // registers hold tile t+1: split / pack / write to the other stage, then fetch tile t+2
        stage(odd ? sA[0] : sA[1], odd ? sB[0] : sB[1]);
        if (t + 2 < nk) fetch(t + 2);
      }
      __syncthreads();
    }
  } else {
    fetch(0);
    h3_scales();
    for (int t = 0; t < nk; ++t) {
      __syncthreads();  // the previous tile has been consumed
      stage(sA[0], sB[0]);
      if (t + 1 < nk) fetch(t + 1);  // next tile's global loads: requested before the barrier, in flight under it and the MFMAs
      __syncthreads();
      mma(sA[0], sB[0]);
    }
  }

  if (AKM && do_rs) {  // thread t summed rows (t % (BM/4)) * 4 .. + 3 over the k-pairs it staged: fold the 8 k-lanes
    __syncthreads();
    constexpr int KL = 256 / (BM / 4) < OA::KP ? 256 / (BM / 4) : OA::KP;  // distinct k-lanes among the threads (item i of a
    float4* red = reinterpret_cast<float4*>(lds);                            // thread has the same rows: 256 % (BM / 4) == 0)
    if (tid < KL * BM / 4) red[tid] = rs;  // [k-lane][BM / 4]
    __syncthreads();
    if (tid < BM) {
      const float* rf = reinterpret_cast<const float*>(lds);
      float v = 0.f;
#pragma unroll
      for (int k = 0; k < KL; ++k) v += rf[k * BM + tid];
      const int m = m0 + tid;
      if (!EDGE || m < p.M) {
        if (p.splits > 1 || SLAB) p.rs_slabs[(long)split * p.M + m] = v;
        else p.rowsum[m] = p.rowsum_acc ? p.rowsum[m] + v : v;
      }
    }
  }

  if constexpr (H16) {  // C = (hh + (lh + hl) 2^-11) 2^-(sa + sb)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = fmaf(acc2[i][j][r], 0x1p-11f, acc[i][j][r]) * inva * invb;
  }
  if (p.splits > 1 || SLAB) {
    float* slab = p.slabs + (long)split * p.M * p.N;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 32 + fr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
          if (!EDGE || (m < p.M && n < p.N)) slab[(long)m * p.N + n] = acc[i][j][r];
        }
      }
    return;
  }
  if constexpr (BM == 128 && BN == 128 && !EDGE) {
    // the ReLU gate as one bit per element (ACT_RELU_BITS / ACT_RELU_GRAD_BITS, gemm_common.h): the forward product of an FFN leaves
    // 8 bytes per thread and tile next to its activation, and dH = (g W) * [h > 0] reads them back instead of the M x N activation
    // (10880 x 2048: 89 MB in 512-byte row segments at the end of every workgroup — what bounds that launch).  Host-checked: no pre /
    // residual / accumulate / row scale / second output with these codes.
    if (p.act == ACT_RELU_BITS || p.act == ACT_RELU_GRAD_BITS) {
      const bool fwd = p.act == ACT_RELU_BITS;
      const long widx = ((long)(m0 / 128) * (p.N / 128) + n0 / 128) * 256 + tid;
      unsigned long long bits = fwd ? 0ull : reinterpret_cast<const unsigned long long*>(p.aux)[widx];
      float amx = 0.f;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int n = n0 + wn * (BN / 2) + j * 32 + fr;
          const float bv = p.bias ? p.bias[n] : 0.f;
          float* crow = p.C + (long)(m0 + wm * (BM / 2) + i * 32 + 4 * g) * p.ldc + n;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int b = (i * NT + j) * 16 + r;
            float v = acc[i][j][r] + bv;
            if (fwd) {
              v = fmaxf(v, 0.f);
              bits |= (unsigned long long)(v > 0.f) << b;
            } else {
              v = ((bits >> b) & 1ull) ? v : 0.f;
            }
            crow[(long)((r & 3) + 8 * (r >> 2)) * p.ldc] = v;
            amx = fmaxf(amx, fabsf(v));
          }
        }
      if (fwd) reinterpret_cast<unsigned long long*>(p.pre)[widx] = bits;
      amax_commit(p.amax_out, amx);
      return;
    }
  }
  const bool plain = !p.pre && p.act == ACT_NONE && !p.resid && !p.accumulate && !p.rowscale && !p.C2;
  // exactly one extra tensor read by the epilogue (aux of act', residual, or old C): its 16 values per tile in one batch
  // (64 x 64 tiles only: on the 128 x 128 one-stage bf16 kernel the 16-register batch costs the third resident workgroup, and on
  // the fp16 one — measured cold, 10880 x 2048 x 256 + residual: 109.7 against 105.8 us — it buys nothing: those launches are
  // bound by the 190 MB their epilogue moves in 512-byte row segments 8 KB apart)
  // (Requesting the 16 values in the PROLOGUE instead, so that they wait in registers through the k loop, measured SLOWER in the
  // step: 34.27 against 34.11 ms per round on one box, two runs each — the in-order load counter makes the second k-step wait
  // for them, and 130 + 32 registers leave the scheduler no slack under the three-workgroup cap.)
  const bool one_extra = BM == 64 && !p.C2 &&
                         ((p.act == ACT_RELU_GRAD || p.act == ACT_GELU_GRAD) ? 1 : 0) + (p.resid ? 1 : 0) + (p.accumulate ? 1 : 0) == 1;
  // an activation (and / or the stored pre-activation) but no tensor to read: epilogue_noload16
#ifdef RSCOTR_NO_NOLOAD  // (A/B builds: scripts/build_variant.sh)
  const bool noload = false;
#else
  const bool noload = !plain && (p.act == ACT_NONE || p.act == ACT_RELU || p.act == ACT_GELU) && !p.resid && !p.accumulate &&
                      !p.rowscale && !p.C2;
#endif
  float amx = 0.f;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + wn * (BN / 2) + j * 32 + fr;
      if (EDGE && n >= p.N) continue;
      const float bv = p.bias ? p.bias[n] : 0.f;
      const int mb = m0 + wm * (BM / 2) + i * 32 + 4 * g;
      float* crow = p.C + (long)mb * p.ldc + n;
      if (one_extra) {
        epilogue_tile16<EDGE>(p, acc[i][j], bv, mb, n, amx);
        __builtin_amdgcn_sched_barrier(0);
      } else if (noload) {
        epilogue_noload16<EDGE>(p, acc[i][j], bv, mb, n, amx);
      } else if (plain) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (!EDGE || mb + (r & 3) + 8 * (r >> 2) < p.M) {
            const float v = acc[i][j][r] + bv;
            crow[(long)((r & 3) + 8 * (r >> 2)) * p.ldc] = v;
            amx = fmaxf(amx, fabsf(v));
          }
      } else {
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          float v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = acc[i][j][4 * g4 + u] + bv;
          epilogue_rows4<EDGE>(p, v, mb + 8 * g4, n, amx);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  amax_commit(p.amax_out, amx);
}

template <int BM, int BN, bool AKM, bool BKM, int PIPE, bool EDGE = false>
__global__ __launch_bounds__(256) void gemm_bf16x6_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) unsigned lds[bf16x6_lds_words<BM, BN, AKM, BKM, PIPE>()];
  gemm_bf16x6_body<BM, BN, AKM, BKM, PIPE, false, EDGE>(p, blockIdx.x, gridDim.x, lds);
}

// OCC: wavefronts per SIMD the register allocation is held to (1: the compiler's own choice).  The interior pipelined 64 x 64
// kernels with a row-major A take 134 / 146 registers on their own and 126 / 128 without a spill when asked: four workgroups
// per CU instead of three
template <int BM, int BN, bool AKM, bool BKM, int PIPE, bool EDGE = false, int OCC = 1, bool BPL = false>
__global__ __launch_bounds__(256, OCC) void gemm_h3_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) unsigned lds[bf16x6_lds_words<BM, BN, AKM, BKM, PIPE, true>()];
  gemm_bf16x6_body<BM, BN, AKM, BKM, PIPE, false, EDGE, true, BPL>(p, blockIdx.x, gridDim.x, lds);
}
// 128 x 128 tiles: two accumulator sets are 128 registers; held to two wavefronts per SIMD (256 registers in all) so that
// two workgroups per CU cover each other's staging phases in the one-stage loop
template <bool AKM, bool BKM, bool EDGE = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_h3_128_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) unsigned lds[bf16x6_lds_words<128, 128, AKM, BKM, 0, true>()];
  gemm_bf16x6_body<128, 128, AKM, BKM, 0, false, EDGE, true>(p, blockIdx.x, gridDim.x, lds);
}

// ---------------------------------------------------------------------------------------------------------------
// bf16x6 with PRE-SPLIT WEIGHTS (rscotr_gemm_f32_wplanes; scripts/lab/planes_lab.hip is the stand-alone version).
// In y = x W^T and dx = dy W the B operand is a parameter: it changes once per optimizer step, yet the kernels above split it
// into bf16 planes again in every workgroup of every launch (a 10880 x 2048 x 256 product converts W 85 times).  Here the
// planes are written ONCE per step by rscotr_gemm_split_weights, in a k-step-major layout [K / 16][Npad][3 planes][16 k]
// bf16 (Npad = N rounded up to 128, zero rows behind N), so that the B stage of a workgroup is one contiguous run of
// 96-byte rows that goes global -> VGPR -> LDS with no VALU work at all; the transposed set (planes of W^T) serves
// dx = dy W.  Only A (the activation, fp32, row-major) is split while it is staged — once per 256 (128) output columns.
// Workgroup = 512 threads = 8 wavefronts; tile 128 x 256 (wave tile 64 x 64) or 128 x 128 (wave tile 32 x 64); two LDS
// stages, one barrier per 16-k step, three register sets of prefetch (tile t + 3 is requested at the top of step t), the
// split / pack / LDS writes of tile t + 1 interleaved with the MFMAs of tile t (sched_group_barrier).  Rows past M are
// clamped loads / guarded stores, columns past N are zero planes / guarded stores: any M, N; K % 16 == 0.
// Lab (MI355X, no epilogue): 10880 x 256 x 2048 in 3 k-slices 70 us against 105 for the in-kernel split, 32768 x 384 x 96
// 26 against 47, 8192 x 768 x 192 21 against 35, 2048 x 1536 x 384 22 against 30, 4096^3 207 TFLOP/s-equivalent against 174.
constexpr int WPL_LDR = 56;  // bf16 per LDS row: 3 planes x 16 k + 8 pad (112 bytes: conflict-free 16-byte fragment reads)

struct WplRegs {
  float4 a;
  uint4 b[3];
  __device__ __forceinline__ void load(const float* a_src, const unsigned short* b_src, long a_off, long b_off, bool b_active) {
    a = *reinterpret_cast<const float4*>(a_src + a_off);
    if (b_active) {
      const uint4* s = reinterpret_cast<const uint4*>(b_src + b_off);
      b[0] = s[0]; b[1] = s[1]; b[2] = s[2];
    }
  }
  __device__ __forceinline__ void store(unsigned* a_s, unsigned* b_s, int tid, bool b_active) const {
    __bf16 x[3], y[3], z[3], w[3];
    split_planes<3>(a.x, x); split_planes<3>(a.y, y); split_planes<3>(a.z, z); split_planes<3>(a.w, w);
    unsigned* dst = a_s + ((tid >> 2) * WPL_LDR + (tid & 3) * 4) / 2;
#pragma unroll
    for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(dst + p * 8) = make_uint2(pack_bf16(x[p], y[p]), pack_bf16(z[p], w[p]));
    if (b_active) {
      uint4* d4 = reinterpret_cast<uint4*>(b_s + ((tid >> 1) * WPL_LDR + (tid & 1) * 24) / 2);
      d4[0] = b[0]; d4[1] = b[1]; d4[2] = b[2];
    }
  }
};

template <int BN> constexpr size_t wplanes_lds_bytes() { return 2 * (size_t)(128 + BN) * WPL_LDR * 2; }

// p.B is unused; `planes` = the pre-split B, npad = its row count per k-step.  p.tiles = tiles_m * tiles_n, p.splits k-slices
// (k-steps divided evenly), slabs as in the kernels above.
template <int BN>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_wplanes_kernel(
    GemmParams p, const unsigned short* __restrict__ planes, int npad) {
  constexpr int BM = 128, DEPTH = 3;
  constexpr int WNW = BN / 64, WMW = 8 / WNW, MT = BM / WMW / 32, NT = 2;
  constexpr int A_WORDS = BM * WPL_LDR / 2, B_WORDS = BN * WPL_LDR / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned wpl_lds[];
  unsigned* sA[2] = {wpl_lds, wpl_lds + A_WORDS};
  unsigned* sB[2] = {wpl_lds + 2 * A_WORDS, wpl_lds + 2 * A_WORDS + B_WORDS};
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WNW, wn = wave % WNW;
  const int tiles_n = (p.N + BN - 1) / BN;
  // grid = tiles x k-slices EXACTLY: one (128 x 256) workgroup is resident per CU, so a grid padded past 256 workgroups
  // (the XCD-run mapping of the kernels above) would leave a handful of them to a second round of the whole chip
  const int idx = xcd_swizzle(blockIdx.x, gridDim.x);
  const int split = idx / p.tiles, tile = idx - split * p.tiles;
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int nk_all = p.K / 16;
  const int kt0 = (int)((long)split * nk_all / p.splits), kt1 = (int)((long)(split + 1) * nk_all / p.splits);
  const int nk = kt1 - kt0;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  WplRegs sets[DEPTH];
  const float* a_src = p.A + (long)min(m0 + (tid >> 2), p.M - 1) * p.lda + (tid & 3) * 4;  // rows past M: clamped reads
  const bool b_active = BN == 256 || tid < 256;
  const unsigned short* b_src = planes + ((long)n0 + (tid >> 1)) * 48 + (tid & 1) * 24;      // (n0 + BN <= npad)
  const long b_step = (long)npad * 48;
  const int fr = lane & 31, g = lane >> 5;
  auto mma = [&](const unsigned* a_s, const unsigned* b_s) {
    bf16x8 af[MT][3], bf[NT][3];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const unsigned* q = a_s + ((wm * (BM / WMW) + i * 32 + fr) * WPL_LDR + 8 * g) / 2;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) af[i][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(q + pl * 8));
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const unsigned* q = b_s + ((wn * 64 + j * 32 + fr) * WPL_LDR + 8 * g) / 2;
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) bf[j][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(q + pl * 8));
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {  // small terms first
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
      }
  };
  constexpr int U = 2 * DEPTH;
  constexpr int NMFMA = MT * NT * 6;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    const int kt = kt0 + min(d, nk - 1);
    sets[d].load(a_src, b_src, (long)kt * 16, kt * b_step, b_active);
  }
  sets[0].store(sA[0], sB[0], tid, b_active);
  __syncthreads();
  for (int t0 = 0; t0 < nk; t0 += U) {
#pragma unroll
    for (int s = 0; s < U; ++s) {
      const int t = t0 + s;
      if (t < nk) {
        {
          const int kt = kt0 + min(t + DEPTH, nk - 1);
          sets[s % DEPTH].load(a_src, b_src, (long)kt * 16, kt * b_step, b_active);
        }
        mma(sA[s & 1], sB[s & 1]);
        sets[(s + 1) % DEPTH].store(sA[(s + 1) & 1], sB[(s + 1) & 1], tid, b_active);
#pragma unroll
        for (int i = 0; i < NMFMA; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);  // VALU
          __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // DS write
        }
        __syncthreads();
      }
    }
  }

  if (p.splits > 1) {
    float* slab = p.slabs + (long)split * p.M * p.N;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = n0 + wn * 64 + j * 32 + fr;
        if (n >= p.N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * (BM / WMW) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
          if (m < p.M) slab[(long)m * p.N + n] = acc[i][j][r];
        }
      }
    return;
  }
  float amx = 0.f;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + wn * 64 + j * 32 + fr;
      if (n >= p.N) continue;
      const float bv = p.bias ? p.bias[n] : 0.f;
      const int mb = m0 + wm * (BM / WMW) + i * 32 + 4 * g;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = acc[i][j][4 * g4 + u] + bv;
        epilogue_rows4<true>(p, v, mb + 8 * g4, n, amx);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
}

// Split weights into the plane layout above: table rows {W, planes, N, K, ldw, npad, first block, transposed} (int64 x 8);
// transposed = 0: planes of W (N rows, reduction over K: y = x W^T); 1: planes of W^T (rows = the K columns of W, reduction
// over N: dx = dy W), N % 16 == 0 then.  One thread per (row, k-step): 16 values, one 96-byte output row.
__global__ __launch_bounds__(256) void split_weights_kernel(const int64_t* __restrict__ table, int n_entries) {
  int e = 0;
  while (e + 1 < n_entries && (long)table[(long)(e + 1) * 8 + 6] <= (long)blockIdx.x) ++e;
  const int64_t* t = table + (long)e * 8;
  const float* W = reinterpret_cast<const float*>(t[0]);
  unsigned short* planes = reinterpret_cast<unsigned short*>(t[1]);
  const int N = (int)t[2], K = (int)t[3], ldw = (int)t[4], npad = (int)t[5], tr = (int)t[7];
  const int rows = tr ? K : N, red = tr ? N : K;  // output rows, reduction length
  const long idx = ((long)blockIdx.x - t[6]) * 256 + threadIdx.x;
  const int nkt = red / 16;
  // consecutive threads take consecutive k-steps of one row (tr = 0: contiguous 64-byte reads) or consecutive rows of one
  // k-step (tr = 1: W is read along its rows)
  const int row = tr ? (int)(idx % npad) : (int)(idx / nkt);
  const int kt = tr ? (int)(idx / npad) : (int)(idx % nkt);
  if (kt >= nkt || row >= npad) return;
  float v[16];
  if (row >= rows) {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = 0.f;
  } else if (!tr) {
    const float4* src = reinterpret_cast<const float4*>(W + (long)row * ldw + kt * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) { const float4 x = src[q]; v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w; }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = W[(long)(kt * 16 + i) * ldw + row];
  }
  unsigned out[3][8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    __bf16 a[3], b[3];
    split_planes<3>(v[2 * q], a);
    split_planes<3>(v[2 * q + 1], b);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) out[pl][q] = pack_bf16(a[pl], b[pl]);
  }
  uint4* dst = reinterpret_cast<uint4*>(planes + ((long)kt * npad + row) * 48);
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    dst[pl * 2] = make_uint4(out[pl][0], out[pl][1], out[pl][2], out[pl][3]);
    dst[pl * 2 + 1] = make_uint4(out[pl][4], out[pl][5], out[pl][6], out[pl][7]);
  }
}

// Planes of weights for PlaneOperandH (rscotr_gemm_split_weights_h3): table rows {W, planes, rows of W, cols of W, ldw, rpad,
// first block, transposed, range word of the parameter} (int64 x 9).  transposed = 0: planes of W (plane rows = rows of W,
// reduction over its columns: y = x W^T); 1: planes of W^T (plane rows = columns of W, reduction over its rows: dx = dy W).
// The reduction length is a multiple of 32; rpad = plane rows rounded up to 64, the rows past the end are zeros.  One thread per
// (k-step, plane row): 32 values in, one 128-byte record {h[32], l[32]} out; an entry takes ceil(rpad * (reduction / 32) / 256)
// blocks.  Consecutive threads take consecutive plane rows of one k-step (the records of a k-step are contiguous; transposed
// reads run along the rows of W).
__global__ __launch_bounds__(256) void split_weights_h3_kernel(const int64_t* __restrict__ table, int n_entries) {
  int e = 0;
  while (e + 1 < n_entries && (long)table[(long)(e + 1) * 9 + 6] <= (long)blockIdx.x) ++e;
  const int64_t* t = table + (long)e * 9;
  const float* W = reinterpret_cast<const float*>(t[0]);
  uint4* planes = reinterpret_cast<uint4*>(t[1]);
  const int wrows = (int)t[2], wcols = (int)t[3], ldw = (int)t[4], rpad = (int)t[5], tr = (int)t[7];
  const int rows = tr ? wcols : wrows, red = tr ? wrows : wcols;
  const int se = h3_scale_exp(amax_read(reinterpret_cast<const unsigned*>(t[8])));
  const H3Scale hs{__uint_as_float((unsigned)se << 23), __uint_as_float((unsigned)(se + 11) << 23)};
  const long idx = ((long)blockIdx.x - t[6]) * 256 + threadIdx.x;
  const int nkt = red / 32;
  const int kt = (int)(idx / rpad), row = (int)(idx % rpad);
  if (kt >= nkt) return;
  float v[32];
  if (row >= rows) {
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = 0.f;
  } else if (!tr) {
    const float4* src = reinterpret_cast<const float4*>(W + (long)row * ldw + kt * 32);
#pragma unroll
    for (int q = 0; q < 8; ++q) { const float4 x = src[q]; v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w; }
  } else {
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = W[(long)(kt * 32 + i) * ldw + row];
  }
  unsigned h[16], l[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    unsigned o[3];
    split_pair_h(v[2 * q], v[2 * q + 1], hs, o);
    h[q] = o[0]; l[q] = o[1];
  }
  uint4* dst = planes + ((long)kt * rpad + row) * 8;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    dst[q] = make_uint4(h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]);
    dst[4 + q] = make_uint4(l[4 * q], l[4 * q + 1], l[4 * q + 2], l[4 * q + 3]);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Grouped launch of deferred weight gradients (rscotr_gemm_dw_group): MANY dW = A^T B problems with small outputs (the
// 256 x 256 projections of the encoder / decoders, the Swin stage 1-2 Linears: ~110 launches of 8-40 us per co-training
// round, each a short grid that ramps up and drains alone) run as ONE launch.  Operands are the k-major activations /
// gradients kept alive until the end of backward; every problem is cut into 64 x 64 tiles x k-slices of about equal length
// (so few slices per problem: the slab traffic of 31-slice launches goes away), slabs + row-sum partials go to the deferred
// combine (rscotr_splitk_flush), which orders problems that share a destination.
// table: device (n, 16) int64 rows {A, B, slabs, rs_slabs | 0, kscale | 0, M, N, K, lda, ldb, ksplit_len, splits,
// first workgroup OF THE BUNDLE, krows_per, 0, workgroups of the problem = tiles * splits}, n a multiple of 8: rows come in
// bundles of 8 (padded with rows of 0 workgroups) that occupy 8 * max(workgroups of the bundle's rows) consecutive ids,
// row x of a bundle taking the ids = x mod 8 (see the kernel).
constexpr size_t GROUP_LDS_BYTES = 4 * (size_t)bf16x6_lds_words<128, 128, true, true, 0>();  // 24 KB (>= the fp32 body's 17 KB)
// VAR 0: fp32 matrix pipe on 64 x 64 tiles with bounds handling (any problem); VAR 6: the six-term bf16 split product on
// 128 x 128 tiles with edge handling (M, N, K multiples of 4, 16-byte aligned operands).  Separate instantiations rather than one
// kernel with both bodies: the 128 x 128 body's registers (114 + 64 accumulators) would halve the residency of the fp32 body's
// workgroups (measured: 950 -> 1500 us for the launch).  (Variants 2 / 3 / 4 of rounds 3-4 — interior-only 128 x 128, 64 x 64
// pipelined, fp32 on 128 x 128 — lost every A/B to variant 6 and left the library in round 5.)
// VAR 7 (round 5): the fp16 split product on the 128 x 128 edge body; table column 14 = (slot of A + 1) << 32 | slot of B + 1,
// indices into `amax_base` (the value-range words of the two operands).
template <int VAR>
__device__ __forceinline__ void gemm_group_dispatch(const int64_t* __restrict__ table, int n, const unsigned* __restrict__ amax_base) {
  extern __shared__ __attribute__((aligned(16))) float gemm_smem[];
  // the problem of this workgroup.  The table comes in BUNDLES of 8 rows that share the first-workgroup column: workgroup
  // first + 8 j + x is the j-th workgroup of the bundle's row x, so that (round-robin dispatch: XCD = id % 8) ALL tiles and
  // k-slices of a problem run on one XCD and its operands are fetched into that L2 once (with the tiles of a problem
  // spread over the XCDs a 256 x 256 x 10880 problem pulled its operands from HBM three times over).  Binary search over
  // the bundles (every thread, uniform: no static LDS in front of the dynamic region the body carves with 16-byte accesses)
  int lo = 0, hi = (n >> 3) - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)table[(long)mid * 8 * 16 + 12] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const int rel = (int)blockIdx.x - (int)table[(long)lo * 8 * 16 + 12];
  const int64_t* t = table + ((long)lo * 8 + (rel & 7)) * 16;
  const int jwg = rel >> 3;
  if (jwg >= (int)t[15]) return;  // (rows of a bundle differ in size; empty rows have 0 workgroups)
  GemmParams p;
  p.A = reinterpret_cast<const float*>(t[0]);
  p.B = reinterpret_cast<const float*>(t[1]);
  p.slabs = reinterpret_cast<float*>(t[2]);
  p.rs_slabs = reinterpret_cast<float*>(t[3]);
  p.kscale = reinterpret_cast<const float*>(t[4]);
  p.M = (int)t[5]; p.N = (int)t[6]; p.K = (int)t[7]; p.lda = (int)t[8]; p.ldb = (int)t[9];
  p.ksplit_len = (int)t[10]; p.splits = (int)t[11];
  p.krows_per = (int)t[13];
  p.C = nullptr; p.C2 = nullptr; p.bias = nullptr; p.aux = nullptr; p.pre = nullptr; p.resid = nullptr; p.rowscale = nullptr;
  p.ldc = p.N; p.act = ACT_NONE; p.accumulate = 0; p.rows_per = 1; p.rowsum_acc = 0;
  p.rowsum = p.rs_slabs;  // non-null = the row sums are wanted (they go to rs_slabs)
  p.vecA = ((t[0] & 15) == 0) && (p.lda % 4 == 0);
  p.vecB = ((t[1] & 15) == 0) && (p.ldb % 4 == 0);
  p.vecC = 0;
  p.nb1 = 0; p.nb2 = 1;
  if constexpr (VAR == 7) {
    p.amax_a = amax_base + (unsigned)((uint64_t)t[14] >> 32) - 1;
    p.amax_b = amax_base + (unsigned)((uint64_t)t[14] & 0xffffffffu) - 1;
  }
  p.tiles = (VAR == 6 || VAR == 7) ? ((p.M + 127) / 128) * ((p.N + 127) / 128) : ((p.M + 63) / 64) * ((p.N + 63) / 64);
  // the bodies decode (tile, k-slice) from a workgroup id laid out for XCD runs (x = id & 7 owns a run of tiles, id >> 3 =
  // slice * run + position in the run): build the id whose decoding is (tile = jwg % tiles, slice = jwg / tiles)
  const int tl_ = jwg % p.tiles, sl_ = jwg / p.tiles;
  const int q_ = p.tiles >> 3, r_ = p.tiles & 7, run_ = q_ + (r_ ? 1 : 0);
  int x_, pos_;
  if (tl_ < r_ * (q_ + 1)) { x_ = tl_ / (q_ + 1); pos_ = tl_ - x_ * (q_ + 1); }
  else { const int u_ = tl_ - r_ * (q_ + 1); x_ = r_ + u_ / q_; pos_ = u_ - (x_ - r_) * q_; }
  const int bx = 8 * (sl_ * run_ + pos_) + x_;
  const int gx = p.splits > 1 ? 8 * run_ * p.splits : p.tiles;  // (one k-slice: the single-slice tile order, result still as slab 0)
  if constexpr (VAR == 6) {
    // bf16x6 on 128 x 128 tiles with edge handling: every member with min(M, N) >= 48 — interior or ragged (Swin stage 1 / 2:
    // 96, 192, 288, 576 rows or columns); on the fp32 pipe of variant 0 the ragged ones ran at 41 TFLOP/s
    gemm_bf16x6_body<128, 128, true, true, 0, true, true>(p, bx, gx, reinterpret_cast<unsigned*>(gemm_smem));
  } else if constexpr (VAR == 7) {
    gemm_bf16x6_body<128, 128, true, true, 0, true, true, true>(p, bx, gx, reinterpret_cast<unsigned*>(gemm_smem));
  } else {
    gemm_f32_body<64, 64, 2, 2, true, true, true, 1, true>(p, bx, gx, 0);
  }
}

template <int VAR>
__global__ __launch_bounds__(256) void gemm_f32_group_kernel(const int64_t* __restrict__ table, int n) {
  gemm_group_dispatch<VAR>(table, n, nullptr);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_h3_group_kernel(
    const int64_t* __restrict__ table, int n, const unsigned* __restrict__ amax_base) {
  gemm_group_dispatch<7>(table, n, amax_base);
}

// Tile / slice choice of the bf16x6 kernel for one problem; bm == 0: not its domain (the fp32 pipe takes it).
struct Split6Cfg {
  int bm, splits, klen;
};

static Split6Cfg choose_split6(const GemmParams& p, int a_kmajor, int b_kmajor, int64_t ws_bytes) {
  Split6Cfg c{0, 1, p.K};
  static const int on = getenv("RSCOTR_BF16X6") ? atoi(getenv("RSCOTR_BF16X6")) : 1;
  static const long t128_min = getenv("RSCOTR_BF16X6_T128") ? atol(getenv("RSCOTR_BF16X6_T128")) : 512;
  static const long t64_min = getenv("RSCOTR_BF16X6_T64") ? atol(getenv("RSCOTR_BF16X6_T64")) : 256;  // (round 4: 256 measures -0.3 ms per round on mtl512 against 512 — Swin stage-2 / -4 products move to the split product, flop share 0.85 -> 0.90; 384: -0.2; 192 and 128 lose 0.7.  It also re-routes the 2500-row stage-4 products of the 800 x 800 det step: that parity run passes with its tensors outside the 1e-3 tier explained by the fp64 anchor)
  static const long dw_t128_min = getenv("RSCOTR_BF16X6_DW_T128") ? atol(getenv("RSCOTR_BF16X6_DW_T128")) : 24;
  static const int k_min = getenv("RSCOTR_BF16X6_KMIN") ? atoi(getenv("RSCOTR_BF16X6_KMIN")) : 192;
  static const int mid_split = getenv("RSCOTR_BF16X6_MIDSPLIT") ? atoi(getenv("RSCOTR_BF16X6_MIDSPLIT")) : 1;
  static const int gelu_ok = getenv("RSCOTR_BF16X6_GELU") ? atoi(getenv("RSCOTR_BF16X6_GELU")) : 1;
  static const long mid_t64 = getenv("RSCOTR_BF16X6_MID_T64") ? atol(getenv("RSCOTR_BF16X6_MID_T64")) : 96;  // (round 4: 96 takes the 512-row Swin stage-4 products with K >= 2304 (96 tiles, 6 k-slices): -0.25 ms per round against 128)
  static const int mid_k = getenv("RSCOTR_BF16X6_MID_K") ? atoi(getenv("RSCOTR_BF16X6_MID_K")) : 1024;
  static const int dw_ok = getenv("RSCOTR_BF16X6_DW") ? atoi(getenv("RSCOTR_BF16X6_DW")) : 1;
  // Measured on the step (profiles/r2_gemm_census.txt against profiles/history/r1_s7_gemm_census_fp32.txt): the split
  // product wins where the MFMA work dominates — the encoder FFN products (117 -> 85 us, 125 -> 100 us), their weight
  // gradients (124 -> 75 us), the 10880- / 2048-row products with K >= 256 (5-15 %) — and loses on small outputs (256 x 256
  // weight gradients: 22.6 -> 32.5 us: too few tiles to hide the staging), on K < 192 (conversion not amortised) and where
  // the epilogue's memory traffic bounds the launch anyway.
  static const int edge_ok = getenv("RSCOTR_BF16X6_EDGE") ? atoi(getenv("RSCOTR_BF16X6_EDGE")) : 1;
  if (!on || !p.vecA || !p.vecB || p.K % 4 || p.K < k_min || p.M < 64 || p.N < 64) return c;
  // ragged shapes (M = 4 x 13 294 rows at 800 x 800, N = 96 / 288 columns of Swin stage 1, K = 53 176 of the 800 x 800 weight
  // gradients) take the EDGE instantiations: clamped loads, zeros past K, guarded stores; a k-major operand is read four
  // rows at a time, so its row count must be a multiple of 4
  const bool ragged = p.M % 64 || p.N % 64 || p.K % 16;
  if (ragged && (!edge_ok || (a_kmajor && p.M % 4) || (b_kmajor && p.N % 4))) return c;
  if (!gelu_ok && (p.act == ACT_GELU || p.act == ACT_GELU_GRAD || p.pre)) return c;
  const long t64 = (long)((p.M + 63) / 64) * ((p.N + 63) / 64);
  const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
  // (a 128-wide tile on a ragged edge wastes up to half a tile per row / column of tiles: only where that is < 1/8 of the work)
  const bool fit128 = (p.M % 128 == 0 || p.M >= 1024) && (p.N % 128 == 0 || p.N >= 1024);
  if (a_kmajor && b_kmajor) {  // weight gradients: small outputs, long reductions -> k-slices through slabs
    if (!dw_ok || p.rowscale || p.K < 1024 || p.M % 128 || p.N % 128 || t128 < dw_t128_min) return c;
    const int bm = 128;
    const long tiles = t128;
    if (tiles > 2048) return c;
    static const long dw_wgs = getenv("RSCOTR_BF16X6_DW_WGS") ? atol(getenv("RSCOTR_BF16X6_DW_WGS")) : 512;  // workgroups a k-sliced weight gradient aims at
    long sp = std::max<long>(1, std::min<long>((dw_wgs + tiles - 1) / tiles, p.K / 256));
    const int64_t per = ((int64_t)p.M * p.N + p.M) * 4;
    if (sp > 1) sp = std::min<long>(sp, ws_bytes / per);
    if (sp < 1) sp = 1;
    int klen = (int)((p.K + sp - 1) / sp);
    klen = (klen + RSCOTR_X6_BK0 - 1) / RSCOTR_X6_BK0 * RSCOTR_X6_BK0;
    c.bm = bm; c.klen = klen; c.splits = (p.K + klen - 1) / klen;
    if (c.splits == 1) c.klen = p.K;
    return c;
  }
  if (p.kscale) return c;
  if (fit128 && t128 >= t128_min) c.bm = 128;
  else if (t64 >= t64_min && p.K <= 4096) c.bm = 64;
  else if (mid_split && t64 >= mid_t64 && p.K >= mid_k) {
    // mid-size outputs with a long reduction (Swin stage 3: 2048 x 384 x 1536): too few 64 x 64 tiles for the chip, so the
    // reduction is cut into k-slices whose slabs the combine launch sums and runs the epilogue on
    static const int mid_kslice = getenv("RSCOTR_BF16X6_MID_KSLICE") ? atoi(getenv("RSCOTR_BF16X6_MID_KSLICE")) : 256;  // shortest k-slice
    long sp = std::min<long>((512 + t64 - 1) / t64, p.K / mid_kslice);
    const int64_t per = ((int64_t)p.M * p.N + p.M) * 4;
    sp = std::min<long>(sp, ws_bytes / per);
    if (sp >= 2) {
      int klen = (int)((p.K + sp - 1) / sp);
      const int kq = p.K % 32 == 0 ? 32 : 16;  // (k-slices of whole 32-k stages for the pipelined 64 x 64 kernel)
      klen = (klen + kq - 1) / kq * kq;
      c.bm = 64; c.klen = klen; c.splits = (p.K + klen - 1) / klen;
      if (c.splits == 1) c.klen = p.K;
    }
  }
  return c;
}

template <int BM, int PIPE, bool EDGE = false>
static void launch_split6(const GemmParams& p, int a_kmajor, int b_kmajor, unsigned nwg, hipStream_t s) {
  if (!a_kmajor && !b_kmajor) gemm_bf16x6_kernel<BM, BM, false, false, PIPE, EDGE><<<dim3(nwg), 256, 0, s>>>(p);
  else if (!a_kmajor) gemm_bf16x6_kernel<BM, BM, false, true, PIPE, EDGE><<<dim3(nwg), 256, 0, s>>>(p);
  else if (!b_kmajor) gemm_bf16x6_kernel<BM, BM, true, false, PIPE, EDGE><<<dim3(nwg), 256, 0, s>>>(p);
  else gemm_bf16x6_kernel<BM, BM, true, true, PIPE, EDGE><<<dim3(nwg), 256, 0, s>>>(p);
}

template <int BM, int PIPE, bool EDGE = false>
static void launch_h3(const GemmParams& p, int a_kmajor, int b_kmajor, unsigned nwg, hipStream_t s) {
  if constexpr (BM == 128 && PIPE == 0) {
    if (!a_kmajor && !b_kmajor) gemm_h3_128_kernel<false, false, EDGE><<<dim3(nwg), 256, 0, s>>>(p);
    else if (!a_kmajor) gemm_h3_128_kernel<false, true, EDGE><<<dim3(nwg), 256, 0, s>>>(p);
    else if (!b_kmajor) gemm_h3_128_kernel<true, false, EDGE><<<dim3(nwg), 256, 0, s>>>(p);
    else gemm_h3_128_kernel<true, true, EDGE><<<dim3(nwg), 256, 0, s>>>(p);
  } else {
    // (the interior pipelined 64 x 64 kernels with a row-major A: the 128-register instantiations, four workgroups per CU —
    // 33.84 against 34.00 ms per round, bit-identical results)
    if (BM == 64 && PIPE == 2 && !EDGE && !a_kmajor && !b_kmajor) gemm_h3_kernel<64, 64, false, false, 2, false, 4><<<dim3(nwg), 256, 0, s>>>(p);
    else if (BM == 64 && PIPE == 2 && !EDGE && !a_kmajor && b_kmajor) gemm_h3_kernel<64, 64, false, true, 2, false, 4><<<dim3(nwg), 256, 0, s>>>(p);
    else if (!a_kmajor && !b_kmajor) gemm_h3_kernel<BM, BM, false, false, PIPE, EDGE><<<dim3(nwg), 256, 0, s>>>(p);
    else if (!a_kmajor) gemm_h3_kernel<BM, BM, false, true, PIPE, EDGE><<<dim3(nwg), 256, 0, s>>>(p);
    else if (!b_kmajor) gemm_h3_kernel<BM, BM, true, false, PIPE, EDGE><<<dim3(nwg), 256, 0, s>>>(p);
    else gemm_h3_kernel<BM, BM, true, true, PIPE, EDGE><<<dim3(nwg), 256, 0, s>>>(p);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Low-latency kernel for the decoders' small products (M x N <= ~1M outputs, K <= 512: the per-layer Linears and
// weight gradients of the DINO / Mask2Former decoders, a few hundred launches per round).  On such shapes the tiled
// kernel above is a chain of dependent memory round trips (k-tile -> LDS -> barrier, 4-16 times) on a fraction of the
// CUs: 11-18 us per launch for microseconds of MFMA work.  Here
//   * the output tile is 32 x 32 (4x the workgroups of a 64 x 64 tiling: M = 200 -> 56, M = 1600 -> 400);
//   * the NW wavefronts of a workgroup split K (each takes a contiguous run of 8-element "octets", <= 32 elements per
//     pass), so the reduction runs on all four SIMDs of the CU at once;
//   * MFMA operand fragments are loaded straight from global memory into registers, all loads of a pass in flight
//     together: ONE memory round trip per pass, no LDS staging, no barrier in the k loop.  A row-major operand is read
//     as float4 = 4 consecutive k per lane (lane half h takes k = 8*octet + 4*h + j for MFMA j: the k order inside an
//     octet is permuted identically for both operands, which a contraction does not see); a k-major operand as
//     128-byte coalesced rows;
//   * partial accumulators meet in LDS in fixed order (deterministic); wavefront q < 4 finishes rows 8q..8q+3 (+4h)
//     of the tile through the staged epilogue.
// Requires K % 8 == 0 and 16-byte loads legal on row-major operands (host-checked); rows past M / N are clamped reads
// whose results are never stored.
template <bool AK, bool BKM, int NW>
__global__ __launch_bounds__(64 * NW) void gemm_small_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) float gemm_smem[];  // [NW][16][64] partial accumulators
  __shared__ float s_rs[NW][32];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int fr = lane & 31, h = lane >> 5;
  const int tiles_n = (p.N + 31) >> 5;
  const int tile = blockIdx.x;
  const int m0 = (tile / tiles_n) * 32, n0 = (tile % tiles_n) * 32;
  const int ar = min(m0 + fr, p.M - 1), br = min(n0 + fr, p.N - 1);
  const int no = p.K >> 3;
  const int o0 = (int)((long)w * no / NW), o1 = (int)((long)(w + 1) * no / NW);
  const bool do_rs = AK && p.rowsum && n0 == 0;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float rs = 0.f;
  const float* Ab = AK ? p.A + ar : p.A + (long)ar * p.lda;
  const float* Bb = BKM ? p.B + br : p.B + (long)br * p.ldb;
  for (int oc = o0; oc < o1; oc += 4) {
    const int nt = min(4, o1 - oc);  // wave-uniform
    float a[4][4], b[4][4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < nt) {
        const int k = (oc + t) * 8 + h * 4;
        if (AK) {
#pragma unroll
          for (int j = 0; j < 4; ++j) a[t][j] = Ab[(long)(k + j) * p.lda];
        } else {
          const float4 q = *reinterpret_cast<const float4*>(Ab + k);
          a[t][0] = q.x; a[t][1] = q.y; a[t][2] = q.z; a[t][3] = q.w;
        }
        if (BKM) {
#pragma unroll
          for (int j = 0; j < 4; ++j) b[t][j] = Bb[(long)(k + j) * p.ldb];
        } else {
          const float4 q = *reinterpret_cast<const float4*>(Bb + k);
          b[t][0] = q.x; b[t][1] = q.y; b[t][2] = q.z; b[t][3] = q.w;
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      if (t < nt) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (AK && do_rs) rs += a[t][j];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][j], b[t][j], acc, 0, 0, 0);
        }
      }
    }
  }

  // partial accumulators -> LDS ([wave][register][lane]: conflict-free), fixed-order sum by wavefronts 0..3
#pragma unroll
  for (int r = 0; r < 16; ++r) gemm_smem[(w * 16 + r) * 64 + lane] = acc[r];
  if (AK && do_rs) {
    rs += __shfl_xor(rs, 32, 64);
    if (h == 0) s_rs[w][fr] = rs;
  }
  __syncthreads();
  if (AK && do_rs && threadIdx.x < 32) {
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < NW; ++g) v += s_rs[g][threadIdx.x];
    const int m = m0 + threadIdx.x;
    if (m < p.M) p.rowsum[m] = p.rowsum_acc ? p.rowsum[m] + v : v;
  }
  if (w >= 4) return;
  const int n = n0 + fr;
  float amx = 0.f;
  if (n < p.N) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float t = 0.f;
#pragma unroll
      for (int g = 0; g < NW; ++g) t += gemm_smem[(g * 16 + 4 * w + u) * 64 + lane];
      v[u] = t;
    }
    if (p.bias) {
      const float bv = p.bias[n];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] += bv;
    }
    epilogue_rows4<true>(p, v, m0 + 8 * w + 4 * h, n, amx);
  }
  amax_commit(p.amax_out, amx);
}

// ---------------------------------------------------------------------------------------------------------------
// Weight-gradient contractions with a SMALL output and a LONG reduction (dW = dY^T X over thousands of tokens: Swin
// stage 1-2 Linears, the 256-wide projections of the encoder): both operands k-major, so an MFMA fragment of k row
// `k` is a 128-byte coalesced load — no transposition, hence no LDS staging.  One wavefront owns a (32 TM) x (32 TN)
// block of the output (TM x TN accumulator tiles in AGPRs: TM + TN fragment loads feed TM * TN MFMAs per k pair, 3x3:
// 6 loads per 9 MFMAs) and streams its k range from global memory through two register buffers of 4 k pairs (the
// loads of block i+1 are in flight under the MFMAs of block i).  The 4 wavefronts of a workgroup take quarters of
// the workgroup's k slice and fold their accumulators through one LDS buffer in fixed order (3 -> 2 -> 1 -> 0), then
// the slice's slab is written for the split-K combine (deterministic).  Against the 64x64-tile kernel on
// M = 288, N = 96, K = 32768: no tile padding (320 x 128 -> 288 x 96), 75 MB instead of 167 MB of L2 -> CU operand
// traffic, no barrier in the k loop.
template <int TM, int TN, bool KS>
__global__ __launch_bounds__(256, 2) void gemm_dw_direct_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) float gemm_smem[];  // [TM*TN*16][64] + [TM][32]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int fr = lane & 31, h = lane >> 5;
  const int tn = (p.N + 32 * TN - 1) / (32 * TN);
  const int tile = blockIdx.x % p.tiles, split = blockIdx.x / p.tiles;
  const int m0 = (tile / tn) * 32 * TM, n0 = (tile % tn) * 32 * TN;
  const int ks0 = split * p.ksplit_len, ks1 = min(p.K, ks0 + p.ksplit_len);
  const int q = (((ks1 - ks0 + 3) >> 2) + 1) & ~1;  // even quarter
  const int k0 = min(ks1, ks0 + w * q), k1 = min(ks1, k0 + q);
  int am[TM], bn[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) am[i] = min(m0 + 32 * i + fr, p.M - 1);  // clamped reads; those rows are never stored
#pragma unroll
  for (int j = 0; j < TN; ++j) bn[j] = min(n0 + 32 * j + fr, p.N - 1);
  const bool do_rs = p.rowsum && n0 == 0;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float rs[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) rs[i] = 0.f;

  // a fragment buffer: 4 k pairs of A / B fragments + the factor of each pair's A rows (0 past the k range, else the
  // per-sample scale).  Nothing is USED at load time, so no wait is placed between the loads; no predicated loads
  // either (they compile to divergent blocks with a wait after each): k past the range reads the last valid row.
  auto load = [&](int kb, float (&a)[4][TM], float (&b)[4][TN], float (&f)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = kb + 2 * u + h;
      const int kc = min(k, p.K - 1);
      const float* ap = p.A + (long)kc * p.lda;
      const float* bp = p.B + (long)kc * p.ldb;
      f[u] = KS ? p.kscale[kc / p.krows_per] : 1.f;
      if (k >= k1) f[u] = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i) a[u][i] = ap[am[i]];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[u][j] = bp[bn[j]];
    }
  };
  auto compute = [&](const float (&a)[4][TM], const float (&b)[4][TN], const float (&f)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const float av = a[u][i] * f[u];
        rs[i] += av;
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[u][j], acc[i][j], 0, 0, 0);
      }
    }
  };
  float a0[4][TM], b0[4][TN], f0[4], a1[4][TM], b1[4][TN], f1[4];
  load(k0, a0, b0, f0);
  for (int kb = k0; kb < k1; kb += 16) {
    load(kb + 8, a1, b1, f1);
    compute(a0, b0, f0);
    load(kb + 16, a0, b0, f0);
    compute(a1, b1, f1);
  }

  // fold the 4 wavefronts' accumulators through LDS: 3 -> 2 -> 1 -> 0 (fixed order)
  float* s_rs = gemm_smem + TM * TN * 16 * 64;
#pragma unroll
  for (int i = 0; i < TM; ++i) rs[i] += __shfl_xor(rs[i], 32, 64);
  for (int g = 3; g >= 1; --g) {
    if (w == g) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) gemm_smem[(((i * TN + j) * 16) + r) * 64 + lane] = acc[i][j][r];
      if (h == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i) s_rs[i * 32 + fr] = rs[i];
      }
    }
    __syncthreads();
    if (w == g - 1) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] += gemm_smem[(((i * TN + j) * 16) + r) * 64 + lane];
#pragma unroll
      for (int i = 0; i < TM; ++i) rs[i] += s_rs[i * 32 + fr];
    }
    __syncthreads();
  }
  if (w != 0) return;
  float* slab = p.slabs + (long)split * p.M * p.N;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + 32 * j + fr;
      if (n >= p.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < p.M) slab[(long)m * p.N + n] = acc[i][j][r];
      }
    }
  if (do_rs && h == 0) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + 32 * i + fr;
      if (m < p.M) p.rs_slabs[(long)split * p.M + m] = rs[i];
    }
  }
}

// Combine split-K slabs (fixed order: deterministic) and apply the epilogue; also the row-sum partials.
// VEC: N % 4 == 0 and 16-byte aligned slabs -> one float4 of one output row per thread per step.
template <bool VEC>
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(GemmParams p) {
  const long total = (long)p.M * p.N;
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  float amx = 0.f;
  if (VEC) {
    const long total4 = total >> 2;
    const float4* sl = reinterpret_cast<const float4*>(p.slabs);
    for (long i = gid; i < total4; i += (long)gridDim.x * 256) {
      float4 v = sl[i];
#pragma unroll 8
      for (int s = 1; s < p.splits; ++s) {
        const float4 t = sl[(long)s * total4 + i];
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      const long e = i << 2;
      const int m = (int)(e / p.N), n = (int)(e - (long)m * p.N);
      if (p.vecC) {
        epilogue_store4(p, v, m, n, amx);
      } else {
        float* c = p.C + (long)m * p.ldc + n;
        c[0] = epilogue_one(p, v.x, m, n, amx);
        c[1] = epilogue_one(p, v.y, m, n + 1, amx);
        c[2] = epilogue_one(p, v.z, m, n + 2, amx);
        c[3] = epilogue_one(p, v.w, m, n + 3, amx);
      }
    }
  } else {
    for (long i = gid; i < total; i += (long)gridDim.x * 256) {
      float v = 0.f;
#pragma unroll 8
      for (int s = 0; s < p.splits; ++s) v += p.slabs[(long)s * total + i];
      const int m = (int)(i / p.N), n = (int)(i - (long)m * p.N);
      p.C[(long)m * p.ldc + n] = epilogue_one(p, v, m, n, amx);
    }
  }
  if (p.rowsum && gid < p.M) {
    float v = 0.f;
    for (int s = 0; s < p.splits; ++s) v += p.rs_slabs[(long)s * p.M + gid];
    p.rowsum[gid] = p.rowsum_acc ? p.rowsum[gid] + v : v;
  }
  amax_commit(p.amax_out, amx);
}

// Same combine for SMALL outputs cut into many slabs (the 256x256 weight gradients of the encoder / decoder
// projections: 16 tiles x ~31 slabs): one thread per output float4 leaves 64 workgroups walking 31 dependent-latency
// loads each (8 us for 8 MB).  Here the 4 wavefronts of a workgroup share 64 output float4s and take the slabs
// round-robin (wave g: slabs g, g+4, ...), partial sums meet in LDS in fixed order (deterministic); 4x the
// workgroups, a quarter of the loads per thread, every load a 1 KB wave-contiguous segment.
__global__ __launch_bounds__(256) void gemm_splitk_reduce_sg_kernel(GemmParams p) {
  __shared__ float4 red[3][64];
  const long total4 = ((long)p.M * p.N) >> 2;
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const long i = (long)blockIdx.x * 64 + lane;
  const float4* sl = reinterpret_cast<const float4*>(p.slabs);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  float amx = 0.f;
  if (i < total4) {
#pragma unroll 8
    for (int s = g; s < p.splits; s += 4) {
      const float4 t = sl[(long)s * total4 + i];
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
  }
  if (g > 0) red[g - 1][lane] = v;
  __syncthreads();
  if (g == 0 && i < total4) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float4 t = red[k][lane];
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    const long e = i << 2;
    const int m = (int)(e / p.N), n = (int)(e - (long)m * p.N);
    if (p.vecC) {
      epilogue_store4(p, v, m, n, amx);
    } else {
      float* c = p.C + (long)m * p.ldc + n;
      c[0] = epilogue_one(p, v.x, m, n, amx);
      c[1] = epilogue_one(p, v.y, m, n + 1, amx);
      c[2] = epilogue_one(p, v.z, m, n + 2, amx);
      c[3] = epilogue_one(p, v.w, m, n + 3, amx);
    }
  }
  const long gid = (long)blockIdx.x * 256 + threadIdx.x;
  if (p.rowsum && gid < p.M) {
    float r = 0.f;
    for (int s = 0; s < p.splits; ++s) r += p.rs_slabs[(long)s * p.M + gid];
    p.rowsum[gid] = p.rowsum_acc ? p.rowsum[gid] + r : r;
  }
  amax_commit(p.amax_out, amx);
}

// Column sums of a row-major (M, N) matrix: out[n] = sum_m X[m, n]  (bias gradients).
// Stage 1: grid (ceil(N/256), GY): a workgroup reduces a (rows x 256 columns) slab with 16-byte loads
// (wave w takes rows r0+w, r0+w+4, ...), folds its 4 wavefronts through LDS and stores one partial
// row.  Stage 2 sums the GY partial rows.  No atomics: same-address fp32 atomics from hundreds of
// workgroups serialise at the memory side.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ X, float* __restrict__ part,
                                                             int M, int N, int ld, int rows_per_block, int vec) {
  __shared__ float4 red[4][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(M, r0 + rows_per_block);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < N) {
    if (vec && c + 3 < N) {
#pragma unroll 4
      for (int r = r0 + w; r < r1; r += 4) {
        const float4 v = *reinterpret_cast<const float4*>(X + (long)r * ld + c);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
    } else {
      for (int r = r0 + w; r < r1; r += 4) {
        const float* src = X + (long)r * ld + c;
        acc.x += src[0];
        if (c + 1 < N) acc.y += src[1];
        if (c + 2 < N) acc.z += src[2];
        if (c + 3 < N) acc.w += src[3];
      }
    }
  }
  red[w][lane] = acc;
  __syncthreads();
  if (w == 0 && c < N) {
    float4 t = red[0][lane];
#pragma unroll
    for (int i = 1; i < 4; ++i) {
      t.x += red[i][lane].x; t.y += red[i][lane].y; t.z += red[i][lane].z; t.w += red[i][lane].w;
    }
    float* dst = part + (long)blockIdx.y * N + c;
    dst[0] = t.x;
    if (c + 1 < N) dst[1] = t.y;
    if (c + 2 < N) dst[2] = t.z;
    if (c + 3 < N) dst[3] = t.w;
  }
}

// out[n] (+)= sum_g part[g][n]
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                           int G, int N, int accumulate) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float s = accumulate ? out[n] : 0.f;
#pragma unroll 8
  for (int g = 0; g < G; ++g) s += part[(long)g * N + n];
  out[n] = s;
}

static int colsum_gy(int M, int N) {
  const int gx = (N + 255) / 256;
  int gy = std::max(1, std::min((M + 63) / 64, std::max(1, 512 / gx)));
  return std::min(gy, 256);
}

template <int BM, int BN, int KG>
constexpr size_t gemm_lds_bytes() {
  return sizeof(float) * std::max<size_t>((size_t)KG * (2 * GEMM_BK * (BM + 4 + BN + 4)), KG > 1 ? (size_t)(KG - 1) * 16 * 256 : 0);
}

// 0: fp32 matrix pipe (v_mfma_f32_32x32x2_f32) everywhere; 3: the split product (three bf16 planes, six MFMAs: fp32-accurate —
// or, with the operands' value ranges, two fp16 planes and three MFMAs) where it pays, fp32 pipe elsewhere.
// RSCOTR_GEMM_PREC=fp32|bf16x6 sets the start value, rscotr_gemm_set_precision() changes it (tests, A/B runs).  (Modes 1 / 2,
// round 1's two-plane bf16 product at 4-6e-6, lost their A/B in rounds 2 and 3 and left the library in round 5.)
static std::atomic<int> g_gemm_prec{[] {
  const char* e = getenv("RSCOTR_GEMM_PREC");
  if (e && (!strcmp(e, "fp32") || !strcmp(e, "0"))) return 0;
  if (e && (!strcmp(e, "bf16x6") || !strcmp(e, "3"))) return 3;
  return RSCOTR_GEMM_PREC_DEFAULT;
}()};

template <typename Kern>
static void launch_kernel(Kern kern, dim3 grid, int threads, size_t lds, hipStream_t s, const GemmParams& p) {
  if (lds > 48 * 1024) {  // opt in to more than the default dynamic LDS once per kernel (all instantiations share
    // this function: the template parameter is the pointer TYPE, so remember the pointers themselves)
    static std::mutex mu;
    static std::set<const void*> raised;
    std::lock_guard<std::mutex> lock(mu);
    if (raised.insert(reinterpret_cast<const void*>(kern)).second)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  }
  kern<<<grid, threads, lds, s>>>(p);
}

template <int BM, int BN, int WM, int WN, bool EDGE, int KG>
static void launch_gemm_edge(const GemmParams& p, int a_kmajor, int b_kmajor, dim3 grid, hipStream_t s) {
  constexpr size_t lds = gemm_lds_bytes<BM, BN, KG>();
  if (!a_kmajor && !b_kmajor)
    launch_kernel(gemm_f32_kernel<BM, BN, WM, WN, false, false, EDGE, KG>, grid, 256 * KG, lds, s, p);
  else if (!a_kmajor && b_kmajor)
    launch_kernel(gemm_f32_kernel<BM, BN, WM, WN, false, true, EDGE, KG>, grid, 256 * KG, lds, s, p);
  else if (a_kmajor && !b_kmajor)
    launch_kernel(gemm_f32_kernel<BM, BN, WM, WN, true, false, EDGE, KG>, grid, 256 * KG, lds, s, p);
  else
    launch_kernel(gemm_f32_kernel<BM, BN, WM, WN, true, true, EDGE, KG>, grid, 256 * KG, lds, s, p);
}

// kgroups: wavefront groups per workgroup sharing the k loop (1, 2 or 4; > 1 only for the one-tile-per-wavefront
// configurations and never together with rowsum)
template <int BM, int BN, int WM, int WN>
static void launch_gemm_cfg(const GemmParams& p, int a_kmajor, int b_kmajor, dim3 grid, hipStream_t s, int kgroups = 1) {
  const bool interior = p.M % BM == 0 && p.N % BN == 0 && p.K % GEMM_BK == 0 && p.ksplit_len % GEMM_BK == 0 &&
                        p.vecA && p.vecB;
  constexpr bool one_tile = (BM / WM == 32) && (BN / WN == 32);
  if (one_tile && kgroups == 4) {
    if (interior) launch_gemm_edge<BM, BN, WM, WN, false, one_tile ? 4 : 1>(p, a_kmajor, b_kmajor, grid, s);
    else launch_gemm_edge<BM, BN, WM, WN, true, one_tile ? 4 : 1>(p, a_kmajor, b_kmajor, grid, s);
  } else if (one_tile && kgroups == 2) {
    if (interior) launch_gemm_edge<BM, BN, WM, WN, false, one_tile ? 2 : 1>(p, a_kmajor, b_kmajor, grid, s);
    else launch_gemm_edge<BM, BN, WM, WN, true, one_tile ? 2 : 1>(p, a_kmajor, b_kmajor, grid, s);
  } else {
    if (interior) launch_gemm_edge<BM, BN, WM, WN, false, 1>(p, a_kmajor, b_kmajor, grid, s);
    else launch_gemm_edge<BM, BN, WM, WN, true, 1>(p, a_kmajor, b_kmajor, grid, s);
  }
}

// Wavefront groups per workgroup for a launch of `wgs` workgroups with `nk` k-tiles each: short grids leave
// most CUs idle, so the k loop of each tile is spread over 2 or 4 groups.
static int choose_kgroups(long wgs, long nk, bool has_rowsum) {
  static const char* force = getenv("RSCOTR_GEMM_KGROUPS");
  static const int rs_ok = getenv("RSCOTR_GEMM_KG_ROWSUM") ? atoi(getenv("RSCOTR_GEMM_KG_ROWSUM")) : 1;
  if (has_rowsum && !rs_ok) return 1;
  if (force) return atoi(force) == 4 ? 4 : (atoi(force) == 2 ? 2 : 1);
  static const long kg4_max = getenv("RSCOTR_GEMM_KG4_MAX") ? atol(getenv("RSCOTR_GEMM_KG4_MAX")) : 256;
  static const long kg2_max = getenv("RSCOTR_GEMM_KG2_MAX") ? atol(getenv("RSCOTR_GEMM_KG2_MAX")) : 768;
  if (wgs <= kg4_max && nk >= 8) return 4;
  if (wgs <= kg2_max && nk >= 4) return 2;
  return 1;
}

// The low-latency kernel's domain: small outputs, short reductions, no per-sample scaling (those are Swin products).
static bool small_gemm_ok(const GemmParams& p, int a_kmajor, int b_kmajor, long nbatch) {
  static const int on = getenv("RSCOTR_GEMM_SMALL") ? atoi(getenv("RSCOTR_GEMM_SMALL")) : 1;
  static const long max_tiles = getenv("RSCOTR_GEMM_SMALL_TILES") ? atol(getenv("RSCOTR_GEMM_SMALL_TILES")) : 512;
  static const int max_k = getenv("RSCOTR_GEMM_SMALL_K") ? atoi(getenv("RSCOTR_GEMM_SMALL_K")) : 512;
  if (!on || p.K % 8 || p.K < 32 || p.rowscale || p.kscale) return false;
  if ((!a_kmajor && !p.vecA) || (!b_kmajor && !p.vecB)) return false;
  const long tiles = (long)((p.M + 31) / 32) * ((p.N + 31) / 32);
  // a handful of output tiles with a longer reduction (the classifier's fc: 2 x 45 x 768) ran as ONE workgroup of the tiled
  // kernel walking 48 dependent k-tiles (26 us); here 16 wavefronts split K
  if (p.K > max_k) return p.K <= 4096 && tiles * nbatch <= 8;
  return tiles * nbatch <= max_tiles;
}

template <int NW>
static void launch_small_nw(const GemmParams& p, int a_kmajor, int b_kmajor, dim3 grid, hipStream_t s) {
  constexpr size_t lds = (size_t)NW * 16 * 64 * sizeof(float);
  if (!a_kmajor && !b_kmajor) launch_kernel(gemm_small_kernel<false, false, NW>, grid, 64 * NW, lds, s, p);
  else if (!a_kmajor && b_kmajor) launch_kernel(gemm_small_kernel<false, true, NW>, grid, 64 * NW, lds, s, p);
  else if (a_kmajor && !b_kmajor) launch_kernel(gemm_small_kernel<true, false, NW>, grid, 64 * NW, lds, s, p);
  else launch_kernel(gemm_small_kernel<true, true, NW>, grid, 64 * NW, lds, s, p);
}

static void launch_small(const GemmParams& p, int a_kmajor, int b_kmajor, unsigned nbatch, hipStream_t s) {
  const int no = p.K / 8;
  const dim3 grid((unsigned)(((p.M + 31) / 32) * ((p.N + 31) / 32)), nbatch, 1);
  // <= 4 octets (one pass) per wavefront where 16 wavefronts allow it
  if (no > 32) launch_small_nw<16>(p, a_kmajor, b_kmajor, grid, s);
  else if (no > 16) launch_small_nw<8>(p, a_kmajor, b_kmajor, grid, s);
  else launch_small_nw<4>(p, a_kmajor, b_kmajor, grid, s);
}

// Wave-tile and split choice of the direct weight-gradient kernel; splits == 0: not its domain.
struct DwCfg {
  int TM, TN, tiles;
  long splits;
  int klen;
};

static DwCfg choose_dw_direct(int M, int N, int K) {
  static const int on = getenv("RSCOTR_GEMM_DW_DIRECT") ? atoi(getenv("RSCOTR_GEMM_DW_DIRECT")) : 1;
  // Measured on the step (profiles/README.md, trip 37): wins where the 64x64 tiling pads badly and the reduction is
  // very long (Swin stage 1: 288x96, 96x384, 384x96 over 32768 tokens: 104 / 80 / 77 us -> 67 / 68 / 69 us); loses on
  // the 256-wide and stage 2-3 gradients (one 24-load block in flight per wavefront is latency-bound below ~K = 16k).
  static const int max_tiles = getenv("RSCOTR_GEMM_DW_TILES") ? atoi(getenv("RSCOTR_GEMM_DW_TILES")) : 4;
  static const int min_k = getenv("RSCOTR_GEMM_DW_MINK") ? atoi(getenv("RSCOTR_GEMM_DW_MINK")) : 16384;
  static const long target = getenv("RSCOTR_GEMM_DW_TARGET") ? atol(getenv("RSCOTR_GEMM_DW_TARGET")) : 256;
  DwCfg c{0, 0, 0, 0, 0};
  if (!on || K < min_k || M < 8 || N < 8) return c;
  // least padded of 96x96, 64x128, 128x64 wave tiles
  const int cand[3][2] = {{3, 3}, {2, 4}, {4, 2}};
  long best = -1;
  for (const auto& t : cand) {
    const long tm = (M + 32 * t[0] - 1) / (32 * t[0]), tn = (N + 32 * t[1] - 1) / (32 * t[1]);
    const long area = tm * 32 * t[0] * tn * 32 * t[1];
    if (best < 0 || area < best) { best = area; c.TM = t[0]; c.TN = t[1]; c.tiles = (int)(tm * tn); }
  }
  if (c.tiles > max_tiles || c.tiles < 3) return c;
  // ~`target` workgroups of 4 wavefronts (one per SIMD of a CU), >= 128 k per workgroup, <= 128 slabs
  long sp = std::min<long>({(target + c.tiles - 1) / c.tiles, (long)K / 128, 128L});
  if (sp < 2) return c;
  int klen = (int)((K + sp - 1) / sp);
  klen = (klen + 7) / 8 * 8;
  c.klen = klen;
  c.splits = (K + klen - 1) / klen;
  if (c.splits < 2) c.splits = 0;
  return c;
}

template <int TM, int TN>
static void launch_dw_direct(const GemmParams& p, dim3 grid, hipStream_t s) {
  constexpr size_t lds = (size_t)(TM * TN * 16 * 64 + TM * 32) * sizeof(float);
  if (p.kscale) launch_kernel(gemm_dw_direct_kernel<TM, TN, true>, grid, 256, lds, s, p);
  else launch_kernel(gemm_dw_direct_kernel<TM, TN, false>, grid, 256, lds, s, p);
}

// combine kernel for the slabs of p (p.splits > 1)
static void launch_splitk_reduce(const GemmParams& p, const float* workspace, hipStream_t s) {
  const int M = p.M, N = p.N;
  const long total = (long)M * N;
  const bool vec = (N % 4 == 0) && aligned16(workspace) && (total % 4 == 0);
  const long work = vec ? total / 4 : total;
  const int blocks = (int)std::min<long>((std::max<long>(work, M) + 255) / 256, 2048);
  ProfScope prof(PROF_HBM, (p.splits + 1.0 + (p.resid ? 1.0 : 0.0) + (p.aux ? 1.0 : 0.0) + (p.accumulate ? 1.0 : 0.0)) * 4.0 * total, s,
                 "rscotr::gemm_splitk_reduce_kernel");
  static const int sg_ok = getenv("RSCOTR_GEMM_REDUCE_SG") ? atoi(getenv("RSCOTR_GEMM_REDUCE_SG")) : 1;
  if (vec && sg_ok && p.splits >= 8 && blocks < 512 && (work + 63) / 64 * 256 >= M)
    gemm_splitk_reduce_sg_kernel<<<(unsigned)((work + 63) / 64), 256, 0, s>>>(p);
  else if (vec) gemm_splitk_reduce_kernel<true><<<blocks, 256, 0, s>>>(p);
  else gemm_splitk_reduce_kernel<false><<<blocks, 256, 0, s>>>(p);
}

void splitk_reduce_launch(const GemmParams& p, const float* workspace, hipStream_t s) { launch_splitk_reduce(p, workspace, s); }

}  // namespace rscotr

using namespace rscotr;

struct GemmCfg {
  int BM, BN;
  long splits;
};

static bool cfg_built(int BM, int BN) {
  return (BM == 64 && BN == 64) || (BM == 128 && (BN == 32 || BN == 64));
}

// Tile / split choice.  The grid must cover 256 CUs a few times over (3 workgroups per CU are
// resident): prefer the largest tile that still gives >= 512 tiles, then smaller tiles, then split K
// (>= 16 k-tiles per split) through caller-provided slabs.
static GemmCfg choose_cfg(int M, int N, int K) {
  static const char* force = getenv("RSCOTR_GEMM_FORCE");  // "BM,BN,splits" — tuning only
  GemmCfg c;
  if (force) {
    int bm = 0, bn = 0, sp = 0;
    if (sscanf(force, "%d,%d,%d", &bm, &bn, &sp) == 3 && cfg_built(bm, bn)) {
      c.BM = bm; c.BN = bn;
      c.splits = sp > 0 ? std::min<long>(sp, std::max(1, K / GEMM_BK)) : 1;
      return c;
    }
  }
  // Measured on MI355X over the step's shapes (scripts/tune_gemm.py, scripts/lab/gemm_lab.hip,
  // profiles/history/r1_gemm_tuning.md): 64x64 tiles (8 resident workgroups per CU) win on nearly every shape;
  // 128x64 wins on the wide, tall products of the encoder FFN (N >= 1024, >= 1360 tiles of 128x64).
  c.BM = 64; c.BN = 64;
  if (N <= 32) { c.BM = 128; c.BN = 32; }
  else if (N >= 1024 && M >= 4096 && M % 128 == 0) { c.BM = 128; c.BN = 64; }
  const long t = (long)((M + c.BM - 1) / c.BM) * ((N + c.BN - 1) / c.BN);
  // Split only long reductions on short grids: a 64x64 tile costs ~0.2 us per k-tile, so K < 1024
  // finishes in a few microseconds on however few CUs, cheaper than a second (combine) launch; longer
  // K is cut (>= 256 elements per slice) until the grid holds ~2 workgroups per CU (each then runs 2 k-groups:
  // 512 workgroups x 2 groups measured equal to 1024 x 1 with half the slab traffic).
  c.splits = 1;
  static const long split_target = getenv("RSCOTR_GEMM_SPLIT_TARGET") ? atol(getenv("RSCOTR_GEMM_SPLIT_TARGET")) : 512;
  if (t < 512 && K >= 1024) {
    long sp = (split_target + t - 1) / t;
    sp = std::min<long>(sp, K / 256);
    c.splits = std::max<long>(1, std::min<long>(sp, 64));
  }
  return c;
}

// Deferred combine (rscotr_gemm_f32_dw_slabs): the split-K launch stops after writing its slabs and reports the split
// count; rscotr_splitk_flush later combines every pending problem of a backward pass in ONE launch.
static thread_local int tl_defer = 0;
static thread_local int tl_last_splits = 1;

// Workspace the split-K path wants for this problem (bytes; 0 = never splits): slabs + row-sum partials.
extern "C" int rscotr_gemm_set_precision(int prec) {
  if (prec != 0 && prec != 3) return fail(RSCOTR_E_ARG, "rscotr_gemm_set_precision: 0 (fp32 matrix pipe everywhere) or 3 (the fp32-accurate split product where it pays)");
  g_gemm_prec.store(prec);
  return RSCOTR_OK;
}

extern "C" int rscotr_gemm_get_precision(void) { return g_gemm_prec.load(); }

extern "C" int64_t rscotr_gemm_f32_workspace(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const GemmCfg c = choose_cfg(M, N, K);
  const DwCfg d = choose_dw_direct(M, N, K);
  int64_t sp = std::max<int64_t>(c.splits > 1 ? c.splits : 0, d.splits);
  if (g_gemm_prec.load(std::memory_order_relaxed) == 3 && M >= 64 && N >= 64 && K % 4 == 0 && K >= 1024) {
    const long t128 = (M % 128 == 0 && N % 128 == 0) ? (long)(M / 128) * (N / 128) : 0;
    const long tiles = t128 >= 16 ? t128 : (long)((M + 63) / 64) * ((N + 63) / 64);
    sp = std::max<int64_t>(sp, std::max<long>(1, std::min<long>((512 + tiles - 1) / tiles, K / 256)));  // as a weight gradient
    const long t64 = (long)((M + 63) / 64) * ((N + 63) / 64);
    if (t64 >= 128 && t64 < 512) sp = std::max<int64_t>(sp, std::min<long>((512 + t64 - 1) / t64, K / 256));  // mid-size k-slices
  }
  return sp * ((int64_t)M * N + M) * 4;
}

static std::atomic<int> g_h3_on{[] {
  const char* e = getenv("RSCOTR_GEMM_H3");
  return e ? atoi(e) : 1;
}()};
extern "C" int rscotr_gemm_set_h3(int on) { return g_h3_on.exchange(on ? 1 : 0); }

static int gemm_f32_impl(const float* A, const float* B, float* C, int M, int N, int K, int lda,
                         int ldb, int ldc, int a_kmajor, int b_kmajor, const float* bias, int act,
                         const float* aux, float* pre, const float* resid, int accumulate,
                         float* rowsum, int rowsum_accumulate, const float* rowscale, int rows_per_scale,
                         const float* kscale, int krows_per_scale, float* out2, float* workspace,
                         int64_t workspace_bytes, void* stream, const uint32_t* amax_a, const uint32_t* amax_b,
                         uint32_t* amax_out, const void* b_planes = nullptr, int b_rpad = 0);

extern "C" int rscotr_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda,
                               int ldb, int ldc, int a_kmajor, int b_kmajor, const float* bias, int act,
                               const float* aux, float* pre, const float* resid, int accumulate,
                               float* rowsum, int rowsum_accumulate, const float* rowscale, int rows_per_scale,
                               const float* kscale, int krows_per_scale, float* out2, float* workspace,
                               int64_t workspace_bytes, void* stream) {
  return gemm_f32_impl(A, B, C, M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, bias, act, aux, pre, resid, accumulate, rowsum,
                       rowsum_accumulate, rowscale, rows_per_scale, kscale, krows_per_scale, out2, workspace, workspace_bytes,
                       stream, nullptr, nullptr, nullptr);
}

extern "C" int rscotr_gemm_f32_r(const float* A, const float* B, float* C, int M, int N, int K, int lda,
                                 int ldb, int ldc, int a_kmajor, int b_kmajor, const float* bias, int act,
                                 const float* aux, float* pre, const float* resid, int accumulate,
                                 float* rowsum, int rowsum_accumulate, const float* rowscale, int rows_per_scale,
                                 const float* kscale, int krows_per_scale, float* out2, float* workspace,
                                 int64_t workspace_bytes, const uint32_t* amax_a, const uint32_t* amax_b, uint32_t* amax_out,
                                 void* stream) {
  return gemm_f32_impl(A, B, C, M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, bias, act, aux, pre, resid, accumulate, rowsum,
                       rowsum_accumulate, rowscale, rows_per_scale, kscale, krows_per_scale, out2, workspace, workspace_bytes,
                       stream, amax_a, amax_b, amax_out);
}

// rscotr_gemm_f32_r with the B operand ALSO given as pre-split fp16 planes (rscotr_gemm_split_weights_h3: the planes of B for a
// row-major B, of its transpose for a k-major one; b_rpad = their padded row count).  Taken where rscotr_gemm_f32_split_route
// answers 2 (the interior pipelined 64 x 64 fp16 kernel); everywhere else the call is rscotr_gemm_f32_r on B itself.
extern "C" int rscotr_gemm_f32_rb(const float* A, const float* B, float* C, int M, int N, int K, int lda,
                                  int ldb, int ldc, int a_kmajor, int b_kmajor, const float* bias, int act,
                                  const float* aux, float* pre, const float* resid, int accumulate,
                                  float* rowsum, int rowsum_accumulate, const float* rowscale, int rows_per_scale,
                                  const float* kscale, int krows_per_scale, float* out2, float* workspace,
                                  int64_t workspace_bytes, const uint32_t* amax_a, const uint32_t* amax_b, uint32_t* amax_out,
                                  const void* b_planes, int b_rpad, void* stream) {
  if (b_planes && (b_rpad < N || b_rpad % 64 || ((uintptr_t)b_planes & 15)))
    return fail(RSCOTR_E_ARG, "rscotr_gemm_f32_rb: the plane set has %d rows for N = %d (a multiple of 64 >= N, 16-byte aligned)", b_rpad, N);
  return gemm_f32_impl(A, B, C, M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, bias, act, aux, pre, resid, accumulate, rowsum,
                       rowsum_accumulate, rowscale, rows_per_scale, kscale, krows_per_scale, out2, workspace, workspace_bytes,
                       stream, amax_a, amax_b, amax_out, b_planes, b_rpad);
}

// 1 if rscotr_gemm_f32 with these arguments (16-byte aligned operands, precision mode 3) runs on the interior 128 x 128
// split-product tiles with one k-slice: the products that may carry the ReLU gate as bits (ACT_RELU_BITS / ACT_RELU_GRAD_BITS)
extern "C" int rscotr_gemm_relu_bits_ok(int M, int N, int K, int lda, int ldb, int a_kmajor, int b_kmajor) {
  if (M <= 0 || N <= 0 || K <= 0 || a_kmajor || g_gemm_prec.load(std::memory_order_relaxed) != 3) return 0;
  if (M % 128 || N % 128 || K % RSCOTR_X6_BK0) return 0;
  GemmParams p{};
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = N;
  p.act = ACT_RELU;
  p.vecA = lda % 4 == 0; p.vecB = ldb % 4 == 0;
  if (small_gemm_ok(p, a_kmajor, b_kmajor, 1)) return 0;
  const Split6Cfg sc = choose_split6(p, a_kmajor, b_kmajor, 0);
  return sc.bm == 128 && sc.splits == 1;
}

static int gemm_f32_impl(const float* A, const float* B, float* C, int M, int N, int K, int lda,
                         int ldb, int ldc, int a_kmajor, int b_kmajor, const float* bias, int act,
                         const float* aux, float* pre, const float* resid, int accumulate,
                         float* rowsum, int rowsum_accumulate, const float* rowscale, int rows_per_scale,
                         const float* kscale, int krows_per_scale, float* out2, float* workspace,
                         int64_t workspace_bytes, void* stream, const uint32_t* amax_a, const uint32_t* amax_b,
                         uint32_t* amax_out, const void* b_planes, int b_rpad) {
  if (M < 0 || N < 0 || K < 0) return fail(RSCOTR_E_SHAPE, "rscotr_gemm_f32: negative dimension");
  if ((rowscale && rows_per_scale <= 0) || (kscale && (krows_per_scale <= 0 || !a_kmajor)))
    return fail(RSCOTR_E_ARG, "rscotr_gemm_f32: rowscale needs rows_per_scale > 0; kscale needs a k-major A and krows_per_scale > 0");
  if (M == 0 || N == 0) return RSCOTR_OK;
  if (!A || !B || !C) return fail(RSCOTR_E_ARG, "rscotr_gemm_f32: null pointer");
  if (act < ACT_NONE || act > ACT_RELU_GRAD_BITS) return fail(RSCOTR_E_ARG, "rscotr_gemm_f32: unknown act %d", act);
  if ((act == ACT_RELU_GRAD || act == ACT_GELU_GRAD || act == ACT_RELU_GRAD_BITS) && !aux)
    return fail(RSCOTR_E_ARG, "rscotr_gemm_f32: act %d needs aux", act);
  const bool relu_bits = act == ACT_RELU_BITS || act == ACT_RELU_GRAD_BITS;
  if (relu_bits) {
    // the one-bit ReLU gate lives in the interior 128 x 128 split-product tiles only (callers ask rscotr_gemm_relu_bits_ok first)
    if (act == ACT_RELU_BITS ? (!pre || ((uintptr_t)pre & 7)) : ((uintptr_t)aux & 7))
      return fail(RSCOTR_E_ARG, "rscotr_gemm_f32: act %d moves the gate bits through %s (8-byte aligned, M * N / 8 bytes)", act,
                  act == ACT_RELU_BITS ? "pre" : "aux");
    if ((act == ACT_RELU_GRAD_BITS && pre) || resid || accumulate || rowscale || out2 || rowsum || kscale)
      return fail(RSCOTR_E_ARG, "rscotr_gemm_f32: act %d takes bias only (no pre / resid / accumulate / rowscale / out2 / rowsum)", act);
    if (!rscotr_gemm_relu_bits_ok(M, N, K, lda, ldb, a_kmajor, b_kmajor))
      return fail(RSCOTR_E_SHAPE, "rscotr_gemm_f32: act %d on a product that does not take the interior 128 x 128 split-product tiles "
                  "(M = %d N = %d K = %d): rscotr_gemm_relu_bits_ok", act, M, N, K);
  }
  if (lda < (a_kmajor ? M : K) || ldb < (b_kmajor ? N : K) || ldc < N)
    return fail(RSCOTR_E_SHAPE, "rscotr_gemm_f32: leading dimension too small");
  if (rowsum && !a_kmajor) return fail(RSCOTR_E_ARG, "rscotr_gemm_f32: rowsum needs a k-major A");
  GemmParams p;
  p.A = A; p.B = B; p.C = C; p.bias = bias; p.aux = aux; p.pre = pre; p.resid = resid; p.C2 = out2;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.act = act; p.accumulate = accumulate;
  p.vecA = aligned16(A) && (lda % 4 == 0);
  p.vecB = aligned16(B) && (ldb % 4 == 0);
  p.vecC = (ldc % 4 == 0) && aligned16(C) && aligned16(bias) && aligned16(aux) && aligned16(pre) && aligned16(resid) && aligned16(out2);
  p.rowsum = rowsum; p.rowsum_acc = rowsum_accumulate;
  p.nb1 = 0; p.nb2 = 1;
  p.rowscale = rowscale; p.rows_per = rows_per_scale; p.kscale = kscale; p.krows_per = krows_per_scale;
  p.amax_out = amax_out;
  hipStream_t s = (hipStream_t)stream;

  if (small_gemm_ok(p, a_kmajor, b_kmajor, 1)) {
    p.ksplit_len = K; p.splits = 1; p.tiles = 0; p.slabs = nullptr; p.rs_slabs = nullptr;
    static const bool prof_shapes_s = getenv("RSCOTR_PROF_SHAPES") != nullptr;
    char sname[112];
    if (prof_shapes_s) snprintf(sname, sizeof(sname), "M=%d N=%d K=%d %d%d splits=0", M, N, K, a_kmajor, b_kmajor);
    else snprintf(sname, sizeof(sname), "rscotr::gemm_small_kernel<%s, %s, *>", a_kmajor ? "true" : "false", b_kmajor ? "true" : "false");
    ProfScope prof(PROF_GEMM, 2.0 * M * N * K, s, "%s", sname);
    launch_small(p, a_kmajor, b_kmajor, 1, s);
    return check_launch("rscotr_gemm_f32 (small)");
  }

  if (a_kmajor && b_kmajor && workspace && !rowscale) {
    const DwCfg d = choose_dw_direct(M, N, K);
    if (d.splits >= 2 && workspace_bytes >= d.splits * ((int64_t)M * N + M) * 4) {
      p.ksplit_len = d.klen; p.splits = (int)d.splits; p.tiles = d.tiles;
      p.slabs = workspace; p.rs_slabs = workspace + d.splits * (int64_t)M * N;
      static const bool prof_shapes_d = getenv("RSCOTR_PROF_SHAPES") != nullptr;
      char dname[112];
      if (prof_shapes_d) snprintf(dname, sizeof(dname), "M=%d N=%d K=%d 11 splits=-%d", M, N, K, (int)d.splits);
      else snprintf(dname, sizeof(dname), "rscotr::gemm_dw_direct_kernel<%d, %d>", d.TM, d.TN);
      ProfScope prof(PROF_GEMM, 2.0 * M * N * K, s, "%s", dname);
      const dim3 grid((unsigned)(d.tiles * d.splits), 1, 1);
      if (d.TM == 3) launch_dw_direct<3, 3>(p, grid, s);
      else if (d.TM == 2) launch_dw_direct<2, 4>(p, grid, s);
      else launch_dw_direct<4, 2>(p, grid, s);
      if (int e = check_launch("rscotr_gemm_f32 (dw direct)")) return e;
      if (tl_defer) { tl_last_splits = (int)d.splits; return RSCOTR_OK; }
      launch_splitk_reduce(p, workspace, s);
      return check_launch("rscotr_gemm_f32 (dw direct reduce)");
    }
  }

  const int prec_mode = g_gemm_prec.load(std::memory_order_relaxed);
  if (prec_mode == 3) {
    const Split6Cfg sc = choose_split6(p, a_kmajor, b_kmajor, workspace ? workspace_bytes : 0);
    if (sc.bm) {
      p.tiles = ((M + sc.bm - 1) / sc.bm) * ((N + sc.bm - 1) / sc.bm);
      const bool ragged = M % sc.bm || N % sc.bm || K % (sc.bm == 128 ? RSCOTR_X6_BK0 : 16);
      p.splits = sc.splits; p.ksplit_len = sc.klen;
      p.slabs = sc.splits > 1 ? workspace : nullptr;
      p.rs_slabs = sc.splits > 1 ? workspace + sc.splits * (int64_t)M * N : nullptr;
      static const bool prof_shapes_6 = getenv("RSCOTR_PROF_SHAPES") != nullptr;
      const bool h3 = amax_a && amax_b && g_h3_on.load(std::memory_order_relaxed);
      p.amax_a = amax_a; p.amax_b = amax_b;
      static const int pipelined = getenv("RSCOTR_BF16X6_PIPE") ? atoi(getenv("RSCOTR_BF16X6_PIPE")) : 1;  // bit 0: 64 x 64 (measured -0.45 ms / round), bit 1: 128 x 128 (measured slower on every layout of the step: +0.65 ms)
      // B from its pre-split planes (rscotr_gemm_f32_rb; row-major by construction): the interior pipelined 64 x 64 fp16 kernel only
      const bool use_planes = h3 && b_planes && !a_kmajor && !ragged && sc.bm == 64 && (pipelined & 1) && K % 32 == 0 && sc.klen % 32 == 0;
      if (use_planes) b_kmajor = 0;  // (what the kernel and its profile name see)
      char xname[112];
      if (prof_shapes_6) snprintf(xname, sizeof(xname), "M=%d N=%d K=%d %d%d %s-%d splits=%d", M, N, K, a_kmajor, b_kmajor, h3 ? "h3" : "bf16x6", sc.bm, sc.splits);
      else snprintf(xname, sizeof(xname), "rscotr::gemm_%s_kernel<%d, %d, %s, %s, *>", h3 ? "h3" : "bf16x6", sc.bm, sc.bm, a_kmajor ? "true" : "false", b_kmajor ? "true" : "false");
      ProfScope prof(PROF_GEMM, 2.0 * M * N * K, s, "%s", xname);
      const unsigned nwg = sc.splits > 1 ? (unsigned)(8 * ((p.tiles >> 3) + ((p.tiles & 7) ? 1 : 0)) * sc.splits) : (unsigned)p.tiles;
      if (h3) {  // the fp16 split product: same tiles, same loops
        if (ragged) {
          if (sc.bm == 128) launch_h3<128, 0, true>(p, a_kmajor, b_kmajor, nwg, s);
          else if (sc.klen % 32 == 0) launch_h3<64, 2, true>(p, a_kmajor, b_kmajor, nwg, s);
          else launch_h3<64, 1, true>(p, a_kmajor, b_kmajor, nwg, s);
        } else if (sc.bm == 128) {
          launch_h3<128, 0>(p, a_kmajor, b_kmajor, nwg, s);
        } else {
          if ((pipelined & 1) && K % 32 == 0 && sc.klen % 32 == 0) {
            if (use_planes) {  // no conversion of B in the loop
              p.B = reinterpret_cast<const float*>(b_planes);
              p.ldb = b_rpad;
              gemm_h3_kernel<64, 64, false, false, 2, false, 4, true><<<dim3(nwg), 256, 0, s>>>(p);
            } else {
              launch_h3<64, 2>(p, a_kmajor, b_kmajor, nwg, s);
            }
          } else {
            launch_h3<64, 1>(p, a_kmajor, b_kmajor, nwg, s);
          }
        }
      } else if (ragged) {  // (EDGE instantiations: the 128 x 128 one-stage loop and the pipelined 64 x 64 loop)
        if (sc.bm == 128) launch_split6<128, 0, true>(p, a_kmajor, b_kmajor, nwg, s);
        else if (sc.klen % 32 == 0) launch_split6<64, 2, true>(p, a_kmajor, b_kmajor, nwg, s);
        else launch_split6<64, 1, true>(p, a_kmajor, b_kmajor, nwg, s);
      } else if (sc.bm == 128) {
        if (pipelined & 2) launch_split6<128, 3>(p, a_kmajor, b_kmajor, nwg, s);
        else launch_split6<128, 0>(p, a_kmajor, b_kmajor, nwg, s);
      } else {
        if ((pipelined & 1) && K % 32 == 0 && sc.klen % 32 == 0) launch_split6<64, 2>(p, a_kmajor, b_kmajor, nwg, s);
        else launch_split6<64, 1>(p, a_kmajor, b_kmajor, nwg, s);
      }
      if (int e = check_launch("rscotr_gemm_f32 (bf16x6)")) return e;
      if (sc.splits > 1) {
        if (tl_defer) { tl_last_splits = sc.splits; return RSCOTR_OK; }
        launch_splitk_reduce(p, workspace, s);
        return check_launch("rscotr_gemm_f32 (bf16x6, split-K reduce)");
      }
      return RSCOTR_OK;
    }
  }
  const GemmCfg cfg = choose_cfg(M, N, K);
  const int BM = cfg.BM, BN = cfg.BN;
  const long tiles = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  long splits = 1;
  if (workspace) {
    splits = std::min<long>(cfg.splits, workspace_bytes / (((int64_t)M * N + M) * 4));
    if (splits < 1) splits = 1;
  }
  // a short grid with a moderately long reduction is better served by wavefront groups sharing the k loop
  // inside the workgroup (no slabs, no combine launch) than by a split across workgroups
  static const int kg_over_split = getenv("RSCOTR_GEMM_KG_OVER_SPLIT") ? atoi(getenv("RSCOTR_GEMM_KG_OVER_SPLIT")) : 0;
  if (kg_over_split && !rowsum && K < 4096 && tiles <= 160 && (BM == 64 || BN == 32)) splits = 1;
  int klen = K;
  if (splits > 1) {
    klen = (int)((K + splits - 1) / splits);
    klen = (klen + GEMM_BK - 1) / GEMM_BK * GEMM_BK;
    splits = (K + klen - 1) / klen;
  }
  p.ksplit_len = klen;
  p.splits = (int)splits;
  p.tiles = (int)tiles;
  p.slabs = splits > 1 ? workspace : nullptr;
  p.rs_slabs = splits > 1 ? workspace + splits * (int64_t)M * N : nullptr;
  dim3 grid((unsigned)tiles, 1, 1);
  if (splits > 1) grid.x = (unsigned)(8 * ((tiles >> 3) + ((tiles & 7) ? 1 : 0)) * splits);
  static const bool prof_shapes = getenv("RSCOTR_PROF_SHAPES") != nullptr;  // per-shape census instead of per-kernel
  char pname[112];
  if (prof_shapes)
    snprintf(pname, sizeof(pname), "M=%d N=%d K=%d %d%d splits=%d", M, N, K, a_kmajor, b_kmajor, (int)splits);
  else
    snprintf(pname, sizeof(pname), "rscotr::gemm_f32_kernel<%d, %d, %d, %d, %s, %s, *>", BM, BN,
             BM == 128 && BN == 32 ? 4 : 2, BM == 128 && BN == 32 ? 1 : 2, a_kmajor ? "true" : "false",
             b_kmajor ? "true" : "false");
  ProfScope prof(PROF_GEMM, 2.0 * M * N * K, s, "%s", pname);
  const int kg = choose_kgroups((long)grid.x, (klen + GEMM_BK - 1) / GEMM_BK, rowsum != nullptr);
  if (BM == 64) launch_gemm_cfg<64, 64, 2, 2>(p, a_kmajor, b_kmajor, grid, s, kg);
  else if (BN == 32) launch_gemm_cfg<128, 32, 4, 1>(p, a_kmajor, b_kmajor, grid, s, kg);
  else launch_gemm_cfg<128, 64, 2, 2>(p, a_kmajor, b_kmajor, grid, s);
  if (int e = check_launch("rscotr_gemm_f32")) return e;
  if (splits > 1) {
    if (tl_defer) { tl_last_splits = (int)splits; return RSCOTR_OK; }
    launch_splitk_reduce(p, workspace, s);
    return check_launch("rscotr_gemm_f32 (split-K reduce)");
  }
  return RSCOTR_OK;
}

// dW = A^T B into slabs only (both operands k-major, result to be ACCUMULATED into C / rowsum later): the launch of
// rscotr_gemm_f32(a_kmajor = b_kmajor = 1, accumulate = 1, rowsum_accumulate = 1) without its combine.  *splits_out = the
// number of slabs written ([splits][M][N] floats at slab_region, then [splits][M] row-sum partials); 1 = the problem
// was not split and C / rowsum already hold the final result.
extern "C" int rscotr_gemm_f32_dw_slabs_r(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb,
                                          int ldc, float* rowsum, const float* kscale, int krows_per_scale,
                                          float* slab_region, int64_t slab_bytes, int32_t* splits_out,
                                          const uint32_t* amax_a, const uint32_t* amax_b, void* stream) {
  if (!splits_out) return fail(RSCOTR_E_ARG, "rscotr_gemm_f32_dw_slabs: splits_out required");
  tl_defer = 1;
  tl_last_splits = 1;
  const int e = gemm_f32_impl(A, B, C, M, N, K, lda, ldb, ldc, 1, 1, nullptr, ACT_NONE, nullptr, nullptr, nullptr, 1, rowsum,
                              1, nullptr, 0, kscale, krows_per_scale, nullptr, slab_region, slab_bytes, stream, amax_a, amax_b,
                              nullptr);
  tl_defer = 0;
  *splits_out = tl_last_splits;
  return e;
}

extern "C" int rscotr_gemm_f32_dw_slabs(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb,
                                        int ldc, float* rowsum, const float* kscale, int krows_per_scale,
                                        float* slab_region, int64_t slab_bytes, int32_t* splits_out, void* stream) {
  return rscotr_gemm_f32_dw_slabs_r(A, B, C, M, N, K, lda, ldb, ldc, rowsum, kscale, krows_per_scale, slab_region, slab_bytes,
                                    splits_out, nullptr, nullptr, stream);
}

// 1 if rscotr_gemm_f32 with these arguments (16-byte aligned operands assumed) runs on the split-product kernels, i.e. as the
// fp16 split product when the value ranges of both operands are supplied (rscotr_gemm_f32_r): callers ask before they
// go looking for ranges.
extern "C" int rscotr_gemm_f32_split_route(int M, int N, int K, int lda, int ldb, int a_kmajor, int b_kmajor, int act,
                                           int has_pre, int has_rowscale, int has_kscale, int64_t workspace_bytes) {
  if (M <= 0 || N <= 0 || K <= 0 || g_gemm_prec.load(std::memory_order_relaxed) != 3 || !g_h3_on.load(std::memory_order_relaxed)) return 0;
  GemmParams p{};
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = N;
  p.act = act;
  p.vecA = lda % 4 == 0; p.vecB = ldb % 4 == 0;
  static float dummy;
  p.pre = has_pre ? &dummy : nullptr;
  p.rowscale = has_rowscale ? &dummy : nullptr;
  p.kscale = has_kscale ? &dummy : nullptr;
  if (small_gemm_ok(p, a_kmajor, b_kmajor, 1)) return 0;
  if (a_kmajor && b_kmajor && workspace_bytes > 0 && !has_rowscale) {
    const DwCfg d = choose_dw_direct(M, N, K);
    if (d.splits >= 2 && workspace_bytes >= d.splits * ((int64_t)M * N + M) * 4) return 0;
  }
  const Split6Cfg sc = choose_split6(p, a_kmajor, b_kmajor, workspace_bytes);
  if (!sc.bm) return 0;
  // 2: the interior pipelined 64 x 64 kernel, which can take its B operand from pre-split planes (rscotr_gemm_f32_rb)
  static const int pipelined = getenv("RSCOTR_BF16X6_PIPE") ? atoi(getenv("RSCOTR_BF16X6_PIPE")) : 1;
  const bool ragged = M % sc.bm || N % sc.bm || K % 16;
  return (sc.bm == 64 && !ragged && !a_kmajor && (pipelined & 1) && K % 32 == 0 && sc.klen % 32 == 0) ? 2 : 1;
}

// Planes of weights for rscotr_gemm_f32_wplanes (layout: gemm_wplanes_kernel).  table: device (n, 8) int64 rows {W, planes, N,
// K, ldw, npad, first block, transposed}; an entry takes ceil(npad * (reduction / 16) / 256) blocks (npad = rows of the plane
// set rounded up to 256; reduction = K, or N when transposed); total_blocks = their sum.
extern "C" int rscotr_gemm_split_weights(const int64_t* table, int n, int total_blocks, void* stream) {
  if (n < 0 || total_blocks < 0) return fail(RSCOTR_E_SHAPE, "rscotr_gemm_split_weights: negative count");
  if (n == 0 || total_blocks == 0) return RSCOTR_OK;
  if (!table) return fail(RSCOTR_E_ARG, "rscotr_gemm_split_weights: null table");
  split_weights_kernel<<<dim3((unsigned)total_blocks), 256, 0, (hipStream_t)stream>>>(table, n);
  return check_launch("rscotr_gemm_split_weights");
}

extern "C" int rscotr_gemm_split_weights_h3(const int64_t* table, int n, int total_blocks, void* stream) {
  if (n < 0 || total_blocks < 0) return fail(RSCOTR_E_SHAPE, "rscotr_gemm_split_weights_h3: negative count");
  if (n == 0 || total_blocks == 0) return RSCOTR_OK;
  if (!table) return fail(RSCOTR_E_ARG, "rscotr_gemm_split_weights_h3: null table");
  split_weights_h3_kernel<<<dim3((unsigned)total_blocks), 256, 0, (hipStream_t)stream>>>(table, n);
  return check_launch("rscotr_gemm_split_weights_h3");
}

// Tile width and k-slices of the pre-split product: one 128 x 256 workgroup is resident per CU (86 KB of LDS), two 128 x 128
// ones; the grid should be a whole number of such rounds over the 256 CUs (340 workgroups take as long as 512).
static void wplanes_cfg(int M, int N, int K, int* bn_out, int* splits_out) {
  static const char* force = getenv("RSCOTR_WPLANES_FORCE");  // "bn,splits" — tuning only
  if (force) {
    int b = 0, sp = 0;
    if (sscanf(force, "%d,%d", &b, &sp) == 2 && (b == 128 || b == 256) && sp >= 1) {
      *bn_out = b; *splits_out = (int)std::min<long>(sp, std::max(1, K / 128));
      return;
    }
  }
  const long tm = (M + 127) / 128;
  const long smax = std::max<long>(1, std::min<long>(8, K / 256));
  double best = -1.0;
  int bbn = 128, bsp = 1;
  for (int bn = 256; bn >= 128; bn -= 128) {
    if (bn == 256 && N <= 128) continue;
    const long t = tm * ((N + bn - 1) / bn), cap = bn == 256 ? 256 : 512;
    for (long sp = 1; sp <= smax; ++sp) {
      const long wgs = t * sp, rounds = (wgs + cap - 1) / cap;
      // time model: rounds x (k-steps per slice + prologue / epilogue worth ~4 steps) x tile area, plus the slab pass
      const double steps = (double)K / 16 / sp + 4.0;
      // (a round of 512 half-width workgroups covers the area of a round of 256 full-width ones, ~1.2x slower: lab)
      double cost = rounds * steps * (bn == 256 ? 1.0 : 1.22);
      if (sp > 1) cost += 2.0 * sp * M * N * 4.0 * 1.9e-7;  // slabs written and read at ~4 TB/s, in k-steps of 1.3 us
      const double score = 1.0 / cost;
      if (score > best) { best = score; bbn = bn; bsp = (int)sp; }
    }
  }
  *bn_out = bbn; *splits_out = bsp;
}

// Where the weight-plane kernel above wins: long reductions over many rows (the encoder's FFN2 / its dX).  (Round 4's tiled
// split-product kernels with their B operand from planes gained 15-28 % per dispatch and lost the round to the per-iteration
// re-split of every weight — profiles/r4_planes_b_tiled.txt — and left the library in round 5.)
static bool wplanes_classic(int M, int K) {
  static const int min_m = getenv("RSCOTR_WPLANES_MIN_M") ? atoi(getenv("RSCOTR_WPLANES_MIN_M")) : 4096;
  static const int min_k = getenv("RSCOTR_WPLANES_MIN_K") ? atoi(getenv("RSCOTR_WPLANES_MIN_K")) : 1024;
  return M >= min_m && K >= min_k;
}

/* 1: rscotr_gemm_f32_wplanes is worth calling for (M, N, K); 0: the caller multiplies with the fp32 weight (rscotr_gemm_f32) */
extern "C" int rscotr_gemm_f32_wplanes_ok(int M, int N, int K, int act_is_gelu) {
  if (M <= 0 || N <= 0 || K < 16 || K % 16) return 0;
  (void)act_is_gelu;
  return wplanes_classic(M, K) && N >= 64;
}

extern "C" int64_t rscotr_gemm_f32_wplanes_workspace(int M, int N, int K) {
  int bn, splits;
  wplanes_cfg(M, N, K, &bn, &splits);
  return splits > 1 ? (int64_t)splits * M * N * 4 : 0;
}

// C = epilogue(A x Bplanes): A (M, K) fp32 row-major (lda), planes = the pre-split B of rscotr_gemm_split_weights (N rows,
// npad >= N rounded up to 256, reduction K, K % 16 == 0); epilogue arguments as rscotr_gemm_f32.
extern "C" int rscotr_gemm_f32_wplanes(const float* A, const void* planes, int npad, float* C, int M, int N, int K, int lda,
                                       int ldc, const float* bias, int act, const float* aux, float* pre, const float* resid,
                                       int accumulate, const float* rowscale, int rows_per_scale, float* out2,
                                       float* workspace, int64_t workspace_bytes, void* stream) {
  if (M < 0 || N < 0 || K < 0) return fail(RSCOTR_E_SHAPE, "rscotr_gemm_f32_wplanes: negative dimension");
  if (M == 0 || N == 0) return RSCOTR_OK;
  if (!A || !planes || !C) return fail(RSCOTR_E_ARG, "rscotr_gemm_f32_wplanes: null pointer");
  if (K % 16 || K < 16 || npad % 256 || npad < N || lda < K || ldc < N || lda % 4 || !aligned16(A) || !aligned16(planes))
    return fail(RSCOTR_E_SHAPE, "rscotr_gemm_f32_wplanes: K %% 16, npad %% 256, 16-byte aligned A rows required (M=%d N=%d K=%d npad=%d lda=%d)",
                M, N, K, npad, lda);
  if (act < ACT_NONE || act > ACT_GELU_GRAD) return fail(RSCOTR_E_ARG, "rscotr_gemm_f32_wplanes: unknown act %d", act);
  if ((act == ACT_RELU_GRAD || act == ACT_GELU_GRAD) && !aux) return fail(RSCOTR_E_ARG, "rscotr_gemm_f32_wplanes: act %d needs aux", act);
  if (rowscale && rows_per_scale <= 0) return fail(RSCOTR_E_ARG, "rscotr_gemm_f32_wplanes: rowscale needs rows_per_scale > 0");
  GemmParams p;
  p.A = A; p.B = nullptr; p.C = C; p.bias = bias; p.aux = aux; p.pre = pre; p.resid = resid; p.C2 = out2;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = K; p.ldc = ldc;
  p.act = act; p.accumulate = accumulate;
  p.vecA = 1; p.vecB = 1;
  p.vecC = (ldc % 4 == 0) && aligned16(C) && aligned16(bias) && aligned16(aux) && aligned16(pre) && aligned16(resid) && aligned16(out2);
  p.rowsum = nullptr; p.rowsum_acc = 0; p.rs_slabs = nullptr;
  p.nb1 = 0; p.nb2 = 1;
  p.rowscale = rowscale; p.rows_per = rows_per_scale; p.kscale = nullptr; p.krows_per = 0;
  hipStream_t s = (hipStream_t)stream;
  int bn, splits;
  wplanes_cfg(M, N, K, &bn, &splits);
  if (splits > 1 && (!workspace || workspace_bytes < (int64_t)splits * M * N * 4))
    splits = (int)std::max<int64_t>(1, workspace ? workspace_bytes / ((int64_t)M * N * 4) : 1);
  const long tm = (M + 127) / 128;
  const long t128 = tm * ((N + 127) / 128), t256 = tm * ((N + 255) / 256);
  p.tiles = (int)(bn == 256 ? t256 : t128);
  p.splits = splits; p.ksplit_len = K;
  p.slabs = splits > 1 ? workspace : nullptr;
  static const bool prof_shapes_w = getenv("RSCOTR_PROF_SHAPES") != nullptr;
  char wname[112];
  if (prof_shapes_w) snprintf(wname, sizeof(wname), "M=%d N=%d K=%d 0p wplanes-%d splits=%d", M, N, K, bn, splits);
  else snprintf(wname, sizeof(wname), "rscotr::gemm_wplanes_kernel<%d>", bn);
  ProfScope prof(PROF_GEMM, 2.0 * M * N * K, s, "%s", wname);
  const unsigned nwg = (unsigned)p.tiles * (unsigned)splits;
  static const bool attr_set = [] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wplanes_kernel<256>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)wplanes_lds_bytes<256>());
    hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_wplanes_kernel<128>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)wplanes_lds_bytes<128>());
    return true;
  }();
  (void)attr_set;
  const unsigned short* pl = reinterpret_cast<const unsigned short*>(planes);
  if (bn == 256) gemm_wplanes_kernel<256><<<dim3(nwg), 512, wplanes_lds_bytes<256>(), s>>>(p, pl, npad);
  else gemm_wplanes_kernel<128><<<dim3(nwg), 512, wplanes_lds_bytes<128>(), s>>>(p, pl, npad);
  if (int e = check_launch("rscotr_gemm_f32_wplanes")) return e;
  if (splits > 1) {
    launch_splitk_reduce(p, workspace, s);
    return check_launch("rscotr_gemm_f32_wplanes (split-K reduce)");
  }
  return RSCOTR_OK;
}

// Grouped launch of deferred weight gradients: see gemm_f32_group_kernel.  table: device (n, 16) int64 (layout there),
// total_wgs = sum of the problems' workgroup counts; flops = sum of 2 M N K over the problems (the table lives on the device:
// the caller, who built it, states the algorithmic work of the launch for the launch-site profiler; 0 = not stated).
extern "C" int rscotr_gemm_dw_group(const int64_t* table, int n, int total_wgs, int variant, double flops,
                                    const uint32_t* amax_base, void* stream) {
  if (n < 0 || total_wgs < 0) return fail(RSCOTR_E_SHAPE, "rscotr_gemm_dw_group: negative count");
  if (n == 0 || total_wgs == 0) return RSCOTR_OK;
  if (!table) return fail(RSCOTR_E_ARG, "rscotr_gemm_dw_group: null table");
  ProfScope prof(PROF_GEMM, flops, (hipStream_t)stream, variant == 7 ? "rscotr::gemm_h3_group_kernel" : "rscotr::gemm_f32_group_kernel<%d>", variant);  // (2 / 3 / 6: bf16x6 bodies)
  if (variant == 7) {
    if (!amax_base) return fail(RSCOTR_E_ARG, "rscotr_gemm_dw_group: variant 7 needs the value-range words (amax_base)");
    gemm_h3_group_kernel<<<dim3((unsigned)total_wgs), 256, GROUP_LDS_BYTES, (hipStream_t)stream>>>(table, n, amax_base);
  } else if (variant == 0) {
    gemm_f32_group_kernel<0><<<dim3((unsigned)total_wgs), 256, gemm_lds_bytes<64, 64, 1>(), (hipStream_t)stream>>>(table, n);
  } else if (variant == 6) {
    gemm_f32_group_kernel<6><<<dim3((unsigned)total_wgs), 256, GROUP_LDS_BYTES, (hipStream_t)stream>>>(table, n);
  } else {
    return fail(RSCOTR_E_ARG, "rscotr_gemm_dw_group: variant must be 0 (fp32 matrix pipe, 64 x 64 tiles, any problem), 6 (six-term bf16 split product, 128 x 128 tiles with edges) or 7 (fp16 split product on the same tiles)");
  }
  return check_launch("rscotr_gemm_dw_group");
}

namespace rscotr {
// One workgroup = 256 output float4s (or row sums) of one pending problem: C[m, n..n+3] += sum_s slab_s (fixed order).
__global__ __launch_bounds__(256) void splitk_flush_kernel(const int64_t* __restrict__ table, const int32_t* __restrict__ wgmap) {
  const int entry = wgmap[2 * blockIdx.x], chunk = wgmap[2 * blockIdx.x + 1];
  const int64_t* t = table + (long)entry * 8;
  const float4* sl = reinterpret_cast<const float4*>(t[0]);
  const float* rsl = reinterpret_cast<const float*>(t[1]);
  float* C = reinterpret_cast<float*>(t[2]);
  float* rowsum = reinterpret_cast<float*>(t[3]);
  const int M = (int)t[4], N = (int)t[5], ldc = (int)t[6], splits = (int)t[7];
  const long total4 = ((long)M * N) >> 2;
  const long i = (long)chunk * 256 + threadIdx.x;
  if (i < total4) {
    float4 v = sl[i];
#pragma unroll 8
    for (int s = 1; s < splits; ++s) {
      const float4 u = sl[(long)s * total4 + i];
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    const long e = i << 2;
    const int m = (int)(e / N), n = (int)(e - (long)m * N);
    float4* c = reinterpret_cast<float4*>(C + (long)m * ldc + n);
    float4 o = *c;
    o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w;
    *c = o;
  }
  if (rowsum && i < M) {
    float r = 0.f;
    for (int s = 0; s < splits; ++s) r += rsl[(long)s * M + i];
    rowsum[i] += r;
  }
}
}  // namespace rscotr

// table: device (n, 8) int64 rows {slabs, row-sum slabs | 0, C, rowsum | 0, M, N, ldc, splits} (N % 4 == 0, ldc % 4 == 0,
// 16-byte aligned pointers: caller-checked); wgmap: device (nwg, 2) int32 rows {table row, chunk of 256 float4s}, with
// ceil(max(M * N / 4, M) / 256) chunks per row.
extern "C" int rscotr_splitk_flush(const int64_t* table, const int32_t* wgmap, int nwg, double bytes, void* stream) {
  if (nwg < 0) return fail(RSCOTR_E_SHAPE, "rscotr_splitk_flush: negative workgroup count");
  if (nwg == 0) return RSCOTR_OK;
  if (!table || !wgmap) return fail(RSCOTR_E_ARG, "rscotr_splitk_flush: null pointer");
  ProfScope prof(PROF_HBM, bytes, (hipStream_t)stream, "rscotr::splitk_flush_kernel");
  splitk_flush_kernel<<<dim3((unsigned)nwg), 256, 0, (hipStream_t)stream>>>(table, wgmap);
  return check_launch("rscotr_splitk_flush");
}

namespace rscotr {
// out[i] = sum_s slabs[s][i] (float4 lanes; n % 4 == 0)
__global__ __launch_bounds__(256) void slab_sum_kernel(const float4* __restrict__ slabs, float4* __restrict__ out, long n4,
                                                       int splits) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 a = slabs[i];
    for (int s = 1; s < splits; ++s) {
      const float4 v = slabs[(long)s * n4 + i];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    out[i] = a;
  }
}
}  // namespace rscotr

// Batched form: nb0 * nb1 independent problems of one shape, problem (b0, b1) at element offsets
// b0*s?0 + b1*s?1 of A, B, C (e.g. b0 = image, b1 = head: the per-head slices of (B, L, heads*32) tensors are
// addressed in place, no permute copies).  No bias / activation; accumulate adds into C.
// ksplits > 1 (row-major A = a_kmajor 0, k-major B = b_kmajor 1 only; K % ksplits == 0): the reduction is cut
// into ksplits slices, slice s of every problem writes slab s = workspace + s * c_elems (laid out like C,
// c_elems = elements of the whole C tensor), and the slabs are summed into C by a second kernel — for the
// attention products with few output tiles and thousands of keys (P v and dS k of the seg decoder's
// cross-attention: 100 queries x 4096 keys per head).
extern "C" int rscotr_gemm_f32_batched(const float* A, const float* B, float* C, int M, int N, int K, int lda,
                                       int ldb, int ldc, int a_kmajor, int b_kmajor, int nb0, int nb1,
                                       int64_t sA0, int64_t sA1, int64_t sB0, int64_t sB1, int64_t sC0,
                                       int64_t sC1, int accumulate, int ksplits, float* workspace,
                                       int64_t c_elems, void* stream) {
  if (M < 0 || N < 0 || K < 0 || nb0 < 0 || nb1 < 0) return fail(RSCOTR_E_SHAPE, "rscotr_gemm_f32_batched: negative dimension");
  if (M == 0 || N == 0 || nb0 == 0 || nb1 == 0) return RSCOTR_OK;
  if (!A || !B || !C) return fail(RSCOTR_E_ARG, "rscotr_gemm_f32_batched: null pointer");
  if (ksplits < 1) ksplits = 1;
  if ((long)nb0 * nb1 * ksplits > 65535) return fail(RSCOTR_E_SHAPE, "rscotr_gemm_f32_batched: more than 65535 problems");
  if (lda < (a_kmajor ? M : K) || ldb < (b_kmajor ? N : K) || ldc < N)
    return fail(RSCOTR_E_SHAPE, "rscotr_gemm_f32_batched: leading dimension too small");
  if (ksplits > 1 && (a_kmajor || !b_kmajor || K % ksplits || accumulate || !workspace || c_elems % 4 || !aligned16(C) ||
                      !aligned16(workspace)))
    return fail(RSCOTR_E_ARG, "rscotr_gemm_f32_batched: ksplits needs row-major A, k-major B, K %% ksplits == 0, a workspace");
  GemmParams p;
  p.A = A; p.B = B; p.C = ksplits > 1 ? workspace : C; p.C2 = nullptr; p.bias = nullptr; p.aux = nullptr; p.pre = nullptr; p.resid = nullptr;
  p.M = M; p.N = N; p.K = K / ksplits; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.act = ACT_NONE; p.accumulate = accumulate;
  const long kl = K / ksplits;
  p.vecC = 0;
  p.vecA = aligned16(A) && (lda % 4 == 0) && (sA0 % 4 == 0) && (sA1 % 4 == 0) && (kl % 4 == 0);
  p.vecB = aligned16(B) && (ldb % 4 == 0) && (sB0 % 4 == 0) && (sB1 % 4 == 0);
  p.rowsum = nullptr; p.rowsum_acc = 0;
  p.rowscale = nullptr; p.kscale = nullptr; p.rows_per = p.krows_per = 1;
  p.nb1 = nb1; p.nb2 = ksplits;
  p.sA0 = sA0; p.sA1 = sA1; p.sB0 = sB0; p.sB1 = sB1; p.sC0 = sC0; p.sC1 = sC1;
  p.sA2 = kl; p.sB2 = kl * ldb; p.sC2 = c_elems;
  p.ksplit_len = p.K; p.splits = 1; p.slabs = nullptr; p.rs_slabs = nullptr;
  int BM = 64, BN = 64;
  if (N <= 32) { BM = 128; BN = 32; }
  const long tiles = (long)((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  p.tiles = (int)tiles;
  dim3 grid((unsigned)tiles, (unsigned)(nb0 * nb1 * ksplits), 1);
  hipStream_t s = (hipStream_t)stream;
  const int kg = choose_kgroups((long)grid.x * grid.y, (p.K + GEMM_BK - 1) / GEMM_BK, false);
  if (BM == 64) launch_gemm_cfg<64, 64, 2, 2>(p, a_kmajor, b_kmajor, grid, s, kg);
  else launch_gemm_cfg<128, 32, 4, 1>(p, a_kmajor, b_kmajor, grid, s, kg);
  if (int e = check_launch("rscotr_gemm_f32_batched")) return e;
  if (ksplits > 1) {
    const long n4 = c_elems / 4;
    slab_sum_kernel<<<(unsigned)std::min<long>((n4 + 255) / 256, 1024), 256, 0, s>>>(
        reinterpret_cast<const float4*>(workspace), reinterpret_cast<float4*>(C), n4, ksplits);
    return check_launch("rscotr_gemm_f32_batched (slab sum)");
  }
  return RSCOTR_OK;
}

extern "C" int64_t rscotr_colsum_f32_workspace(int M, int N) {
  if (M <= 0 || N <= 0) return 0;
  return (int64_t)colsum_gy(M, N) * N * 4;
}

extern "C" int rscotr_colsum_f32(const float* X, float* out, int M, int N, int ld, int accumulate,
                                 float* workspace, int64_t workspace_bytes, void* stream) {
  if (M < 0 || N < 0) return fail(RSCOTR_E_SHAPE, "rscotr_colsum_f32: negative dimension");
  if (N == 0) return RSCOTR_OK;
  if (!X || !out) return fail(RSCOTR_E_ARG, "rscotr_colsum_f32: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (M == 0) {
    if (!accumulate) hipMemsetAsync(out, 0, (size_t)N * 4, s);
    return RSCOTR_OK;
  }
  const int gy = colsum_gy(M, N);
  if (!workspace || workspace_bytes < (int64_t)gy * N * 4)
    return fail(RSCOTR_E_ARG, "rscotr_colsum_f32: workspace of rscotr_colsum_f32_workspace() bytes required");
  const int rpb = (M + gy - 1) / gy;
  const int gyu = (M + rpb - 1) / rpb;
  const int vec = aligned16(X) && (ld % 4 == 0);
  colsum_partial_kernel<<<dim3((N + 255) / 256, gyu), 256, 0, s>>>(X, workspace, M, N, ld, rpb, vec);
  colsum_final_kernel<<<(N + 255) / 256, 256, 0, s>>>(workspace, out, gyu, N, accumulate);
  return check_launch("rscotr_colsum_f32");
}

// ---------------------------------------------------------------------------------------------------------------
// Value range of a tensor for the fp16 split product: slot = max(slot, bit pattern of max |X[r, c]|) over rows x cols with
// row stride ld (slot = a range word of kAmaxPlanes sub-words, gemm_common.h).  The caller zeroes the slot (one memset for all slots of an iteration); the maximum is taken per lane,
// per wavefront (shuffles), per workgroup (LDS), and the workgroup marks the byte of its binade in the word (common.h: an idempotent plain
// store — no atomics, deterministic).  NaNs compare above every finite pattern.
namespace rscotr {
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ X, long rows, int cols, int ld, int vec,
                                                   unsigned* __restrict__ slot) {
  __shared__ unsigned sm[4];
  unsigned m = 0u;
  const long tid = (long)blockIdx.x * 256 + threadIdx.x, nth = (long)gridDim.x * 256;
  if (vec) {
    const int c4 = cols >> 2;
    const long n4 = rows * c4;
    for (long i = tid; i < n4; i += nth) {
      const long r = i / c4;
      const int c = (int)(i - r * c4) << 2;
      const uint4 v = *reinterpret_cast<const uint4*>(X + r * ld + c);
      m = max(max(m, v.x & 0x7fffffffu), max(max(v.y & 0x7fffffffu, v.z & 0x7fffffffu), v.w & 0x7fffffffu));
    }
  } else {
    const long n = rows * cols;
    for (long i = tid; i < n; i += nth) {
      const long r = i / cols;
      m = max(m, __float_as_uint(X[r * ld + (i - r * cols)]) & 0x7fffffffu);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(sm[0], sm[1]), max(sm[2], sm[3]));
    if (m) range_mark(slot, range_byte(m));  // (a plain byte store: common.h)
  }
}
}  // namespace rscotr

namespace rscotr {
// The same for MANY tensors in one launch (the operands of the grouped weight-gradient launch that arrived without a range):
// table rows {X, rows, cols, ld, slot, first block}; entry e owns blocks [first_e, first_{e+1}) (the last one up to gridDim.x).
__global__ __launch_bounds__(256) void amax_group_kernel(const int64_t* __restrict__ table, int n) {
  __shared__ unsigned sm[4];
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int)table[(long)mid * 6 + 5] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const int64_t* t = table + (long)lo * 6;
  const float* X = reinterpret_cast<const float*>(t[0]);
  const long rows = t[1];
  const int cols = (int)t[2], ld = (int)t[3];
  unsigned* slot = reinterpret_cast<unsigned*>(t[4]);
  const int first = (int)t[5], nb = (lo + 1 < n ? (int)table[(long)(lo + 1) * 6 + 5] : (int)gridDim.x) - first;
  const long tid = (long)(blockIdx.x - first) * 256 + threadIdx.x, nth = (long)nb * 256;
  unsigned m = 0u;
  if (((t[0] & 15) == 0) && cols % 4 == 0 && ld % 4 == 0) {
    const int c4 = cols >> 2;
    const long n4 = rows * c4;
    for (long i = tid; i < n4; i += nth) {
      const long r = i / c4;
      const int c = (int)(i - r * c4) << 2;
      const uint4 v = *reinterpret_cast<const uint4*>(X + r * ld + c);
      m = max(max(m, v.x & 0x7fffffffu), max(max(v.y & 0x7fffffffu, v.z & 0x7fffffffu), v.w & 0x7fffffffu));
    }
  } else {
    const long nn = rows * cols;
    for (long i = tid; i < nn; i += nth) {
      const long r = i / cols;
      m = max(m, __float_as_uint(X[r * ld + (i - r * cols)]) & 0x7fffffffu);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(sm[0], sm[1]), max(sm[2], sm[3]));
    if (m) range_mark(slot, range_byte(m));  // (a plain byte store: common.h)
  }
}
}  // namespace rscotr

extern "C" int rscotr_amax_group(const int64_t* table, int n, int total_blocks, void* stream) {
  if (n < 0 || total_blocks < 0) return rscotr::fail(RSCOTR_E_SHAPE, "rscotr_amax_group: negative count");
  if (n == 0 || total_blocks == 0) return RSCOTR_OK;
  if (!table) return rscotr::fail(RSCOTR_E_ARG, "rscotr_amax_group: null table");
  rscotr::amax_group_kernel<<<dim3((unsigned)total_blocks), 256, 0, (hipStream_t)stream>>>(table, n);
  return rscotr::check_launch("rscotr_amax_group");
}

extern "C" int rscotr_amax_f32(const float* X, int64_t rows, int cols, int ld, uint32_t* slot, void* stream) {
  if (rows < 0 || cols < 0 || ld < cols) return rscotr::fail(RSCOTR_E_SHAPE, "rscotr_amax_f32: bad shape");
  if (rows == 0 || cols == 0) return RSCOTR_OK;
  if (!X || !slot) return rscotr::fail(RSCOTR_E_ARG, "rscotr_amax_f32: null pointer");
  const int vec = rscotr::aligned16(X) && cols % 4 == 0 && ld % 4 == 0;
  const long n = rows * (long)cols;
  const unsigned grid = (unsigned)std::max<long>(1, std::min<long>(256, (n + 8191) / 8192));
  rscotr::amax_kernel<<<dim3(grid), 256, 0, (hipStream_t)stream>>>(X, rows, cols, ld, vec, slot);
  return rscotr::check_launch("rscotr_amax_f32");
}
