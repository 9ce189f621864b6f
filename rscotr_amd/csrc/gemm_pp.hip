// Planes x planes split product fed by LDS-DMA (rscotr_gemm_pp, rscotr_split_planes).
//
// The six-term bf16 product of precision mode 3 (csrc/gemm.hip: x = h + m + l, l*h + h*l + m*m + m*h + h*m + h*h, fp32
// accumulate, the error class of an fp32 FMA chain) with BOTH operands arriving as bf16 plane sets: the product kernel does no
// conversion, no VGPR staging and no LDS write instruction — its stages are filled by global_load_lds_dwordx4 (LDS-DMA, requests in
// flight across the barrier) and its fragments are conflict-free 16-byte reads.  This is the substrate under the large Linears
// of the step (reference call sites: the FFN 256 -> 2048 -> 256 of the shared encoder, configs/multi/MTL_slvlcls_...potsdam.py:44-49;
// the MSDeformAttn projections, :39-43; Swin qkv / proj / MLP, :9-25) and their two backward contractions.
//
// ST32 plane layout of a stored tensor X (R rows, C contiguous columns), rows and columns padded with zeros to multiples of 32:
//   super-tile (rt, ct) = 32 x 32 elements, tiles in row-major order; per super-tile three planes (h, m, l) of 2 KB; inside a plane
//   128 units of 16 bytes, unit (r, c8) = row r, columns 8 c8 .. 8 c8 + 7, at
//       slot(r, c8) = 32 c8 + 16 (r / 16) + 4 ((r / 4 + c8) & 3) + (r & 3).
// One plane set serves both uses of a tensor in a product  C[m, n] = sum_k Aop[m, k] Bop[n, k]:
//   ROW mode (MFMA rows = stored rows, reduction over stored columns: x in y = x W^T, dy in dx = dy W, W in y = x W^T): lane
//     (r = lane & 31, g = lane >> 5) reads unit (r, 2 ks + g) with one ds_read_b128; the 16 rows of a b128 lane group fall on 16
//     distinct slots mod 16: conflict-free.
//   COL mode (MFMA rows = stored columns, reduction over stored rows: dy and x in dW = dy^T x, W in dx = dy W):
//     ds_read_b64_tr_b16 (the LDS transpose read of gfx950) on the same image; the 4 rows x 4 c8 of a 32-lane group fall on 16
//     distinct slots mod 16: conflict-free.
// A stage of the product = 16 k of a 128 x 128 tile = 24 pieces of 1 KB (32 rows x 16 k of one plane), each filled by ONE
// global_load_lds_dwordx4 of one wavefront from one contiguous KB (ROW) / four 256-byte runs (COL) of the plane set.
//
// Kernel: 256 threads = 4 wavefronts (2 x 2, wave tile 64 x 64), ring of three stages (72 KB: two workgroups per CU cover each
// other's fill, drain and epilogue), ONE raw s_barrier per stage: wait vmcnt(6) (stage t landed, t + 1 may fly) -> barrier ->
// fragment reads of stage t -> 24 MFMAs with the six LDS-DMA pieces of stage t + 2 issued one per four MFMAs.  Measured
// (scripts/lab/pp_lab.hip, profiles/r4_pp_lab.txt): 10880 x 2048 x 256 in 65 us against 89 for the in-kernel split, dy W 67 against
// 101, weight gradients 256 x 2048 x 10880 in 16 k-slices 67 against 98; the kernel sits at 0.87 of its own MFMA-only + epilogue
// floor.  Split-K through caller slabs, the deferred combine and the fused epilogue are shared with csrc/gemm.hip.
#include "gemm_common.h"
#include <algorithm>

namespace rscotr {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

__host__ __device__ inline int st32_slot(int r, int c8) { return c8 * 32 + ((r >> 4) << 4) + ((((r >> 2) + c8) & 3) << 2) + (r & 3); }
// 16-byte unit index of (row, col8 = col / 8) of plane pl; CT = column tiles of the stored tensor
__host__ __device__ inline long st32_unit(int row, int col8, int CT, int pl) {
  return (((long)(row >> 5) * CT + (col8 >> 2)) * 3 + pl) * 128 + st32_slot(row & 31, col8 & 3);
}

// ---------------------------------------------------------------------------------------------------------------
// fp32 (R x C, row stride ld) -> ST32 planes.  Workgroup = 256 rows x 32 columns (one column tile): wavefront w takes rows
// 64 it + 16 w .. + 15 of the group for it = 0..3, lane = (row lane >> 2, c8 = lane & 3): full 128-byte lines in, 256-byte runs of
// units out.  rowscale: row r is multiplied by rowscale[r / rows_per] BEFORE the split (the per-sample factor of a DropPath /
// Mixup folded into a Linear: the planes of s * dy serve dx = (s dy) W and dW = (s dy)^T x alike).  colsum: part[rg][c] = sum of
// the (scaled) rows of row group rg — the bias gradient of the Linear whose dy is being split, folded in fixed order by the
// deferred combine (rscotr_splitk_flush reads it as row-sum partials).
struct SplitParams {
  const float* X;
  uint4* planes;
  const float* rowscale;
  float* colsum;      // [row groups][C] or null
  int R, C, ld, RT, CT, rows_per, vec;
};

__device__ __forceinline__ void split_planes_block(const SplitParams& p, const int bx, const int by, float (*red)[32]) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int ct = bx, c8 = lane & 3, col = ct * 32 + c8 * 8;
  float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int row = by * 256 + it * 64 + w * 16 + (lane >> 2);
    if (row >= p.RT * 32) break;
    float v[8];
    if (row < p.R && p.vec && col + 8 <= p.C) {
      const float4* src = reinterpret_cast<const float4*>(p.X + (long)row * p.ld + col);
      const float4 a = src[0], b = src[1];
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (row < p.R && col + j < p.C) ? p.X[(long)row * p.ld + col + j] : 0.f;
    }
    if (p.rowscale && row < p.R) {
      const float f = p.rowscale[row / p.rows_per];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] *= f;
    }
    unsigned short q[8][3];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      __bf16 pl[3];
      split_planes<3>(v[j], pl);
      cs[j] += v[j];
#pragma unroll
      for (int k = 0; k < 3; ++k) q[j][k] = __builtin_bit_cast(unsigned short, pl[k]);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      uint4 o;
      o.x = q[0][k] | ((unsigned)q[1][k] << 16); o.y = q[2][k] | ((unsigned)q[3][k] << 16);
      o.z = q[4][k] | ((unsigned)q[5][k] << 16); o.w = q[6][k] | ((unsigned)q[7][k] << 16);
      p.planes[st32_unit(row, ct * 4 + c8, p.CT, k)] = o;
    }
  }
  if (p.colsum) {  // (uniform)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float s = cs[j];
      s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 8, 64); s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
      if (lane < 4) red[w][lane * 8 + j] = s;
    }
    __syncthreads();
    if (threadIdx.x < 32) {
      const int c = ct * 32 + threadIdx.x;
      if (c < p.C) p.colsum[(long)by * p.C + c] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
    }
  }
}

__global__ __launch_bounds__(256) void split_planes_kernel(SplitParams p) {
  __shared__ float red[4][32];
  split_planes_block(p, blockIdx.x, blockIdx.y, red);
}

// Many tensors in one launch (the parameters of a task, once per optimizer step): table rows {X, planes, R, C, ld, first block,
// 0, 0} (int64 x 8), entry e owns the blocks first_e .. first_{e+1} - 1 = CT x ceil(32 RT / 256), column tile fastest.
__global__ __launch_bounds__(256) void split_planes_group_kernel(const int64_t* __restrict__ table, int n) {
  __shared__ float red[4][32];
  int e = 0;
  while (e + 1 < n && (long)table[(long)(e + 1) * 8 + 5] <= (long)blockIdx.x) ++e;
  const int64_t* t = table + (long)e * 8;
  SplitParams p;
  p.X = reinterpret_cast<const float*>(t[0]);
  p.planes = reinterpret_cast<uint4*>(t[1]);
  p.rowscale = nullptr; p.colsum = nullptr;
  p.R = (int)t[2]; p.C = (int)t[3]; p.ld = (int)t[4];
  p.RT = (p.R + 31) / 32; p.CT = (p.C + 31) / 32; p.rows_per = 0;
  p.vec = (t[0] & 15) == 0 && p.ld % 4 == 0;
  const int b = (int)((long)blockIdx.x - t[5]);
  split_planes_block(p, b % p.CT, b / p.CT, red);
}

// ---------------------------------------------------------------------------------------------------------------
struct PPOperands {
  const char* A;   // plane set of the tensor behind the A operand
  const char* B;
  int ctA, ctB;    // column tiles of the stored tensors
  int mtA, mtB;    // 32-row tiles of the operands (ceil(M / 32), ceil(N / 32)): tiles past them are clamped reads
  int tiles_n;
};

constexpr int PP_BM = 128, PP_BN = 128, PP_NST = 3;
constexpr int PP_NA = PP_BM / 32 * 3, PP_NB = PP_BN / 32 * 3, PP_NP = PP_NA + PP_NB;  // 1 KB pieces per stage
constexpr int PP_PW = PP_NP / 4;                                                       // pieces per wavefront per stage
constexpr int PP_STAGE = PP_NP * 1024;
constexpr size_t PP_LDS_BYTES = (size_t)PP_NST * PP_STAGE;
static_assert(PP_NP % 4 == 0, "pieces divide evenly over the four wavefronts");

#define RSCOTR_WAIT_VM(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// Epilogue operands through LDS.  After the product loop the ring (72 KB) is idle; a 128 x 128 fp32 tile of an epilogue operand
// (aux, residual, old C) is 64 KB = 64 pieces of 1 KB (two rows of 128 floats), brought in by LDS-DMA — sixteen pieces per
// wavefront, all in flight at once: ONE memory round trip per operand and workgroup.  A short grid has one workgroup per CU and
// nothing else to hide the ~2 us of an HBM-cold load; loads into registers in row groups (csrc/gemm_common.h) pay that latency
// eight to sixteen times per wavefront (measured in the step: 10880 x 256 x 256 with a residual 32 us against 15 without).
// Needs 16-byte aligned rows (p.vecC).  Rows past M are clamped, columns past N start at N - 4 (in-bounds garbage, never used).
template <bool EDGE>
__device__ __forceinline__ void pp_stage_tile(const GemmParams& p, const float* __restrict__ T, char* lds, const int m0,
                                              const int n0, const int wave, const int lane) {
  const int col = EDGE ? min(n0 + (lane & 31) * 4, p.N - 4) : n0 + (lane & 31) * 4;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int piece = wave * 16 + i;
    const int row = EDGE ? min(m0 + 2 * piece + (lane >> 5), p.M - 1) : m0 + 2 * piece + (lane >> 5);
    const float* src = T + (long)row * p.ldc + col;
    __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(lds + piece * 1024), 16, 0, 0);
  }
}

// p: the epilogue / split-K fields of GemmParams (A, B, lda, ldb unused); p.tiles = output tiles, p.splits = k-slices (the 16-k
// steps divided evenly), p.slabs as in csrc/gemm.hip.
template <bool ACOL, bool BCOL, bool EDGE>
__global__ __launch_bounds__(256, 2) void gemm_pp_kernel(GemmParams p, PPOperands o) {
  extern __shared__ __attribute__((aligned(1024))) char pp_lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int idx = xcd_swizzle(blockIdx.x, gridDim.x);
  const int split = idx / p.tiles, tile = idx - split * p.tiles;  // (the tiles of a k-slice are neighbours: they share operands)
  const int tm = tile / o.tiles_n, tn = tile - tm * o.tiles_n;
  const int nk_all = (p.K + 15) >> 4;
  const int kt0 = (int)((long)split * nk_all / p.splits), nk = (int)((long)(split + 1) * nk_all / p.splits) - kt0;

  // per-lane source offset inside a 2 KB plane for a 1 KB LDS piece (lane L fills LDS bytes 16 L ..): ROW: the k-step's half of
  // the plane is contiguous; COL: four 256-byte runs (one per c8 block), the k-step selects the 16-row half of each
  const unsigned src_row = lane * 16, src_col = (lane >> 4) * 512 + (lane & 15) * 16;
  // per-lane read offsets inside a 1 KB piece.  ROW: unit (r, c8 = 2 ks + g): half-local slot 32 g + 16 (r / 16) +
  // 4 ((r / 4 + g + 2 ks) & 3) + (r & 3) — the k-step's parity flips bit 7 of the byte offset
  const int r = lane & 31, g = lane >> 5;
  const int rd_row0 = (g * 32 + ((r >> 4) << 4) + ((((r >> 2) + g) & 3) << 2) + (r & 3)) * 16;
  // COL: ds_read_b64_tr_b16 hands lane i' of a 16-lane group element (i' % 4) of the 8-byte chunks addressed by lanes 4 j + i' / 4
  // (j = 0..3) of the group: lane c_l = lane & 15 addresses the chunk (row j = c_l / 4 of a row quad, column quad c_l % 4) of the
  // columns 16 (G & 1) .. of its group G = lane >> 4; the quad is rows 8 (G >> 1) + 4 hr .. of the k-step (hr = 0, 1: two reads)
  const int G = lane >> 4, cl = lane & 15, c8c = 2 * (G & 1) + ((cl & 3) >> 1), gq = G >> 1;
  const int rd_col0 = c8c * 256 + ((((2 * gq + 0 + c8c) & 3) << 2) + (cl >> 2)) * 16 + (cl & 1) * 8;
  const int rd_col1 = c8c * 256 + ((((2 * gq + 1 + c8c) & 3) << 2) + (cl >> 2)) * 16 + (cl & 1) * 8;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // the six pieces this wavefront brings in per stage: piece q = wave + 4 i -> (operand, 32-row tile, plane); everything that
  // does not depend on the k-step is fixed here.  Row tiles past the operand are clamped to its last tile (the rows they feed
  // are never stored).
  const char* pbase[PP_PW];
  bool pisA[PP_PW];
  unsigned plane_off[PP_PW], plds[PP_PW];
#pragma unroll
  for (int i = 0; i < PP_PW; ++i) {
    const int q = wave + i * 4;
    const bool isA = q < PP_NA;
    const int qq = isA ? q : q - PP_NA;
    const int j = qq / 3, pl = qq - 3 * j;
    pisA[i] = isA;
    plds[i] = q * 1024;
    if (isA) {
      const int mt = min(tm * (PP_BM / 32) + j, o.mtA - 1);
      pbase[i] = o.A + ((ACOL ? (long)mt : (long)mt * o.ctA) * 3 + pl) * 2048;
      plane_off[i] = ACOL ? src_col : src_row;
    } else {
      const int nt = min(tn * (PP_BN / 32) + j, o.mtB - 1);
      pbase[i] = o.B + ((BCOL ? (long)nt : (long)nt * o.ctB) * 3 + pl) * 2048;
      plane_off[i] = BCOL ? src_col : src_row;
    }
  }
  // k-step dependent part of a source address.  ROW: super-tile kt / 2 along the row of tiles, half kt & 1 of the plane;
  // COL: super-tile row kt / 2 (ct tiles each), 16-row half kt & 1 of every c8 block
  auto koffA = [&](int kt) -> long { return ACOL ? (long)(kt >> 1) * o.ctA * 6144 + (kt & 1) * 256 : (long)(kt >> 1) * 6144 + (kt & 1) * 1024; };
  auto koffB = [&](int kt) -> long { return BCOL ? (long)(kt >> 1) * o.ctB * 6144 + (kt & 1) * 256 : (long)(kt >> 1) * 6144 + (kt & 1) * 1024; };
  auto issue_one = [&](int i, long ka, long kb, int st_off) {
    const char* src = pbase[i] + (pisA[i] ? ka : kb) + plane_off[i];
    __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(pp_lds + st_off + plds[i]), 16, 0, 0);
  };
  auto issue = [&](int kt, int st_off) {
    const long ka = koffA(kt), kb = koffB(kt);
#pragma unroll
    for (int i = 0; i < PP_PW; ++i) issue_one(i, ka, kb, st_off);
  };
  auto frag_row = [&](const char* piece, int par) -> bf16x8 {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(piece + (rd_row0 ^ (par << 7))));
  };
  auto frag_col = [&](const char* piece) -> bf16x8 {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(piece + rd_col0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(piece + rd_col1));
    const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  };

  if (nk > 0) issue(kt0, 0);
  if (nk > 1) issue(kt0 + 1, PP_STAGE);
  // the bias of this lane's two output columns: requested now, used after the loop (hipcc counts this load: its wait comes with
  // the first use, behind every LDS-DMA piece — all of which have landed by then)
  float bias_v[2] = {0.f, 0.f};
  if (p.bias && p.splits == 1) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = tn * PP_BN + wn * 64 + j * 32 + (lane & 31);
      bias_v[j] = (!EDGE || n < p.N) ? p.bias[n] : 0.f;
    }
  }
  int st_off = 0;
  for (int t = 0; t < nk; ++t) {
    // stage t has landed for this wavefront (the six pieces of stage t + 1 may still fly); after the barrier it has landed for
    // all four, and every wavefront is done reading stage t - 1 — the ring slot stage t + 2 goes into
    if (t + 1 < nk) { RSCOTR_WAIT_VM(PP_PW); } else { RSCOTR_WAIT_VM(0); }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const bool more = t + 2 < nk;
    const int nst_off = st_off + 2 * PP_STAGE >= PP_NST * PP_STAGE ? st_off + 2 * PP_STAGE - PP_NST * PP_STAGE : st_off + 2 * PP_STAGE;
    const long ka = koffA(kt0 + t + 2), kb = koffB(kt0 + t + 2);
    const char* sb = pp_lds + st_off;
    const int par = (kt0 + t) & 1;
    bf16x8 af[2][3], bf[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        const char* pa = sb + ((wm * 2 + i) * 3 + pl) * 1024;
        af[i][pl] = ACOL ? frag_col(pa) : frag_row(pa, par);
        const char* pb = sb + (PP_NA + (wn * 2 + i) * 3 + pl) * 1024;
        bf[i][pl] = BCOL ? frag_col(pb) : frag_row(pb, par);
      }
    // small terms first; term-major over the 2 x 2 accumulators: consecutive MFMAs write different accumulators
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    constexpr int EVERY = 24 / PP_PW;
#pragma unroll
    for (int tmm = 0; tmm < 6; ++tmm)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA[tmm]], bf[j][PB[tmm]], acc[i][j], 0, 0, 0);
          const int n = tmm * 4 + i * 2 + j;  // MFMA number inside the stage: one LDS-DMA piece after every fourth
          if (n % EVERY == EVERY - 1 && more) issue_one(n / EVERY, ka, kb, nst_off);
        }
    st_off = st_off + PP_STAGE == PP_NST * PP_STAGE ? 0 : st_off + PP_STAGE;
  }

  const int fr = lane & 31;
  const int m0 = tm * PP_BM, n0 = tn * PP_BN;
  if (p.splits > 1) {
    float* slab = p.slabs + (long)split * p.M * p.N;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + fr;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int m = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * g;
          if (!EDGE || (m < p.M && n < p.N)) slab[(long)m * p.N + n] = acc[i][j][e];
        }
      }
    return;
  }
  // bias / ReLU / GELU only (every forward Linear without a residual): nothing to load, the accumulators go out as they are
  const bool simple = !p.pre && !p.resid && !p.accumulate && !p.rowscale && !p.C2 && p.act <= ACT_GELU;
  const int act = p.act, ldc = p.ldc;
  if (simple) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + fr;
        if (EDGE && n >= p.N) continue;
        const int mb = m0 + wm * 64 + i * 32 + 4 * g;
        float* crow = p.C + (long)mb * ldc + n;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          float v = acc[i][j][e] + bias_v[j];
          if (act == ACT_RELU) v = fmaxf(v, 0.f);
          else if (act == ACT_GELU) v = gelu_f(v);
          if (!EDGE || mb + (e & 3) + 8 * (e >> 2) < p.M) crow[((e & 3) + 8 * (e >> 2)) * ldc] = v;
        }
      }
    return;
  }
  // general epilogue: v += bias; pre; act / act'(aux); row scale; residual / second output; old C — operands staged through LDS
  // one after the other (most epilogues have one), values read back per 32 x 32 accumulator tile
  const bool need_aux = act == ACT_RELU_GRAD || act == ACT_GELU_GRAD;
  const bool staged = p.vecC != 0 && p.N % 4 == 0;
  auto fetch = [&](const float* T) {  // -> the tile of T at (m0, n0) sits in LDS as [128][128] floats
    __builtin_amdgcn_s_barrier();     // every wavefront is done with what the LDS held
    asm volatile("" ::: "memory");
    pp_stage_tile<EDGE>(p, T, pp_lds, m0, n0, wave, lane);
    RSCOTR_WAIT_VM(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  // element (i, j, e) of this lane in the staged tile / in a row-major tensor
  auto lds_at = [&](int i, int j, int e) -> float {
    return *reinterpret_cast<const float*>(pp_lds + (((wm * 64 + i * 32 + 4 * g + (e & 3) + 8 * (e >> 2)) * 128) + wn * 64 + j * 32 + fr) * 4);
  };
  auto row_of = [&](int i, int e) -> int { return m0 + wm * 64 + i * 32 + 4 * g + (e & 3) + 8 * (e >> 2); };
  auto col_of = [&](int j) -> int { return n0 + wn * 64 + j * 32 + fr; };
  auto live = [&](int i, int j, int e) -> bool { return !EDGE || (row_of(i, e) < p.M && col_of(j) < p.N); };
  auto get = [&](const float* T, int i, int j, int e) -> float {
    return staged ? lds_at(i, j, e) : (live(i, j, e) ? T[(long)row_of(i, e) * ldc + col_of(j)] : 0.f);
  };
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] += bias_v[j];
  if (p.pre) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (live(i, j, e)) p.pre[(long)row_of(i, e) * ldc + col_of(j)] = acc[i][j][e];
  }
  if (need_aux) {
    if (staged) fetch(p.aux);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float x = get(p.aux, i, j, e);
          acc[i][j][e] = act == ACT_RELU_GRAD ? (x > 0.f ? acc[i][j][e] : 0.f) : acc[i][j][e] * gelu_grad_f(x);
        }
  } else if (act == ACT_RELU) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = fmaxf(acc[i][j][e], 0.f);
  } else if (act == ACT_GELU) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = gelu_f(acc[i][j][e]);
  }
  if (p.rowscale) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float f = (!EDGE || row_of(i, e) < p.M) ? p.rowscale[row_of(i, e) / p.rows_per] : 1.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j][e] *= f;
      }
  }
  if (p.accumulate) {
    if (staged) fetch(p.C);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] += get(p.C, i, j, e);
  }
  if (p.C2) {  // C = value (+ old C), C2 = C + resid
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (live(i, j, e)) p.C[(long)row_of(i, e) * ldc + col_of(j)] = acc[i][j][e];
  }
  if (p.resid) {
    if (staged) fetch(p.resid);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] += get(p.resid, i, j, e);
  }
  float* dst = p.C2 ? p.C2 : p.C;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (live(i, j, e)) dst[(long)row_of(i, e) * ldc + col_of(j)] = acc[i][j][e];
}

}  // namespace rscotr

using namespace rscotr;

extern "C" int64_t rscotr_planes_bytes(int rows, int cols) {
  if (rows <= 0 || cols <= 0) return 0;
  return (int64_t)((rows + 31) / 32) * ((cols + 31) / 32) * 6144;
}

extern "C" int rscotr_split_planes_parts(int rows) { return rows <= 0 ? 0 : (rows + 255) / 256; }

extern "C" int rscotr_split_planes(const float* X, int rows, int cols, int ld, void* planes, const float* rowscale,
                                   int rows_per_scale, float* colsum_parts, void* stream) {
  if (rows < 0 || cols < 0) return fail(RSCOTR_E_SHAPE, "rscotr_split_planes: negative dimension");
  if (rows == 0 || cols == 0) return RSCOTR_OK;
  if (!X || !planes) return fail(RSCOTR_E_ARG, "rscotr_split_planes: null pointer");
  if (ld < cols) return fail(RSCOTR_E_SHAPE, "rscotr_split_planes: row stride %d < %d columns", ld, cols);
  if (!aligned16(planes)) return fail(RSCOTR_E_ALIGN, "rscotr_split_planes: the plane set must be 16-byte aligned");
  if (rowscale && rows_per_scale <= 0) return fail(RSCOTR_E_ARG, "rscotr_split_planes: rowscale needs rows_per_scale > 0");
  SplitParams p;
  p.X = X; p.planes = reinterpret_cast<uint4*>(planes); p.rowscale = rowscale; p.colsum = colsum_parts;
  p.R = rows; p.C = cols; p.ld = ld; p.RT = (rows + 31) / 32; p.CT = (cols + 31) / 32; p.rows_per = rows_per_scale;
  p.vec = aligned16(X) && ld % 4 == 0;
  const dim3 grid((unsigned)p.CT, (unsigned)((p.RT * 32 + 255) / 256));
  split_planes_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(p);
  return check_launch("rscotr_split_planes");
}

extern "C" int rscotr_split_planes_group(const int64_t* table, int n, int total_blocks, void* stream) {
  if (n < 0 || total_blocks < 0) return fail(RSCOTR_E_SHAPE, "rscotr_split_planes_group: negative count");
  if (n == 0 || total_blocks == 0) return RSCOTR_OK;
  if (!table) return fail(RSCOTR_E_ARG, "rscotr_split_planes_group: null table");
  split_planes_group_kernel<<<dim3((unsigned)total_blocks), 256, 0, (hipStream_t)stream>>>(table, n);
  return check_launch("rscotr_split_planes_group");
}

// k-slices of a product: the grid should hold ~2 workgroups per CU; a slice keeps >= 8 steps of 16 k
// (measured, scripts/bench_pp.py: a second slice of 10880 x 256 x 256 — 170 tiles — costs 11 us of slab traffic and a combine
// launch to save 3; 48 tiles x 1536 k in 4-6 slices halve the time): no slices from 192 tiles on, >= 256 k per slice
static int pp_splits(int M, int N, int K) {
  const long tiles = (long)((M + PP_BM - 1) / PP_BM) * ((N + PP_BN - 1) / PP_BN);
  const int nk = (K + 15) / 16;
  static const long target = getenv("RSCOTR_PP_SPLIT_TARGET") ? atol(getenv("RSCOTR_PP_SPLIT_TARGET")) : 384;
  if (tiles >= target / 2) return 1;
  long sp = (target + tiles - 1) / tiles;
  sp = std::min<long>(sp, nk / 16);
  return (int)std::max<long>(1, std::min<long>(sp, 128));
}

extern "C" int64_t rscotr_gemm_pp_workspace(int M, int N, int K) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int sp = pp_splits(M, N, K);
  return sp > 1 ? (int64_t)sp * M * N * 4 : 0;
}

extern "C" int rscotr_gemm_pp(const void* a_planes, int a_ct, int a_col, const void* b_planes, int b_ct, int b_col, float* C,
                              int M, int N, int K, int ldc, const float* bias, int act, const float* aux, float* pre,
                              const float* resid, int accumulate, const float* rowscale, int rows_per_scale, float* out2,
                              float* workspace, int64_t workspace_bytes, int defer, int32_t* splits_out, void* stream) {
  if (M < 0 || N < 0 || K < 0) return fail(RSCOTR_E_SHAPE, "rscotr_gemm_pp: negative dimension");
  if (splits_out) *splits_out = 1;
  if (M == 0 || N == 0) return RSCOTR_OK;
  if (!a_planes || !b_planes || !C || K == 0) return fail(RSCOTR_E_ARG, "rscotr_gemm_pp: null pointer / empty reduction");
  if (!aligned16(a_planes) || !aligned16(b_planes)) return fail(RSCOTR_E_ALIGN, "rscotr_gemm_pp: plane sets must be 16-byte aligned");
  if (act < ACT_NONE || act > ACT_GELU_GRAD) return fail(RSCOTR_E_ARG, "rscotr_gemm_pp: unknown act %d", act);
  if ((act == ACT_RELU_GRAD || act == ACT_GELU_GRAD) && !aux) return fail(RSCOTR_E_ARG, "rscotr_gemm_pp: act %d needs aux", act);
  if (ldc < N) return fail(RSCOTR_E_SHAPE, "rscotr_gemm_pp: leading dimension too small");
  if ((int64_t)M * ldc >= ((int64_t)1 << 31)) return fail(RSCOTR_E_SHAPE, "rscotr_gemm_pp: output of %lld elements (32-bit row offsets in the epilogue)", (long long)M * ldc);
  if (rowscale && rows_per_scale <= 0) return fail(RSCOTR_E_ARG, "rscotr_gemm_pp: rowscale needs rows_per_scale > 0");
  const int need_a = a_col ? (M + 31) / 32 : (K + 31) / 32, need_b = b_col ? (N + 31) / 32 : (K + 31) / 32;
  if (a_ct < need_a || b_ct < need_b)
    return fail(RSCOTR_E_SHAPE, "rscotr_gemm_pp: column tiles (%d, %d) of the plane sets do not cover the operands (%d, %d)", a_ct, b_ct, need_a, need_b);
  if (defer && !splits_out) return fail(RSCOTR_E_ARG, "rscotr_gemm_pp: defer needs splits_out");
  GemmParams p;
  p.A = nullptr; p.B = nullptr; p.C = C; p.bias = bias; p.aux = aux; p.pre = pre; p.resid = resid; p.C2 = out2;
  p.M = M; p.N = N; p.K = K; p.lda = 0; p.ldb = 0; p.ldc = ldc;
  p.act = act; p.accumulate = accumulate;
  p.vecA = p.vecB = 0;
  p.vecC = (ldc % 4 == 0) && aligned16(C) && aligned16(bias) && aligned16(aux) && aligned16(pre) && aligned16(resid) && aligned16(out2);
  p.rowsum = nullptr; p.rowsum_acc = 0; p.rs_slabs = nullptr;
  p.nb1 = 0; p.nb2 = 1;
  p.rowscale = rowscale; p.rows_per = rows_per_scale; p.kscale = nullptr; p.krows_per = 0;
  const int tiles_m = (M + PP_BM - 1) / PP_BM, tiles_n = (N + PP_BN - 1) / PP_BN;
  p.tiles = tiles_m * tiles_n;
  int sp = pp_splits(M, N, K);
  if (sp > 1 && (!workspace || workspace_bytes < (int64_t)sp * M * N * 4))
    sp = workspace ? (int)std::max<int64_t>(1, workspace_bytes / ((int64_t)M * N * 4)) : 1;
  p.splits = sp;
  p.ksplit_len = K;
  p.slabs = sp > 1 ? workspace : nullptr;
  PPOperands o;
  o.A = reinterpret_cast<const char*>(a_planes); o.B = reinterpret_cast<const char*>(b_planes);
  o.ctA = a_ct; o.ctB = b_ct; o.mtA = (M + 31) / 32; o.mtB = (N + 31) / 32; o.tiles_n = tiles_n;
  hipStream_t s = (hipStream_t)stream;
  static bool once = false;
  if (!once) {
#define PP_ATTR(A_, B_, E_) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pp_kernel<A_, B_, E_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PP_LDS_BYTES)
    PP_ATTR(false, false, false); PP_ATTR(false, true, false); PP_ATTR(true, false, false); PP_ATTR(true, true, false);
    PP_ATTR(false, false, true); PP_ATTR(false, true, true); PP_ATTR(true, false, true); PP_ATTR(true, true, true);
#undef PP_ATTR
    once = true;
  }
  const bool edge = M % PP_BM || N % PP_BN;
  static const bool prof_shapes = getenv("RSCOTR_PROF_SHAPES") != nullptr;
  char name[112];
  if (prof_shapes) snprintf(name, sizeof(name), "M=%d N=%d K=%d %d%d pp splits=%d", M, N, K, a_col, b_col, sp);
  else snprintf(name, sizeof(name), "rscotr::gemm_pp_kernel<%s, %s, *>", a_col ? "true" : "false", b_col ? "true" : "false");
  ProfScope prof(PROF_GEMM, 2.0 * M * N * K, s, "%s", name);
  const dim3 grid((unsigned)(p.tiles * sp));
#define PP_LAUNCH(A_, B_)                                                                          \
  do {                                                                                             \
    if (edge) gemm_pp_kernel<A_, B_, true><<<grid, 256, PP_LDS_BYTES, s>>>(p, o);                  \
    else gemm_pp_kernel<A_, B_, false><<<grid, 256, PP_LDS_BYTES, s>>>(p, o);                      \
  } while (0)
  if (!a_col && !b_col) PP_LAUNCH(false, false);
  else if (!a_col) PP_LAUNCH(false, true);
  else if (!b_col) PP_LAUNCH(true, false);
  else PP_LAUNCH(true, true);
#undef PP_LAUNCH
  if (int e = check_launch("rscotr_gemm_pp")) return e;
  if (splits_out) *splits_out = sp;
  if (sp > 1 && !defer) {
    splitk_reduce_launch(p, workspace, s);
    return check_launch("rscotr_gemm_pp (split-K reduce)");
  }
  return RSCOTR_OK;
}
