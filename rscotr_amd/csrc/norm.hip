// LayerNorm forward / backward over the last dimension (gfx950), HBM-streaming kernels.
//
// Replaces torch.nn.LayerNorm as built by the un-vendored layers the reference instantiates:
// mmdet SwinTransformer norm1/norm2/norm{0..3}, PatchEmbed / PatchMerging norms
// (configs/multi/MTL_slvlcls_...potsdam.py:9-25), the `norm` steps of mmcv BaseTransformerLayer in the
// shared encoder and both decoders (:34-50, :76-98, :139-160), enc_output_norm / decoder norm
// (models/multi/bbox_head/transformer.py:38-41,151-158) and post_norm
// (models/multi/seg_head/mask2former_head.py:60-83).  eps = 1e-5 everywhere.
//
// Mapping: a group of G lanes (G = 8..64, power of two >= C/4/NV) owns one row and keeps it in
// registers as NV float4 per lane; a wavefront therefore normalises 64/G rows at once with 16-byte
// coalesced loads, statistics are lane-group butterflies (no LDS).  Two-pass (mean, then centred
// variance) in registers, so one HBM read and one write per element.  Backward keeps the same
// ownership: every lane owns fixed columns, so dgamma/dbeta accumulate in registers across the
// rows a block visits and leave the block as one atomic per column.
#include "common.h"

namespace rscotr {

// Patch-merging mode (MG): the normalised row is mmcv PatchMerging's nn.Unfold(2, stride 2) row of the token map x
// (B, H, W, Cin) — element c * 4 + kh * 2 + kw of output token (b, i, j) is x[b, 2i + kh, 2j + kw, c], zero past H / W —
// gathered by the LOADS of the norm itself: float4 column c of the unfold-ordered row is channel c at the four window
// positions, so a lane reads its columns as four 4-byte loads (one per position, coalesced across the lanes' adjacent
// channels) into the same registers the plain kernel fills with one 16-byte load.  Ownership, summation order, affine,
// dgamma / dbeta and the y / dy accesses are the plain kernels': the result equals LayerNorm of the gathered copy bit for bit.
struct MergeGeom {
  int H, W, Cin;  // input map; output tokens (H + 1) / 2 x (W + 1) / 2, row width 4 Cin
};

// element offsets of the four window positions of output row `row` (-1: past the map, reads as zero)
__device__ __forceinline__ void merge_rows(const MergeGeom& mg, long row, long (&off)[4]) {
  const int Ho = (mg.H + 1) >> 1, Wo = (mg.W + 1) >> 1;
  const int oj = (int)(row % Wo);
  const long t = row / Wo;
  const int oi = (int)(t % Ho);
  const long b = t / Ho;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int yy = 2 * oi + (k >> 1), xx = 2 * oj + (k & 1);
    off[k] = (yy < mg.H && xx < mg.W) ? ((b * mg.H + yy) * mg.W + xx) * (long)mg.Cin : -1;
  }
}

__device__ __forceinline__ float4 merge_load(const float* __restrict__ x, const long (&off)[4], int c) {
  return make_float4(off[0] >= 0 ? x[off[0] + c] : 0.f, off[1] >= 0 ? x[off[1] + c] : 0.f,
                     off[2] >= 0 ? x[off[2] + c] : 0.f, off[3] >= 0 ? x[off[3] + c] : 0.f);
}

template <int G, int NV, bool MG = false>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ b, float* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            int M, int C, float eps,
                                                            const float* __restrict__ add = nullptr, int add_rows = 1,
                                                            float* __restrict__ y2 = nullptr, MergeGeom mg = MergeGeom{},
                                                            unsigned* __restrict__ amax_out = nullptr) {
  // amax_out: value-range word of the output(s) (common.h: amax_commit) — max |y| (and |y2|) for the fp16 split product
  // of the GEMM that multiplies with them
  constexpr int RW = kWave / G;  // rows per wavefront
  float amx = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % G, rw = lane / G;
  const int C4 = C >> 2;
  float4 wv[NV], bv[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = sub + i * G;
    wv[i] = (c < C4 && w) ? reinterpret_cast<const float4*>(w)[c] : make_float4(1.f, 1.f, 1.f, 1.f);
    bv[i] = (c < C4 && b) ? reinterpret_cast<const float4*>(b)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float invC = 1.f / (float)C;
  for (long row0 = ((long)blockIdx.x * 4 + wave) * RW; row0 < M; row0 += (long)gridDim.x * 4 * RW) {
    const long row = row0 + rw;
    const bool ok = row < M;
    float4 v[NV];
    float s = 0.f;
    long off[4];
    if (MG) merge_rows(mg, ok ? row : 0, off);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = sub + i * G;
      if (MG) v[i] = (ok && c < C4) ? merge_load(x, off, c) : make_float4(0.f, 0.f, 0.f, 0.f);
      else v[i] = (ok && c < C4) ? reinterpret_cast<const float4*>(x + row * C)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
      s += v[i].x + v[i].y + v[i].z + v[i].w;
    }
    const float mu = group_sum<G>(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = sub + i * G;
      if (c < C4) {
        const float dx = v[i].x - mu, dy = v[i].y - mu, dz = v[i].z - mu, dw = v[i].w - mu;
        q += dx * dx + dy * dy + dz * dz + dw * dw;
      }
    }
    const float rs = rsqrtf(group_sum<G>(q) * invC + eps);
    if (ok) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = sub + i * G;
        if (c < C4) {
          float4 o;
          o.x = (v[i].x - mu) * rs * wv[i].x + bv[i].x;
          o.y = (v[i].y - mu) * rs * wv[i].y + bv[i].y;
          o.z = (v[i].z - mu) * rs * wv[i].z + bv[i].z;
          o.w = (v[i].w - mu) * rs * wv[i].w + bv[i].w;
          reinterpret_cast<float4*>(y + row * C)[c] = o;
          amx = amax4(amx, o);
          if (y2) {  // second output: y + add[row % add_rows] (the positional embedding the next attention adds to its query)
            const float4 a = reinterpret_cast<const float4*>(add + (row % add_rows) * C)[c];
            const float4 o2 = make_float4(o.x + a.x, o.y + a.y, o.z + a.z, o.w + a.w);
            reinterpret_cast<float4*>(y2 + row * C)[c] = o2;
            amx = amax4(amx, o2);
          }
        }
      }
      if (sub == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
      }
    }
  }
  amax_commit(amax_out, amx);
}

template <int G, int NV, bool MG = false>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy,
                                                            const float* __restrict__ x,
                                                            const float* __restrict__ w,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ dx,
                                                            const float* __restrict__ dres,
                                                            float* __restrict__ part, int M, int C,
                                                            MergeGeom mg = MergeGeom{},
                                                            unsigned* __restrict__ amax_out = nullptr) {
  // no implicit fused multiply-adds in this body: the plain and the patch-merging instantiation must round alike
  // (tests/test_norm_gpu.py holds them bit-equal), whatever the optimiser would contract in each of them
#pragma clang fp contract(off)
  constexpr int RW = kWave / G;
  float amx = 0.f;  // max |dx| -> amax_out (as the forward kernel)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane % G, rw = lane / G;
  const int C4 = C >> 2;
  float4 wv[NV], aw[NV], ab[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = sub + i * G;
    wv[i] = (c < C4 && w) ? reinterpret_cast<const float4*>(w)[c] : make_float4(1.f, 1.f, 1.f, 1.f);
    aw[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float invC = 1.f / (float)C;
  for (long row0 = ((long)blockIdx.x * 4 + wave) * RW; row0 < M; row0 += (long)gridDim.x * 4 * RW) {
    const long row = row0 + rw;
    const bool ok = row < M;
    const float mu = ok ? mean[row] : 0.f, rs = ok ? rstd[row] : 0.f;
    float4 g[NV], xh[NV];
    float s1 = 0.f, s2 = 0.f;
    long off[4];
    if (MG) merge_rows(mg, ok ? row : 0, off);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = sub + i * G;
      const bool in = ok && c < C4;
      const float4 d = in ? reinterpret_cast<const float4*>(dy + row * C)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
      const float4 xv = !in ? make_float4(mu, mu, mu, mu)
                            : (MG ? merge_load(x, off, c) : reinterpret_cast<const float4*>(x + row * C)[c]);
      xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
      ab[i].x += d.x; ab[i].y += d.y; ab[i].z += d.z; ab[i].w += d.w;
      aw[i].x += d.x * xh[i].x; aw[i].y += d.y * xh[i].y; aw[i].z += d.z * xh[i].z; aw[i].w += d.w * xh[i].w;
      g[i] = make_float4(d.x * wv[i].x, d.y * wv[i].y, d.z * wv[i].z, d.w * wv[i].w);
      s1 += g[i].x + g[i].y + g[i].z + g[i].w;
      s2 += g[i].x * xh[i].x + g[i].y * xh[i].y + g[i].z * xh[i].z + g[i].w * xh[i].w;
    }
    const float m1 = group_sum<G>(s1) * invC, m2 = group_sum<G>(s2) * invC;
    if (ok && dx) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = sub + i * G;
        if (c < C4) {
          float4 o;
          o.x = rs * fmaf(-xh[i].x, m2, g[i].x - m1);
          o.y = rs * fmaf(-xh[i].y, m2, g[i].y - m1);
          o.z = rs * fmaf(-xh[i].z, m2, g[i].z - m1);
          o.w = rs * fmaf(-xh[i].w, m2, g[i].w - m1);
          if (!MG) {
            if (dres) {  // gradient of the residual branch that forks at this LayerNorm's input
              const float4 r = reinterpret_cast<const float4*>(dres + row * C)[c];
              o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
            }
            reinterpret_cast<float4*>(dx + row * C)[c] = o;
            amx = amax4(amx, o);
          } else {  // channel c back to its four window positions
            if (off[0] >= 0) dx[off[0] + c] = o.x;
            if (off[1] >= 0) dx[off[1] + c] = o.y;
            if (off[2] >= 0) dx[off[2] + c] = o.z;
            if (off[3] >= 0) dx[off[3] + c] = o.w;
            amx = amax4(amx, o);  // (positions past the map are not stored: still a bound)
          }
        }
      }
    }
  }
  amax_commit(amax_out, amx);
  if (!part) return;
  // fold the RW row-slots of the wavefront, then the 4 wavefronts through LDS, then store this
  // workgroup's partial row: part[blockIdx.x][{dgamma, dbeta}][C].  (Hundreds of workgroups adding
  // atomically into the same C addresses serialise at the memory side.)
  __shared__ float4 red[2][4][NV * 64];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
#pragma unroll
    for (int o = G; o < kWave; o <<= 1) {
      aw[i].x += __shfl_xor(aw[i].x, o, 64); aw[i].y += __shfl_xor(aw[i].y, o, 64);
      aw[i].z += __shfl_xor(aw[i].z, o, 64); aw[i].w += __shfl_xor(aw[i].w, o, 64);
      ab[i].x += __shfl_xor(ab[i].x, o, 64); ab[i].y += __shfl_xor(ab[i].y, o, 64);
      ab[i].z += __shfl_xor(ab[i].z, o, 64); ab[i].w += __shfl_xor(ab[i].w, o, 64);
    }
    if (rw == 0) {
      red[0][wave][sub + i * G] = aw[i];
      red[1][wave][sub + i * G] = ab[i];
    }
  }
  __syncthreads();
  float4* pw = reinterpret_cast<float4*>(part + (long)blockIdx.x * 2 * C);
  float4* pb = reinterpret_cast<float4*>(part + (long)blockIdx.x * 2 * C + C);
  for (int j = threadIdx.x; j < NV * G; j += 256) {
    if (j >= C4) continue;
    float4 a = red[0][0][j], bsum = red[1][0][j];
#pragma unroll
    for (int wv_ = 1; wv_ < 4; ++wv_) {
      const float4 t = red[0][wv_][j], u = red[1][wv_][j];
      a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
      bsum.x += u.x; bsum.y += u.y; bsum.z += u.z; bsum.w += u.w;
    }
    pw[j] = a;
    pb[j] = bsum;
  }
}

// dweight[c] += sum_g part[g][0][c]; dbias[c] += sum_g part[g][1][c].
// One workgroup per 64 columns of the (2C)-wide partial rows: 16 column lanes (one float4 each) x 16 row
// lanes; a row lane strides over the G partial rows (independent 16-byte loads in flight), the 16 row
// lanes fold through LDS.  (The first version walked all G rows serially in one thread per column.)
__global__ __launch_bounds__(256) void layernorm_bwd_final_kernel(const float* __restrict__ part,
                                                                  float* __restrict__ dw, float* __restrict__ db,
                                                                  int G, int C) {
  __shared__ float4 red[16][16];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = (blockIdx.x * 16 + cl) * 4;  // column in [0, 2C), C % 4 == 0
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < 2 * C) {
#pragma unroll 4
    for (int g = rl; g < G; g += 16) {
      const float4 v = *reinterpret_cast<const float4*>(part + (long)g * 2 * C + c);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  red[rl][cl] = a;
  __syncthreads();
  if (rl == 0 && c < 2 * C) {
#pragma unroll
    for (int r = 1; r < 16; ++r) {
      const float4 t = red[r][cl];
      a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
    }
    float* dst = c < C ? dw : db;
    if (dst) {
      float4* d4 = reinterpret_cast<float4*>(dst + (c < C ? c : c - C));
      float4 o = *d4;
      o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
      *d4 = o;
    }
  }
}

// The same fold for MANY pending LayerNorm backward passes in one launch (rscotr_layernorm_flush): workgroup b takes
// 64 columns `wgmap[b].y` of table row `wgmap[b].x` = {partial rows, dweight | 0, dbias | 0, G, C}.
__global__ __launch_bounds__(256) void layernorm_flush_kernel(const int64_t* __restrict__ table,
                                                              const int32_t* __restrict__ wgmap) {
  __shared__ float4 red[16][16];
  const int64_t* t = table + (long)wgmap[2 * blockIdx.x] * 5;
  const float* part = reinterpret_cast<const float*>(t[0]);
  float* dw = reinterpret_cast<float*>(t[1]);
  float* db = reinterpret_cast<float*>(t[2]);
  const int G = (int)t[3], C = (int)t[4];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = (wgmap[2 * blockIdx.x + 1] * 16 + cl) * 4;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < 2 * C) {
#pragma unroll 4
    for (int g = rl; g < G; g += 16) {
      const float4 v = *reinterpret_cast<const float4*>(part + (long)g * 2 * C + c);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  }
  red[rl][cl] = a;
  __syncthreads();
  if (rl == 0 && c < 2 * C) {
#pragma unroll
    for (int r = 1; r < 16; ++r) {
      const float4 u = red[r][cl];
      a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
    }
    float* dst = c < C ? dw : db;
    if (dst) {
      float4* d4 = reinterpret_cast<float4*>(dst + (c < C ? c : c - C));
      float4 o = *d4;
      o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
      *d4 = o;
    }
  }
}

#ifndef RSCOTR_LN_FWD_CAP
#define RSCOTR_LN_FWD_CAP 1024
#endif
#ifndef RSCOTR_LN_BWD_CAP
#define RSCOTR_LN_BWD_CAP 1024  // workgroups (= partial dgamma / dbeta rows) of a backward launch.  512 (rounds 1-5) left the 32768- / 10880- / 8192-row
#endif                          // launches at two wavefronts per SIMD with 5-8 dependent trips each; 1024: -0.05 to -0.2 ms per round by box, 768 /
                                // 1536 / 2048 within noise of it, a forward cap of 2048-4096 no gain (profiles/README.md)
static int ln_blocks(int M, int rows_per_block, long cap = RSCOTR_LN_FWD_CAP) {
  long b = ((long)M + rows_per_block - 1) / rows_per_block;
  return (int)std::max<long>(1, std::min<long>(b, cap));
}

}  // namespace rscotr

using namespace rscotr;

// (G, NV) with G*NV*4 >= C, G a power of two in [8, 64], NV <= 8.
#define RSCOTR_LN_DISPATCH(C, CALL)                       \
  do {                                                    \
    const int c4 = (C) >> 2;                              \
    if (c4 <= 8) { CALL(8, 1); }                          \
    else if (c4 <= 16) { CALL(16, 1); }                   \
    else if (c4 <= 32) { CALL(32, 1); }                   \
    else if (c4 <= 64) { CALL(64, 1); }                   \
    else if (c4 <= 128) { CALL(64, 2); }                  \
    else if (c4 <= 192) { CALL(64, 3); }                  \
    else if (c4 <= 256) { CALL(64, 4); }                  \
    else if (c4 <= 384) { CALL(64, 6); }                  \
    else { CALL(64, 8); }                                 \
  } while (0)

extern "C" int rscotr_layernorm_fwd(const float* x, const float* weight, const float* bias, float* y,
                                    float* mean, float* rstd, int M, int C, float eps, uint32_t* amax_out, void* stream) {
  if (M < 0 || C <= 0) return fail(RSCOTR_E_SHAPE, "rscotr_layernorm_fwd: bad shape M=%d C=%d", M, C);
  if (C % 4 != 0 || C > 2048) return fail(RSCOTR_E_SHAPE, "rscotr_layernorm_fwd: C=%d must be a multiple of 4, <= 2048", C);
  if (M == 0) return RSCOTR_OK;
  if (!x || !y) return fail(RSCOTR_E_ARG, "rscotr_layernorm_fwd: null pointer");
  if (!aligned16(x) || !aligned16(y) || (weight && !aligned16(weight)) || (bias && !aligned16(bias)))
    return fail(RSCOTR_E_ALIGN, "rscotr_layernorm_fwd: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(PROF_HBM, 8.0 * M * C, s, "rscotr::layernorm_fwd_kernel");
#define CALL(G, NV)                                                                                       \
  layernorm_fwd_kernel<G, NV><<<ln_blocks(M, 4 * (64 / G)), 256, 0, s>>>(x, weight, bias, y, mean, rstd, M, C, eps, nullptr, 1, \
                                                                         nullptr, MergeGeom{}, amax_out)
  RSCOTR_LN_DISPATCH(C, CALL);
#undef CALL
  return check_launch("rscotr_layernorm_fwd");
}

extern "C" int rscotr_layernorm_fwd_sum(const float* x, const float* weight, const float* bias, float* y, float* mean,
                                        float* rstd, const float* add, int add_rows, float* y2, int M, int C, float eps,
                                        uint32_t* amax_out, void* stream) {
  if (M < 0 || C <= 0 || add_rows <= 0) return fail(RSCOTR_E_SHAPE, "rscotr_layernorm_fwd_sum: bad shape M=%d C=%d add_rows=%d", M, C, add_rows);
  if (C % 4 != 0 || C > 2048) return fail(RSCOTR_E_SHAPE, "rscotr_layernorm_fwd_sum: C=%d must be a multiple of 4, <= 2048", C);
  if (M == 0) return RSCOTR_OK;
  if (!x || !y || !add || !y2) return fail(RSCOTR_E_ARG, "rscotr_layernorm_fwd_sum: null pointer");
  if (!aligned16(x) || !aligned16(y) || !aligned16(add) || !aligned16(y2) || (weight && !aligned16(weight)) ||
      (bias && !aligned16(bias)))
    return fail(RSCOTR_E_ALIGN, "rscotr_layernorm_fwd_sum: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(PROF_HBM, 12.0 * M * C, s, "rscotr::layernorm_fwd_kernel");
#define CALL(G, NV)                                                                                                  \
  layernorm_fwd_kernel<G, NV><<<ln_blocks(M, 4 * (64 / G)), 256, 0, s>>>(x, weight, bias, y, mean, rstd, M, C, eps, add, \
                                                                         add_rows, y2, MergeGeom{}, amax_out)
  RSCOTR_LN_DISPATCH(C, CALL);
#undef CALL
  return check_launch("rscotr_layernorm_fwd_sum");
}

static int ln_bwd_blocks(int M, int C) {
  const int c4 = C >> 2;
  const int G = c4 <= 8 ? 8 : c4 <= 16 ? 16 : c4 <= 32 ? 32 : 64;
  return ln_blocks(M, 4 * (64 / G), RSCOTR_LN_BWD_CAP);
}

extern "C" int64_t rscotr_layernorm_bwd_workspace(int M, int C) {
  if (M <= 0 || C <= 0) return 0;
  return (int64_t)ln_bwd_blocks(M, C) * 2 * C * 4;
}

// dweight / dbias are ACCUMULATED into: the caller zeroes them (or passes a gradient buffer to add
// to).  Any of dx / dweight / dbias may be null; `workspace` (rscotr_layernorm_bwd_workspace() bytes,
// 16-byte aligned) is required when dweight or dbias is given.
// `dx_add` (M,C) or null: added to dx on the way out (the gradient of a residual branch forking at the input).
extern "C" int rscotr_layernorm_bwd(const float* dy, const float* x, const float* weight, const float* mean,
                                    const float* rstd, float* dx, const float* dx_add, float* dweight, float* dbias, int M, int C,
                                    float* workspace, int64_t workspace_bytes, uint32_t* amax_out, void* stream) {
  if (M < 0 || C <= 0) return fail(RSCOTR_E_SHAPE, "rscotr_layernorm_bwd: bad shape M=%d C=%d", M, C);
  if (C % 4 != 0 || C > 2048) return fail(RSCOTR_E_SHAPE, "rscotr_layernorm_bwd: C=%d must be a multiple of 4, <= 2048", C);
  if (M == 0) return RSCOTR_OK;
  if (!dy || !x || !mean || !rstd) return fail(RSCOTR_E_ARG, "rscotr_layernorm_bwd: null pointer");
  if (!aligned16(dy) || !aligned16(x) || (dx && !aligned16(dx)) || (weight && !aligned16(weight)))
    return fail(RSCOTR_E_ALIGN, "rscotr_layernorm_bwd: pointers must be 16-byte aligned");
  const bool params = dweight || dbias;
  if ((dweight && !aligned16(dweight)) || (dbias && !aligned16(dbias)))
    return fail(RSCOTR_E_ALIGN, "rscotr_layernorm_bwd: dweight / dbias must be 16-byte aligned");
  const int nb = ln_bwd_blocks(M, C);
  if (params && (!workspace || !aligned16(workspace) || workspace_bytes < (int64_t)nb * 2 * C * 4))
    return fail(RSCOTR_E_ARG, "rscotr_layernorm_bwd: workspace of rscotr_layernorm_bwd_workspace() bytes required");
  hipStream_t s = (hipStream_t)stream;
  float* part = params ? workspace : nullptr;
  ProfScope prof(PROF_HBM, (dx_add ? 16.0 : 12.0) * M * C, s, "rscotr::layernorm_bwd_kernel");
#define CALL(G, NV) \
  layernorm_bwd_kernel<G, NV><<<nb, 256, 0, s>>>(dy, x, weight, mean, rstd, dx, dx_add, part, M, C, MergeGeom{}, amax_out)
  RSCOTR_LN_DISPATCH(C, CALL);
#undef CALL
  if (int e = check_launch("rscotr_layernorm_bwd")) return e;
  if (params) {
    layernorm_bwd_final_kernel<<<(2 * C + 63) / 64, 256, 0, s>>>(part, dweight, dbias, nb, C);
    return check_launch("rscotr_layernorm_bwd (final)");
  }
  return RSCOTR_OK;
}

// Backward without the parameter-gradient fold: the per-workgroup partial rows ([G][2C], G = workspace bytes / (8 C))
// stay in `part` (caller-owned until rscotr_layernorm_flush), dx is written as usual.
extern "C" int rscotr_layernorm_bwd_partials(const float* dy, const float* x, const float* weight, const float* mean,
                                             const float* rstd, float* dx, const float* dx_add, int M, int C, float* part, int64_t part_bytes,
                                             uint32_t* amax_out, void* stream) {
  if (M <= 0 || C <= 0) return fail(RSCOTR_E_SHAPE, "rscotr_layernorm_bwd_partials: bad shape M=%d C=%d", M, C);
  if (C % 4 != 0 || C > 2048) return fail(RSCOTR_E_SHAPE, "rscotr_layernorm_bwd_partials: C=%d must be a multiple of 4, <= 2048", C);
  if (!dy || !x || !mean || !rstd || !part) return fail(RSCOTR_E_ARG, "rscotr_layernorm_bwd_partials: null pointer");
  if (!aligned16(dy) || !aligned16(x) || (dx && !aligned16(dx)) || (dx_add && !aligned16(dx_add)) || (weight && !aligned16(weight)) || !aligned16(part))
    return fail(RSCOTR_E_ALIGN, "rscotr_layernorm_bwd_partials: pointers must be 16-byte aligned");
  const int nb = ln_bwd_blocks(M, C);
  if (part_bytes < (int64_t)nb * 2 * C * 4)
    return fail(RSCOTR_E_ARG, "rscotr_layernorm_bwd_partials: region of rscotr_layernorm_bwd_workspace() bytes required");
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(PROF_HBM, (dx_add ? 16.0 : 12.0) * M * C, s, "rscotr::layernorm_bwd_kernel");
#define CALL(G, NV) \
  layernorm_bwd_kernel<G, NV><<<nb, 256, 0, s>>>(dy, x, weight, mean, rstd, dx, dx_add, part, M, C, MergeGeom{}, amax_out)
  RSCOTR_LN_DISPATCH(C, CALL);
#undef CALL
  return check_launch("rscotr_layernorm_bwd_partials");
}

// ---- mmcv PatchMerging's unfold + LayerNorm(4 Cin) as one launch per direction (MG instantiations above) ----------------
static int pm_bwd_blocks(long M, int Cin) { return ln_bwd_blocks((int)M, 4 * Cin); }

static int pm_check(const char* who, int B, int H, int W, int Cin, long* M) {
  if (B < 0 || H <= 0 || W <= 0 || Cin <= 0) return fail(RSCOTR_E_SHAPE, "%s: bad shape B=%d H=%d W=%d Cin=%d", who, B, H, W, Cin);
  if (Cin > 512) return fail(RSCOTR_E_SHAPE, "%s: Cin=%d must be <= 512", who, Cin);
  *M = (long)B * ((H + 1) / 2) * ((W + 1) / 2);
  if (*M > 0x7fffffffL) return fail(RSCOTR_E_SHAPE, "%s: too many output tokens", who);
  return RSCOTR_OK;
}

extern "C" int rscotr_patch_merge_norm_fwd(const float* x, const float* weight, const float* bias, float* y, float* mean,
                                           float* rstd, int B, int H, int W, int Cin, float eps, uint32_t* amax_out, void* stream) {
  long M;
  if (int e = pm_check("rscotr_patch_merge_norm_fwd", B, H, W, Cin, &M)) return e;
  if (M == 0) return RSCOTR_OK;
  if (!x || !y) return fail(RSCOTR_E_ARG, "rscotr_patch_merge_norm_fwd: null pointer");
  if (!aligned16(x) || !aligned16(y) || (weight && !aligned16(weight)) || (bias && !aligned16(bias)))
    return fail(RSCOTR_E_ALIGN, "rscotr_patch_merge_norm_fwd: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const MergeGeom mg{H, W, Cin};
  const int C = 4 * Cin;
#define CALL(G, NV)                                                                                                     \
  layernorm_fwd_kernel<G, NV, true><<<ln_blocks((int)M, 4 * (64 / G)), 256, 0, s>>>(x, weight, bias, y, mean, rstd, (int)M, C, \
                                                                                    eps, nullptr, 1, nullptr, mg, amax_out)
  RSCOTR_LN_DISPATCH(C, CALL);
#undef CALL
  return check_launch("rscotr_patch_merge_norm_fwd");
}

extern "C" int64_t rscotr_patch_merge_norm_bwd_workspace(int B, int H, int W, int Cin) {
  long M;
  if (pm_check("rscotr_patch_merge_norm_bwd_workspace", B, H, W, Cin, &M) || M == 0) return 0;
  return (int64_t)pm_bwd_blocks(M, Cin) * 2 * (4 * Cin) * 4;
}

extern "C" int rscotr_patch_merge_norm_bwd(const float* dy, const float* x, const float* weight, const float* mean,
                                           const float* rstd, float* dx, float* dweight, float* dbias, int B, int H, int W,
                                           int Cin, float* workspace, int64_t workspace_bytes, int fold, uint32_t* amax_out, void* stream) {
  long M;
  if (int e = pm_check("rscotr_patch_merge_norm_bwd", B, H, W, Cin, &M)) return e;
  if (M == 0) return RSCOTR_OK;
  if (!dy || !x || !mean || !rstd) return fail(RSCOTR_E_ARG, "rscotr_patch_merge_norm_bwd: null pointer");
  if (!aligned16(dy) || !aligned16(x) || (dx && !aligned16(dx)) || (weight && !aligned16(weight)) ||
      (dweight && !aligned16(dweight)) || (dbias && !aligned16(dbias)))
    return fail(RSCOTR_E_ALIGN, "rscotr_patch_merge_norm_bwd: pointers must be 16-byte aligned");
  const int C = 4 * Cin;
  const bool params = !fold || dweight || dbias;
  const int nb = pm_bwd_blocks(M, Cin);
  if (params && (!workspace || !aligned16(workspace) || workspace_bytes < (int64_t)nb * 2 * C * 4))
    return fail(RSCOTR_E_ARG, "rscotr_patch_merge_norm_bwd: workspace of rscotr_patch_merge_norm_bwd_workspace() bytes required");
  hipStream_t s = (hipStream_t)stream;
  float* part = params ? workspace : nullptr;
  const MergeGeom mg{H, W, Cin};
#define CALL(G, NV) \
  layernorm_bwd_kernel<G, NV, true><<<nb, 256, 0, s>>>(dy, x, weight, mean, rstd, dx, nullptr, part, (int)M, C, mg, amax_out)
  RSCOTR_LN_DISPATCH(C, CALL);
#undef CALL
  if (int e = check_launch("rscotr_patch_merge_norm_bwd")) return e;
  if (fold && params) {
    layernorm_bwd_final_kernel<<<(2 * C + 63) / 64, 256, 0, s>>>(part, dweight, dbias, nb, C);
    return check_launch("rscotr_patch_merge_norm_bwd (final)");
  }
  return RSCOTR_OK;
}

// table: device (n, 5) int64 rows {partial rows, dweight | 0, dbias | 0, G, C}; wgmap: device (nwg, 2) int32 rows
// {table row, block of 64 of the 2C partial columns}, ceil(2C / 64) blocks per row.  dweight / dbias are ADDED to.
extern "C" int rscotr_layernorm_flush(const int64_t* table, const int32_t* wgmap, int nwg, void* stream) {
  if (nwg < 0) return fail(RSCOTR_E_SHAPE, "rscotr_layernorm_flush: negative workgroup count");
  if (nwg == 0) return RSCOTR_OK;
  if (!table || !wgmap) return fail(RSCOTR_E_ARG, "rscotr_layernorm_flush: null pointer");
  layernorm_flush_kernel<<<dim3((unsigned)nwg), 256, 0, (hipStream_t)stream>>>(table, wgmap);
  return check_launch("rscotr_layernorm_flush");
}
