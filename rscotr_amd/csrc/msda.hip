// Multi-scale deformable attention (MSDA) sampling kernels for gfx950.
//
// Replaces the operator the reference reaches through mmcv's
//   MultiScaleDeformableAttnFunction.apply(value, spatial_shapes, level_start_index,
//                                          sampling_locations, attention_weights, im2col_step)
// (ext_module.ms_deform_attn_forward / _backward), called from
//   /root/reference/models/multi/seg_head/pixel_decoder.py:134-146   (shared encoder, seg)
//   /root/reference/models/multi/bbox_head/transformer.py:211-221    (shared encoder, det)
//   /root/reference/models/multi/bbox_head/transformer.py:258-269    (DINO decoder cross-attn)
//
//   out[b,q,h,:] = sum_{l<L} sum_{p<P} A[b,q,h,l,p] * bilinear(V_l[b,:,h,:], loc[b,q,h,l,p,:])
// with pixel coords x = loc_x*W_l - 0.5, y = loc_y*H_l - 0.5, zero padding outside the map
// (== F.grid_sample(bilinear, zeros, align_corners=False)).
//
// CDNA4 mapping (not a CUDA one-thread-per-channel translation):
//   * one (b, q-tile, head) per 256-thread workgroup; G = D/4 lanes hold the D channels of
//     one query as float4, so a wavefront covers 64/G queries and every tap is one 16-byte
//     load per lane = whole 128-byte lines per query (D = 32);
//   * blockIdx % H == head: with H = 8 heads and the dispatcher's round-robin over the
//     8 XCDs, each XCD's private 4 MiB L2 only ever sees ONE head's 128-byte slice of every
//     value token (680 KB per image at N = 5440), so the 16x4 tap re-reads are L2 hits;
//   * sampling locations / attention weights for the tile are staged once through LDS with
//     coalesced loads and re-read as LDS broadcasts by the G lanes of a query;
//   * backward: per-lane partial sums over 4 channels, lane-group butterfly (ds_swizzle /
//     DPP via __shfl_xor) over the G lanes, results gathered in LDS and written back
//     coalesced; grad_value scatter uses the hardware fp32 atomic (global_atomic_add_f32).
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "rscotr.h"

namespace rscotr {

struct Bilinear {
  int i1, i2, i3, i4;      // token offsets inside the level (valid only if ok*)
  bool ok1, ok2, ok3, ok4;  // tap inside the map
  bool in;                  // sample inside (-1, size) on both axes
  float hh, hw, lh, lw;
  int h_low, w_low;         // top-left tap (-1 .. size - 1 when `in`)
};

// Pixel coordinate of a normalised sampling location: loc * extent - 0.5 as ONE explicitly fused multiply-add.  Every kernel that
// derives a bin (floor) or a fractional weight from a location calls this: the bin written by the sample kernel and the
// weights re-derived by the tile kernel must come from the same float, whatever the compiler would contract on its own (a
// sample within one ulp of an integer coordinate would otherwise land in bin n with a weight of ~0 instead of ~1: ADVICE r3)
__device__ __forceinline__ float msda_pix(float loc, int extent) { return __fmaf_rn(loc, (float)extent, -0.5f); }

__device__ __forceinline__ Bilinear bilinear_setup(float lx, float ly, int Hl, int Wl) {
  Bilinear t;
  const float h_im = msda_pix(ly, Hl);
  const float w_im = msda_pix(lx, Wl);
  t.in = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)Hl) && (w_im < (float)Wl);
  const float hf = floorf(h_im), wf = floorf(w_im);
  const int h_low = t.in ? (int)hf : 0, w_low = t.in ? (int)wf : 0;
  const int h_high = h_low + 1, w_high = w_low + 1;
  t.h_low = h_low;
  t.w_low = w_low;
  t.lh = h_im - hf;
  t.lw = w_im - wf;
  t.hh = 1.f - t.lh;
  t.hw = 1.f - t.lw;
  if (!t.in) t.lh = t.lw = t.hh = t.hw = 0.f;  // sample skipped entirely (also NaN/inf locations)
  t.ok1 = t.in && h_low >= 0 && w_low >= 0;
  t.ok2 = t.in && h_low >= 0 && w_high <= Wl - 1;
  t.ok3 = t.in && h_high <= Hl - 1 && w_low >= 0;
  t.ok4 = t.in && h_high <= Hl - 1 && w_high <= Wl - 1;
  t.i1 = h_low * Wl + w_low;
  t.i2 = t.i1 + 1;
  t.i3 = t.i1 + Wl;
  t.i4 = t.i3 + 1;
  return t;
}

__device__ __forceinline__ float4 ld4(const float* p, bool ok) {
  return ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
// One sample of a (query, head) as the gather loop wants it: the bilinear set-up is done ONCE per sample by the thread that
// stages it — the G = D / 4 lanes that share a (query, head) used to repeat it, ~50 of the ~85 instructions a lane issued per
// sample, and the launch was bound by instruction issue (22 us of issue on the whole chip for the 1.39 M samples of an
// encoder call), not by the L2.  Same arithmetic on the same inputs: results are bit-identical.
struct alignas(16) MsdaSample {
  float aw, w1, w2, w3;  // attention weight; bilinear weights of the four taps
  float w4;
  int e1;                // element offset of the top-left tap's channel row from the (image, head) base (level start included)
  int ok;                // bit t: tap t lies inside the map
  int pad;
};

// the backward sample kernel's record: the fractional weights themselves (its derivative terms want them)
struct alignas(16) MsdaSampleB {
  float aw, hh, hw, lh;
  float lw;
  int e1;   // element offset of the top-left tap's channel row from the LEVEL's first token
  int ok;   // bits 0-3: tap inside the map; bit 4: the sample counts (inside (-1, size) on both axes)
  int pad;
};

#ifndef RSCOTR_MSDA_FWD_DEDUP
#define RSCOTR_MSDA_FWD_DEDUP 1  // (0: the per-lane set-up of rounds 1-4, for A/B builds)
#endif

// PREP (rscotr_msda_fwd_prep; L * P == 16): the kernel does the element-wise prologue of the attention module itself — the softmax
// over the 16 logits of a (query, head) and the location arithmetic, by the 16 consecutive threads that stage its samples — and
// leaves loc / attn in global memory for the backward, instead of reading what msda_prep_fwd_kernel wrote a launch earlier.
struct MsdaPrepIn {
  const float* off;    // raw sampling offsets, row (b, q) at (b Nq + q) ld_off, head h at + h L P 2
  const float* logit;  // raw attention logits, row (b, q) at (b Nq + q) ld_logit, head h at + h L P
  const float* ref;    // reference points (B, Nq, ref_levels, refdim)
  const float* norm;   // (L, 2) = (W_l, H_l) for 2-d reference points
  float* loc;          // out (B, Nq, H, L, P, 2)
  float* attn;         // out (B, Nq, H, L, P)
  int ld_off, ld_logit, refdim, ref_levels;
};

template <int D, int P, bool DEDUP = true, bool PREP = false>
__global__ __launch_bounds__(256) void msda_fwd_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const float* __restrict__ loc,
    const float* __restrict__ attn, float* __restrict__ out, int Nk, int Nq, int H, int L,
    int ntiles, MsdaPrepIn pi = MsdaPrepIn{}) {
  static_assert(!PREP || DEDUP, "the prologue rides the record staging");
  constexpr int G = D / 4;        // lanes per (query, head)
  constexpr int QW = kWave / G;   // queries per wavefront
  constexpr int QB = 4 * QW;      // queries per workgroup
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int LP = L * P;

  const int bid = blockIdx.x;
  const int h = bid % H;
  const int t = bid / H;
  const int tile = t % ntiles;
  const int b = t / ntiles;
  const int q0 = tile * QB;
  const int tid = threadIdx.x;
  const int tok_stride = H * D;
  if constexpr (DEDUP) {
  MsdaSample* recs = reinterpret_cast<MsdaSample*>(smem);  // [QB][LP]
  // stage the tile's samples: locations + weights read in coalesced 128-byte rows, set up once, left as records
  for (int i = tid; i < QB * LP; i += 256) {
    const int r = i / LP, s_ = i - r * LP;
    const int q = q0 + r;
    MsdaSample m;
    m.aw = m.w1 = m.w2 = m.w3 = m.w4 = 0.f;
    m.e1 = m.ok = m.pad = 0;
    float2 xy = make_float2(0.f, 0.f);
    float aw_ = 0.f;
    const long e = (((long)b * Nq + q) * H + h) * LP + s_;
    const int l = s_ / P;
    if constexpr (PREP) {  // (msda_prep_fwd_kernel<16>'s arithmetic: whole 16-lane groups stay together for the shuffles)
      const bool in = q < Nq;
      const long bq = (long)b * Nq + (in ? q : 0);
      const float lg = in ? pi.logit[bq * pi.ld_logit + h * LP + s_] : -3.0e38f;
      if (in) {
        const float* rp = pi.ref + (bq * pi.ref_levels + (pi.ref_levels > 1 ? l : 0)) * pi.refdim;
        const float2 o = *reinterpret_cast<const float2*>(pi.off + bq * pi.ld_off + (h * LP + s_) * 2);
        xy = msda_location(rp, o, pi.norm, l, P, pi.refdim);
        reinterpret_cast<float2*>(pi.loc)[e] = xy;
      }
      const float mx = group_max<16>(lg);
      const float ex = in ? expf(lg - mx) : 0.f;
      const float sum = group_sum<16>(ex);
      if (in) { aw_ = ex / sum; pi.attn[e] = aw_; }
    } else if (q < Nq) {
      xy = *reinterpret_cast<const float2*>(loc + e * 2);
      aw_ = attn[e];
    }
    if (q < Nq) {
      const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
      const Bilinear g = bilinear_setup(xy.x, xy.y, Hl, Wl);
      m.aw = aw_;
      m.w1 = g.hh * g.hw; m.w2 = g.hh * g.lw; m.w3 = g.lh * g.hw; m.w4 = g.lh * g.lw;
      m.e1 = ((int)lsi[l] + g.i1) * tok_stride;
      m.ok = (g.ok1 ? 1 : 0) | (g.ok2 ? 2 : 0) | (g.ok3 ? 4 : 0) | (g.ok4 ? 8 : 0);
    }
    recs[i] = m;
  }
  __syncthreads();

  const int lane = tid & 63, w = tid >> 6;
  const int r = w * QW + lane / G;
  const int sub = lane % G;
  const int q = q0 + r;
  if (q >= Nq) return;

  const float* vb = value + ((long)b * Nk * H + h) * D + sub * 4;  // + element offset of a token's channel row
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const MsdaSample* mine = recs + r * LP;
  for (int l = 0; l < L; ++l) {
    const int rowstep = (int)shapes[2 * l + 1] * tok_stride;
    float4 ra[P], rb[P];
    float4 v1[P], v2[P], v3[P], v4[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float4* rp = reinterpret_cast<const float4*>(mine + l * P + p);
      ra[p] = rp[0];
      rb[p] = rp[1];
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const int ok = __float_as_int(rb[p].z);
      const float* t1 = vb + __float_as_int(rb[p].y);
      v1[p] = ld4(t1, ok & 1);
      v2[p] = ld4(t1 + tok_stride, ok & 2);
      v3[p] = ld4(t1 + rowstep, ok & 4);
      v4[p] = ld4(t1 + rowstep + tok_stride, ok & 8);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float aw = ra[p].x, w1 = ra[p].y, w2 = ra[p].z, w3 = ra[p].w, w4 = rb[p].x;
      acc.x += aw * (w1 * v1[p].x + w2 * v2[p].x + w3 * v3[p].x + w4 * v4[p].x);
      acc.y += aw * (w1 * v1[p].y + w2 * v2[p].y + w3 * v3[p].y + w4 * v4[p].y);
      acc.z += aw * (w1 * v1[p].z + w2 * v2[p].z + w3 * v3[p].z + w4 * v4[p].z);
      acc.w += aw * (w1 * v1[p].w + w2 * v2[p].w + w3 * v3[p].w + w4 * v4[p].w);
    }
  }
  *reinterpret_cast<float4*>(out + (((long)b * Nq + q) * H + h) * D + sub * 4) = acc;
  } else {  // the per-lane set-up (records of a tile past 48 KB of LDS, or element offsets past 2^31)
  float* s_loc = smem;                // [QB][LP*2]
  float* s_attn = smem + QB * LP * 2;  // [QB][LP]

  // stage sampling locations + attention weights of the tile (coalesced 128-byte rows)
  for (int i = tid; i < QB * LP * 2; i += 256) {
    const int r = i / (LP * 2), c = i - r * (LP * 2);
    const int q = q0 + r;
    s_loc[i] = (q < Nq) ? loc[(((long)b * Nq + q) * H + h) * (LP * 2) + c] : 0.f;
  }
  for (int i = tid; i < QB * LP; i += 256) {
    const int r = i / LP, c = i - r * LP;
    const int q = q0 + r;
    s_attn[i] = (q < Nq) ? attn[(((long)b * Nq + q) * H + h) * LP + c] : 0.f;
  }
  __syncthreads();

  const int lane = tid & 63, w = tid >> 6;
  const int r = w * QW + lane / G;
  const int sub = lane % G;
  const int q = q0 + r;
  if (q >= Nq) return;

  const float* vb = value + ((long)b * Nk * H + h) * D + sub * 4;  // + token*H*D
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* my_loc = s_loc + r * LP * 2;
  const float* my_attn = s_attn + r * LP;

  for (int l = 0; l < L; ++l) {
    const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
    const float* vl = vb + (long)lsi[l] * tok_stride;
    Bilinear g[P];
    float aw[P];
    float4 v1[P], v2[P], v3[P], v4[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float2 xy = *reinterpret_cast<const float2*>(my_loc + (l * P + p) * 2);
      aw[p] = my_attn[l * P + p];
      g[p] = bilinear_setup(xy.x, xy.y, Hl, Wl);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      v1[p] = ld4(vl + (long)g[p].i1 * tok_stride, g[p].ok1);
      v2[p] = ld4(vl + (long)g[p].i2 * tok_stride, g[p].ok2);
      v3[p] = ld4(vl + (long)g[p].i3 * tok_stride, g[p].ok3);
      v4[p] = ld4(vl + (long)g[p].i4 * tok_stride, g[p].ok4);
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float w1 = g[p].hh * g[p].hw, w2 = g[p].hh * g[p].lw;
      const float w3 = g[p].lh * g[p].hw, w4 = g[p].lh * g[p].lw;
      acc.x += aw[p] * (w1 * v1[p].x + w2 * v2[p].x + w3 * v3[p].x + w4 * v4[p].x);
      acc.y += aw[p] * (w1 * v1[p].y + w2 * v2[p].y + w3 * v3[p].y + w4 * v4[p].y);
      acc.z += aw[p] * (w1 * v1[p].z + w2 * v2[p].z + w3 * v3[p].z + w4 * v4[p].z);
      acc.w += aw[p] * (w1 * v1[p].w + w2 * v2[p].w + w3 * v3[p].w + w4 * v4[p].w);
    }
  }
  *reinterpret_cast<float4*>(out + (((long)b * Nq + q) * H + h) * D + sub * 4) = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void atomic_add4(float* p, float4 v, bool ok) {
  if (ok) {
    unsafeAtomicAdd(p + 0, v.x);
    unsafeAtomicAdd(p + 1, v.y);
    unsafeAtomicAdd(p + 2, v.z);
    unsafeAtomicAdd(p + 3, v.w);
  }
}

__device__ __forceinline__ float dot4(float4 a, float4 b) {
  return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}

__device__ __forceinline__ float4 scale4(float4 a, float s) {
  return make_float4(a.x * s, a.y * s, a.z * s, a.w * s);
}

// Reduce-scatter over a lane group (see msda_bwd_kernel): one butterfly step at lane offset O on N live values per lane; the
// lane whose `sub & O` is clear keeps the first ceil(N / 2) values, its partner the rest (zero-padded), each adding what the
// other sends.  rs_final<N, O>() = values per lane after the steps O, O / 2, ..., 1.
template <int N, int O>
constexpr int rs_final() {
  if constexpr (O == 0) return N; else return rs_final<(N + 1) / 2, O / 2>();
}
template <int N0, int N, int O>
__device__ __forceinline__ void rs_steps(float (&cur)[N0], int sub, int& base, int& rend) {
  if constexpr (O > 0) {
    constexpr int KEEP = (N + 1) / 2;
    const bool hi = (sub & O) != 0;
#pragma unroll
    for (int i = 0; i < KEEP; ++i) {  // (writes slots < KEEP only: slot i + KEEP is still this step's input)
      const float lo_v = cur[i], hi_v = (i + KEEP < N) ? cur[i + KEEP] : 0.f;
      const float mine = hi ? hi_v : lo_v, other = hi ? lo_v : hi_v;
      cur[i] = mine + __shfl_xor(other, O, 64);
    }
    if (hi) base += KEEP; else rend = min(rend, base + KEEP);
    rs_steps<N0, KEEP, O / 2>(cur, sub, base, rend);
  }
}

// Tile geometry the sample kernel needs for the BLOCK MASKS of the tile-accumulation backward (see that section): per level,
// reciprocal tile edges (in bins) and tiles per row.  mask[(b h, level, query tile of the sample kernel)] has bit
// (tile & 63) set iff one of the block's samples has its bin in that tile: the tile workgroups skip the other blocks.
struct MsdaMaskGeom {
  float itx[8], ity[8];
  int ntx[8];
};

// SCATTER: 0 = grad_loc / grad_attn only (grad_value comes from the pull kernel), 1 = also scatter grad_value with
// atomics, 2 = scatter iff the level pyramid has more than `bins_cap` extended bins (the sorted path stood down).
// TILE: also leave what the tile-accumulation backward needs (see that section): one bin word per sample + the block masks.
template <int D, int P, int SCATTER, bool TILE = false>
__global__ __launch_bounds__(256, P <= 4 && SCATTER == 0 ? 4 : 2) void msda_bwd_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const float* __restrict__ loc,
    const float* __restrict__ attn, const float* __restrict__ grad_out,
    float* __restrict__ grad_value, float* __restrict__ grad_loc, float* __restrict__ grad_attn,
    int* __restrict__ binw, unsigned long long* __restrict__ mask, MsdaMaskGeom MG, int Nk, int Nq,
    int H, int L, int ntiles, int bins_cap) {
  constexpr int G = D / 4;
  constexpr int QW = kWave / G;
  constexpr int QB = 4 * QW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int LP = L * P;
  bool scatter = SCATTER == 1;
  if (SCATTER == 2) {
    int NE = 0;
    for (int l = 0; l < L; ++l) NE += ((int)shapes[2 * l] + 1) * ((int)shapes[2 * l + 1] + 1);
    scatter = NE > bins_cap;
  }
  // one record per sample (the set-up is done ONCE, by the thread that stages the sample — as in the forward kernel: the G lanes
  // of a (query, head) used to repeat it inside the gather loop), the gradients gathered for a coalesced store, and (TILE) one
  // 4-byte bin word per sample (bin of the top-left tap on the extended grid | -1), staged [L][QB][P]
  MsdaSampleB* recs = reinterpret_cast<MsdaSampleB*>(smem);  // [QB][LP]
  float* s_gattn = smem + QB * LP * 8;       // [QB][LP]    out: grad_attn
  float* s_gloc = smem + QB * LP * 9;        // [QB][LP*2]  out: grad_loc
  int* s_bin = reinterpret_cast<int*>(smem + QB * LP * 11);
  unsigned* s_mask = reinterpret_cast<unsigned*>(smem + QB * LP * 12);  // [L][2] (TILE only)
  if (TILE && threadIdx.x < 2 * L) s_mask[threadIdx.x] = 0u;  // (ordered before the atomics by the barrier below)

  const int bid = blockIdx.x;
  const int h = bid % H;
  const int t = bid / H;
  const int tile = t % ntiles;
  const int b = t / ntiles;
  const int q0 = tile * QB;
  const int tid = threadIdx.x;
  const int tok_stride = H * D;
  if (TILE) __syncthreads();

  for (int i = tid; i < QB * LP; i += 256) {
    const int rr = i / LP, c = i - rr * LP, l = c / P, pp = c - l * P;
    const int q = q0 + rr;
    MsdaSampleB m;
    m.aw = m.hh = m.hw = m.lh = m.lw = 0.f;
    m.e1 = m.ok = m.pad = 0;
    bool in = false;
    int h_low = 0, w_low = 0;
    if (q < Nq) {
      const long e = (((long)b * Nq + q) * H + h) * LP + c;
      const float2 xy = *reinterpret_cast<const float2*>(loc + e * 2);
      const Bilinear gg = bilinear_setup(xy.x, xy.y, (int)shapes[2 * l], (int)shapes[2 * l + 1]);
      m.aw = attn[e];
      m.hh = gg.hh; m.hw = gg.hw; m.lh = gg.lh; m.lw = gg.lw;
      m.e1 = gg.i1 * tok_stride;  // (from the level's first token: grad_value is addressed with the same offset)
      m.ok = (gg.ok1 ? 1 : 0) | (gg.ok2 ? 2 : 0) | (gg.ok3 ? 4 : 0) | (gg.ok4 ? 8 : 0) | (gg.in ? 16 : 0);
      in = gg.in; h_low = gg.h_low; w_low = gg.w_low;
    }
    recs[i] = m;
    if (TILE) {
      s_bin[(l * QB + rr) * P + pp] = in ? ((h_low + 1) << 16) | (w_low + 1) : -1;
      if (in) {  // (+ 0.5: the quotient is at least 1 / 32 away from an integer, far above the rounding of the product)
        const int tt = (int)(((float)(h_low + 1) + 0.5f) * MG.ity[l]) * MG.ntx[l] + (int)(((float)(w_low + 1) + 0.5f) * MG.itx[l]);
        atomicOr(&s_mask[2 * l + ((tt >> 5) & 1)], 1u << (tt & 31));
      }
    }
  }
  __syncthreads();

  const int lane = tid & 63, w = tid >> 6;
  const int r = w * QW + lane / G;
  const int sub = lane % G;
  const int q = q0 + r;
  const bool qok = q < Nq;  // keep whole groups alive for the butterflies

  const long voff = ((long)b * Nk * H + h) * D + sub * 4;
  const float* vb = value + voff;
  float* gvb = grad_value + voff;
  const float4 go = qok ? *reinterpret_cast<const float4*>(
                              grad_out + (((long)b * Nq + q) * H + h) * D + sub * 4)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
  const MsdaSampleB* mine = recs + r * LP;  // (rows past Nq hold zero records: nothing is loaded, nothing counts)

  for (int l = 0; l < L; ++l) {
    const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
    const long lofs = (long)lsi[l] * tok_stride;
    const float* vl = vb + lofs;
    float* gvl = gvb + lofs;
    const int rowstep = Wl * tok_stride;
    MsdaSampleB g[P];
    float aw[P];
    float4 v1[P], v2[P], v3[P], v4[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float4* rp = reinterpret_cast<const float4*>(mine + l * P + p);
      const float4 ra = rp[0], rb = rp[1];
      g[p].aw = ra.x; g[p].hh = ra.y; g[p].hw = ra.z; g[p].lh = ra.w;
      g[p].lw = rb.x; g[p].e1 = __float_as_int(rb.y); g[p].ok = __float_as_int(rb.z);
      aw[p] = ra.x;
    }
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float* t1 = vl + g[p].e1;
      v1[p] = ld4(t1, g[p].ok & 1);
      v2[p] = ld4(t1 + tok_stride, g[p].ok & 2);
      v3[p] = ld4(t1 + rowstep, g[p].ok & 4);
      v4[p] = ld4(t1 + rowstep + tok_stride, g[p].ok & 8);
    }
    float part[3 * P];
    unsigned inmask = 0u;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float hh = g[p].hh, hw = g[p].hw, lh = g[p].lh, lw = g[p].lw;
      const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
      const float4 top = scale4(go, aw[p]);  // grad_out * attention weight
      if (scatter) {
        float* t1 = gvl + g[p].e1;
        atomic_add4(t1, scale4(top, w1), g[p].ok & 1);
        atomic_add4(t1 + tok_stride, scale4(top, w2), g[p].ok & 2);
        atomic_add4(t1 + rowstep, scale4(top, w3), g[p].ok & 4);
        atomic_add4(t1 + rowstep + tok_stride, scale4(top, w4), g[p].ok & 8);
      }
      // d(sample)/d(h_im), d(sample)/d(w_im), and the sample itself, dotted with the grads
      const float d1 = dot4(top, v1[p]), d2 = dot4(top, v2[p]);
      const float d3 = dot4(top, v3[p]), d4 = dot4(top, v4[p]);
      // this lane's share (its 4 channels) of the sample's three sums: [3 p] = d/d(w_im), [3 p + 1] = d/d(h_im), [3 p + 2] = d/d(weight)
      part[3 * p + 0] = -hh * d1 + hh * d2 - lh * d3 + lh * d4;
      part[3 * p + 1] = -hw * d1 - lw * d2 + hw * d3 + lw * d4;
      part[3 * p + 2] = w1 * dot4(go, v1[p]) + w2 * dot4(go, v2[p]) + w3 * dot4(go, v3[p]) + w4 * dot4(go, v4[p]);
      if (g[p].ok & 16) inmask |= 1u << p;
    }
    // the 3 P sums of the level over the G lanes of the group as a REDUCE-SCATTER: at every butterfly step a lane keeps one
    // half of the values and sends the other (12 values on 8 lanes: 6 + 3 + 2 = 11 exchanges against 36 for one all-reduce
    // per value); the pairing of the steps is the butterfly's (offsets G/2 ... 1), so every sum is the same float as before.
    // A lane ends with the values base .. base + NF - 1 (those below rend are real)
    {
      constexpr int N0 = 3 * P;
      int base = 0, rend = N0;
      float cur[N0];
#pragma unroll
      for (int i = 0; i < N0; ++i) cur[i] = part[i];
      rs_steps<N0, N0, G / 2>(cur, sub, base, rend);
      constexpr int NF = rs_final<N0, G / 2>();
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        const int idx = base + i;
        if (idx < rend) {
          const int pp = (idx * 11) >> 5, k = idx - 3 * pp;  // idx / 3 for idx < 32
          const bool in = (inmask >> pp) & 1u;
          const float val = cur[i];
          if (k == 2) s_gattn[r * LP + l * P + pp] = in ? val : 0.f;
          else s_gloc[(r * LP + l * P + pp) * 2 + k] = in ? (k == 0 ? (float)Wl : (float)Hl) * val : 0.f;
        }
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < QB * LP * 2; i += 256) {
    const int rr = i / (LP * 2), c = i - rr * (LP * 2);
    const int qq = q0 + rr;
    if (qq < Nq) grad_loc[(((long)b * Nq + qq) * H + h) * (LP * 2) + c] = s_gloc[i];
  }
  for (int i = tid; i < QB * LP; i += 256) {
    const int rr = i / LP, c = i - rr * LP;
    const int qq = q0 + rr;
    if (qq < Nq) grad_attn[(((long)b * Nq + qq) * H + h) * LP + c] = s_gattn[i];
  }
  if (TILE) {
    const long SP = (long)Nq * P;
    for (int i = tid; i < L * QB * P; i += 256) {
      const int l = i / (QB * P), rem = i - l * (QB * P);
      if (q0 + rem / P < Nq) binw[((long)(b * H + h) * L + l) * ((SP + 3) & ~3L) + (long)q0 * P + rem] = s_bin[i];
    }
    if (tid < L) mask[((long)(b * H + h) * L + tid) * ntiles + tile] = (unsigned long long)s_mask[2 * tid] | ((unsigned long long)s_mask[2 * tid + 1] << 32);
  }
}

// ---------------------------------------------------------------------------------------------
// backward, grad_value by destination ("pull") — no fp32 atomics in the common case
// ---------------------------------------------------------------------------------------------
// Device-scope fp32 atomics execute at the memory side on MI355X (one fabric transaction per
// dword): the 4 taps x D channels of every sample made the scatter formulation ~50x slower than
// the forward gather.  Instead the samples of one (batch, head) are counting-sorted by the token
// their TOP-LEFT tap lands on (an "extended" (H_l+1) x (W_l+1) grid per level, so that top-left
// taps one pixel outside the map have a bin too); every value token then PULLS its gradient from
// the four bins whose 2x2 footprint covers it, D lanes per token, with plain 128-byte gathers of
// grad_out rows (L2-resident: one head's slice per XCD) and a plain coalesced store.  Tokens with
// long lists (the coarse levels) are cut into chunks of MSDA_CH taps that combine with atomics —
// a few hundred lines per launch instead of millions.
//
// Workspace (int32 words, per bh = b*H + h, NE = extended bins <= 2*Nk + 2*L):
//   cnt[BH][NEmax], then per bh: start[NEmax+1] | keyrank[2*Nq*LP] | sorted[Nq*LP] x int4 | itemoff[Nk+1] |
//   items[2*maxItems] | nitems
// taps per work item of the pull kernel (RSCOTR_MSDA_CH overrides, for A/B runs)
static int msda_ch() {
  // 32, not 128: with 128 (fewer atomics, plan kernel 34 -> 18 us, round time unchanged) AND the bf16x3 weight-gradient route
  // on, the 512^2 seg step lost parity whenever earlier processes had left data in device memory (140-440 of 459 gradient
  // tensors outside the tight tier; 10-13 with either switch alone, 8 of 8 runs) — an unwritten word is read somewhere
  // on that combination (both use the shared workspace); not found yet, so the long-standing value stays.
  static const int v = [] { const char* e = getenv("RSCOTR_MSDA_CH"); const int x = e ? atoi(e) : 32; return x >= 8 ? x : 32; }();
  return v;
}
constexpr int MSDA_MAXL = 16;    // levels
// LDS words of the bin histogram: the host only knows the bound NE <= 2 Nk + 2 L + 2 (the level shapes live on the
// device); the kernels know NE = sum (H_l + 1)(W_l + 1) (~1.03 Nk for image pyramids) and all take the same
// decision: NE > lds_words -> the sorted path stands down and the sample kernel scatters with atomics instead.
constexpr int MSDA_LDS_WORDS = (156 * 1024) / 4;
constexpr int MSDA_MAXCHUNK = 64;  // sample chunks (one wavefront each) per (b,h) in the histogram pass

struct MsdaWs {
  long chunkcnt;  // word offset of chunkcnt[BH][C][NEmax] (cnt[BH][NEmax] sits at offset 0)
  long body;      // word offset of the first per-(b,h) block
  long per_bh;    // words per (b,h) block
  long start, keyrank, sorted, itemoff, items, nitems, cpart, mclist;  // word offsets inside a bh block
  int NEmax, maxItems, C, CH;
  int lds_words;  // bins the LDS histogram of the hist / plan kernels can hold (<= NEmax)
};

static MsdaWs msda_ws_layout(int BH, int Nk, int Nq, int L, int P) {
  MsdaWs w;
  const long S = (long)Nq * L * P;
  w.NEmax = 2 * Nk + 2 * L + 2;
  w.lds_words = std::min(w.NEmax, MSDA_LDS_WORDS);
  w.CH = msda_ch();
  w.maxItems = (int)(Nk + (S * 4 + w.CH - 1) / w.CH + 1);
  w.C = (int)std::max<long>(1, std::min<long>(MSDA_MAXCHUNK, S / 1024));
  w.chunkcnt = ((long)BH * w.NEmax + 3) & ~3L;
  w.body = (w.chunkcnt + (long)BH * w.C * w.NEmax + 3) & ~3L;
  long o = 0;
  w.start = o; o += w.NEmax + 1;
  o = (o + 1) & ~1L;
  w.keyrank = o; o += 2 * S;
  o = (o + 3) & ~3L;
  w.sorted = o; o += 4 * S;  // one 16-byte record per sample: {query, weight, lh, lw}
  w.itemoff = o; o += Nk + 1;
  o = (o + 1) & ~1L;
  w.items = o; o += 2L * w.maxItems;
  w.nitems = o; o += 2;
  o = (o + 3) & ~3L;
  w.cpart = o; o += (long)w.maxItems * 64;  // one partial row (<= 64 channels) per work item of a multi-chunk token
  w.mclist = o; o += Nk + 2;                // [0] = number of multi-chunk tokens, then their ids (ascending)
  w.per_bh = (o + 3) & ~3L;
  return w;
}

struct LevelGeom {
  int Hl[MSDA_MAXL], Wl[MSDA_MAXL], lsi[MSDA_MAXL], ext[MSDA_MAXL + 1];
};

__device__ __forceinline__ void load_geom(LevelGeom* g, const int64_t* shapes, const int64_t* lsi, int L) {
  if (threadIdx.x == 0) {
    int e = 0;
    for (int l = 0; l < L; ++l) {
      g->Hl[l] = (int)shapes[2 * l];
      g->Wl[l] = (int)shapes[2 * l + 1];
      g->lsi[l] = (int)lsi[l];
      g->ext[l] = e;
      e += (g->Hl[l] + 1) * (g->Wl[l] + 1);
    }
    g->ext[L] = e;
  }
  __syncthreads();
}

// grid (C, BH), ONE wavefront per workgroup: LDS histogram of one chunk of the samples of (b,h) over the extended bins;
// the LDS atomic's return value is the sample's rank inside (chunk, bin).  One wavefront walks its chunk in program order,
// so the ranks depend on nothing but the data (the LDS serialises the equal-bin lanes of one instruction in a fixed
// order): the sorted record order, hence the summation order of the pull kernel, is the same in every run.  (With four
// wavefronts per chunk — round 1 — their atomics interleaved by timing and grad_value was reproducible to rounding only.)
__global__ __launch_bounds__(64) void msda_hist_kernel(const int64_t* __restrict__ shapes,
                                                        const int64_t* __restrict__ lsi,
                                                        const float* __restrict__ loc, int* __restrict__ ws,
                                                        MsdaWs W, int Nq, int H, int L, int P) {
  extern __shared__ int s_cnt[];
  __shared__ LevelGeom g;
  load_geom(&g, shapes, lsi, L);
  const int NE = g.ext[L];
  if (NE > W.lds_words) return;  // scatter fallback (see MSDA_LDS_WORDS)
  for (int i = threadIdx.x; i < NE; i += 64) s_cnt[i] = 0;
  __syncthreads();
  const int LP = L * P;
  const long S = (long)Nq * LP;
  const int c = blockIdx.x, bh = blockIdx.y;
  const int b = bh / H, h = bh % H;
  int* base = ws + W.body + (long)bh * W.per_bh;
  const long s0 = S * c / W.C, s1 = S * (c + 1) / W.C;
  // four rounds of locations in flight per wavefront (the chain load -> LDS atomic -> store is latency-bound otherwise)
  for (long r0 = s0; r0 < s1; r0 += 4 * 64) {
    float2 xy[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long sid = r0 + u * 64 + threadIdx.x;
      xy[u] = make_float2(-9.f, -9.f);
      if (sid < s1) {
        const int q = (int)(sid / LP), lp = (int)(sid - (long)q * LP);
        xy[u] = *reinterpret_cast<const float2*>(loc + ((((long)b * Nq + q) * H + h) * LP + lp) * 2);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long sid = r0 + u * 64 + threadIdx.x;
      if (sid >= s1) continue;
      const int lp = (int)(sid % LP), l = lp / P;
      const int Hl = g.Hl[l], Wl = g.Wl[l];
      const float h_im = msda_pix(xy[u].y, Hl), w_im = msda_pix(xy[u].x, Wl);
      const bool in = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)Hl) && (w_im < (float)Wl);
      int key = -1, rank = 0;
      if (in) {
        const int ye = (int)floorf(h_im) + 1, xe = (int)floorf(w_im) + 1;
        key = g.ext[l] + ye * (Wl + 1) + xe;
        rank = atomicAdd(&s_cnt[key], 1);
      }
      *reinterpret_cast<int2*>(base + W.keyrank + 2 * sid) = make_int2(key, rank);
    }
  }
  __syncthreads();
  int* out = ws + W.chunkcnt + ((long)bh * W.C + c) * W.NEmax;
  for (int i = threadIdx.x; i < NE; i += 64) out[i] = s_cnt[i];
}

// grid (ceil(NEmax/256), BH): per bin, exclusive prefix over the chunks (in place) and the total
__global__ __launch_bounds__(256) void msda_binsum_kernel(const int64_t* __restrict__ shapes, int* __restrict__ ws,
                                                          MsdaWs W, int L) {
  int NE = 0;
  for (int l = 0; l < L; ++l) NE += ((int)shapes[2 * l] + 1) * ((int)shapes[2 * l + 1] + 1);
  const int i = blockIdx.x * 256 + threadIdx.x, bh = blockIdx.y;
  if (i >= NE || NE > W.lds_words) return;
  int* cc = ws + W.chunkcnt + (long)bh * W.C * W.NEmax + i;
  int run = 0;
  for (int c = 0; c < W.C; ++c) {
    const int t = cc[(long)c * W.NEmax];
    cc[(long)c * W.NEmax] = run;
    run += t;
  }
  ws[(long)bh * W.NEmax + i] = run;
}

// exclusive prefix over the 1024 threads of the block
__device__ __forceinline__ int block_exclusive_scan(int v, int* s_part, int* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) s_part[w] = inc;
  __syncthreads();
  int off = 0, tot = 0;
  for (int i = 0; i < 16; ++i) {
    if (i < w) off += s_part[i];
    tot += s_part[i];
  }
  *total = tot;
  __syncthreads();
  return off + inc - v;
}

// one 1024-thread workgroup per (b,h): bin starts, per-token tap counts -> work items of the pull kernel
template <int D>
__global__ __launch_bounds__(1024) void msda_plan_kernel(const int64_t* __restrict__ shapes,
                                                         const int64_t* __restrict__ lsi, int* __restrict__ ws,
                                                         MsdaWs W, float* __restrict__ grad_value, int Nk, int H,
                                                         int L) {
  extern __shared__ int s_cnt[];
  __shared__ LevelGeom g;
  __shared__ int s_part[16];
  load_geom(&g, shapes, lsi, L);
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  int* base = ws + W.body + (long)bh * W.per_bh;
  const int* cnt = ws + (long)bh * W.NEmax;
  int* start = base + W.start;
  const int NE = g.ext[L];
  if (NE > W.lds_words) return;
  const int tid = threadIdx.x;
  for (int i = tid; i < NE; i += 1024) s_cnt[i] = cnt[i];
  __syncthreads();
  // A: exclusive scan of the bin counts
  {
    const int per = (NE + 1023) / 1024;
    const int i0 = min(NE, tid * per), i1 = min(NE, i0 + per);
    int sum = 0;
    for (int i = i0; i < i1; ++i) sum += s_cnt[i];
    int total;
    int run = block_exclusive_scan(sum, s_part, &total);
    for (int i = i0; i < i1; ++i) {
      start[i] = run;
      run += s_cnt[i];
    }
    if (tid == 0) start[NE] = total;
  }
  // B: taps per token = its four covering bins; chunks of MSDA_CH -> items
  {
    int* itemoff = base + W.itemoff;
    int2* items = reinterpret_cast<int2*>(base + W.items);
    const int per = (Nk + 1023) / 1024;
    const int t0 = min(Nk, tid * per), t1 = min(Nk, t0 + per);
    int sum = 0;
    for (int tok = t0; tok < t1; ++tok) {
      int l = 0;
      while (l + 1 < L && tok >= g.lsi[l + 1]) ++l;
      const int Wl = g.Wl[l], r = tok - g.lsi[l];
      const int y = r / Wl, x = r - y * Wl;
      const int e = g.ext[l] + (y + 1) * (Wl + 1) + (x + 1);
      const int taps = s_cnt[e] + s_cnt[e - 1] + s_cnt[e - (Wl + 1)] + s_cnt[e - (Wl + 1) - 1];
      sum += max(1, (taps + W.CH - 1) / W.CH);
    }
    int total;
    int run = block_exclusive_scan(sum, s_part, &total);
    int nmc = 0;
    for (int tok = t0; tok < t1; ++tok) {
      int l = 0;
      while (l + 1 < L && tok >= g.lsi[l + 1]) ++l;
      const int Wl = g.Wl[l], r = tok - g.lsi[l];
      const int y = r / Wl, x = r - y * Wl;
      const int e = g.ext[l] + (y + 1) * (Wl + 1) + (x + 1);
      const int taps = s_cnt[e] + s_cnt[e - 1] + s_cnt[e - (Wl + 1)] + s_cnt[e - (Wl + 1) - 1];
      const int nch = max(1, (taps + W.CH - 1) / W.CH);
      itemoff[tok] = run;
      for (int j = 0; j < nch; ++j) items[run + j] = make_int2(tok, j);
      run += nch;
      nmc += nch > 1;
    }
    // tokens whose list was cut into several items, in ascending order (the chunk-combine kernel walks this list)
    int mtotal;
    int mrun = block_exclusive_scan(nmc, s_part, &mtotal);
    int* mclist = base + W.mclist;
    for (int tok = t0; tok < t1; ++tok)
      if (itemoff[tok] + 1 < ((tok + 1 < t1) ? itemoff[tok + 1] : run)) mclist[1 + mrun++] = tok;
    if (tid == 0) mclist[0] = mtotal;
    if (tid == 0) {
      itemoff[Nk] = total;
      base[W.nitems] = total;
    }
  }
}

// grid (C, BH): scatter the samples to their sorted slots as 16-byte records {query, attention weight, lh, lw}.
// loc / attn are read here in sample order (coalesced), so that the pull kernel's dependent chain is
// item -> bin -> record -> row instead of item -> bin -> sample id -> loc / attn -> row.
__global__ __launch_bounds__(256) void msda_fill_kernel(const int64_t* __restrict__ shapes,
                                                        const int64_t* __restrict__ lsi,
                                                        const float* __restrict__ loc,
                                                        const float* __restrict__ attn, int* __restrict__ ws, MsdaWs W,
                                                        int Nq, int H, int L, int P) {
  __shared__ LevelGeom g;
  load_geom(&g, shapes, lsi, L);
  if (g.ext[L] > W.lds_words) return;
  const int c = blockIdx.x, bh = blockIdx.y;
  const int b = bh / H, h = bh % H;
  const int LP = L * P;
  const long S = (long)Nq * LP;
  int* base = ws + W.body + (long)bh * W.per_bh;
  int4* rec = reinterpret_cast<int4*>(base + W.sorted);
  const int* cbase = ws + W.chunkcnt + ((long)bh * W.C + c) * W.NEmax;
  const long s0 = S * c / W.C, s1 = S * (c + 1) / W.C;
  for (long sid = s0 + threadIdx.x; sid < s1; sid += 256) {
    const int2 kr = *reinterpret_cast<const int2*>(base + W.keyrank + 2 * sid);
    if (kr.x < 0) continue;
    const int q = (int)(sid / LP), lp = (int)(sid - (long)q * LP), l = lp / P;
    const long so = (((long)b * Nq + q) * H + h) * LP + lp;
    const float2 xy = *reinterpret_cast<const float2*>(loc + so * 2);
    const float a = attn[so];
    const float h_im = msda_pix(xy.y, g.Hl[l]), w_im = msda_pix(xy.x, g.Wl[l]);
    const float lh = h_im - floorf(h_im), lw = w_im - floorf(w_im);
    rec[base[W.start + kr.x] + cbase[kr.x] + kr.y] = make_int4(q, __float_as_int(a), __float_as_int(lh), __float_as_int(lw));
  }
}

// D lanes per work item (token, chunk): gather-accumulate grad_out rows of the chunk's taps.  The kernel is bound
// by its dependent loads (item -> bin counts / starts -> record -> row), not by bandwidth: every lane group works
// on U independent items at once, the loads of each level issued together, which doubles the memory-level
// parallelism of a wavefront at the same occupancy.
template <int D, int U>
__global__ __launch_bounds__(256) void msda_pull_kernel(const int64_t* __restrict__ shapes,
                                                        const int64_t* __restrict__ lsi,
                                                        const float* __restrict__ grad_out,
                                                        float* __restrict__ grad_value, int* __restrict__ ws,
                                                        MsdaWs W, int Nk, int Nq, int H, int L, int blocks_per_bh) {
  constexpr int GPB = 256 / D;  // lane groups per workgroup
  __shared__ LevelGeom g;
  load_geom(&g, shapes, lsi, L);
  if (g.ext[L] > W.lds_words) return;
  const int bh = blockIdx.x / blocks_per_bh, blk = blockIdx.x - bh * blocks_per_bh;
  const int b = bh / H, h = bh % H;
  int* base = ws + W.body + (long)bh * W.per_bh;
  const int* cnt = ws + (long)bh * W.NEmax;
  const int grp = threadIdx.x / D, ln = threadIdx.x % D;
  const int nitems = base[W.nitems];
  if ((long)blk * U * GPB >= nitems) return;  // whole workgroup past the end
  const int4* rec = reinterpret_cast<const int4*>(base + W.sorted);
  const float* go_b = grad_out + ((long)b * Nq * H + h) * D + ln;

  bool live[U];
  int2 it[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int item = (blk * U + u) * GPB + grp;
    live[u] = item < nitems;
    it[u] = live[u] ? reinterpret_cast<const int2*>(base + W.items)[item] : make_int2(0, 0);
  }
  int c[U][4], s[U][4], nch[U], p0[U], p1[U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int tok = it[u].x;
    int l = 0;
    while (l + 1 < L && tok >= g.lsi[l + 1]) ++l;
    const int Wl = g.Wl[l], r = tok - g.lsi[l];
    const int y = r / Wl, x = r - y * Wl;
    const int e0 = g.ext[l] + (y + 1) * (Wl + 1) + (x + 1);
    const int eb[4] = {e0, e0 - 1, e0 - (Wl + 1), e0 - (Wl + 1) - 1};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      c[u][k] = cnt[eb[k]];
      s[u][k] = base[W.start + eb[k]];
    }
    nch[u] = base[W.itemoff + tok + 1] - base[W.itemoff + tok];
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int total = live[u] ? c[u][0] + c[u][1] + c[u][2] + c[u][3] : 0;
    p0[u] = it[u].y * W.CH;
    p1[u] = max(p0[u], min(total, p0[u] + W.CH));
  }
  float acc[U];
#pragma unroll
  for (int u = 0; u < U; ++u) acc[u] = 0.f;
  for (int pb = 0; pb < W.CH; pb += D) {
    // lane ln resolves tap p0 + pb + ln of each item: which bin, which record, its coefficient
    float coef[U];
    int q[U];
    int nb = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int pos = p0[u] + pb + ln;
      coef[u] = 0.f;
      q[u] = 0;
      if (pos < p1[u]) {
        int k = 0, off = pos;
        if (off >= c[u][0]) { off -= c[u][0]; k = 1;
          if (off >= c[u][1]) { off -= c[u][1]; k = 2;
            if (off >= c[u][2]) { off -= c[u][2]; k = 3; } } }
        const int sk = (k == 0) ? s[u][0] : (k == 1) ? s[u][1] : (k == 2) ? s[u][2] : s[u][3];
        const int4 rc = rec[sk + off];
        q[u] = rc.x;
        const float a = __int_as_float(rc.y), lh = __int_as_float(rc.z), lw = __int_as_float(rc.w);
        coef[u] = a * ((k & 2) ? lh : 1.f - lh) * ((k & 1) ? lw : 1.f - lw);
      }
      nb = max(nb, min(D, p1[u] - p0[u] - pb));
    }
#pragma unroll
    for (int o = D; o < kWave; o <<= 1) nb = max(nb, __shfl_xor(nb, o, 64));  // wave-uniform trip count
    if (nb <= 0) break;
    // 8 independent row gathers in flight per item and step; lanes past the end carry coef 0 / row 0
    for (int j0 = 0; j0 < nb; j0 += 8) {
      float cj[U][8], gj[U][8];
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          cj[u][t] = __shfl(coef[u], j0 + t, D);
          const int qj = __shfl(q[u], j0 + t, D);
          gj[u][t] = go_b[(long)qj * H * D];
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[u] += cj[u][t] * gj[u][t];
      }
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) {
    if (!live[u]) continue;
    if (nch[u] > 1) {  // chunk of a long list: partial row, folded in chunk order by msda_chunk_combine_kernel (no atomics)
      const int item = (blk * U + u) * GPB + grp;
      reinterpret_cast<float*>(base + W.cpart)[(long)item * D + ln] = acc[u];
    } else {
      grad_value[(((long)b * Nk + it[u].x) * H + h) * D + ln] = acc[u];
    }
  }
}

// grad_value rows of the tokens whose tap list was cut into several work items: partial rows summed in chunk order.
// grid (blocks, BH): D lanes per token of the (b,h)'s multi-chunk list (msda_plan_kernel).
template <int D>
__global__ __launch_bounds__(256) void msda_chunk_combine_kernel(const int64_t* __restrict__ shapes, float* __restrict__ grad_value,
                                                                 const int* __restrict__ ws, MsdaWs W, int Nk, int H, int L) {
  int NE = 0;
  for (int l = 0; l < L; ++l) NE += ((int)shapes[2 * l] + 1) * ((int)shapes[2 * l + 1] + 1);
  if (NE > W.lds_words) return;  // the sorted path stood down
  constexpr int TPB = 256 / D;
  const int bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int* base = ws + W.body + (long)bh * W.per_bh;
  const int* mclist = base + W.mclist;
  const int n = mclist[0], ln = threadIdx.x % D;
  const float* part = reinterpret_cast<const float*>(base + W.cpart);
  for (int k = blockIdx.x * TPB + threadIdx.x / D; k < n; k += gridDim.x * TPB) {
    const int tok = mclist[1 + k];
    const int i0 = base[W.itemoff + tok], i1 = base[W.itemoff + tok + 1];
    float v = 0.f;
    for (int j = i0; j < i1; ++j) v += part[(long)j * D + ln];
    grad_value[(((long)b * Nk + tok) * H + h) * D + ln] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// backward, grad_value by TILE ACCUMULATION — deterministic, no global sort (round 2, second design; the default)
// ---------------------------------------------------------------------------------------------
// A sample belongs to the BIN of its top-left tap, (h_low + 1, w_low + 1) on the (H_l + 1) x (W_l + 1) "extended" grid of its
// level; its four taps land on the cells (bin, bin + 1 right, bin + 1 down, both).  The bins of a level are cut into tiles of
// at most 16 x 16 bins (17 x 17 cells with the one-cell halo to the right / bottom), and the samples of a level — in sample
// order — into `nch` chunks; one 256-thread workgroup per (b, h, level, tile, chunk):
//   0. the sample kernel (msda_bwd_kernel<.., TILE = true>) leaves one 4-byte bin word {bin | -1} per sample, laid out
//      (b h, level, q, p), and one 64-bit tile mask per block of its queries and level (a pass of its own over the staged
//      locations);
//   1. SCAN: the workgroup reads the bin words of its chunk (coalesced, L2-resident: all workgroups of a (b, h) run on one
//      XCD) and keeps those whose bin lies in its tile — ballot compaction, so the kept list is in sample order.  No sort of
//      the whole sample set: filtering 16 x redundantly costs less than the counting sort did (5 launches, ~110 us);
//   2. every MSDA_T_CAP kept records (and at the end): stable counting sort of the list by bin inside LDS (one wavefront
//      per quarter of the list, LDS atomics return the rank), then lane groups of D/4 lanes walk the runs of equal bin, bins
//      of one parity class (x & 1, y & 1) at a time: ONE 128-byte gather of the sample's grad_out row serves all four taps
//      (the pull formulation gathers it once per tap), four register accumulators per run, added to the tile's LDS cells
//      at the end of the run — bins of one parity class never share a cell, so plain read-add-write;
//   3. the tile's 17 x 17 cell block goes to a partial buffer; msda_tile_combine_kernel sums, per token, the <= 4 tiles that
//      hold its cell x nch chunks in fixed order and stores grad_value (fully overwritten).
// Every float sum runs in an order fixed by the data layout alone: bit-reproducible.
constexpr int MSDA_T_MAXL = 8;     // levels the tile path handles
constexpr int MSDA_T_TS = 16;      // bins per tile edge (at most)
constexpr int MSDA_T_CW = 17;      // cells per tile edge
constexpr int MSDA_T_CAP = 3072;   // kept records per sort + accumulate round (list entries: 8 bytes)
constexpr int MSDA_T_TARGET = 10;  // mean run length (samples per bin and thread) a sample chunk is sized for
#ifndef MSDA_T_OCC
#define MSDA_T_OCC(D) ((D) <= 32 ? 3 : 2)
#endif
constexpr int MSDA_T_SEGB = 2048;  // blocks of the sample kernel per scan segment (their numbers live in LDS)

struct MsdaTiles {
  int L, NW;                                       // levels, workgroups (= partial tiles) per (b, h)
  int Hl[MSDA_T_MAXL], Wl[MSDA_T_MAXL], lsi[MSDA_T_MAXL];
  int tsx[MSDA_T_MAXL], tsy[MSDA_T_MAXL];          // bins per tile along x / y (<= 16)
  int ntx[MSDA_T_MAXL], nty[MSDA_T_MAXL];          // tiles along x / y
  int nch[MSDA_T_MAXL];                            // sample chunks
  int wbase[MSDA_T_MAXL];                          // first workgroup of the level
};

static bool msda_tiles_build(MsdaTiles* T, const int64_t* shapes_host, int L, int Nk, long SP, int D) {
  const int tsy_max = D >= 32 ? 8 : 16;  // MsdaTileGeom<D>::TSY
  if (!shapes_host || L < 1 || L > MSDA_T_MAXL) return false;
  static const int target = [] { const char* e = getenv("RSCOTR_MSDA_TILE_RUN"); const int v = e ? atoi(e) : MSDA_T_TARGET; return v > 0 ? v : MSDA_T_TARGET; }();
  T->L = L;
  int nw = 0, tok = 0;
  for (int l = 0; l < L; ++l) {
    const int Hh = (int)shapes_host[2 * l], Ww = (int)shapes_host[2 * l + 1];
    if (Hh < 1 || Ww < 1 || Hh > 32766 || Ww > 32766) return false;
    T->Hl[l] = Hh; T->Wl[l] = Ww; T->lsi[l] = tok;
    T->ntx[l] = (Ww + 1 + MSDA_T_TS - 1) / MSDA_T_TS;
    T->nty[l] = (Hh + 1 + tsy_max - 1) / tsy_max;
    T->tsx[l] = (Ww + 1 + T->ntx[l] - 1) / T->ntx[l];
    T->tsy[l] = (Hh + 1 + T->nty[l] - 1) / T->nty[l];
    const long tiles = (long)T->ntx[l] * T->nty[l];
    // sample chunks: the walk of the tile kernel is a chain of gathers per thread as long as the longest run of equal bin,
    // so a level is cut into as many chunks as keep the MEAN run (samples of the chunk per bin, per thread sharing a bin)
    // near `target` — the coarse levels receive as many samples as the fine ones on a fraction of the bins
    const long nbt = (long)T->tsx[l] * T->tsy[l], sf = std::max<long>(1, std::min<long>(4, (D >= 32 ? 128 : 256) / nbt));
    long nch = (SP + (long)(Hh + 1) * (Ww + 1) * sf * target - 1) / ((long)(Hh + 1) * (Ww + 1) * sf * target);
    nch = std::max<long>(1, std::min<long>(std::min<long>(nch, 64), SP / 512));
    T->nch[l] = (int)nch;
    T->wbase[l] = nw;
    if (tiles * nch > (1 << 20)) return false;
    nw += (int)(tiles * nch);
    tok += Hh * Ww;
  }
  for (int l = L; l < MSDA_T_MAXL; ++l) {
    T->Hl[l] = T->Wl[l] = 1; T->lsi[l] = tok; T->tsx[l] = T->tsy[l] = 2; T->ntx[l] = T->nty[l] = 1; T->nch[l] = 1; T->wbase[l] = nw;
  }
  T->NW = nw;
  return tok == Nk && nw <= (1 << 20);
}

struct MsdaTileWs {
  long binw, mask, part, total;  // byte offsets
};

static MsdaTileWs msda_tile_ws(const MsdaTiles& T, int BH, int Nq, int P, int D) {
  MsdaTileWs w;
  const long SP = (long)Nq * P;
  long o = 0;
  w.binw = o; o += (long)BH * T.L * ((SP + 3) & ~3L) * 4;  // (rows padded to whole 16-byte loads)
  const int QB = 4 * (kWave / (D / 4));  // queries per workgroup of the sample kernel
  const long nqt = (Nq + QB - 1) / QB;
  w.mask = o; o += (((long)BH * T.L * nqt * 8) + 15) & ~15L;
  w.part = o; o += (long)BH * T.NW * MSDA_T_CW * ((D >= 32 ? 8 : 16) + 1) * D * 4;
  w.total = o;
  return w;
}

// Per-D geometry of the tile kernel: TB threads own one BIN (two for D >= 32: CH = D / TB channels each), 256 threads per
// workgroup, so a tile has 256 / TB bins: 16 x 16 (D = 16) or 16 x 8 (D = 32, 64).
template <int D>
struct MsdaTileGeom {
  static constexpr int TB = D >= 32 ? 2 : 1;
  static constexpr int CH = D / TB;
  static constexpr int V = CH / 4;                 // float4 per thread and row
#ifndef MSDA_T_U16
#define MSDA_T_U16 3  // (4 spilled 17 registers under the 168-register cap of three wavefronts per SIMD once the walk took balanced work items: +21 MiB of scratch writes per launch, encoder call 85 -> 78 us in the lab with 3)
#endif
  static constexpr int U = CH <= 16 ? MSDA_T_U16 : 2;  // samples in flight per thread in the walk
  static constexpr int TSY = 256 / TB / MSDA_T_TS;  // bins per tile along y
  static constexpr int NBIN = MSDA_T_TS * TSY;
  static constexpr int NCELL = MSDA_T_CW * (TSY + 1);
  static constexpr size_t lds_bytes() {
    return (size_t)NCELL * D * 4 + (size_t)MSDA_T_CAP * (4 + 2 + 2) + (4 * NBIN + NBIN + 4 + 8 + 4) * 4 + MSDA_T_SEGB * 2;
  }
};

// exclusive prefix sum of one int per thread over the 256 threads of the workgroup; *total = the sum.  `scratch`: 4 ints
// of LDS nobody else touches between the two barriers inside.
__device__ __forceinline__ int block_exclusive_scan_256(int v, int* scratch, int* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(inc, o, 64);
    if (lane >= o) inc += u;
  }
  if (lane == 63) scratch[w] = inc;
  __syncthreads();
  const int s0 = scratch[0], s1 = scratch[1], s2 = scratch[2], s3 = scratch[3];
  const int before = (w > 0 ? s0 : 0) + (w > 1 ? s1 : 0) + (w > 2 ? s2 : 0);
  *total = s0 + s1 + s2 + s3;
  __syncthreads();
  return before + inc - v;
}

// (lab builds only — scripts/lab/msda_lab.hip defines MSDA_T_PROFILE: shader-clock cycles per phase of every tile workgroup)
#ifdef MSDA_T_PROFILE
__device__ long long g_msda_tprof[1 << 16][8];
#define MSDA_TP_INIT long long tp_last_ = clock64(), tp_acc_[5] = {0, 0, 0, 0, 0}; int tp_n_ = 0;
#define MSDA_TP(i) { const long long t_ = clock64(); tp_acc_[i] += t_ - tp_last_; tp_last_ = t_; }
#define MSDA_TP_N(n) tp_n_ += (n);
#define MSDA_TP_DONE(level) if (threadIdx.x == 0 && blockIdx.x < (1 << 16)) { for (int i_ = 0; i_ < 5; ++i_) g_msda_tprof[blockIdx.x][i_] = tp_acc_[i_]; g_msda_tprof[blockIdx.x][5] = (level); g_msda_tprof[blockIdx.x][6] = tp_n_; g_msda_tprof[blockIdx.x][7] = 1; }
#else
#define MSDA_TP_INIT
#define MSDA_TP(i)
#define MSDA_TP_N(n)
#define MSDA_TP_DONE(level)
#endif

// One 256-thread workgroup per (b, h, level, tile, chunk): see the header of this section.  A thread keeps the four tap
// rows of ITS bin (its CH channels) in registers for the whole life of the workgroup: the walk over the sorted list needs
// no barrier and no LDS accumulator — a thread reads the records of its bin in order, gathers each sample's grad_out row
// (its part) once and feeds the four accumulators; the rows meet in the tile's cells only at the very end.
template <int D, int P>
__global__ __launch_bounds__(256, MSDA_T_OCC(D)) void msda_tile_kernel(const float* __restrict__ go, const float* __restrict__ loc,
                                                        const float* __restrict__ attn,
                                                        const int* __restrict__ binw, const unsigned long long* __restrict__ mask,
                                                        float* __restrict__ part, MsdaTiles T, int Nq, int bshift,
                                                        int nqt, int H, int BH) {
  constexpr int pshift = P == 1 ? 0 : P == 2 ? 1 : P == 4 ? 2 : 3;
  using Gm = MsdaTileGeom<D>;
  constexpr int TB = Gm::TB, CH = Gm::CH, V = Gm::V, U = Gm::U, NBIN = Gm::NBIN, NCELL = Gm::NCELL;
  constexpr int R = 4;  // consecutive records per thread and scan round (one 16-byte load of bin words)
  extern __shared__ __attribute__((aligned(16))) float t_lds[];
  float* acc = t_lds;                                                   // [NCELL][D] (filled at the very end)
  int* lrec = reinterpret_cast<int*>(acc + NCELL * D);                  // [CAP] sample index << 8 | local bin (kept list)
  unsigned short* order = reinterpret_cast<unsigned short*>(lrec + MSDA_T_CAP);  // [CAP] list positions sorted by bin
  unsigned short* rank = order + MSDA_T_CAP;                            // [CAP]
  int* hist = reinterpret_cast<int*>(rank + MSDA_T_CAP);                // [4][NBIN]
  int* binstart = hist + 4 * NBIN;                                      // [NBIN + 1]
  int* wtot = binstart + NBIN + 4;                                      // [2][4] kept records per wavefront (two buffers)
  int* scratch = wtot + 8;                                              // [4]
  unsigned short* blist = reinterpret_cast<unsigned short*>(scratch + 4);  // [MSDA_T_SEGB] blocks of the segment to scan

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // all workgroups of a (b, h) on one XCD (round-robin dispatch: XCD = id % 8): its records, grad_out slices and
  // partial tiles stay in that L2
  const int x8 = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int bh = x8 + 8 * (j / T.NW), e = j % T.NW;
  if (bh >= BH) return;
  const int b = bh / H, h = bh - b * H;
  int l = 0;
  while (l + 1 < T.L && e >= T.wbase[l + 1]) ++l;
  const int r = e - T.wbase[l];
  const int nch = T.nch[l], chunk = r % nch, tile = r / nch;
  const int ty = tile / T.ntx[l], tx = tile - ty * T.ntx[l];
  const int bx0 = tx * T.tsx[l], by0 = ty * T.tsy[l], bx1 = bx0 + T.tsx[l], by1 = by0 + T.tsy[l];
  const int SP = Nq << pshift;  // (< 2^23: the host checks Nq < 2^20; a multiple of 4 or the host keeps nch = 1 ... see c0)
  // chunk bounds on whole blocks of the sample kernel (BS = 1 << bshift records, >= 16: 16-byte loads of bin words)
  const int BS = 1 << bshift;
  const int c0 = (int)((long)SP * chunk / nch) & ~(BS - 1), c1 = chunk + 1 == nch ? SP : (int)((long)SP * (chunk + 1) / nch) & ~(BS - 1);
  const int* bsrc = binw + ((long)bh * T.L + l) * ((SP + 3) & ~3);

  // Work ITEMS of the walk: after the sort every bin's run is cut into parts of at most R0 records, R0 chosen per sort so
  // that the parts number at most NPAIR (the thread pairs of the workgroup): pair j takes item j.  A bin that collects far
  // more samples than its neighbours (the coarse levels; the padded denoising slots of a DINO decoder call, which all carry
  // the SAME reference box and so put hundreds of samples into one bin: a 150-sample run walked by one pair was a chain of
  // 75 dependent gathers, 80 us for a decoder call against 33 with well-spread queries) is walked by as many pairs as the
  // tile has to spare; the parts of a bin meet in its cells in part order (below), so the sums stay fixed by the data alone
  constexpr int NPAIR = 256 / TB;
  constexpr int RMIN = 16;
  const int pair = tid / TB, sub = tid % TB;
  const float* gob = go + ((long)b * Nq * H + h) * D + sub * CH;  // + q * H * D
  const int qstride = H * D;
  // the walk re-derives a sample's tap weights from its sampling location and attention weight (12 algorithmic bytes per
  // sample, L2-resident) with the sample kernel's arithmetic, hence the same floats — round 2 read a 16-byte record per
  // sample that the sample kernel had written: 45 MB of HBM traffic per launch at the encoder shape of configs[1]
  const int Hl = T.Hl[l], Wl = T.Wl[l];
  const int LP = T.L << pshift;
  const float* locb = loc + (((long)b * Nq * H + h) * T.L + l) * (2 << pshift);   // + q * H * LP * 2 + p * 2
  const float* attb = attn + (((long)b * Nq * H + h) * T.L + l) * (1 << pshift);  // + q * H * LP + p
  const long lstride = (long)H * LP * 2, astride = (long)H * LP;
  typedef float v2f __attribute__((ext_vector_type(2)));  // (pairs: v_pk_fma_f32 does two channels per instruction)
  v2f a1[2 * V], a2[2 * V], a3[2 * V], a4[2 * V];  // the bin's four tap rows (this thread's channels)
  for (int i = tid; i < NCELL * D / 4; i += 256) reinterpret_cast<float4*>(acc)[i] = make_float4(0.f, 0.f, 0.f, 0.f);  // (ordered before the first add by the barriers of the scan)

  MSDA_TP_INIT
  // sort the n kept records by bin (stable), then every thread adds the records of its bin to its accumulators
  auto flush = [&](int n) {
    MSDA_TP(1) MSDA_TP_N(n)
    for (int i = tid; i < 4 * NBIN; i += 256) hist[i] = 0;
    __syncthreads();
    const int nw = ((n + 3) / 4 + 63) & ~63;  // records per wavefront (whole rounds of 64)
    const int i0 = w * nw, i1 = min(n, i0 + nw);
    // one wavefront walks its quarter in program order: the rank inside (wavefront, bin) depends on the data only
    for (int i = i0 + lane; i < i1; i += 64) rank[i] = (unsigned short)atomicAdd(&hist[w * NBIN + (lrec[i] & 255)], 1);
    __syncthreads();
    {
      int h0 = 0, h1 = 0, h2 = 0, h3 = 0;
      if (tid < NBIN) { h0 = hist[tid]; h1 = hist[NBIN + tid]; h2 = hist[2 * NBIN + tid]; h3 = hist[3 * NBIN + tid]; }
      int total;
      const int start = block_exclusive_scan_256(h0 + h1 + h2 + h3, scratch, &total);
      if (tid < NBIN) {
        binstart[tid] = start;
        hist[tid] = start; hist[NBIN + tid] = start + h0; hist[2 * NBIN + tid] = start + h0 + h1; hist[3 * NBIN + tid] = start + h0 + h1 + h2;
      }
      if (tid == 0) binstart[NBIN] = total;
    }
    __syncthreads();
    for (int i = i0 + lane; i < i1; i += 64) order[hist[w * NBIN + (lrec[i] & 255)] + rank[i]] = (unsigned short)i;
    __syncthreads();
    MSDA_TP(2)
    // items: parts of at most R0 records per bin, at most NPAIR in all (n / R0 + non-empty bins <= NPAIR)
    const int runlen = tid < NBIN ? binstart[tid + 1] - binstart[tid] : 0;
    const int spare = max(NPAIR - __syncthreads_count(runlen > 0), 1);
    const int R0 = max(RMIN, (n + spare - 1) / spare);
    const int nit = (runlen + R0 - 1) / R0;
    int* itab = hist;  // (the histogram is dead once `order` is written)
    int nitems;
    const int istart = block_exclusive_scan_256(nit, scratch, &nitems);
    for (int k = 0; k < nit; ++k) itab[istart + k] = tid | (k << 8);
    __syncthreads();
    const bool active = pair < nitems;
    const int item = active ? itab[pair] : 0;
    const int bin = item & 255, part_k = item >> 8;
    int s0 = binstart[bin] + part_k * R0, s1 = min(binstart[bin + 1], s0 + R0);
    if (!active) s0 = s1 = 0;
#pragma unroll
    for (int v = 0; v < 2 * V; ++v) a1[v] = a2[v] = a3[v] = a4[v] = v2f{0.f, 0.f};
#pragma unroll 1
    for (int i = s0; i < s1; i += U) {  // U samples in flight per thread, applied in list (= sample) order
      float2 xy[U];
      float aws[U];
      float4 g[U][V];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int sidx = lrec[order[min(i + u, s1 - 1)]] >> 8;
        const int q = sidx >> pshift, pp = sidx & (P - 1);
        xy[u] = *reinterpret_cast<const float2*>(locb + q * lstride + pp * 2);
        aws[u] = attb[q * astride + pp];
        const float4* row = reinterpret_cast<const float4*>(gob + (long)q * qstride);
#pragma unroll
        for (int v = 0; v < V; ++v) g[u][v] = row[v];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (i + u < s1) {
          const float aw = aws[u];  // (bilinear_setup's arithmetic)
          const float h_im = msda_pix(xy[u].y, Hl), w_im = msda_pix(xy[u].x, Wl);
          const float lh = h_im - floorf(h_im), lw = w_im - floorf(w_im);
          const float hw = 1.f - lw, hh = 1.f - lh;
          const float ah = aw * hh, al = aw * lh;  // the tap weights carry the attention weight
          const float w1 = ah * hw, w2 = ah * lw, w3 = al * hw, w4 = al * lw;
          const v2f W1 = {w1, w1}, W2 = {w2, w2}, W3 = {w3, w3}, W4 = {w4, w4};
#pragma unroll
          for (int v = 0; v < V; ++v) {
            const v2f lo = {g[u][v].x, g[u][v].y}, hi = {g[u][v].z, g[u][v].w};
            a1[2 * v] += lo * W1; a1[2 * v + 1] += hi * W1;
            a2[2 * v] += lo * W2; a2[2 * v + 1] += hi * W2;
            a3[2 * v] += lo * W3; a3[2 * v + 1] += hi * W3;
            a4[2 * v] += lo * W4; a4[2 * v + 1] += hi * W4;
          }
        }
      }
    }
    MSDA_TP(3)
    // The parts of one bin are consecutive items, i.e. neighbouring pairs.  (1) Inside a wavefront their rows are summed by a
    // suffix scan over the pairs (Hillis-Steele, shuffles; a fixed tree): the FIRST pair of a bin in each wavefront ends up
    // with the sum of the bin's parts in that wavefront.  (2) Those heads add their rows to the tile's cells — tap k of bin
    // (x, y) belongs to cell (x + (k & 1), y + (k >> 1)) — one tap at a time and, for a bin whose parts straddle wavefronts,
    // one wavefront after the other (at most four rounds): no two threads touch a cell together, and every sum runs in an
    // order the data alone fixes
    {
      constexpr int PPW = 64 / TB;  // pairs per wavefront
      const int pl = lane / TB;
      const int segid = active ? bin : -1 - pl;  // (idle pairs: segments of their own)
      int maxk = active ? part_k : 0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) maxk = max(maxk, __shfl_xor(maxk, o, 64));
      for (int d = 1; d <= maxk && d < PPW; d <<= 1) {  // (wave-uniform: a segment is at most maxk + 1 pairs long)
        const int seg_there = __shfl_down(segid, d * TB, 64);  // (by every lane: a shuffle inside `a && b` would run with the top lanes — the partners — switched off)
        const bool take = (pl + d < PPW) && seg_there == segid;
#pragma unroll
        for (int v = 0; v < 2 * V; ++v) {
          const float x1 = __shfl_down(a1[v].x, d * TB, 64), y1 = __shfl_down(a1[v].y, d * TB, 64);
          const float x2 = __shfl_down(a2[v].x, d * TB, 64), y2 = __shfl_down(a2[v].y, d * TB, 64);
          const float x3 = __shfl_down(a3[v].x, d * TB, 64), y3 = __shfl_down(a3[v].y, d * TB, 64);
          const float x4 = __shfl_down(a4[v].x, d * TB, 64), y4 = __shfl_down(a4[v].y, d * TB, 64);
          if (take) {
            a1[v] += v2f{x1, y1}; a2[v] += v2f{x2, y2}; a3[v] += v2f{x3, y3}; a4[v] += v2f{x4, y4};
          }
        }
      }
      const bool head = active && (part_k == 0 || pl == 0);
      const int round = head ? pair / PPW - (pair - part_k) / PPW : 0;  // wavefronts between the bin's first item and this one
      const int lbx = bin & (MSDA_T_TS - 1), lby = bin / MSDA_T_TS;
      float4* c = reinterpret_cast<float4*>(acc + (lby * MSDA_T_CW + lbx) * D + sub * CH);
      constexpr int CS = D / 4;  // float4 per cell
      auto add = [](float4* p, const float4& v) { float4 o = *p; o.x += v.x; o.y += v.y; o.z += v.z; o.w += v.w; *p = o; };
      for (int k = 0; __syncthreads_or(head && round >= k); ++k) {
        const bool mine = head && round == k;
        if (mine) {
#pragma unroll
          for (int v = 0; v < V; ++v) add(c + v, make_float4(a1[2 * v].x, a1[2 * v].y, a1[2 * v + 1].x, a1[2 * v + 1].y));
        }
        __syncthreads();
        if (mine) {
#pragma unroll
          for (int v = 0; v < V; ++v) add(c + CS + v, make_float4(a2[2 * v].x, a2[2 * v].y, a2[2 * v + 1].x, a2[2 * v + 1].y));
        }
        __syncthreads();
        if (mine) {
#pragma unroll
          for (int v = 0; v < V; ++v) add(c + MSDA_T_CW * CS + v, make_float4(a3[2 * v].x, a3[2 * v].y, a3[2 * v + 1].x, a3[2 * v + 1].y));
        }
        __syncthreads();
        if (mine) {
#pragma unroll
          for (int v = 0; v < V; ++v) add(c + (MSDA_T_CW + 1) * CS + v, make_float4(a4[2 * v].x, a4[2 * v].y, a4[2 * v + 1].x, a4[2 * v + 1].y));
        }
        __syncthreads();
      }
    }
    MSDA_TP(4)
  };

  // scan: only the blocks whose mask names this tile (kept in order in `blist`, a segment of MSDA_T_SEGB blocks at a
  // time); one 16-byte load of R = 4 consecutive bin words per thread and round (one barrier per 1024 records), two rounds
  // requested ahead; the kept list is in sample order (blocks ascending, thread-major inside a round = index order)
  int n = 0, it = 0;
  const int4 none = make_int4(-1, -1, -1, -1);
  const unsigned long long* msrc = mask + ((long)bh * T.L + l) * nqt;
  const int mbit = tile & 63;
  const int blk0 = c0 >> bshift, nblk = (c1 - c0 + BS - 1) >> bshift;
  const int slot = (tid * R) >> bshift, off = (tid * R) & (BS - 1), BPR = (256 * R) >> bshift;  // blocks per round
  for (int seg = 0; seg < nblk; seg += MSDA_T_SEGB) {
    const int segn = min(nblk - seg, MSDA_T_SEGB);
    int cnt = 0;
    for (int j0 = 0; j0 < segn; j0 += 256, ++it) {
      const int jj = j0 + tid;
      const bool keep = jj < segn && ((msrc[blk0 + seg + jj] >> mbit) & 1ull) != 0ull;
      const unsigned long long m = __ballot(keep);
      int* wt = wtot + (it & 1) * 4;
      if (lane == 0) wt[w] = __popcll(m);
      __syncthreads();
      const int t0 = wt[0], t1 = wt[1], t2 = wt[2], t3 = wt[3];
      if (keep) blist[cnt + (w > 0 ? t0 : 0) + (w > 1 ? t1 : 0) + (w > 2 ? t2 : 0) + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)jj;
      cnt += t0 + t1 + t2 + t3;
    }
    __syncthreads();
    MSDA_TP(0)
    auto index = [&](int rb) {  // first record of this thread in the round that starts at list position rb (-1: none)
      const int k = rb + slot;
      return k < cnt ? ((blk0 + seg + (int)blist[k]) << bshift) + off : -1;
    };
    auto fetch = [&](int i) { return i >= 0 ? *reinterpret_cast<const int4*>(bsrc + i) : none; };
    int i0 = index(0), i1 = index(BPR);
    int4 nx0 = fetch(i0), nx1 = fetch(i1);
    for (int rb = 0; rb < cnt; rb += BPR, ++it) {
      const int4 c4 = nx0;
      const int ib = i0;
      nx0 = nx1; i0 = i1;
      i1 = index(rb + 2 * BPR);
      nx1 = fetch(i1);
      const int cur[R] = {c4.x, c4.y, c4.z, c4.w};
      bool sel[R];
      int before = 0, wsum = 0;
#pragma unroll
      for (int k = 0; k < R; ++k) {
        const int bx = cur[k] & 0xffff, by = cur[k] >> 16;  // (-1: by = -1: outside every tile)
        sel[k] = ib >= 0 && cur[k] >= 0 && bx >= bx0 && bx < bx1 && by >= by0 && by < by1 && ib + k < SP;
        const unsigned long long m = __ballot(sel[k]);
        before += __popcll(m & ((1ull << lane) - 1ull));
        wsum += __popcll(m);
      }
      int* wt = wtot + (it & 1) * 4;
      if (lane == 0) wt[w] = wsum;
      __syncthreads();
      const int t0 = wt[0], t1 = wt[1], t2 = wt[2], t3 = wt[3];
      int pos = n + before + (w > 0 ? t0 : 0) + (w > 1 ? t1 : 0) + (w > 2 ? t2 : 0);
#pragma unroll
      for (int k = 0; k < R; ++k) {
        if (sel[k]) {
          const int bx = cur[k] & 0xffff, by = cur[k] >> 16;
          lrec[pos++] = ((ib + k) << 8) | ((by - by0) * MSDA_T_TS + (bx - bx0));
        }
      }
      n += t0 + t1 + t2 + t3;
      if (n > MSDA_T_CAP - 256 * R) {
        __syncthreads();
        flush(n);
        n = 0;
      }
    }
    __syncthreads();  // (blist is rebuilt)
  }
  __syncthreads();
  if (n > 0) flush(n);
  MSDA_TP(1)
  float4* dst = reinterpret_cast<float4*>(part + ((long)bh * T.NW + e) * NCELL * D);
  for (int i = tid; i < NCELL * D / 4; i += 256) dst[i] = reinterpret_cast<const float4*>(acc)[i];
  MSDA_TP(4) MSDA_TP_DONE(l)
}

// grad_value row of every token = the cells that alias it in the (at most four) tiles that hold it, every sample chunk, in
// fixed order.  D/4 lanes per token; workgroups mapped like msda_tile_kernel (one XCD per (b, h)).
template <int D>
__global__ __launch_bounds__(256) void msda_tile_combine_kernel(const float* __restrict__ part, float* __restrict__ grad_value,
                                                                MsdaTiles T, int Nk, int H, int BH, int bpb,
                                                                unsigned* __restrict__ amax_out) {
  constexpr int G = D / 4, TPB = 256 / G;
  constexpr int NCELL = MsdaTileGeom<D>::NCELL;
  const int x8 = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int bh = x8 + 8 * (j / bpb), blk = j % bpb;
  if (bh >= BH) return;
  const int tok = blk * TPB + threadIdx.x / G, c4 = threadIdx.x % G;
  float amx = 0.f;  // max |grad_value| of this lane -> the tensor's range word (the value projection's dX / dW operand)
  if (tok < Nk) {
    const int b = bh / H, h = bh - b * H;
    int l = 0;
    while (l + 1 < T.L && tok >= T.lsi[l + 1]) ++l;
    const int Wl = T.Wl[l], tsx = T.tsx[l], tsy = T.tsy[l], ntx = T.ntx[l], nch = T.nch[l];
    const int rr = tok - T.lsi[l], y = rr / Wl, x = rr - y * Wl;
    const int cx = x + 1, cy = y + 1;  // extended-grid cell of the token
    const int tx = cx / tsx, ty = cy / tsy, lx = cx - tx * tsx, ly = cy - ty * tsy;
    const float* base = part + ((long)bh * T.NW + T.wbase[l]) * NCELL * D + c4 * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    auto tile_cells = [&](int ttx, int tty, int ccy, int ccx) {
      const float* p = base + ((long)(tty * ntx + ttx) * nch * NCELL + ccy * MSDA_T_CW + ccx) * D;
      for (int c = 0; c < nch; ++c) {
        const float4 u = *reinterpret_cast<const float4*>(p + (long)c * NCELL * D);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
    };
    const bool hx = lx == 0 && tx > 0, hy = ly == 0 && ty > 0;  // also the halo column / row of the left / upper tile
    if (hx && hy) tile_cells(tx - 1, ty - 1, tsy, tsx);
    if (hy) tile_cells(tx, ty - 1, tsy, lx);
    if (hx) tile_cells(tx - 1, ty, ly, tsx);
    tile_cells(tx, ty, ly, lx);
    *reinterpret_cast<float4*>(grad_value + (((long)b * Nk + tok) * H + h) * D + c4 * 4) = v;
    amx = amax4(0.f, v);
  }
  amax_commit(amax_out, amx);  // (every lane of the wavefront, also those past the last token)
}

// dynamic LDS of msda_bwd_kernel: per sample a 32-byte record, grad_attn, grad_loc (2), one bin word; + the level masks
static size_t msda_bwd_lds(int QB, int L, int P) { return (size_t)QB * L * P * 12 * sizeof(float) + 64; }

template <int D, int P>
static void launch_bwd_tiled(const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc, const float* attn,
                             const float* go, float* gv, float* gl, float* ga, int B, int Nk, int Nq, int H, int L,
                             const MsdaTiles& T, char* ws, hipStream_t s, unsigned* amax_gv) {
  constexpr int QB = 4 * (kWave / (D / 4));
  const int ntiles = (Nq + QB - 1) / QB;
  const size_t shm = msda_bwd_lds(QB, L, P);  // records + gathered gradients + bin words staged for a coalesced store + masks
  if (shm > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(&msda_bwd_kernel<D, P, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
  const int BH = B * H;
  const MsdaTileWs W = msda_tile_ws(T, BH, Nq, P, D);
  int* binw = reinterpret_cast<int*>(ws + W.binw);
  unsigned long long* mask = reinterpret_cast<unsigned long long*>(ws + W.mask);
  float* part = reinterpret_cast<float*>(ws + W.part);
  MsdaMaskGeom MG;
  for (int l = 0; l < 8; ++l) { MG.itx[l] = 1.f / (float)T.tsx[l]; MG.ity[l] = 1.f / (float)T.tsy[l]; MG.ntx[l] = T.ntx[l]; }
  // grad_loc / grad_attn by sample + one bin word per sample + one tile mask per block of QB queries
  msda_bwd_kernel<D, P, 0, true><<<dim3((unsigned)((long)B * ntiles * H)), dim3(256), shm, s>>>(
      value, shapes, lsi, loc, attn, go, gv, gl, ga, binw, mask, MG, Nk, Nq, H, L, ntiles, 0);
  constexpr size_t lds = MsdaTileGeom<D>::lds_bytes();
  static const bool attr_set = [] {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&msda_tile_kernel<D, P>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return true;
  }();
  (void)attr_set;
  const unsigned bh8 = (unsigned)((BH + 7) / 8) * 8;
  int bshift = 0;
  while ((1 << bshift) < QB * P) ++bshift;
  msda_tile_kernel<D, P><<<dim3(bh8 * (unsigned)T.NW), 256, lds, s>>>(go, loc, attn, binw, mask, part, T, Nq, bshift, ntiles, H, BH);
  const int bpb = (Nk + 256 / (D / 4) - 1) / (256 / (D / 4));
  msda_tile_combine_kernel<D><<<dim3(bh8 * (unsigned)bpb), 256, 0, s>>>(part, gv, T, Nk, H, BH, bpb, amax_gv);
}

// ---------------------------------------------------------------------------------------------
// host dispatch
// ---------------------------------------------------------------------------------------------
static int check_shape(const char* fn, int B, int Nk, int Nq, int H, int D, int L, int P) {
  if (B < 0 || Nk < 0 || Nq < 0 || H <= 0 || L <= 0)
    return fail(RSCOTR_E_SHAPE, "%s: negative/zero dimension (B=%d Nk=%d Nq=%d H=%d L=%d)", fn, B,
                Nk, Nq, H, L);
  if (!(D == 16 || D == 32 || D == 64))
    return fail(RSCOTR_E_SHAPE, "%s: channels per head D=%d not in {16,32,64}", fn, D);
  if (!(P == 1 || P == 2 || P == 4 || P == 8))
    return fail(RSCOTR_E_SHAPE, "%s: num_points P=%d not in {1,2,4,8}", fn, P);
  if ((long)L * P > 64) return fail(RSCOTR_E_SHAPE, "%s: L*P=%d exceeds 64", fn, L * P);
  return RSCOTR_OK;
}

template <int D, int P>
static void launch_fwd(const float* value, const int64_t* shapes, const int64_t* lsi,
                       const float* loc, const float* attn, float* out, int B, int Nk, int Nq,
                       int H, int L, hipStream_t s, const MsdaPrepIn* prep = nullptr) {
  constexpr int QB = 4 * (kWave / (D / 4));
  const int ntiles = (Nq + QB - 1) / QB;
  const size_t shm_rec = (size_t)QB * L * P * sizeof(MsdaSample);
  if (prep) {  // (rscotr_msda_fwd_prep checked rscotr_msda_fused_ok)
    msda_fwd_kernel<D, P, true, true><<<dim3((unsigned)((long)B * ntiles * H)), dim3(256), shm_rec, s>>>(
        value, shapes, lsi, nullptr, nullptr, out, Nk, Nq, H, L, ntiles, *prep);
    return;
  }
  if (RSCOTR_MSDA_FWD_DEDUP && shm_rec <= 48 * 1024 && (long)(Nk + 1) * H * D < (1l << 31)) {
    msda_fwd_kernel<D, P, true><<<dim3((unsigned)((long)B * ntiles * H)), dim3(256), shm_rec, s>>>(
        value, shapes, lsi, loc, attn, out, Nk, Nq, H, L, ntiles);
    return;
  }
  const size_t shm = (size_t)QB * L * P * 3 * sizeof(float);
  msda_fwd_kernel<D, P, false><<<dim3((unsigned)((long)B * ntiles * H)), dim3(256), shm, s>>>(
      value, shapes, lsi, loc, attn, out, Nk, Nq, H, L, ntiles);
}

template <int D, int P>
static void launch_bwd(const float* value, const int64_t* shapes, const int64_t* lsi,
                       const float* loc, const float* attn, const float* go, float* gv, float* gl,
                       float* ga, int B, int Nk, int Nq, int H, int L, hipStream_t s) {
  constexpr int QB = 4 * (kWave / (D / 4));
  const int ntiles = (Nq + QB - 1) / QB;
  const size_t shm = msda_bwd_lds(QB, L, P);
  msda_bwd_kernel<D, P, 1><<<dim3((unsigned)((long)B * ntiles * H)), dim3(256), shm, s>>>(
      value, shapes, lsi, loc, attn, go, gv, gl, ga, nullptr, nullptr, MsdaMaskGeom(), Nk, Nq, H, L, ntiles, 0);
}

// sorted / pull strategy: grad_loc + grad_attn by sample, grad_value by destination token
template <int D, int P>
static void launch_bwd_sorted(const float* value, const int64_t* shapes, const int64_t* lsi,
                              const float* loc, const float* attn, const float* go, float* gv, float* gl,
                              float* ga, int B, int Nk, int Nq, int H, int L, int* ws, hipStream_t s) {
  constexpr int QB = 4 * (kWave / (D / 4));
  const int ntiles = (Nq + QB - 1) / QB;
  const size_t shm = msda_bwd_lds(QB, L, P);
  const int BH = B * H;
  const MsdaWs W = msda_ws_layout(BH, Nk, Nq, L, P);
  const long S = (long)Nq * L * P;
  const size_t hist_lds = (size_t)W.lds_words * sizeof(int);
  const bool may_stand_down = W.NEmax > W.lds_words;  // only the device knows whether the bins fit
  if (hist_lds > 48 * 1024) {  // opt in to large dynamic LDS (up to the 160 KB of a CU)
    hipFuncSetAttribute(reinterpret_cast<const void*>(&msda_hist_kernel),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)hist_lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&msda_plan_kernel<D>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)hist_lds);
  }
  msda_hist_kernel<<<dim3(W.C, BH), 64, hist_lds, s>>>(shapes, lsi, loc, ws, W, Nq, H, L, P);
  if (may_stand_down) {
    // grad_value zeroed for the scatter the sample kernel falls back to; on the sorted path the pull kernel overwrites it
    hipMemsetAsync(gv, 0, (size_t)B * Nk * H * D * sizeof(float), s);
    msda_bwd_kernel<D, P, 2><<<dim3((unsigned)((long)B * ntiles * H)), dim3(256), shm, s>>>(
        value, shapes, lsi, loc, attn, go, gv, gl, ga, nullptr, nullptr, MsdaMaskGeom(), Nk, Nq, H, L, ntiles, W.lds_words);
  } else {
    msda_bwd_kernel<D, P, 0><<<dim3((unsigned)((long)B * ntiles * H)), dim3(256), shm, s>>>(
        value, shapes, lsi, loc, attn, go, gv, gl, ga, nullptr, nullptr, MsdaMaskGeom(), Nk, Nq, H, L, ntiles, 0);
  }
  msda_binsum_kernel<<<dim3((W.NEmax + 255) / 256, BH), 256, 0, s>>>(shapes, ws, W, L);
  msda_plan_kernel<D><<<BH, 1024, hist_lds, s>>>(shapes, lsi, ws, W, gv, Nk, H, L);
  msda_fill_kernel<<<dim3(W.C, BH), 256, 0, s>>>(shapes, lsi, loc, attn, ws, W, Nq, H, L, P);
  constexpr int GPB = 256 / D;
  static const int pull_u = [] { const char* e = getenv("RSCOTR_MSDA_PULL_U"); return e ? atoi(e) : 1; }();
  if (pull_u == 1) {
    const int bpb = (W.maxItems + GPB - 1) / GPB;
    msda_pull_kernel<D, 1><<<dim3((unsigned)((long)BH * bpb)), 256, 0, s>>>(shapes, lsi, go, gv, ws, W, Nk, Nq, H, L, bpb);
  } else {
    const int bpb = (W.maxItems + 2 * GPB - 1) / (2 * GPB);
    msda_pull_kernel<D, 2><<<dim3((unsigned)((long)BH * bpb)), 256, 0, s>>>(shapes, lsi, go, gv, ws, W, Nk, Nq, H, L, bpb);
  }
  msda_chunk_combine_kernel<D><<<dim3(64, BH), 256, 0, s>>>(shapes, gv, ws, W, Nk, H, L);
}

#define RSCOTR_DISPATCH_DP(D, P, CALL)                         \
  switch ((D) * 16 + (P)) {                                    \
    case 16 * 16 + 1: { CALL(16, 1); } break;                  \
    case 16 * 16 + 2: { CALL(16, 2); } break;                  \
    case 16 * 16 + 4: { CALL(16, 4); } break;                  \
    case 16 * 16 + 8: { CALL(16, 8); } break;                  \
    case 32 * 16 + 1: { CALL(32, 1); } break;                  \
    case 32 * 16 + 2: { CALL(32, 2); } break;                  \
    case 32 * 16 + 4: { CALL(32, 4); } break;                  \
    case 32 * 16 + 8: { CALL(32, 8); } break;                  \
    case 64 * 16 + 1: { CALL(64, 1); } break;                  \
    case 64 * 16 + 2: { CALL(64, 2); } break;                  \
    case 64 * 16 + 4: { CALL(64, 4); } break;                  \
    case 64 * 16 + 8: { CALL(64, 8); } break;                  \
  }

}  // namespace rscotr

using namespace rscotr;

extern "C" int rscotr_msda_fwd(const float* value, const int64_t* spatial_shapes,
                               const int64_t* level_start_index, const float* loc,
                               const float* attn, float* out, int B, int Nk, int Nq, int H, int D,
                               int L, int P, void* stream) {
  if (int e = check_shape("rscotr_msda_fwd", B, Nk, Nq, H, D, L, P)) return e;
  if (B == 0 || Nq == 0) return RSCOTR_OK;  // empty query set: nothing to write
  if (!value || !spatial_shapes || !level_start_index || !loc || !attn || !out)
    return fail(RSCOTR_E_ARG, "rscotr_msda_fwd: null pointer");
  if (!aligned16(value) || !aligned16(out))
    return fail(RSCOTR_E_ALIGN, "rscotr_msda_fwd: value/out must be 16-byte aligned");
  if (B == 0 || Nq == 0) return RSCOTR_OK;
  hipStream_t s = (hipStream_t)stream;
  // algorithmic bytes: read value + loc + attn, write out
  ProfScope prof(PROF_MSDA_FWD, 4.0 * B * ((double)Nk * H * D + (double)Nq * H * L * P * 3 + (double)Nq * H * D), s,
                 "rscotr::msda_fwd_kernel<%d, %d>", D, P);
#define CALL(DD, PP) \
  launch_fwd<DD, PP>(value, spatial_shapes, level_start_index, loc, attn, out, B, Nk, Nq, H, L, s)
  RSCOTR_DISPATCH_DP(D, P, CALL)
#undef CALL
  return check_launch("rscotr_msda_fwd");
}

// 1 if the fused entries (rscotr_msda_fwd_prep / rscotr_msda_bwd_prep) take this geometry: 16 samples per (query, head) — the
// prologue's softmax is a 16-lane reduction of the threads that stage them —, a tile's sample records within 48 KB of LDS and
// element offsets within 2^31
extern "C" int rscotr_msda_fused_ok(int Nk, int H, int D, int L, int P) {
  if (!(D == 16 || D == 32 || D == 64) || !(P == 1 || P == 2 || P == 4 || P == 8) || L < 1 || L > MSDA_MAXL || L * P != 16) return 0;
  const int QB = 4 * (kWave / (D / 4));
  return RSCOTR_MSDA_FWD_DEDUP && (size_t)QB * L * P * sizeof(MsdaSample) <= 48 * 1024 && (long)(Nk + 1) * H * D < (1l << 31);
}

extern "C" int rscotr_msda_fwd_prep(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                                    const float* off, const float* logit, int ld_off, int ld_logit, const float* ref,
                                    const float* norm, int refdim, int ref_levels, float* loc, float* attn, float* out, int B,
                                    int Nk, int Nq, int H, int D, int L, int P, void* stream) {
  if (int e = check_shape("rscotr_msda_fwd_prep", B, Nk, Nq, H, D, L, P)) return e;
  if (B == 0 || Nq == 0) return RSCOTR_OK;
  if (!rscotr_msda_fused_ok(Nk, H, D, L, P))
    return fail(RSCOTR_E_SHAPE, "rscotr_msda_fwd_prep: geometry outside rscotr_msda_fused_ok (L * P = %d, D = %d)", L * P, D);
  if ((refdim != 2 && refdim != 4) || (ref_levels != 1 && ref_levels != L) || ld_off < H * L * P * 2 || (ld_off & 1) || ld_logit < H * L * P)
    return fail(RSCOTR_E_SHAPE, "rscotr_msda_fwd_prep: refdim 2 | 4, ref_levels 1 | L, ld_off >= 2 H L P (even), ld_logit >= H L P");
  if (!value || !spatial_shapes || !level_start_index || !off || !logit || !ref || !loc || !attn || !out || (refdim == 2 && !norm))
    return fail(RSCOTR_E_ARG, "rscotr_msda_fwd_prep: null pointer");
  if (!aligned16(value) || !aligned16(out) || ((uintptr_t)off & 7) || ((uintptr_t)loc & 7))
    return fail(RSCOTR_E_ALIGN, "rscotr_msda_fwd_prep: value / out 16-byte, off / loc 8-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  MsdaPrepIn pi{off, logit, ref, norm, loc, attn, ld_off, ld_logit, refdim, ref_levels};
  // algorithmic bytes: rscotr_msda_fwd's, with the raw offsets / logits read and loc / attn written instead of read
  ProfScope prof(PROF_MSDA_FWD, 4.0 * B * ((double)Nk * H * D + (double)Nq * H * L * P * 6 + (double)Nq * H * D), s,
                 "rscotr::msda_fwd_kernel<%d, %d>", D, P);
#define CALL(DD, PP) launch_fwd<DD, PP>(value, spatial_shapes, level_start_index, nullptr, nullptr, out, B, Nk, Nq, H, L, s, &pi)
  RSCOTR_DISPATCH_DP(D, P, CALL)
#undef CALL
  return check_launch("rscotr_msda_fwd_prep");
}

extern "C" int64_t rscotr_msda_bwd_workspace(int B, int Nk, int Nq, int H, int L, int P) {
  if (B <= 0 || Nk <= 0 || Nq <= 0 || H <= 0 || L <= 0 || P <= 0 || L > MSDA_MAXL) return 0;
  const MsdaWs W = msda_ws_layout(B * H, Nk, Nq, L, P);
  return (int64_t)(W.body + (long)B * H * W.per_bh) * 4;
}

extern "C" int64_t rscotr_msda_bwd_tiled_workspace(const int64_t* shapes_host, int B, int Nk, int Nq, int H, int D, int L,
                                                   int P) {
  MsdaTiles T;
  if (B <= 0 || Nq <= 0 || H <= 0 || P <= 0 || Nq >= (1 << 20) || !msda_tiles_build(&T, shapes_host, L, Nk, (long)Nq * P, D)) return 0;
  return msda_tile_ws(T, B * H, Nq, P, D).total;
}

extern "C" int rscotr_msda_bwd(const float* value, const int64_t* spatial_shapes,
                               const int64_t* level_start_index, const float* loc,
                               const float* attn, const float* grad_out, float* grad_value,
                               float* grad_loc, float* grad_attn, int B, int Nk, int Nq, int H,
                               int D, int L, int P, const int64_t* shapes_host, void* workspace,
                               int64_t workspace_bytes, uint32_t* amax_grad_value, void* stream) {
  if (int e = check_shape("rscotr_msda_bwd", B, Nk, Nq, H, D, L, P)) return e;
  if (B == 0 || Nq == 0) return RSCOTR_OK;  // grad_value stays as zeroed by the caller
  if (!value || !spatial_shapes || !level_start_index || !loc || !attn || !grad_out ||
      !grad_value || !grad_loc || !grad_attn)
    return fail(RSCOTR_E_ARG, "rscotr_msda_bwd: null pointer");
  if (!aligned16(value) || !aligned16(grad_out) || !aligned16(grad_value))
    return fail(RSCOTR_E_ALIGN, "rscotr_msda_bwd: value/grad_out/grad_value must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  // algorithmic bytes: read value, read-modify-write grad_value, read loc/attn/grad_out, write grad_loc/grad_attn
  ProfScope prof(PROF_MSDA_BWD, 4.0 * B * (3.0 * Nk * H * D + (double)Nq * H * L * P * 6 + (double)Nq * H * D), s,
                 "rscotr_msda_bwd<%d, %d> (sample + tile + combine kernels)", D, P);
  if (workspace && shapes_host && Nk > 0) {
    MsdaTiles T;
    const int64_t need_t = rscotr_msda_bwd_tiled_workspace(shapes_host, B, Nk, Nq, H, D, L, P);
    if (need_t > 0 && workspace_bytes >= need_t && msda_tiles_build(&T, shapes_host, L, Nk, (long)Nq * P, D)) {
      if (!aligned16(workspace)) return fail(RSCOTR_E_ALIGN, "rscotr_msda_bwd: workspace must be 16-byte aligned");
#define CALL(DD, PP)                                                                                    \
  launch_bwd_tiled<DD, PP>(value, spatial_shapes, level_start_index, loc, attn, grad_out, grad_value, \
                           grad_loc, grad_attn, B, Nk, Nq, H, L, T, (char*)workspace, s, amax_grad_value)
      RSCOTR_DISPATCH_DP(D, P, CALL)
#undef CALL
      return check_launch("rscotr_msda_bwd (tiled)");
    }
  }
  const int64_t need = rscotr_msda_bwd_workspace(B, Nk, Nq, H, L, P);
  if (workspace && need > 0 && workspace_bytes >= need && Nk > 0) {
    if (!aligned16(workspace)) return fail(RSCOTR_E_ALIGN, "rscotr_msda_bwd: workspace must be 16-byte aligned");
    int* ws = (int*)workspace;
#define CALL(DD, PP)                                                                                    \
  launch_bwd_sorted<DD, PP>(value, spatial_shapes, level_start_index, loc, attn, grad_out, grad_value, \
                            grad_loc, grad_attn, B, Nk, Nq, H, L, ws, s)
    RSCOTR_DISPATCH_DP(D, P, CALL)
#undef CALL
    if (int e = check_launch("rscotr_msda_bwd (sorted)")) return e;
    // (only the tiled strategy's combine kernel folds the range of grad_value itself: measure it here)
    return amax_grad_value ? rscotr_amax_f32(grad_value, (int64_t)B * Nk, H * D, H * D, amax_grad_value, stream) : RSCOTR_OK;
  }
#define CALL(DD, PP)                                                                            \
  launch_bwd<DD, PP>(value, spatial_shapes, level_start_index, loc, attn, grad_out, grad_value, \
                     grad_loc, grad_attn, B, Nk, Nq, H, L, s)
  RSCOTR_DISPATCH_DP(D, P, CALL)
#undef CALL
  if (int e = check_launch("rscotr_msda_bwd")) return e;
  return amax_grad_value ? rscotr_amax_f32(grad_value, (int64_t)B * Nk, H * D, H * D, amax_grad_value, stream) : RSCOTR_OK;
}
