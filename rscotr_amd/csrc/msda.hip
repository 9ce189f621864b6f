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

namespace rscotr {

struct Bilinear {
  int i1, i2, i3, i4;      // token offsets inside the level (valid only if ok*)
  bool ok1, ok2, ok3, ok4;  // tap inside the map
  bool in;                  // sample inside (-1, size) on both axes
  float hh, hw, lh, lw;
};

__device__ __forceinline__ Bilinear bilinear_setup(float lx, float ly, int Hl, int Wl) {
  Bilinear t;
  const float h_im = ly * (float)Hl - 0.5f;
  const float w_im = lx * (float)Wl - 0.5f;
  t.in = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)Hl) && (w_im < (float)Wl);
  const float hf = floorf(h_im), wf = floorf(w_im);
  const int h_low = t.in ? (int)hf : 0, w_low = t.in ? (int)wf : 0;
  const int h_high = h_low + 1, w_high = w_low + 1;
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
template <int D, int P>
__global__ __launch_bounds__(256) void msda_fwd_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const float* __restrict__ loc,
    const float* __restrict__ attn, float* __restrict__ out, int Nk, int Nq, int H, int L,
    int ntiles) {
  constexpr int G = D / 4;        // lanes per (query, head)
  constexpr int QW = kWave / G;   // queries per wavefront
  constexpr int QB = 4 * QW;      // queries per workgroup
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int LP = L * P;
  float* s_loc = smem;                // [QB][LP*2]
  float* s_attn = smem + QB * LP * 2;  // [QB][LP]

  const int bid = blockIdx.x;
  const int h = bid % H;
  const int t = bid / H;
  const int tile = t % ntiles;
  const int b = t / ntiles;
  const int q0 = tile * QB;
  const int tid = threadIdx.x;

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
  const int tok_stride = H * D;
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

// SCATTER: 0 = grad_loc / grad_attn only (grad_value comes from the pull kernel), 1 = also scatter grad_value with
// atomics, 2 = scatter iff the level pyramid has more than `bins_cap` extended bins (the sorted path stood down).
template <int D, int P, int SCATTER>
__global__ __launch_bounds__(256) void msda_bwd_kernel(
    const float* __restrict__ value, const int64_t* __restrict__ shapes,
    const int64_t* __restrict__ lsi, const float* __restrict__ loc,
    const float* __restrict__ attn, const float* __restrict__ grad_out,
    float* __restrict__ grad_value, float* __restrict__ grad_loc, float* __restrict__ grad_attn,
    int Nk, int Nq, int H, int L, int ntiles, int bins_cap) {
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
  float* s_loc = smem;                      // [QB][LP*2]  in: locations, out: grad_loc
  float* s_attn = smem + QB * LP * 2;        // [QB][LP]    in: weights
  float* s_gattn = smem + QB * LP * 3;       // [QB][LP]    out: grad_attn
  float* s_gloc = smem + QB * LP * 4;        // [QB][LP*2]  out: grad_loc

  const int bid = blockIdx.x;
  const int h = bid % H;
  const int t = bid / H;
  const int tile = t % ntiles;
  const int b = t / ntiles;
  const int q0 = tile * QB;
  const int tid = threadIdx.x;

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
  const bool qok = q < Nq;  // keep whole groups alive for the butterflies

  const long voff = ((long)b * Nk * H + h) * D + sub * 4;
  const float* vb = value + voff;
  float* gvb = grad_value + voff;
  const int tok_stride = H * D;
  const float4 go = qok ? *reinterpret_cast<const float4*>(
                              grad_out + (((long)b * Nq + q) * H + h) * D + sub * 4)
                        : make_float4(0.f, 0.f, 0.f, 0.f);
  const float* my_loc = s_loc + r * LP * 2;
  const float* my_attn = s_attn + r * LP;

  for (int l = 0; l < L; ++l) {
    const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
    const long lofs = (long)lsi[l] * tok_stride;
    const float* vl = vb + lofs;
    float* gvl = gvb + lofs;
    Bilinear g[P];
    float aw[P];
    float4 v1[P], v2[P], v3[P], v4[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float2 xy = *reinterpret_cast<const float2*>(my_loc + (l * P + p) * 2);
      aw[p] = my_attn[l * P + p];
      g[p] = bilinear_setup(xy.x, xy.y, Hl, Wl);
      if (!qok) g[p].in = g[p].ok1 = g[p].ok2 = g[p].ok3 = g[p].ok4 = false;
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
      const float hh = g[p].hh, hw = g[p].hw, lh = g[p].lh, lw = g[p].lw;
      const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
      const float4 top = scale4(go, aw[p]);  // grad_out * attention weight
      if (scatter) {
        atomic_add4(gvl + (long)g[p].i1 * tok_stride, scale4(top, w1), g[p].ok1);
        atomic_add4(gvl + (long)g[p].i2 * tok_stride, scale4(top, w2), g[p].ok2);
        atomic_add4(gvl + (long)g[p].i3 * tok_stride, scale4(top, w3), g[p].ok3);
        atomic_add4(gvl + (long)g[p].i4 * tok_stride, scale4(top, w4), g[p].ok4);
      }
      // d(sample)/d(h_im), d(sample)/d(w_im), and the sample itself, dotted with the grads
      const float d1 = dot4(top, v1[p]), d2 = dot4(top, v2[p]);
      const float d3 = dot4(top, v3[p]), d4 = dot4(top, v4[p]);
      float gh = -hw * d1 - lw * d2 + hw * d3 + lw * d4;
      float gw = -hh * d1 + hh * d2 - lh * d3 + lh * d4;
      float ga = w1 * dot4(go, v1[p]) + w2 * dot4(go, v2[p]) + w3 * dot4(go, v3[p]) +
                 w4 * dot4(go, v4[p]);
      gh = group_sum<G>(gh);
      gw = group_sum<G>(gw);
      ga = group_sum<G>(ga);
      if (sub == 0) {
        const bool in = g[p].in;
        s_gloc[(r * LP + l * P + p) * 2 + 0] = in ? (float)Wl * gw : 0.f;
        s_gloc[(r * LP + l * P + p) * 2 + 1] = in ? (float)Hl * gh : 0.f;
        s_gattn[r * LP + l * P + p] = in ? ga : 0.f;
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
      const float h_im = xy[u].y * (float)Hl - 0.5f, w_im = xy[u].x * (float)Wl - 0.5f;
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
    const float h_im = xy.y * (float)g.Hl[l] - 0.5f, w_im = xy.x * (float)g.Wl[l] - 0.5f;
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
// backward, grad_value by TILE ACCUMULATION — deterministic (round 2; replaces the sort pipeline above as the default)
// ---------------------------------------------------------------------------------------------
// Every level's value map is cut into square tiles of ts x ts tokens (ts = 16 on the largest level, halved per octave so
// that every level has about the same number of tiles: the levels receive the same number of samples).  A sample belongs
// to the tile of its TOP-LEFT tap; its four taps then lie inside the tile's (ts+1) x (ts+1) block (one-token halo to the
// right / bottom).
//   1. msda_tile_part_kernel: one WAVEFRONT per chunk of 1024 consecutive samples of one (b, h): tile id per sample, STABLE
//      rank inside (chunk, tile) — lanes of equal tile found with ballots, running per-tile counters in LDS updated by one
//      wavefront in program order — then the chunk's samples are written, partitioned by tile, as 16-byte records
//      {query | local top-left cell, attention weight, lw, lh} plus a (chunk, tile) -> (offset, count) table.  No atomics
//      whose order matters: the record order inside a tile is the sample order, always.
//   2. msda_tile_acc_kernel: one workgroup per (b, h, tile) with the block's accumulators in LDS.  Its four wavefronts own
//      the four tap parities (x & 1, y & 1) — the 2x2 footprint of a sample has exactly one tap of each parity — so no two
//      wavefronts ever touch the same accumulator; inside a wavefront the records are processed in list order, 64 / (D/4)
//      samples per step, D/4 lanes x float4 per sample: grad_out row gather (L2-resident: one head per XCD) and a plain
//      LDS read-add-write of the cell's row; the samples of one step that hit the same cell are applied in sample order
//      (rank among equal cells from wave shuffles, one round per rank).  The block is then written to a partial buffer.
//   3. msda_tile_combine_kernel: grad_value[token] = its own tile's cell + the halo cells of the left / upper / upper-left
//      neighbour tiles, fixed order.
// The result is bit-reproducible; traffic: 16-byte records written + read once, partial blocks ~1.2x grad_value.
constexpr int MSDA_T_MAXL = 8;      // levels the tile path handles
constexpr int MSDA_T_CH = 1024;     // samples per partition chunk = 16 rounds of one wavefront
constexpr int MSDA_T_MAXNT = 1024;  // tiles per (b, h)
constexpr int MSDA_T_MAXCH = 2048;  // chunks per (b, h)

struct MsdaTiles {
  int L, NT, prows;  // levels, tiles per (b, h), partial rows (tokens incl. halos) per (b, h)
  int Hl[MSDA_T_MAXL], Wl[MSDA_T_MAXL], lsi[MSDA_T_MAXL];
  int tsh[MSDA_T_MAXL], ntx[MSDA_T_MAXL];      // log2 of the tile edge, tiles per row
  int tbase[MSDA_T_MAXL], pbase[MSDA_T_MAXL];  // first tile / first partial row of the level
};

static bool msda_tiles_build(MsdaTiles* T, const int64_t* shapes_host, int L, int Nk) {
  if (!shapes_host || L < 1 || L > MSDA_T_MAXL) return false;
  static const int ts0 = [] { const char* e = getenv("RSCOTR_MSDA_TS"); const int v = e ? atoi(e) : 16; return (v == 8 || v == 16) ? v : 16; }();
  int maxdim0 = 1;
  for (int l = 0; l < L; ++l) maxdim0 = std::max<int>(maxdim0, (int)std::max(shapes_host[2 * l], shapes_host[2 * l + 1]));
  T->L = L;
  int tiles = 0, prows = 0, tok = 0;
  for (int l = 0; l < L; ++l) {
    const int Hh = (int)shapes_host[2 * l], Ww = (int)shapes_host[2 * l + 1];
    if (Hh < 1 || Ww < 1) return false;
    int ts = ts0;
    for (int d = std::max(Hh, Ww); 2 * d <= maxdim0 + d / 2 && ts > 2; d *= 2) ts >>= 1;  // one halving per octave below the largest level
    int tsh = 0;
    while ((1 << tsh) < ts) ++tsh;
    T->Hl[l] = Hh; T->Wl[l] = Ww; T->lsi[l] = tok; T->tsh[l] = tsh;
    T->ntx[l] = (Ww + ts - 1) >> tsh;
    const int nty = (Hh + ts - 1) >> tsh;
    T->tbase[l] = tiles; T->pbase[l] = prows;
    tiles += T->ntx[l] * nty;
    prows += T->ntx[l] * nty * (ts + 1) * (ts + 1);
    tok += Hh * Ww;
  }
  for (int l = L; l < MSDA_T_MAXL; ++l) {
    T->Hl[l] = T->Wl[l] = 1; T->lsi[l] = tok; T->tsh[l] = 1; T->ntx[l] = 1; T->tbase[l] = tiles; T->pbase[l] = prows;
  }
  T->NT = tiles; T->prows = prows;
  return tok == Nk && tiles <= MSDA_T_MAXNT;
}

struct MsdaTileWs {
  long tbl, rec, part, total;  // byte offsets
  int NCH;
};

static MsdaTileWs msda_tile_ws(const MsdaTiles& T, int BH, long S, int D) {
  MsdaTileWs w;
  w.NCH = (int)((S + MSDA_T_CH - 1) / MSDA_T_CH);
  long o = 0;
  w.tbl = o; o += (long)BH * w.NCH * T.NT * 8;
  o = (o + 15) & ~15L;
  w.rec = o; o += (long)BH * S * 16;
  w.part = o; o += (long)BH * T.prows * D * 4;
  w.total = o;
  return w;
}

template <int P>
__global__ __launch_bounds__(64) void msda_tile_part_kernel(const float* __restrict__ loc, const float* __restrict__ attn,
                                                            int2* __restrict__ tbl, int4* __restrict__ rec, MsdaTiles T,
                                                            int Nq, int H, int S, int NCH) {
  __shared__ int run[MSDA_T_MAXNT], off[MSDA_T_MAXNT];
  __shared__ int gH[MSDA_T_MAXL], gW[MSDA_T_MAXL], gS[MSDA_T_MAXL], gN[MSDA_T_MAXL], gB[MSDA_T_MAXL];
  constexpr int R = MSDA_T_CH / 64;
  const int lane = threadIdx.x, chunk = blockIdx.x, bh = blockIdx.y, b = bh / H, h = bh - b * H;
  const int LP = T.L * P;
  if (lane < MSDA_T_MAXL) {
    gH[lane] = T.Hl[lane]; gW[lane] = T.Wl[lane]; gS[lane] = T.tsh[lane]; gN[lane] = T.ntx[lane]; gB[lane] = T.tbase[lane];
  }
  for (int i = lane; i < T.NT; i += 64) run[i] = 0;
  __syncthreads();
  float2 xy[R];
  int key[R], rank[R];
  const long rowbase = ((long)b * Nq * H + h) * LP;  // + q * H * LP + lp
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int sid = chunk * MSDA_T_CH + j * 64 + lane;
    xy[j] = make_float2(-9.f, -9.f);  // "outside": no bin
    if (sid < S) {
      const int q = sid / LP, lp = sid - q * LP;
      xy[j] = *reinterpret_cast<const float2*>(loc + (rowbase + (long)q * H * LP + lp) * 2);
    }
  }
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int sid = chunk * MSDA_T_CH + j * 64 + lane;
    const int lp = sid % LP, l = lp / P;
    const int Hl = gH[l], Wl = gW[l];
    const float h_im = xy[j].y * (float)Hl - 0.5f, w_im = xy[j].x * (float)Wl - 0.5f;
    const bool in = sid < S && (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)Hl) && (w_im < (float)Wl);
    int k = -1;
    if (in) {
      const int x0 = (int)floorf(w_im), y0 = (int)floorf(h_im);
      const int tsh = gS[l];
      k = gB[l] + (max(y0, 0) >> tsh) * gN[l] + (max(x0, 0) >> tsh);
    }
    // stable rank among the lanes of equal tile (wave-uniform loop over the distinct tiles of this round)
    int rk = 0, cnt = 0;
    bool leader = false;
    unsigned long long rem = __ballot(k >= 0);
    while (rem) {
      const int src = __ffsll((long long)rem) - 1;
      const int kk = __shfl(k, src, 64);
      const unsigned long long m = __ballot(k == kk);
      if (k == kk) {
        rk = __popcll(m & below);
        cnt = __popcll(m);
        leader = lane == src;
      }
      rem &= ~m;
    }
    const int old = k >= 0 ? run[k] : 0;  // every lane reads before the leaders write (one wavefront, program order)
    if (leader) run[k] = old + cnt;
    key[j] = k;
    rank[j] = old + rk;
  }
  __syncthreads();
  {  // exclusive scan of the chunk's per-tile totals -> off[]; table row of the chunk
    const int per = (T.NT + 63) / 64;
    const int i0 = min(T.NT, lane * per), i1 = min(T.NT, i0 + per);
    int sum = 0;
    for (int i = i0; i < i1; ++i) sum += run[i];
    int inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    int r = inc - sum;
    int2* row = tbl + ((long)bh * NCH + chunk) * T.NT;
    for (int i = i0; i < i1; ++i) {
      const int c = run[i];
      off[i] = r;
      row[i] = make_int2(r, c);
      r += c;
    }
  }
  __syncthreads();
  int4* out = rec + (long)bh * S + (long)chunk * MSDA_T_CH;
#pragma unroll
  for (int j = 0; j < R; ++j) {
    if (key[j] < 0) continue;
    const int sid = chunk * MSDA_T_CH + j * 64 + lane;
    const int q = sid / LP, lp = sid - q * LP, l = lp / P;
    const int Hl = gH[l], Wl = gW[l], tsh = gS[l];
    const float h_im = xy[j].y * (float)Hl - 0.5f, w_im = xy[j].x * (float)Wl - 0.5f;
    const float hf = floorf(h_im), wf = floorf(w_im);
    const int x0 = (int)wf, y0 = (int)hf;
    // local top-left cell (+1: -1 .. ts-1 -> 0 .. ts) inside the tile's block
    const int lx = x0 - ((max(x0, 0) >> tsh) << tsh) + 1, ly = y0 - ((max(y0, 0) >> tsh) << tsh) + 1;
    const float a = attn[rowbase + (long)q * H * LP + lp];
    out[off[key[j]] + rank[j]] = make_int4(q | (lx << 20) | (ly << 25), __float_as_int(a), __float_as_int(w_im - wf),
                                           __float_as_int(h_im - hf));
  }
}

struct TileRec {
  int4 r;
  bool ok;
};

template <int D>
__global__ __launch_bounds__(256) void msda_tile_acc_kernel(const float* __restrict__ grad_out, const int2* __restrict__ tbl,
                                                            const int4* __restrict__ rec, float* __restrict__ part, MsdaTiles T,
                                                            int Nq, int H, int S, int NCH, int BH) {
  constexpr int G = D / 4, SPW = kWave / G, U = 2;  // lanes per sample, samples per wavefront step, steps per iteration
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int s_scan[2][4];
  const int bh = blockIdx.x % BH, tile = blockIdx.x / BH;  // blockIdx % 8 == head (H = 8): one head's grad_out rows per XCD's L2
  const int b = bh / H, h = bh - b * H;
  int l = 0;
  while (l + 1 < T.L && tile >= T.tbase[l + 1]) ++l;
  const int tsh = T.tsh[l], ts = 1 << tsh, ntx = T.ntx[l];
  const int tl = tile - T.tbase[l], ty = tl / ntx, tx = tl - ty * ntx;
  const int Wl = T.Wl[l], Hl = T.Hl[l];
  const int bw = ts + 1, ncell = bw * bw;
  float* acc = smem;                                          // [ncell][D]
  int* s_pre = reinterpret_cast<int*>(smem + 17 * 17 * D);    // [K + 1] exclusive prefix of the segment lengths
  int* s_adr = s_pre + NCH + 1;                               // [K] first record of the segment
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < ncell * D; i += 256) acc[i] = 0.f;
  // non-empty (chunk, tile) segments in chunk order: thread t takes a contiguous run of chunks
  int K, n;
  {
    constexpr int PER = MSDA_T_MAXCH / 256;
    const int per = (NCH + 255) / 256;
    int2 e[PER];
    int mine = 0, msum = 0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int c = tid * per + i;
      e[i] = make_int2(0, 0);
      if (i < per && c < NCH) e[i] = tbl[((long)bh * NCH + c) * T.NT + tile];
      if (e[i].y > 0) { ++mine; msum += e[i].y; }
    }
    int ia = mine, ib = msum;  // inclusive scans over the workgroup
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int ta = __shfl_up(ia, o, 64), tb = __shfl_up(ib, o, 64);
      if (lane >= o) { ia += ta; ib += tb; }
    }
    if (lane == 63) { s_scan[0][wave] = ia; s_scan[1][wave] = ib; }
    __syncthreads();
    int oa = 0, ob = 0, ta = 0, tb = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w < wave) { oa += s_scan[0][w]; ob += s_scan[1][w]; }
      ta += s_scan[0][w]; tb += s_scan[1][w];
    }
    K = ta; n = tb;
    int k = oa + ia - mine, p = ob + ib - msum;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      if (e[i].y > 0) {
        s_pre[k] = p;
        s_adr[k] = (tid * per + i) * MSDA_T_CH + e[i].x;
        ++k;
        p += e[i].y;
      }
    }
    if (tid == 0) s_pre[K] = n;
  }
  __syncthreads();

  const int grp = lane / G, sub = lane - grp * G;
  const int px = wave & 1, py = wave >> 1;  // tap parity this wavefront owns
  const int4* rb = rec + (long)bh * S;
  const float* gb = grad_out + ((long)b * Nq * H + h) * D + sub * 4;
  const int niter = (n + SPW * U - 1) / (SPW * U);
  int cur = 0;  // segment cursor of this lane group (its sample index only grows)
  auto load_rec = [&](int it, TileRec (&r)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = (it * U + u) * SPW + grp;
      r[u].ok = j < n;
      r[u].r = make_int4(0, 0, 0, 0);
      if (r[u].ok) {
        while (j >= s_pre[cur + 1]) ++cur;
        r[u].r = rb[s_adr[cur] + (j - s_pre[cur])];
      }
    }
  };
  auto load_go = [&](const TileRec (&r)[U], float4 (&g)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u)
      g[u] = r[u].ok ? *reinterpret_cast<const float4*>(gb + (long)(r[u].r.x & 0xFFFFF) * H * D) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto accumulate = [&](const TileRec (&r)[U], const float4 (&g)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int cell = -1;
      float w = 0.f;
      if (r[u].ok) {
        const int lx = (r[u].r.x >> 20) & 31, ly = (r[u].r.x >> 25) & 31;  // local top-left cell + 1
        const int dx = px ^ ((lx + 1) & 1), dy = py ^ ((ly + 1) & 1);      // the tap of this wavefront's parity (ts is even)
        const int cx = lx - 1 + dx, cy = ly - 1 + dy;
        const int x = (tx << tsh) + cx, y = (ty << tsh) + cy;
        if (x >= 0 && y >= 0 && x < Wl && y < Hl) {
          const float a = __int_as_float(r[u].r.y), lw = __int_as_float(r[u].r.z), lh = __int_as_float(r[u].r.w);
          w = a * (dy ? lh : 1.f - lh) * (dx ? lw : 1.f - lw);
          cell = cy * bw + cx;
        }
      }
      // The SPW samples of this step may hit the same cell: they are applied in sample order, one round per rank among
      // the samples of equal cell (plain LDS read-add-write: no LDS float atomics — those run at about one lane per
      // clock per CU on gfx950 and made this kernel 10x slower).  Nearly always one round.
      int rank = 0;
#pragma unroll
      for (int g2 = 0; g2 < SPW; ++g2) {
        const int c2 = __shfl(cell, g2 * G, 64);
        if (g2 < grp && c2 == cell) ++rank;
      }
      for (int rd = 0; __any(cell >= 0 && rank >= rd); ++rd) {
        if (cell >= 0 && rank == rd) {
          float4* dst = reinterpret_cast<float4*>(acc + cell * D + sub * 4);
          float4 v = *dst;
          v.x += w * g[u].x; v.y += w * g[u].y; v.z += w * g[u].z; v.w += w * g[u].w;
          *dst = v;
        }
      }
    }
  };
  if (niter > 0) {  // three-stage software pipeline: records two iterations ahead, grad_out rows one iteration ahead
    TileRec r0[U], r1[U], r2[U];
    float4 g0[U], g1[U];
    load_rec(0, r0);
    load_rec(1, r1);
    load_go(r0, g0);
    for (int it = 0; it < niter; ++it) {
      load_rec(it + 2, r2);
      load_go(r1, g1);
      accumulate(r0, g0);
#pragma unroll
      for (int u = 0; u < U; ++u) { r0[u] = r1[u]; r1[u] = r2[u]; g0[u] = g1[u]; }
    }
  }
  __syncthreads();
  float* prow = part + ((long)bh * T.prows + T.pbase[l] + (long)tl * ncell) * D;
  for (int i = tid; i < ncell * D; i += 256) prow[i] = acc[i];
}

// grad_value row of every token = own tile's cell + the neighbours' halo cells that alias it, fixed order
template <int D>
__global__ __launch_bounds__(256) void msda_tile_combine_kernel(const float* __restrict__ part, float* __restrict__ grad_value,
                                                                MsdaTiles T, int Nk, int H, int BH) {
  constexpr int G = D / 4;
  const long total = (long)BH * Nk * G;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int c4 = (int)(i % G);
    const long r = i / G;
    const int tok = (int)(r % Nk), bh = (int)(r / Nk);
    const int b = bh / H, h = bh - b * H;
    int l = 0;
    while (l + 1 < T.L && tok >= T.lsi[l + 1]) ++l;
    const int Wl = T.Wl[l], tsh = T.tsh[l], ts = 1 << tsh, bw = ts + 1, ntx = T.ntx[l];
    const int rr = tok - T.lsi[l], y = rr / Wl, x = rr - y * Wl;
    const int tx = x >> tsh, ty = y >> tsh, lx = x & (ts - 1), ly = y & (ts - 1);
    const float* base = part + ((long)bh * T.prows + T.pbase[l]) * D + c4 * 4;
    auto cellp = [&](int ttx, int tty, int cy, int cx) {
      return *reinterpret_cast<const float4*>(base + ((long)(tty * ntx + ttx) * bw * bw + cy * bw + cx) * D);
    };
    float4 v = cellp(tx, ty, ly, lx);
    if (lx == 0 && tx > 0) { const float4 u = cellp(tx - 1, ty, ly, ts); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
    if (ly == 0 && ty > 0) { const float4 u = cellp(tx, ty - 1, ts, lx); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
    if (lx == 0 && tx > 0 && ly == 0 && ty > 0) { const float4 u = cellp(tx - 1, ty - 1, ts, ts); v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w; }
    *reinterpret_cast<float4*>(grad_value + (((long)b * Nk + tok) * H + h) * D + c4 * 4) = v;
  }
}

template <int D, int P>
static void launch_bwd_tiled(const float* value, const int64_t* shapes, const int64_t* lsi, const float* loc, const float* attn,
                             const float* go, float* gv, float* gl, float* ga, int B, int Nk, int Nq, int H, int L,
                             const MsdaTiles& T, char* ws, hipStream_t s) {
  constexpr int QB = 4 * (kWave / (D / 4));
  const int ntiles = (Nq + QB - 1) / QB;
  const size_t shm = (size_t)QB * L * P * 6 * sizeof(float);
  const int BH = B * H;
  const long S = (long)Nq * L * P;
  const MsdaTileWs W = msda_tile_ws(T, BH, S, D);
  int2* tbl = reinterpret_cast<int2*>(ws + W.tbl);
  int4* rec = reinterpret_cast<int4*>(ws + W.rec);
  float* part = reinterpret_cast<float*>(ws + W.part);
  msda_tile_part_kernel<P><<<dim3(W.NCH, BH), 64, 0, s>>>(loc, attn, tbl, rec, T, Nq, H, (int)S, W.NCH);
  // grad_loc / grad_attn by sample (independent of the partition: the two kernels overlap at the launch boundary)
  msda_bwd_kernel<D, P, 0><<<dim3((unsigned)((long)B * ntiles * H)), dim3(256), shm, s>>>(
      value, shapes, lsi, loc, attn, go, gv, gl, ga, Nk, Nq, H, L, ntiles, 0);
  const size_t acc_lds = ((size_t)17 * 17 * D + 2 * (W.NCH + 1)) * sizeof(float);
  if (acc_lds > 48 * 1024)
    hipFuncSetAttribute(reinterpret_cast<const void*>(&msda_tile_acc_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)acc_lds);
  msda_tile_acc_kernel<D><<<dim3((unsigned)((long)BH * T.NT)), 256, acc_lds, s>>>(go, tbl, rec, part, T, Nq, H, (int)S, W.NCH, BH);
  const long items = (long)BH * Nk * (D / 4);
  msda_tile_combine_kernel<D><<<(unsigned)std::min<long>((items + 255) / 256, 4096), 256, 0, s>>>(part, gv, T, Nk, H, BH);
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
                       int H, int L, hipStream_t s) {
  constexpr int QB = 4 * (kWave / (D / 4));
  const int ntiles = (Nq + QB - 1) / QB;
  const size_t shm = (size_t)QB * L * P * 3 * sizeof(float);
  msda_fwd_kernel<D, P><<<dim3((unsigned)((long)B * ntiles * H)), dim3(256), shm, s>>>(
      value, shapes, lsi, loc, attn, out, Nk, Nq, H, L, ntiles);
}

template <int D, int P>
static void launch_bwd(const float* value, const int64_t* shapes, const int64_t* lsi,
                       const float* loc, const float* attn, const float* go, float* gv, float* gl,
                       float* ga, int B, int Nk, int Nq, int H, int L, hipStream_t s) {
  constexpr int QB = 4 * (kWave / (D / 4));
  const int ntiles = (Nq + QB - 1) / QB;
  const size_t shm = (size_t)QB * L * P * 6 * sizeof(float);
  msda_bwd_kernel<D, P, 1><<<dim3((unsigned)((long)B * ntiles * H)), dim3(256), shm, s>>>(
      value, shapes, lsi, loc, attn, go, gv, gl, ga, Nk, Nq, H, L, ntiles, 0);
}

// sorted / pull strategy: grad_loc + grad_attn by sample, grad_value by destination token
template <int D, int P>
static void launch_bwd_sorted(const float* value, const int64_t* shapes, const int64_t* lsi,
                              const float* loc, const float* attn, const float* go, float* gv, float* gl,
                              float* ga, int B, int Nk, int Nq, int H, int L, int* ws, hipStream_t s) {
  constexpr int QB = 4 * (kWave / (D / 4));
  const int ntiles = (Nq + QB - 1) / QB;
  const size_t shm = (size_t)QB * L * P * 6 * sizeof(float);
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
        value, shapes, lsi, loc, attn, go, gv, gl, ga, Nk, Nq, H, L, ntiles, W.lds_words);
  } else {
    msda_bwd_kernel<D, P, 0><<<dim3((unsigned)((long)B * ntiles * H)), dim3(256), shm, s>>>(
        value, shapes, lsi, loc, attn, go, gv, gl, ga, Nk, Nq, H, L, ntiles, 0);
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

extern "C" int64_t rscotr_msda_bwd_workspace(int B, int Nk, int Nq, int H, int L, int P) {
  if (B <= 0 || Nk <= 0 || Nq <= 0 || H <= 0 || L <= 0 || P <= 0 || L > MSDA_MAXL) return 0;
  const MsdaWs W = msda_ws_layout(B * H, Nk, Nq, L, P);
  return (int64_t)(W.body + (long)B * H * W.per_bh) * 4;
}

extern "C" int64_t rscotr_msda_bwd_tiled_workspace(const int64_t* shapes_host, int B, int Nk, int Nq, int H, int D, int L,
                                                   int P) {
  MsdaTiles T;
  if (B <= 0 || Nq <= 0 || H <= 0 || P <= 0 || !msda_tiles_build(&T, shapes_host, L, Nk)) return 0;
  const long S = (long)Nq * L * P;
  if ((S + MSDA_T_CH - 1) / MSDA_T_CH > MSDA_T_MAXCH || Nq >= (1 << 20)) return 0;
  return msda_tile_ws(T, B * H, S, D).total;
}

extern "C" int rscotr_msda_bwd(const float* value, const int64_t* spatial_shapes,
                               const int64_t* level_start_index, const float* loc,
                               const float* attn, const float* grad_out, float* grad_value,
                               float* grad_loc, float* grad_attn, int B, int Nk, int Nq, int H,
                               int D, int L, int P, const int64_t* shapes_host, void* workspace,
                               int64_t workspace_bytes, void* stream) {
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
                 "rscotr_msda_bwd<%d, %d> (hist + sample + plan + fill + pull kernels)", D, P);
  if (workspace && shapes_host && Nk > 0) {
    MsdaTiles T;
    const int64_t need_t = rscotr_msda_bwd_tiled_workspace(shapes_host, B, Nk, Nq, H, D, L, P);
    if (need_t > 0 && workspace_bytes >= need_t && msda_tiles_build(&T, shapes_host, L, Nk)) {
      if (!aligned16(workspace)) return fail(RSCOTR_E_ALIGN, "rscotr_msda_bwd: workspace must be 16-byte aligned");
#define CALL(DD, PP)                                                                                    \
  launch_bwd_tiled<DD, PP>(value, spatial_shapes, level_start_index, loc, attn, grad_out, grad_value, \
                           grad_loc, grad_attn, B, Nk, Nq, H, L, T, (char*)workspace, s)
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
    return check_launch("rscotr_msda_bwd (sorted)");
  }
#define CALL(DD, PP)                                                                            \
  launch_bwd<DD, PP>(value, spatial_shapes, level_start_index, loc, attn, grad_out, grad_value, \
                     grad_loc, grad_attn, B, Nk, Nq, H, L, s)
  RSCOTR_DISPATCH_DP(D, P, CALL)
#undef CALL
  return check_launch("rscotr_msda_bwd");
}
