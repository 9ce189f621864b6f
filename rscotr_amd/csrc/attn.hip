// Dense masked attention pieces for the det / seg decoders: torch.nn.MultiheadAttention as wrapped by mmcv
// MultiheadAttention (configs/multi/MTL_slvlcls_...potsdam.py:81-85,144-151; reached from
// models/multi/bbox_head/transformer.py:103-108 and models/multi/seg_head/mask2former_head.py:183-192).
// The two products per head run on rscotr_gemm_f32_batched (per-head slices addressed in place); this file
// holds what sits between them: P = softmax(scale * S + mask) over the key axis, and its backward
// dS = scale * P * (dP - sum_k P_k dP_k), both in place.  A boolean mask (True = blocked, as in torch) is
// indexed by `mask_mode`: 0 none, 1 shared (Lq, Lk), 2 per image (B, Lq, Lk), 3 per image and head
// (B*heads, Lq, Lk).
//
// Mapping: one wavefront per row (rows are 64 ... 4096 keys: at most 16 KB, they stay in L1/L2 across the
// passes), four rows per workgroup; max / sum by wave butterflies.
#include "common.h"
#include <stdlib.h>

namespace rscotr {

__device__ __forceinline__ const unsigned char* mask_row(const unsigned char* mask, int mode, long row, int Lq,
                                                         int Lk, int heads) {
  if (mode == 0) return nullptr;
  const long i = row % Lq, bh = row / Lq;
  const long blk = mode == 1 ? 0 : (mode == 2 ? bh / heads : bh);
  return mask + (blk * Lq + i) * Lk;
}

__global__ __launch_bounds__(256) void softmax_mask_fwd_kernel(float* __restrict__ S, const unsigned char* __restrict__ mask,
                                                               int mode, long rows, int Lq, int Lk, int heads, float scale) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float* s = S + row * Lk;
  const unsigned char* m = mask_row(mask, mode, row, Lq, Lk, heads);
  float mx = -3.0e38f;
  for (int j = lane; j < Lk; j += 64) {
    const bool blocked = m && m[j];
    if (!blocked) mx = fmaxf(mx, s[j] * scale);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < Lk; j += 64) {
    const bool blocked = m && m[j];
    const float e = blocked ? 0.f : expf(s[j] * scale - mx);
    s[j] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  const float inv = sum > 0.f ? 1.f / sum : 0.f;  // a fully blocked row (never produced on this path) -> zeros
  for (int j = lane; j < Lk; j += 64) s[j] *= inv;
}

// Rows of Lk = 4 * 64 * NV keys at most (Lk a multiple of 4) stay in registers as NV float4 per lane: one read and one write
// per element with 16-byte accesses instead of three scalar passes (the det decoder's 800-key rows: 22 -> ~10 us per call).
template <int NV>
__global__ __launch_bounds__(256) void softmax_mask_fwd_vec_kernel(float* __restrict__ S, const unsigned char* __restrict__ mask,
                                                                   int mode, long rows, int Lq, int Lk, int heads, float scale) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63, Lk4 = Lk >> 2;
  float4* s4 = reinterpret_cast<float4*>(S + row * Lk);
  const unsigned char* m = mask_row(mask, mode, row, Lq, Lk, heads);
  float4 v[NV];
  unsigned long long blk = 0ull;  // bit 4 i + k: element k of vector i is blocked (or past the row)
  float mx = -3.0e38f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    if (c < Lk4) {
      v[i] = s4[c];
      unsigned b4 = 0u;
      if (m) {
        const uchar4 mm = reinterpret_cast<const uchar4*>(m)[c];
        b4 = (mm.x ? 1u : 0u) | (mm.y ? 2u : 0u) | (mm.z ? 4u : 0u) | (mm.w ? 8u : 0u);
      }
      v[i].x *= scale; v[i].y *= scale; v[i].z *= scale; v[i].w *= scale;
      if (!(b4 & 1u)) mx = fmaxf(mx, v[i].x);
      if (!(b4 & 2u)) mx = fmaxf(mx, v[i].y);
      if (!(b4 & 4u)) mx = fmaxf(mx, v[i].z);
      if (!(b4 & 8u)) mx = fmaxf(mx, v[i].w);
      blk |= (unsigned long long)b4 << (4 * i);
    } else {
      blk |= 15ull << (4 * i);
    }
  }
  mx = wave_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const unsigned b4 = (unsigned)(blk >> (4 * i)) & 15u;
    v[i].x = (b4 & 1u) ? 0.f : expf(v[i].x - mx);
    v[i].y = (b4 & 2u) ? 0.f : expf(v[i].y - mx);
    v[i].z = (b4 & 4u) ? 0.f : expf(v[i].z - mx);
    v[i].w = (b4 & 8u) ? 0.f : expf(v[i].w - mx);
    sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  sum = wave_sum(sum);
  const float inv = sum > 0.f ? 1.f / sum : 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    if (c < Lk4) s4[c] = make_float4(v[i].x * inv, v[i].y * inv, v[i].z * inv, v[i].w * inv);
  }
}

template <int NV>
__global__ __launch_bounds__(256) void softmax_bwd_vec_kernel(const float* __restrict__ P, float* __restrict__ dP, long rows,
                                                              int Lk, float scale) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63, Lk4 = Lk >> 2;
  const float4* p4 = reinterpret_cast<const float4*>(P + row * Lk);
  float4* d4 = reinterpret_cast<float4*>(dP + row * Lk);
  float4 p[NV], d[NV];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    if (c < Lk4) {
      p[i] = p4[c];
      d[i] = d4[c];
      dot += (p[i].x * d[i].x + p[i].y * d[i].y) + (p[i].z * d[i].z + p[i].w * d[i].w);
    }
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane + i * 64;
    if (c < Lk4)
      d4[c] = make_float4(scale * p[i].x * (d[i].x - dot), scale * p[i].y * (d[i].y - dot), scale * p[i].z * (d[i].z - dot),
                          scale * p[i].w * (d[i].w - dot));
  }
}

// dP <- scale * P * (dP - sum_k P_k dP_k)
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ P, float* __restrict__ dP, long rows,
                                                          int Lk, float scale) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* p = P + row * Lk;
  float* d = dP + row * Lk;
  float dot = 0.f;
  for (int j = lane; j < Lk; j += 64) dot += p[j] * d[j];
  dot = wave_sum(dot);
  for (int j = lane; j < Lk; j += 64) d[j] = scale * p[j] * (d[j] - dot);
}

// ---- sampling locations + attention weights of mmcv MultiScaleDeformableAttention.forward in one pass -----------
// (models reach it at seg_head/pixel_decoder.py:134-146, bbox_head/transformer.py:211-221,258-269)
//   attn = softmax over the L*P logits of a (query, head);
//   loc  = ref_xy + off / (W_l, H_l)                      (2-d reference points)
//   loc  = ref_xy + off / P * ref_wh * 0.5                (4-d reference points)
// 16 consecutive lanes own one (image, query, head) when L*P == 16 (the configs' 4 levels x 4 points); general
// L*P <= 64 uses one wavefront slice of LP lanes rounded up to a power of two.
template <int G>
__global__ __launch_bounds__(256) void msda_prep_fwd_kernel(const float* __restrict__ off, const float* __restrict__ logit,
                                                            const float* __restrict__ ref, const float* __restrict__ norm,
                                                            float* __restrict__ loc, float* __restrict__ attn, long groups,
                                                            int Nq, int H, int L, int P, int refdim, int ld_off,
                                                            int ld_logit, int ref_levels) {
  const long gid = ((long)blockIdx.x * 256 + threadIdx.x) / G;
  const int s = threadIdx.x % G, LP = L * P;
  const bool in = gid < groups && s < LP;
  const long e = gid * LP + s;
  float lg = -3.0e38f;
  if (in) {
    const long bq = gid / H;                // b * Nq + q
    const int h = (int)(gid - bq * H);
    lg = logit[bq * ld_logit + h * LP + s];
    const int l = s / P;
    const float* r = ref + (bq * ref_levels + (ref_levels > 1 ? l : 0)) * refdim;
    const float2 o = *reinterpret_cast<const float2*>(off + bq * ld_off + (h * LP + s) * 2);
    reinterpret_cast<float2*>(loc)[e] = msda_location(r, o, norm, l, P, refdim);
  }
  const float m = group_max<G>(lg);
  const float ex = in ? expf(lg - m) : 0.f;
  const float sum = group_sum<G>(ex);
  if (in) attn[e] = ex / sum;
}

// grad_off = grad_loc * d(loc)/d(off); grad_logit = attn * (grad_attn - sum attn * grad_attn)
template <int G>
__global__ __launch_bounds__(256) void msda_prep_bwd_kernel(const float* __restrict__ gloc, const float* __restrict__ gattn,
                                                            const float* __restrict__ attn, const float* __restrict__ ref,
                                                            const float* __restrict__ norm, float* __restrict__ goff,
                                                            float* __restrict__ glogit, long groups, int Nq, int H, int L,
                                                            int P, int refdim, int ld_off, int ld_logit, int ref_levels,
                                                            unsigned* __restrict__ amax_out) {
  const long gid = ((long)blockIdx.x * 256 + threadIdx.x) / G;
  const int s = threadIdx.x % G, LP = L * P;
  const bool in = gid < groups && s < LP;
  float amx = 0.f;  // max |grad_off|, |grad_logit| -> the range word of the gradient (common.h: amax_commit)
  const long e = gid * LP + s;
  const long bq = gid / H;
  const int h = (int)(gid - bq * H);
  float p = 0.f, ga = 0.f;
  if (in) {
    p = attn[e];
    ga = gattn[e];
    const int l = s / P;
    const float2 g = reinterpret_cast<const float2*>(gloc)[e];
    float2 out;
    if (refdim == 2) {
      out.x = g.x / norm[2 * l];
      out.y = g.y / norm[2 * l + 1];
    } else {
      const float* r = ref + (bq * ref_levels + (ref_levels > 1 ? l : 0)) * refdim;
      out.x = g.x * (r[2] * 0.5f) / (float)P;
      out.y = g.y * (r[3] * 0.5f) / (float)P;
    }
    *reinterpret_cast<float2*>(goff + bq * ld_off + (h * LP + s) * 2) = out;
    amx = fmaxf(fabsf(out.x), fabsf(out.y));
  }
  const float dot = group_sum<G>(p * ga);
  if (in) {
    const float gl = p * (ga - dot);
    glogit[bq * ld_logit + h * LP + s] = gl;
    amx = fmaxf(amx, fabsf(gl));
  }
  amax_commit(amax_out, amx);
}

}  // namespace rscotr

using namespace rscotr;

static int msda_prep_check(const char* fn, int B, int Nq, int H, int L, int P, int refdim, int ld_off, int ld_logit,
                           int ref_levels) {
  if (B < 0 || Nq < 0 || H <= 0 || L <= 0 || P <= 0) return fail(RSCOTR_E_SHAPE, "%s: bad shape", fn);
  if (ld_off < H * L * P * 2 || (ld_off & 1) || ld_logit < H * L * P)
    return fail(RSCOTR_E_SHAPE, "%s: row strides must cover a row (offsets: even)", fn);
  if (ref_levels != L && ref_levels != 1) return fail(RSCOTR_E_SHAPE, "%s: ref_levels must be L or 1", fn);
  if (L * P > 64) return fail(RSCOTR_E_SHAPE, "%s: L*P = %d > 64", fn, L * P);
  if (refdim != 2 && refdim != 4) return fail(RSCOTR_E_SHAPE, "%s: reference points must be 2- or 4-d", fn);
  return RSCOTR_OK;
}

#define MSDA_PREP_DISPATCH(LP, CALL) \
  do {                               \
    if ((LP) <= 8) { CALL(8); }      \
    else if ((LP) <= 16) { CALL(16); } \
    else if ((LP) <= 32) { CALL(32); } \
    else { CALL(64); }               \
  } while (0)

extern "C" int rscotr_msda_prep_fwd(const float* off, const float* logit, const float* ref, const float* norm, float* loc,
                                    float* attn, int B, int Nq, int H, int L, int P, int refdim, int ld_off, int ld_logit,
                                    int ref_levels, void* stream) {
  if (int e = msda_prep_check("rscotr_msda_prep_fwd", B, Nq, H, L, P, refdim, ld_off, ld_logit, ref_levels)) return e;
  const long groups = (long)B * Nq * H;
  if (groups == 0) return RSCOTR_OK;
  if (!off || !logit || !ref || !loc || !attn || (refdim == 2 && !norm))
    return fail(RSCOTR_E_ARG, "rscotr_msda_prep_fwd: null pointer");
#define CALL(G) \
  msda_prep_fwd_kernel<G><<<(unsigned)((groups * G + 255) / 256), 256, 0, (hipStream_t)stream>>>(off, logit, ref, norm, loc, attn, groups, Nq, H, L, P, refdim, ld_off, ld_logit, ref_levels)
  MSDA_PREP_DISPATCH(L * P, CALL);
#undef CALL
  return check_launch("rscotr_msda_prep_fwd");
}

extern "C" int rscotr_msda_prep_bwd(const float* grad_loc, const float* grad_attn, const float* attn, const float* ref,
                                    const float* norm, float* grad_off, float* grad_logit, int B, int Nq, int H, int L,
                                    int P, int refdim, int ld_off, int ld_logit, int ref_levels, uint32_t* amax_out, void* stream) {
  if (int e = msda_prep_check("rscotr_msda_prep_bwd", B, Nq, H, L, P, refdim, ld_off, ld_logit, ref_levels)) return e;
  const long groups = (long)B * Nq * H;
  if (groups == 0) return RSCOTR_OK;
  if (!grad_loc || !grad_attn || !attn || !ref || !grad_off || !grad_logit || (refdim == 2 && !norm))
    return fail(RSCOTR_E_ARG, "rscotr_msda_prep_bwd: null pointer");
#define CALL(G) \
  msda_prep_bwd_kernel<G><<<(unsigned)((groups * G + 255) / 256), 256, 0, (hipStream_t)stream>>>(grad_loc, grad_attn, attn, ref, norm, grad_off, grad_logit, groups, Nq, H, L, P, refdim, ld_off, ld_logit, ref_levels, amax_out)
  MSDA_PREP_DISPATCH(L * P, CALL);
#undef CALL
  return check_launch("rscotr_msda_prep_bwd");
}

namespace rscotr {

// DINO decoder query positions: gen_sineembed_for_position (models/multi/bbox_head/transformer.py:43-76):
// pos (rows, 4) = (x, y, w, h) in [0,1] -> out (rows, 512) = [emb(y) | emb(x) | emb(w) | emb(h)], each 128 wide with
// emb(v)[2i] = sin(2*pi*v / 10000^(2i/128)), emb(v)[2i+1] = cos(same).  One thread per output element.
__global__ __launch_bounds__(256) void sine_embed4_kernel(const float* __restrict__ pos, float* __restrict__ out, long rows) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * 512) return;
  const long row = i >> 9;
  const int c = (int)(i & 511), which = c >> 7, d = c & 127;
  const int src = which == 0 ? 1 : (which == 1 ? 0 : which);  // order (y, x, w, h)
  const float dim_t = powf(10000.0f, (float)(2 * (d / 2)) / 128.0f);
  const float v = pos[row * 4 + src] * 6.283185307179586f / dim_t;
  out[i] = (d & 1) ? cosf(v) : sinf(v);
}

}  // namespace rscotr

using namespace rscotr;

extern "C" int rscotr_sine_embed4(const float* pos, float* out, int64_t rows, void* stream) {
  if (rows < 0) return fail(RSCOTR_E_SHAPE, "rscotr_sine_embed4: negative rows");
  if (rows == 0) return RSCOTR_OK;
  if (!pos || !out) return fail(RSCOTR_E_ARG, "rscotr_sine_embed4: null pointer");
  sine_embed4_kernel<<<(unsigned)((rows * 512 + 255) / 256), 256, 0, (hipStream_t)stream>>>(pos, out, rows);
  return check_launch("rscotr_sine_embed4");
}

extern "C" int rscotr_softmax_mask_fwd(float* S, const unsigned char* mask, int mask_mode, int B, int heads, int Lq,
                                       int Lk, float scale, void* stream) {
  if (B < 0 || heads <= 0 || Lq < 0 || Lk <= 0) return fail(RSCOTR_E_SHAPE, "rscotr_softmax_mask_fwd: bad shape");
  if (mask_mode < 0 || mask_mode > 3 || (mask_mode > 0 && !mask)) return fail(RSCOTR_E_ARG, "rscotr_softmax_mask_fwd: bad mask");
  const long rows = (long)B * heads * Lq;
  if (rows == 0) return RSCOTR_OK;
  if (!S) return fail(RSCOTR_E_ARG, "rscotr_softmax_mask_fwd: null pointer");
  const unsigned grid = (unsigned)((rows + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
  static const int vec_on = getenv("RSCOTR_SOFTMAX_VEC") ? atoi(getenv("RSCOTR_SOFTMAX_VEC")) : 3;  // (A/B switch: bit 0 forward, bit 1 backward, bits 4.. = smallest row length)
  const bool vec = (vec_on & 1) && (Lk & 3) == 0 && Lk <= 4096 && Lk >= ((vec_on >> 4) & 0xffff) && aligned16(S) && (!mask || (reinterpret_cast<uintptr_t>(mask) & 3u) == 0);
  if (vec && Lk <= 256) softmax_mask_fwd_vec_kernel<1><<<grid, 256, 0, st>>>(S, mask, mask_mode, rows, Lq, Lk, heads, scale);
  else if (vec && Lk <= 1024) softmax_mask_fwd_vec_kernel<4><<<grid, 256, 0, st>>>(S, mask, mask_mode, rows, Lq, Lk, heads, scale);
  else if (vec && Lk <= 2048) softmax_mask_fwd_vec_kernel<8><<<grid, 256, 0, st>>>(S, mask, mask_mode, rows, Lq, Lk, heads, scale);
  else if (vec) softmax_mask_fwd_vec_kernel<16><<<grid, 256, 0, st>>>(S, mask, mask_mode, rows, Lq, Lk, heads, scale);
  else softmax_mask_fwd_kernel<<<grid, 256, 0, st>>>(S, mask, mask_mode, rows, Lq, Lk, heads, scale);
  return check_launch("rscotr_softmax_mask_fwd");
}

extern "C" int rscotr_softmax_bwd(const float* P, float* dP, int64_t rows, int Lk, float scale, void* stream) {
  if (rows < 0 || Lk <= 0) return fail(RSCOTR_E_SHAPE, "rscotr_softmax_bwd: bad shape");
  if (rows == 0) return RSCOTR_OK;
  if (!P || !dP) return fail(RSCOTR_E_ARG, "rscotr_softmax_bwd: null pointer");
  const unsigned grid = (unsigned)((rows + 3) / 4);
  hipStream_t st = (hipStream_t)stream;
  static const int vec_on = getenv("RSCOTR_SOFTMAX_VEC") ? atoi(getenv("RSCOTR_SOFTMAX_VEC")) : 3;
  const bool vec = (vec_on & 2) && (Lk & 3) == 0 && Lk <= 4096 && Lk >= ((vec_on >> 4) & 0xffff) && aligned16(P) && aligned16(dP);
  if (vec && Lk <= 256) softmax_bwd_vec_kernel<1><<<grid, 256, 0, st>>>(P, dP, rows, Lk, scale);
  else if (vec && Lk <= 1024) softmax_bwd_vec_kernel<4><<<grid, 256, 0, st>>>(P, dP, rows, Lk, scale);
  else if (vec && Lk <= 2048) softmax_bwd_vec_kernel<8><<<grid, 256, 0, st>>>(P, dP, rows, Lk, scale);
  else if (vec) softmax_bwd_vec_kernel<16><<<grid, 256, 0, st>>>(P, dP, rows, Lk, scale);
  else softmax_bwd_kernel<<<grid, 256, 0, st>>>(P, dP, rows, Lk, scale);
  return check_launch("rscotr_softmax_bwd");
}
