// Fused global-norm clip + AdamW over a flat parameter arena (gfx950).
//
// Replaces, for the whole model in two launches, what the reference does with hundreds of tiny
// kernels per step through mmcv's OptimizerHook + torch.optim.AdamW (torch 1.11 for-each path):
//   clip_grad_norm_(params with grad, max_norm=0.1, norm_type=2); optimizer.step()
// (cfg configs/multi/MTL_slvlcls_...potsdam.py:203-213; hook registered at mtl/apis/train.py:66-83;
// one param group PER PARAMETER from mtl/utils/optimizer.py:40-55).
//
// Layout: parameters, gradients and both moments live in four flat fp32 arenas with identical
// offsets; each tensor ("segment") starts on a 16-byte boundary.  The arena is cut into chunks of
// at most CHUNK elements that never straddle a segment; chunk tables are static, the per-segment
// dynamic row {lr, weight_decay, 1/bias_correction1, 1/sqrt(bias_correction2), live} is
// refreshed by the host each step (bias corrections are per segment because, with torch-1.11
// semantics, a tensor's step count starts when it first receives a gradient).
// Pure HBM streaming: 16 B read + 12 B written per element in the update, 4 B in the norm pass.
#include "common.h"

namespace rscotr {

constexpr int SEG_STRIDE = 8;  // floats per segment row

// Grid-stride over the chunk table: a workgroup folds all its chunks locally and stores ONE partial; a one-workgroup
// kernel folds the <= SUMSQ_PARTS partials in fixed order (no atomics: the clip coefficient, hence every updated weight,
// is bit-reproducible).  (One atomic per 4096-element chunk = 15 k same-address fp32 atomics per step, which execute one
// after the other at the memory side on MI355X: 143 us for a 200 MB pass.)
constexpr int SUMSQ_PARTS = 1024;
__global__ __launch_bounds__(256) void grad_sumsq_kernel(
    const float* __restrict__ grad, const int32_t* __restrict__ chunk_seg,
    const int64_t* __restrict__ chunk_off, const int32_t* __restrict__ chunk_len,
    const float* __restrict__ seg_dyn, float* __restrict__ sumsq, int nchunks) {
  float acc = 0.f;
  for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const int seg = chunk_seg[c];
    if (seg_dyn[seg * SEG_STRIDE + 4] == 0.f) continue;  // tensor has never received a gradient
    const float4* g = reinterpret_cast<const float4*>(grad + chunk_off[c]);
    const int n4 = chunk_len[c] >> 2;
    for (int i = threadIdx.x; i < n4; i += 256) {
      const float4 v = g[i];
      acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
  }
  acc = wave_sum(acc);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) sumsq[1 + blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(256) void grad_sumsq_final_kernel(float* __restrict__ sumsq, int nparts) {
  float acc = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) acc += sumsq[1 + i];
  acc = wave_sum(acc);
  __shared__ float part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) sumsq[0] = part[0] + part[1] + part[2] + part[3];
}

// Value ranges of the parameters (round 5: the fp16 split product of csrc/gemm.hip scales a weight operand by a power of two
// taken from max |w| of its tensor): seg_amax[segment] = bit pattern of max |w|.  The update kernel below leaves the maximum
// of every wavefront's share of a chunk in chunk_amax[4 * chunk + wavefront] (plain stores), seg_amax_reduce_kernel — one
// wavefront per segment, the segment's chunks found by bisection of the sorted chunk_seg — folds them into the words of the
// segments that were stepped; the others keep theirs.  No atomics: the first cut folded each workgroup's maximum into the
// word with atomicMax (one per workgroup behind a look at the word: 285 -> 345 us per update of the 62 M elements of
// configs[1]; one per wavefront against a look taken at the start: 575-1000 us — a few thousand atomics on a handful of
// lines serialise at ~50-100 ns each and the kernel cannot retire before them).
__global__ __launch_bounds__(256) void seg_amax_reduce_kernel(const int32_t* __restrict__ chunk_seg, const unsigned* __restrict__ chunk_amax,
                                                              const float* __restrict__ seg_dyn, unsigned* __restrict__ seg_amax,
                                                              int nseg, int nchunks) {
  const int sgm = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (sgm >= nseg || (seg_dyn && seg_dyn[sgm * SEG_STRIDE + 4] == 0.f)) return;  // (wavefront-uniform)
  // first chunk of this segment and of the next: two bisections of the ascending chunk_seg, stepped together (one memory
  // latency per step for both)
  int lo0 = 0, hi0 = nchunks, lo1 = 0, hi1 = nchunks;
  while (lo0 < hi0 || lo1 < hi1) {
    const int m0 = (lo0 + hi0) >> 1, m1 = (lo1 + hi1) >> 1;
    const int v0 = lo0 < hi0 ? chunk_seg[m0] : 0, v1 = lo1 < hi1 ? chunk_seg[m1] : 0;
    if (lo0 < hi0) { if (v0 < sgm) lo0 = m0 + 1; else hi0 = m0; }
    if (lo1 < hi1) { if (v1 < sgm + 1) lo1 = m1 + 1; else hi1 = m1; }
  }
  const int c0 = lo0, c1 = lo1;
  unsigned m = 0u;  // (bit patterns of non-negative floats order like the floats)
  for (int c = c0 + lane; c < c1; c += 64) {
    const uint4 v = reinterpret_cast<const uint4*>(chunk_amax)[c];
    m = max(max(m, max(v.x, v.y)), max(v.z, v.w));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
  // the parameter's word in the exponent-map form of common.h: persistent, so all 32 sub-words are rewritten (one lane each)
  if (lane < kAmaxPlanes) {
    const int i = m ? range_byte(m) : -1;
    seg_amax[sgm + (long)lane * kAmaxStride] = (i >= 0 && (i >> 2) == lane) ? (1u << (8 * (i & 3))) : 0u;
  }
}

__device__ __forceinline__ void chunk_amax_store(unsigned* __restrict__ chunk_amax, int c, float amx) {
  amx = wave_max(amx);
  if ((threadIdx.x & 63) == 0) chunk_amax[4 * c + (threadIdx.x >> 6)] = __float_as_uint(amx);
}

__global__ __launch_bounds__(256) void param_amax_kernel(const float* __restrict__ param, const int32_t* __restrict__ chunk_seg,
                                                         const int64_t* __restrict__ chunk_off,
                                                         const int32_t* __restrict__ chunk_len, unsigned* __restrict__ chunk_amax) {
  const int c = blockIdx.x;
  const float4* p4 = reinterpret_cast<const float4*>(param + chunk_off[c]);
  const int n4 = chunk_len[c] >> 2;
  float amx = 0.f;
  for (int i = threadIdx.x; i < n4; i += 256) {
    const float4 p = p4[i];
    amx = fmaxf(fmaxf(amx, fmaxf(fabsf(p.x), fabsf(p.y))), fmaxf(fabsf(p.z), fabsf(p.w)));
  }
  chunk_amax_store(chunk_amax, c, amx);
}

__global__ __launch_bounds__(256) void adamw_clip_kernel(
    float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ exp_avg,
    float* __restrict__ exp_avg_sq, const int32_t* __restrict__ chunk_seg,
    const int64_t* __restrict__ chunk_off, const int32_t* __restrict__ chunk_len,
    const float* __restrict__ seg_dyn, const float* __restrict__ sumsq, float max_norm, float beta1,
    float beta2, float eps, unsigned* __restrict__ chunk_amax) {
  const int c = blockIdx.x;
  const float* d = seg_dyn + chunk_seg[c] * SEG_STRIDE;
  if (d[4] == 0.f) return;
  const float lr = d[0], wd = d[1], inv_bc1 = d[2], inv_sqrt_bc2 = d[3];
  float coef = 1.f;
  if (max_norm > 0.f) {
    // torch.nn.utils.clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1
    coef = fminf(max_norm / (sqrtf(*sumsq) + 1e-6f), 1.f);
  }
  const int64_t off = chunk_off[c];
  float4* p4 = reinterpret_cast<float4*>(param + off);
  const float4* g4 = reinterpret_cast<const float4*>(grad + off);
  float4* m4 = reinterpret_cast<float4*>(exp_avg + off);
  float4* v4 = reinterpret_cast<float4*>(exp_avg_sq + off);
  const int n4 = chunk_len[c] >> 2;
  const float decay = 1.f - lr * wd;
  const float step_size = lr * inv_bc1;
  const float omb1 = 1.f - beta1, omb2 = 1.f - beta2;
  float amx = 0.f;
#define RSCOTR_ADAMW_1(X)                                   \
  {                                                         \
    const float gg = g.X * coef;                            \
    p.X *= decay;                                           \
    m.X = m.X * beta1 + omb1 * gg;                          \
    v.X = v.X * beta2 + omb2 * gg * gg;                     \
    const float denom = sqrtf(v.X) * inv_sqrt_bc2 + eps;    \
    p.X -= step_size * (m.X / denom);                       \
  }
  auto step4 = [&](int i, float4 p, const float4& g, float4 m, float4 v) {
    RSCOTR_ADAMW_1(x) RSCOTR_ADAMW_1(y) RSCOTR_ADAMW_1(z) RSCOTR_ADAMW_1(w)
    p4[i] = p;
    m4[i] = m;
    v4[i] = v;
    amx = fmaxf(fmaxf(amx, fmaxf(fabsf(p.x), fabsf(p.y))), fmaxf(fabsf(p.z), fabsf(p.w)));
  };
  // two elements of the thread's stride requested before either is stepped (the running maximum kept the compiler from
  // unrolling the plain loop: one iteration's four loads in flight)
  int i = threadIdx.x;
  for (; i + 256 < n4; i += 512) {
    const int j = i + 256;
    const float4 p0 = p4[i], g0 = g4[i], m0 = m4[i], v0 = v4[i];
    const float4 p1 = p4[j], g1 = g4[j], m1 = m4[j], v1 = v4[j];
    asm volatile("" ::: "memory");  // (all eight requests before the first store: the scheduler sank three of them behind it)
    step4(i, p0, g0, m0, v0);
    step4(j, p1, g1, m1, v1);
  }
  if (i < n4) step4(i, p4[i], g4[i], m4[i], v4[i]);
#undef RSCOTR_ADAMW_1
  if (chunk_amax) chunk_amax_store(chunk_amax, c, amx);
}

}  // namespace rscotr

using namespace rscotr;

extern "C" int rscotr_grad_sumsq(const float* grad, const int32_t* chunk_seg, const int64_t* chunk_off,
                                 const int32_t* chunk_len, const float* seg_dyn, int nchunks,
                                 float* sumsq, void* stream) {
  if (nchunks < 0) return fail(RSCOTR_E_SHAPE, "rscotr_grad_sumsq: negative chunk count");
  if (nchunks == 0) return RSCOTR_OK;
  if (!grad || !chunk_seg || !chunk_off || !chunk_len || !seg_dyn || !sumsq)
    return fail(RSCOTR_E_ARG, "rscotr_grad_sumsq: null pointer");
  if (!aligned16(grad)) return fail(RSCOTR_E_ALIGN, "rscotr_grad_sumsq: grad must be 16-byte aligned");
  const int nparts = nchunks < SUMSQ_PARTS ? nchunks : SUMSQ_PARTS;
  grad_sumsq_kernel<<<dim3(nparts), dim3(256), 0, (hipStream_t)stream>>>(
      grad, chunk_seg, chunk_off, chunk_len, seg_dyn, sumsq, nchunks);
  grad_sumsq_final_kernel<<<1, 256, 0, (hipStream_t)stream>>>(sumsq, nparts);
  return check_launch("rscotr_grad_sumsq");
}

extern "C" int rscotr_adamw_clip_step_r(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                                        const int32_t* chunk_seg, const int64_t* chunk_off,
                                        const int32_t* chunk_len, const float* seg_dyn, int nchunks,
                                        const float* sumsq, float max_norm, float beta1, float beta2,
                                        float eps, uint32_t* seg_amax, int nseg, uint32_t* chunk_amax, void* stream);

extern "C" int rscotr_adamw_clip_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                                      const int32_t* chunk_seg, const int64_t* chunk_off,
                                      const int32_t* chunk_len, const float* seg_dyn, int nchunks,
                                      const float* sumsq, float max_norm, float beta1, float beta2,
                                      float eps, void* stream) {
  return rscotr_adamw_clip_step_r(param, grad, exp_avg, exp_avg_sq, chunk_seg, chunk_off, chunk_len, seg_dyn, nchunks, sumsq,
                                  max_norm, beta1, beta2, eps, nullptr, 0, nullptr, stream);
}

extern "C" int rscotr_param_amax(const float* param, const int32_t* chunk_seg, const int64_t* chunk_off,
                                 const int32_t* chunk_len, int nchunks, uint32_t* seg_amax, int nseg, uint32_t* chunk_amax,
                                 void* stream) {
  if (nchunks < 0 || nseg < 0) return fail(RSCOTR_E_SHAPE, "rscotr_param_amax: negative count");
  if (nchunks == 0 || nseg == 0) return RSCOTR_OK;
  if (!param || !chunk_seg || !chunk_off || !chunk_len || !seg_amax || !chunk_amax) return fail(RSCOTR_E_ARG, "rscotr_param_amax: null pointer");
  if (!aligned16(chunk_amax)) return fail(RSCOTR_E_ALIGN, "rscotr_param_amax: chunk_amax must be 16-byte aligned");
  param_amax_kernel<<<dim3(nchunks), dim3(256), 0, (hipStream_t)stream>>>(param, chunk_seg, chunk_off, chunk_len, chunk_amax);
  seg_amax_reduce_kernel<<<dim3((nseg + 3) / 4), dim3(256), 0, (hipStream_t)stream>>>(chunk_seg, chunk_amax, nullptr, seg_amax, nseg, nchunks);
  return check_launch("rscotr_param_amax");
}

extern "C" int rscotr_adamw_clip_step_r(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                                        const int32_t* chunk_seg, const int64_t* chunk_off,
                                        const int32_t* chunk_len, const float* seg_dyn, int nchunks,
                                        const float* sumsq, float max_norm, float beta1, float beta2,
                                        float eps, uint32_t* seg_amax, int nseg, uint32_t* chunk_amax, void* stream) {
  if (nchunks < 0) return fail(RSCOTR_E_SHAPE, "rscotr_adamw_clip_step: negative chunk count");
  if (nchunks == 0) return RSCOTR_OK;
  if (!param || !grad || !exp_avg || !exp_avg_sq || !chunk_seg || !chunk_off || !chunk_len || !seg_dyn)
    return fail(RSCOTR_E_ARG, "rscotr_adamw_clip_step: null pointer");
  if (max_norm > 0.f && !sumsq) return fail(RSCOTR_E_ARG, "rscotr_adamw_clip_step: sumsq required when clipping");
  if (!aligned16(param) || !aligned16(grad) || !aligned16(exp_avg) || !aligned16(exp_avg_sq))
    return fail(RSCOTR_E_ALIGN, "rscotr_adamw_clip_step: arenas must be 16-byte aligned");
  const bool ranges = seg_amax && nseg > 0;
  if (ranges && !chunk_amax) return fail(RSCOTR_E_ARG, "rscotr_adamw_clip_step_r: chunk_amax (4 * nchunks words) required with seg_amax");
  if (ranges && !aligned16(chunk_amax)) return fail(RSCOTR_E_ALIGN, "rscotr_adamw_clip_step_r: chunk_amax must be 16-byte aligned");
  // (work 0: which segments are live is device data — the caller prices the launch: 28 bytes per stepped element)
  ProfScope prof(PROF_HBM, 0.0, (hipStream_t)stream, "rscotr::adamw_clip_kernel");
  adamw_clip_kernel<<<dim3(nchunks), dim3(256), 0, (hipStream_t)stream>>>(
      param, grad, exp_avg, exp_avg_sq, chunk_seg, chunk_off, chunk_len, seg_dyn, sumsq, max_norm, beta1,
      beta2, eps, ranges ? chunk_amax : nullptr);
  if (ranges)
    seg_amax_reduce_kernel<<<dim3((nseg + 3) / 4), dim3(256), 0, (hipStream_t)stream>>>(chunk_seg, chunk_amax, seg_dyn, seg_amax, nseg, nchunks);
  return check_launch("rscotr_adamw_clip_step");
}
