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
// taken from max |w| of its tensor): seg_amax[segment] = bit pattern of max |w|.  The update kernel below refreshes the word
// of every segment it touches — amax_reset_kernel zeroes the live segments' words, each chunk's workgroup then folds its
// maximum in with one atomicMax on the bit pattern (a maximum does not depend on the order: deterministic; the plain read
// in front only skips atomics that cannot raise the word).  Segments that are not stepped keep their word.
__global__ __launch_bounds__(256) void amax_reset_kernel(const float* __restrict__ seg_dyn, unsigned* __restrict__ seg_amax, int nseg) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < nseg && (!seg_dyn || seg_dyn[i * SEG_STRIDE + 4] != 0.f)) seg_amax[i] = 0u;
}

__device__ __forceinline__ void chunk_amax_commit(unsigned* __restrict__ word, float amx) {
  __shared__ float part_amax[4];
  amx = wave_max(amx);
  if ((threadIdx.x & 63) == 0) part_amax[threadIdx.x >> 6] = amx;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned m = __float_as_uint(fmaxf(fmaxf(part_amax[0], part_amax[1]), fmaxf(part_amax[2], part_amax[3])));
    if (m > *reinterpret_cast<volatile unsigned*>(word)) atomicMax(word, m);
  }
}

__global__ __launch_bounds__(256) void param_amax_kernel(const float* __restrict__ param, const int32_t* __restrict__ chunk_seg,
                                                         const int64_t* __restrict__ chunk_off,
                                                         const int32_t* __restrict__ chunk_len, unsigned* __restrict__ seg_amax) {
  const int c = blockIdx.x;
  const float4* p4 = reinterpret_cast<const float4*>(param + chunk_off[c]);
  const int n4 = chunk_len[c] >> 2;
  float amx = 0.f;
  for (int i = threadIdx.x; i < n4; i += 256) {
    const float4 p = p4[i];
    amx = fmaxf(fmaxf(amx, fmaxf(fabsf(p.x), fabsf(p.y))), fmaxf(fabsf(p.z), fabsf(p.w)));
  }
  chunk_amax_commit(seg_amax + chunk_seg[c], amx);
}

__global__ __launch_bounds__(256) void adamw_clip_kernel(
    float* __restrict__ param, const float* __restrict__ grad, float* __restrict__ exp_avg,
    float* __restrict__ exp_avg_sq, const int32_t* __restrict__ chunk_seg,
    const int64_t* __restrict__ chunk_off, const int32_t* __restrict__ chunk_len,
    const float* __restrict__ seg_dyn, const float* __restrict__ sumsq, float max_norm, float beta1,
    float beta2, float eps, unsigned* __restrict__ seg_amax) {
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
  for (int i = threadIdx.x; i < n4; i += 256) {
    float4 p = p4[i], g = g4[i], m = m4[i], v = v4[i];
#define RSCOTR_ADAMW_1(X)                                   \
  {                                                         \
    const float gg = g.X * coef;                            \
    p.X *= decay;                                           \
    m.X = m.X * beta1 + omb1 * gg;                          \
    v.X = v.X * beta2 + omb2 * gg * gg;                     \
    const float denom = sqrtf(v.X) * inv_sqrt_bc2 + eps;    \
    p.X -= step_size * (m.X / denom);                       \
  }
    RSCOTR_ADAMW_1(x) RSCOTR_ADAMW_1(y) RSCOTR_ADAMW_1(z) RSCOTR_ADAMW_1(w)
#undef RSCOTR_ADAMW_1
    p4[i] = p;
    m4[i] = m;
    v4[i] = v;
    amx = fmaxf(fmaxf(amx, fmaxf(fabsf(p.x), fabsf(p.y))), fmaxf(fabsf(p.z), fabsf(p.w)));
  }
  if (seg_amax) chunk_amax_commit(seg_amax + chunk_seg[c], amx);
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
                                        float eps, uint32_t* seg_amax, int nseg, void* stream);

extern "C" int rscotr_adamw_clip_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                                      const int32_t* chunk_seg, const int64_t* chunk_off,
                                      const int32_t* chunk_len, const float* seg_dyn, int nchunks,
                                      const float* sumsq, float max_norm, float beta1, float beta2,
                                      float eps, void* stream) {
  return rscotr_adamw_clip_step_r(param, grad, exp_avg, exp_avg_sq, chunk_seg, chunk_off, chunk_len, seg_dyn, nchunks, sumsq,
                                  max_norm, beta1, beta2, eps, nullptr, 0, stream);
}

extern "C" int rscotr_param_amax(const float* param, const int32_t* chunk_seg, const int64_t* chunk_off,
                                 const int32_t* chunk_len, int nchunks, uint32_t* seg_amax, int nseg, void* stream) {
  if (nchunks < 0 || nseg < 0) return fail(RSCOTR_E_SHAPE, "rscotr_param_amax: negative count");
  if (nchunks == 0 || nseg == 0) return RSCOTR_OK;
  if (!param || !chunk_seg || !chunk_off || !chunk_len || !seg_amax) return fail(RSCOTR_E_ARG, "rscotr_param_amax: null pointer");
  amax_reset_kernel<<<dim3((nseg + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(nullptr, seg_amax, nseg);
  param_amax_kernel<<<dim3(nchunks), dim3(256), 0, (hipStream_t)stream>>>(param, chunk_seg, chunk_off, chunk_len, seg_amax);
  return check_launch("rscotr_param_amax");
}

extern "C" int rscotr_adamw_clip_step_r(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                                        const int32_t* chunk_seg, const int64_t* chunk_off,
                                        const int32_t* chunk_len, const float* seg_dyn, int nchunks,
                                        const float* sumsq, float max_norm, float beta1, float beta2,
                                        float eps, uint32_t* seg_amax, int nseg, void* stream) {
  if (nchunks < 0) return fail(RSCOTR_E_SHAPE, "rscotr_adamw_clip_step: negative chunk count");
  if (nchunks == 0) return RSCOTR_OK;
  if (!param || !grad || !exp_avg || !exp_avg_sq || !chunk_seg || !chunk_off || !chunk_len || !seg_dyn)
    return fail(RSCOTR_E_ARG, "rscotr_adamw_clip_step: null pointer");
  if (max_norm > 0.f && !sumsq) return fail(RSCOTR_E_ARG, "rscotr_adamw_clip_step: sumsq required when clipping");
  if (!aligned16(param) || !aligned16(grad) || !aligned16(exp_avg) || !aligned16(exp_avg_sq))
    return fail(RSCOTR_E_ALIGN, "rscotr_adamw_clip_step: arenas must be 16-byte aligned");
  if (seg_amax && nseg > 0)
    amax_reset_kernel<<<dim3((nseg + 255) / 256), dim3(256), 0, (hipStream_t)stream>>>(seg_dyn, seg_amax, nseg);
  // (work 0: which segments are live is device data — the caller prices the launch: 28 bytes per stepped element)
  ProfScope prof(PROF_HBM, 0.0, (hipStream_t)stream, "rscotr::adamw_clip_kernel");
  adamw_clip_kernel<<<dim3(nchunks), dim3(256), 0, (hipStream_t)stream>>>(
      param, grad, exp_avg, exp_avg_sq, chunk_seg, chunk_off, chunk_len, seg_dyn, sumsq, max_norm, beta1,
      beta2, eps, seg_amax);
  return check_launch("rscotr_adamw_clip_step");
}
