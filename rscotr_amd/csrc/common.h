// Shared helpers for the rscotr HIP kernels (gfx950 / CDNA4 only).
//
// Conventions of the C ABI (see include/rscotr.h):
//   * every entry returns int: 0 = ok, <0 = RSCOTR_E_*; the message is kept in a
//     thread-local buffer readable through rscotr_last_error();
//   * the caller owns every buffer (inputs, outputs, workspaces); kernels never
//     allocate, free, or keep pointers past return;
//   * every device entry takes the hipStream_t to launch on and never synchronises.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#define RSCOTR_OK 0
#define RSCOTR_E_SHAPE (-1)
#define RSCOTR_E_ALIGN (-2)
#define RSCOTR_E_ARCH (-3)
#define RSCOTR_E_LAUNCH (-4)
#define RSCOTR_E_ARG (-5)

namespace rscotr {

char* err_buf();  // thread-local, 512 bytes (defined in abi.hip)

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(RSCOTR_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return RSCOTR_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

constexpr int kWave = 64;  // CDNA wavefront width

// Launch-site profiling (abi.hip): when enabled through rscotr_prof_enable(), an entry brackets its launches
// with a pair of HIP events recorded on the launch stream from inside the library (no host code between the
// event and the launch) and remembers the algorithmic work of the call; rscotr_prof_get() returns the event
// durations.  Disabled (the default) it costs one branch.  Never enable it while a stream is capturing.
// PROF_HBM / PROF_MFMA (round 5): every other kernel family worth >= 1 ms of a round, priced by its algorithmic bytes / flops
// under the name of its kernel (LayerNorm, split-K combines, AdamW; Swin window attention) — on whenever any kind is on.
enum { PROF_GEMM = 0, PROF_MSDA_FWD = 1, PROF_MSDA_BWD = 2, PROF_HBM = 3, PROF_MFMA = 4, PROF_KINDS = 5 };
struct ProfScope {
  int slot;
  hipStream_t stream;
  ProfScope(int kind, double work, hipStream_t s, const char* fmt, ...);
  ~ProfScope();
};

// Sum across the lanes of an aligned power-of-two lane group (G <= 64).
template <int G>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int G>
__device__ __forceinline__ float group_max(float v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }
__device__ __forceinline__ float wave_max(float v) { return group_max<64>(v); }

// A value-range WORD is kAmaxPlanes sub-words at stride kAmaxStride (the caller's buffer is [kAmaxPlanes][kAmaxStride] words;
// include/rscotr.h) = 128 BYTES read as an exponent map: byte i != 0 <=> the tensor holds an element whose biased fp32 exponent
// is kRangeExpLo + i (clamped into the window 2^-47 .. 2^80: a smaller maximum is taken for 2^-47 — an upper bound, and the planes keep full
// relative precision down to maxima of 2^-73; nothing a training step holds is larger than 2^80).  A producer wavefront commits its maximum with ONE PLAIN BYTE
// STORE of the constant 1 — idempotent, so any number of wavefronts on any XCD may write the same word with no atomic and no
// order — and a consumer reads the 32 sub-words with one load per lane and takes the highest byte set.  What the word delivers is
// the BINADE of max |x| (2^e <= max |x| < 2^(e+1)), which is all the fp16 split product uses (h3_scale_exp reads the exponent field
// only: the scales are bit for bit the ones the full maximum gave); bounds built from words take the upper end, amax_hi.
// (Round 6.  Until then a wavefront folded its maximum into sub-word (XCD, wavefront) with an atomicMax on the bit pattern: a few
// hundred to a few thousand L2 atomics per producer launch on four lines per XCD.  In the step, one box, alternating: 29.20 / 29.30 ms
// per round with the byte marks, 29.39 / 29.68 with the atomics.  Two timing-only builds had promised 2.1 ms — a plain store of the
// wavefront's maximum, and a racy load - compare - store — but both leave words that are too SMALL, and a step whose planes overflow
// runs a different, shorter course through its losses: profiles/r6_range_words.txt.)
constexpr int kAmaxPlanes = 32;
constexpr int kAmaxStride = 16384;
constexpr int kRangeExpLo = 80;  // biased exponent of byte 0

// byte index (0 .. 127) of a non-zero |x| bit pattern
__device__ __forceinline__ int range_byte(unsigned bits) {
  return min(max((int)((bits >> 23) & 0xffu), kRangeExpLo), kRangeExpLo + 4 * kAmaxPlanes - 1) - kRangeExpLo;
}
// the plain store that marks byte i of the word at `slot`
__device__ __forceinline__ void range_mark(unsigned* slot, int i) {
  reinterpret_cast<unsigned char*>(slot + (long)(i >> 2) * kAmaxStride)[i & 3] = 1;
}

// (lane l holds sub-word l % kAmaxPlanes, loaded by the caller as early as it likes) -> the bit pattern of 2^e, e = the binade of
// the tensor's maximum (0 for a word nobody marked: an all-zero tensor); wave-uniform
__device__ __forceinline__ unsigned amax_fold(unsigned v) {
  int c = v ? 4 * (int)(threadIdx.x & (kAmaxPlanes - 1)) + ((31 - __clz((int)v)) >> 3) + 1 : 0;  // byte index + 1
#pragma unroll
  for (int o = kAmaxPlanes / 2; o > 0; o >>= 1) c = max(c, __shfl_xor(c, o, 64));
  c = __builtin_amdgcn_readfirstlane(c);
  return c ? (unsigned)(kRangeExpLo + c - 1) << 23 : 0u;
}
__device__ __forceinline__ unsigned amax_read(const unsigned* word) {
  return amax_fold(word[(long)(threadIdx.x & (kAmaxPlanes - 1)) * kAmaxStride]);
}
// upper end of the binade amax_fold returned: max |x| < amax_hi (for bounds computed from words)
__device__ __forceinline__ float amax_hi(unsigned folded) { return folded ? __uint_as_float(folded + (1u << 23)) : 0.f; }

// End of a kernel: the wavefront's maximum marks its byte.  Every lane of the wavefront must reach this call.
__device__ __forceinline__ void amax_commit(unsigned* slot, float amx) {
  if (!slot) return;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amx = fmaxf(amx, __shfl_xor(amx, o, 64));
  if ((threadIdx.x & 63) == 0) {
    const unsigned b = __float_as_uint(amx) & 0x7fffffffu;
#ifdef RSCOTR_RANGE_UNDER  // (experiment, profiles/r6_range_words.txt: words RSCOTR_RANGE_UNDER binades too small — what a lost maximum does to a step)
    if (b) range_mark(slot, max(range_byte(b) - RSCOTR_RANGE_UNDER, 0));
#else
    if (b) range_mark(slot, range_byte(b));
#endif
  }
}

// Sampling location of mmcv MultiScaleDeformableAttention.forward from a reference point and a raw offset (ONE definition for the
// stand-alone prologue kernel, csrc/attn.hip, and the forward kernel that does the prologue itself, csrc/msda.hip):
//   2-d reference points: ref_xy + off / (W_l, H_l);   4-d: ref_xy + off / P * ref_wh * 0.5
__device__ __forceinline__ float2 msda_location(const float* __restrict__ r, float2 o, const float* __restrict__ norm, int l, int P,
                                                int refdim) {
  float2 out;
  if (refdim == 2) {
    out.x = r[0] + o.x / norm[2 * l];
    out.y = r[1] + o.y / norm[2 * l + 1];
  } else {
    out.x = r[0] + o.x / (float)P * r[2] * 0.5f;
    out.y = r[1] + o.y / (float)P * r[3] * 0.5f;
  }
  return out;
}

__device__ __forceinline__ float amax4(float a, const float4& v) {
  return fmaxf(fmaxf(a, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}

}  // namespace rscotr
