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
// include/rscotr.h): its value is the maximum of the sub-words.  Producers spread their atomics over the sub-words (a
// wavefront takes sub-word (4 * workgroup + wavefront) % kAmaxPlanes; the planes are 64 KB apart: different memory channels),
// consumers read all of them with one load per lane.
constexpr int kAmaxPlanes = 32;
constexpr int kAmaxStride = 16384;

// (lane l holds sub-word l % kAmaxPlanes, loaded by the caller as early as it likes) -> the word's value, wave-uniform
__device__ __forceinline__ unsigned amax_fold(unsigned v) {
#pragma unroll
  for (int o = kAmaxPlanes / 2; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o, 64));
  return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ unsigned amax_read(const unsigned* word) {
  return amax_fold(word[(long)(threadIdx.x & (kAmaxPlanes - 1)) * kAmaxStride]);
}

// Same-address atomics execute one after the other at the memory side (~10 ns each: thousands of wavefronts finishing
// together would cost more than a short product itself), so a wavefront first LOOKS at the word (a plain, possibly stale
// read: staleness only costs an atomic that changes nothing) and stays silent unless it raises it — operands of one
// tensor are of one magnitude, the first few finishers settle the word.
__device__ __forceinline__ void amax_commit(unsigned* slot, float amx) {
  if (!slot) return;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amx = fmaxf(amx, __shfl_xor(amx, o, 64));
  if ((threadIdx.x & 63) == 0) {
    const unsigned b = __float_as_uint(amx);
#ifdef RSCOTR_AMAX_BY_BLOCK
    unsigned* sub = slot + (long)((blockIdx.x * 4 + (threadIdx.x >> 6)) & (kAmaxPlanes - 1)) * kAmaxStride;
#else
    // sub-word by (XCD, wavefront of the workgroup): a line is only ever touched by the atomics of ONE XCD's L2
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));  // HW_REG_XCC_ID
    unsigned* sub = slot + (long)(((xcc & 7) * 4 + ((threadIdx.x >> 6) & 3)) & (kAmaxPlanes - 1)) * kAmaxStride;
#endif
    if (b) atomicMax(sub, b);  // (fire and forget: nothing waits for the result; a look at the word first would put a second memory round trip at the end of every wavefront)
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
