// Do VALU instructions of ONE wavefront issue while ANOTHER wavefront of the same SIMD runs MFMAs?  512-thread workgroups =
// two wavefronts per SIMD: wavefronts 0-3 run `iters` x 24 v_mfma_f32_32x32x16_bf16, wavefronts 4-7 run `iters` x 160 VALU
// instructions of the three-plane conversion (v_cvt_pk_bf16_f32, shifts / masks, v_pk_add_f32); each kind also alone.
// hipcc --offload-arch=gfx950 -O3 scripts/lab/mfma_valu_overlap.hip -o /tmp/ov && /tmp/ov
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void ov_kernel(float* out, long long* clk, int iters, int mode) {  // mode 1: MFMA only, 2: VALU only, 3: both
  const int wave = threadIdx.x >> 6;
  const bool is_mfma = wave < 4;
  float res = 0.f;
  const long long c0 = clock64();
  if (is_mfma && (mode & 1)) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
    f32x16 acc[4];
    for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 6; ++rep)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
    }
    for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) res += acc[n][r];
  } else if (!is_mfma && (mode & 2)) {
    f32x2 e[8];
    for (int i = 0; i < 8; ++i) e[i] = f32x2{1.0f + 0.001f * threadIdx.x + i, 2.0f - 0.002f * threadIdx.x - i};
    unsigned sink = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 4; ++rep)   // 4 x (8 x (cvt + shift + mask + pk_add + ...)) ~ 160 VALU
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const bf16x2 h = __builtin_convertvector(e[i], bf16x2);
          const unsigned hu = __builtin_bit_cast(unsigned, h);
          const f32x2 f = {__uint_as_float(hu << 16), __uint_as_float(hu & 0xffff0000u)};
          e[i] = e[i] - f + f32x2{1.0f, 2.0f};
          sink ^= hu;
        }
    }
    for (int i = 0; i < 8; ++i) res += e[i].x + e[i].y;
    res += (float)sink;
  }
  const long long c1 = clock64();
  out[blockIdx.x * 512 + threadIdx.x] = res;
  if ((threadIdx.x & 63) == 0) clk[blockIdx.x * 8 + wave] = c1 - c0;
}

int main() {
  float* out;
  long long *clk, h[256 * 8];
  (void)hipMalloc(&out, 256 * 512 * 4);
  (void)hipMalloc(&clk, 256 * 8 * 8);
  const int iters = 2000;
  for (int mode = 1; mode <= 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) ov_kernel<<<256, 512>>>(out, clk, iters, mode);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0, v = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += h[b * 8 + w];
    printf("mode %d (%s): MFMA wavefronts %.0f cycles per 24 MFMAs, VALU wavefronts %.0f cycles per iteration\n", mode,
           mode == 1 ? "MFMA alone" : mode == 2 ? "VALU alone" : "both on every SIMD", m / 1024 / iters, v / 1024 / iters);
  }
  return 0;
}
