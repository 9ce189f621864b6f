// Calibration: sustained rate of v_mfma_f32_32x32x16_bf16 from registers (no memory traffic) at 1 / 2 wavefronts per SIMD, with 4
// independent accumulators (the product loops' pattern) — what "MFMA-bound" means in wall time on this part.
// hipcc --offload-arch=gfx950 -O3 scripts/lab/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
  f32x16 acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 6; ++rep)
#pragma unroll
      for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[n], 0, 0, 0);
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 4096 * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  for (int wgs : {256, 512, 1024}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      rate_kernel<4><<<wgs, 256>>>(out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double mfma_per_wave = (double)iters * 24, flops = mfma_per_wave * 4 * wgs * 32768.0;
      if (rep) printf("%4d workgroups x 4 waves: %.3f ms, %.1f ns per 24 MFMAs per wave-slot, %.0f TFLOP/s bf16\n", wgs, ms,
                      ms * 1e6 / iters / ((wgs + 255) / 256), flops / ms / 1e9);
    }
  }
  return 0;
}
