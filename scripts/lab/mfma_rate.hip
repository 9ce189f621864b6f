// Calibration: sustained rate of v_mfma_f32_32x32x16_bf16 from registers (no memory traffic) at 1 / 2 / 4 wavefronts per SIMD, four
// independent accumulators (the product loops' pattern), and the shader clock the chip holds meanwhile (clock64 = s_memtime
// against wall_clock64 = the 100 MHz constant counter) — what "MFMA-bound" means in wall time on this part.  Variant LDS = 1:
// every 24 MFMAs are preceded by 12 ds_read_b128 of fresh fragments (the product loops' LDS read volume, no barrier).
// hipcc --offload-arch=gfx950 -O3 scripts/lab/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int LDS>
__global__ __launch_bounds__(256) void rate_kernel(float* out, long long* clk, int iters) {
  __shared__ __attribute__((aligned(16))) unsigned lds[128 * 28 * 2];
  for (int i = threadIdx.x; i < 128 * 28 * 2; i += 256) lds[i] = 0x3f803f80u + i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned* qa = lds + ((wave >> 1) * 64 + (lane & 31)) * 28 + 4 * (lane >> 5);
  const unsigned* qb = lds + 128 * 28 + ((wave & 1) * 64 + (lane & 31)) * 28 + 4 * (lane >> 5);
  bf16x8 a[2][3], b[2][3];
  for (int i = 0; i < 2; ++i) for (int pl = 0; pl < 3; ++pl) {
    a[i][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(qa + i * 32 * 28 + pl * 8));
    b[i][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(qb + i * 32 * 28 + pl * 8));
  }
  f32x16 acc[4];
  for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (LDS) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          a[i][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const volatile uint4*>(qa + i * 32 * 28 + pl * 8));
          b[i][pl] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const volatile uint4*>(qb + i * 32 * 28 + pl * 8));
        }
    }
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int tm = 0; tm < 6; ++tm)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[n >> 1][PA[tm]], b[n & 1][PB[tm]], acc[n], 0, 0, 0);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

int main() {
  float* out;
  long long *clk, hclk[2 * 1024];
  (void)hipMalloc(&out, 4096 * 256 * 4);
  (void)hipMalloc(&clk, 2 * 1024 * 8);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 4000;
  for (int lds = 0; lds < 2; ++lds)
    for (int wgs : {256, 512, 1024}) {
      float ms = 0.f;
      for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        if (lds) rate_kernel<1><<<wgs, 256>>>(out, clk, iters);
        else rate_kernel<0><<<wgs, 256>>>(out, clk, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
      }
      (void)hipMemcpy(hclk, clk, 2 * wgs * 8, hipMemcpyDeviceToHost);
      double cs = 0, ws = 0;
      for (int i = 0; i < wgs; ++i) { cs += hclk[2 * i]; ws += hclk[2 * i + 1]; }
      const double flops = (double)iters * 24 * 4 * wgs * 32768.0;
      printf("lds=%d %4d workgroups x 4 waves: %.3f ms, %.0f TFLOP/s bf16, shader clock %.0f MHz (clock64 / wall_clock64 x 100 MHz), "
             "%.1f cycles per MFMA and wavefront slot\n", lds, wgs, ms, flops / ms / 1e9, cs / ws * 100.0,
             cs / wgs / ((double)iters * 24) / ((wgs + 255) / 256));
    }
  return 0;
}
