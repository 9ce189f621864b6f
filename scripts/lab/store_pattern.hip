// How fast does a (M x N) fp32 result leave the chip under the store patterns a GEMM epilogue can produce?  One wavefront owns a
// 64 x 64 tile (4 wavefronts = a 128 x 128 workgroup tile), values come from registers.
//   0: MFMA C-layout stores — 4 bytes per lane, a half-wave = 128 contiguous bytes of one row (the shipped epilogues)
//   1: 16 bytes per lane, 16 lanes = one 256-byte row segment of the wave tile, 4 rows per instruction (after an LDS transpose)
//   2: 16 bytes per lane, 32 lanes = one 512-byte row segment (two wavefronts' columns), 2 rows per instruction
//   3: 16 bytes per lane, 1 KB contiguous per instruction (a fill: the ceiling)
// hipcc --offload-arch=gfx950 -O3 scripts/lab/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(float* __restrict__ C, int M, int N, float seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles_n = N / 128;
  const int m0 = (blockIdx.x / tiles_n) * 128 + (wave >> 1) * 64, n0 = (blockIdx.x % tiles_n) * 128 + (wave & 1) * 64;
  if (MODE == 0) {
    const int fr = lane & 31, g = lane >> 5;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g, n = n0 + j * 32 + fr;
          C[(long)m * N + n] = seed + r;
        }
  } else if (MODE == 1) {
    const int c4 = lane & 15, rr = lane >> 4;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + r * 4 + rr;
      *reinterpret_cast<float4*>(C + (long)m * N + n0 + c4 * 4) = make_float4(seed, seed + r, seed, seed);
    }
  } else if (MODE == 2) {  // the two column-neighbour wavefronts together cover 128 columns: this one takes rows of its parity
    const int c4 = lane & 31, rr = lane >> 5;
    const int nb = (blockIdx.x % tiles_n) * 128, mb = (blockIdx.x / tiles_n) * 128 + (wave >> 1) * 64;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mb + (r * 2 + rr) * 2 + (wave & 1);
      *reinterpret_cast<float4*>(C + (long)m * N + nb + c4 * 4) = make_float4(seed, seed + r, seed, seed);
    }
  } else {
    const long base = ((long)blockIdx.x * 4 + wave) * 4096;
#pragma unroll
    for (int r = 0; r < 16; ++r) *reinterpret_cast<float4*>(C + base + r * 256 + lane * 4) = make_float4(seed, seed + r, seed, seed);
  }
}

int main() {
  const int M = 10880, N = 2048;
  float* C;
  hipMalloc(&C, (size_t)M * N * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int wgs = (M / 128) * (N / 128);
  for (int mode = 0; mode < 4; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) store_kernel<0><<<wgs, 256>>>(C, M, N, (float)rep);
      if (mode == 1) store_kernel<1><<<wgs, 256>>>(C, M, N, (float)rep);
      if (mode == 2) store_kernel<2><<<wgs, 256>>>(C, M, N, (float)rep);
      if (mode == 3) store_kernel<3><<<wgs, 256>>>(C, M, N, (float)rep);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    printf("mode %d: %.1f us for %.1f MB = %.2f TB/s\n", mode, best * 1e3, M * (double)N * 4 / 1e6, M * (double)N * 4 / best / 1e9);
  }
  return 0;
}
