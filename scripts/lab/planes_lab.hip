// Stand-alone lab for the "pre-split weights" form of the bf16x6 product (round 2):
//   C[M,N] = A[M,K] (fp32, row-major, split into three bf16 planes WHILE it is staged) x B^T, B = weights given as
//   PRE-SPLIT planes in a k-step-major layout [K/16][N][3 planes][16 k] bf16 (written once per optimizer step by a split
//   kernel), so that a 256-row stage of B is one contiguous 24 KB run that goes global -> VGPR -> LDS without any VALU work.
// Workgroup = 512 threads (8 wavefronts as 2 x 4), tile 128 x 256 (wave tile 64 x 64 = 2 x 2 MFMA tiles, 24 MFMAs per
// 16-k step), A converted once per 256 columns; two LDS stages, one barrier per step, DEPTH register sets of prefetch.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/lab/planes_lab.hip -o scripts/lab/planes_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void split3(float x, __bf16 (&p)[3]) {
  p[0] = (__bf16)x;
  const float r1 = x - (float)p[0];
  p[1] = (__bf16)r1;
  p[2] = (__bf16)(r1 - (float)p[1]);
}
__device__ __forceinline__ unsigned pk(__bf16 a, __bf16 b) {
  return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}

// W [N][K] fp32 -> planes [K/16][N][3][16] bf16 (one thread per (n, k-step): 16 floats)
__global__ void split_weights(const float* __restrict__ W, unsigned short* __restrict__ planes, int N, int K) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int kt = (int)(idx / N), n = (int)(idx % N);
  if (kt >= K / 16) return;
  const float4* src = reinterpret_cast<const float4*>(W + (long)n * K + kt * 16);
  unsigned out[3][8];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 v = src[q];
    __bf16 a[3], b[3], c[3], d[3];
    split3(v.x, a); split3(v.y, b); split3(v.z, c); split3(v.w, d);
#pragma unroll
    for (int p = 0; p < 3; ++p) { out[p][q * 2] = pk(a[p], b[p]); out[p][q * 2 + 1] = pk(c[p], d[p]); }
  }
  uint4* dst = reinterpret_cast<uint4*>(planes + ((long)kt * N + n) * 48);
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    dst[p * 2] = make_uint4(out[p][0], out[p][1], out[p][2], out[p][3]);
    dst[p * 2 + 1] = make_uint4(out[p][4], out[p][5], out[p][6], out[p][7]);
  }
}

constexpr int LDR = 56;  // bf16 per LDS row: 3 planes x 16 k + 8 pad (112 bytes)

template <int BM, int BN, int DEPTH, int SCHED>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_bplanes(const float* __restrict__ A, const unsigned short* __restrict__ Bp,
                                                    float* __restrict__ C, int M, int N, int K, int splits) {
  static_assert(BM == 128 && (BN == 256 || BN == 128), "tile");
  constexpr int WNW = BN / 64;            // waves along n: 4 (BN 256) or 2 (BN 128: wave tile 32 x 64... see MT)
  constexpr int WMW = 8 / WNW;            // waves along m
  constexpr int MT = BM / WMW / 32;       // MFMA row tiles per wave
  constexpr int NT = 2;
  constexpr int A_WORDS = BM * LDR / 2, B_WORDS = BN * LDR / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned lds[];
  unsigned* sA[2] = {lds, lds + A_WORDS};
  unsigned* sB[2] = {lds + 2 * A_WORDS, lds + 2 * A_WORDS + B_WORDS};
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WNW, wn = wave % WNW;
  const int tiles_n = N / BN, tiles = (M / BM) * tiles_n;
  int tile = blockIdx.x % tiles;
  const int split = blockIdx.x / tiles;
  {  // XCD-aware order
    const int n = tiles, q = n >> 3, r = n & 7, x = tile & 7, j = tile >> 3;
    tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
    if (tile >= tiles) tile = tiles - 1;
  }
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  const int nk_all = K / 16;
  const int kt0 = (int)((long)split * nk_all / splits), kt1 = (int)((long)(split + 1) * nk_all / splits);
  const int nk = kt1 - kt0;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging roles: A: thread -> (row = tid >> 2, 4 k at (tid & 3) * 4), one float4 (BM = 128: 512 items)
  //                B: thread -> (row = tid >> 1 (+ 256 per extra pass), 48 bytes at half = tid & 1): BN / 256 passes... BN = 256: 1
  struct Set {
    float4 a;
    uint4 b[3];
    __device__ __forceinline__ void load(const float* a_src, const unsigned short* b_src, long a_off, long b_off, bool b_active) {
      a = *reinterpret_cast<const float4*>(a_src + a_off);
      if (b_active) {
        const uint4* s = reinterpret_cast<const uint4*>(b_src + b_off);
        b[0] = s[0]; b[1] = s[1]; b[2] = s[2];
      }
    }
    __device__ __forceinline__ void store(unsigned* a_s, unsigned* b_s, int tid, bool b_active) const {
      __bf16 x[3], y[3], z[3], w[3];
      split3(a.x, x); split3(a.y, y); split3(a.z, z); split3(a.w, w);
      unsigned* dst = a_s + ((tid >> 2) * LDR + (tid & 3) * 4) / 2;
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<uint2*>(dst + p * 8) = make_uint2(pk(x[p], y[p]), pk(z[p], w[p]));
      if (b_active) {
        uint4* d4 = reinterpret_cast<uint4*>(b_s + ((tid >> 1) * LDR + (tid & 1) * 24) / 2);
        d4[0] = b[0]; d4[1] = b[1]; d4[2] = b[2];
      }
    }
  };
  Set sets[DEPTH];
  const float* a_src = A + (long)(m0 + (tid >> 2)) * K + (tid & 3) * 4;
  const bool b_active = BN == 256 || tid < 256;
  const unsigned short* b_src = Bp + ((long)n0 + (tid >> 1)) * 48 + (tid & 1) * 24;
  const int fr = lane & 31, g = lane >> 5;
  auto mma = [&](const unsigned* a_s, const unsigned* b_s) {
    bf16x8 af[MT][3], bf[NT][3];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const unsigned* q = a_s + ((wm * (BM / WMW) + i * 32 + fr) * LDR + 8 * g) / 2;
#pragma unroll
      for (int p = 0; p < 3; ++p) af[i][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(q + p * 8));
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const unsigned* q = b_s + ((wn * 64 + j * 32 + fr) * LDR + 8 * g) / 2;
#pragma unroll
      for (int p = 0; p < 3; ++p) bf[j][p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(q + p * 8));
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
      }
  };
  constexpr int U = (DEPTH % 2 == 0) ? DEPTH : 2 * DEPTH;
  constexpr int NMFMA = MT * NT * 6;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) {
    const int kt = kt0 + min(d, nk - 1);
    sets[d].load(a_src, b_src, (long)kt * 16, (long)kt * N * 48, b_active);
  }
  sets[0].store(sA[0], sB[0], tid, b_active);
  __syncthreads();
  for (int t0 = 0; t0 < nk; t0 += U) {
#pragma unroll
    for (int s = 0; s < U; ++s) {
      const int t = t0 + s;
      if (t < nk) {
        {
          const int kt = kt0 + min(t + DEPTH, nk - 1);
          sets[s % DEPTH].load(a_src, b_src, (long)kt * 16, (long)kt * N * 48, b_active);
        }
        mma(sA[s & 1], sB[s & 1]);
        sets[(s + 1) % DEPTH].store(sA[(s + 1) & 1], sB[(s + 1) & 1], tid, b_active);
        if (SCHED) {
#pragma unroll
          for (int i = 0; i < NMFMA; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
          }
        }
        __syncthreads();
      }
    }
  }
  float* Cs = C + (long)split * M * N;
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (BM / WMW) + i * 32 + 8 * (r >> 2) + 4 * g + (r & 3);
        const int col = n0 + wn * 64 + j * 32 + fr;
        Cs[(long)row * N + col] = acc[i][j][r];
      }
}

__global__ void slab_sum(const float4* __restrict__ slabs, float4* __restrict__ out, long n4, int splits) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 a = slabs[i];
    for (int s = 1; s < splits; ++s) {
      const float4 v = slabs[(long)s * n4 + i];
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    out[i] = a;
  }
}

typedef void (*Kern)(const float*, const unsigned short*, float*, int, int, int, int);
struct Variant { const char* name; Kern k; int bm, bn; };
static size_t lds_bytes(int bm, int bn) { return 2 * (size_t)(bm + bn) * LDR * 2; }
#define VAR(BM, BN, D, S) { #BM "x" #BN " d" #D " s" #S, gemm_bplanes<BM, BN, D, S>, BM, BN }

int main() {
  const Variant vars[] = {VAR(128, 256, 2, 0), VAR(128, 256, 3, 0), VAR(128, 256, 3, 1), VAR(128, 256, 4, 1), VAR(128, 128, 3, 1)};
  // {M, N, K, splits, current step time us}
  const int shapes[][5] = {{10880, 2048, 256, 1, 85}, {10880, 256, 2048, 1, 105}, {10880, 256, 2048, 3, 105}, {10880, 256, 2048, 2, 105},
                           {10880, 256, 256, 1, 21}, {32768, 384, 96, 1, 47}, {8192, 768, 192, 1, 35}, {2048, 1536, 384, 1, 30},
                           {4096, 4096, 4096, 1, 0}};
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (auto& sh : shapes) {
    const int M = sh[0], N = sh[1], K = sh[2], splits = sh[3];
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    for (auto& v : hA) v = nd(rng);
    for (auto& v : hB) v = nd(rng) * 0.05f;
    float *dA, *dB, *dC, *dO;
    unsigned short* dP;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4));
    CK(hipMalloc(&dC, (size_t)M * N * 4 * splits)); CK(hipMalloc(&dO, (size_t)M * N * 4));
    CK(hipMalloc(&dP, (size_t)N * K * 6));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    const long nsplit = (long)N * (K / 16);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < 20; ++i) split_weights<<<(unsigned)((nsplit + 255) / 256), 256>>>(dB, dP, N, K);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const int ns = 2048;
    std::vector<int> smp(2 * ns);
    for (int s = 0; s < ns; ++s) { smp[2 * s] = rng() % M; smp[2 * s + 1] = rng() % N; }
    std::vector<double> ref(ns);
    double refmax = 0;
    for (int s = 0; s < ns; ++s) {
      double a = 0;
      for (int k = 0; k < K; ++k) a += (double)hA[(size_t)smp[2 * s] * K + k] * (double)hB[(size_t)smp[2 * s + 1] * K + k];
      ref[s] = a;
      refmax = std::max(refmax, std::fabs(a));
    }
    const double flop = 2.0 * M * N * K;
    printf("M=%5d N=%5d K=%5d splits=%d  now %3d us   (weight split: %.1f us)\n", M, N, K, splits, sh[4], ms / 20 * 1e3);
    std::vector<float> hC((size_t)M * N);
    for (const Variant& v : vars) {
      if (M % v.bm || N % v.bn || K % 16) continue;
      const dim3 grid((M / v.bm) * (N / v.bn) * splits);
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(v.k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(v.bm, v.bn)));
      auto run = [&]() {
        v.k<<<grid, 512, lds_bytes(v.bm, v.bn)>>>(dA, dP, splits > 1 ? dC : dO, M, N, K, splits);
        if (splits > 1) slab_sum<<<1024, 256>>>(reinterpret_cast<const float4*>(dC), reinterpret_cast<float4*>(dO), (long)M * N / 4, splits);
      };
      for (int i = 0; i < 3; ++i) run();
      const int iters = flop > 5e10 ? 10 : 40;
      CK(hipEventRecord(e0));
      for (int i = 0; i < iters; ++i) run();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      const float us = ms / iters * 1e3f;
      CK(hipMemcpy(hC.data(), dO, hC.size() * 4, hipMemcpyDeviceToHost));
      double e = 0;
      for (int s = 0; s < ns; ++s) e = std::max(e, std::fabs(hC[(size_t)smp[2 * s] * N + smp[2 * s + 1]] - ref[s]));
      printf("    %-16s %5d wgs %8.1f us %7.1f TF-eq  err %.1e\n", v.name, grid.x, us, flop / us * 1e-6, e / refmax);
    }
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dO)); CK(hipFree(dP));
  }
  return 0;
}
