// bf16x6 lab: an fp32-ACCURATE product on the bf16 matrix pipe.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/lab/bf16x6_lab.hip -o scripts/lab/bf16x6_lab && scripts/lab/bf16x6_lab
// x = h + m + l with h = rne_bf16(x), m = rne_bf16(x - h), l = rne_bf16(x - h - m) (both subtractions exact in fp32): 24
// significand bits in three bf16 planes, no range problem (bf16 keeps the fp32 exponent).  The product keeps the six plane
// pairs of order <= 2^-16: l*h + h*l + m*m + m*h + h*m + h*h, fp32 accumulate (v_mfma_f32_32x32x16_bf16): what is dropped
// (m*l, l*m, l*l) is <= 2^-23 of |a||b| per product — the rounding class of an fp32 FMA chain.  Six MFMAs of 32 cycles per
// 32x32x16 block against eight of 64 cycles on the fp32 pipe: 2500 / 6 = 417 TFLOP/s-equivalent peak against 157.
// TERMS = 3 is round 1's two-plane product (error ~5e-6) for comparison.
//
// Structure under test (PIPE = 1): BK = 16 per stage, LDS double-buffered, ONE barrier per k-tile; the global loads of
// tile t+2 are issued before the MFMAs of tile t, tile t+1 is split / converted / written to the other LDS buffer after
// them (a second resident workgroup's MFMAs cover it).  PIPE = 0: round 1's single-buffered two-barrier loop.
// Operands row-major ([rows][K]) or k-major ([K][rows]) as in rscotr_gemm_f32.  Interior shapes only (lab).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int BK = 16;

template <int NPL>
__device__ __forceinline__ void split(float x, __bf16 (&p)[3]) {
#ifdef ABL_NOCVT
  const unsigned short t = (unsigned short)(__builtin_bit_cast(unsigned, x) >> 16);
  p[0] = p[1] = p[2] = __builtin_bit_cast(__bf16, t);
  return;
#endif
  p[0] = (__bf16)x;
  const float r1 = x - (float)p[0];
  p[1] = (__bf16)r1;
  if (NPL == 3) p[2] = (__bf16)(r1 - (float)p[1]);
}

__device__ __forceinline__ unsigned pk(__bf16 a, __bf16 b) {
  return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}

// One operand tile of R rows x 16 k.  LDS image per stage:
//   row-major source: [R][LDR] bf16, LDR = 16 * NPL + 8: planes side by side (h | m | l | pad); a fragment (row, 8 k at 8g,
//     plane p) is one 16-byte read at row * LDR + 16 p + 8 g.  Row pitch 112 B (NPL 3) / 80 B (NPL 2): conflict-free b128 reads.
//   k-major source:   [NPL][8 k-pairs][R] dwords (two consecutive k of one row per dword), written as 16-byte rows,
//     a fragment = four dword reads (pairs 4g .. 4g+3).  Same k order inside a fragment for both layouts.
template <int R, bool KM, int NPL>
struct Operand {
  static constexpr int LDR = 16 * NPL + 8;
  static constexpr int WORDS = KM ? NPL * 8 * R : R * LDR / 2;  // dwords per stage
  static constexpr int NV = KM ? (8 * R / 4 + 255) / 256 : (R * 4 + 255) / 256;
  float4 v[NV], w[NV];  // row-major: v only; k-major: v = even k row, w = odd k row of a pair

  __device__ __forceinline__ void load(const float* __restrict__ P, int ld, int row0, int k0, int tid) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * 256;
      if (!KM) {
        if (R * 4 % 256 == 0 || idx < R * 4)
          v[i] = *reinterpret_cast<const float4*>(P + (long)(row0 + (idx >> 2)) * ld + k0 + (idx & 3) * 4);
      } else {
        if (8 * R / 4 % 256 == 0 || idx < 8 * R / 4) {
          const int kp = idx / (R / 4), r4 = (idx % (R / 4)) * 4;
          const float* src = P + (long)(k0 + 2 * kp) * ld + row0 + r4;
          v[i] = *reinterpret_cast<const float4*>(src);
          w[i] = *reinterpret_cast<const float4*>(src + ld);
        }
      }
    }
  }
  __device__ __forceinline__ void store(unsigned* S, int tid) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * 256;
      if (!KM) {
        if (R * 4 % 256 == 0 || idx < R * 4) {
          const int row = idx >> 2, kq = (idx & 3) * 4;
          __bf16 a[3], b[3], c[3], d[3];
          split<NPL>(v[i].x, a); split<NPL>(v[i].y, b); split<NPL>(v[i].z, c); split<NPL>(v[i].w, d);
          unsigned* dst = S + (row * LDR + kq) / 2;
#pragma unroll
          for (int p = 0; p < NPL; ++p) {
            uint2 q;
            q.x = pk(a[p], b[p]);
            q.y = pk(c[p], d[p]);
            *reinterpret_cast<uint2*>(dst + p * 8) = q;
          }
        }
      } else {
        if (8 * R / 4 % 256 == 0 || idx < 8 * R / 4) {
          const int kp = idx / (R / 4), r4 = (idx % (R / 4)) * 4;
          __bf16 e0[3], o0[3], e1[3], o1[3], e2[3], o2[3], e3[3], o3[3];
          split<NPL>(v[i].x, e0); split<NPL>(w[i].x, o0);
          split<NPL>(v[i].y, e1); split<NPL>(w[i].y, o1);
          split<NPL>(v[i].z, e2); split<NPL>(w[i].z, o2);
          split<NPL>(v[i].w, e3); split<NPL>(w[i].w, o3);
#pragma unroll
          for (int p = 0; p < NPL; ++p) {
            uint4 q;
            q.x = pk(e0[p], o0[p]); q.y = pk(e1[p], o1[p]); q.z = pk(e2[p], o2[p]); q.w = pk(e3[p], o3[p]);
            *reinterpret_cast<uint4*>(S + (p * 8 + kp) * R + r4) = q;
          }
        }
      }
    }
  }
  static __device__ __forceinline__ void frag(const unsigned* S, int row, int g, bf16x8 (&f)[3]) {
    if (!KM) {
      const unsigned* q = S + (row * LDR + 8 * g) / 2;
#pragma unroll
      for (int p = 0; p < NPL; ++p) f[p] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(q + p * 8));
    } else {
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
        const unsigned* q = S + (p * 8 + 4 * g) * R + row;
        uint4 t;
        t.x = q[0]; t.y = q[R]; t.z = q[2 * R]; t.w = q[3 * R];
        f[p] = __builtin_bit_cast(bf16x8, t);
      }
    }
  }
};

template <int BM, int BN, int TERMS, bool AKM, bool BKM, int PIPE>
__global__ __launch_bounds__(256) void gemm_split(const float* __restrict__ A, const float* __restrict__ B,
                                                  float* __restrict__ C, int M, int N, int K, int lda, int ldb) {
  constexpr int NPL = TERMS == 6 ? 3 : 2;
  constexpr int MT = BM / 64, NT = BN / 64;
  using OA = Operand<BM, AKM, NPL>;
  using OB = Operand<BN, BKM, NPL>;
  constexpr int NBUF = PIPE ? 2 : 1;
  __shared__ __attribute__((aligned(16))) unsigned sA[NBUF][OA::WORDS], sB[NBUF][OB::WORDS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = N / BN;
  // XCD-aware order: each XCD takes a contiguous run of tiles (nwg % 8 == 0 in the lab shapes or small remainder ignored)
  int tile = blockIdx.x;
  {
    const int n = gridDim.x, q = n >> 3, r = n & 7, x = tile & 7, j = tile >> 3;
    tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j;
  }
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  OA la;
  OB lb;
  const int fr = lane & 31, g = lane >> 5;
  auto mma = [&](const unsigned* a_s, const unsigned* b_s) {
#ifdef ABL_NOMMA
    return;
#endif
    bf16x8 af[MT][3], bf[NT][3];
#pragma unroll
    for (int i = 0; i < MT; ++i) OA::frag(a_s, wm * (BM / 2) + i * 32 + fr, g, af[i]);
#pragma unroll
    for (int j = 0; j < NT; ++j) OB::frag(b_s, wn * (BN / 2) + j * 32 + fr, g, bf[j]);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (TERMS == 6) {  // small terms first
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
        }
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
      }
  };
  const int nk = K / BK;
  if (PIPE) {
    la.load(A, lda, m0, 0, tid);
    lb.load(B, ldb, n0, 0, tid);
    la.store(sA[0], tid);
    lb.store(sB[0], tid);
    if (nk > 1) {
      la.load(A, lda, m0, BK, tid);
      lb.load(B, ldb, n0, BK, tid);
    }
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
      const int cur = t & 1;
      mma(sA[cur], sB[cur]);
      if (t + 1 < nk) {  // registers hold tile t+1: split / convert / write to the other buffer, then fetch tile t+2
        la.store(sA[cur ^ 1], tid);
        lb.store(sB[cur ^ 1], tid);
        if (t + 2 < nk) {
          la.load(A, lda, m0, (t + 2) * BK, tid);
          lb.load(B, ldb, n0, (t + 2) * BK, tid);
        }
      }
      __syncthreads();
    }
  } else {
    la.load(A, lda, m0, 0, tid);
    lb.load(B, ldb, n0, 0, tid);
    for (int t = 0; t < nk; ++t) {
#ifdef ABL_NOSTAGE
      if (t == 0) {
        la.store(sA[0], tid);
        lb.store(sB[0], tid);
        __syncthreads();
      }
#else
      __syncthreads();
      la.store(sA[0], tid);
      lb.store(sB[0], tid);
      __syncthreads();
#ifndef ABL_NOLOAD
      if (t + 1 < nk) {
        la.load(A, lda, m0, (t + 1) * BK, tid);
        lb.load(B, ldb, n0, (t + 1) * BK, tid);
      }
#endif
#endif
      mma(sA[0], sB[0]);
    }
  }
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (BM / 2) + i * 32 + 8 * (r >> 2) + 4 * g + (r & 3);
        const int col = n0 + wn * (BN / 2) + j * 32 + fr;
#ifdef ABL_NOEPI
        if (acc[i][j][r] == 123.456f)
#endif
        C[(long)row * N + col] = acc[i][j][r];
      }
}

// fp32 FMA chain on sampled entries (error yardstick)
__global__ void ref_f32(const float* A, const float* B, float* out, int K, int lda, int ldb, int akm, int bkm,
                        const int* samples, int ns) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= ns) return;
  const int m = samples[2 * s], n = samples[2 * s + 1];
  float acc = 0.f;
  for (int k = 0; k < K; ++k)
    acc = fmaf(akm ? A[(long)k * lda + m] : A[(long)m * lda + k], bkm ? B[(long)k * ldb + n] : B[(long)n * ldb + k], acc);
  out[s] = acc;
}

typedef void (*Kern)(const float*, const float*, float*, int, int, int, int, int);
struct Variant { const char* name; Kern k[4]; int bm, bn; };  // k[akm * 2 + bkm]

#define VAR(BM, BN, T, P) { #BM "x" #BN " t" #T " p" #P, { gemm_split<BM, BN, T, false, false, P>, gemm_split<BM, BN, T, false, true, P>, \
                                                         gemm_split<BM, BN, T, true, false, P>, gemm_split<BM, BN, T, true, true, P> }, BM, BN }

int main(int argc, char** argv) {
  const Variant vars[] = {VAR(128, 128, 6, 0), VAR(128, 128, 3, 0), VAR(64, 64, 6, 1), VAR(64, 64, 6, 0)};
  // {M, N, K, a_kmajor, b_kmajor, fp32-pipe us (profiles/history/r1_s7_gemm_census_fp32.txt)}
  const int shapes[][6] = {{10880, 2048, 256, 0, 0, 117}, {10880, 256, 256, 0, 0, 23}, {10880, 256, 2048, 0, 0, 125}, {4096, 4096, 4096, 0, 0, 0}};
  std::mt19937 rng(1);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (auto& sh : shapes) {
    const int M = sh[0], N = sh[1], K = sh[2], akm = sh[3], bkm = sh[4];
    const int lda = akm ? M : K, ldb = bkm ? N : K;
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    for (auto& v : hA) v = nd(rng);
    for (auto& v : hB) v = nd(rng) * 0.05f;
    float *dA, *dB, *dC, *dR;
    CK(hipMalloc(&dA, hA.size() * 4)); CK(hipMalloc(&dB, hB.size() * 4)); CK(hipMalloc(&dC, (size_t)M * N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    const int ns = 2048;
    std::vector<int> smp(2 * ns);
    for (int s = 0; s < ns; ++s) { smp[2 * s] = rng() % M; smp[2 * s + 1] = rng() % N; }
    int* dS;
    CK(hipMalloc(&dS, smp.size() * 4)); CK(hipMalloc(&dR, ns * 4));
    CK(hipMemcpy(dS, smp.data(), smp.size() * 4, hipMemcpyHostToDevice));
    std::vector<double> ref(ns);
    double refmax = 0;
    for (int s = 0; s < ns; ++s) {
      const int m = smp[2 * s], n = smp[2 * s + 1];
      double a = 0;
      for (int k = 0; k < K; ++k)
        a += (double)(akm ? hA[(size_t)k * lda + m] : hA[(size_t)m * lda + k]) * (double)(bkm ? hB[(size_t)k * ldb + n] : hB[(size_t)n * ldb + k]);
      ref[s] = a;
      refmax = std::max(refmax, std::fabs(a));
    }
    ref_f32<<<(ns + 255) / 256, 256>>>(dA, dB, dR, K, lda, ldb, akm, bkm, dS, ns);
    std::vector<float> hR(ns), hC((size_t)M * N);
    CK(hipMemcpy(hR.data(), dR, ns * 4, hipMemcpyDeviceToHost));
    double e32 = 0;
    for (int s = 0; s < ns; ++s) e32 = std::max(e32, std::fabs(hR[s] - ref[s]));
    const double flop = 2.0 * M * N * K;
    printf("M=%5d N=%5d K=%5d %d%d  fp32 pipe (r1 census): %3d us   fp32 FMA chain err %.1e\n", M, N, K, akm, bkm, sh[5], e32 / refmax);
    for (const Variant& v : vars) {
      if (M % v.bm || N % v.bn || K % BK) continue;
      const dim3 grid((M / v.bm) * (N / v.bn));
      Kern kern = v.k[akm * 2 + bkm];
      const int iters = flop > 5e10 ? 10 : 40;
      for (int i = 0; i < 3; ++i) kern<<<grid, 256>>>(dA, dB, dC, M, N, K, lda, ldb);
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      for (int i = 0; i < iters; ++i) kern<<<grid, 256>>>(dA, dB, dC, M, N, K, lda, ldb);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const float us = ms / iters * 1e3f;
      CK(hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost));
      double e = 0;
      for (int s = 0; s < ns; ++s) e = std::max(e, std::fabs(hC[(size_t)smp[2 * s] * N + smp[2 * s + 1]] - ref[s]));
      printf("    %-16s %5d wgs %8.1f us %7.1f TF-eq  err %.1e\n", v.name, grid.x, us, flop / us * 1e-6, e / refmax);
    }
    CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC)); CK(hipFree(dS)); CK(hipFree(dR));
  }
  return 0;
}
