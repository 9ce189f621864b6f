// Two-layer FFN block of the shared encoder as ONE launch (round 6): y = act(x W1^T + b1) W2^T + b2 (+ resid) with the wide
// intermediate leaving the kernel only for the backward pass, never to be read again by the forward.
//
// Replaces (reference call sites): mmcv FFN inside the encoder's BaseTransformerLayer (cfg
// configs/multi/MTL_slvlcls_swin-t-p4-w7_1x1_resisc&dior&potsdam.py:44-49, built at models/multi/multitask_learner.py:51,
// called from models/multi/seg_head/pixel_decoder.py:134-146 and models/multi/bbox_head/transformer.py:211-221): two
// nn.Linear around a ReLU.  As two products (rscotr_gemm_f32_r twice) the 10880 x 2048 hidden tensor is written (89 MB), read
// back by the second product (89 MB), and each launch pays its own ramp / drain and converts its operands again: 122 us per
// layer forward, 118 us for the mirrored pair dH = (g W2) * gate, dX = dH W1 of backward (cold operands, scripts/lab/ffn_cold.py)
// — the largest block of the round.
//
// Structure.  A workgroup (512 threads = 8 wavefronts, one per CU) owns BM = 32 or 48 rows (the host picks what balances the
// 256 CUs: 10880 rows = 227 workgroups of 48).  The rows' fp16 planes (h | l of the "h3" split product, gemm.hip) are staged in
// LDS ONCE.  Then the wavefronts split into two ROLES that work on different chunks of 128 hidden columns at the same time:
//   wavefronts 0-3 (A)  hidden chunk c (BM x 128) = x W1[chunk]^T, 32 columns each; bias, ReLU (gate bits out) or gate (bits in);
//                       the fp32 chunk to global memory (the weight gradients of backward read it), its planes into LDS image c % 2
//                       — this epilogue one chunk late, spread over the k loop of chunk c + 1;
//   wavefronts 4-7 (B)  y (BM x 256) += chunk (c - 2) W2[:, chunk]^T from its LDS image, 64 output columns each;
// one barrier per chunk.  Every SIMD holds one wavefront of each role: while an A wavefront converts and stores its chunk
// (VALU, LDS and memory instructions), its B partner has the matrix pipe to itself — the first version of this kernel ran both
// phases on all eight wavefronts in lockstep and left the pipe idle for a quarter of its life (profiles/r6_ffn_lab.txt).
// Both products are computed TRANSPOSED (the weight fragment is the MFMA's A operand, the activation fragment its B operand): a
// lane's four accumulator registers are then four consecutive COLUMNS of one row — 16-byte stores, 8-byte LDS writes, float4
// bias / residual loads instead of four scalar ones each.
// The weights arrive as FRAGMENT-MAJOR fp16 planes (rscotr_gemm_split_weights_frag: [row tile of 16][k step of 32][h | l][lane][8
// halfs], written once per optimizer step): the weight operand of a 16 x 16 x 32 MFMA is ONE contiguous 1 KB load per wavefront,
// no wavefront shares a fragment with another, so nothing goes through LDS and every weight byte is read once per workgroup
// (4 MB for 256 -> 2048 -> 256; L2-resident).  A ring of eight such loads per wavefront runs ahead of the MFMAs across chunks.
//
// Numerics: the same three-term fp16 split product as gemm_h3_* (l h, h l into a second accumulator that enters with 2^-11; fp32
// accumulate) — on v_mfma_f32_16x16x32_f16 instead of 32x32x16, so sums associate differently: equal to rscotr_gemm_f32_r's
// result at fp32 rounding, not bit for bit.  The planes of the hidden chunk are scaled by a power of two taken from an A-PRIORI
// bound, C * max|x| * max|W1| + max|b1| (the true maximum is only known after the last workgroup): any upper bound is a valid
// range (ops/ranges.py); a bound 2^t too loose moves the point where the planes start to lose RELATIVE precision from 2^-26 to
// 2^(t-26) of the maximum — below it the error is 2^-48 2^t of the maximum absolute, invisible in an fp32 product.
#include "gemm_common.h"
#include "rscotr.h"
#include <type_traits>

namespace rscotr {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int FFN_HC = 128;    // hidden columns per chunk (4 A wavefronts x 32)
constexpr int FFN_LDRB = 160;  // bytes per LDS row of a 32-k stage: h[32] | l[32] | 32 bytes pad (conflict-free ds_read_b128 of the
                               // 16 x 16 x 32 fragment pattern: row = lane & 15, 16 bytes at k = 8 (lane >> 4))
#ifndef FFN_DB
#define FFN_DB 1  // k steps (of 32) the B role's weight fragment loads run ahead of their MFMAs
#endif

// MODE: what happens to a hidden chunk between the two products
enum { FFN_RELU = 0,       // hidden = relu(x W1^T + b1); one bit per element [hidden > 0] -> bits
       FFN_RELU_GATE = 1,  // hidden = (x W1^T) * bit                       (dH = (g W2) * [h > 0] of the backward pass)
       FFN_GELU = 2,       // pre = x W1^T + b1 -> Pre; hidden = gelu(pre)  (the MLP of a Swin block: mmdet SwinBlock's FFN, erf GELU)
       FFN_GELU_GRAD = 3 };// hidden = (x W1^T) * gelu'(Pre)               (dH = (g W2) * gelu'(pre))

struct FfnParams {
  const float* X;
  int M, H;
  const uint4* W1f;
  const float* b1;
  const uint4* W2f;
  const float* b2;
  unsigned* bits;      // FFN_RELU: written, FFN_RELU_GATE: read
  float* Pre;          // FFN_GELU: written, FFN_GELU_GRAD: read (M, H)
  float* Hid;
  const float* resid;
  float* Y;
  const float* xscale; // per-sample factor on the rows of X while they are staged (the upstream gradient of a DropPath'ed block) | null
  const float* yscale; // per-sample factor on Y before the residual (DropPath folded into the block's last Linear) | null
  int rows_per;        // rows per sample for the two
  const unsigned *amax_x, *amax_w1, *amax_w2, *amax_b1;
  unsigned *amax_hid, *amax_y;
  int splits;          // > 1: run s takes the hidden chunks [s, s + 1) * (H / 128 / splits) and stores its PARTIAL y (no bias /
  float* slabs;        //      scale / residual / range word) into slabs[s] (M x C); gemm_splitk_reduce_kernel adds them up and finishes
  // LN instantiations (a pre-norm block, x + MLP(LayerNorm(x)): mmdet SwinBlock's norm2 + FFN): X is the block input, the rows are
  // normalised while they are staged; LayerNorm(X) leaves the kernel too (the weight gradient of the first Linear and the norm's own
  // backward read it, with the row statistics)
  const float *ln_g, *ln_b;
  float ln_eps;
  float *ln_out, *ln_mean, *ln_rstd;
  const unsigned *amax_g, *amax_bt;
  unsigned* amax_ln;
  int by_xcd;          // splits > 1 and splits | 8: 1-D grid of 8 * tpx workgroups, workgroup b (XCD b % 8) = run (b % 8) % splits,
  int tpx;             //      row tile (b % 8) / splits + (8 / splits) * (b / 8): an XCD's L2 then holds ONE run's slice of the weights
};

// C: model width (reduction of the first product, columns of the second); NT: 16-row tiles per workgroup
template <int C, int NT>
constexpr size_t ffn_lds_bytes() { return (size_t)(C / 32 + 2 * (FFN_HC / 32)) * (16 * NT) * FFN_LDRB; }

// 16-byte buffer store with a SCALAR offset register.  LLVM's hazard recognizer assumes that a buffer store of more than 8 bytes
// whose soffset is an SGPR has read its data registers by the time the next instruction issues, and inserts no wait state; on
// gfx950 that does not hold: a v_pk_mul_f32 right behind the store, allocated onto the same registers, reached memory instead of
// the stored values in lanes 12-15 of every 16 (found by scripts/lab/ffn_debug.py: exactly the elements x 2^(s+11), the split's
// second product).  One wait state pinned behind the store, as the ISA asks for the immediate-offset form.
__device__ __forceinline__ void store_b128(float4 v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned)))) unsigned, v), r, voff, soff, 0);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 1");
  __builtin_amdgcn_sched_barrier(0);
}

#ifndef FFN_STAG_MINC
#define FFN_STAG_MINC 192  // narrowest width whose launches rotate their chunk order: 8192 x 192 -> 768 cold 45.6 / 41.5 -> 38.2 / 37.6 us with it; C = 96 (held to
                           // 128 registers, where the rotation's index arithmetic spills) 57 -> 61: stays in order
#endif
#ifndef FFN_PHASES
#define FFN_PHASES 4  // distinct starting chunks among the workgroups that share an L2.  16 (every chunk of 256 -> 2048 a starting point) spreads the
                      // cold requests widest but puts all 4 MB of both weights' planes into the working set of a 4 MB L2 next to the 89 MB hidden
                      // stream: 2 x 155 MiB fetched per launch against 2 x 27 without rotation (scripts/lab/pmc_ffn_traffic.sh); 4: 2 x 77 MiB, the
                      // same launch time cold and in the step (29.66-29.72 ms per round; 8: 29.64-29.65; 16: 29.72-29.79; 2: 29.96-29.99)
#endif
// first chunk of workgroup j (of those sharing the weights through one L2) in a run of n chunks
__device__ __forceinline__ int ffn_phase(int j, int n) {
  const int ph = n < FFN_PHASES ? n : FFN_PHASES;
  return (int)((unsigned)j % (unsigned)ph) * n / ph;
}

__device__ __forceinline__ f32x4_t mfma16(uint4 a, uint4 b, f32x4_t c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// Bit layout of FFN_RELU / FFN_RELU_GATE (opaque to callers, the same in both directions): uint32 [row tile][chunk][A wavefront][lane],
// bit (it * NT + n) * 4 + r = accumulator register r of the lane's 16 x 16 tile (column tile it, row tile n).
template <int C, int NT, int MODE, bool LN = false>
#ifndef FFN_SWIN_WAVES
#define FFN_SWIN_WAVES 4  // wavefronts per SIMD the C = 96 kernels are held to (4 = 128 registers: two 8-wavefront workgroups per CU; stage 1 of
                          // Swin-T at 512^2 is 1024 workgroups; C = 192 is 256 workgroups = one per CU, C = 256 needs its 123 KB of LDS alone)
#endif
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(C == 96 ? FFN_SWIN_WAVES : 2, C == 96 ? FFN_SWIN_WAVES : 2))) void ffn_h3_kernel(FfnParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ffn_lds[];
  constexpr bool GATE = MODE == FFN_RELU_GATE, GELU = MODE == FFN_GELU, GGRAD = MODE == FFN_GELU_GRAD;
  constexpr bool FWD = MODE == FFN_RELU || MODE == FFN_GELU;
  constexpr int BM = 16 * NT, STG = BM * FFN_LDRB;  // bytes per 32-k stage of a plane image
  constexpr int KS1 = C / 32, HST = FFN_HC / 32;    // k steps of the first product; stages (= k steps) of a chunk image
  constexpr int DA = KS1 % 2 == 0 ? 2 : 1;          // k steps the A role's weight loads run ahead (ring position static: DA | KS1)
  constexpr int TPB = C == 96 ? 2 : C / 64;         // 16-column output tiles per B wavefront
  constexpr int NB = C / 16 / TPB;                  // B wavefronts with work (3 for C = 96, else 4)
  constexpr int RING = 4 * DA > 2 * TPB * FFN_DB ? 4 * DA : 2 * TPB * FFN_DB;
  constexpr int SPK = (2 * NT + KS1 - 1) / KS1;     // epilogue tiles riding in one k step of the next chunk
  static_assert(C % 32 == 0 && (C / 16) % TPB == 0 && NB <= 4, "column tiles split evenly over the B wavefronts");
  unsigned char* xs = ffn_lds;                      // [KS1][BM][160]
  unsigned char* hs = ffn_lds + KS1 * STG;          // [2][HST][BM][160]
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kg = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // (provably uniform: the role branches below must be scalar branches)
  const bool role_a = wv < 4;
  const int wr = wv & 3;
  // row tile bx, run by of the hidden width, rank jx of this workgroup among the ones that share its weights through one L2
  int bx = blockIdx.x, by = blockIdx.y, jx = blockIdx.x >> 3;
  if (p.by_xcd) {
    const int xcd = (int)(blockIdx.x & 7);
    jx = (int)(blockIdx.x >> 3);
    by = xcd % p.splits;
    bx = xcd / p.splits + (8 / p.splits) * jx;
    if (bx * BM >= p.M) return;  // (the grid is rounded up to whole XCD rounds; before any barrier)
  }
  const int m0 = bx * BM;
  const int nch = p.H / FFN_HC;                   // chunks of the hidden width
  const int nloc = nch / p.splits;                // ... of this workgroup: chunks cbase .. cbase + nloc - 1
  const int cbase = by * nloc;

  // value-range words: requested first, reduced after the operand loads have been requested too (cold lines)
  const long sub = (long)(lane & (kAmaxPlanes - 1)) * kAmaxStride;
  const unsigned rx = LN ? 0u : p.amax_x[sub], r1 = p.amax_w1[sub], r2 = p.amax_w2[sub], rb = p.amax_b1 ? p.amax_b1[sub] : 0u;
  const unsigned rg = LN ? p.amax_g[sub] : 0u, rbt = (LN && p.amax_bt) ? p.amax_bt[sub] : 0u;

  // weight fragments: buffer loads — ONE address register per wavefront (its lane and its tiles), the chunk / k step as a scalar
  // offset.  A: tile = c * 8 + wr * 2 + it of W1op (H rows, KS1 k steps);  B: tile = wr * TPB + it of W2op (C rows, H / 32 k steps)
  const int wbytes = p.H * C * 4;
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(role_a ? p.W1f : p.W2f), 0, wbytes, 0x00020000);
  const int nks2 = p.H / 32;
  const int vW = lane * 16 + (role_a ? wr * 2 * KS1 * 2048 : wr * TPB * nks2 * 2048);
  auto ld_a = [&](int c, int ks, int it, int pl) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rW, vW, c * (8 * KS1 * 2048) + it * (KS1 * 2048) + ks * 2048 + pl * 1024, 0));
  };
  auto ld_b = [&](int t, int it, int pl) {  // t: global k-step index c * 4 + ks = the k step of W2op
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rW, vW, it * nks2 * 2048 + t * 2048 + pl * 1024, 0));
  };
#ifdef FFN_NO_STAGGER
  const int c0r = cbase;
#else
  const int c0r = cbase + (C >= FFN_STAG_MINC ? ffn_phase(jx, nloc) : 0);  // (= chunk_of(0) below: the first chunk of this workgroup)
#endif
  uint4 ring[RING];
  if (role_a) {
#pragma unroll
    for (int q = 0; q < 4 * DA; ++q) ring[q] = ld_a(c0r, q >> 2, (q >> 1) & 1, q & 1);  // (DA <= KS1: the first chunk)
  } else if (wr < NB) {
#pragma unroll
    for (int q = 0; q < 2 * TPB * FFN_DB; ++q) ring[q] = ld_b(c0r * HST + min(q / (2 * TPB), HST - 1), (q >> 1) % TPB, q & 1);  // (FFN_DB <= HST)
  }

  // the row tile's tensors through descriptors that end at row M: rows past the end read zeros and drop their stores — no
  // per-element guards (which cost a branch and a live 64-bit address per store of the chunk epilogue)
  const int rows_ok = min(BM, p.M - m0);
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X + (long)m0 * C), 0, rows_ok * C * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc(p.Hid + (long)m0 * p.H, 0, rows_ok * p.H * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((GELU || GGRAD) ? p.Pre + (long)m0 * p.H : p.Hid, 0, rows_ok * p.H * 4, 0x00020000);
  float* const ybase = p.splits > 1 ? p.slabs + (long)by * p.M * C : p.Y;
  const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(ybase + (long)m0 * C, 0, rows_ok * C * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.resid ? p.resid + (long)m0 * C : p.X), 0, rows_ok * C * 4, 0x00020000);

  // the rows' planes -> LDS (all eight wavefronts).  LN: |LayerNorm(x)| <= sqrt(C) max|gamma| + max|beta| — a normalised row has
  // no entry above sqrt(C - 1) — is the range the planes are scaled with (the row maxima are only known afterwards; a loose bound
  // costs nothing above 2^-26 of it, see the header)
  // (words deliver binades, common.h: amax_fold = 2^e <= max, amax_hi its upper end; a scale taken from a word uses e, a bound the upper end)
  const unsigned fx = LN ? 0u : amax_fold(rx);
  const float xbound = LN ? sqrtf((float)C) * amax_hi(amax_fold(rg)) + amax_hi(amax_fold(rbt)) : amax_hi(fx);  // max |staged row entry| <= xbound
  const unsigned ux = LN ? __float_as_uint(xbound) : fx;
  if constexpr (!LN) {
    constexpr int TOT = BM * (C / 4), NV = (TOT + 511) / 512;
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rX, (tid + i * 512) * 16, 0, 0));
    const int ex0 = h3_scale_exp(ux);
    const H3Scale hx{__uint_as_float((unsigned)ex0 << 23), __uint_as_float((unsigned)(ex0 + 11) << 23)};
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * 512;
      if (TOT % 512 == 0 || idx < TOT) {
        const int row = idx / (C / 4), kq = (idx % (C / 4)) * 4;
        if (p.xscale) {  // (the range word bounds the unscaled rows: a DropPath factor 1 / keep rides in the 8 x headroom of the scale)
          const float f = p.xscale[min(m0 + row, p.M - 1) / p.rows_per];
          v[i].x *= f; v[i].y *= f; v[i].z *= f; v[i].w *= f;
        }
        unsigned ab[3], cd[3];
        split_pair_h(v[i].x, v[i].y, hx, ab);
        split_pair_h(v[i].z, v[i].w, hx, cd);
        unsigned char* dst = xs + (kq / 32) * STG + row * FFN_LDRB + (kq % 32) * 2;
        *reinterpret_cast<uint2*>(dst) = make_uint2(ab[0], cd[0]);
        *reinterpret_cast<uint2*>(dst + 64) = make_uint2(ab[1], cd[1]);
      }
    }
  } else {
    // G = C / 12 lanes per row, three float4 each (columns 4 (sub + j G)): 512 / G rows per pass; mean, then the centred sum of
    // squares, both in registers (the arithmetic of layernorm_fwd_kernel on another lane layout: equal at rounding, not bit for bit)
    static_assert(C % 48 == 0 && (C / 12 == 8 || C / 12 == 16 || C / 12 == 32), "LN rows: 8 / 16 / 32 lanes x 3 float4");
    constexpr int G = C / 12, RPP = 512 / G, NP = (BM + RPP - 1) / RPP;
    const int sub = tid % G, rr = tid / G;
    const __amdgpu_buffer_rsrc_t rL = __builtin_amdgcn_make_buffer_rsrc(p.ln_out + (long)m0 * C, 0, (by == 0 ? rows_ok : 0) * C * 4, 0x00020000);
    const int ex0 = h3_scale_exp(ux);
    const H3Scale hx{__uint_as_float((unsigned)ex0 << 23), __uint_as_float((unsigned)(ex0 + 11) << 23)};
    float amx = 0.f;
    if (RPP <= BM || wv < 8 * BM / RPP) {  // (C = 96: 64 rows per pass, 32 to stage — wavefronts 4-7 have none)
      float4 gw[3], gb[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        gw[j] = reinterpret_cast<const float4*>(p.ln_g)[sub + j * G];
        gb[j] = p.ln_b ? reinterpret_cast<const float4*>(p.ln_b)[sub + j * G] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int ps = 0; ps < NP; ++ps) {
        const int row = ps * RPP + rr;
        float4 v[3];
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          v[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rX, (row * C + 4 * (sub + j * G)) * 4, 0, 0));
          sm += v[j].x + v[j].y + v[j].z + v[j].w;
        }
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
        const float mu = sm * (1.f / (float)C);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          v[j].x -= mu; v[j].y -= mu; v[j].z -= mu; v[j].w -= mu;
          q += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
        }
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        const float rs = rsqrtf(q * (1.f / (float)C) + p.ln_eps);
        if (sub == 0 && by == 0 && row < rows_ok) { p.ln_mean[m0 + row] = mu; p.ln_rstd[m0 + row] = rs; }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          float4 o;
          o.x = v[j].x * rs * gw[j].x + gb[j].x; o.y = v[j].y * rs * gw[j].y + gb[j].y;
          o.z = v[j].z * rs * gw[j].z + gb[j].z; o.w = v[j].w * rs * gw[j].w + gb[j].w;
          if (row >= rows_ok) o = make_float4(0.f, 0.f, 0.f, 0.f);  // (rows past M: beta would leak into the planes — harmless, their outputs are dropped, but not into the range word)
          const int kq = 4 * (sub + j * G);
          store_b128(o, rL, (row * C + kq) * 4, 0);
          amx = amax4(amx, o);
          unsigned ab[3], cd[3];
          split_pair_h(o.x, o.y, hx, ab);
          split_pair_h(o.z, o.w, hx, cd);
          unsigned char* dst = xs + (kq / 32) * STG + row * FFN_LDRB + (kq % 32) * 2;
          *reinterpret_cast<uint2*>(dst) = make_uint2(ab[0], cd[0]);
          *reinterpret_cast<uint2*>(dst + 64) = make_uint2(ab[1], cd[1]);
        }
      }
    }
    if (by == 0) amax_commit(p.amax_ln, amx);
  }
  const unsigned u1 = amax_fold(r1), u2 = amax_fold(r2), ub = amax_fold(rb);
  const int ex = h3_scale_exp(ux), e1 = h3_scale_exp(u1), e2 = h3_scale_exp(u2);
  // |hidden| <= C max|x| max|W1| + max|b1|: relu / a gate only shrink it, |gelu(t)| <= |t|, |gelu'| < 1.13 (hence the 1.25); a
  // DropPath factor on the rows of x (<= 2 for keep >= 0.5) rides in the scale's 8 x headroom
  const float bound = ((float)C * xbound * amax_hi(u1) + amax_hi(ub)) * (GGRAD ? 1.25f : 1.f);
  const int eh = h3_scale_exp(__float_as_uint(bound));
  const H3Scale hh{__uint_as_float((unsigned)eh << 23), __uint_as_float((unsigned)(eh + 11) << 23)};
  const float invx = __uint_as_float((unsigned)(254 - ex) << 23), inv1 = __uint_as_float((unsigned)(254 - e1) << 23);
  const float invh = __uint_as_float((unsigned)(254 - eh) << 23), inv2 = __uint_as_float((unsigned)(254 - e2) << 23);

  const int fo = li * FFN_LDRB + kg * 16;  // the lane's part of a fragment address: row li of a 16-row tile, 16 bytes at k = 8 kg
  // Chunk ORDER is rotated per workgroup: the ~28 workgroups of an XCD otherwise all want the same weight fragments at the same time —
  // cold for that L2 in the step, where twelve layers' plane sets pass through it per iteration — and then all hit the same lines.
  // Loop index c -> chunk (c + c0) % nch; the image parity stays the loop index's.  (The order of the output's partial sums
  // becomes a function of the row tile: still fixed by the launch geometry, bit-reproducible.)
#ifdef FFN_NO_STAGGER
  const int c0 = 0;
#else
  const int c0 = C >= FFN_STAG_MINC ? ffn_phase(jx, nloc) : 0;  // (C = 96 stays in order: FFN_STAG_MINC)
#endif
  auto chunk_of = [&](int c) {
    if constexpr (C < FFN_STAG_MINC) return cbase + c;
    const int v = c + c0;
    return cbase + (v >= nloc ? v - nloc : v);
  };
  __syncthreads();

  // The two roles are separate code paths (a scalar branch: wv is wave-uniform by construction), each with its own loop and its
  // own nch + 2 barriers, so that neither role's accumulators are live in the other's code.
  if (role_a) {
    // The epilogue of chunk c - 1 (bias, activation / gate, planes -> LDS image, fp32 chunk -> memory) RIDES IN THE K LOOP OF CHUNK c, SPK
    // 16 x 16 tiles per k step: the matrix pipe takes an MFMA every 16 cycles and leaves three issue slots in between, so the VALU / LDS /
    // memory instructions of a tile disappear between the MFMAs of a step instead of standing alone at the end of the chunk,
    // where the B partner — done with its own chunk — waited at the barrier (+20 us of 88: profiles/r6_ffn_lab.txt).  The k loop is
    // fully unrolled (tile and ring indices static) with a scheduling barrier per k step (unfenced, the scheduler hoists the
    // fragment reads of all steps to the top and spills).  The image of chunk c - 1 is complete at the barrier that ends
    // chunk c's loop: the B role runs two barriers behind.
    unsigned amxu = 0u;
    const int vH = (li * p.H + wr * 32 + 4 * kg) * 4;
    float4 pv[2][NT];  // chunk c - 1 before bias / activation
    float4 bvs[2];
    float4 pres[GGRAD ? 2 : 1][GGRAD ? NT : 1];
    unsigned bits = 0u;
    auto slice = [&](int cl, int it, int n) {  // cl: loop index of the chunk (image parity), cp: the chunk itself
      const int cp = chunk_of(cl);
      float v[4] = {pv[it][n].x, pv[it][n].y, pv[it][n].z, pv[it][n].w};
      const int soff = (n * 16 * p.H + cp * FFN_HC + it * 16) * 4;
      if constexpr (MODE == FFN_RELU) {
        // relu as compare + select (fmaxf canonicalises its operands: a v_cmp_class + v_cndmask per element on top of the v_max);
        // the compare is the gate bit
        const float t[4] = {v[0] + bvs[it].x, v[1] + bvs[it].y, v[2] + bvs[it].z, v[3] + bvs[it].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool pos = t[r] > 0.f;
          v[r] = pos ? t[r] : 0.f;
          bits |= (unsigned)pos << ((it * NT + n) * 4 + r);
        }
      } else if constexpr (GATE) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          v[r] = __uint_as_float(__float_as_uint(v[r]) & (unsigned)__builtin_amdgcn_sbfe((int)bits, (it * NT + n) * 4 + r, 1));
      } else if constexpr (GELU) {
        const float4 t = make_float4(v[0] + bvs[it].x, v[1] + bvs[it].y, v[2] + bvs[it].z, v[3] + bvs[it].w);
        store_b128(t, rP, vH, soff);
        v[0] = gelu_f(t.x); v[1] = gelu_f(t.y); v[2] = gelu_f(t.z); v[3] = gelu_f(t.w);
      } else {
        const float4 t = pres[GGRAD ? it : 0][GGRAD ? n : 0];
        v[0] *= gelu_grad_f(t.x); v[1] *= gelu_grad_f(t.y); v[2] *= gelu_grad_f(t.z); v[3] *= gelu_grad_f(t.w);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) amxu = max(amxu, __float_as_uint(v[r]) & 0x7fffffffu);  // (bit patterns of |v| order like the values)
      unsigned ab[3], cd[3];
      split_pair_h(v[0], v[1], hh, ab);
      split_pair_h(v[2], v[3], hh, cd);
#ifndef FFN_ABL_NOIMG
      unsigned char* dst = hs + (cl & 1) * (HST * STG) + wr * STG + li * FFN_LDRB + kg * 8 + n * 16 * FFN_LDRB + it * 32;
      *reinterpret_cast<uint2*>(dst) = make_uint2(ab[0], cd[0]);
      *reinterpret_cast<uint2*>(dst + 64) = make_uint2(ab[1], cd[1]);
#else
      asm volatile("" ::"v"(ab[0]), "v"(ab[1]), "v"(cd[0]), "v"(cd[1]));
#endif
#ifndef FFN_ABL_NOHID  // (FFN_ABL_*: timing ablations of scripts/lab/ffn_abl.sh — results are wrong with any of them)
      store_b128(make_float4(v[0], v[1], v[2], v[3]), rH, vH, soff);
#endif
    };
    auto pre_epi = [&](int cl) {  // what the tiles of chunk cl need from memory: requested a k loop ahead of their first use
      const int cp = chunk_of(cl);
#pragma unroll
      for (int it = 0; it < 2; ++it)
        bvs[it] = (FWD && p.b1) ? *reinterpret_cast<const float4*>(p.b1 + cp * FFN_HC + wr * 32 + it * 16 + 4 * kg) : make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (GATE) bits = p.bits[(((long)bx * nch + cp) * 4 + wr) * 64 + lane];
      else bits = 0u;
      if constexpr (GGRAD) {
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            pres[it][n] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rP, vH, (n * 16 * p.H + cp * FFN_HC + it * 16) * 4, 0));
      }
    };
    auto kloop = [&](int c, auto with_epi) {
      constexpr bool EPI = decltype(with_epi)::value;
      f32x4_t ha[2][NT], hb[2][NT];
#pragma unroll
      for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int n = 0; n < NT; ++n) { ha[it][n] = f32x4_t{0.f, 0.f, 0.f, 0.f}; hb[it][n] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
      const int cw = chunk_of(c), cn = chunk_of(min(c + 1, nloc - 1));  // (past the last chunk: a harmless re-read)
#pragma unroll
      for (int ks = 0; ks < KS1; ++ks) {
        uint4 xh[NT], xl[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          const unsigned char* q = xs + ks * STG + n * 16 * FFN_LDRB + fo;
          xh[n] = *reinterpret_cast<const uint4*>(q);
          xl[n] = *reinterpret_cast<const uint4*>(q + 64);
        }
        // terms of gemm_h3_*: l h, h l into the second accumulator, h h into the first (weight = the MFMA's A operand).  A ring
        // slot is refilled AFTER its last use, into the same registers (refilling it first made the compiler rotate the ring
        // through copies, and every copy waits for the load it copies: the pipeline drained once per k step)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int slot = (ks % DA) * 4 + it * 2;
          const uint4 wh = ring[slot], wl = ring[slot + 1];
#pragma unroll
          for (int n = 0; n < NT; ++n) hb[it][n] = mfma16(wh, xl[n], hb[it][n]);
#pragma unroll
          for (int n = 0; n < NT; ++n) hb[it][n] = mfma16(wl, xh[n], hb[it][n]);
#pragma unroll
          for (int n = 0; n < NT; ++n) ha[it][n] = mfma16(wh, xh[n], ha[it][n]);
#ifndef FFN_ABL_NOB
          if (ks + DA < KS1) { ring[slot] = ld_a(cw, ks + DA, it, 0); ring[slot + 1] = ld_a(cw, ks + DA, it, 1); }
          else { ring[slot] = ld_a(cn, ks + DA - KS1, it, 0); ring[slot + 1] = ld_a(cn, ks + DA - KS1, it, 1); }
#endif
        }
        if constexpr (EPI) {
#pragma unroll
          for (int t = ks * SPK; t < (ks + 1) * SPK; ++t)
            if (t < 2 * NT) slice(c - 1, t / NT, t % NT);  // (ends in store_b128's scheduling barrier)
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (EPI && MODE == FFN_RELU) p.bits[(((long)bx * nch + chunk_of(c - 1)) * 4 + wr) * 64 + lane] = bits;
      pre_epi(c);
#pragma unroll
      for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          pv[it][n].x = fmaf(hb[it][n][0], 0x1p-11f, ha[it][n][0]) * invx * inv1;
          pv[it][n].y = fmaf(hb[it][n][1], 0x1p-11f, ha[it][n][1]) * invx * inv1;
          pv[it][n].z = fmaf(hb[it][n][2], 0x1p-11f, ha[it][n][2]) * invx * inv1;
          pv[it][n].w = fmaf(hb[it][n][3], 0x1p-11f, ha[it][n][3]) * invx * inv1;
        }
    };
#ifdef FFN_ABL_NOPIPE  // (timing ablation: every chunk's epilogue right behind its own k loop; the B role then reads an image too early)
#pragma unroll 1
    for (int c = 0; c < nloc; ++c) {
      kloop(c, std::false_type{});
#pragma unroll
      for (int t = 0; t < 2 * NT; ++t) slice(c, t / NT, t % NT);
      if constexpr (MODE == FFN_RELU) p.bits[(((long)bx * nch + chunk_of(c)) * 4 + wr) * 64 + lane] = bits;
      __syncthreads();
    }
    __syncthreads();
    __syncthreads();
#else
    kloop(0, std::false_type{});
    __syncthreads();
#pragma unroll 1
    for (int c = 1; c < nloc; ++c) {
      kloop(c, std::true_type{});
      __syncthreads();  // image (c - 1) % 2 is complete
    }
#pragma unroll
    for (int t = 0; t < 2 * NT; ++t) slice(nloc - 1, t / NT, t % NT);
    if constexpr (MODE == FFN_RELU) p.bits[(((long)bx * nch + chunk_of(nloc - 1)) * 4 + wr) * 64 + lane] = bits;
    __syncthreads();  // the last image is complete
    __syncthreads();  // (the B role's last barrier)
#endif
    amax_commit(p.amax_hid, __uint_as_float(amxu));
  } else {
    f32x4_t ya[TPB][NT], yb[TPB][NT];  // y tiles: columns (wr * TPB + it) * 16 .., rows n * 16 ..
#pragma unroll
    for (int it = 0; it < TPB; ++it)
#pragma unroll
      for (int n = 0; n < NT; ++n) { ya[it][n] = f32x4_t{0.f, 0.f, 0.f, 0.f}; yb[it][n] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    const bool active = wr < NB;
    __syncthreads();
    __syncthreads();  // (image c is complete one barrier later: its epilogue rides in the k loop of chunk c + 1)
#pragma unroll 1
    for (int c = 0; c < nloc; ++c) {
      // ---- y += chunk c W2[:, chunk]^T: 4 k steps of 32
      const unsigned char* img = hs + (c & 1) * (HST * STG) + fo;
      const int cb = chunk_of(c);
#ifndef FFN_ABL_NOPHASEB
      if (active) {
#pragma unroll 1
        for (int kq = 0; kq < HST / FFN_DB; ++kq)
#pragma unroll
          for (int u = 0; u < FFN_DB; ++u) {
            const int ks = kq * FFN_DB + u;
            uint4 gh[NT], gl[NT];
#pragma unroll
            for (int n = 0; n < NT; ++n) {
              const unsigned char* q = img + ks * STG + n * 16 * FFN_LDRB;
              gh[n] = *reinterpret_cast<const uint4*>(q);
              gl[n] = *reinterpret_cast<const uint4*>(q + 64);
            }
            // (k step ks + FFN_DB of this chunk, or the first ones of the next chunk in this workgroup's order)
            int tn;
            if constexpr (C < FFN_STAG_MINC) tn = cbase * HST + min(c * HST + ks + FFN_DB, nloc * HST - 1);
            else tn = ks + FFN_DB < HST ? cb * HST + ks + FFN_DB : chunk_of(min(c + 1, nloc - 1)) * HST + (ks + FFN_DB - HST);
#pragma unroll
            for (int it = 0; it < TPB; ++it) {
              const int slot = u * 2 * TPB + it * 2;
              const uint4 wh = ring[slot], wl = ring[slot + 1];
#pragma unroll
              for (int n = 0; n < NT; ++n) yb[it][n] = mfma16(wh, gl[n], yb[it][n]);
#pragma unroll
              for (int n = 0; n < NT; ++n) yb[it][n] = mfma16(wl, gh[n], yb[it][n]);
#pragma unroll
              for (int n = 0; n < NT; ++n) ya[it][n] = mfma16(wh, gh[n], ya[it][n]);
#ifndef FFN_ABL_NOB
              ring[slot] = ld_b(tn, it, 0);  // (after the slot's last use: see the A role)
              ring[slot + 1] = ld_b(tn, it, 1);
#endif
            }
          }
      }
#endif
      __syncthreads();  // image c % 2 has been consumed
    }
    // ---- y = ((h h + (l h + h l) 2^-11) 2^-(s_hidden + s_w2) + b2) * yscale (+ resid): four consecutive columns of a row per lane
    if (active) {
      const int vY = (li * C + wr * TPB * 16 + 4 * kg) * 4;
      float amy = 0.f;
#pragma unroll
      for (int it = 0; it < TPB; ++it) {
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.b2) bv = *reinterpret_cast<const float4*>(p.b2 + wr * TPB * 16 + it * 16 + 4 * kg);
        float4 e[NT];
        float ysc[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          e[n] = p.resid ? __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rR, vY, (n * 16 * C + it * 16) * 4, 0)) : make_float4(0.f, 0.f, 0.f, 0.f);
          ysc[n] = p.yscale ? p.yscale[min(m0 + n * 16 + li, p.M - 1) / p.rows_per] : 1.f;
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
          float4 v;
          v.x = fmaf(fmaf(yb[it][n][0], 0x1p-11f, ya[it][n][0]) * invh * inv2 + bv.x, ysc[n], e[n].x);
          v.y = fmaf(fmaf(yb[it][n][1], 0x1p-11f, ya[it][n][1]) * invh * inv2 + bv.y, ysc[n], e[n].y);
          v.z = fmaf(fmaf(yb[it][n][2], 0x1p-11f, ya[it][n][2]) * invh * inv2 + bv.z, ysc[n], e[n].z);
          v.w = fmaf(fmaf(yb[it][n][3], 0x1p-11f, ya[it][n][3]) * invh * inv2 + bv.w, ysc[n], e[n].w);
          store_b128(v, rY, vY, (n * 16 * C + it * 16) * 4);
          amy = amax4(amy, v);
        }
      }
      amax_commit(p.amax_y, amy);
    }
  }
}

// ONE Linear on the same machinery for the TALL, NARROW products of Swin stages 1 and 2 (mmdet WindowMSA's qkv / proj Linears and their
// input gradients: 32768 x 96 -> 288, 32768 x 96 -> 96, 32768 x 288 -> 96; 8192 x 192 -> 576 ...; cfg configs/multi/...potsdam.py:9-25):
// y = (x W^T + b) * yscale (+ resid).  These move 25-50 MB for 0.6-1.8 GFLOP — memory-bound shapes on which the tiled 64 x 64
// kernels spend 17-36 us (each of the 5-9 column tiles of a row tile stages and converts the rows again, six k steps of fixed cost
// per workgroup) against 8-11 us of HBM time.  Here a workgroup (512 threads) stages the planes of 32 rows ONCE (optionally
// normalising them on the way: the LayerNorm in front of qkv) and its eight wavefronts take 32 output columns each, 256 per sweep,
// weight fragments straight from the fragment-major planes; no barrier after the staging one.  (The same form for the 256-wide
// 10880-row Linears of the encoder lost to the tiled kernel in the step — profiles/r6_ffn_lab.txt item 4: those are not memory-bound.)
struct LinParams {
  const float* X;
  int M, N;            // rows; output columns (N % 32 == 0)
  const uint4* Wf;     // fragment-major planes of the weight operand (N rows, reduction C)
  const float* bias;
  const float* resid;
  float* Y;
  const float* xscale; // per-sample factor on the rows of X while they are staged | null
  const float* yscale; // per-sample factor on Y before the residual | null
  int rows_per;
  const unsigned *amax_x, *amax_w;
  unsigned* amax_y;
  int col_groups;      // grid.y: 1 = sweeps inside the workgroup
  const float *ln_g, *ln_b;  // LN instantiations: see FfnParams
  float ln_eps;
  float *ln_out, *ln_mean, *ln_rstd;
  const unsigned *amax_g, *amax_bt;
  unsigned* amax_ln;
};

template <int C, bool LN, int NT = 2>
__global__ __launch_bounds__(512) void lin_h3_kernel(LinParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ffn_lds[];
  constexpr int BM = 16 * NT, STG = BM * FFN_LDRB, KS1 = C / 32;
  constexpr int DA = KS1 % 2 == 0 ? 2 : KS1 % 3 == 0 ? 3 : 1;  // k steps the weight loads run ahead (DA | KS1: the ring position of a k step is the same in every sweep)
  static_assert(C % 32 == 0 && KS1 % DA == 0, "reduction in whole 32-k steps");
  unsigned char* xs = ffn_lds;  // [KS1][BM][160]
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 15, kg = lane >> 4;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * BM;
  const long sub = (long)(lane & (kAmaxPlanes - 1)) * kAmaxStride;
  const unsigned rx = LN ? 0u : p.amax_x[sub], rw = p.amax_w[sub];
  const unsigned rg = LN ? p.amax_g[sub] : 0u, rbt = (LN && p.amax_bt) ? p.amax_bt[sub] : 0u;
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(p.Wf), 0, p.N * C * 4, 0x00020000);
  auto ld_w = [&](int col0, int ks, int it, int pl) {  // fragment of column tile col0 / 16 + it, k step ks, plane pl
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rW, lane * 16, ((col0 >> 4) + it) * (KS1 * 2048) + ks * 2048 + pl * 1024, 0));
  };
  // (the first sweep's fragments are requested before the rows: cold lines).  gridDim.y > 1: FEW rows (Swin stages 3 / 4: 2048 / 512) — a
  // workgroup per (row tile, group of 256 columns) instead of sweeps, every group staging the rows again (from L2)
  const int c_first = (int)blockIdx.y * 256 + wv * 32;
  const int c_step = 256 * (int)gridDim.y;
  uint4 ring[4 * DA];
  if (c_first < p.N) {
#pragma unroll
    for (int q = 0; q < 4 * DA; ++q) ring[q] = ld_w(c_first, q >> 2, (q >> 1) & 1, q & 1);
  }
  const int rows_ok = min(BM, p.M - m0);
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X + (long)m0 * C), 0, rows_ok * C * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(p.Y + (long)m0 * p.N, 0, rows_ok * p.N * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.resid ? p.resid + (long)m0 * p.N : p.X), 0, rows_ok * p.N * 4, 0x00020000);
  // (words deliver binades, common.h: amax_fold = 2^e <= max, amax_hi its upper end; a scale taken from a word uses e, a bound the upper end)
  const unsigned fx = LN ? 0u : amax_fold(rx);
  const float xbound = LN ? sqrtf((float)C) * amax_hi(amax_fold(rg)) + amax_hi(amax_fold(rbt)) : amax_hi(fx);  // max |staged row entry| <= xbound
  const unsigned ux = LN ? __float_as_uint(xbound) : fx;
  const int ex = h3_scale_exp(ux);
  const H3Scale hx{__uint_as_float((unsigned)ex << 23), __uint_as_float((unsigned)(ex + 11) << 23)};
  if constexpr (!LN) {
    constexpr int TOT = BM * (C / 4), NV = (TOT + 511) / 512;
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rX, (tid + i * 512) * 16, 0, 0));
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * 512;
      if (TOT % 512 == 0 || idx < TOT) {
        const int row = idx / (C / 4), kq = (idx % (C / 4)) * 4;
        if (p.xscale) {
          const float f = p.xscale[min(m0 + row, p.M - 1) / p.rows_per];
          v[i].x *= f; v[i].y *= f; v[i].z *= f; v[i].w *= f;
        }
        unsigned ab[3], cd[3];
        split_pair_h(v[i].x, v[i].y, hx, ab);
        split_pair_h(v[i].z, v[i].w, hx, cd);
        unsigned char* dst = xs + (kq / 32) * STG + row * FFN_LDRB + (kq % 32) * 2;
        *reinterpret_cast<uint2*>(dst) = make_uint2(ab[0], cd[0]);
        *reinterpret_cast<uint2*>(dst + 64) = make_uint2(ab[1], cd[1]);
      }
    }
  } else {  // (ffn_h3_kernel's LN staging: C / 12 lanes per row, three float4 each)
    static_assert(!LN || (C % 48 == 0 && (C / 12 == 8 || C / 12 == 16 || C / 12 == 32)), "LN rows: 8 / 16 / 32 lanes x 3 float4");
    constexpr int G = C / 12, RPP = 512 / G, NP = (BM + RPP - 1) / RPP;
    const int sb = tid % G, rr = tid / G;
    const bool own = blockIdx.y == 0;  // (one column group writes the norm's output and statistics)
    const __amdgpu_buffer_rsrc_t rL = __builtin_amdgcn_make_buffer_rsrc(p.ln_out + (long)m0 * C, 0, (own ? rows_ok : 0) * C * 4, 0x00020000);
    float amx = 0.f;
    if (RPP <= BM || wv < 8 * BM / RPP) {
      float4 gw[3], gb[3];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        gw[j] = reinterpret_cast<const float4*>(p.ln_g)[sb + j * G];
        gb[j] = p.ln_b ? reinterpret_cast<const float4*>(p.ln_b)[sb + j * G] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int ps = 0; ps < NP; ++ps) {
        const int row = ps * RPP + rr;
        float4 v[3];
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          v[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rX, (row * C + 4 * (sb + j * G)) * 4, 0, 0));
          sm += v[j].x + v[j].y + v[j].z + v[j].w;
        }
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
        const float mu = sm * (1.f / (float)C);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          v[j].x -= mu; v[j].y -= mu; v[j].z -= mu; v[j].w -= mu;
          q += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
        }
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
        const float rs = rsqrtf(q * (1.f / (float)C) + p.ln_eps);
        if (sb == 0 && own && row < rows_ok) { p.ln_mean[m0 + row] = mu; p.ln_rstd[m0 + row] = rs; }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          float4 o;
          o.x = v[j].x * rs * gw[j].x + gb[j].x; o.y = v[j].y * rs * gw[j].y + gb[j].y;
          o.z = v[j].z * rs * gw[j].z + gb[j].z; o.w = v[j].w * rs * gw[j].w + gb[j].w;
          if (row >= rows_ok) o = make_float4(0.f, 0.f, 0.f, 0.f);
          const int kq = 4 * (sb + j * G);
          store_b128(o, rL, (row * C + kq) * 4, 0);
          amx = amax4(amx, o);
          unsigned ab[3], cd[3];
          split_pair_h(o.x, o.y, hx, ab);
          split_pair_h(o.z, o.w, hx, cd);
          unsigned char* dst = xs + (kq / 32) * STG + row * FFN_LDRB + (kq % 32) * 2;
          *reinterpret_cast<uint2*>(dst) = make_uint2(ab[0], cd[0]);
          *reinterpret_cast<uint2*>(dst + 64) = make_uint2(ab[1], cd[1]);
        }
      }
    }
    if (own) amax_commit(p.amax_ln, amx);
  }
  const int ew = h3_scale_exp(amax_fold(rw));
  const float inv = __uint_as_float((unsigned)(254 - ex) << 23) * __uint_as_float((unsigned)(254 - ew) << 23);
  const int fo = li * FFN_LDRB + kg * 16;
  __syncthreads();

  float amy = 0.f;
#pragma unroll 1
  for (int col0 = c_first; col0 < p.N; col0 += c_step) {  // (wave-uniform)
    f32x4_t ha[2][NT], hb[2][NT];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int n = 0; n < NT; ++n) { ha[it][n] = f32x4_t{0.f, 0.f, 0.f, 0.f}; hb[it][n] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    const int cnext = col0 + c_step < p.N ? col0 + c_step : col0;  // (past the last sweep: a harmless re-read)
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
      uint4 xh[NT], xl[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const unsigned char* q = xs + ks * STG + n * 16 * FFN_LDRB + fo;
        xh[n] = *reinterpret_cast<const uint4*>(q);
        xl[n] = *reinterpret_cast<const uint4*>(q + 64);
      }
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int slot = (ks % DA) * 4 + it * 2;
        const uint4 wh = ring[slot], wl = ring[slot + 1];
#pragma unroll
        for (int n = 0; n < NT; ++n) hb[it][n] = mfma16(wh, xl[n], hb[it][n]);
#pragma unroll
        for (int n = 0; n < NT; ++n) hb[it][n] = mfma16(wl, xh[n], hb[it][n]);
#pragma unroll
        for (int n = 0; n < NT; ++n) ha[it][n] = mfma16(wh, xh[n], ha[it][n]);
        if (ks + DA < KS1) { ring[slot] = ld_w(col0, ks + DA, it, 0); ring[slot + 1] = ld_w(col0, ks + DA, it, 1); }
        else { ring[slot] = ld_w(cnext, ks + DA - KS1, it, 0); ring[slot + 1] = ld_w(cnext, ks + DA - KS1, it, 1); }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // y = ((h h + (l h + h l) 2^-11) 2^-(s_x + s_w) + b) * yscale (+ resid): four consecutive columns of a row per lane
    const int vY = (li * p.N + col0 + 4 * kg) * 4;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias) bv = *reinterpret_cast<const float4*>(p.bias + col0 + it * 16 + 4 * kg);
      float4 e[NT];
      float ysc[NT];
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        e[n] = p.resid ? __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rR, vY, (n * 16 * p.N + it * 16) * 4, 0)) : make_float4(0.f, 0.f, 0.f, 0.f);
        ysc[n] = p.yscale ? p.yscale[min(m0 + n * 16 + li, p.M - 1) / p.rows_per] : 1.f;
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        float4 v;
        v.x = fmaf(fmaf(hb[it][n][0], 0x1p-11f, ha[it][n][0]) * inv + bv.x, ysc[n], e[n].x);
        v.y = fmaf(fmaf(hb[it][n][1], 0x1p-11f, ha[it][n][1]) * inv + bv.y, ysc[n], e[n].y);
        v.z = fmaf(fmaf(hb[it][n][2], 0x1p-11f, ha[it][n][2]) * inv + bv.z, ysc[n], e[n].z);
        v.w = fmaf(fmaf(hb[it][n][3], 0x1p-11f, ha[it][n][3]) * inv + bv.w, ysc[n], e[n].w);
        store_b128(v, rY, vY, (n * 16 * p.N + it * 16) * 4);
        amy = amax4(amy, v);
      }
    }
  }
  amax_commit(p.amax_y, amy);
}

// (Measured and removed, profiles/r6_ffn_lab.txt: the same machinery for the ONE-product 256-wide Linears of the encoder — value /
//  output projections, offsets | logits, their input gradients; rows' planes staged once, weight fragments from fragment-major
//  planes, 16-byte stores — 12.9 us per 10880 x 256 x 256 launch against 14.1 for the tiled 64 x 64 kernel in the cold lab, but
//  31.22 against 31.06 ms per round in the step: with one product per launch there is nothing to overlap the staging phase with.)

// (Measured and removed, second attempt, profiles/r6_ffn_lab.txt item 8: a rows-resident ONE-product kernel for the small row counts
//  — Swin stage 3 / 4 at 2048 / 512 rows, the decoders' 1600- / 200-row Linears; a workgroup = 32 rows x a column slice, all of K staged
//  once, no barrier in the k loop.  Kernel durations 11.1 us against 15.3 (2048 x 384 x 384), 11.1 against 20.7 (512 x 768 x 768), 8.0
//  against 9.5-10.1 (1600 x 256 x 256), equal at N >= 1536, 36-40 against 19.7 at K = 2048; in the step +1.35 ms per round.)

// Fragment-major fp16 planes of a weight for ffn_h3_kernel (rscotr_gemm_split_weights_frag): table rows {W, planes, rows of W,
// cols of W, ldw, 0, first block, transposed, range word of the parameter} (int64 x 9, as rscotr_gemm_split_weights_h3).
// transposed = 0: plane rows n = rows of W, reduction k over its columns (y = x W^T); 1: plane rows = columns of W, reduction over
// its rows (dx = dy W).  Layout: uint4 [n / 16][k / 32][h | l][lane], lane l holding k = 32 ks + 8 (l >> 4) .. + 7 of row
// n = 16 nt + (l & 15) — one operand of v_mfma_f32_16x16x32_f16, one contiguous 1 KB load per wavefront.  Plane rows % 16 == 0,
// reduction % 32 == 0 (host-checked); one thread per (row tile, k step, lane): 8 values in, two 16-byte records out.
__global__ __launch_bounds__(256) void split_weights_frag_kernel(const int64_t* __restrict__ table, int n_entries) {
  int e = 0;
  while (e + 1 < n_entries && (long)table[(long)(e + 1) * 9 + 6] <= (long)blockIdx.x) ++e;
  const int64_t* t = table + (long)e * 9;
  const float* W = reinterpret_cast<const float*>(t[0]);
  uint4* planes = reinterpret_cast<uint4*>(t[1]);
  const int wrows = (int)t[2], wcols = (int)t[3], ldw = (int)t[4], tr = (int)t[7];
  const int rows = tr ? wcols : wrows, red = tr ? wrows : wcols;
  const int se = h3_scale_exp(amax_read(reinterpret_cast<const unsigned*>(t[8])));
  const H3Scale hs{__uint_as_float((unsigned)se << 23), __uint_as_float((unsigned)(se + 11) << 23)};
  const long idx = ((long)blockIdx.x - t[6]) * 256 + threadIdx.x;
  const int nks = red / 32;
  const int lane = (int)(idx & 63);
  const long rest = idx >> 6;
  const int ks = (int)(rest % nks), nt = (int)(rest / nks);
  if (nt >= rows / 16) return;
  const int n = nt * 16 + (lane & 15), k0 = ks * 32 + 8 * (lane >> 4);
  float v[8];
  if (!tr) {
    const float4* src = reinterpret_cast<const float4*>(W + (long)n * ldw + k0);
    const float4 a = src[0], b = src[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = W[(long)(k0 + j) * ldw + n];
  }
  unsigned h[4], l[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned o[3];
    split_pair_h(v[2 * q], v[2 * q + 1], hs, o);
    h[q] = o[0]; l[q] = o[1];
  }
  uint4* dst = planes + ((long)nt * nks + ks) * 128 + lane;
  dst[0] = make_uint4(h[0], h[1], h[2], h[3]);
  dst[64] = make_uint4(l[0], l[1], l[2], l[3]);
}

}  // namespace rscotr

using namespace rscotr;

extern "C" int rscotr_gemm_split_weights_frag(const int64_t* table, int n, int total_blocks, void* stream) {
  if (!table || n <= 0 || total_blocks <= 0) return fail(RSCOTR_E_ARG, "split_weights_frag: empty table");
  hipLaunchKernelGGL(split_weights_frag_kernel, dim3((unsigned)total_blocks), dim3(256), 0, static_cast<hipStream_t>(stream), table, n);
  return check_launch("split_weights_frag");
}

// (C = 384 — Swin-T stage 3, 2048 rows — was instantiated and measured: 71 / 67 us against 45 / 46 for the two products; 64 workgroups each
// stream 4.7 MB of weight planes: profiles/r6_ffn_lab.txt.  C = 128 is Swin-B's stage 1.)
// rows per workgroup: C = 256 (the encoder FFN, one 123 KB workgroup per CU): what leaves the fewest rounds x rows on 256 CUs, 48 on
// ties (fewer weight reads); the Swin widths: 32 (two workgroups per CU: 56 / 72 KB of LDS)
static int ffn_rows(int M, int C) {
  if (C != 256) return 32;
  const long c48 = ((long)((M + 47) / 48) + 255) / 256 * 48, c32 = ((long)((M + 31) / 32) + 255) / 256 * 32;
  return c32 < c48 ? 32 : 48;
}

// FEW ROWS (Swin-T stage 3: 2048 x 384 -> 1536 -> 384, the detection decoder's FFN: 1600 x 256 -> 2048 -> 256): 64 / 50 row tiles leave
// three quarters of the CUs idle and each workgroup streams all of both weights (the C = 384 instantiation alone: 71 us against 45 for the
// two tiled products).  The hidden width is then CUT into `splits` runs of chunks, one workgroup per (row tile, run): every run makes its
// own partial y from its own slice of both weights, stored into a slab; gemm_splitk_reduce_kernel adds the slabs in fixed order and
// applies bias / scale / residual / range word.  splits = the smallest divisor of the chunk count that brings the launch to 200
// workgroups, runs of at least two chunks (the roles' pipeline); 1 for every launch that fills the chip by its rows.
extern "C" int rscotr_ffn_h3_splits(int M, int C, int H) {
  static const int forced = getenv("RSCOTR_FFN_SPLITS") ? atoi(getenv("RSCOTR_FFN_SPLITS")) : 0;  // (A/B runs: 1 = never)
  if (H < FFN_HC || H % FFN_HC) return 1;
  const int bm = ffn_rows(M, C), tiles = (M + bm - 1) / bm, nch = H / FFN_HC;
  if (forced > 0) return nch % forced == 0 ? forced : 1;
  if (tiles >= 200) return 1;
  for (int d = 2; d <= nch / 2; ++d)
    if (nch % d == 0 && tiles * d >= 200) return d;
  return 1;
}

extern "C" int rscotr_ffn_h3_ok(int M, int C, int H) {
  return ((C == 384 || C == 256 || C == 192 || C == 128 || C == 96) && H >= FFN_HC && H % FFN_HC == 0 && M >= 1) ? 1 : 0;
}

extern "C" int64_t rscotr_ffn_h3_bits_words(int M, int C, int H) {
  const int bm = ffn_rows(M, C);
  return (int64_t)((M + bm - 1) / bm) * (H / FFN_HC) * 256;
}

template <int C, int NT, int MODE, bool LN = false>
static void ffn_launch1(const FfnParams& p, hipStream_t s) {
  constexpr size_t lds = ffn_lds_bytes<C, NT>();
  static bool attr_set = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_h3_kernel<C, NT, MODE, LN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return true;
  }();
  (void)attr_set;
  const unsigned tiles = (unsigned)((p.M + 16 * NT - 1) / (16 * NT));
  const dim3 grid = p.by_xcd ? dim3(8u * (unsigned)p.tpx) : dim3(tiles, (unsigned)p.splits);
  hipLaunchKernelGGL((ffn_h3_kernel<C, NT, MODE, LN>), grid, dim3(512), lds, s, p);
}

template <int C, int NT>
static void ffn_launch(const FfnParams& p, int mode, hipStream_t s) {
  switch (mode) {
    case FFN_RELU: ffn_launch1<C, NT, FFN_RELU>(p, s); break;
    case FFN_RELU_GATE: ffn_launch1<C, NT, FFN_RELU_GATE>(p, s); break;
    case FFN_GELU: ffn_launch1<C, NT, FFN_GELU>(p, s); break;
    default: ffn_launch1<C, NT, FFN_GELU_GRAD>(p, s); break;
  }
}

struct FfnNorm {  // the LayerNorm in front of the block (rscotr_ffn_h3_ln) | all null
  const float *g, *b;
  float eps;
  float *out, *mean, *rstd;
  const uint32_t *amax_g, *amax_b;
  uint32_t* amax_out;
};

static int ffn_h3_run(const float* X, int M, int C, int H, const void* W1f, const float* b1, const void* W2f, const float* b2,
                      int mode, void* bits, float* Pre, float* Hid, const float* resid, float* Y, const float* xscale,
                      const float* yscale, int rows_per, const uint32_t* amax_x, const uint32_t* amax_w1,
                      const uint32_t* amax_w2, const uint32_t* amax_b1, uint32_t* amax_hid, uint32_t* amax_y, float* workspace,
                      int64_t workspace_bytes, const FfnNorm& ln, void* stream) {
  if (!rscotr_ffn_h3_ok(M, C, H)) return fail(RSCOTR_E_SHAPE, "ffn_h3: M=%d C=%d H=%d (C in {96, 128, 192, 256, 384}, H %% 128 == 0)", M, C, H);
  const int splits = rscotr_ffn_h3_splits(M, C, H);
  if (splits > 1 && (!workspace || workspace_bytes < (int64_t)splits * M * C * 4 || ((uintptr_t)workspace & 15)))
    return fail(RSCOTR_E_ARG, "ffn_h3: M=%d C=%d H=%d runs as %d partial sums: workspace of %lld bytes (16-byte aligned), got %lld", M, C, H,
                splits, (long long)splits * M * C * 4, (long long)workspace_bytes);
  if (mode < 0 || mode > 3) return fail(RSCOTR_E_ARG, "ffn_h3: mode %d", mode);
  if (!X || !W1f || !W2f || !Hid || !Y || (!amax_x && !ln.g) || !amax_w1 || !amax_w2) return fail(RSCOTR_E_ARG, "ffn_h3: null argument");
  if (ln.g) {
    if (mode != FFN_GELU || !(C == 96 || C == 192 || C == 384))
      return fail(RSCOTR_E_SHAPE, "ffn_h3_ln: the norm prologue exists for mode 2 (GELU forward) at C in {96, 192, 384} (mode %d, C=%d)", mode, C);
    if (!ln.out || !ln.mean || !ln.rstd || !ln.amax_g || xscale) return fail(RSCOTR_E_ARG, "ffn_h3_ln: null argument (or a row scale on X)");
    if (((uintptr_t)ln.g | (uintptr_t)ln.b | (uintptr_t)ln.out) & 15) return fail(RSCOTR_E_ALIGN, "ffn_h3_ln: operands must be 16-byte aligned");
  }
  if (mode <= FFN_RELU_GATE ? !bits : !Pre) return fail(RSCOTR_E_ARG, "ffn_h3: mode %d needs %s", mode, mode <= FFN_RELU_GATE ? "bits" : "Pre");
  if ((xscale || yscale) && rows_per <= 0) return fail(RSCOTR_E_ARG, "ffn_h3: rows_per with a row scale");
  if (((uintptr_t)X | (uintptr_t)W1f | (uintptr_t)W2f | (uintptr_t)Hid | (uintptr_t)Y | (uintptr_t)resid | (uintptr_t)b1 | (uintptr_t)b2 |
       (uintptr_t)Pre) & 15)
    return fail(RSCOTR_E_ALIGN, "ffn_h3: operands must be 16-byte aligned");
  if ((long)M * H * 4 >= (1l << 32)) return fail(RSCOTR_E_SHAPE, "ffn_h3: M * H too large for 32-bit buffer offsets");
  const bool fwd = mode == FFN_RELU || mode == FFN_GELU;
  FfnParams p{};
  p.X = X; p.M = M; p.H = H;
  p.W1f = static_cast<const uint4*>(W1f); p.b1 = fwd ? b1 : nullptr;
  p.W2f = static_cast<const uint4*>(W2f); p.b2 = b2;
  p.bits = static_cast<unsigned*>(bits); p.Pre = Pre; p.Hid = Hid; p.resid = resid; p.Y = Y;
  p.xscale = xscale; p.yscale = yscale; p.rows_per = rows_per > 0 ? rows_per : 1;
  p.amax_x = amax_x; p.amax_w1 = amax_w1; p.amax_w2 = amax_w2; p.amax_b1 = fwd ? amax_b1 : nullptr;
  p.amax_hid = amax_hid; p.amax_y = amax_y;
  p.splits = splits; p.slabs = workspace;
  p.ln_g = ln.g; p.ln_b = ln.b; p.ln_eps = ln.eps; p.ln_out = ln.out; p.ln_mean = ln.mean; p.ln_rstd = ln.rstd;
  p.amax_g = ln.amax_g; p.amax_bt = ln.amax_b; p.amax_ln = ln.amax_out;
  static const int xcd_ok = getenv("RSCOTR_FFN_SPLIT_XCD") ? atoi(getenv("RSCOTR_FFN_SPLIT_XCD")) : 1;  // (A/B runs)
  if (splits > 1 && 8 % splits == 0 && xcd_ok) {
    const int bm0 = ffn_rows(M, C), tiles = (M + bm0 - 1) / bm0, per = 8 / splits;  // XCDs per run
    p.by_xcd = 1; p.tpx = (tiles + per - 1) / per;
  }
  if (splits > 1) { p.b2 = nullptr; p.resid = nullptr; p.yscale = nullptr; p.amax_y = nullptr; }  // (the combine's)
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int bm = ffn_rows(M, C);
  {
    ProfScope prof(PROF_GEMM, 4.0 * M * (double)C * H, s, "rscotr::ffn_h3_kernel<%d, %d, %d>", C, bm / 16, mode);
    if (ln.g) {
      if (C == 384) ffn_launch1<384, 2, FFN_GELU, true>(p, s);
      else if (C == 192) ffn_launch1<192, 2, FFN_GELU, true>(p, s);
      else ffn_launch1<96, 2, FFN_GELU, true>(p, s);
    } else if (C == 384) ffn_launch<384, 2>(p, mode, s);
    else if (C == 256) { if (bm == 32) ffn_launch<256, 2>(p, mode, s); else ffn_launch<256, 3>(p, mode, s); }
    else if (C == 192) ffn_launch<192, 2>(p, mode, s);
    else if (C == 128) ffn_launch<128, 2>(p, mode, s);
    else ffn_launch<96, 2>(p, mode, s);
  }
  if (int rc = check_launch("ffn_h3")) return rc;
  if (splits > 1) {  // y = (sum of the slabs + b2) * yscale + resid, its range word
    GemmParams q{};
    q.C = Y; q.M = M; q.N = C; q.ldc = C; q.vecC = 1;
    q.bias = b2; q.resid = resid; q.rowscale = yscale; q.rows_per = p.rows_per;
    q.slabs = workspace; q.splits = splits; q.amax_out = amax_y;
    splitk_reduce_launch(q, workspace, s);
    return check_launch("ffn_h3 combine");
  }
  return 0;
}


extern "C" int rscotr_ffn_h3(const float* X, int M, int C, int H, const void* W1f, const float* b1, const void* W2f, const float* b2,
                             int mode, void* bits, float* Pre, float* Hid, const float* resid, float* Y, const float* xscale,
                             const float* yscale, int rows_per, const uint32_t* amax_x, const uint32_t* amax_w1,
                             const uint32_t* amax_w2, const uint32_t* amax_b1, uint32_t* amax_hid, uint32_t* amax_y, float* workspace,
                             int64_t workspace_bytes, void* stream) {
  return ffn_h3_run(X, M, C, H, W1f, b1, W2f, b2, mode, bits, Pre, Hid, resid, Y, xscale, yscale, rows_per, amax_x, amax_w1, amax_w2, amax_b1,
                    amax_hid, amax_y, workspace, workspace_bytes, FfnNorm{}, stream);
}

extern "C" int rscotr_ffn_h3_ln(const float* X, int M, int C, int H, const float* ln_weight, const float* ln_bias, float ln_eps,
                                float* ln_out, float* ln_mean, float* ln_rstd, const void* W1f, const float* b1, const void* W2f,
                                const float* b2, float* Pre, float* Hid, const float* resid, float* Y, const float* yscale, int rows_per,
                                const uint32_t* amax_ln_weight, const uint32_t* amax_ln_bias, const uint32_t* amax_w1,
                                const uint32_t* amax_w2, const uint32_t* amax_b1, uint32_t* amax_ln_out, uint32_t* amax_hid,
                                uint32_t* amax_y, float* workspace, int64_t workspace_bytes, void* stream) {
  if (!ln_weight) return fail(RSCOTR_E_ARG, "ffn_h3_ln: null LayerNorm weight");
  const FfnNorm ln{ln_weight, ln_bias, ln_eps, ln_out, ln_mean, ln_rstd, amax_ln_weight, amax_ln_bias, amax_ln_out};
  return ffn_h3_run(X, M, C, H, W1f, b1, W2f, b2, FFN_GELU, nullptr, Pre, Hid, resid, Y, nullptr, yscale, rows_per, nullptr, amax_w1, amax_w2,
                    amax_b1, amax_hid, amax_y, workspace, workspace_bytes, ln, stream);
}

// ---- the one-Linear launch (lin_h3_kernel)
template <int C, bool LN, int NT = 2>
static void lin_launch(const LinParams& p, hipStream_t s) {
  constexpr size_t lds = (size_t)(C / 32) * (16 * NT) * FFN_LDRB;
  static bool attr_set = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lin_h3_kernel<C, LN, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return true;
  }();
  (void)attr_set;
  hipLaunchKernelGGL((lin_h3_kernel<C, LN, NT>), dim3((unsigned)((p.M + 16 * NT - 1) / (16 * NT)), (unsigned)p.col_groups), dim3(512), lds, s, p);
}

extern "C" int rscotr_lin_h3_ok(int M, int N, int K) {
  const bool k_ok = K == 96 || K == 128 || K == 192 || K == 256 || K == 288 || K == 384 || K == 576 || K == 768 || K == 1152;
  return (k_ok && N >= 32 && N % 32 == 0 && M >= 1 && (long)M * N * 4 < (1l << 32) && (long)M * K * 4 < (1l << 32)) ? 1 : 0;
}

static int lin_h3_run(const float* X, int M, int N, int K, const void* Wf, const float* bias, const float* resid, float* Y,
                      const float* xscale, const float* yscale, int rows_per, const uint32_t* amax_x, const uint32_t* amax_w,
                      uint32_t* amax_y, const FfnNorm& ln, void* stream) {
  if (!rscotr_lin_h3_ok(M, N, K)) return fail(RSCOTR_E_SHAPE, "lin_h3: M=%d N=%d K=%d (K in {96, 128, 192, 256, 288, 384, 576, 768, 1152}, N %% 32 == 0)", M, N, K);
  if (!X || !Wf || !Y || (!amax_x && !ln.g) || !amax_w) return fail(RSCOTR_E_ARG, "lin_h3: null argument");
  if ((xscale || yscale) && rows_per <= 0) return fail(RSCOTR_E_ARG, "lin_h3: rows_per with a row scale");
  if (((uintptr_t)X | (uintptr_t)Wf | (uintptr_t)Y | (uintptr_t)resid | (uintptr_t)bias) & 15)
    return fail(RSCOTR_E_ALIGN, "lin_h3: operands must be 16-byte aligned");
  if (ln.g) {
    if (!(K == 96 || K == 192 || K == 384)) return fail(RSCOTR_E_SHAPE, "lin_h3_ln: the norm prologue exists at K in {96, 192, 384} (K=%d)", K);
    if (!ln.out || !ln.mean || !ln.rstd || !ln.amax_g || xscale) return fail(RSCOTR_E_ARG, "lin_h3_ln: null argument (or a row scale on X)");
    if (((uintptr_t)ln.g | (uintptr_t)ln.b | (uintptr_t)ln.out) & 15) return fail(RSCOTR_E_ALIGN, "lin_h3_ln: operands must be 16-byte aligned");
  }
  LinParams p{};
  p.X = X; p.M = M; p.N = N; p.Wf = static_cast<const uint4*>(Wf); p.bias = bias; p.resid = resid; p.Y = Y;
  p.xscale = xscale; p.yscale = yscale; p.rows_per = rows_per > 0 ? rows_per : 1;
  p.amax_x = amax_x; p.amax_w = amax_w; p.amax_y = amax_y;
  p.ln_g = ln.g; p.ln_b = ln.b; p.ln_eps = ln.eps; p.ln_out = ln.out; p.ln_mean = ln.mean; p.ln_rstd = ln.rstd;
  p.amax_g = ln.amax_g; p.amax_bt = ln.amax_b; p.amax_ln = ln.amax_out;
  // fewer than 200 row tiles: one workgroup per (row tile, 256 columns)
  p.col_groups = (M + 31) / 32 >= 200 ? 1 : (N + 255) / 256;
  if (K == 1152 && M >= 8192) return fail(RSCOTR_E_SHAPE, "lin_h3: K = 1152 is a few-row width (M=%d)", M);
  hipStream_t s = static_cast<hipStream_t>(stream);
  ProfScope prof(PROF_GEMM, 2.0 * M * (double)N * K, s, "rscotr::lin_h3_kernel<%d, %s>", K, ln.g ? "true" : "false");
  static const int rows64 = getenv("RSCOTR_LIN_ROWS64") ? atoi(getenv("RSCOTR_LIN_ROWS64")) : 0;  // (A/B: 64-row workgroups at K = 96)
  if (rows64 && K == 96 && M >= 16384) {
    if (ln.g) lin_launch<96, true, 4>(p, s); else lin_launch<96, false, 4>(p, s);
  } else if (ln.g) {
    if (K == 96) lin_launch<96, true>(p, s);
    else if (K == 192) lin_launch<192, true>(p, s);
    else lin_launch<384, true>(p, s);
  } else switch (K) {
    case 96: lin_launch<96, false>(p, s); break;
    case 128: lin_launch<128, false>(p, s); break;
    case 192: lin_launch<192, false>(p, s); break;
    case 256: lin_launch<256, false>(p, s); break;
    case 288: lin_launch<288, false>(p, s); break;
    case 384: lin_launch<384, false>(p, s); break;
    case 576: lin_launch<576, false>(p, s); break;
    case 768: {
      static const int rows16 = getenv("RSCOTR_LIN_ROWS16") ? atoi(getenv("RSCOTR_LIN_ROWS16")) : 0;  // (A/B: 16-row workgroups for the 512-row launches of Swin stage 4)
      if (rows16 && M <= 1024) lin_launch<768, false, 1>(p, s); else lin_launch<768, false>(p, s);
      break;
    }
    default: lin_launch<1152, false, 1>(p, s); break;  // (16-row workgroups: 92 KB of planes)
  }
  return check_launch("lin_h3");
}

extern "C" int rscotr_lin_h3(const float* X, int M, int N, int K, const void* Wf, const float* bias, const float* resid, float* Y,
                             const float* xscale, const float* yscale, int rows_per, const uint32_t* amax_x, const uint32_t* amax_w,
                             uint32_t* amax_y, void* stream) {
  return lin_h3_run(X, M, N, K, Wf, bias, resid, Y, xscale, yscale, rows_per, amax_x, amax_w, amax_y, FfnNorm{}, stream);
}

extern "C" int rscotr_lin_h3_ln(const float* X, int M, int N, int K, const float* ln_weight, const float* ln_bias, float ln_eps,
                                float* ln_out, float* ln_mean, float* ln_rstd, const void* Wf, const float* bias, const float* resid,
                                float* Y, const float* yscale, int rows_per, const uint32_t* amax_ln_weight, const uint32_t* amax_ln_bias,
                                const uint32_t* amax_w, uint32_t* amax_ln_out, uint32_t* amax_y, void* stream) {
  if (!ln_weight) return fail(RSCOTR_E_ARG, "lin_h3_ln: null LayerNorm weight");
  const FfnNorm ln{ln_weight, ln_bias, ln_eps, ln_out, ln_mean, ln_rstd, amax_ln_weight, amax_ln_bias, amax_ln_out};
  return lin_h3_run(X, M, N, K, Wf, bias, resid, Y, nullptr, yscale, rows_per, nullptr, amax_w, amax_y, ln, stream);
}
