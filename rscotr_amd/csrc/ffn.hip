// Two-layer FFN block of the shared encoder as ONE launch (round 6): y = act(x W1^T + b1) W2^T + b2 (+ resid) with the wide
// intermediate leaving the kernel only for the backward pass, never to be read again by the forward.
//
// Replaces (reference call sites): mmcv FFN inside the encoder's BaseTransformerLayer (cfg
// configs/multi/MTL_slvlcls_swin-t-p4-w7_1x1_resisc&dior&potsdam.py:44-49, built at models/multi/multitask_learner.py:51,
// called from models/multi/seg_head/pixel_decoder.py:134-146 and models/multi/bbox_head/transformer.py:211-221): two
// nn.Linear around a ReLU.  As two products (rscotr_gemm_f32_r twice) the 10880 x 2048 hidden tensor is written (89 MB), read
// back by the second product (89 MB), and each launch pays its own ramp / drain and converts its operands again: 73 + 59 us per
// layer forward, 65-76 + 57 us for the mirrored pair dH = (g W2) * gate, dX = dH W1 of backward — the largest block of the round.
//
// Structure.  A workgroup (512 threads = 8 wavefronts, one per CU: 144 KB of LDS) owns 64 rows.  The rows' fp16 planes
// (h | l of the "h3" split product, gemm.hip) are staged in LDS ONCE; then, per chunk of 256 hidden columns:
//   phase A   hidden chunk (64 x 256) = x W1[chunk]^T: wavefront w owns hidden columns 32 w .. 32 w + 31 of the chunk, both
//             32-row tiles; A fragments from the LDS planes, B fragments straight from L2 into registers (below);
//   epilogue  bias, ReLU (gate bits out) or gate (bits in), the fp32 chunk to global memory (the weight gradients of backward
//             read it), its planes into LDS — the k block a wavefront writes is its own;
//   phase B   y (64 x 256) += chunk W2[:, chunk]^T: wavefront w owns output columns 32 w .., A fragments from the chunk's planes.
// The weights arrive as FRAGMENT-MAJOR fp16 planes (rscotr_gemm_split_weights_frag: [n tile of 32][k step of 16][h | l][lane][8
// halfs], written once per optimizer step): the B operand of a wavefront's 32 x 32 x 16 MFMA is ONE contiguous 1 KB load, no
// wavefront shares a fragment with another (each owns its n tile), so nothing goes through LDS and every weight byte is read
// once per workgroup (4 MB for 256 -> 2048 -> 256; L2-resident).  A ring of eight such loads per wavefront runs ahead of the MFMAs
// across phase and chunk boundaries.
//
// Numerics: the same three-MFMA fp16 split product as gemm_h3_* (same term order per k step: the hidden chunk is BIT-IDENTICAL to
// rscotr_gemm_f32_r's output for the first Linear).  The planes of the hidden chunk are scaled by a power of two taken from an
// A-PRIORI bound, C * max|x| * max|W1| + max|b1| (the true maximum is only known after the last workgroup): any upper bound is a
// valid range (ops/ranges.py); a bound 2^t too loose moves the point where the planes start to lose RELATIVE precision from
// 2^-26 to 2^(t-26) of the maximum — below it the error is 2^-48 2^t of the maximum absolute, invisible in an fp32 product.
#include "gemm_common.h"
#include "rscotr.h"

namespace rscotr {

constexpr int FFN_BM = 64;    // rows per workgroup
constexpr int FFN_HC = 256;   // hidden columns per chunk (8 wavefronts x 32)
constexpr int FFN_LDR = 72;   // halfs per LDS row of a 32-k stage: h[32] | l[32] | 8 pad (144 bytes: conflict-free 16-byte fragment reads, as SplitOperand)
constexpr int FFN_STAGE = FFN_BM * FFN_LDR / 2;  // dwords per 32-k stage
constexpr int FFN_RING = 8;   // B fragment loads in flight per wavefront (= 4 k steps)

struct FfnParams {
  const float* X;
  int M, H;
  const uint4* W1f;
  const float* b1;
  const uint4* W2f;
  const float* b2;
  unsigned* bits;
  float* Hid;
  const float* resid;
  float* Y;
  const unsigned *amax_x, *amax_w1, *amax_w2, *amax_b1;
  unsigned *amax_hid, *amax_y;
};

template <int C>
constexpr size_t ffn_lds_bytes() { return (size_t)(C / 32 + FFN_HC / 32) * FFN_STAGE * 4; }

// GATE = false: hidden = relu(x W1^T + b1), one bit per element [hidden > 0] written to p.bits; true: hidden = (x W1^T) gated by
// the bits a forward launch of the same shape left (dH = (g W2) * [h > 0]).  Bit layout (opaque to callers, the same in both
// directions): uint32 [row tile][chunk][wavefront][lane], bit i * 16 + r = accumulator element r of the lane's 32 x 32 tile i.
template <int C, bool GATE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void ffn_h3_kernel(FfnParams p) {
  static_assert(C == 256, "phase B maps one 32-column output tile to each of the 8 wavefronts");
  extern __shared__ __attribute__((aligned(16))) unsigned ffn_lds[];
  constexpr int KS1 = C / 16, KS2 = FFN_HC / 16, XST = C / 32;
  constexpr int D = FFN_RING;
  static_assert((2 * KS1) % D == 0 && (2 * KS2) % D == 0, "the ring position is static across phases");
  unsigned* xs = ffn_lds;
  unsigned* hs = ffn_lds + XST * FFN_STAGE;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, fr = lane & 31, g = lane >> 5;
  const int m0 = blockIdx.x * FFN_BM;
  const int nchunks = p.H / FFN_HC;

  // value-range words: requested first, reduced after the operand loads have been requested too (cold lines)
  const long sub = (long)(lane & (kAmaxPlanes - 1)) * kAmaxStride;
  const unsigned rx = p.amax_x[sub], r1 = p.amax_w1[sub], r2 = p.amax_w2[sub], rb = p.amax_b1 ? p.amax_b1[sub] : 0u;

  // B fragments: ring of D loads; sequence A(0) B(0) A(1) B(1) ... of 2 * KS loads each
  // (buffer loads: ONE address register per weight — the lane's and the wavefront's part — and the chunk / k-step part as a scalar
  //  offset; with flat pointers the unrolled loop kept a 64-bit address per load alive and spilled)
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(p.W1f), 0, p.H * C * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(p.W2f), 0, p.H * C * 4, 0x00020000);
  const int vA = (w * KS1 * 128 + lane) * 16;           // chunk c: + c * 8 * KS1 * 2048 bytes
  const int vB = (w * (p.H / 16) * 128 + lane) * 16;    // chunk c: + c * KS2 * 2048 bytes
  auto ldA = [&](int c, int j) { return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rA, vA, c * (8 * KS1 * 2048) + j * 1024, 0)); };
  auto ldB = [&](int c, int j) { return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rB, vB, c * (KS2 * 2048) + j * 1024, 0)); };
  uint4 ring[D];
#pragma unroll
  for (int j = 0; j < D; ++j) ring[j] = ldA(0, j);
  // the row tile's tensors through descriptors that end at row M: rows past the end read zeros and drop their stores — no
  // per-element guards (which cost a branch and a live 64-bit address per store of the chunk epilogue)
  const int rows_ok = min(FFN_BM, p.M - m0);
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X + (long)m0 * C), 0, rows_ok * C * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc(p.Hid + (long)m0 * p.H, 0, rows_ok * p.H * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc(p.Y + (long)m0 * C, 0, rows_ok * C * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.resid ? p.resid + (long)m0 * C : p.X), 0, rows_ok * C * 4, 0x00020000);

  // the rows' planes -> LDS
  {
    constexpr int Q = C / 4, NV = FFN_BM * Q / 512;
    float4 v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * 512;  // (rows past M: out of the descriptor's range, zeros)
      v[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rX, idx * 16, 0, 0));
    }
    const int ex = h3_scale_exp(amax_fold(rx));
    const H3Scale hx{__uint_as_float((unsigned)ex << 23), __uint_as_float((unsigned)(ex + 11) << 23)};
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = tid + i * 512;
      const int row = idx / Q, kq = (idx % Q) * 4;
      unsigned ab[3], cd[3];
      split_pair_h(v[i].x, v[i].y, hx, ab);
      split_pair_h(v[i].z, v[i].w, hx, cd);
      unsigned* dst = xs + (kq / 32) * FFN_STAGE + (row * FFN_LDR + (kq % 32)) / 2;
      *reinterpret_cast<uint2*>(dst) = make_uint2(ab[0], cd[0]);
      *reinterpret_cast<uint2*>(dst + 16) = make_uint2(ab[1], cd[1]);
    }
  }
  const unsigned ux = amax_fold(rx), u1 = amax_fold(r1), u2 = amax_fold(r2), ub = amax_fold(rb);
  const int ex = h3_scale_exp(ux), e1 = h3_scale_exp(u1), e2 = h3_scale_exp(u2);
  // |hidden| <= C max|x| max|W1| + max|b1| (gating only shrinks it)
  const float bound = (float)C * __uint_as_float(ux) * __uint_as_float(u1) + __uint_as_float(ub);
  const int eh = h3_scale_exp(__float_as_uint(bound));
  const H3Scale hh{__uint_as_float((unsigned)eh << 23), __uint_as_float((unsigned)(eh + 11) << 23)};
  const float invx = __uint_as_float((unsigned)(254 - ex) << 23), inv1 = __uint_as_float((unsigned)(254 - e1) << 23);
  const float invh = __uint_as_float((unsigned)(254 - eh) << 23), inv2 = __uint_as_float((unsigned)(254 - e2) << 23);

  f32x16 ya[2], yb[2];  // y tiles (rows 32 i .., columns 32 w ..): h h terms / (l h + h l) 2^11
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) { ya[i][r] = 0.f; yb[i][r] = 0.f; }
  float amx = 0.f;
  __syncthreads();

  // Four k steps (64 k) of a phase: A fragments of both 32-row tiles from the plane image S (lane part of the address in `lo`), the
  // eight ring entries as B fragments, each replaced by the load eight positions ahead (LAST: the first eight of the next phase).
  // The k loop runs over such groups (not fully unrolled: the scheduler hoisted every fragment read of an unrolled phase to its
  // top and spilled a hundred registers).
  const int lo = (fr * FFN_LDR + 8 * g) / 2;  // dwords
  auto steps4 = [&](const unsigned* S, int kg, f32x16 (&ta)[2], f32x16 (&tb)[2], auto ld_cur, auto ld_next, bool last) {
    const unsigned* q0 = S + (kg >> 1) * FFN_STAGE + lo;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      f16x8 ah[2], al[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const unsigned* q = q0 + (u >> 1) * FFN_STAGE + (i * 32 * FFN_LDR + 16 * (u & 1)) / 2;
        ah[i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(q));
        al[i] = __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(q + 16));
      }
      const f16x8 bh = __builtin_bit_cast(f16x8, ring[2 * u]), bl = __builtin_bit_cast(f16x8, ring[2 * u + 1]);
      if (last) {
        ring[2 * u] = ld_next(2 * u);
        ring[2 * u + 1] = ld_next(2 * u + 1);
      } else {
        ring[2 * u] = ld_cur(2 * kg + 2 * u + D);
        ring[2 * u + 1] = ld_cur(2 * kg + 2 * u + 1 + D);
      }
      // term order of gemm_h3_*: l h, h l into the second accumulator, h h into the first
#pragma unroll
      for (int i = 0; i < 2; ++i) tb[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh, tb[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) tb[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl, tb[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) ta[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh, ta[i], 0, 0, 0);
    }
  };

  for (int c = 0; c < nchunks; ++c) {
    const int cn = min(c + 1, nchunks - 1);  // (past the last chunk: a harmless re-read)
    auto a_cur = [&](int j) { return ldA(c, j); };
    auto b_cur = [&](int j) { return ldB(c, j); };
    auto a_next = [&](int j) { return ldA(cn, j); };
    // ---- phase A
    f32x16 ha[2], hb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) { ha[i][r] = 0.f; hb[i][r] = 0.f; }
#pragma unroll 1
    for (int kg = 0; kg < KS1 - 4; kg += 4) steps4(xs, kg, ha, hb, a_cur, b_cur, false);
    steps4(xs, KS1 - 4, ha, hb, a_cur, b_cur, true);
    __syncthreads();  // every wavefront has finished phase B of the previous chunk: the chunk image is free
    // ---- epilogue of the chunk: column n of the hidden tensor = k of phase B, k block w of the image is this wavefront's
    {
      const int n = c * FFN_HC + w * 32 + fr;
      const int vH = (4 * g * p.H + n) * 4;
      const float bv = (!GATE && p.b1) ? p.b1[n] : 0.f;
      const long widx = (((long)blockIdx.x * nchunks + c) * 8 + w) * 64 + lane;
      unsigned bits = GATE ? p.bits[widx] : 0u;
      unsigned short* img = reinterpret_cast<unsigned short*>(hs + w * FFN_STAGE);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
          float v = fmaf(hb[i][r], 0x1p-11f, ha[i][r]) * invx * inv1;
          if (!GATE) {
            v = fmaxf(v + bv, 0.f);
            bits |= (unsigned)(v > 0.f) << (i * 16 + r);
          } else {
            v = ((bits >> (i * 16 + r)) & 1u) ? v : 0.f;
          }
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rH, vH, (i * 32 + (r & 3) + 8 * (r >> 2)) * p.H * 4, 0);
          amx = fmaxf(amx, fabsf(v));
          const float y = v * hh.sc;
          const _Float16 h16 = (_Float16)y;
          const _Float16 l16 = (_Float16)fmaf((float)h16, -2048.f, v * hh.sc2);
          img[row * FFN_LDR + fr] = __builtin_bit_cast(unsigned short, h16);
          img[row * FFN_LDR + 32 + fr] = __builtin_bit_cast(unsigned short, l16);
        }
      if (!GATE) p.bits[widx] = bits;
    }
    __syncthreads();
    // ---- phase B
#pragma unroll 1
    for (int kg = 0; kg < KS2 - 4; kg += 4) steps4(hs, kg, ya, yb, b_cur, a_next, false);
    steps4(hs, KS2 - 4, ya, yb, b_cur, a_next, true);
  }
  amax_commit(p.amax_hid, amx);

  // ---- y = (h h + (l h + h l) 2^-11) 2^-(s_hidden + s_w2) + b2 (+ resid)
  {
    const int n = w * 32 + fr;
    const int vY = (4 * g * C + n) * 4;
    const float bv = p.b2 ? p.b2[n] : 0.f;
    float amy = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float e[16];
#pragma unroll
      for (int r = 0; r < 16; ++r)
        e[r] = p.resid ? __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rR, vY, (i * 32 + (r & 3) + 8 * (r >> 2)) * C * 4, 0)) : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = fmaf(yb[i][r], 0x1p-11f, ya[i][r]) * invh * inv2 + bv + e[r];
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rY, vY, (i * 32 + (r & 3) + 8 * (r >> 2)) * C * 4, 0);
        amy = fmaxf(amy, fabsf(v));
      }
    }
    amax_commit(p.amax_y, amy);
  }
}

// Fragment-major fp16 planes of a weight for ffn_h3_kernel (rscotr_gemm_split_weights_frag): table rows {W, planes, rows of W,
// cols of W, ldw, 0, first block, transposed, range word of the parameter} (int64 x 9, as rscotr_gemm_split_weights_h3).
// transposed = 0: plane rows n = rows of W, reduction k over its columns (y = x W^T); 1: plane rows = columns of W, reduction over
// its rows (dx = dy W).  Layout: uint4 [n / 32][k / 16][h | l][lane], lane l holding k = 16 ks + 8 (l >> 5) .. + 7 of row
// n = 32 nt + (l & 31) — the B operand of v_mfma_f32_32x32x16_f16, one contiguous 1 KB load per wavefront.  Plane rows % 32 == 0,
// reduction % 16 == 0 (host-checked); one thread per (n tile, k step, lane): 8 values in, two 16-byte records out.
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
  const int nks = red / 16;
  const int lane = (int)(idx & 63);
  const long rest = idx >> 6;
  const int ks = (int)(rest % nks), nt = (int)(rest / nks);
  if (nt >= rows / 32) return;
  const int n = nt * 32 + (lane & 31), k0 = ks * 16 + 8 * (lane >> 5);
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

extern "C" int rscotr_ffn_h3_ok(int M, int C, int H) { return (C == 256 && H >= FFN_HC && H % FFN_HC == 0 && M >= 1) ? 1 : 0; }

extern "C" int64_t rscotr_ffn_h3_bits_words(int M, int H) { return (int64_t)((M + FFN_BM - 1) / FFN_BM) * (H / FFN_HC) * 512; }

extern "C" int rscotr_ffn_h3(const float* X, int M, int C, int H, const void* W1f, const float* b1, const void* W2f, const float* b2,
                             void* bits, int gate, float* Hid, const float* resid, float* Y, const uint32_t* amax_x,
                             const uint32_t* amax_w1, const uint32_t* amax_w2, const uint32_t* amax_b1, uint32_t* amax_hid,
                             uint32_t* amax_y, void* stream) {
  if (!rscotr_ffn_h3_ok(M, C, H)) return fail(RSCOTR_E_SHAPE, "ffn_h3: M=%d C=%d H=%d (C == 256, H %% 256 == 0)", M, C, H);
  if (!X || !W1f || !W2f || !bits || !Hid || !Y || !amax_x || !amax_w1 || !amax_w2) return fail(RSCOTR_E_ARG, "ffn_h3: null argument");
  if (((uintptr_t)X | (uintptr_t)W1f | (uintptr_t)W2f) & 15) return fail(RSCOTR_E_ALIGN, "ffn_h3: operands must be 16-byte aligned");
  FfnParams p{};
  p.X = X; p.M = M; p.H = H;
  p.W1f = static_cast<const uint4*>(W1f); p.b1 = b1;
  p.W2f = static_cast<const uint4*>(W2f); p.b2 = b2;
  p.bits = static_cast<unsigned*>(bits); p.Hid = Hid; p.resid = resid; p.Y = Y;
  p.amax_x = amax_x; p.amax_w1 = amax_w1; p.amax_w2 = amax_w2; p.amax_b1 = gate ? nullptr : amax_b1;
  p.amax_hid = amax_hid; p.amax_y = amax_y;
  constexpr size_t lds = ffn_lds_bytes<256>();
  static bool attr_set = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_h3_kernel<256, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_h3_kernel<256, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    return true;
  }();
  (void)attr_set;
  const dim3 grid((unsigned)((M + FFN_BM - 1) / FFN_BM));
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (gate) hipLaunchKernelGGL((ffn_h3_kernel<256, true>), grid, dim3(512), lds, s, p);
  else hipLaunchKernelGGL((ffn_h3_kernel<256, false>), grid, dim3(512), lds, s, p);
  return check_launch("ffn_h3");
}
