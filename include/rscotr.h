/* rscotr.h — C ABI of librscotr.so, the MI355X (gfx950) kernels behind the RSCoTr
 * multi-task co-training step.
 *
 * Every entry is what the reference's operator layer would bind for this path; the reference
 * file:line each one replaces is cited next to it (paths relative to the reference repo;
 * "mmcv"/"mmdet"/"mmseg"/"scipy" name the un-vendored dependency the reference calls there).
 *
 * Conventions
 *   - plain pointers and sizes only; all tensors are contiguous row-major fp32 unless noted;
 *   - the CALLER allocates every buffer (inputs, outputs, workspaces); entries never allocate,
 *     free, or keep a pointer after returning;
 *   - device entries take the hipStream_t to launch on as `void* stream` and never synchronise;
 *   - return value: 0 = ok, negative = RSCOTR_E_*; rscotr_last_error() returns a thread-local
 *     message for the last failure on the calling thread. No exception crosses the ABI.
 */
#ifndef RSCOTR_H
#define RSCOTR_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define RSCOTR_OK 0
#define RSCOTR_E_SHAPE (-1)
#define RSCOTR_E_ALIGN (-2)
#define RSCOTR_E_ARCH (-3)
#define RSCOTR_E_LAUNCH (-4)
#define RSCOTR_E_ARG (-5)

/* ABI revision: bumped whenever an exported entry changes its argument list (round 5 inserted amax_out / nbytes before
 * `stream` in the LayerNorm, window-attention, MSDA backward, pack4, splitk_flush and grouped-dW entries = revision 6;
 * round 6 = 7).  rscotr_version() returns the revision the shared object was BUILT with; a binding compares the two before
 * its first call (rscotr_amd/_lib.py does) — a stale .so would take a stream handle for a pointer. */
#define RSCOTR_ABI_VERSION 10
int rscotr_version(void);
const char* rscotr_last_error(void);
int rscotr_device_count(void);

/* Launch-site profiling for the benchmark's rooflines: while enabled, rscotr_gemm_f32 / rscotr_msda_fwd /
 * rscotr_msda_bwd bracket one launch in every_* of their kind with a pair of HIP events recorded on the launch
 * stream from inside the library, and keep the algorithmic work of that call (flop for the GEMM, bytes for
 * MSDA).  rscotr_prof_pause() stops recording and returns the number of records; rscotr_prof_get(i, ...) returns
 * record i (the stream must be idle); rscotr_prof_disable() releases the events.  Not capturable in a hipGraph. */
int rscotr_prof_enable(int every_gemm, int every_msda_fwd, int every_msda_bwd, int max_records);
int rscotr_prof_pause(void);
int rscotr_prof_empty(int n, void* stream); /* n empty brackets (name "rscotr::empty_bracket"): the cost of the bracketing itself */
int rscotr_prof_get(int i, int* kind, double* work, float* ms, char* name, int name_len);
int rscotr_prof_disable(void);

/* ---- multi-scale deformable attention ----------------------------------------------------
 * Replaces mmcv MultiScaleDeformableAttnFunction (ext_module.ms_deform_attn_forward/backward),
 * reached from models/multi/seg_head/pixel_decoder.py:134-146,
 * models/multi/bbox_head/transformer.py:211-221 and :258-269.
 *   value (B,Nk,H,D) | spatial_shapes (L,2) int64 rows (H_l,W_l), DEVICE memory |
 *   level_start_index (L) int64, DEVICE memory | loc (B,Nq,H,L,P,2) as (x,y) in [0,1] |
 *   attn (B,Nq,H,L,P) | out (B,Nq,H*D).  D in {16,32,64}, P in {1,2,4,8}.
 * Backward: grad_loc / grad_attn are fully overwritten.  grad_value (B,Nk,H,D), by what the caller provides:
 *   shapes_host = HOST copy of spatial_shapes (L <= 8) + rscotr_msda_bwd_tiled_workspace() bytes: TILE ACCUMULATION (the
 *     default of rscotr_amd.ops) — the sample kernel leaves one 4-byte bin word per sample; one workgroup per (b, h, level,
 *     tile of 16 x 8 bins, sample chunk) scans the level's bin words, keeps those of its tile in sample order, sorts them by
 *     bin in LDS, re-derives every kept sample's tap weights from loc / attn and sums every bin's four tap rows in registers; a combine kernel folds the tiles' cells per token in fixed
 *     order: fully overwritten, no atomics, BIT-REPRODUCIBLE, 3 launches;
 *   shapes_host NULL + a `workspace` of rscotr_msda_bwd_workspace() bytes (16-byte aligned, contents irrelevant): the
 *     samples are counting-sorted by destination token (one wavefront per chunk: the ranks, hence the summation order,
 *     depend on the data only) and every token pulls its taps; long tap lists are cut into work items whose partial
 *     rows are folded in order: fully overwritten, no fp32 atomics, bit-reproducible, 8 launches (kept as an independent
 *     formulation);
 *   workspace NULL / too small: scatter with fp32 atomics into a grad_value the caller has ZEROED (order-dependent). */
int rscotr_msda_fwd(const float* value, const int64_t* spatial_shapes,
                    const int64_t* level_start_index, const float* loc, const float* attn,
                    float* out, int B, int Nk, int Nq, int H, int D, int L, int P, void* stream);
int rscotr_msda_bwd(const float* value, const int64_t* spatial_shapes,
                    const int64_t* level_start_index, const float* loc, const float* attn,
                    const float* grad_out, float* grad_value, float* grad_loc, float* grad_attn,
                    int B, int Nk, int Nq, int H, int D, int L, int P, const int64_t* shapes_host,
                    void* workspace, int64_t workspace_bytes, uint32_t* amax_grad_value,
                    void* stream);
int64_t rscotr_msda_bwd_workspace(int B, int Nk, int Nq, int H, int L, int P);
int64_t rscotr_msda_bwd_tiled_workspace(const int64_t* shapes_host, int B, int Nk, int Nq, int H, int D, int L, int P);
/* The element-wise prologue of mmcv MultiScaleDeformableAttention.forward in one launch per direction:
 *   attn (B,Nq,H,L,P) = softmax over L*P of logit (B,Nq,H,L*P);
 *   loc (B,Nq,H,L,P,2) = ref_xy + off / norm[l]            for ref (B,Nq,L,2), norm (L,2) = (W_l, H_l), or
 *                      = ref_xy + off / P * ref_wh * 0.5   for ref (B,Nq,L,4) (norm unused, may be NULL);
 * backward: grad_off from grad_loc, grad_logit = attn * (grad_attn - sum attn*grad_attn); the reference points get
 * no gradient (they are detached on this path: bbox_head/transformer.py:115-121).  L*P <= 64.
 * Row (b, q) of off / grad_off starts at element (b*Nq+q) * ld_off (>= H*L*P*2, even), of logit / grad_logit at
 * (b*Nq+q) * ld_logit (>= H*L*P): the two projections may be the column blocks of ONE (B*Nq, 3*H*L*P) product.
 * ref_levels = L, or 1 for reference points shared by the levels (ref (B,Nq,1,.): valid ratios all one). */
int rscotr_msda_prep_fwd(const float* off, const float* logit, const float* ref, const float* norm, float* loc,
                         float* attn, int B, int Nq, int H, int L, int P, int refdim, int ld_off, int ld_logit,
                         int ref_levels, void* stream);
int rscotr_msda_prep_bwd(const float* grad_loc, const float* grad_attn, const float* attn, const float* ref,
                         const float* norm, float* grad_off, float* grad_logit, int B, int Nq, int H, int L, int P,
                         int refdim, int ld_off, int ld_logit, int ref_levels, uint32_t* amax_out, void* stream);
/* The prologue INSIDE the sampling kernel (round 5).  rscotr_msda_fwd_prep = rscotr_msda_prep_fwd + rscotr_msda_fwd in one launch: the
 * threads that stage a tile's samples compute the softmax over the 16 logits of a (query, head) and the sampling locations
 * themselves (the prologue kernel's arithmetic: same loc / attn, bit for bit) and still leave loc / attn in global memory for
 * the backward — one launch less and no second pass over the 12 bytes per sample.  Geometries rscotr_msda_fused_ok answers 1 for
 * (L * P == 16 — the configs' 4 levels x 4 points —, D in {16, 32, 64}); elsewhere the entry fails and callers use the pair. */
int rscotr_msda_fused_ok(int Nk, int H, int D, int L, int P);
int rscotr_msda_fwd_prep(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index,
                         const float* off, const float* logit, int ld_off, int ld_logit, const float* ref,
                         const float* norm, int refdim, int ref_levels, float* loc, float* attn, float* out, int B,
                         int Nk, int Nq, int H, int D, int L, int P, void* stream);

/* ---- fp32 GEMM on the matrix cores, fused epilogue ----------------------------------------------
 * Replaces torch F.linear / nn.Linear and 1x1 / patchify nn.Conv2d (and the two backward
 * contractions autograd derives from them) wherever the un-vendored layers the reference builds
 * use them: mmdet SwinTransformer qkv/proj/MLP and PatchMerging.reduction
 * (configs/multi/MTL_slvlcls_...potsdam.py:9-25), mmcv FFN and the MultiScaleDeformableAttention /
 * MultiheadAttention projections of the shared encoder and both decoders (:34-50, :76-98, :139-160),
 * ChannelMapper 1x1 convs (:26-33), the heads' Linear branches (models/multi/bbox_head/dino_head.py:40-47,
 * models/multi/seg_head/mask2former_head.py:60-83, models/multi/cls_head/slvl_cls_head.py:14-23).
 *   C[m,n] = epilogue( sum_k Aop[m,k] * Bop[n,k] ),  Aop[m,k] = a_kmajor ? A[k*lda+m] : A[m*lda+k],
 *   Bop[n,k] = b_kmajor ? B[k*ldb+n] : B[n*ldb+k]
 *   epilogue(v): v += bias[n] (bias may be NULL); if (pre) pre[m*ldc+n] = v;
 *                act 0 none | 1 relu | 2 gelu(erf) | 3 v *= (aux>0) | 4 v *= gelu'(aux)   [aux: (M,N), ldc] | 5 / 6: rscotr_gemm_relu_bits_ok;
 *                v += resid[m*ldc+n] (resid may be NULL); if (accumulate) v += C[m*ldc+n].
 * F.linear(x,W,b) = (A=x,B=W,0,0); dx = (A=dy,B=W,0,1); dW = (A=dy,B=x,1,1).
 * rowsum (may be NULL; needs a_kmajor): rowsum[m] (+)= sum_k Aop[m,k] — with A = dy this is the bias
 * gradient of the Linear whose dW the same call computes (replaces a separate column-sum pass).
 * rowscale (may be NULL): after bias / activation, row m is multiplied by rowscale[m / rows_per_scale] (before
 * resid is added); kscale (may be NULL; needs a_kmajor): Aop[m,k] is multiplied by kscale[k / krows_per_scale]
 * (the row sums see the scaled operand).  Together they fold the per-sample DropPath factor of the Swin blocks
 * (mmdet SwinBlock: x + drop_path(attn/ffn(...)), cfg ...potsdam.py:20) into the proj / fc2 Linear: forward
 * y = x + s_b (h W^T + b); backward dH = s_b (g W), dW = (s g)^T h, db = sum s g.
 * out2 (may be NULL; same ldc as C): a second output C2 = C + resid, while C itself is then stored WITHOUT the residual
 * (C = epilogue value [+ old C]; C2 = C + resid).  Backward of an attention block wants one product both alone (the
 * gradient of the positional embedding) and merged with the gradient of the residual path (the gradient of the block
 * input): this replaces the element-wise add autograd would launch (mmcv MultiheadAttention / MultiScaleDeformableAttention
 * `query + query_pos`, `identity + dropout(out)`).
 * `workspace` (may be NULL) holds split-K slabs (long reductions on short grids are cut along K and combined
 * in fixed order by a second kernel); rscotr_gemm_f32_workspace() returns the bytes the split path wants
 * for a problem (0 = it never splits).
 * One entry, three kernels behind it (same results up to fp32 summation order; all deterministic): the LDS-tiled
 * kernel (64x64 / 128x64 / 128x32 tiles, k-groups, split-K); a low-latency kernel for small products (K <= 512,
 * K % 8 == 0, <= 512 output tiles of 32x32: wavefronts split K, fragments loaded straight from global memory); a
 * direct (LDS-free) kernel for k-major x k-major weight gradients with K >= 16384 and a 3-4 block output. */
/* Inner product of the tiled GEMM kernel (rscotr_gemm_f32, rscotr_gemm_f32_batched, rscotr_gemm_f32_dw_slabs):
 * 0 = fp32 matrix pipe (v_mfma_f32_32x32x2_f32) everywhere; 3 = "bf16x6" (THE START VALUE): the large products as SIX bf16
 * MFMAs on three-plane splits (h + m + l carry all 24 significand bits) with fp32 accumulation -- 3e-7 of max|C|, the error
 * class of an fp32 FMA chain (gemm_bf16x6_kernel; ragged M / N take its edge instantiations), fp32 pipe for the small ones.
 * Process-wide; start value 3, or from RSCOTR_GEMM_PREC=fp32|bf16x6.  (Modes 1 / 2 of rounds 1-4, the two-plane bf16
 * product at 4-6e-6, are gone.) */
/* The same product with PRE-SPLIT WEIGHTS.  In y = x W^T and dx = dy W of a Linear layer (torch F.linear behind mmcv's FFN,
 * MultiheadAttention, MultiScaleDeformableAttention, mmdet's WindowMSA / FFN) the B operand is a parameter that changes once
 * per optimizer step; rscotr_gemm_split_weights writes its three bf16 planes once (layout [K/16][npad][3][16], npad = rows
 * rounded up to 256 with zero rows behind; `transposed` entries hold the planes of W^T for dx = dy W, N % 16 == 0), and
 * rscotr_gemm_f32_wplanes multiplies an fp32 row-major A (split on the fly, once per 128 / 256 output columns) with them:
 * same six-term product and error class as precision mode 3, same epilogue arguments as rscotr_gemm_f32 (no row sums, no
 * k scaling: those belong to weight gradients).  table of rscotr_gemm_split_weights: device (n, 8) int64 rows {W, planes, N, K,
 * ldw, npad, first block, transposed}; an entry takes ceil(npad * (reduction / 16) / 256) blocks, reduction = K (or N when
 * transposed, the plane set then has K rows).  Plane buffers hold npad * reduction * 6 bytes. */
int rscotr_gemm_split_weights(const int64_t* table, int n, int total_blocks, void* stream);
int64_t rscotr_gemm_f32_wplanes_workspace(int M, int N, int K);
/* 1 if rscotr_gemm_f32_wplanes is worth calling for the shape (the weight-plane kernel's domain: M >= 4096 rows, K >= 1024,
 * N >= 64); 0: multiply with the fp32 weight (rscotr_gemm_f32).  act_is_gelu: reserved. */
int rscotr_gemm_f32_wplanes_ok(int M, int N, int K, int act_is_gelu);
int rscotr_gemm_f32_wplanes(const float* A, const void* planes, int npad, float* C, int M, int N, int K, int lda, int ldc,
                            const float* bias, int act, const float* aux, float* pre, const float* resid, int accumulate,
                            const float* rowscale, int rows_per_scale, float* out2, float* workspace,
                            int64_t workspace_bytes, void* stream);
int rscotr_gemm_set_precision(int prec);
int rscotr_gemm_get_precision(void);
int64_t rscotr_gemm_f32_workspace(int M, int N, int K);
int rscotr_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb,
                    int ldc, int a_kmajor, int b_kmajor, const float* bias, int act, const float* aux,
                    float* pre, const float* resid, int accumulate, float* rowsum, int rowsum_accumulate,
                    const float* rowscale, int rows_per_scale, const float* kscale, int krows_per_scale,
                    float* out2, float* workspace, int64_t workspace_bytes, void* stream);
/* The same product with the VALUE RANGES of its operands (round 5).  amax_a / amax_b: device words holding the bit pattern of
 * max |x| over (a superset of) each operand — written by rscotr_amax_f32, by the epilogue of the product that made the tensor
 * (amax_out of that call), by the optimizer for parameters.  With both given, the products that precision mode 3 routes to
 * the split kernels run as the FP16 SPLIT PRODUCT "h3" instead: x 2^s = h + l 2^-11 with h = rne_f16(x 2^s),
 * l = rne_f16((x 2^s - h) 2^11), s from the operand's amax so that the scaled amax lies in [2^12, 2^13); three
 * v_mfma_f32_32x32x16_f16 per 16 k (h h into one accumulator set, l h + h l into a second that enters with 2^-11), fp32
 * accumulate.  The two planes carry an element to 2^-24 relative down to 2^-26 of the tensor's amax (2^-48 of amax absolute
 * below): the error class of an fp32 FMA chain, like the six-term bf16 product, at half the MFMA issues and two thirds of
 * the conversion / LDS traffic.  A RANGE WORD is RSCOTR_RANGE_PLANES (32) sub-words at a stride of RSCOTR_RANGE_STRIDE (16384)
 * words — the caller owns one buffer of [32][16384] uint32 and names a word by the address of its sub-word 0 — read as an EXPONENT
 * MAP of 128 bytes (round 6): byte i != 0 <=> the tensor holds an element whose biased fp32 exponent is RSCOTR_RANGE_EXP_LO + i
 * (clamped into the window).  Producers mark with plain one-byte stores of the constant 1 (idempotent: no atomics, any order, any
 * XCD); consumers take the highest byte set: the binade of max |x|, which is all the scale of a split product uses.  A word is
 * zeroed by the caller before its producers run (one memset per iteration); a parameter's persistent word is rewritten whole by the
 * optimizer kernels.  Until round 6 the word was the maximum's bit pattern folded by atomicMax (L2 atomics at the tails of ~800
 * producer launches per co-training round: ~0.3 ms).  Everything else (tiles, k-slices, epilogue, row sums, k scaling, workspace) is
 * rscotr_gemm_f32; null ranges = exactly rscotr_gemm_f32.  rscotr_gemm_set_h3(0 | 1): A/B switch (RSCOTR_GEMM_H3),
 * returns the previous setting.  amax_out (optional): the binade of max |C| of the stored result is marked in this word (the
 * caller zeroes it).  rscotr_amax_f32: marks the binade of max |X| over rows x cols, row stride ld. */
#define RSCOTR_RANGE_EXP_LO 80
#define RSCOTR_RANGE_PLANES 32
#define RSCOTR_RANGE_STRIDE 16384
int rscotr_gemm_f32_r(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb,
                      int ldc, int a_kmajor, int b_kmajor, const float* bias, int act, const float* aux,
                      float* pre, const float* resid, int accumulate, float* rowsum, int rowsum_accumulate,
                      const float* rowscale, int rows_per_scale, const float* kscale, int krows_per_scale,
                      float* out2, float* workspace, int64_t workspace_bytes, const uint32_t* amax_a,
                      const uint32_t* amax_b, uint32_t* amax_out, void* stream);
int rscotr_gemm_set_h3(int on);
/* The ReLU gate of an FFN as ONE BIT per element (round 5).  mmcv FFN.forward is Linear -> ReLU -> Linear (cfg ...potsdam.py:86-93,
 * reached from models/multi/bbox_head/transformer.py:78-131 and seg_head/pixel_decoder.py:134-146); autograd's ReLU backward reads the
 * M x N activation again to gate dH = (dY W2) * [h > 0] — at 10880 x 2048 that is 89 MB fetched in 512-byte row segments at the end
 * of every workgroup of the dX product, which is what bounds that launch.  act 5 (forward: relu, and the words [y > 0] leave through
 * `pre`, M * N / 8 bytes, 8-byte aligned) and act 6 (backward: v *= bit, the words arrive through `aux`) replace act 1 / act 3 on
 * products that run on the interior 128 x 128 split-product tiles with one k-slice — rscotr_gemm_relu_bits_ok answers 1 for those
 * (precision mode 3, M % 128 == N % 128 == 0, K % 32 == 0, row-major A) and rscotr_gemm_f32* refuses the codes anywhere else.  The
 * word layout (uint64 [128 x 128 tile, row-major][thread]) is private to the two kernels; the activation itself is still written
 * (the weight gradient multiplies with it).  With acts 5 / 6: bias only — no pre-activation output, residual, accumulate, row
 * scale, second output. */
int rscotr_gemm_relu_bits_ok(int M, int N, int K, int lda, int ldb, int a_kmajor, int b_kmajor);
/* The B operand of the fp16 split product from PRE-SPLIT PLANES (round 5).  In y = x W^T and dx = dy W the B tile of a
 * workgroup is a weight — it changes once per optimizer step, yet every row tile of every launch converts it again, and the
 * k loop of the 64 x 64 kernel is bound by that conversion (profiles/r5_h3_64_pmc.txt).  rscotr_gemm_split_weights_h3 writes the
 * two fp16 planes of a weight once per step, scaled by the power of two of its range word: layout [reduction / 32][rpad][h | l][32]
 * fp16, rpad = plane rows rounded up to 64 (zero rows behind the end).  table: device (n, 9) int64 rows {W, planes, rows of W,
 * cols of W, ldw, rpad, first block, transposed, address of the parameter's range word}; transposed = 0: the planes of W (for a
 * row-major B: y = x W^T), 1: of W^T (for a k-major B: dx = dy W); reduction % 32 == 0; an entry takes
 * ceil(rpad * (reduction / 32) / 256) blocks, total_blocks = their sum.  rscotr_gemm_f32_rb = rscotr_gemm_f32_r with the plane
 * set of B given as well: where rscotr_gemm_f32_split_route answers 2 the kernel stages B from the planes (bit-identical to
 * the in-kernel split: same planes), everywhere else B itself is used.  The range word must not have changed since the split. */
int rscotr_gemm_split_weights_h3(const int64_t* table, int n, int total_blocks, void* stream);
int rscotr_gemm_f32_rb(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb,
                       int ldc, int a_kmajor, int b_kmajor, const float* bias, int act, const float* aux,
                       float* pre, const float* resid, int accumulate, float* rowsum, int rowsum_accumulate,
                       const float* rowscale, int rows_per_scale, const float* kscale, int krows_per_scale,
                       float* out2, float* workspace, int64_t workspace_bytes, const uint32_t* amax_a,
                       const uint32_t* amax_b, uint32_t* amax_out, const void* b_planes, int b_rpad, void* stream);
/* A two-layer MLP block as ONE launch (round 6, csrc/ffn.hip).  Replaces
 *   - mmcv FFN (two nn.Linear around a ReLU) inside the shared encoder's BaseTransformerLayer —
 *     configs/multi/MTL_slvlcls_swin-t-p4-w7_1x1_resisc&dior&potsdam.py:44-49, reached from
 *     models/multi/seg_head/pixel_decoder.py:134-146 and models/multi/bbox_head/transformer.py:211-221 (C = 256, mode 0 / 1);
 *   - the FFN of mmdet's SwinBlock (Linear - GELU - Linear with DropPath) in stages 1 and 2 of the backbone —
 *     configs/multi/MTL_slvlcls_...potsdam.py:9-25, reached from models/multi/multitask_learner.py:81-85 (C = 96 / 192, mode 2 / 3);
 * and the mirrored pair of each one's backward pass.  mode:
 *     0  hid = relu(X' W1op^T + b1), the gate [hid > 0] written to `bits`        1  hid = (X' W1op^T) * bit read from `bits`
 *     2  Pre = X' W1op^T + b1 stored, hid = gelu(Pre) (erf form)                  3  hid = (X' W1op^T) * gelu'(Pre), Pre read
 *     Y = (hid W2op^T + b2) * yscale[row / rows_per] + resid,   X' = X * xscale[row / rows_per]   (scales / resid optional)
 * hid (M, H) is stored fp32 (the weight gradients read it).  X (M, C) row-major fp32, C in {96, 128, 192, 256, 384}, H % 128 == 0
 * (rscotr_ffn_h3_ok).  Launches with few rows (< 200 row tiles of 32: Swin-T stage 3, the detection decoder's FFN) cut the hidden width
 * into rscotr_ffn_h3_splits(M, C, H) runs, one workgroup per (row tile, run), whose partial Y meet in `workspace` (splits * M * C floats,
 * 16-byte aligned; unused and may be null when splits == 1) and are combined in fixed order by a second launch.  W1f / W2f: FRAGMENT-MAJOR fp16 planes of the two weight operands, W1op (H rows, reduction C) and W2op (C rows,
 * reduction H), written by rscotr_gemm_split_weights_frag — table rows as rscotr_gemm_split_weights_h3 (column 5 unused); plane
 * rows % 16 == 0, reduction % 32 == 0; layout uint4 [row / 16][k / 32][h | l][lane]: the weight operand of a wavefront's
 * 16 x 16 x 32 MFMA is one contiguous 1 KB load; an entry takes rows * reduction / 8 / 256 blocks.  bits:
 * rscotr_ffn_h3_bits_words(M, C, H) uint32 words (layout private to the kernel).  amax_x / amax_w1 / amax_w2 (required), amax_b1
 * (optional): range words of X, of the two weights (the ones the planes were split with) and of b1; amax_hid / amax_y
 * (optional): range words of hid / Y, committed.  The three-term fp16 split product of rscotr_gemm_f32_r throughout (on
 * 16 x 16 x 32 MFMAs: equal to that entry's results at fp32 rounding, not bit for bit); the planes of hid for the second product
 * are scaled from the a-priori bound C max|X| max|W1| + max|b1|. */
int rscotr_ffn_h3_ok(int M, int C, int H);
int rscotr_ffn_h3_splits(int M, int C, int H);
int64_t rscotr_ffn_h3_bits_words(int M, int C, int H);
int rscotr_gemm_split_weights_frag(const int64_t* table, int n, int total_blocks, void* stream);
/* rscotr_ffn_h3 mode 2 with the LayerNorm of a pre-norm block in front (mmdet SwinBlock.forward: x = x + ffn(norm2(x)), reached from
 * models/multi/multitask_learner.py:83 through the backbone of configs/multi/...potsdam.py:9-25): X is the block input; its rows are
 * normalised (two-pass mean / variance, eps inside the root, affine ln_weight / ln_bias) while they are staged, and LayerNorm(X)
 * leaves the launch as ln_out (M, C) with ln_mean / ln_rstd (M) — what rscotr_layernorm_fwd would have written, equal at fp32
 * rounding (the sums run over another lane layout) — for the weight gradient of the first Linear and rscotr_layernorm_bwd.
 * C in {96, 192, 384}.  amax_ln_weight (required) / amax_ln_bias: range words of the affine parameters — the planes of the
 * normalised rows are scaled from sqrt(C) max|weight| + max|bias|; amax_ln_out (optional): range word of ln_out, committed.
 * Everything else as rscotr_ffn_h3 (no bits, no xscale). */
/* ONE Linear on the fused kernel's machinery for the tall, narrow products of Swin stages 1 / 2 (mmdet WindowMSA's qkv / proj Linears
 * and their input gradients, reached through ShiftWindowMSA.forward of the backbone, cfg :9-25): Y (M, N) = (X Wop^T + bias) *
 * yscale[row / rows_per] + resid, X' = X * xscale[row / rows_per]; X (M, K) row-major fp32, K in {96, 128, 192, 256, 288, 384, 576, 768, 1152 (few rows only)},
 * N % 32 == 0 (rscotr_lin_h3_ok).  Wf: fragment-major fp16 planes of Wop (N rows, reduction K; rscotr_gemm_split_weights_frag).  A
 * workgroup stages the planes of 32 rows once and its eight wavefronts take 32 output columns each — memory-bound shapes, where the
 * tiled kernels re-stage the rows per column tile.  The three-term fp16 split product (rscotr_gemm_f32_r's arithmetic on 16 x 16 x 32
 * MFMAs: equal at fp32 rounding).  rscotr_lin_h3_ln: with the LayerNorm in front (K in {96, 192, 384}), as rscotr_ffn_h3_ln. */
int rscotr_lin_h3_ok(int M, int N, int K);
int rscotr_lin_h3(const float* X, int M, int N, int K, const void* Wf, const float* bias, const float* resid, float* Y,
                  const float* xscale, const float* yscale, int rows_per, const uint32_t* amax_x, const uint32_t* amax_w,
                  uint32_t* amax_y, void* stream);
int rscotr_lin_h3_ln(const float* X, int M, int N, int K, const float* ln_weight, const float* ln_bias, float ln_eps, float* ln_out,
                     float* ln_mean, float* ln_rstd, const void* Wf, const float* bias, const float* resid, float* Y,
                     const float* yscale, int rows_per, const uint32_t* amax_ln_weight, const uint32_t* amax_ln_bias,
                     const uint32_t* amax_w, uint32_t* amax_ln_out, uint32_t* amax_y, void* stream);
int rscotr_ffn_h3_ln(const float* X, int M, int C, int H, const float* ln_weight, const float* ln_bias, float ln_eps, float* ln_out,
                     float* ln_mean, float* ln_rstd, const void* W1f, const float* b1, const void* W2f, const float* b2, float* Pre,
                     float* Hid, const float* resid, float* Y, const float* yscale, int rows_per, const uint32_t* amax_ln_weight,
                     const uint32_t* amax_ln_bias, const uint32_t* amax_w1, const uint32_t* amax_w2, const uint32_t* amax_b1,
                     uint32_t* amax_ln_out, uint32_t* amax_hid, uint32_t* amax_y, float* workspace, int64_t workspace_bytes,
                     void* stream);
int rscotr_ffn_h3(const float* X, int M, int C, int H, const void* W1f, const float* b1, const void* W2f, const float* b2,
                  int mode, void* bits, float* Pre, float* Hid, const float* resid, float* Y, const float* xscale,
                  const float* yscale, int rows_per, const uint32_t* amax_x, const uint32_t* amax_w1,
                  const uint32_t* amax_w2, const uint32_t* amax_b1, uint32_t* amax_hid, uint32_t* amax_y, float* workspace,
                  int64_t workspace_bytes, void* stream);
/* > 0 if rscotr_gemm_f32 with these arguments (aligned operands) takes the split-product kernels, i.e. runs as the fp16 split
 * product once both value ranges are supplied: callers ask before they go looking for ranges.  2: the interior pipelined
 * 64 x 64 kernel, which can take a weight operand B from pre-split planes (rscotr_gemm_f32_rb). */
int rscotr_gemm_f32_split_route(int M, int N, int K, int lda, int ldb, int a_kmajor, int b_kmajor, int act, int has_pre,
                                int has_rowscale, int has_kscale, int64_t workspace_bytes);
int rscotr_amax_f32(const float* X, int64_t rows, int cols, int ld, uint32_t* slot, void* stream);
/* rscotr_amax_f32 for n tensors in one launch: table = device (n, 6) int64 rows {X, rows, cols, ld, range word, first block};
 * entry e is worked on by blocks [first_e, first_{e+1}) of the total_blocks workgroups (>= 1 each, ascending). */
int rscotr_amax_group(const int64_t* table, int n, int total_blocks, void* stream);
/* Deferred split-K combine for weight gradients.  rscotr_gemm_f32_dw_slabs = rscotr_gemm_f32(a_kmajor = b_kmajor = 1,
 * accumulate = 1, rowsum_accumulate = 1) WITHOUT its combine launch: the slabs stay in `slab_region` (caller-owned until
 * the flush; rscotr_gemm_f32_workspace() bytes), *splits_out (HOST int) = number of slabs written ([splits][M][N] floats,
 * then [splits][M] row-sum partials), 1 = the problem was not split and C / rowsum already hold the result.
 * rscotr_splitk_flush combines every pending problem of a backward pass in ONE launch (C += sum of slabs, rowsum +=
 * sum of partials, fixed order): table = device (n, 8) int64 rows {slabs, row-sum slabs | 0, C, rowsum | 0, M, N, ldc,
 * splits} (N % 4 == 0, ldc % 4 == 0, 16-byte aligned C / slabs); wgmap = device (nwg, 2) int32 rows {table row, chunk},
 * ceil(max(M*N/4, M) / 256) chunks per row.  Replaces ~450 combine launches per co-training round (one per split
 * weight-gradient contraction of mmcv / torch autograd's Linear backward) by 3. */
int rscotr_gemm_f32_dw_slabs(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                             float* rowsum, const float* kscale, int krows_per_scale, float* slab_region,
                             int64_t slab_bytes, int32_t* splits_out, void* stream);
/* ... with the value ranges of the operands (rscotr_gemm_f32_r): the fp16 split product where mode 3 takes the split kernels */
int rscotr_gemm_f32_dw_slabs_r(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                               float* rowsum, const float* kscale, int krows_per_scale, float* slab_region,
                               int64_t slab_bytes, int32_t* splits_out, const uint32_t* amax_a, const uint32_t* amax_b,
                               void* stream);
/* (bytes: the algorithmic traffic of the launch — slabs read, destinations read and written — stated by the caller who built the
 * device table, for the launch-site profiler; 0 = not stated) */
int rscotr_splitk_flush(const int64_t* table, const int32_t* wgmap, int nwg, double bytes, void* stream);
/* Grouped launch of deferred weight gradients: n problems dW_i = A_i^T B_i (both operands k-major) with small outputs run
 * as ONE launch on tiles x k-slices; every problem leaves `splits` slabs (+ row-sum partials when rs_slabs != 0) for
 * rscotr_splitk_flush.  variant 0: fp32 matrix pipe, 64 x 64 tiles with bounds handling (any problem); variant 2 / 3: bf16x6
 * split product (fp32-accurate, three bf16 planes per operand) on 128 x 128 / 64 x 64 tiles — M, N multiples of the tile edge,
 * K and ksplit_len multiples of 16 / 32, 16-byte aligned operands with lda, ldb multiples of 4 (caller-checked).  table =
 * device (n, 16) int64 rows {A, B, slabs, rs_slabs | 0, kscale | 0, M, N, K, lda, ldb, ksplit_len (a multiple of 16 / 32
 * unless splits == 1), splits, first workgroup of the row's BUNDLE, krows_per_scale, 0, workgroups of the problem = tiles *
 * splits}, tiles = ceil(M / 64) * ceil(N / 64) (variant 2: (M / 128) * (N / 128)), n a multiple of 8: rows come in bundles of
 * 8 (padded with all-zero rows) that occupy 8 * max(workgroups of the bundle's rows) consecutive workgroup ids, row x of a
 * bundle taking the ids = x mod 8 — one problem per XCD, so that its operands enter one L2 once; total_wgs = the sum over
 * the bundles; flops = sum of 2 M N K over the problems (the table is device data: the caller states the launch's algorithmic
 * work for the launch-site profiler, rscotr_prof_*; 0 = not stated).  Replaces ~110 short launches per co-training round (torch autograd's per-Linear weight-gradient GEMMs behind
 * mmcv's FFN / MultiheadAttention / MultiScaleDeformableAttention modules).  variant 6: the bf16x6 product on 128 x 128
 * tiles with edge handling (M, N, K multiples of 4); variant 7 (round 5): the fp16 split product (rscotr_gemm_f32_r) on the same
 * tiles — table column 14 = (index of A's value-range word + 1) << 32 | index of B's + 1, both into `amax_base` (required). */
int rscotr_gemm_dw_group(const int64_t* table, int n, int total_wgs, int variant, double flops,
                         const uint32_t* amax_base, void* stream);

/* nb0 * nb1 independent products of one shape, problem (b0, b1) at element offsets b0*s?0 + b1*s?1 of A, B, C
 * (b0 = image, b1 = head: the per-head slices of (B, L, heads*32) tensors are addressed in place).  No bias /
 * activation; accumulate != 0 adds into C.  ksplits > 1 (row-major A, k-major B, K % ksplits == 0, no
 * accumulate): the reduction is cut into slices that write ksplits slabs shaped like the whole C tensor
 * (c_elems elements each, in `workspace`) and a second kernel sums them into C — for products with few output
 * tiles and thousands of keys.  With rscotr_softmax_mask_fwd / rscotr_softmax_bwd this is
 * torch.nn.MultiheadAttention (mmcv MultiheadAttention, cfg ...potsdam.py:81-85,144-151; reached from
 * models/multi/bbox_head/transformer.py:103-108, models/multi/seg_head/mask2former_head.py:183-192):
 * S = q k^T per head; P = softmax(scale*S + mask) in place (mask bool, True = blocked; mask_mode 0 none,
 * 1 (Lq,Lk) shared, 2 (B,Lq,Lk) per image, 3 (B*heads,Lq,Lk)); O = P v; backward dP <- scale*P*(dP - sum P dP). */
int rscotr_gemm_f32_batched(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb,
                            int ldc, int a_kmajor, int b_kmajor, int nb0, int nb1, int64_t sA0, int64_t sA1,
                            int64_t sB0, int64_t sB1, int64_t sC0, int64_t sC1, int accumulate, int ksplits,
                            float* workspace, int64_t c_elems, void* stream);
int rscotr_softmax_mask_fwd(float* S, const unsigned char* mask, int mask_mode, int B, int heads, int Lq, int Lk,
                            float scale, void* stream);
int rscotr_softmax_bwd(const float* P, float* dP, int64_t rows, int Lk, float scale, void* stream);
/* The same attention core as ONE pass per direction for head dim 32 (csrc/attn_core.hip): the chain
 * rscotr_gemm_f32_batched (q k^T) -> rscotr_softmax_mask_fwd -> rscotr_gemm_f32_batched (P v) of torch.nn.MultiheadAttention
 * (models/multi/bbox_head/transformer.py:103-108, models/multi/seg_head/mask2former_head.py:183-192) without the
 * (B, heads, Lq, Lk) score tensor: out[b, i, h*32 + d] = sum_j softmax_j(scale * q_i . k_j + mask_ij) v_j[d], online softmax over
 * 32 x 32 score tiles held in MFMA accumulators (v_mfma_f32_32x32x2_f32).  q (B, Lq, .) / k, v (B, Lk, .) / out (B, Lq, .) are
 * addressed in place through their row strides ld? (head h = columns h*32 .. h*32+31; strides multiples of 4, pointers
 * 16-byte aligned: q | k may be the column halves of one projected tensor); lse (B, heads, Lq) receives log sum exp of every
 * score row (+inf for a fully blocked row, whose output is 0) and is what backward recomputes the probabilities from.
 * Backward writes dq (B, Lq, .), dk, dv (B, Lk, .) (overwritten) from dout (row stride ldo, like out): a query-side pass
 * (dQ, and D = rowsum(dout * out) into the workspace) then a key-side pass (dK, dV); every sum in a fixed order (no atomics).
 * mask / mask_mode as rscotr_softmax_mask_fwd.  workspace: rscotr_attn_core_workspace(B, heads, Lq, Lk) bytes (backward always;
 * forward only when the key axis is cut into chunks: few query blocks against thousands of keys). */
int64_t rscotr_attn_core_workspace(int B, int heads, int Lq, int Lk);
int rscotr_attn_core_fwd(const float* q, const float* k, const float* v, const unsigned char* mask, int mask_mode, float* out,
                         float* lse, int B, int heads, int Lq, int Lk, int hd, int ldq, int ldk, int ldv, int ldo, float scale,
                         void* workspace, int64_t workspace_bytes, void* stream);
int rscotr_attn_core_bwd(const float* q, const float* k, const float* v, const unsigned char* mask, int mask_mode, const float* out,
                         const float* dout, const float* lse, float* dq, float* dk, float* dv, int B, int heads, int Lq, int Lk,
                         int hd, int ldq, int ldk, int ldv, int ldo, int lddq, int lddk, int lddv, float scale, void* workspace,
                         int64_t workspace_bytes, void* stream);
/* out[n] (+)= sum_m X[m*ld+n]  (bias gradients of the Linears above); two-stage, deterministic;
 * accumulate != 0 adds into out; workspace of rscotr_colsum_f32_workspace(M, N) bytes required. */
int64_t rscotr_colsum_f32_workspace(int M, int N);
int rscotr_colsum_f32(const float* X, float* out, int M, int N, int ld, int accumulate, float* workspace,
                      int64_t workspace_bytes, void* stream);

/* ---- LayerNorm over the last dimension ---------------------------------------------------------
 * Replaces torch.nn.LayerNorm (eps 1e-5) as instantiated by mmdet SwinTransformer / mmcv
 * BaseTransformerLayer / the heads (cfg ...potsdam.py:9-25,34-50,76-98,139-160;
 * models/multi/bbox_head/transformer.py:38-41,151-158; models/multi/seg_head/mask2former_head.py:60-83).
 * x,y,dy,dx (M,C) row-major, C % 4 == 0, C <= 2048; mean/rstd (M) saved by forward (may be NULL
 * in forward when no backward follows).  Backward ACCUMULATES dweight/dbias (caller zeroes them
 * or passes the gradient buffer to add into); dx/dweight/dbias may each be NULL; dx_add (M,C) or NULL is added to
 * dx on the way out (pre-norm blocks: the gradient of the residual branch that forks at the LayerNorm input --
 * mmdet SwinBlock `x = x + attn(norm1(x))`, swin.py of mmdet 2.25.1 -- instead of a separate element-wise add); a workspace of
 * rscotr_layernorm_bwd_workspace(M, C) bytes (16-byte aligned) holds per-workgroup partial sums.
 * amax_out (all entries of the family; NULL = not wanted): value-range word (rscotr_gemm_f32_r) that receives max |y| (and
 * |y2|) of a forward, max |dx| of a backward — the range of the operand the next Linear multiplies with. */
int rscotr_layernorm_fwd(const float* x, const float* weight, const float* bias, float* y, float* mean,
                         float* rstd, int M, int C, float eps, uint32_t* amax_out, void* stream);
/* ---- two-stage proposal selection of the DINO transformer (models/multi/bbox_head/transformer.py:226-241) ---------------
 * topk_idx (B, K) int64 = torch.topk(enc_cls.max(-1)[0], K, dim=1)[1] (descending scores; equal scores: lower index first —
 * torch leaves that order unspecified), topk_score (B, K, C) = gather(enc_cls), topk_unact (B, K, 4) = gather(enc_reg +
 * proposals), topk_anchor = sigmoid(topk_unact); inv (B, N) int32 = rank of token n among the K winners or -1 (for the
 * backward).  enc_cls (B, N, C) = cls_branches[num_layers](output_memory), enc_reg (B, N, 4) = reg_branches[num_layers](
 * output_memory), proposals (B | 1, N, 4) = output_proposals (proposals_batched = 0: one set for all images).  One workgroup
 * per image: K <= 1024, N <= 36864.  Backward: d_cls (B, N, C) / d_reg (B, N, 4) are written completely (zeros for tokens that
 * were not selected): d_cls[b, n] = d_score[b, inv], d_reg[b, n] = d_anchor[b, inv] * a (1 - a); d_score / d_anchor may be NULL
 * (= zero), d_cls / d_reg may be NULL (not wanted). */
int rscotr_det_proposals(const float* enc_cls, const float* enc_reg, const float* proposals, int proposals_batched,
                         int64_t* topk_idx, float* topk_score, float* topk_unact, float* topk_anchor, int32_t* inv, int B, int N,
                         int C, int K, void* stream);
int rscotr_det_proposals_bwd(const float* d_score, const float* d_anchor, const float* topk_anchor, const int32_t* inv,
                             float* d_cls, float* d_reg, int B, int N, int C, int K, void* stream);

/* Classification / box targets of the Hungarian assignment for S prediction sets x B images in one launch
 * (mmdet_detr_head/detr_head.py:475-543 `_get_target_single`, batched): q_for_gt (S, B, G) int32 = the query assigned to
 * ground truth g of image b in set s (rscotr_lsap_dev_f32's output; -1 = padding column), gt_lab (B, G) int64, gt_boxn
 * (B, G, 4) normalised cxcywh.  labels (S, B, Q) int64 = gt_lab of the assigned ground truth or num_classes (background),
 * bbox_targets (S, B, Q, 4) = its box or zeros, bbox_weights (S, B, Q, 4) = 1 or 0: every row written once. */
int rscotr_det_targets(const int32_t* q_for_gt, const int64_t* gt_lab, const float* gt_boxn, int64_t* labels, float* bbox_targets,
                       float* bbox_weights, int S, int B, int Q, int G, int num_classes, void* stream);

/* rscotr_layernorm_fwd with a second output y2 = y + add[row % add_rows] (add (add_rows, C)): mmcv BaseTransformerLayer's
 * `norm` step followed by an attention whose wrapper forms `query + query_pos` (mmcv MultiheadAttention.forward /
 * MultiScaleDeformableAttention.forward as built from cfg ...potsdam.py:34-50,76-98,139-160) — the sum leaves the norm's
 * own launch instead of an element-wise add; y2 carries no gradient of its own (the attention's backward produces
 * d(query) and d(query_pos) from its projections). */
int rscotr_layernorm_fwd_sum(const float* x, const float* weight, const float* bias, float* y, float* mean, float* rstd,
                             const float* add, int add_rows, float* y2, int M, int C, float eps, uint32_t* amax_out,
                             void* stream);
int64_t rscotr_layernorm_bwd_workspace(int M, int C);
int rscotr_layernorm_bwd(const float* dy, const float* x, const float* weight, const float* mean,
                         const float* rstd, float* dx, const float* dx_add, float* dweight, float* dbias, int M, int C,
                         float* workspace, int64_t workspace_bytes, uint32_t* amax_out, void* stream);

/* mmcv PatchMerging's `nn.Unfold(kernel_size=2, stride=2)` (zero "corner" padding for odd H / W) followed by its
 * LayerNorm(4 Cin) (mmdet 2.25.1 SwinTransformer stages' `downsample`, cfg ...potsdam.py:9-25), one launch per direction:
 * x (B, H, W, Cin) token map, y (B * ceil(H/2) * ceil(W/2), 4 Cin) with element c * 4 + kh * 2 + kw of token (i, j) =
 * LN over that row of x[b, 2i + kh, 2j + kw, c]; the unfold is done by the norm's own loads (forward) and stores (dx,
 * (B, H, W, Cin): every position written exactly once); results equal rscotr_layernorm_* on the gathered copy bit for
 * bit.  Cin <= 512.  Backward ADDS dweight / dbias
 * (4 Cin each); fold = 0 leaves the per-workgroup partial rows in `workspace` for rscotr_layernorm_flush (table row
 * {workspace, dweight, dbias, workspace_bytes / (32 Cin), 4 Cin}). */
int rscotr_patch_merge_norm_fwd(const float* x, const float* weight, const float* bias, float* y, float* mean, float* rstd,
                                int B, int H, int W, int Cin, float eps, uint32_t* amax_out, void* stream);
int64_t rscotr_patch_merge_norm_bwd_workspace(int B, int H, int W, int Cin);
int rscotr_patch_merge_norm_bwd(const float* dy, const float* x, const float* weight, const float* mean, const float* rstd,
                                float* dx, float* dweight, float* dbias, int B, int H, int W, int Cin, float* workspace,
                                int64_t workspace_bytes, int fold, uint32_t* amax_out, void* stream);

/* Deferred parameter-gradient fold: rscotr_layernorm_bwd_partials = rscotr_layernorm_bwd without its second launch (the
 * per-workgroup partial rows stay in `part`, rscotr_layernorm_bwd_workspace() bytes, caller-owned until the flush);
 * rscotr_layernorm_flush folds EVERY pending pass of a backward in one launch: table = device (n, 5) int64 rows
 * {partial rows, dweight | 0, dbias | 0, G = partial rows, C}, wgmap = device (nwg, 2) int32 {table row, block of 64 of the
 * 2C columns}.  Two rows with the same destination must go to different launches (the fold is a plain read-add-write). */
int rscotr_layernorm_bwd_partials(const float* dy, const float* x, const float* weight, const float* mean,
                                  const float* rstd, float* dx, const float* dx_add, int M, int C, float* part, int64_t part_bytes,
                                  uint32_t* amax_out, void* stream);
int rscotr_layernorm_flush(const int64_t* table, const int32_t* wgmap, int nwg, void* stream);


/* ---- Swin (shifted-)window attention core -------------------------------------------------------
 * Replaces, between the qkv Linear and the proj Linear, mmdet ShiftWindowMSA + WindowMSA
 * (F.pad -> torch.roll -> window partition -> q k^T/sqrt(32) + relative_position_bias_table[(dy+6)*13+(dx+6)]
 * [+ -100 shift mask] -> softmax -> @v -> window reverse -> roll back -> crop) of the Swin-T backbone built
 * from configs/multi/MTL_slvlcls_...potsdam.py:9-25 and run at models/multi/multitask_learner.py:83.
 *   qkv (B, H*W, 3C): output of the qkv Linear on the UNPADDED tokens, channel = which*C + head*32 + d;
 *   qkv_bias (3C) or NULL: value of q/k/v on zero-padding tokens (the upstream Linear runs on them);
 *   bias_table (169, heads); out / dout (B, H*W, C), channel = head*32 + d.  ws must be 7, C = heads*32.
 * Backward recomputes the probabilities; dqkv (B, H*W, 3C) is fully overwritten; dqkv_bias (3C, pad-token
 * contributions only) and dbias_table (169, heads) are ACCUMULATED (caller zeroes); either may be NULL.  workspace:
 * rscotr_swin_wattn_bwd_workspace() bytes of per-workgroup partial rows of those two gradients, folded in fixed order
 * by a second launch (no global atomics: bit-reproducible).  With BOTH gradient pointers NULL the partial rows simply
 * stay in `workspace` (then caller-owned) and rscotr_swin_wattn_flush folds the rows of many backward passes in one
 * launch.  `out` (may be NULL): the forward output of the same inputs; with it the softmax backward's
 * delta_i = sum_j P_ij dP_ij is taken as dO_i . O_i (a 32-channel dot product) instead of a cross-lane reduction.
 * Default kernels: four wavefronts per (window, head) item, matrix cores for the five 49x49x32 products. */
int rscotr_swin_wattn_fwd(const float* qkv, const float* qkv_bias, const float* bias_table, float* out,
                          int B, int H, int W, int C, int heads, int ws, int shift, uint32_t* amax_out,
                         void* stream);
int64_t rscotr_swin_wattn_bwd_workspace(int B, int H, int W, int C, int heads);
int rscotr_swin_wattn_bwd(const float* qkv, const float* qkv_bias, const float* bias_table, const float* dout,
                          float* dqkv, float* dqkv_bias, float* dbias_table, int B, int H, int W, int C,
                          int heads, int ws, int shift, const float* out, float* workspace,
                          int64_t workspace_bytes, uint32_t* amax_out, void* stream);
/* table: device (n, 16) int64 rows {partial rows, dbias_table | 0, dqkv_bias | 0, heads, C, rows per head (= workspace bytes
 * / (heads * 268 * 4)), running sum of `heads` over the previous rows, 0 ...}; total_heads = sum of heads. */
int rscotr_swin_wattn_flush(const int64_t* table, int n, int total_heads, void* stream);

/* ---- ChannelMapper neck in token layout -----------------------------------------------------------
 * mmdet ChannelMapper (configs/multi/MTL_slvlcls_...potsdam.py:26-33; models/multi/multitask_learner.py:84):
 * the 1x1 convolutions are rscotr_gemm_f32 on the (B*H*W, C_in) token matrix; the 3x3 stride-2 padding-1
 * "extra" convolution is rscotr_gemm_f32 on the matrix gathered by rscotr_im2col3x3s2_tokens (row = output
 * position, column = c*9 + ky*3 + kx, the Conv2d weight's own (C, kH, kW) flattening), its input gradient
 * is rscotr_col2im3x3s2_tokens of the GEMM's dcol; GroupNorm(32, 256) runs on tokens.
 *   x, y, dy, dx (B, L, C) with C in {64, 128, 256}; mean_rstd (B, G, 2) written by forward; proj_ws (B, G, 2)
 *   scratch; dweight/dbias (C) ACCUMULATED (caller zeroes), may be NULL.  Ho = (H+1)/2, Wo = (W+1)/2.
 *   dy_batch_stride (elements, >= L * C, % 4 == 0): dy may be one level's rows of a gradient laid out over the
 *   concatenated levels (B, sum L_l, C) — the encoder input's gradient — read in place.
 *   workspace: rscotr_groupnorm_tokens_workspace() bytes of per-workgroup partial sums, folded in fixed order (no
 *   atomics: the statistics are bit-reproducible). */
int64_t rscotr_groupnorm_tokens_workspace(int B, int L, int C, int G);
int rscotr_groupnorm_tokens_fwd(const float* x, const float* weight, const float* bias, float* y,
                                float* mean_rstd, int B, int L, int C, int G, float eps, float* workspace,
                                int64_t workspace_bytes, void* stream);
int rscotr_groupnorm_tokens_bwd(const float* dy, const float* x, const float* weight, const float* mean_rstd,
                                float* dx, float* dweight, float* dbias, float* proj_ws, int B, int L, int C,
                                int G, int64_t dy_batch_stride, float* workspace, int64_t workspace_bytes, void* stream);
int rscotr_im2col3x3s2_tokens(const float* x, float* col, int B, int H, int W, int C, void* stream);
int rscotr_col2im3x3s2_tokens(const float* dcol, float* dx, int B, int H, int W, int C, void* stream);

/* ---- fused bilinear upsample + cross-entropy + top-1 accuracy (seg loss) -------------------------------
 * Replaces mmseg BaseDecodeHead.losses (resize(bilinear, align_corners=False) -> F.cross_entropy(ignore_index,
 * reduction='none') -> mean over ALL pixels; accuracy over the non-ignored ones) reached from
 * models/multi/seg_head/mask2former_head.py:204, without materialising the (B,C,H,W) upsampled logits.
 *   logit (B,C,h,w); label (B,H,W) int64; lse (B,H,W) per-pixel log-sum-exp saved for backward;
 *   sums[3] = {sum of CE over non-ignored pixels, #correct, #non-ignored} (written; per-workgroup partials in
 *   `workspace` (rscotr_upsample_ce_workspace() bytes) folded in fixed order: bit-reproducible, no atomics);
 *   backward: dlogit (B,C,h,w) = grad_scale[0] * d(sums[0])/d(logit), grad_scale a DEVICE scalar. */
int64_t rscotr_upsample_ce_workspace(void);
int rscotr_upsample_ce_fwd(const float* logit, const int64_t* label, float* lse, float* sums, int B, int C,
                           int h, int w, int H, int W, int ignore_index, float* workspace, int64_t workspace_bytes,
                           void* stream);
int rscotr_upsample_ce_bwd(const float* logit, const int64_t* label, const float* lse, const float* grad_scale,
                           float* dlogit, int B, int C, int h, int w, int H, int W, int ignore_index, void* stream);

/* Query position embedding of the DINO decoder (models/multi/bbox_head/transformer.py:43-76): pos (rows,4) = (x,y,w,h)
 * -> out (rows,512) = [emb(y)|emb(x)|emb(w)|emb(h)], emb(v)[2i] = sin(2 pi v / 10000^(2i/128)), [2i+1] = cos.  No
 * gradient (the reference points are detached). */
int rscotr_sine_embed4(const float* pos, float* out, int64_t rows, void* stream);

/* Level embeddings of the multi-scale token maps: out[b, t, :] = x[b, t, :] + cst[b | 0, t, :] + w[level(t), :] over the
 * concatenated levels (sizes[l] tokens each, host array, sum = N); x and cst may be null, image b of x starts at element
 * b * x_bstride (rows dense), cst is (B, N, C) when cst_batched else (N, C); out is dense.  Replaces the per-level `pos + level_embed[lvl].view(1, 1, -1)` adds + concatenation of
 * models/multi/bbox_head/transformer.py:196-207 (lvl_pos_embed), models/multi/seg_head/pixel_decoder.py:108-118
 * (level_encoding) and models/multi/seg_head/mask2former_head.py:152-156 (level_embed).  C a multiple of 4, L <= 8. */
int rscotr_level_embed_fwd(const float* x, int64_t x_bstride, const float* cst, int cst_batched, const float* w,
                           float* out, const int* sizes, int L, int B, int N, int C, void* stream);
/* Gradient of the embedding rows: dw[l, :] (+)= sum over b and the tokens of level l of g[b, t, :], fixed summation
 * order (bit-reproducible), one launch.  workspace: rscotr_level_embed_bwd_workspace(L, C) bytes; counters: L ints, zero
 * before the first call (the kernel returns them to zero). */
/* Contrastive denoising queries of DINO in slot layout (models/multi/bbox_head/query_denoising.py:104-178: label flip,
 * box jitter of the positive / negative copies, clamp, cxcywh, inverse_sigmoid(eps = 1e-3), label-embedding lookup) in ONE
 * launch.  Slot s copies ground truth slot_src[s] (index into gt_lab (.,) / gt_boxn (., 4), normalised cxcywh); slot_valid /
 * slot_neg (n_slots) floats 0 | 1; u (n_slots, 10) = [label_p, new_label, sign x 4, part x 4]: raw uniforms in [0, 1) when
 * uniform != 0 (new_label = floor(u * num_classes), sign = u >= 0.5), else the reference's own draws (integer-valued).
 * label_thr = label_noise_scale / 2 (<= 0: no flips), box_scale = box_noise_scale (<= 0: no jitter).  Outputs: kl_out
 * (n_slots) the noised labels, q_label (n_slots, C) = embed[kl] (zero rows for invalid slots), q_bbox (n_slots, 4).
 * rscotr_cdn_embed_grad: d(embed)[r] (+)= sum of g[s] over the valid slots with kl[s] == r, in slot order. */
int rscotr_cdn_queries(const int64_t* gt_lab, const float* gt_boxn, const int64_t* slot_src, const float* slot_valid,
                       const float* slot_neg, const float* u, int uniform, const float* embed, float label_thr,
                       float box_scale, int num_classes, int64_t* kl_out, float* q_label, float* q_bbox, int n_slots, int C,
                       void* stream);
int rscotr_cdn_embed_grad(const float* g, const int64_t* kl, const float* slot_valid, float* dw, int rows, int n_slots,
                          int C, int accumulate, void* stream);
/* Classification head (models/multi/cls_head/slvl_cls_head.py:14-23; mmcls GlobalAveragePooling + LabelSmoothLoss 'original'):
 * rscotr_gap_tokens_fwd: out (B, C) = mean over the T tokens of x (B, T, C); _bwd: dx (B, T, C) = g (B, C) / T, dense.
 * rscotr_soft_ce: loss (1) = sum over rows and classes of -t * log_softmax(score) / avg_factor with
 * t = label * (1 - smooth) + smooth / C (label (B, C): one-hot or mixed soft labels), and dscore (B, C) = d loss / d score
 * in the same launch.  B <= 1024. */
int rscotr_gap_tokens_fwd(const float* x, float* out, int B, int T, int C, void* stream);
int rscotr_gap_tokens_bwd(const float* g, float* dx, int B, int T, int C, void* stream);
int rscotr_soft_ce(const float* score, const float* label, float* loss, float* dscore, int B, int C, float smooth,
                   float avg_factor, void* stream);
/* out = p0 + p1 + ... + p(n-1) (n <= 8 dense fp32 arrays of `count` elements, added left to right; out may be p0): the
 * gradients that meet at a tensor with several consumers (torch autograd adds them pairwise, one launch per consumer). */
int rscotr_sum8(const float* p0, const float* p1, const float* p2, const float* p3, const float* p4, const float* p5,
                const float* p6, const float* p7, int n, float* out, int64_t count, void* stream);
/* out = [a | b | c | d]: flat concatenation of up to four fp32 arrays in one launch (packs the rows and biases of Linear
 * layers that read the same operand, e.g. mmcv MultiScaleDeformableAttention's sampling_offsets and attention_weights,
 * so that they run as one product). */
int rscotr_pack4(const float* a, int64_t na, const float* b, int64_t nb, const float* c, int64_t nc, const float* d,
                 int64_t nd, float* out, uint32_t* amax_out, void* stream);
int64_t rscotr_level_embed_bwd_workspace(int L, int C);
int rscotr_level_embed_bwd(const float* g, float* dw, const int* sizes, int L, int B, int N, int C, int accumulate,
                           float* workspace, int* counters, void* stream);

/* Masked-attention mask of the seg decoder (models/multi/seg_head/mask2former_head.py:126-136, :177-178):
 * mask_pred (rows, h, w) -> bilinear resize to (th, tw), align_corners=False -> sigmoid < 0.5 -> rows that are
 * all-True reset to all-False -> out (rows, th*tw) bool (1 byte each, 1 = blocked). */
int rscotr_seg_attn_mask(const float* mask_pred, unsigned char* out, int rows, int h, int w, int th, int tw,
                         void* stream);

/* ---- detection loss arithmetic ---------------------------------------------------------------------------------
 * rscotr_match_cost: mmdet FocalLossCost + BBoxL1Cost(xywh) + IoUCost(giou) (HungarianAssigner.assign reached from
 * models/multi/bbox_head/mmdet_detr_head/detr_head.py:513-515; weights / alpha / gamma / eps from cfg ...potsdam.py:169-174)
 * for S prediction sets and padded ground truth: cls (S,B,Q,C) logits, box (S,B,Q,4) cxcywh normalised, gt_box (B,G,4)
 * xyxy pixels, gt_lab (B,G) int64, factors (B,4) = (w,h,w,h) -> cost (S,B,Q,G).
 * rscotr_focal_sum: mmcv sigmoid_focal_loss summed per set (detr_head.py:384-385, dino_head.py:272-273): pred (S,N,C),
 * target (S,N) int64 in [0,C] (C = background), weight (S,N) or NULL -> sums (S) and dpred (S,N,C) = d sums / d pred.
 * rscotr_box_loss: L1 (cxcywh, normalised; weight (S,B,Q,4)) and GIoU (xyxy * factors, eps; weight = mean of the 4)
 * sums per set (detr_head.py:392-415): sums (2,S) = {l1, giou}; d_l1, d_giou (S,B,Q,4) = their gradients wrt pred. */
/* out = sigmoid(delta + inverse_sigmoid(ref, eps)) — the box refinement step of the DINO decoder / head
 * (bbox_head/transformer.py:112-118, dino_head.py:133-137) — and its backward (d_delta / d_ref may be NULL). */
int rscotr_refine_box_fwd(const float* delta, const float* ref, float* out, int64_t n, float eps, void* stream);
int rscotr_refine_box_bwd(const float* grad_out, const float* out, const float* ref, float* d_delta, float* d_ref,
                          int64_t n, float eps, void* stream);
int rscotr_match_cost(const float* cls, const float* box, const float* gt_box, const int64_t* gt_lab,
                      const float* factors, float* cost, int S, int B, int Q, int C, int G, float w_cls, float w_l1,
                      float w_iou, float alpha, float gamma, float eps, void* stream);
int rscotr_focal_sum(const float* pred, const int64_t* target, const float* weight, float* sums, float* dpred, int S,
                     int N, int C, float gamma, float alpha, void* stream);
int rscotr_box_loss(const float* pred, const float* target, const float* weight, const float* factors, float* sums,
                    float* d_l1, float* d_giou, int S, int B, int Q, float eps, void* stream);

/* ---- Hungarian matching (host, fp64) ---------------------------------------------------------
 * Replaces scipy.optimize.linear_sum_assignment as called by mmdet HungarianAssigner.assign,
 * reached from models/multi/bbox_head/mmdet_detr_head/detr_head.py:513-515.  Pure CPU,
 * thread-safe.  rscotr_lsap_f64 returns the number of assignments (>=0) or a negative error;
 * row_ind is ascending, as SciPy returns it.  The batch entry solves n independent problems with
 * fp32 costs (problem k: rows[k] x cols[k] row-major at cost+offsets[k]; results at
 * row_ind/col_ind + out_offsets[k], min(rows,cols) entries) and returns 0 or an error. */
int rscotr_lsap_f64(const double* cost, int nr, int nc, int64_t* row_ind, int64_t* col_ind);
/* Device-side solver for the matcher's own problem shape (same algorithm, fp64, same tie-breaks, one
 * wavefront per problem; no host round trip, capturable in a hipGraph): P problems, cost[p] (Q, ld) fp32
 * row-major = queries x ground truths with the first gcount[p] (<= min(ld, Q)) columns real (DEVICE int32);
 * q_for_gt[p][i] (P, ld) int32 = query assigned to ground truth i, -1 for i >= gcount[p].
 * Q <= 1024, ld <= 256. */
int rscotr_lsap_dev_f32(const float* cost, const int32_t* gcount, int P, int Q, int ld, int32_t* q_for_gt,
                        void* stream);
int rscotr_lsap_batch_f32(const float* cost, const int64_t* offsets, const int* rows, const int* cols,
                          int n, const int64_t* out_offsets, int64_t* row_ind, int64_t* col_ind);

/* ---- fused global-norm clip + AdamW over a flat arena ----------------------------------------
 * Replaces mmcv OptimizerHook.after_train_iter (clip_grad_norm_(max_norm=0.1) + AdamW.step(),
 * registered at mtl/apis/train.py:66-83; per-parameter groups from mtl/utils/optimizer.py:40-55;
 * cfg configs/multi/MTL_slvlcls_...potsdam.py:203-213).  param/grad/exp_avg/exp_avg_sq are flat fp32
 * arenas with identical offsets; chunk k covers chunk_len[k] (multiple of 4) elements at
 * chunk_off[k] (multiple of 4) inside segment chunk_seg[k]; seg_dyn holds 8 floats per segment:
 * {lr, weight_decay, 1/bias_correction1, 1/sqrt(bias_correction2), live(0/1), 0, 0, 0}.
 * rscotr_grad_sumsq WRITES the squared L2 norm of the live segments' gradients to sumsq[0]; sumsq is a
 * buffer of 1 + 1024 floats (sumsq[1..] = per-workgroup partials, folded in fixed order: bit-reproducible,
 * no atomics); rscotr_adamw_clip_step applies coef = min(1, max_norm/(sqrt(*sumsq)+1e-6)) (skipped
 * when max_norm <= 0) and the decoupled-weight-decay Adam update of torch 1.11 to live segments. */
int rscotr_grad_sumsq(const float* grad, const int32_t* chunk_seg, const int64_t* chunk_off,
                      const int32_t* chunk_len, const float* seg_dyn, int nchunks, float* sumsq,
                      void* stream);
int rscotr_adamw_clip_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                           const int32_t* chunk_seg, const int64_t* chunk_off, const int32_t* chunk_len,
                           const float* seg_dyn, int nchunks, const float* sumsq, float max_norm,
                           float beta1, float beta2, float eps, void* stream);
/* The same step keeping the VALUE RANGES of the parameters (round 5; consumed by rscotr_gemm_f32_r as amax_b of a weight
 * operand): seg_amax[segment] = the range word (exponent map, rscotr_gemm_f32_r; all RSCOTR_RANGE_PLANES sub-words of it: seg_amax is
 * a row of the caller's range buffer) of max |w| over the segment, refreshed for every live segment by the update
 * itself (each chunk's wavefronts store their maxima into the scratch chunk_amax — 4 * nchunks uint32, 16-byte aligned — and
 * a second small launch folds them per segment, found by bisection of the ascending chunk_seg: no atomics, deterministic);
 * other segments keep their word.  chunk_amax is required when seg_amax is given.
 * rscotr_param_amax computes all nseg words from the arena (construction, checkpoint load, restore). */
int rscotr_adamw_clip_step_r(float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                             const int32_t* chunk_seg, const int64_t* chunk_off, const int32_t* chunk_len,
                             const float* seg_dyn, int nchunks, const float* sumsq, float max_norm,
                             float beta1, float beta2, float eps, uint32_t* seg_amax, int nseg, uint32_t* chunk_amax,
                             void* stream);
int rscotr_param_amax(const float* param, const int32_t* chunk_seg, const int64_t* chunk_off,
                      const int32_t* chunk_len, int nchunks, uint32_t* seg_amax, int nseg, uint32_t* chunk_amax,
                      void* stream);

/* ---- gradient exchange on RCCL, called directly ------------------------------------------------------------------------
 * Replaces the bucket all-reduces of torch DDP / c10d ProcessGroupNCCL behind the reference's MMDistributedDataParallel
 * (mtl/apis/train.py:37-46; launched from tools/train.py:173-182 `init_dist`) for the DATA path of the exchange: a bucket of
 * the flat gradient arena is averaged in place by one ncclAllReduce(ncclFloat32, ncclAvg) on the stream the caller names —
 * ordered against backward by the caller's own events (fork / join, also inside a hipGraph capture), with no watchdog thread
 * polling events of a capturing stream (c10d's aborted the overlapped exchange in 2 of 8 runs: DESIGN.md section 6).  The
 * control traffic (the unique id, plan hashes, agreements) stays with whatever process group the job has.  RCCL is resolved
 * at run time from the instance already in the process (torch's) or librccl.so on the loader path.
 * rscotr_comm_unique_id: 128 bytes, made by ONE rank and handed to all; rscotr_comm_init: collective, each rank with its
 * device current; rscotr_comm_allreduce_avg: buf[0 .. count) = mean over the ranks, in place, on `stream` (capturable). */
int rscotr_comm_available(void);
int rscotr_comm_unique_id(void* id128);
int rscotr_comm_init(const void* id128, int rank, int nranks, void** comm_out);
int rscotr_comm_allreduce_avg(void* comm, float* buf, int64_t count, void* stream);
int rscotr_comm_destroy(void* comm);

/* ---- device-side input pipeline --------------------------------------------------------
 * One launch turns a batch of ragged decoded uint8 images into the collated network input: crop window ->
 * horizontal flip -> BGR->RGB + (x - mean) / std -> zero pad to (Hout, Wout) -> (B, 3, Hout, Wout) float32.
 * Replaces the per-sample pipeline tail + collate of the reference's dataset configs (mmcv imflip / imnormalize /
 * impad, ImageToTensor / DefaultFormatBundle): configs/_base_/cls/resisc_swin_224.py:14,36-38,
 * configs/_base_/det/dior.py:15-19, configs/_base_/seg/potsdam_IRRG_all.py:12-19.
 * src: device byte buffer holding every sample's HWC (3 bytes per pixel) image; meta: device (B, 10) int64 rows
 * {byte offset, H, W, row stride in bytes, crop x0, crop y0, crop w, crop h, flip (0/1), reserved}; the crop window must
 * lie inside the image and crop w / h <= Wout / Hout (caller-checked: the entry cannot read device memory).
 * mean3 / std3: HOST pointers to 3 floats each, in output-channel order (RGB when to_rgb).
 * rscotr_seg_label_prep_u8: the same geometry for a 1-byte-per-pixel label map -> (B, 1, Hout, Wout) int64, padded
 * with pad_val (seg_pad_val); reduce_zero_label applies mmseg LoadAnnotations' 0 -> 255, l -> l - 1. */
int rscotr_img_prep_u8(const uint8_t* src, const int64_t* meta, float* out, int B, int Hout, int Wout,
                       const float* mean3, const float* std3, int to_rgb, void* stream);
int rscotr_seg_label_prep_u8(const uint8_t* src, const int64_t* meta, int64_t* out, int B, int Hout, int Wout,
                             int reduce_zero_label, int pad_val, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RSCOTR_H */
