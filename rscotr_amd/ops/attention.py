"""Dense attention: the Swin window attention node, torch.nn.MultiheadAttention as one autograd node (`mha`), the mask
logits of the seg decoder and its attention-mask kernel."""
import torch
from torch.autograd import Function

from .core import _WS, _Prof, _chk, _f32c, _off_path, _ptr, _sink, _stream, lib
from .matmul import DEFER, RANGE_OUT, _linear_param_grad, colsum, gemm, gemm_batched, linear
from .ranges import RANGES
from .state import STATE

class _SwinWindowAttn(Function):
    """softmax(q k^T/sqrt(32) + rel-pos bias [+ shift mask]) v over 7x7 (shifted) windows, with the
    pad / roll / partition of mmdet ShiftWindowMSA as index arithmetic (rscotr_swin_wattn_*)."""

    @staticmethod
    def forward(ctx, qkv, qkv_b, table, H, W, heads, ws, shift):
        ctx.table_param, ctx.qkvb_param = table, qkv_b  # handles for the gradient sink
        qkv, table = _f32c(qkv), _f32c(table)
        qkv_b = None if qkv_b is None else _f32c(qkv_b)
        _chk(qkv, qkv_b, table)
        B, L, C3 = qkv.shape
        C = C3 // 3
        out = torch.empty((B, L, C), dtype=torch.float32, device=qkv.device)
        with _Prof('swin_wattn_fwd', 4 * B * L * 4 * C):
            slot = RANGES.out_slot(qkv.device)  # (max |out|: the proj Linear multiplies with it)
            lib.call('rscotr_swin_wattn_fwd', qkv.data_ptr(), _ptr(qkv_b), table.data_ptr(), out.data_ptr(),
                     B, H, W, C, heads, ws, shift, slot, _stream())
        ctx.save_for_backward(qkv, qkv_b, table, out)  # (out: the proj Linear keeps it alive anyway)
        ctx.geom = (B, H, W, C, heads, ws, shift)
        return RANGES.tag(out, slot)

    @staticmethod
    def backward(ctx, dout):
        qkv, qkv_b, table, out = ctx.saved_tensors
        B, H, W, C, heads, ws, shift = ctx.geom
        dout = _f32c(dout)
        dqkv = torch.empty_like(qkv)
        slot = RANGES.out_slot(qkv.device)  # (max |dqkv|: the qkv Linear's backward multiplies with it)
        # the kernel ACCUMULATES the bias-table and pad-token (qkv-bias) gradients: with the gradient sink they go
        # straight into the arena (no zero-filled temporaries, no accumulate-adds afterwards)
        skt = _sink(ctx.table_param) if ctx.needs_input_grad[2] else None
        skb = _sink(ctx.qkvb_param) if (qkv_b is not None and ctx.needs_input_grad[1]) else None
        dtable = None if skt is not None else torch.zeros_like(table)
        dqkv_b = None if (skb is not None or qkv_b is None) else torch.zeros_like(qkv_b)
        dt_ptr = skt[1].data_ptr() if skt is not None else dtable.data_ptr()
        db_ptr = skb[1].data_ptr() if skb is not None else _ptr(dqkv_b)
        nws = lib.rscotr_swin_wattn_bwd_workspace(B, H, W, C, heads)
        if (skt is not None and (skb is not None or qkv_b is None) and DEFER.enabled and STATE.side is None
                and STATE.profile is None):
            # arena-direct: the fold of the partial rows joins the end-of-pass flush (one launch for all 12 blocks)
            part = DEFER.reserve(nws, qkv.device)
            lib.call('rscotr_swin_wattn_bwd', qkv.data_ptr(), _ptr(qkv_b), table.data_ptr(), dout.data_ptr(),
                     dqkv.data_ptr(), 0, 0, B, H, W, C, heads, ws, shift, out.data_ptr(), part, nws, slot, _stream())
            DEFER.wattn_entries.append((part, dt_ptr, db_ptr, heads, C, nws // (heads * 268 * 4)))
        else:
            with _Prof('swin_wattn_bwd', 4 * B * H * W * 8 * C):
                lib.call('rscotr_swin_wattn_bwd', qkv.data_ptr(), _ptr(qkv_b), table.data_ptr(), dout.data_ptr(),
                         dqkv.data_ptr(), db_ptr, dt_ptr, B, H, W, C, heads, ws, shift, out.data_ptr(),
                         _WS.get(nws, qkv.device).data_ptr(), nws, slot, _stream())
        for sk in (skt, skb):
            if sk is not None:
                STATE.grad_sink.grad_written(sk[0])
        return RANGES.tag(dqkv, slot), dqkv_b, dtable, None, None, None, None, None


def swin_window_attention(x, hw, qkv_w, qkv_b, bias_table, rel_index, proj_w, proj_b, heads, ws, shift,
                          identity=None, out_scale=None):
    """mmdet ShiftWindowMSA + WindowMSA on (B, H*W, C) tokens: qkv GEMM on the real
    tokens, fused window-attention kernel (pad / shift / partition / bias / mask / softmax / PV /
    reverse by index arithmetic), proj GEMM.  `rel_index` is unused: the kernel uses the closed form
    (dy+6)*13 + (dx+6) of the buffer."""
    H, W = hw
    qkv = linear(x, qkv_w, qkv_b, range_out=False)  # (read by the window-attention kernel, which writes the word of ITS output)
    o = _SwinWindowAttn.apply(qkv, qkv_b, bias_table, H, W, heads, ws, shift)
    return linear(o, proj_w, proj_b, resid=identity, out_scale=out_scale)  # x + s_b * proj(...): one epilogue


MASK_NONE, MASK_SHARED, MASK_PER_IMAGE, MASK_PER_HEAD = 0, 1, 2, 3


class _MHA(Function):
    """torch.nn.MultiheadAttention (batch-first) + the positional adds and the identity add of mmcv's wrapper, forward and
    backward, entirely on the C ABI: in-proj GEMMs (bias fused), per-head q k^T and P v on the batched GEMM with the
    (B, L, heads*hd) tensors addressed in place (no head transposes), masked softmax / its backward in
    place, out-proj GEMM with bias + identity fused; backward = the transposed contractions, parameter
    gradients accumulated straight into the arena (packed in_proj rows addressed as sub-blocks), and the gradients that
    meet at the block inputs merged inside GEMM epilogues (second epilogue output / accumulate) instead of by autograd's
    element-wise adds.
    args: x (B,Lq,C) query content, q_pos | None, kx (B,Lk,C) key content | None (= x: self-attention), k_pos | None (the
    SAME object as q_pos in self-attention = one q|k projection), vx value content | None (= the key content), then the
    packed parameters, identity (Tensor | None, may be x), heads, mask, mask_mode."""

    @staticmethod
    def forward(ctx, x, q_pos, kx, k_pos, vx, in_w, in_b, out_w, out_b, identity, heads, mask, mask_mode, q_sum=None,
                k_sum=None):
        B, Lq, C = x.shape
        hd = C // heads
        dev = x.device
        self_attn = kx is None
        x2 = _f32c(x).reshape(B * Lq, C)
        # q_sum / k_sum: x + q_pos / kx + k_pos already formed by a producer (ops.layer_norm_sum, a per-level constant):
        # plain data, no gradient of their own — d(x), d(q_pos), ... come from the projections below either way
        if q_pos is None:
            q2 = x2
        else:
            q2 = _f32c(torch.add(x, q_pos) if q_sum is None else q_sum).reshape(B * Lq, C)
        # self-attention with one positional embedding for both sides (query + pos feeds q and k): the q and k projections
        # are one GEMM over the first 2C rows of the packed in_proj weight; q / k are then the column halves of one
        # (B*L, 2C) tensor, addressed in place by the batched products (row stride 2C, element offset C for k)
        fused = self_attn and (k_pos is q_pos)
        if self_attn:
            kx2 = x2
            k2 = q2 if fused else (x2 if k_pos is None else _f32c(torch.add(x, k_pos)).reshape(B * Lq, C))
        else:
            kx2 = _f32c(kx).reshape(-1, C)
            k2 = kx2 if k_pos is None else _f32c(torch.add(kx, k_pos) if k_sum is None else k_sum).reshape(-1, C)
        Lk = k2.shape[0] // B
        v_is_kx = vx is None or vx is (x if self_attn else kx)
        v2 = kx2 if v_is_kx else _f32c(vx).reshape(B * Lk, C)
        id_is_x = identity is x
        in_w, in_b = in_w.contiguous(), in_b.contiguous()
        ldq = 2 * C if fused else C
        core = STATE.attn_core and hd == 32
        nr = RANGE_OUT.want(not core)  # (q, k, v are read by the attention core, not by a product; the chain of batched products measures)
        if fused:
            q = k = gemm(q2, in_w[:2 * C], B * Lq, 2 * C, C, C, C, 0, 0, bias=in_b[:2 * C], range_out=nr)
        else:
            q = gemm(q2, in_w[:C], B * Lq, C, C, C, C, 0, 0, bias=in_b[:C], range_out=nr)
            k = gemm(k2, in_w[C:2 * C], B * Lk, C, C, C, C, 0, 0, bias=in_b[C:2 * C], range_out=nr)
        v = gemm(v2, in_w[2 * C:], B * Lk, C, C, C, C, 0, 0, bias=in_b[2 * C:], range_out=nr)
        if mask is not None:
            mask = mask.contiguous()
            assert mask.dtype == torch.bool and mask.is_cuda
        mode = int(mask_mode) if mask is not None else 0
        o = torch.empty((B * Lq, C), dtype=torch.float32, device=dev)
        if core:
            # one pass over the keys, scores in MFMA accumulators; P below = the log-sum-exp of every score row (what backward
            # recomputes the probabilities from) instead of the (B, heads, Lq, Lk) probabilities
            P = torch.empty((B, heads, Lq), dtype=torch.float32, device=dev)
            nws = lib.rscotr_attn_core_workspace(B, heads, Lq, Lk)
            lib.call('rscotr_attn_core_fwd', q.data_ptr(), k.data_ptr() + (4 * C if fused else 0), v.data_ptr(), _ptr(mask), mode,
                     o.data_ptr(), P.data_ptr(), B, heads, Lq, Lk, hd, ldq, ldq, C, C, float(hd ** -0.5),
                     _WS.get(nws, dev).data_ptr(), nws, _stream())
        else:
            P = torch.empty((B, heads, Lq, Lk), dtype=torch.float32, device=dev)
            sq, sk, sp = (Lq * C, hd), (Lk * C, hd), (heads * Lq * Lk, Lq * Lk)
            sqp, skp = (Lq * ldq, hd), (Lk * ldq, hd)  # strides of the projected q / k
            gemm_batched(q, k, P, Lq, Lk, hd, ldq, ldq, Lk, 0, 0, B, heads, sqp, skp, sp, offB=C if fused else 0)
            lib.call('rscotr_softmax_mask_fwd', P.data_ptr(), _ptr(mask), mode, B, heads, Lq, Lk, float(hd ** -0.5), _stream())
            gemm_batched(P, v, o, Lq, hd, Lk, Lk, C, C, 0, 1, B, heads, sp, sk, sq, ksplit=True)
        ctx.core, ctx.mask_t, ctx.mask_mode = core, mask, mode
        id2 = x2 if id_is_x else (None if identity is None else _f32c(identity).reshape(B * Lq, C))
        y = gemm(o, out_w, B * Lq, C, C, C, C, 0, 0, bias=out_b, resid=id2, range_out=RANGE_OUT.want(id2 is None))
        ctx.save_for_backward(q2, k2, v2, q, k, v, P, o, in_w, out_w)
        ctx.params = (in_w, in_b, out_w, out_b)  # handles for the gradient sink
        ctx.geom = (B, Lq, Lk, C, heads, hd)
        ctx.flags = (fused, self_attn, v_is_kx, id_is_x, q_pos is not None, k_pos is not None, identity is not None)
        ctx.shapes = (x.shape, None if q_pos is None else q_pos.shape, None if kx is None else kx.shape,
                      None if k_pos is None else k_pos.shape, None if vx is None else vx.shape,
                      None if identity is None else identity.shape)
        return y.view(B, Lq, C)

    @staticmethod
    def backward(ctx, dy):
        q2, k2, v2, q, k, v, P, o, in_w, out_w = ctx.saved_tensors
        p_in_w, p_in_b, p_out_w, p_out_b = ctx.params
        B, Lq, Lk, C, heads, hd = ctx.geom
        fused, self_attn, v_is_kx, id_is_x, has_qpos, has_kpos, has_id = ctx.flags
        dev = dy.device
        g = _f32c(dy).reshape(B * Lq, C)
        need = ctx.needs_input_grad  # x, q_pos, kx, k_pos, vx, in_w, in_b, out_w, out_b, identity
        sq, sk, sp = (Lq * C, hd), (Lk * C, hd), (heads * Lq * Lk, Lq * Lk)

        # out projection
        gw_o, gb_o, skw_o, skb_o = _linear_param_grad(g, o, C, C, B * Lq, p_out_w, p_out_b, 0, need[7], need[8])
        do = gemm(g, out_w, B * Lq, C, C, C, C, 0, 1, range_out=RANGE_OUT.want(not ctx.core))
        # attention core
        dv = torch.empty((B * Lk, C), dtype=torch.float32, device=dev)
        ldq = 2 * C if fused else C
        if fused:  # dq | dk as the column halves of one (B*L, 2C) tensor, like q | k
            dq = dk = torch.empty((B * Lq, 2 * C), dtype=torch.float32, device=dev)
        else:
            dq = torch.empty((B * Lq, C), dtype=torch.float32, device=dev)
            dk = torch.empty((B * Lk, C), dtype=torch.float32, device=dev)
        koff = C if fused else 0
        if ctx.core:  # P = lse: probabilities recomputed tile by tile; a query-side pass (dQ) and a key-side pass (dK, dV)
            nws = lib.rscotr_attn_core_workspace(B, heads, Lq, Lk)
            lib.call('rscotr_attn_core_bwd', q.data_ptr(), k.data_ptr() + 4 * koff, v.data_ptr(), _ptr(ctx.mask_t), ctx.mask_mode,
                     o.data_ptr(), do.data_ptr(), P.data_ptr(), dq.data_ptr(), dk.data_ptr() + 4 * koff, dv.data_ptr(), B, heads, Lq,
                     Lk, hd, ldq, ldq, C, C, ldq, ldq, C, float(hd ** -0.5), _WS.get(nws, dev).data_ptr(), nws, _stream())
        else:
            gemm_batched(P, do, dv, Lk, hd, Lq, Lk, C, C, 1, 1, B, heads, sp, sq, sk)                 # dV = P^T dO
            dP = torch.empty_like(P)
            gemm_batched(do, v, dP, Lq, Lk, hd, C, C, Lk, 0, 0, B, heads, sq, sk, sp)                  # dP = dO V^T
            lib.call('rscotr_softmax_bwd', P.data_ptr(), dP.data_ptr(), B * heads * Lq, Lk, float(hd ** -0.5), _stream())
            sqp, skp = (Lq * ldq, hd), (Lk * ldq, hd)
            # dQ first: its key-split combine sums whole-tensor slabs (when fused that sweeps the dk half too, with
            # whatever the workspace held) and the dK product below then writes the dk half
            gemm_batched(dP, k, dq, Lq, hd, Lk, Lk, ldq, ldq, 0, 1, B, heads, sp, skp, sqp, offB=koff, ksplit=True)  # dQ = dS K
            gemm_batched(dP, q, dk, Lk, hd, Lq, Lk, ldq, ldq, 1, 1, B, heads, sp, sqp, skp, offC=koff)  # dK = dS^T Q
        # in projections (packed (3C, C) weight / (3C) bias: three row blocks; q and k as one block when fused)
        want_w, want_b = need[5], need[6]
        sink_w = _sink(p_in_w) if want_w else None
        sink_b = _sink(p_in_b) if want_b else None
        gw_in = None if (not want_w or sink_w is not None) else torch.empty((3 * C, C), dtype=torch.float32, device=dev)
        gb_in = None if (not want_b or sink_b is not None) else torch.empty(3 * C, dtype=torch.float32, device=dev)
        blocks = ((dq, q2, B * Lq, 0, 2 * C), (dv, v2, B * Lk, 2 * C, C)) if fused else \
            ((dq, q2, B * Lq, 0, C), (dk, k2, B * Lk, C, C), (dv, v2, B * Lk, 2 * C, C))
        for dproj, x2, M_, r0, R in blocks:
            rs = None if not want_b else (sink_b[1][r0:r0 + R] if sink_b is not None else gb_in[r0:r0 + R])
            if want_w:
                out_w_blk = sink_w[1][r0:r0 + R] if sink_w is not None else gw_in[r0:r0 + R]
                call = lambda dproj=dproj, x2=x2, M_=M_, o=out_w_blk, rs=rs, R=R: gemm(
                    dproj, x2, R, C, M_, R, C, 1, 1, out=o, accumulate=sink_w is not None, rowsum=rs,
                    rowsum_accumulate=sink_b is not None)
                if sink_w is not None and (sink_b is not None or not want_b):
                    _off_path(call, dproj, x2)
                else:
                    call()
            elif want_b:
                colsum(dproj, M_, R, out=rs, accumulate=sink_b is not None)
        for sk_ in (skw_o, skb_o, sink_w, sink_b):
            if sk_ is not None:
                STATE.grad_sink.grad_written(sk_[0])

        # (the input gradients go to a norm's backward or a merge, not straight into a product: no range words — RANGE_OUT)
        gin = lambda *a_, **k_: gemm(*a_, range_out=RANGE_OUT.want(False), **k_)
        # ---- input gradients: what meets at x (and at the key content) is merged in the epilogues -----------------
        Mq, Mk = B * Lq, B * Lk
        w_q, w_k, w_v, w_qk = in_w[:C], in_w[C:2 * C], in_w[2 * C:], in_w[:2 * C]
        want_x, want_qpos = need[0], has_qpos and need[1]
        merge_id = id_is_x and want_x
        d_x = d_qpos = d_kx = d_kpos = d_vx = None
        if self_attn:
            want_kpos = has_kpos and need[3] and not fused   # (fused: k_pos is q_pos, one gradient)
            v_to_x = v_is_kx and want_x
            if fused:
                # d(q side) + d(k side) in one product over K = 2C (both reach x and the shared positional embedding)
                if want_qpos and want_x and (merge_id or v_to_x):
                    d_x = torch.empty((Mq, C), dtype=torch.float32, device=dev)
                    d_qpos = gin(dq, w_qk, Mq, C, 2 * C, 2 * C, C, 0, 1, out2=d_x, resid=g if merge_id else None)
                elif want_x or want_qpos:
                    pure = gin(dq, w_qk, Mq, C, 2 * C, 2 * C, C, 0, 1, resid=g if (merge_id and not want_qpos) else None)
                    d_x = pure if want_x else None
                    d_qpos = pure if want_qpos else None
            else:
                dq_in = gin(dq, w_q, Mq, C, C, C, C, 0, 1) if (want_x or want_qpos) else None
                dk_in = gin(dk, w_k, Mk, C, C, C, C, 0, 1) if (want_x or want_kpos) else None
                d_qpos = dq_in if want_qpos else None
                d_kpos = dk_in if want_kpos else None
                if want_x:  # (rare on this path: distinct positional embeddings for the two sides)
                    d_x = dq_in + dk_in
                    if merge_id:
                        d_x = d_x + g
            if v_to_x:
                if d_x is None:
                    d_x = gin(dv, w_v, Mk, C, C, C, C, 0, 1, resid=g if merge_id else None)
                elif d_x is d_qpos or d_x is d_kpos:  # shared with a positional gradient: must not be modified
                    d_x = gin(dv, w_v, Mk, C, C, C, C, 0, 1, resid=d_x)
                else:
                    gin(dv, w_v, Mk, C, C, C, C, 0, 1, out=d_x, accumulate=True)
            elif not v_is_kx and need[4]:
                d_vx = gin(dv, w_v, Mk, C, C, C, C, 0, 1)
        else:
            want_kx, want_kpos = need[2], has_kpos and need[3]
            if want_x or want_qpos:
                if want_qpos and merge_id:
                    d_x = torch.empty((Mq, C), dtype=torch.float32, device=dev)
                    d_qpos = gin(dq, w_q, Mq, C, C, C, C, 0, 1, out2=d_x, resid=g)
                else:
                    pure = gin(dq, w_q, Mq, C, C, C, C, 0, 1, resid=g if merge_id else None)
                    d_x = pure if want_x else None
                    d_qpos = pure if want_qpos else None
            v_to_kx = v_is_kx and want_kx
            dv_in = gin(dv, w_v, Mk, C, C, C, C, 0, 1) if (v_to_kx or (not v_is_kx and need[4])) else None
            if want_kx or want_kpos:
                if want_kpos and v_to_kx:
                    d_kx = torch.empty((Mk, C), dtype=torch.float32, device=dev)
                    d_kpos = gin(dk, w_k, Mk, C, C, C, C, 0, 1, out2=d_kx, resid=dv_in)
                elif v_to_kx:
                    d_kx = gin(dk, w_k, Mk, C, C, C, C, 0, 1, out=dv_in, accumulate=True)
                else:
                    pure = gin(dk, w_k, Mk, C, C, C, C, 0, 1)
                    d_kx = pure if want_kx else None
                    d_kpos = pure if want_kpos else None
            elif v_to_kx:
                d_kx = dv_in
            if not v_is_kx and need[4]:
                d_vx = dv_in
        d_id = g if (has_id and not id_is_x and need[9]) else None
        sh = ctx.shapes

        def shaped(t, i):
            return None if t is None else t.view(sh[i])
        return (shaped(d_x, 0), shaped(d_qpos, 1), shaped(d_kx, 2), shaped(d_kpos, 3), shaped(d_vx, 4),
                gw_in, gb_in, gw_o, gb_o, shaped(d_id, 5), None, None, None, None, None)


def mha(x, kx, vx, in_w, in_b, out_w, out_b, heads, attn_mask=None, identity=None, mask_mode=None, q_pos=None, k_pos=None,
        q_sum=None, k_sum=None):
    """torch.nn.MultiheadAttention semantics on batch-first tensors (+ the positional adds and the `identity` residual of
    mmcv's wrapper): query = x + q_pos (B,Lq,C), key = kx + k_pos, value = vx (B,Lk,C); kx None or x itself =
    self-attention (k_pos None then means q_pos), vx None = the key content; attn_mask bool, True = blocked: (Lq,Lk)
    shared, (B,Lq,Lk) per image (mask_mode=MASK_PER_IMAGE) or (B*heads,Lq,Lk).  q_sum / k_sum: x + q_pos / kx + k_pos where a
    producer already formed them (values only, detached)."""
    if attn_mask is not None and mask_mode is None:
        if attn_mask.dim() == 2:
            mask_mode = MASK_SHARED
        else:
            mask_mode = MASK_PER_IMAGE if attn_mask.shape[0] == x.shape[0] and heads > 1 else MASK_PER_HEAD
    same_pos = k_pos is q_pos
    if q_pos is not None and q_pos.shape != x.shape:
        q_pos = q_pos.expand_as(x)
    if kx is x:
        if vx is kx:
            vx = None
        kx = None
        if k_pos is None:
            same_pos = True
    elif vx is kx:
        vx = None
    if same_pos and kx is None:
        k_pos = q_pos
    elif k_pos is not None and k_pos.shape != (x if kx is None else kx).shape:
        k_pos = k_pos.expand_as(x if kx is None else kx)
    if q_sum is not None:
        assert q_pos is not None and q_sum.shape == x.shape
        q_sum = RANGES.carry(q_sum, q_sum.detach())
    if k_sum is not None:
        assert kx is not None and k_pos is not None and k_sum.shape == kx.shape
        k_sum = RANGES.carry(k_sum, k_sum.detach())
    return _MHA.apply(x, q_pos, kx, k_pos, vx, in_w, in_b, out_w, out_b, identity, heads, attn_mask, mask_mode or 0, q_sum,
                      k_sum)


class _MaskLogits(Function):
    """mask_pred[b, q, p] = sum_d e[b, q, d] * mf[b, p, d]  (torch.einsum('bqd,bdhw->bqhw') of
    mask2former_head.py:117 with the mask features kept in token layout (B, h*w, C)): per-image products on the
    batched MFMA GEMM, both gradients likewise."""

    @staticmethod
    def forward(ctx, e, mf):
        e, mf = _f32c(e), _f32c(mf)
        _chk(e, mf)
        B, Q, D = e.shape
        P = mf.shape[1]
        out = torch.empty((B, Q, P), dtype=torch.float32, device=e.device)
        gemm_batched(e, mf, out, Q, P, D, D, D, P, 0, 0, B, 1, (Q * D, 0), (P * D, 0), (Q * P, 0))
        ctx.save_for_backward(e, mf)
        return out

    @staticmethod
    def backward(ctx, g):
        e, mf = ctx.saved_tensors
        B, Q, D = e.shape
        P = mf.shape[1]
        g = _f32c(g)
        de = dmf = None
        if ctx.needs_input_grad[0]:
            de = torch.empty_like(e)
            gemm_batched(g, mf, de, Q, D, P, P, D, D, 0, 1, B, 1, (Q * P, 0), (P * D, 0), (Q * D, 0), ksplit=True)
        if ctx.needs_input_grad[1]:
            dmf = torch.empty_like(mf)
            gemm_batched(g, e, dmf, P, D, Q, P, D, D, 1, 1, B, 1, (Q * P, 0), (Q * D, 0), (P * D, 0))
        return de, dmf


def mask_logits(e, mask_tokens):
    """e (B,Q,C) query embeddings, mask_tokens (B,h*w,C) mask features in token layout -> (B,Q,h*w)."""
    return _MaskLogits.apply(e, mask_tokens)


def seg_attn_mask(mask_pred, target_size, heads):
    """mask2former_head.py:126-136 + :177-178: bilinear resize to the next level, sigmoid < 0.5,
    rows that are all-True reset to all-False -> bool (B, Q, h*w), one kernel (rscotr_seg_attn_mask).  The
    reference tiles it over the heads ((B*heads, Q, h*w)); the attention kernel indexes the per-image mask
    for every head instead."""
    mp = _f32c(mask_pred.detach())
    _chk(mp)
    B, Q, h, w = mp.shape
    th, tw = int(target_size[0]), int(target_size[1])
    out = torch.empty((B, Q, th * tw), dtype=torch.bool, device=mp.device)
    lib.call('rscotr_seg_attn_mask', mp.data_ptr(), out.data_ptr(), B * Q, h, w, th, tw, _stream())
    return out

