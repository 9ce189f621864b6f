"""Multi-scale deformable attention: the sampling op (mmcv MultiScaleDeformableAttnFunction contract), its element-wise
prologue, and the whole attention block as one autograd node."""
import torch
from torch.autograd import Function

from .core import _WS, _Prof, _chk, _f32c, _ptr, _stream, lib
from .matmul import RANGE_OUT, _linear_param_grad, gemm
from .ranges import RANGES
from .state import STATE

# ------------------------------------------------------------------------------------------
# multi-scale deformable attention sampling (mmcv MultiScaleDeformableAttnFunction contract)
# ------------------------------------------------------------------------------------------
# 'tiled' (default): per-tile scan + LDS sort + register accumulation in sample order, bit-reproducible, 3 launches;
# 'sorted': counting sort by destination token + pull, bit-reproducible, 8 launches; 'scatter': atomic accumulation
# (order-dependent).  Tests run all three.
# STATE.msda_bwd selects (RSCOTR_MSDA_BWD).

# Host copies of the level-shape tensors.  The mmcv contract keeps spatial_shapes on the device; the tile-accumulation
# backward sizes its launches from the shapes, so whoever builds the device tensor registers its host twin
# (layers.LevelGeometry, which also keeps the tensor alive).  An entry is trusted only while the registered tensor object is
# ALIVE (its address cannot have been handed to another tensor) and UNMODIFIED (same autograd version counter: an in-place
# edit bumps it); anything else — an unregistered tensor, a dead owner whose address the caching allocator reused, an edited
# one — is read back from the device on every call (a sync: eager callers only; inside a hipGraph capture it is an error).
_MSDA_HOST_SHAPES = {}  # data_ptr -> (weakref to the registered tensor, its version, int64 (L, 2) array)


def msda_register_shapes(spatial_shapes, shapes):
    import weakref
    import numpy as np
    _MSDA_HOST_SHAPES[spatial_shapes.data_ptr()] = (
        weakref.ref(spatial_shapes), spatial_shapes._version,
        np.ascontiguousarray(np.asarray(shapes, dtype=np.int64).reshape(-1, 2)))
    if len(_MSDA_HOST_SHAPES) > 256:  # (dead owners: forget them)
        for k in [k for k, e in _MSDA_HOST_SHAPES.items() if e[0]() is None]:
            del _MSDA_HOST_SHAPES[k]


def _msda_host_shapes(spatial_shapes):
    e = _MSDA_HOST_SHAPES.get(spatial_shapes.data_ptr())
    if e is not None:
        owner = e[0]()
        if (owner is not None and owner._version == e[1] and spatial_shapes._version == e[1]
                and owner.shape == spatial_shapes.shape):
            return e[2]
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError('rscotr_msda_bwd (tiled): the level shapes of this spatial_shapes tensor are not known on the host; '
                           'register them (ops.msda_register_shapes) before the iteration is captured')
    import numpy as np
    return np.ascontiguousarray(spatial_shapes.detach().cpu().numpy().astype(np.int64).reshape(-1, 2))


def _msda_fwd_raw(value, spatial_shapes, level_start_index, loc, attn):
    B, Nk, H, D = value.shape
    _, Nq, _, L, P, _ = loc.shape
    out = torch.empty((B, Nq, H * D), dtype=torch.float32, device=value.device)
    # algorithmic bytes: read value + loc + attn, write out
    nbytes = 4 * B * (Nk * H * D + Nq * H * L * P * 3 + Nq * H * D)
    with _Prof('msda_fwd', nbytes):
        lib.call('rscotr_msda_fwd', value.data_ptr(), spatial_shapes.data_ptr(),
                 level_start_index.data_ptr(), loc.data_ptr(), attn.data_ptr(), out.data_ptr(),
                 B, Nk, Nq, H, D, L, P, _stream())
    return out


def _msda_bwd_raw(value, spatial_shapes, level_start_index, loc, attn, grad_out):
    B, Nk, H, D = value.shape
    _, Nq, _, L, P, _ = loc.shape
    hs, hs_ptr, nws = None, 0, 0
    if STATE.msda_bwd != 'scatter':
        nws = lib.rscotr_msda_bwd_workspace(B, Nk, Nq, H, L, P)
        if STATE.msda_bwd == 'tiled':
            hs = _msda_host_shapes(spatial_shapes)
            hs_ptr = hs.ctypes.data
            nws = max(nws, lib.rscotr_msda_bwd_tiled_workspace(hs_ptr, B, Nk, Nq, H, D, L, P))
    ws = _WS.get(nws, value.device) if nws else None
    grad_value = torch.zeros_like(value) if ws is None else torch.empty_like(value)
    grad_loc = torch.empty_like(loc)
    grad_attn = torch.empty_like(attn)
    # algorithmic bytes: read value, RMW grad_value, read loc/attn/grad_out, write grad_loc/attn
    nbytes = 4 * B * (3 * Nk * H * D + Nq * H * L * P * 3 + Nq * H * D + Nq * H * L * P * 3)
    slot = RANGES.out_slot(value.device)  # (max |grad_value|: the value projection's backward multiplies with it)
    with _Prof('msda_bwd', nbytes):
        lib.call('rscotr_msda_bwd', value.data_ptr(), spatial_shapes.data_ptr(),
                 level_start_index.data_ptr(), loc.data_ptr(), attn.data_ptr(), grad_out.data_ptr(),
                 grad_value.data_ptr(), grad_loc.data_ptr(), grad_attn.data_ptr(),
                 B, Nk, Nq, H, D, L, P, hs_ptr, 0 if ws is None else ws.data_ptr(), nws, slot, _stream())
    return RANGES.tag(grad_value, slot), grad_loc, grad_attn


class _MSDA(Function):
    @staticmethod
    def forward(ctx, value, spatial_shapes, level_start_index, loc, attn):
        value, loc, attn = _f32c(value), _f32c(loc), _f32c(attn)
        spatial_shapes = spatial_shapes.contiguous()
        level_start_index = level_start_index.contiguous()
        _chk(value, spatial_shapes, level_start_index, loc, attn)
        assert spatial_shapes.dtype == torch.int64 and level_start_index.dtype == torch.int64
        out = _msda_fwd_raw(value, spatial_shapes, level_start_index, loc, attn)
        ctx.save_for_backward(value, spatial_shapes, level_start_index, loc, attn)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        value, spatial_shapes, level_start_index, loc, attn = ctx.saved_tensors
        grad_value, grad_loc, grad_attn = _msda_bwd_raw(value, spatial_shapes, level_start_index, loc, attn, _f32c(grad_out))
        return grad_value, None, None, grad_loc, grad_attn


def msda(value, spatial_shapes, level_start_index, loc, attn):
    """value (B,Nk,H,D), spatial_shapes (L,2) int64 [device], level_start_index (L,) int64
    [device], loc (B,Nq,H,L,P,2), attn (B,Nq,H,L,P) -> (B,Nq,H*D).  Same argument meaning as
    mmcv's MultiScaleDeformableAttnFunction.apply (im2col_step is not needed)."""
    return _MSDA.apply(value, spatial_shapes, level_start_index, loc, attn)


def _msda_fused_ok(Nk, H, D, L, P):
    """the prologue rides the sampling kernel's staging (rscotr_msda_fwd_prep): 16 samples per (query, head)"""
    return STATE.msda_fused and bool(lib.rscotr_msda_fused_ok(Nk, H, D, L, P))


def _msda_fwd_prep_raw(value, spatial_shapes, level_start_index, off, logit, ref, norm, L, P, ld_off, ld_logit):
    """-> (loc, attn, out): _msda_prep_fwd_raw + _msda_fwd_raw in one launch"""
    B, Nk, H, D = value.shape
    Nq = ref.shape[1]
    loc = torch.empty((B, Nq, H, L, P, 2), dtype=torch.float32, device=value.device)
    attn = torch.empty((B, Nq, H, L, P), dtype=torch.float32, device=value.device)
    out = torch.empty((B, Nq, H * D), dtype=torch.float32, device=value.device)
    nbytes = 4 * B * (Nk * H * D + Nq * H * L * P * 6 + Nq * H * D)
    with _Prof('msda_fwd', nbytes):
        lib.call('rscotr_msda_fwd_prep', value.data_ptr(), spatial_shapes.data_ptr(), level_start_index.data_ptr(),
                 off.data_ptr(), logit.data_ptr(), ld_off, ld_logit, ref.data_ptr(), _ptr(norm), ref.shape[-1], ref.shape[-2],
                 loc.data_ptr(), attn.data_ptr(), out.data_ptr(), B, Nk, Nq, H, D, L, P, _stream())
    return loc, attn, out


def _msda_prep_fwd_raw(off, logit, ref, norm, B, Nq, H, L, P, ld_off=None, ld_logit=None):
    """off / logit: dense (B*Nq, H*L*P*2) / (B*Nq, H*L*P), or column blocks of one wider row (ld_* = its row stride);
    ref (B,Nq,L|1,2|4)."""
    refdim = ref.shape[-1]
    loc = torch.empty((B, Nq, H, L, P, 2), dtype=torch.float32, device=off.device)
    attn = torch.empty((B, Nq, H, L, P), dtype=torch.float32, device=off.device)
    lib.call('rscotr_msda_prep_fwd', off.data_ptr(), logit.data_ptr(), ref.data_ptr(), _ptr(norm), loc.data_ptr(),
             attn.data_ptr(), B, Nq, H, L, P, refdim, ld_off or H * L * P * 2, ld_logit or H * L * P, ref.shape[-2], _stream())
    return loc, attn


def _msda_prep_bwd_raw(gloc, gattn, attn, ref, norm, B, Nq, H, L, P, packed=False):
    """-> (grad_off, grad_logit); packed: the two as column blocks [0, 2n) and [2n, 3n) of ONE (B*Nq, 3n) tensor (n = H*L*P),
    returned as (that tensor, None)."""
    n = H * L * P
    if packed:
        both = torch.empty((B * Nq, 3 * n), dtype=torch.float32, device=attn.device)
        goff, glogit, ldo, ldl = both, both[:, 2 * n:], 3 * n, 3 * n
    else:
        goff = torch.empty((B, Nq, H, L * P * 2), dtype=torch.float32, device=attn.device)
        glogit = torch.empty((B, Nq, H, L * P), dtype=torch.float32, device=attn.device)
        ldo, ldl = 2 * n, n
    slot = RANGES.out_slot(attn.device)  # (one word bounds both gradients: they are the operands of the projections' backward)
    lib.call('rscotr_msda_prep_bwd', gloc.data_ptr(), gattn.data_ptr(), attn.data_ptr(), ref.data_ptr(), _ptr(norm),
             goff.data_ptr(), glogit.data_ptr(), B, Nq, H, L, P, ref.shape[-1], ldo, ldl, ref.shape[-2], slot, _stream())
    return (RANGES.tag(both, slot), None) if packed else (RANGES.tag(goff, slot), RANGES.tag(glogit, slot))


class _MSDAPrep(Function):
    @staticmethod
    def forward(ctx, off, logit, ref, norm, L, P):
        off, logit, ref = _f32c(off), _f32c(logit), _f32c(ref.detach())
        _chk(off, logit, ref, norm)
        B, Nq, H = logit.shape[:3]
        loc, attn = _msda_prep_fwd_raw(off, logit, ref, norm, B, Nq, H, L, P)
        ctx.save_for_backward(attn, ref, norm)
        ctx.geom = (B, Nq, H, L, P)
        return loc, attn

    @staticmethod
    def backward(ctx, gloc, gattn):
        attn, ref, norm = ctx.saved_tensors
        goff, glogit = _msda_prep_bwd_raw(_f32c(gloc), _f32c(gattn), attn, ref, norm, *ctx.geom)
        return goff, glogit, None, None, None, None


def msda_prep(off, logit, reference_points, offset_norm, L, P):
    """off (B,Nq,H*L*P*2) raw sampling offsets, logit (B,Nq,H,L*P) raw attention logits, reference_points
    (B,Nq,L,2|4) (no gradient), offset_norm (L,2) = (W_l,H_l) -> (loc (B,Nq,H,L,P,2), attn (B,Nq,H,L,P))."""
    assert not reference_points.requires_grad, 'reference points are detached on this path'
    B, Nq, H = logit.shape[:3]
    return _MSDAPrep.apply(off.view(B, Nq, H, L * P * 2), logit, reference_points, offset_norm, L, P)




class _MSDAAttn(Function):
    """mmcv MultiScaleDeformableAttention.forward as ONE autograd node: q = x + query_pos, value / offset /
    weight projections, softmax + location arithmetic, the sampling kernel, output projection + identity — and a backward
    that MERGES the gradients meeting at the block input inside GEMM epilogues instead of leaving them to autograd's
    element-wise adds: d(x) = d(offsets) W_off + d(weights) W_aw [+ dy when x is the identity] [+ d(value) W_v when x is
    the value]; d(query_pos) is the same product without the merged terms (second output of the epilogue).
    args: x (B,Nq,C), q_pos (B,Nq,C)|None, value_in (B,Nk,C)|None (= x), identity Tensor|None (may be x), key_padding_mask
    (B,Nk) bool|None, reference_points (no gradient), spatial_shapes, level_start_index, offset_norm, heads, L, P, then
    W/b of sampling_offsets, attention_weights, value_proj, output_proj."""

    @staticmethod
    def forward(ctx, x, q_pos, value_in, identity, kpm, ref, spatial_shapes, lsi, norm, heads, L, P,
                w_off, b_off, w_aw, b_aw, w_v, b_v, w_o, b_o, q_sum=None):
        B, Nq, C = x.shape
        H, D = heads, C // heads
        M = B * Nq
        x2 = RANGES.carry(x, _f32c(x).reshape(M, C))
        # q_sum: x + q_pos already formed by the producer of x (ops.layer_norm_sum): data only, no gradient of its own
        q2 = x2 if q_pos is None else (_f32c(torch.add(x, q_pos)).reshape(M, C) if q_sum is None else
                                       RANGES.carry(q_sum, _f32c(q_sum).reshape(M, C)))
        v_is_x = value_in is None or value_in is x
        id_is_x = identity is x
        val2 = x2 if v_is_x else RANGES.carry(value_in, _f32c(value_in).reshape(-1, C))
        Mk = val2.shape[0]
        Nk = Mk // B
        ws = [w if w.is_contiguous() else w.contiguous() for w in (w_off, w_aw, w_v, w_o)]
        ref = _f32c(ref.detach())
        spatial_shapes, lsi = spatial_shapes.contiguous(), lsi.contiguous()
        _chk(x2, q2, val2, ref, spatial_shapes, lsi, norm)
        v = gemm(val2, ws[2], Mk, C, C, C, C, 0, 0, bias=b_v)
        if kpm is not None:
            v.view(B, Nk, C).masked_fill_(kpm[..., None], 0.0)
        n_off, n_aw = H * L * P * 2, H * L * P
        # sampling_offsets | attention_weights as ONE product over the packed rows of the two weights (one small packing
        # launch instead of a second GEMM on the same operand; backward: one d(query) product over K = 3 n)
        packed = STATE.msda_packed and b_off is not None and b_aw is not None and n_off % 4 == 0 and C % 4 == 0
        if packed:
            n3 = n_off + n_aw
            wb = torch.empty(n3 * C + n3, dtype=torch.float32, device=x2.device)
            wslot = RANGES.out_slot(x2.device)  # (max |.| of the packed weights and biases: the range of the B operand below)
            lib.call('rscotr_pack4', ws[0].data_ptr(), n_off * C, ws[1].data_ptr(), n_aw * C, b_off.data_ptr(), n_off,
                     b_aw.data_ptr(), n_aw, wb.data_ptr(), wslot, _stream())
            w_cat = RANGES.tag(wb[:n3 * C].view(n3, C), wslot)
            both = gemm(q2, w_cat, M, n3, C, C, C, 0, 0, bias=wb[n3 * C:], range_out=RANGE_OUT.want(False))  # (read by the prologue)
            off, logit, ldo, ldl = both, both.view(-1)[n_off:], n3, n3
        else:
            w_cat = None
            off = gemm(q2, ws[0], M, n_off, C, C, C, 0, 0, bias=b_off)
            logit = gemm(q2, ws[1], M, n_aw, C, C, C, 0, 0, bias=b_aw)
            ldo, ldl = n_off, n_aw
        if _msda_fused_ok(Nk, H, D, L, P):  # softmax + location arithmetic by the threads that stage the samples
            loc, attn, out = _msda_fwd_prep_raw(v.view(B, Nk, H, D), spatial_shapes, lsi, off, logit, ref, norm, L, P, ldo, ldl)
        else:
            loc, attn = _msda_prep_fwd_raw(off, logit, ref, norm, B, Nq, H, L, P, ld_off=ldo, ld_logit=ldl)
            out = _msda_fwd_raw(v.view(B, Nk, H, D), spatial_shapes, lsi, loc, attn)
        id2 = x2 if id_is_x else (None if identity is None else _f32c(identity).reshape(M, C))
        # (every output element is a convex combination of value entries — softmax weights x bilinear weights, zeros outside
        # the maps: max |out| <= max |v|, so the value's range word serves the output projection's operand too)
        y = gemm(RANGES.carry(v, out.view(M, C)), ws[3], M, C, C, C, C, 0, 0, bias=b_o, resid=id2, range_out=RANGE_OUT.want(id2 is None))
        ctx.save_for_backward(q2, val2, v, loc, attn, ref, norm, out, spatial_shapes, lsi, *ws)
        ctx.slots = (RANGES.saved(q2), RANGES.saved(val2))
        ctx.kpm = kpm
        ctx.w_cat = w_cat  # (a temporary of this node: not an autograd-tracked tensor)
        ctx.params = (w_off, b_off, w_aw, b_aw, w_v, b_v, w_o, b_o)  # handles for the gradient sink
        ctx.geom = (B, Nq, Nk, C, H, D, L, P)
        ctx.flags = (v_is_x, id_is_x, q_pos is not None, identity is not None)
        ctx.shapes = (x.shape, None if q_pos is None else q_pos.shape, None if value_in is None else value_in.shape,
                      None if identity is None else identity.shape)
        return RANGES.carry(y, y.view(B, Nq, C))

    @staticmethod
    def backward(ctx, dy):
        q2, val2, v, loc, attn, ref, norm, out, spatial_shapes, lsi, w_off, w_aw, w_v, w_o = ctx.saved_tensors
        p_off, pb_off, p_aw, pb_aw, p_v, pb_v, p_o, pb_o = ctx.params
        B, Nq, Nk, C, H, D, L, P = ctx.geom
        v_is_x, id_is_x, has_pos, has_id = ctx.flags
        need = ctx.needs_input_grad
        M, Mk = B * Nq, B * Nk
        n_off, n_aw = H * L * P * 2, H * L * P
        g = RANGES.carry(dy, _f32c(dy).reshape(M, C))
        RANGES.restore(q2, ctx.slots[0])  # (void after a begin() since the forward: ADVICE r5)
        RANGES.restore(val2, ctx.slots[1])
        sinks = []
        # output projection
        gw_o, gb_o, s1, s2 = _linear_param_grad(g, out.view(M, C), C, C, M, p_o, pb_o, 0, need[18], pb_o is not None and need[19])
        sinks += [s1, s2]
        d_out = gemm(g, w_o, M, C, C, C, C, 0, 1, range_out=RANGE_OUT.want(False))  # (read by the sampling kernel's backward)
        # sampling kernel and the location / softmax arithmetic
        gv, gloc, gattn = _msda_bwd_raw(v.view(B, Nk, H, D), spatial_shapes, lsi, loc, attn, d_out.view(B, Nq, C))
        w_cat = ctx.w_cat
        packed = w_cat is not None
        n3 = n_off + n_aw
        goff, glogit = _msda_prep_bwd_raw(gloc, gattn, attn, ref, norm, B, Nq, H, L, P, packed=packed)
        gv = RANGES.carry(gv, gv.view(Mk, C))
        if packed:
            both = goff                             # (M, 3 n): [d(offsets) | d(logits)]
            goff, glogit, ldg = both.view(-1), both.view(-1)[n_off:], n3   # (flat aliases: column blocks with row stride 3 n)
            RANGES.carry(both, goff)
            RANGES.carry(both, glogit)
        else:
            goff, glogit, ldg = goff.view(M, n_off), glogit.view(M, n_aw), None
        if ctx.kpm is not None:
            gv.view(B, Nk, C).masked_fill_(ctx.kpm[..., None], 0.0)
        gw_off, gb_off, s1, s2 = _linear_param_grad(goff, q2, n_off, C, M, p_off, pb_off, 0, need[12], pb_off is not None and need[13], lda=ldg)
        sinks += [s1, s2]
        gw_aw, gb_aw, s1, s2 = _linear_param_grad(glogit, q2, n_aw, C, M, p_aw, pb_aw, 0, need[14], pb_aw is not None and need[15], lda=ldg)
        sinks += [s1, s2]
        gw_v, gb_v, s1, s2 = _linear_param_grad(gv, val2, C, C, Mk, p_v, pb_v, 0, need[16], pb_v is not None and need[17])
        sinks += [s1, s2]
        for sk_ in sinks:
            if sk_ is not None:
                STATE.grad_sink.grad_written(sk_[0])
        # input gradients, merged in the epilogues (they go to a norm's backward or a merge: no range words — RANGE_OUT)
        gin = lambda *a_, **k_: gemm(*a_, range_out=RANGE_OUT.want(False), **k_)
        want_pos = has_pos and need[1]
        want_x = need[0]
        want_val = (not v_is_x) and need[2]
        d_x = d_pos = d_val = None
        merge_id = id_is_x and want_x
        if want_x or want_pos:
            two = want_pos and want_x and (merge_id or v_is_x)
            res = g if (merge_id and (two or not want_pos)) else None
            if two:
                d_x = torch.empty((M, C), dtype=torch.float32, device=g.device)  # pure = d(query_pos); d_x = pure (+ dy) (+ d(value) below)
            if packed:
                pure = gin(both, w_cat, M, C, n3, n3, C, 0, 1, out2=d_x if two else None, resid=res)
            else:
                pure = gin(goff, w_off, M, C, n_off, n_off, C, 0, 1)
                gin(glogit, w_aw, M, C, n_aw, n_aw, C, 0, 1, out=pure, accumulate=True, out2=d_x if two else None, resid=res)
            if two:
                d_pos = pure
            else:
                d_x = pure if want_x else None
                d_pos = pure if want_pos else None
            if v_is_x and want_x:
                gin(gv, w_v, Mk, C, C, C, C, 0, 1, out=d_x, accumulate=True)
        elif v_is_x and want_x:
            d_x = gin(gv, w_v, Mk, C, C, C, C, 0, 1, resid=g if merge_id else None)
        if want_val:
            d_val = gin(gv, w_v, Mk, C, C, C, C, 0, 1).view(ctx.shapes[2])
        d_id = g.view(ctx.shapes[3]) if (has_id and not merge_id and need[3]) else None
        if id_is_x and not merge_id:
            d_id = None
        return (None if d_x is None else RANGES.carry(d_x, d_x.view(ctx.shapes[0])),
                None if d_pos is None else RANGES.carry(d_pos, d_pos.view(ctx.shapes[1])),
                d_val, d_id, None, None, None, None, None, None, None, None,
                gw_off, gb_off, gw_aw, gb_aw, gw_v, gb_v, gw_o, gb_o, None)


def msda_attention(x, q_pos, value, identity, key_padding_mask, reference_points, spatial_shapes, level_start_index,
                   offset_norm, heads, L, P, w_off, b_off, w_aw, b_aw, w_v, b_v, w_o, b_o, q_sum=None):
    """The whole of mmcv MultiScaleDeformableAttention.forward on batch-first tensors (see _MSDAAttn); `value` None or x
    itself = self-attention over the token map (encoder), `identity` None = no residual, x = the usual one."""
    assert not reference_points.requires_grad, 'reference points are detached on this path'
    if reference_points.shape[-1] not in (2, 4):
        raise ValueError(f'Last dim of reference_points must be 2 or 4, got {reference_points.shape[-1]}')
    if q_pos is not None and q_pos.shape != x.shape:
        q_pos = q_pos.expand_as(x)
    if q_sum is not None:
        assert q_pos is not None and q_sum.shape == x.shape
        q_sum = RANGES.carry(q_sum, q_sum.detach())
    return _MSDAAttn.apply(x, q_pos, value, identity, key_padding_mask, reference_points, spatial_shapes,
                           level_start_index, offset_norm, heads, L, P, w_off, b_off, w_aw, b_aw, w_v, b_v, w_o, b_o, q_sum)

