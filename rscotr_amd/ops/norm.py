"""LayerNorm (plain and forked for pre-norm residual blocks) and GroupNorm on token layout."""
import torch
from torch.autograd import Function

from .core import _WS, _Prof, _chk, _f32c, _ptr, _sink, _stream, lib
from .matmul import DEFER
from .ranges import RANGES
from .state import STATE

LN_EPS = 1e-5


class LazyNorm:
    """A LayerNorm whose forward launch has been left to its ONLY consumer (layer_norm_fork(lazy=True) -> ops.mlp): the fused
    two-Linear kernel normalises the rows while it stages them (rscotr_ffn_h3_ln) and fills `y` / `stats` on the way; a consumer
    that cannot take that route calls run() first.  The autograd node of the norm is the ordinary one: it saved x, the weight and
    the `stats` tensor, which is full by the time backward runs."""

    def __init__(self, x2, w, b, eps, y, stats, slot):
        self.x2, self.w, self.b, self.eps, self.y, self.stats, self.slot = x2, w, b, eps, y, stats, slot
        self.done = False

    def run(self):
        if not self.done:
            M, C = self.x2.shape
            with _Prof('layernorm_fwd', 8 * M * C):
                lib.call('rscotr_layernorm_fwd', self.x2.data_ptr(), _ptr(self.w), _ptr(self.b), self.y.data_ptr(),
                         self.stats[0].data_ptr(), self.stats[1].data_ptr(), M, C, float(self.eps), self.slot, _stream())
            self.done = True


def materialize(t):
    """Run the launch a lazy norm output `t` is still waiting for (no-op for every other tensor)."""
    lz = getattr(t, '_lazy_ln', None)
    if lz is not None:
        lz.run()
    return t


class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, w, b, eps, lazy=False):
        C = x.shape[-1]
        x2 = _f32c(x).reshape(-1, C)
        M = x2.shape[0]
        _chk(x2, w, b)
        y = torch.empty_like(x2)
        stats = torch.empty((2, M), dtype=torch.float32, device=x2.device)
        slot = RANGES.out_slot(x2.device)
        lz = LazyNorm(x2, w, b, eps, y, stats, slot)
        if not lazy:
            lz.run()
        ctx.save_for_backward(x2, w, stats)
        ctx.has_b = b is not None
        ctx.bias = b
        out = RANGES.tag(y.view(x.shape), slot)
        if lazy:
            out._lazy_ln = lz
        return out

    @staticmethod
    def backward(ctx, dy, dres=None):
        return _LayerNorm._backward(ctx, dy, dres)

    @staticmethod
    def _backward(ctx, dy, dres):
        x2, w, stats = ctx.saved_tensors
        merge = getattr(ctx, 'merge', None)   # (B, H, W, Cin): patch-merging rows gathered by the kernel (_PatchMergeNorm)
        if merge is None:
            M, C = x2.shape
        else:
            M, C = stats.shape[1], 4 * merge[3]
            assert dres is None
        g = _f32c(dy).reshape(M, C)
        r = None if dres is None else _f32c(dres).reshape(M, C)  # residual-branch gradient, added inside the kernel
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        slot = RANGES.out_slot(x2.device) if dx is not None else 0  # (the range of dx: the next backward multiplies with it)
        skw, skb = _sink(w), _sink(ctx.bias)
        direct = skw is not None and (skb is not None or not ctx.has_b)
        dwb = None if direct else torch.zeros((2, C), dtype=torch.float32, device=x2.device)
        dw_ptr = skw[1].data_ptr() if direct else dwb[0].data_ptr()
        db_ptr = (skb[1].data_ptr() if ctx.has_b else 0) if direct else dwb[1].data_ptr()
        nws = lib.rscotr_layernorm_bwd_workspace(M, C) if merge is None else lib.rscotr_patch_merge_norm_bwd_workspace(*merge)
        dx_shape = dy.shape if merge is None else x2.shape
        defer = direct and DEFER.enabled and STATE.side is None and STATE.profile is None
        if merge is not None:
            ws_ptr = DEFER.reserve(nws, x2.device) if defer else _WS.get(nws, x2.device).data_ptr()
            lib.call('rscotr_patch_merge_norm_bwd', g.data_ptr(), x2.data_ptr(), _ptr(w), stats[0].data_ptr(),
                     stats[1].data_ptr(), _ptr(dx), dw_ptr, db_ptr, *merge, ws_ptr, nws, 0 if defer else 1, slot, _stream())
            if defer:
                DEFER.ln_entries.append((ws_ptr, dw_ptr, db_ptr, nws // (8 * C), C))
        elif defer:
            # the fold of the per-workgroup partial rows into dgamma / dbeta joins the end-of-pass flush (one launch for
            # all ~55 LayerNorms of a backward pass instead of one each)
            part = DEFER.reserve(nws, x2.device)
            lib.call('rscotr_layernorm_bwd_partials', g.data_ptr(), x2.data_ptr(), _ptr(w), stats[0].data_ptr(),
                     stats[1].data_ptr(), _ptr(dx), _ptr(r), M, C, part, nws, slot, _stream())
            DEFER.ln_entries.append((part, dw_ptr, db_ptr, nws // (8 * C), C))
        else:
            ws = _WS.get(nws, x2.device)
            with _Prof('layernorm_bwd', 12 * M * C):
                lib.call('rscotr_layernorm_bwd', g.data_ptr(), x2.data_ptr(), _ptr(w), stats[0].data_ptr(),
                         stats[1].data_ptr(), _ptr(dx), _ptr(r), dw_ptr, db_ptr, M, C, ws.data_ptr(), nws, slot, _stream())
        if defer:
            STATE.grad_sink.grad_written(skw[0])
            if ctx.has_b:
                STATE.grad_sink.grad_written(skb[0])
            return (None if dx is None else RANGES.tag(dx.view(dx_shape), slot)), None, None, None
        dxv = None if dx is None else RANGES.tag(dx.view(dx_shape), slot)
        if direct:  # dgamma / dbeta were accumulated straight into the gradient arena
            STATE.grad_sink.grad_written(skw[0])
            if ctx.has_b:
                STATE.grad_sink.grad_written(skb[0])
            return dxv, None, None, None
        return dxv, dwb[0] if w is not None else None, dwb[1] if ctx.has_b else None, None


class _LayerNormFork(Function):
    """(LayerNorm(x), x): a pre-norm block reads x twice -- through the norm and as the residual the branch's last
    GEMM adds back (mmdet SwinBlock: x = x + attn(norm1(x)); x = x + ffn(norm2(x))).  Returning x through this
    node brings both gradients to one backward call, where the LayerNorm backward kernel adds the residual one
    on its way out (dx_add of rscotr_layernorm_bwd) instead of autograd launching an element-wise add."""

    @staticmethod
    def forward(ctx, x, w, b, eps, lazy=False):
        ctx.set_materialize_grads(False)
        return _LayerNorm.forward(ctx, x, w, b, eps, lazy), x

    @staticmethod
    def backward(ctx, dy, dres):
        if dy is None:  # norm output unused: only the residual gradient flows
            return dres, None, None, None, None
        return _LayerNorm._backward(ctx, dy, dres) + (None,)


class _LayerNormSum(Function):
    """(LayerNorm(x), LayerNorm(x) + add) from one launch; the second output carries NO gradient: it is the `query +
    query_pos` an attention wrapper would form (ops._MHA / ops._MSDAAttn take it as `q_sum` and still return d(query) and
    d(query_pos) from their own backward), so the norm's backward is the plain one."""

    @staticmethod
    def forward(ctx, x, w, b, eps, add):
        C = x.shape[-1]
        x2 = _f32c(x).reshape(-1, C)
        M = x2.shape[0]
        if add.dim() == 3 and add.shape[0] > 1 and add.stride(0) == 0:   # one embedding for every image (expanded view)
            a2 = _f32c(add[0])
        else:
            a2 = _f32c(add).reshape(-1, C)
        rows = a2.shape[0]
        assert M % rows == 0 and a2.shape[-1] == C, (tuple(x.shape), tuple(add.shape))
        _chk(x2, w, b, a2)
        y, y2 = torch.empty_like(x2), torch.empty_like(x2)
        stats = torch.empty((2, M), dtype=torch.float32, device=x2.device)
        with _Prof('layernorm_fwd', 16 * M * C):
            slot = RANGES.out_slot(x2.device)
            lib.call('rscotr_layernorm_fwd_sum', x2.data_ptr(), _ptr(w), _ptr(b), y.data_ptr(), stats[0].data_ptr(),
                     stats[1].data_ptr(), a2.data_ptr(), rows, y2.data_ptr(), M, C, float(eps), slot, _stream())
        ctx.save_for_backward(x2, w, stats)
        ctx.has_b = b is not None
        ctx.bias = b
        y, y2 = RANGES.tag(y.view(x.shape), slot), RANGES.tag(y2.view(x.shape), slot)  # (one word bounds both outputs)
        ctx.mark_non_differentiable(y2)
        ctx.set_materialize_grads(False)   # (no zero-filled stand-in for the sum's absent gradient)
        ctx.out_shape = tuple(x.shape)
        return y, y2

    @staticmethod
    def backward(ctx, dy, _):
        if dy is None:  # (the norm's output was not used by anything that needs a gradient)
            dy = torch.zeros(ctx.out_shape, dtype=torch.float32, device=ctx.saved_tensors[0].device)
        return _LayerNorm._backward(ctx, dy, None) + (None,)


class _PatchMergeNorm(Function):
    """LayerNorm(4 Cin) of mmcv PatchMerging's nn.Unfold(2, stride 2) rows of a (B, H*W, Cin) token map, the unfold done by
    the norm kernels' own loads / stores (no gathered copy in either direction)."""

    @staticmethod
    def forward(ctx, x, w, b, eps, H, W):
        B, L, Cin = x.shape
        assert L == H * W
        x3 = _f32c(x)
        Ho, Wo = (H + 1) // 2, (W + 1) // 2
        M, C = B * Ho * Wo, 4 * Cin
        _chk(x3, w, b)
        y = torch.empty((B, Ho * Wo, C), dtype=torch.float32, device=x3.device)
        stats = torch.empty((2, M), dtype=torch.float32, device=x3.device)
        with _Prof('layernorm_fwd', 8 * M * C):
            slot = RANGES.out_slot(x3.device)
            lib.call('rscotr_patch_merge_norm_fwd', x3.data_ptr(), _ptr(w), _ptr(b), y.data_ptr(), stats[0].data_ptr(),
                     stats[1].data_ptr(), B, H, W, Cin, float(eps), slot, _stream())
        ctx.save_for_backward(x3, w, stats)
        ctx.has_b, ctx.bias, ctx.merge = b is not None, b, (B, H, W, Cin)
        return RANGES.tag(y, slot)

    @staticmethod
    def backward(ctx, dy):
        return _LayerNorm._backward(ctx, dy, None) + (None, None)


def layer_norm(x, w, b, eps=LN_EPS):
    return _LayerNorm.apply(x, w, b, eps)


def patch_merge_norm(x, hw, w, b, eps=LN_EPS):
    """(B, H*W, Cin) -> (LayerNorm over the (B, ceil(H/2)*ceil(W/2), 4 Cin) nn.Unfold(2, stride 2) rows (c*4 + kh*2 + kw, zero
    padding for odd H / W), (ceil(H/2), ceil(W/2))): ops.layer_norm(ops.patch_merge_gather(x, hw)) without the gathered copy."""
    H, W = hw
    return _PatchMergeNorm.apply(x, w, b, eps, int(H), int(W)), ((H + 1) // 2, (W + 1) // 2)


def layer_norm_sum(x, w, b, add, eps=LN_EPS):
    """(LayerNorm(x), LayerNorm(x) + add.detach()): `add` (same shape as x, or one embedding expanded over the batch) is
    the positional embedding of the attention that consumes the norm's output; see _LayerNormSum."""
    return _LayerNormSum.apply(x, w, b, eps, add.detach())


def layer_norm_fork(x, w, b, eps=LN_EPS, lazy=False):
    """(LayerNorm(x), x) for pre-norm residual blocks: use the second output as the residual.  lazy=True: the norm's output goes
    to ops.mlp and nowhere else — its launch is left to that call (LazyNorm), which may fold it into the fused MLP kernel."""
    return _LayerNormFork.apply(x, w, b, eps, bool(lazy))


class _GroupNormTokens(Function):
    @staticmethod
    def forward(ctx, x, w, b, groups, eps):
        x = _f32c(x)
        _chk(x, w, b)
        B, L, C = x.shape
        y = torch.empty_like(x)
        stats = torch.empty((B, groups, 2), dtype=torch.float32, device=x.device)
        nws = lib.rscotr_groupnorm_tokens_workspace(B, L, C, groups)
        lib.call('rscotr_groupnorm_tokens_fwd', x.data_ptr(), _ptr(w), _ptr(b), y.data_ptr(), stats.data_ptr(),
                 B, L, C, groups, float(eps), _WS.get(nws, x.device).data_ptr(), nws, _stream())
        ctx.save_for_backward(x, w, stats)
        ctx.groups, ctx.has_b, ctx.bias = groups, b is not None, b
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, stats = ctx.saved_tensors
        B, L, C = x.shape
        # one level's rows of a gradient over the concatenated levels (dense rows, batch stride of all levels): read in place
        if not (dy.dtype == torch.float32 and dy.dim() == 3 and dy.stride(2) == 1 and dy.stride(1) == C
                and dy.stride(0) >= L * C and dy.stride(0) % 4 == 0 and dy.data_ptr() % 16 == 0):
            dy = _f32c(dy)
        dx = torch.empty_like(x)
        # dgamma / dbeta are ADDED by the kernel: straight into the gradient arena when both parameters are sunk (no
        # zero-filled staging rows, no accumulate launches by autograd)
        skw, skb = _sink(w), _sink(ctx.bias)
        direct = w is not None and skw is not None and (skb is not None or not ctx.has_b)
        dwb = None if direct else torch.zeros((2, C), dtype=torch.float32, device=x.device)
        dw_ptr = skw[1].data_ptr() if direct else dwb[0].data_ptr()
        db_ptr = (skb[1].data_ptr() if ctx.has_b else 0) if direct else dwb[1].data_ptr()
        proj = torch.empty((B, ctx.groups, 2), dtype=torch.float32, device=x.device)
        nws = lib.rscotr_groupnorm_tokens_workspace(B, L, C, ctx.groups)
        lib.call('rscotr_groupnorm_tokens_bwd', dy.data_ptr(), x.data_ptr(), _ptr(w), stats.data_ptr(), dx.data_ptr(),
                 dw_ptr, db_ptr, proj.data_ptr(), B, L, C, ctx.groups, dy.stride(0),
                 _WS.get(nws, x.device).data_ptr(), nws, _stream())
        if direct:
            STATE.grad_sink.grad_written(skw[0])
            if ctx.has_b:
                STATE.grad_sink.grad_written(skb[0])
            return dx, None, None, None, None
        return dx, dwb[0] if w is not None else None, dwb[1] if ctx.has_b else None, None, None


def group_norm_tokens(x, groups, w, b, eps=1e-5):
    """nn.GroupNorm(groups, C) on token layout (B, L, C): statistics per (image, group) over all L tokens."""
    return _GroupNormTokens.apply(x, w, b, groups, eps)

