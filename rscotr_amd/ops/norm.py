"""LayerNorm (plain and forked for pre-norm residual blocks) and GroupNorm on token layout."""
import torch
from torch.autograd import Function

from .core import _WS, _Prof, _chk, _f32c, _ptr, _sink, _stream, lib
from .matmul import DEFER
from .state import STATE

LN_EPS = 1e-5


class _LayerNorm(Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        C = x.shape[-1]
        x2 = _f32c(x).reshape(-1, C)
        M = x2.shape[0]
        _chk(x2, w, b)
        y = torch.empty_like(x2)
        stats = torch.empty((2, M), dtype=torch.float32, device=x2.device)
        with _Prof('layernorm_fwd', 8 * M * C):
            lib.call('rscotr_layernorm_fwd', x2.data_ptr(), _ptr(w), _ptr(b), y.data_ptr(), stats[0].data_ptr(),
                     stats[1].data_ptr(), M, C, float(eps), _stream())
        ctx.save_for_backward(x2, w, stats)
        ctx.has_b = b is not None
        ctx.bias = b
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy, dres=None):
        return _LayerNorm._backward(ctx, dy, dres)

    @staticmethod
    def _backward(ctx, dy, dres):
        x2, w, stats = ctx.saved_tensors
        M, C = x2.shape
        g = _f32c(dy).reshape(M, C)
        r = None if dres is None else _f32c(dres).reshape(M, C)  # residual-branch gradient, added inside the kernel
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        skw, skb = _sink(w), _sink(ctx.bias)
        direct = skw is not None and (skb is not None or not ctx.has_b)
        dwb = None if direct else torch.zeros((2, C), dtype=torch.float32, device=x2.device)
        dw_ptr = skw[1].data_ptr() if direct else dwb[0].data_ptr()
        db_ptr = (skb[1].data_ptr() if ctx.has_b else 0) if direct else dwb[1].data_ptr()
        nws = lib.rscotr_layernorm_bwd_workspace(M, C)
        if direct and DEFER.enabled and STATE.side is None and STATE.profile is None:
            # the fold of the per-workgroup partial rows into dgamma / dbeta joins the end-of-pass flush (one launch for
            # all ~55 LayerNorms of a backward pass instead of one each)
            part = DEFER.reserve(nws, x2.device)
            lib.call('rscotr_layernorm_bwd_partials', g.data_ptr(), x2.data_ptr(), _ptr(w), stats[0].data_ptr(),
                     stats[1].data_ptr(), _ptr(dx), _ptr(r), M, C, part, nws, _stream())
            DEFER.ln_entries.append((part, dw_ptr, db_ptr, nws // (8 * C), C))
            STATE.grad_sink.grad_written(skw[0])
            if ctx.has_b:
                STATE.grad_sink.grad_written(skb[0])
            return (None if dx is None else dx.view(dy.shape)), None, None, None
        ws = _WS.get(nws, x2.device)
        with _Prof('layernorm_bwd', 12 * M * C):
            lib.call('rscotr_layernorm_bwd', g.data_ptr(), x2.data_ptr(), _ptr(w), stats[0].data_ptr(),
                     stats[1].data_ptr(), _ptr(dx), _ptr(r), dw_ptr, db_ptr, M, C, ws.data_ptr(), nws, _stream())
        dxv = None if dx is None else dx.view(dy.shape)
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
    def forward(ctx, x, w, b, eps):
        ctx.set_materialize_grads(False)
        return _LayerNorm.forward(ctx, x, w, b, eps), x

    @staticmethod
    def backward(ctx, dy, dres):
        if dy is None:  # norm output unused: only the residual gradient flows
            return dres, None, None, None
        return _LayerNorm._backward(ctx, dy, dres)


def layer_norm(x, w, b, eps=LN_EPS):
    return _LayerNorm.apply(x, w, b, eps)


def layer_norm_fork(x, w, b, eps=LN_EPS):
    """(LayerNorm(x), x) for pre-norm residual blocks: use the second output as the residual."""
    return _LayerNormFork.apply(x, w, b, eps)


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
                 dw_ptr, db_ptr, proj.data_ptr(), B, L, C, ctx.groups,
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

