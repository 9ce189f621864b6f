"""Small fused ops between the big kernels: level embeddings, gradient fan-in, denoising queries, the sine position
embedding, the neck's im2col, layout helpers, the cls head's pooling and loss."""
import ctypes

import torch
import torch.nn.functional as F
from torch.autograd import Function

from .core import _WS, _chk, _f32c, _ptr, _sink, _stream, lib
from .matmul import linear
from .ranges import RANGES
from .state import STATE

def sine_embed4(pos):
    """gen_sineembed_for_position of the DINO decoder: pos (B,Q,4) (no gradient) -> (B,Q,512), one kernel."""
    p = _f32c(pos.detach())
    _chk(p)
    out = torch.empty(p.shape[:-1] + (512,), dtype=torch.float32, device=p.device)
    lib.call('rscotr_sine_embed4', p.data_ptr(), out.data_ptr(), p.numel() // 4, _stream())
    return out


_LEVEL_COUNTERS = {}


class _LevelEmbedAdd(Function):
    """out[b, t] = x[b, t] + const[b | 0, t] + weight[row0 + level(t)] over concatenated levels, ONE launch; backward:
    d(x) = the incoming gradient itself, d(weight) = fixed-order segment sums (one launch, straight into the gradient arena
    when the parameter is sunk) — instead of a broadcast add + concatenation per level forward and a sum-reduce / select
    zero-fill / copy / accumulate chain per level in backward.  x may be a batch-strided view with dense rows."""

    @staticmethod
    def forward(ctx, x, weight, const, sizes, batch, row0):
        L = len(sizes)
        N = int(sum(sizes))
        C = weight.shape[-1]
        w = weight if weight.is_contiguous() else weight.contiguous()
        if x is not None and not (x.dtype == torch.float32 and x.dim() == 3 and x.stride(2) == 1 and x.stride(1) == C):
            x = _f32c(x)
        c2 = None if const is None else _f32c(const)
        B = x.shape[0] if x is not None else (batch or c2.shape[0])
        assert weight.shape[0] >= row0 + L and (x is None or x.shape == (B, N, C))
        assert c2 is None or (c2.shape[1:] == (N, C) and c2.shape[0] in (1, B))
        _chk(w, c2)
        assert x is None or x.is_cuda
        out = torch.empty((B, N, C), dtype=torch.float32, device=w.device)
        arr = (ctypes.c_int * L)(*[int(v) for v in sizes])
        lib.call('rscotr_level_embed_fwd', _ptr(x), 0 if x is None else x.stride(0), _ptr(c2),
                 int(c2 is not None and c2.shape[0] == B and B > 1), w.data_ptr() + row0 * C * 4, out.data_ptr(), arr, L, B, N,
                 C, _stream())
        ctx.sizes, ctx.geom, ctx.weight = tuple(int(v) for v in sizes), (B, N, C, L, row0), weight
        return out

    @staticmethod
    def backward(ctx, g):
        B, N, C, L, row0 = ctx.geom
        need = ctx.needs_input_grad
        gw = None
        if need[1]:
            g2 = _f32c(g)
            dev = g2.device
            sk = _sink(ctx.weight)
            if sk is not None:
                dw, acc = sk[1], 1
            else:
                dw = gw = (torch.zeros if ctx.weight.shape[0] > L else torch.empty)(
                    tuple(ctx.weight.shape), dtype=torch.float32, device=dev)
                acc = 0
            cnt = _LEVEL_COUNTERS.get((dev, _stream()))
            if cnt is None:  # (zero once; the kernel returns its counters to zero)
                cnt = _LEVEL_COUNTERS[(dev, _stream())] = torch.zeros(8, dtype=torch.int32, device=dev)
            nws = lib.rscotr_level_embed_bwd_workspace(L, C)
            ws = _WS.get(nws, dev)
            arr = (ctypes.c_int * L)(*ctx.sizes)
            lib.call('rscotr_level_embed_bwd', g2.data_ptr(), dw.data_ptr() + row0 * C * 4, arr, L, B, N, C, acc, ws.data_ptr(),
                     cnt.data_ptr(), _stream())
            if sk is not None:
                STATE.grad_sink.grad_written(sk[0])
        return (g if need[0] else None), gw, None, None, None, None


def level_embed_add(x, weight, sizes, const=None, batch=None, row0=0):
    """x (B,N,C) | None, const (B|1, N, C) | None (no gradient), weight (>= len(sizes), C): every token of level l gets
    weight[l] added (levels concatenated along N, sizes[l] tokens each; level l takes row row0 + l); batch = B of the result when x is None."""
    if const is not None:
        const = const.detach()
    return _LevelEmbedAdd.apply(x, weight, const, tuple(sizes), batch, row0)




class _FanOut(Function):
    """n handles of one tensor for n consumers: the gradients of all of them arrive in ONE backward call and are summed by
    ONE launch per 8 of them (rscotr_sum8, fixed order) instead of by n - 1 pairwise adds of the autograd engine."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(RANGES.carry(x, x.view_as(x)) for _ in range(n))  # (the handles hold x's values: and its value range)

    @staticmethod
    def backward(ctx, *grads):
        gs = [g for g in grads if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        gs = [_f32c(g) for g in gs]
        _chk(*gs)
        count = gs[0].numel()
        if count % 4 or any(g.shape != gs[0].shape for g in gs):
            out = gs[0]
            for g in gs[1:]:
                out = out + g
            return out, None
        out = torch.empty_like(gs[0])
        cur, rest = None, gs
        while rest:
            take = rest[:8] if cur is None else [cur] + rest[:7]
            rest = rest[8:] if cur is None else rest[7:]
            ptrs = [t.data_ptr() for t in take] + [0] * (8 - len(take))
            lib.call('rscotr_sum8', *ptrs, len(take), out.data_ptr(), count, _stream())
            cur = out
        return out, None


def fan_out(x, n):
    """-> n handles of x, one per consumer (see _FanOut); x itself when nothing is to be gained (n <= 2, no gradient)."""
    if n <= 2 or not STATE.fan_out or not (torch.is_tensor(x) and x.requires_grad and x.is_cuda):
        return [x] * n
    return list(_FanOut.apply(x, n))


class _BatchParam(Function):
    """p[None].expand(B, ...) of a parameter (query embeddings shared by the images of a batch) whose backward sums the B
    gradient slices AND adds them to the parameter's rows of the gradient arena in one launch (rscotr_sum8 with the arena as
    first addend and output) — instead of autograd's reduction over the batch axis plus an accumulate launch."""

    @staticmethod
    def forward(ctx, p, B):
        ctx.param, ctx.B = p, B
        return p.detach()[None].expand(B, *p.shape)

    @staticmethod
    def backward(ctx, g):
        p, B = ctx.param, ctx.B
        sk = _sink(p)
        n = p.numel()
        # (a batch-strided gradient with dense per-image blocks — a slice of a concatenation along the query axis — is read in place)
        dense = g.dtype == torch.float32 and all(g[b].is_contiguous() and g[b].data_ptr() % 16 == 0 for b in range(B))
        if not dense:
            g = _f32c(g)
        if sk is None or B > 7 or n % 4 or not g.is_cuda:
            return g.sum(0), None
        ptrs = [sk[1].data_ptr()] + [g[b].data_ptr() for b in range(B)] + [0] * (7 - B)
        lib.call('rscotr_sum8', *ptrs, B + 1, sk[1].data_ptr(), n, _stream())
        STATE.grad_sink.grad_written(sk[0])
        return None, None


def batch_param(p, B):
    """(B, *p.shape) expanded view of parameter p, one copy per image of the batch (see _BatchParam)."""
    if not (p.requires_grad and p.is_cuda and p.is_contiguous()):
        return p[None].expand(B, *p.shape)
    return _BatchParam.apply(p, B)


class _CdnQueries(Function):
    """The denoising queries of a det batch in slot layout, one launch (rscotr_cdn_queries); the only gradient is the
    label embedding's (fixed-order scatter, straight into the arena when the parameter is sunk)."""

    @staticmethod
    def forward(ctx, weight, gt_lab, gt_boxn, slot_src, slot_valid, slot_neg, u, uniform, label_thr, box_scale, num_classes):
        w = weight if weight.is_contiguous() else weight.contiguous()
        gt_lab, gt_boxn, slot_src = gt_lab.contiguous(), _f32c(gt_boxn), slot_src.contiguous()
        slot_valid, slot_neg, u = _f32c(slot_valid), _f32c(slot_neg), _f32c(u)
        _chk(w, gt_lab, gt_boxn, slot_src, slot_valid, slot_neg, u)
        assert gt_lab.dtype == torch.int64 and slot_src.dtype == torch.int64 and u.shape == slot_src.shape + (10,)
        n, C = slot_src.numel(), w.shape[1]
        kl = torch.empty(slot_src.shape, dtype=torch.int64, device=w.device)
        q_label = torch.empty(slot_src.shape + (C,), dtype=torch.float32, device=w.device)
        q_bbox = torch.empty(slot_src.shape + (4,), dtype=torch.float32, device=w.device)
        lib.call('rscotr_cdn_queries', gt_lab.data_ptr(), gt_boxn.data_ptr(), slot_src.data_ptr(), slot_valid.data_ptr(),
                 slot_neg.data_ptr(), u.data_ptr(), int(uniform), w.data_ptr(), float(label_thr), float(box_scale),
                 int(num_classes), kl.data_ptr(), q_label.data_ptr(), q_bbox.data_ptr(), n, C, _stream())
        ctx.save_for_backward(kl, slot_valid)
        ctx.weight = weight
        ctx.mark_non_differentiable(q_bbox)
        return q_label, q_bbox

    @staticmethod
    def backward(ctx, g_label, g_bbox):
        kl, slot_valid = ctx.saved_tensors
        if not ctx.needs_input_grad[0]:
            return (None,) * 11
        g = _f32c(g_label)
        rows, C = ctx.weight.shape
        sk = _sink(ctx.weight)
        dw = sk[1] if sk is not None else torch.empty((rows, C), dtype=torch.float32, device=g.device)
        lib.call('rscotr_cdn_embed_grad', g.data_ptr(), kl.data_ptr(), slot_valid.data_ptr(), dw.data_ptr(), rows, kl.numel(), C,
                 int(sk is not None), _stream())
        if sk is not None:
            STATE.grad_sink.grad_written(sk[0])
            return (None,) * 11
        return (dw,) + (None,) * 10


def cdn_queries(weight, gt_lab, gt_boxn, slot_src, slot_valid, slot_neg, u, uniform, label_noise_scale, box_noise_scale,
                num_classes):
    """-> (q_label (B,PC,C), q_bbox (B,PC,4)): see include/rscotr.h, rscotr_cdn_queries.  u (B,PC,10)."""
    return _CdnQueries.apply(weight, gt_lab, gt_boxn, slot_src, slot_valid, slot_neg, u, bool(uniform),
                             label_noise_scale * 0.5 if label_noise_scale > 0 else 0.0, max(box_noise_scale, 0.0), num_classes)


class _Im2Col3x3s2(Function):
    @staticmethod
    def forward(ctx, x, H, W):
        x = _f32c(x)
        _chk(x)
        B, L, C = x.shape
        Ho, Wo = (H + 1) // 2, (W + 1) // 2
        col = torch.empty((B, Ho * Wo, C * 9), dtype=torch.float32, device=x.device)
        lib.call('rscotr_im2col3x3s2_tokens', x.data_ptr(), col.data_ptr(), B, H, W, C, _stream())
        ctx.geom = (B, H, W, C)
        return col

    @staticmethod
    def backward(ctx, dcol):
        B, H, W, C = ctx.geom
        dcol = _f32c(dcol)
        dx = torch.empty((B, H * W, C), dtype=torch.float32, device=dcol.device)
        lib.call('rscotr_col2im3x3s2_tokens', dcol.data_ptr(), dx.data_ptr(), B, H, W, C, _stream())
        return dx, None, None


def conv3x3s2_tokens(x, hw, w):
    """Conv2d(C, O, 3, stride=2, padding=1, bias=False) on tokens (B, H*W, C) -> ((B, Ho*Wo, O), (Ho, Wo)):
    im2col gather kernel + MFMA GEMM against the weight flattened (O, C*9)."""
    H, W = hw
    col = _Im2Col3x3s2.apply(x, H, W)
    return linear(col, w.reshape(w.shape[0], -1), None), ((H + 1) // 2, (W + 1) // 2)


def map_to_tokens(x):
    """(B, C, H, W) -> (B, H*W, C); free for the channels-last views tokens_to_map returns."""
    B, C, H, W = x.shape
    return RANGES.carry(x, x.permute(0, 2, 3, 1).reshape(B, H * W, C)), (H, W)  # (the same values: the range word travels along)


def residual_droppath(x, y, keep, rate):
    """x + DropPath(y): mmcv drop_path = y / keep_prob * floor(keep_prob + U); `keep` (B,) are
    the 0/1 floors drawn by the caller."""
    if keep is None or rate == 0.0:
        return x + y
    scale = (keep / (1.0 - rate)).view(-1, *([1] * (y.dim() - 1)))
    return x + y * scale


def patch_embed(img, w, b, k):
    """Conv2d(3, C, k, stride=k) (+ "corner" padding to a multiple of k) as an MFMA GEMM: the
    non-overlapping patches are a pure re-indexing of the image, K = 3*k*k ordered (c, ky, kx)."""
    H, W = img.shape[-2:]
    if H % k or W % k:
        img = F.pad(img, (0, (k - W % k) % k, 0, (k - H % k) % k))
        H, W = img.shape[-2:]
    B, Cin = img.shape[:2]
    hw = (H // k, W // k)
    patches = img.view(B, Cin, hw[0], k, hw[1], k).permute(0, 2, 4, 1, 3, 5).reshape(B, hw[0] * hw[1], Cin * k * k)
    return linear(patches, w.reshape(w.shape[0], -1), b), hw


def patch_merge_gather(x, hw):
    """(B, H*W, C) -> (B, H/2*W/2, 4C) in nn.Unfold(2, stride 2) order (c*4 + kh*2 + kw)."""
    B, L, C = x.shape
    H, W = hw
    y = x.view(B, H, W, C)
    if H % 2 or W % 2:
        y = F.pad(y, (0, 0, 0, W % 2, 0, H % 2))
        H, W = y.shape[1], y.shape[2]
    y = y.view(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 5, 2, 4)
    return y.reshape(B, (H // 2) * (W // 2), 4 * C), (H // 2, W // 2)


def tokens_to_map(x, hw):
    """(B, H*W, C) -> (B, C, H, W) as a channels-last VIEW (no copy): every consumer either flattens it
    back to tokens (free) or reduces over H, W."""
    B, L, C = x.shape
    return RANGES.carry(x, x.view(B, hw[0], hw[1], C).permute(0, 3, 1, 2))


class _GapTokens(Function):
    @staticmethod
    def forward(ctx, tok):
        B, T, C = tok.shape
        out = torch.empty((B, C), dtype=torch.float32, device=tok.device)
        lib.call('rscotr_gap_tokens_fwd', tok.data_ptr(), out.data_ptr(), B, T, C, _stream())
        ctx.geom = (B, T, C)
        return out

    @staticmethod
    def backward(ctx, g):
        B, T, C = ctx.geom
        g = _f32c(g)
        dx = torch.empty((B, T, C), dtype=torch.float32, device=g.device)
        lib.call('rscotr_gap_tokens_bwd', g.data_ptr(), dx.data_ptr(), B, T, C, _stream())
        return dx


def global_avg_pool(x):
    """mmcls GlobalAveragePooling of a (B, C, H, W) map.  The maps of this path are channels-last views of token tensors
    (tokens_to_map): pooled by one kernel over the tokens, with a dense gradient (the generic mean's expanded gradient costs
    the consumer a copy); any other layout goes through the device library."""
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] % 4 == 0:
        tok = x.permute(0, 2, 3, 1)
        if tok.is_contiguous():
            B, H, W, C = tok.shape
            return _GapTokens.apply(tok.reshape(B, H * W, C))
    return x.mean(dim=(2, 3))


class _SoftCE(Function):
    @staticmethod
    def forward(ctx, score, soft_label, smooth, avg_factor):
        score, soft_label = _f32c(score), _f32c(soft_label.detach())
        _chk(score, soft_label)
        B, C = score.shape
        loss = torch.empty((), dtype=torch.float32, device=score.device)
        dscore = torch.empty_like(score)
        lib.call('rscotr_soft_ce', score.data_ptr(), soft_label.data_ptr(), loss.data_ptr(), dscore.data_ptr(), B, C,
                 float(smooth), float(avg_factor), _stream())
        ctx.save_for_backward(dscore)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dscore,) = ctx.saved_tensors
        return dscore * g, None, None, None


def soft_ce_label_smooth(score, soft_label, smooth, avg_factor):
    """mmcls LabelSmoothLoss('original') + soft cross-entropy, sum / avg_factor: loss and d(loss)/d(score) in one launch
    (rscotr_soft_ce) instead of a smoothing / log-softmax / multiply / sum / divide chain and its five backward nodes."""
    if score.is_cuda and score.dim() == 2 and score.shape[0] <= 1024 and not soft_label.requires_grad:
        return _SoftCE.apply(score, soft_label, float(smooth), float(avg_factor))
    C = score.shape[-1]
    t = soft_label * (1 - smooth) + smooth / C
    return (-t * F.log_softmax(score, dim=-1)).sum() / avg_factor

