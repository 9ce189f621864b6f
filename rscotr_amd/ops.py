"""Operator layer of the product path: torch.autograd.Function wrappers over the C ABI.

Each `*_hip` function launches hand-written gfx950 kernels from librscotr.so on the current
torch stream with raw device pointers (PyTorch only provides memory, streams and autograd
bookkeeping).  There is NO CPU / eager fallback: a missing library, a CPU tensor, or a non-zero
return code raises.
"""
import ctypes
import os

import torch
from torch.autograd import Function

from ._lib import lib


def _stream():
    # raw hipStream_t of torch's current stream (torch.cuda.current_stream() costs ~10 us of host time)
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


_WS_POISON = os.environ.get('RSCOTR_WS_POISON') == '1'


class _Workspace:
    """Grow-only scratch buffer per device for kernel workspaces (split-K slabs, reduction partials,
    MSDA sort buffers).  Kernels that use it run on the same stream, so consecutive users are
    ordered; the buffer is never handed to autograd."""

    MIN_WORDS = 16 << 20  # 64 MB up front: covers every workspace of the 512x512 step

    def __init__(self):
        self.buf = {}
        self.retired = []  # outgrown buffers stay alive: captured hipGraphs hold their addresses

    def get(self, nbytes, device):
        # one buffer per (device, stream): the weight-gradient contractions run on a side stream (SIDE) next to
        # the main chain and must not share slabs with it
        key = (device, _stream())
        b = self.buf.get(key)
        if b is None or b.numel() * 4 < nbytes:
            if b is not None:
                self.retired.append(b)
            b = torch.empty(max((nbytes + 3) // 4, self.MIN_WORDS), dtype=torch.int32, device=device)
            self.buf[key] = b
        if _WS_POISON:  # debugging aid: every user finds NaN bit patterns in whatever it did not write itself
            b.fill_(0x7FC00000)
        return b


_WS = _Workspace()
_gemm_ws_bytes = {}


class _WeightPlanes:
    """bf16 plane sets of the parameters that serve as the B operand of y = x W^T (and dx = dy W): rscotr_gemm_split_weights
    writes them ONCE per optimizer step and rscotr_gemm_f32_wplanes multiplies fp32 activations with them (include/rscotr.h).
    A parameter is recognised by its address inside the optimizer's flat arena (`GRAD_SINK.is_param_ptr`); the sets a task
    uses are remembered under the task's name (`begin`), and the first product of an iteration that finds them stale
    re-splits ALL of them in one grouped launch (inside the per-task hipGraph when the iteration is replayed).  `bump()` =
    "the parameters have changed" (optimizer step, checkpoint load, snapshot restore)."""

    def __init__(self):
        self.enabled = os.environ.get('RSCOTR_WPLANES', '1') != '0'
        self.min_m = int(os.environ.get('RSCOTR_WPLANES_MIN_M', 4096))
        self.min_k = int(os.environ.get('RSCOTR_WPLANES_MIN_K', 1024))
        self.version = 1
        self.entries, self.groups, self.tables = {}, {}, {}
        self.current = None

    def begin(self, group):
        self.current = group

    def reset(self):
        """Forget every plane set (a new optimizer arena: addresses may be reused by other parameters)."""
        self.entries, self.groups, self.tables = {}, {}, {}
        self.version += 1

    def bump(self):
        self.version += 1

    def eligible(self, A, B, M, N, K, lda, ldb, a_kmajor, b_kmajor):
        if not self.enabled or a_kmajor or GRAD_SINK is None or K % 16 or K < self.min_k or M < self.min_m or N < 64:
            return False
        if lda % 4 or A.data_ptr() % 16 or (b_kmajor and ldb % 4):
            return False
        if lib.rscotr_gemm_get_precision() != 3:
            return False
        return GRAD_SINK.is_param_ptr(B.data_ptr())

    def get(self, B, N, K, ldb, b_kmajor):
        """-> (planes pointer, npad) of the weight behind operand B (N output rows, reduction K), fresh."""
        key = (B.data_ptr(), N, K, ldb, int(b_kmajor))
        e = self.entries.get(key)
        if e is None:
            npad = (N + 255) // 256 * 256
            e = self.entries[key] = dict(planes=torch.empty(npad * K * 3, dtype=torch.int16, device=B.device), npad=npad,
                                         version=0, blocks=(npad * (K // 16) + 255) // 256)
        keys = self.groups.setdefault(self.current, [])
        if key not in keys:
            keys.append(key)
        if e['version'] != self.version:
            self._refresh(keys, B.device)
        return e['planes'].data_ptr(), e['npad']

    def _refresh(self, keys, dev):
        stale = tuple(k for k in keys if self.entries[k]['version'] != self.version)
        hit = self.tables.get(stale)
        if hit is None:
            import numpy as np
            rows, first = [], 0
            for (ptr, N, K, ldb, tr) in stale:
                e = self.entries[(ptr, N, K, ldb, tr)]
                # table row {W, planes, rows of W, cols of W, ldw, npad, first block, transposed}: operand B (N, K) row-major is
                # W itself; operand B k-major is the (K, N) matrix W whose TRANSPOSE is multiplied (planes of W^T)
                rows.append((ptr, e['planes'].data_ptr(), K if tr else N, N if tr else K, ldb, e['npad'], first, tr))
                first += e['blocks']
            hit = self.tables[stale] = (torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(dev), len(rows), first)
        lib.call('rscotr_gemm_split_weights', hit[0].data_ptr(), hit[1], hit[2], _stream())
        for k in stale:
            self.entries[k]['version'] = self.version


WPLANES = _WeightPlanes()


class _Side:
    """Second stream for the weight-gradient (dW / db) contractions of backward.  They only feed the gradient
    arena, which nobody reads before the optimizer step, so they need not sit on the critical path: each one
    is forked off the main stream (event after its inputs exist) and the main stream joins once, after
    backward (`side_join`).  The step's small GEMMs occupy a fraction of the 256 CUs, so the two chains overlap.
    Tensors a side-stream kernel reads are kept alive until the join (the caching allocator would otherwise hand
    their memory to later main-stream allocations — also inside a hipGraph capture).  Only active together with
    the gradient sink (results never flow back into autograd) and never while gradient buckets are exchanged
    from backward hooks."""

    def __init__(self):
        self.stream = torch.cuda.Stream()
        self.keep = []
        self.forked = False

    def run(self, fn, *keep):
        ev = torch.cuda.Event()
        ev.record()
        self.stream.wait_event(ev)
        with torch.cuda.stream(self.stream):
            fn()
        self.keep.extend(keep)
        self.forked = True

    def join(self):
        if self.forked:
            torch.cuda.current_stream().wait_stream(self.stream)
            self.forked = False
        self.keep = []


SIDE = None  # active side stream: set by the runner around a hipGraph-replayed iteration's backward
_SIDE_OBJ = None


def side_enable(on=True):
    global SIDE, _SIDE_OBJ
    if on and _SIDE_OBJ is None:
        _SIDE_OBJ = _Side()  # one stream (and one split-K workspace) for the life of the process
    SIDE = _SIDE_OBJ if on else None
    return SIDE


def side_join():
    if SIDE is not None:
        SIDE.join()


def _off_path(fn, *keep):
    """Run a weight-gradient contraction: on the side stream when one is active, else inline."""
    if SIDE is None or GRAD_SINK is None:
        fn()
    else:
        SIDE.run(fn, *keep)


# When set to a list (bench.py), every launch of a profiled HIP kernel appends
# dict(kind, bytes, e0, e1): HIP events recorded on the launch stream around the kernel and the
# ALGORITHMIC bytes of that launch.  None = no overhead.
PROFILE = None


PROFILE_EVERY = {'gemm': 8}  # record every n-th launch of a kind (HIP events cost host time)
_prof_count = {}


class _Prof:
    def __init__(self, kind, nbytes, name=None, shape=None):
        self.rec = None
        if PROFILE is not None:
            n = _prof_count.get(kind, 0)
            _prof_count[kind] = n + 1
            if n % PROFILE_EVERY.get(kind, 1) == 0:
                self.rec = dict(kind=kind, bytes=nbytes, name=name, shape=shape,
                                e0=torch.cuda.Event(enable_timing=True), e1=torch.cuda.Event(enable_timing=True))

    def __enter__(self):
        if self.rec is not None:
            self.rec['e0'].record()
        return self

    def __exit__(self, *exc):
        if self.rec is not None:
            self.rec['e1'].record()
            PROFILE.append(self.rec)
        return False


def _chk(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('rscotr HIP op called with a CPU tensor: the product path has no CPU fallback')
        if not t.is_contiguous():
            raise RuntimeError('rscotr HIP op requires contiguous tensors')


def _f32c(t):
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


# ------------------------------------------------------------------------------------------
# multi-scale deformable attention sampling (mmcv MultiScaleDeformableAttnFunction contract)
# ------------------------------------------------------------------------------------------
# 'tiled' (default): per-tile scan + LDS sort + register accumulation in sample order, bit-reproducible, 3 launches;
# 'sorted': counting sort by destination token + pull, bit-reproducible, 8 launches; 'scatter': atomic accumulation
# (order-dependent).  Tests run all three.
MSDA_BWD_STRATEGY = os.environ.get('RSCOTR_MSDA_BWD', 'tiled')

# host copies of the level-shape tensors (the mmcv contract keeps spatial_shapes on the device; the tile-accumulation
# backward sizes its launches from the shapes): data_ptr -> int64 numpy array, registered by whoever builds the device
# tensor (layers.LevelGeometry); an unregistered tensor is read back once (a device sync, eager callers only)
_MSDA_HOST_SHAPES = {}


def msda_register_shapes(spatial_shapes, shapes):
    import numpy as np
    _MSDA_HOST_SHAPES[spatial_shapes.data_ptr()] = np.ascontiguousarray(np.asarray(shapes, dtype=np.int64).reshape(-1, 2))


def _msda_host_shapes(spatial_shapes):
    a = _MSDA_HOST_SHAPES.get(spatial_shapes.data_ptr())
    if a is None:
        msda_register_shapes(spatial_shapes, spatial_shapes.detach().cpu().numpy())
        a = _MSDA_HOST_SHAPES[spatial_shapes.data_ptr()]
    return a


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
    if MSDA_BWD_STRATEGY != 'scatter':
        nws = lib.rscotr_msda_bwd_workspace(B, Nk, Nq, H, L, P)
        if MSDA_BWD_STRATEGY == 'tiled':
            hs = _msda_host_shapes(spatial_shapes)
            hs_ptr = hs.ctypes.data
            nws = max(nws, lib.rscotr_msda_bwd_tiled_workspace(hs_ptr, B, Nk, Nq, H, D, L, P))
    ws = _WS.get(nws, value.device) if nws else None
    grad_value = torch.zeros_like(value) if ws is None else torch.empty_like(value)
    grad_loc = torch.empty_like(loc)
    grad_attn = torch.empty_like(attn)
    # algorithmic bytes: read value, RMW grad_value, read loc/attn/grad_out, write grad_loc/attn
    nbytes = 4 * B * (3 * Nk * H * D + Nq * H * L * P * 3 + Nq * H * D + Nq * H * L * P * 3)
    with _Prof('msda_bwd', nbytes):
        lib.call('rscotr_msda_bwd', value.data_ptr(), spatial_shapes.data_ptr(),
                 level_start_index.data_ptr(), loc.data_ptr(), attn.data_ptr(), grad_out.data_ptr(),
                 grad_value.data_ptr(), grad_loc.data_ptr(), grad_attn.data_ptr(),
                 B, Nk, Nq, H, D, L, P, hs_ptr, 0 if ws is None else ws.data_ptr(), nws, _stream())
    return grad_value, grad_loc, grad_attn


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
    lib.call('rscotr_msda_prep_bwd', gloc.data_ptr(), gattn.data_ptr(), attn.data_ptr(), ref.data_ptr(), _ptr(norm),
             goff.data_ptr(), glogit.data_ptr(), B, Nq, H, L, P, ref.shape[-1], ldo, ldl, ref.shape[-2], _stream())
    return (both, None) if packed else (goff, glogit)


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
                GRAD_SINK.grad_written(sk[0])
        return (g if need[0] else None), gw, None, None, None, None


def level_embed_add(x, weight, sizes, const=None, batch=None, row0=0):
    """x (B,N,C) | None, const (B|1, N, C) | None (no gradient), weight (>= len(sizes), C): every token of level l gets
    weight[l] added (levels concatenated along N, sizes[l] tokens each; level l takes row row0 + l); batch = B of the result when x is None."""
    if const is not None:
        const = const.detach()
    return _LevelEmbedAdd.apply(x, weight, const, tuple(sizes), batch, row0)


FAN_OUT = os.environ.get('RSCOTR_FAN_OUT', '1') != '0'  # (A/B switch)


class _FanOut(Function):
    """n handles of one tensor for n consumers: the gradients of all of them arrive in ONE backward call and are summed by
    ONE launch per 8 of them (rscotr_sum8, fixed order) instead of by n - 1 pairwise adds of the autograd engine."""

    @staticmethod
    def forward(ctx, x, n):
        ctx.set_materialize_grads(False)
        return tuple(x.view_as(x) for _ in range(n))

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
    if n <= 2 or not FAN_OUT or not (torch.is_tensor(x) and x.requires_grad and x.is_cuda):
        return [x] * n
    return list(_FanOut.apply(x, n))


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
            GRAD_SINK.grad_written(sk[0])
            return (None,) * 11
        return (dw,) + (None,) * 10


def cdn_queries(weight, gt_lab, gt_boxn, slot_src, slot_valid, slot_neg, u, uniform, label_noise_scale, box_noise_scale,
                num_classes):
    """-> (q_label (B,PC,C), q_bbox (B,PC,4)): see include/rscotr.h, rscotr_cdn_queries.  u (B,PC,10)."""
    return _CdnQueries.apply(weight, gt_lab, gt_boxn, slot_src, slot_valid, slot_neg, u, bool(uniform),
                             label_noise_scale * 0.5 if label_noise_scale > 0 else 0.0, max(box_noise_scale, 0.0), num_classes)


def msda_prep(off, logit, reference_points, offset_norm, L, P):
    """off (B,Nq,H*L*P*2) raw sampling offsets, logit (B,Nq,H,L*P) raw attention logits, reference_points
    (B,Nq,L,2|4) (no gradient), offset_norm (L,2) = (W_l,H_l) -> (loc (B,Nq,H,L,P,2), attn (B,Nq,H,L,P))."""
    assert not reference_points.requires_grad, 'reference points are detached on this path'
    B, Nq, H = logit.shape[:3]
    return _MSDAPrep.apply(off.view(B, Nq, H, L * P * 2), logit, reference_points, offset_norm, L, P)


# ==========================================================================================
# Device-library ("plumbing") ops.  These run ATen / hipBLASLt kernels on the GPU and are the
# hook points that hand-written HIP kernels take over one by one (DESIGN.md §kernels keeps the
# list of which op is HIP and which is still library code).  They are NOT a fallback for the
# HIP ops above: the step cannot run without librscotr.so.
# ==========================================================================================
import torch.nn.functional as F  # noqa: E402

LN_EPS = 1e-5


# Set by rscotr_amd.optim.FlatAdamW: object with grad_view(tensor) -> (index, arena view) | None and
# grad_written(index).  When present, backward kernels ADD parameter gradients straight into the flat
# gradient arena (epilogue accumulate) and return None to autograd for them.
GRAD_SINK = None


def _sink(t):
    return None if GRAD_SINK is None or t is None else GRAD_SINK.grad_view(t)


ACT_NONE, ACT_RELU, ACT_GELU, ACT_RELU_GRAD, ACT_GELU_GRAD = 0, 1, 2, 3, 4
_ACT = {None: ACT_NONE, 'relu': ACT_RELU, 'gelu': ACT_GELU}


def _ptr(t):
    return 0 if t is None else t.data_ptr()


class _DeferredCombine:
    """Split-K weight-gradient contractions whose result is ACCUMULATED into the gradient arena leave their slabs in a
    private region and are combined by ONE launch at the end of the backward pass (`flush_deferred`, called by the
    runner / optimizer before anything reads the arena) instead of one combine launch each: ~450 launches per
    co-training round become ~10 (one per task, plus one per repeated use of a shared parameter).  The (slab, destination, shape) table of a pass is static across iterations (slab
    regions are handed out in call order, destinations are arena addresses), so its device copy is cached by content
    and a captured hipGraph replays the same flush."""

    BLOCK = 256 << 20

    def __init__(self):
        self.enabled = os.environ.get('RSCOTR_DEFER_SPLITK', '1') != '0'
        self.blocks, self.cur, self.off = [], 0, 0
        self.entries, self.notify, self.cache = [], [], {}
        self.ln_entries, self.ln_cache = [], {}
        # flush tables are addressed by raw pointer from captured hipGraphs: a table that was looked up while a graph
        # was being warmed up / captured (`pin = True`, set by runner.GraphedTask) is never evicted; the others are
        # dropped oldest-first once more than MAX_TABLES signatures have been seen
        self.pin = False
        self.pinned = set()
        # weight gradients with small outputs are not launched one by one: their operands are kept alive and ONE grouped
        # launch at the end of backward computes them all (rscotr_gemm_dw_group), then the combine below folds the slabs
        self.group_enabled = os.environ.get('RSCOTR_DW_GROUP', '1') != '0'
        self.group_x6 = int(os.environ.get('RSCOTR_DW_GROUP_X6', '0'))  # 0: fp32 tiles only, 1: bf16x6 128 x 128 tiles for interior problems
        self.group, self.group_keep, self.group_cache = [], [], {}
        self.pinned_pool, self.pinned_live = [], []
        self.wattn_entries, self.wattn_cache = [], {}

    MAX_TABLES = 64
    GROUP_MAX_OUT = int(os.environ.get('RSCOTR_DW_GROUP_MAX', 160000))     # M * N of a grouped problem
    GROUP_TARGET_WGS = int(os.environ.get('RSCOTR_DW_GROUP_WGS', 3072))    # workgroups a grouped launch aims at

    def _plan_group(self):
        """Slices and slab regions of the pending grouped problems -> ([(device table, problems, workgroups, variant)],
        combine entries).  Interior problems (M, N multiples of 128, aligned operands) go to the bf16x6 128 x 128 variant of
        the grouped kernel, the rest to the fp32 64 x 64 variant: one launch each."""
        import numpy as np
        probs = self.group

        def kind(p):
            a, b, _, _, _, M, N, K, lda, ldb, _ = p
            ok = self.group_x6 and K % 16 == 0 and K >= 64 and lda % 4 == 0 and ldb % 4 == 0 and a % 16 == 0 and b % 16 == 0
            if self.group_x6 == 3:  # bf16x6 on 64 x 64 tiles, 32 k per step
                return 3 if ok and M % 64 == 0 and N % 64 == 0 and K % 32 == 0 and K >= 512 else 0
            return 2 if ok and M % 128 == 0 and N % 128 == 0 else 0
        kinds = [kind(p) for p in probs]
        tiles = [(M // 128) * (N // 128) if k == 2 else (M // 64) * (N // 64) if k == 3 else ((M + 63) // 64) * ((N + 63) // 64)
                 for k, (_, _, _, _, _, M, N, K, _, _, _) in zip(kinds, probs)]
        # k-slices of about equal WORK per workgroup (a 128 x 128 tile does four times the work of a 64 x 64 one per k)
        work = sum(t * p[7] * (4 if k == 2 else 1) for t, k, p in zip(tiles, kinds, probs))
        klen_t = max(256, -(-work // self.GROUP_TARGET_WGS))
        dev = self.group_keep[0].device
        launches, ents = [], []
        for variant in (0, 2, 3):
            rows = []
            for t, x6, (a, b, out, rs, ks, M, N, K, lda, ldb, kper) in zip(tiles, kinds, probs):
                if x6 != variant:
                    continue
                sp = max(1, -(-K // max(256, klen_t // (4 if x6 == 2 else 1))))
                kq = 32 if x6 == 3 else 16
                klen = -(-(-(-K // sp)) // kq) * kq
                sp = -(-K // klen)
                if sp == 1:
                    klen = K
                slab = self.reserve(sp * (M * N + M) * 4, dev)
                rs_slab = slab + sp * M * N * 4 if rs else 0
                rows.append([a, b, slab, rs_slab, ks, M, N, K, lda, ldb, klen, sp, 0, max(kper, 1), 0, t * sp])
                ents.append((slab, rs_slab, out, rs, M, N, N, sp))
            if rows:
                # bundles of 8 problems of similar size, one problem per XCD (the kernel's id layout): largest first
                rows.sort(key=lambda r: -r[15])
                rows += [[0] * 16 for _ in range(-len(rows) % 8)]
                first = 0
                for b0 in range(0, len(rows), 8):
                    for r in rows[b0:b0 + 8]:
                        r[12] = first
                    first += 8 * rows[b0][15]
                launches.append((self._upload(np.asarray(rows, dtype=np.int64), dev), len(rows), first, variant))
                if os.environ.get('RSCOTR_DW_GROUP_DUMP'):  # (tuning aid: the problems of one grouped launch)
                    print(f'[dw group] variant {variant}: {len(rows)} problems, {first} workgroups, k-slice target {klen_t}')
                    for r in rows:
                        print(f'    M={r[5]} N={r[6]} K={r[7]} klen={r[10]} splits={r[11]} rowsum={int(r[3] != 0)}')
        return launches, ents

    def prepare_capture(self, n=4):
        """Pinned staging buffers for tables that have to be built WHILE a hipGraph is being captured (the grouped launch's
        table holds activation addresses, which differ between the warm-up iterations and the capture): a pageable
        host-to-device copy is not capturable, a pinned one is — and the replayed copy node re-reads the pinned buffer,
        which therefore lives as long as the cache entry."""
        while len(self.pinned_pool) < n:
            self.pinned_pool.append(torch.empty((4096, 16), dtype=torch.int64).pin_memory())

    def _upload(self, arr, dev):
        if dev.type == 'cuda' and torch.cuda.is_current_stream_capturing():
            assert arr.shape[0] <= 4096 and self.pinned_pool, 'DEFER.prepare_capture() must run before a capture'
            host = self.pinned_pool.pop()
            host[:arr.shape[0]].copy_(torch.from_numpy(arr))
            d = torch.empty(arr.shape, dtype=torch.int64, device=dev)
            d.copy_(host[:arr.shape[0]], non_blocking=True)
            self.pinned_live.append(host)
            return d
        return torch.from_numpy(arr).to(dev)

    def _remember(self, cache, sig, hit):
        if self.pin:
            self.pinned.add(sig)
        if sig not in cache:
            cache[sig] = hit
            if len(cache) > self.MAX_TABLES:
                for k in list(cache):
                    if len(cache) <= self.MAX_TABLES:
                        break
                    if k not in self.pinned and k != sig:
                        del cache[k]

    def reserve(self, nbytes, device):
        nbytes = (nbytes + 255) // 256 * 256
        while True:
            if self.cur == len(self.blocks):
                self.blocks.append(torch.empty(max(self.BLOCK, nbytes) // 4, dtype=torch.float32, device=device))
            b = self.blocks[self.cur]
            if self.off + nbytes <= b.numel() * 4:
                ptr = b.data_ptr() + self.off
                self.off += nbytes
                return ptr
            self.cur, self.off = self.cur + 1, 0

    def pending(self):
        return bool(self.entries or self.ln_entries or self.group or self.wattn_entries)

    def drop(self):
        self.entries, self.notify, self.ln_entries = [], [], []
        self.group, self.group_keep = [], []
        self.wattn_entries = []
        self.cur = self.off = 0

    @staticmethod
    def _rounds(entries, dests):
        """Entries that share a destination go to successive launches (the combine is a plain read-add-write)."""
        seen, rounds = {}, []
        for e in entries:
            ds = [d for d in dests(e) if d]
            k = max([seen.get(d, 0) for d in ds] or [0])
            for d in ds:
                seen[d] = k + 1
            while len(rounds) <= k:
                rounds.append([])
            rounds[k].append(e)
        return rounds

    def _flush_ln(self):
        sig = tuple(self.ln_entries)
        hit = self.ln_cache.get(sig)
        if hit is None:
            import numpy as np
            dev = self.blocks[0].device
            hit = []
            for ents in self._rounds(self.ln_entries, lambda e: (e[1], e[2])):
                wg = [(r, c) for r, e in enumerate(ents) for c in range((2 * e[4] + 63) // 64)]
                hit.append((torch.from_numpy(np.asarray(ents, dtype=np.int64)).to(dev),
                            torch.from_numpy(np.asarray(wg, dtype=np.int32)).to(dev), len(wg)))
        self._remember(self.ln_cache, sig, hit)
        for tab, wg, nwg in hit:
            lib.call('rscotr_layernorm_flush', tab.data_ptr(), wg.data_ptr(), nwg, _stream())
        self.ln_entries = []

    def _flush_group(self):
        sig = (tuple(self.group), self.cur, self.off)  # (the slab regions continue where this pass's reserves stand)
        hit = self.group_cache.get(sig)
        if hit is None:
            launches, ents = self._plan_group()
            hit = (launches, ents, self.cur, self.off)
        else:
            self.cur, self.off = hit[2], hit[3]
        self._remember(self.group_cache, sig, hit)
        for table, n, total, variant in hit[0]:
            lib.call('rscotr_gemm_dw_group', table.data_ptr(), n, total, variant, _stream())
        self.entries.extend(hit[1])
        self.group, self.group_keep = [], []

    def _flush_wattn(self):
        """Partial rows of the window-attention backward passes (bias-table / pad-token gradients): one fold launch."""
        sig = tuple(self.wattn_entries)
        hit = self.wattn_cache.get(sig)
        if hit is None:
            import numpy as np
            rows, first = [], 0
            for part, dt, db, heads, C, nrows in self.wattn_entries:
                rows.append((part, dt, db, heads, C, nrows, first) + (0,) * 9)
                first += heads
            hit = (self._upload(np.asarray(rows, dtype=np.int64), self.blocks[0].device), len(rows), first)
        self._remember(self.wattn_cache, sig, hit)
        lib.call('rscotr_swin_wattn_flush', hit[0].data_ptr(), hit[1], hit[2], _stream())
        self.wattn_entries = []

    def flush(self):
        if self.group:
            self._flush_group()
        if self.ln_entries:
            self._flush_ln()
        if self.wattn_entries:
            self._flush_wattn()
        if self.entries:
            sig = tuple(self.entries)
            hit = self.cache.get(sig)
            if hit is None:
                import numpy as np
                # a parameter used several times in one pass (ref_point_head and the shared heads of the DINO decoder:
                # 6-7 contractions into one destination) must not be combined by concurrent workgroups: entry k of a
                # destination goes to launch k
                seen_c, seen_r, rounds = {}, {}, []
                for e in self.entries:
                    k = max(seen_c.get(e[2], 0), seen_r.get(e[3], 0) if e[3] else 0)
                    seen_c[e[2]] = k + 1
                    if e[3]:
                        seen_r[e[3]] = k + 1
                    while len(rounds) <= k:
                        rounds.append([])
                    rounds[k].append(e)
                dev = self.blocks[0].device
                hit = []
                for ents in rounds:
                    tab = np.asarray(ents, dtype=np.int64)
                    wg = []
                    for r, e in enumerate(ents):
                        M, N = e[4], e[5]
                        wg.extend((r, c) for c in range((max(M * N // 4, M) + 255) // 256))
                    hit.append((torch.from_numpy(tab).to(dev), torch.from_numpy(np.asarray(wg, dtype=np.int32)).to(dev), len(wg)))
            self._remember(self.cache, sig, hit)
            for tab, wg, nwg in hit:
                lib.call('rscotr_splitk_flush', tab.data_ptr(), wg.data_ptr(), nwg, _stream())
        notify, self.notify = self.notify, []
        self.entries = []
        self.cur = self.off = 0
        if GRAD_SINK is not None:
            for i in notify:
                GRAD_SINK._on_ready(i)


DEFER = _DeferredCombine()


def flush_deferred():
    """Compute the grouped weight gradients and combine the pending split-K weight gradients / LayerNorm parameter
    gradients into the arena (no-op when nothing is pending)."""
    if DEFER.pending() or DEFER.notify:
        DEFER.flush()


def _try_defer_dw(A, B, out, M, N, K, lda, ldb, rowsum, kscale, krows_per, nws):
    """-> True if the contraction was issued as slabs for the deferred combine."""
    sink = GRAD_SINK
    if sink is None or not DEFER.enabled or SIDE is not None or N % 4 or out.data_ptr() % 16:
        return False
    fg = sink.flat_g
    lo = fg.data_ptr()
    if not (lo <= out.data_ptr() < lo + fg.numel() * 4):
        return False
    if DEFER.group_enabled and M * N <= DEFER.GROUP_MAX_OUT and K >= 16:
        # small output: joins the grouped launch at the end of backward (operands stay alive until then)
        DEFER.group.append((A.data_ptr(), B.data_ptr(), out.data_ptr(), _ptr(rowsum), _ptr(kscale), M, N, K, lda, ldb,
                            int(krows_per)))
        DEFER.group_keep.extend(t for t in (A, B, kscale) if t is not None)
        return True
    if nws == 0:
        return False
    import ctypes
    ptr = DEFER.reserve(nws, A.device)
    splits = ctypes.c_int32(1)
    lib.call('rscotr_gemm_f32_dw_slabs', A.data_ptr(), B.data_ptr(), out.data_ptr(), M, N, K, lda, ldb, N, _ptr(rowsum),
             _ptr(kscale), int(krows_per), ptr, nws, ctypes.byref(splits), _stream())
    sp = splits.value
    if sp > 1:
        DEFER.entries.append((ptr, ptr + sp * M * N * 4 if rowsum is not None else 0, out.data_ptr(), _ptr(rowsum), M, N, N, sp))
    return True


def gemm(A, B, M, N, K, lda, ldb, a_kmajor, b_kmajor, out=None, bias=None, act=ACT_NONE, aux=None, pre=None,
         resid=None, accumulate=False, rowsum=None, rowsum_accumulate=False, rowscale=None, rows_per=0, kscale=None,
         krows_per=0, out2=None):
    """C[m,n] = epilogue(sum_k Aop[m,k] Bop[n,k]) on the fp32 matrix cores (include/rscotr.h,
    rscotr_gemm_f32).  A, B, out are contiguous fp32 device tensors; out (M,N) is allocated here
    unless given.  `rowsum` (M,) (+)= sum_k Aop[m,k] (k-major A only: the bias gradient riding the dW
    contraction).  `out2` (M,N): second output out + resid, `out` itself then stays without the residual.
    Returns out."""
    _chk(A, B, out, bias, aux, pre, resid, rowsum, out2)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    if (rowsum is None and kscale is None and PROFILE is None
            and WPLANES.eligible(A, B, M, N, K, lda, ldb, a_kmajor, b_kmajor)):
        # B is a parameter: multiply with its pre-split bf16 planes (written once per optimizer step)
        planes, npad = WPLANES.get(B, N, K, ldb, b_kmajor)
        nws = lib.rscotr_gemm_f32_wplanes_workspace(M, N, K)
        ws = _WS.get(nws, A.device).data_ptr() if nws else 0
        lib.call('rscotr_gemm_f32_wplanes', A.data_ptr(), planes, npad, out.data_ptr(), M, N, K, lda, N, _ptr(bias), int(act),
                 _ptr(aux), _ptr(pre), _ptr(resid), int(accumulate), _ptr(rowscale), int(rows_per), _ptr(out2), ws, nws,
                 _stream())
        return out
    key = (M, N, K, lib.rscotr_gemm_get_precision())  # the workspace a shape wants depends on the precision mode
    nws = _gemm_ws_bytes.get(key)
    if nws is None:
        nws = _gemm_ws_bytes[key] = lib.rscotr_gemm_f32_workspace(M, N, K)
    if (accumulate and a_kmajor and b_kmajor and bias is None and act == ACT_NONE and resid is None and pre is None
            and rowscale is None and (rowsum is None or rowsum_accumulate) and PROFILE is None
            and _try_defer_dw(A, B, out, M, N, K, lda, ldb, rowsum, kscale, krows_per, nws)):
        return out
    ws = _WS.get(nws, A.device).data_ptr() if nws else 0
    args = (A.data_ptr(), B.data_ptr(), out.data_ptr(), M, N, K, lda, ldb, N, int(a_kmajor), int(b_kmajor),
            _ptr(bias), int(act), _ptr(aux), _ptr(pre), _ptr(resid), int(accumulate), _ptr(rowsum),
            int(rowsum_accumulate), _ptr(rowscale), int(rows_per), _ptr(kscale), int(krows_per), _ptr(out2), ws, nws,
            _stream())
    if PROFILE is None:
        lib.call('rscotr_gemm_f32', *args)
    else:
        with _Prof('gemm', 2 * M * N * K, gemm_kernel_name(M, N, K, a_kmajor, b_kmajor),
                   shape=(M, N, K, int(a_kmajor), int(b_kmajor))):
            lib.call('rscotr_gemm_f32', *args)
    return out


def gemm_kernel_name(M, N, K, a_kmajor, b_kmajor):
    """Name of the kernel instantiation rscotr_gemm_f32 launches for this problem (mirrors the tile
    choice in csrc/gemm.hip; used to label roofline samples so they can be matched with rocprof)."""
    if N <= 32:
        bm, bn, wm, wn = 128, 32, 4, 1
    elif N >= 1024 and M >= 4096 and M % 128 == 0:
        bm, bn, wm, wn = 128, 64, 2, 2
    else:
        bm, bn, wm, wn = 64, 64, 2, 2
    return f'rscotr::gemm_f32_kernel<{bm}, {bn}, {wm}, {wn}, {"true" if a_kmajor else "false"}, ' \
           f'{"true" if b_kmajor else "false"}, *>'


def colsum(X, M, N, out=None, accumulate=False):
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=X.device)
    nws = lib.rscotr_colsum_f32_workspace(M, N)
    ws = _WS.get(nws, X.device)
    lib.call('rscotr_colsum_f32', X.data_ptr(), out.data_ptr(), M, N, N, int(accumulate), ws.data_ptr(), nws,
             _stream())
    return out


class _MLP(Function):
    """y = L_n(act(L_{n-1}(... act(L_1(x))))) [+ identity], L_i(h) = h W_i^T + b_i: every Linear is
    one MFMA GEMM with bias/activation/residual fused in its epilogue; backward folds act' into the
    epilogue of the dX GEMM of the following layer (no separate element-wise passes).
    `out_scale` (B,) or None: per-sample factor on the last layer's output before the identity is added (the
    DropPath of a Swin block folded into its proj / fc2 Linear): forward rides the epilogue, backward the
    epilogue of dH and the operand staging of dW / db.
    args: x, identity (Tensor | None), act code, out_scale, then W_1, b_1, ..., W_n, b_n (b may be None)."""

    @staticmethod
    def forward(ctx, x, identity, act, out_scale, *wb):
        n = len(wb) // 2
        ws, bs = wb[0::2], wb[1::2]
        K0 = x.shape[-1]
        x2 = _f32c(x).reshape(-1, K0)
        M = x2.shape[0]
        id_is_x = identity is x  # mmcv FFN: identity defaults to the input itself
        rows_per = 0
        if out_scale is not None:
            out_scale = _f32c(out_scale)
            assert x.dim() == 3 and out_scale.numel() == x.shape[0]
            rows_per = x.shape[1]
        id2 = None if identity is None else (x2 if id_is_x else _f32c(identity).reshape(M, -1))
        hs, auxs = [x2], []
        h = x2
        for i in range(n):
            W = ws[i] if ws[i].is_contiguous() else ws[i].contiguous()
            N, K = W.shape
            last = i == n - 1
            pre = None
            if not last and act == ACT_GELU:
                pre = torch.empty((M, N), dtype=torch.float32, device=x2.device)
            sc = out_scale if last else None
            h = gemm(h, W, M, N, K, K, K, 0, 0, bias=bs[i], act=ACT_NONE if last else act, pre=pre,
                     resid=id2 if last else None, rowscale=sc, rows_per=rows_per if sc is not None else 0)
            if not last:
                hs.append(h)
                auxs.append(pre if act == ACT_GELU else h)
        ctx.save_for_backward(*hs, *auxs, *ws)
        ctx.out_scale, ctx.rows_per = out_scale, rows_per
        ctx.n, ctx.act, ctx.has_id, ctx.id_is_x = n, act, identity is not None, id_is_x
        ctx.has_bias = [b is not None for b in bs]
        ctx.biases = bs  # parameter handles only (for the gradient sink); not needed as saved tensors
        ctx.x_shape = x.shape
        ctx.id_shape = None if identity is None else identity.shape
        return h.view(*x.shape[:-1], h.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        n, act = ctx.n, ctx.act
        saved = ctx.saved_tensors
        hs, auxs, ws = saved[:n], saved[n:2 * n - 1], saved[2 * n - 1:]
        M = hs[0].shape[0]
        g = _f32c(dy).reshape(M, -1)
        g_out = g
        d_id = g.view(ctx.id_shape) if ctx.has_id and not ctx.id_is_x and ctx.needs_input_grad[1] else None
        grads_wb = [None] * (2 * n)
        gact = ACT_RELU_GRAD if act == ACT_RELU else ACT_GELU_GRAD
        dx = None
        for i in range(n - 1, -1, -1):
            W = ws[i] if ws[i].is_contiguous() else ws[i].contiguous()
            N, K = W.shape
            want_w = ctx.needs_input_grad[4 + 2 * i]
            want_b = ctx.has_bias[i] and ctx.needs_input_grad[5 + 2 * i]
            # the last layer's upstream gradient is s_b * dy: folded into the three contractions that read it
            sc = ctx.out_scale if i == n - 1 else None
            sck = dict(kscale=sc, krows_per=ctx.rows_per) if sc is not None else {}
            scr = dict(rowscale=sc, rows_per=ctx.rows_per) if sc is not None else {}
            rs, rs_acc, skb = None, False, None
            if want_b:
                skb = _sink(ctx.biases[i])
                if skb is None:
                    rs = grads_wb[2 * i + 1] = torch.empty(N, dtype=torch.float32, device=g.device)
                else:  # straight into the gradient arena
                    rs, rs_acc = skb[1], True
            if want_w:
                # dW = g^T h; the bias gradient (column sums of g = row sums of the k-major A) rides along
                sk = _sink(ws[i])
                if sk is None:
                    grads_wb[2 * i] = gemm(g, hs[i], N, K, M, N, K, 1, 1, rowsum=rs, rowsum_accumulate=rs_acc, **sck)
                elif skb is not None or not want_b:  # everything lands in the arena: off the critical path
                    _off_path(lambda g=g, h=hs[i], o=sk[1], rs=rs: gemm(g, h, N, K, M, N, K, 1, 1, out=o, accumulate=True,
                                                                        rowsum=rs, rowsum_accumulate=rs_acc, **sck),
                              g, hs[i], sc)
                    GRAD_SINK.grad_written(sk[0])
                else:
                    gemm(g, hs[i], N, K, M, N, K, 1, 1, out=sk[1], accumulate=True, rowsum=rs,
                         rowsum_accumulate=rs_acc, **sck)
                    GRAD_SINK.grad_written(sk[0])
            elif want_b:
                if sc is not None:
                    raise RuntimeError('out_scale with a bias-only gradient is not supported')
                colsum(g, M, N, out=rs, accumulate=rs_acc)
            if skb is not None:
                GRAD_SINK.grad_written(skb[0])
            if i > 0:
                g = gemm(g, W, M, K, N, N, K, 0, 1, act=gact, aux=auxs[i - 1], **scr)  # dH = (g W) * act'
            elif ctx.needs_input_grad[0]:
                # identity == input: its gradient (dy) rides in this epilogue instead of a separate add
                dx = gemm(g, W, M, K, N, N, K, 0, 1, resid=g_out if ctx.id_is_x else None, **scr).view(ctx.x_shape)
        return (dx, d_id, None, None, *grads_wb)


def mlp(x, layers, act='relu', identity=None, out_scale=None):
    """layers: [(W, b), ...]; activation between layers, none after the last; `identity` (same shape
    as the output) is added in the last epilogue (mmcv FFN add_identity); `out_scale` (B,) multiplies the
    output per sample before that (DropPath)."""
    flat = []
    for w, b in layers:
        flat += [w, b]
    return _MLP.apply(x, identity, _ACT[act], out_scale, *flat)


def linear(x, w, b=None, act=None, resid=None, out_scale=None):
    """F.linear(x, w, b) [* out_scale per sample] (+ resid) on the matrix cores.  Activations belong to `mlp`."""
    if act is not None:
        raise RuntimeError('ops.linear has no activation: use ops.mlp')
    return _MLP.apply(x, resid, ACT_NONE, out_scale, w, b)


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
        if direct and DEFER.enabled and SIDE is None and PROFILE is None:
            # the fold of the per-workgroup partial rows into dgamma / dbeta joins the end-of-pass flush (one launch for
            # all ~55 LayerNorms of a backward pass instead of one each)
            part = DEFER.reserve(nws, x2.device)
            lib.call('rscotr_layernorm_bwd_partials', g.data_ptr(), x2.data_ptr(), _ptr(w), stats[0].data_ptr(),
                     stats[1].data_ptr(), _ptr(dx), _ptr(r), M, C, part, nws, _stream())
            DEFER.ln_entries.append((part, dw_ptr, db_ptr, nws // (8 * C), C))
            GRAD_SINK.grad_written(skw[0])
            if ctx.has_b:
                GRAD_SINK.grad_written(skb[0])
            return (None if dx is None else dx.view(dy.shape)), None, None, None
        ws = _WS.get(nws, x2.device)
        with _Prof('layernorm_bwd', 12 * M * C):
            lib.call('rscotr_layernorm_bwd', g.data_ptr(), x2.data_ptr(), _ptr(w), stats[0].data_ptr(),
                     stats[1].data_ptr(), _ptr(dx), _ptr(r), dw_ptr, db_ptr, M, C, ws.data_ptr(), nws, _stream())
        dxv = None if dx is None else dx.view(dy.shape)
        if direct:  # dgamma / dbeta were accumulated straight into the gradient arena
            GRAD_SINK.grad_written(skw[0])
            if ctx.has_b:
                GRAD_SINK.grad_written(skb[0])
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
            GRAD_SINK.grad_written(skw[0])
            if ctx.has_b:
                GRAD_SINK.grad_written(skb[0])
            return dx, None, None, None, None
        return dx, dwb[0] if w is not None else None, dwb[1] if ctx.has_b else None, None, None


def group_norm_tokens(x, groups, w, b, eps=1e-5):
    """nn.GroupNorm(groups, C) on token layout (B, L, C): statistics per (image, group) over all L tokens."""
    return _GroupNormTokens.apply(x, w, b, groups, eps)


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
    return x.permute(0, 2, 3, 1).reshape(B, H * W, C), (H, W)


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
    return x.view(B, hw[0], hw[1], C).permute(0, 3, 1, 2)


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
            lib.call('rscotr_swin_wattn_fwd', qkv.data_ptr(), _ptr(qkv_b), table.data_ptr(), out.data_ptr(),
                     B, H, W, C, heads, ws, shift, _stream())
        ctx.save_for_backward(qkv, qkv_b, table, out)  # (out: the proj Linear keeps it alive anyway)
        ctx.geom = (B, H, W, C, heads, ws, shift)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, qkv_b, table, out = ctx.saved_tensors
        B, H, W, C, heads, ws, shift = ctx.geom
        dout = _f32c(dout)
        dqkv = torch.empty_like(qkv)
        # the kernel ACCUMULATES the bias-table and pad-token (qkv-bias) gradients: with the gradient sink they go
        # straight into the arena (no zero-filled temporaries, no accumulate-adds afterwards)
        skt = _sink(ctx.table_param) if ctx.needs_input_grad[2] else None
        skb = _sink(ctx.qkvb_param) if (qkv_b is not None and ctx.needs_input_grad[1]) else None
        dtable = None if skt is not None else torch.zeros_like(table)
        dqkv_b = None if (skb is not None or qkv_b is None) else torch.zeros_like(qkv_b)
        dt_ptr = skt[1].data_ptr() if skt is not None else dtable.data_ptr()
        db_ptr = skb[1].data_ptr() if skb is not None else _ptr(dqkv_b)
        nws = lib.rscotr_swin_wattn_bwd_workspace(B, H, W, C, heads)
        if (skt is not None and (skb is not None or qkv_b is None) and DEFER.enabled and SIDE is None
                and PROFILE is None):
            # arena-direct: the fold of the partial rows joins the end-of-pass flush (one launch for all 12 blocks)
            part = DEFER.reserve(nws, qkv.device)
            lib.call('rscotr_swin_wattn_bwd', qkv.data_ptr(), _ptr(qkv_b), table.data_ptr(), dout.data_ptr(),
                     dqkv.data_ptr(), 0, 0, B, H, W, C, heads, ws, shift, out.data_ptr(), part, nws, _stream())
            DEFER.wattn_entries.append((part, dt_ptr, db_ptr, heads, C, nws // (heads * 268 * 4)))
        else:
            with _Prof('swin_wattn_bwd', 4 * B * H * W * 8 * C):
                lib.call('rscotr_swin_wattn_bwd', qkv.data_ptr(), _ptr(qkv_b), table.data_ptr(), dout.data_ptr(),
                         dqkv.data_ptr(), db_ptr, dt_ptr, B, H, W, C, heads, ws, shift, out.data_ptr(),
                         _WS.get(nws, qkv.device).data_ptr(), nws, _stream())
        for sk in (skt, skb):
            if sk is not None:
                GRAD_SINK.grad_written(sk[0])
        return dqkv, dqkv_b, dtable, None, None, None, None, None


def swin_window_attention(x, hw, qkv_w, qkv_b, bias_table, rel_index, proj_w, proj_b, heads, ws, shift,
                          identity=None, out_scale=None):
    """mmdet ShiftWindowMSA + WindowMSA on (B, H*W, C) tokens: qkv GEMM on the real
    tokens, fused window-attention kernel (pad / shift / partition / bias / mask / softmax / PV /
    reverse by index arithmetic), proj GEMM.  `rel_index` is unused: the kernel uses the closed form
    (dy+6)*13 + (dx+6) of the buffer."""
    H, W = hw
    qkv = linear(x, qkv_w, qkv_b)
    o = _SwinWindowAttn.apply(qkv, qkv_b, bias_table, H, W, heads, ws, shift)
    return linear(o, proj_w, proj_b, resid=identity, out_scale=out_scale)  # x + s_b * proj(...): one epilogue


def _attn_ksplits(M, N, K, nb):
    """Slices of the key axis for an attention product with few output tiles (P v, dS k): aim at >= 512
    workgroups, >= 128 keys per slice, K divisible."""
    tiles = ((M + 127) // 128) * nb if N <= 32 else ((M + 63) // 64) * ((N + 63) // 64) * nb
    sp = 1
    while tiles * sp < 512 and K % (sp * 2) == 0 and K // (sp * 2) >= 128 and (K // (sp * 2)) % 16 == 0:
        sp *= 2
    return sp


def gemm_batched(A, B, C, M, N, K, lda, ldb, ldc, a_kmajor, b_kmajor, nb0, nb1, sA, sB, sC, offA=0, offB=0, offC=0,
                 accumulate=False, ksplit=False):
    """nb0*nb1 products of one shape addressed in place (rscotr_gemm_f32_batched); s? = (stride b0, stride b1)
    and off? = element offset of the first problem inside the tensor."""
    _chk(A, B, C)
    flops = 2 * M * N * K * nb0 * nb1
    sp = _attn_ksplits(M, N, K, nb0 * nb1) if (ksplit and not a_kmajor and b_kmajor and not accumulate and offC == 0) else 1
    ws = _WS.get(sp * C.numel() * 4, C.device).data_ptr() if sp > 1 else 0
    args = (A.data_ptr() + 4 * offA, B.data_ptr() + 4 * offB, C.data_ptr() + 4 * offC, M, N, K, lda, ldb, ldc,
            int(a_kmajor), int(b_kmajor), nb0, nb1, sA[0], sA[1], sB[0], sB[1], sC[0], sC[1], int(accumulate), sp, ws,
            C.numel(), _stream())
    if PROFILE is None:
        lib.call('rscotr_gemm_f32_batched', *args)
    else:
        with _Prof('gemm_batched', flops, 'rscotr::gemm_f32_kernel (batched attention products)'):
            lib.call('rscotr_gemm_f32_batched', *args)
    return C


MASK_NONE, MASK_SHARED, MASK_PER_IMAGE, MASK_PER_HEAD = 0, 1, 2, 3


def _linear_param_grad(A, Bm, M, N, K, w_handle, b_handle, row0, want_w, want_b, lda=None):
    """Parameter gradients of y = x W^T + b from A = dy (K rows, M columns as the k-major operand) and Bm = x:
    dW[row0:row0+M] (+)= A^T Bm, db[row0:row0+M] (+)= column sums of A (riding the dW contraction); straight into the
    gradient arena when the parameter is sunk (then nothing is returned for it).  `lda`: row stride of A when it is a column
    block of a wider tensor.  Returns (gw, gb, sink_w, sink_b)."""
    dev = A.device
    skw = _sink(w_handle) if want_w else None
    skb = _sink(b_handle) if want_b else None
    gw = gb = None
    rs, rs_acc = None, False
    if want_b:
        if skb is not None:
            rs, rs_acc = skb[1][row0:row0 + M], True
        else:
            rs = gb = torch.empty(M, dtype=torch.float32, device=dev)
    if want_w:
        if skw is not None and (skb is not None or not want_b):
            _off_path(lambda: gemm(A, Bm, M, N, K, lda or M, N, 1, 1, out=skw[1][row0:row0 + M], accumulate=True,
                                   rowsum=rs, rowsum_accumulate=rs_acc), A, Bm)
        elif skw is not None:
            gemm(A, Bm, M, N, K, lda or M, N, 1, 1, out=skw[1][row0:row0 + M], accumulate=True, rowsum=rs,
                 rowsum_accumulate=rs_acc)
        else:
            gw = gemm(A, Bm, M, N, K, lda or M, N, 1, 1, rowsum=rs, rowsum_accumulate=rs_acc)
    elif want_b:
        assert lda is None or lda == M
        colsum(A, K, M, out=rs, accumulate=rs_acc)
    return gw, gb, skw, skb


MSDA_PACKED_PROJ = os.environ.get('RSCOTR_MSDA_PACKED', '1') != '0'  # (A/B switch)


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
                w_off, b_off, w_aw, b_aw, w_v, b_v, w_o, b_o):
        B, Nq, C = x.shape
        H, D = heads, C // heads
        M = B * Nq
        x2 = _f32c(x).reshape(M, C)
        q2 = x2 if q_pos is None else _f32c(torch.add(x, q_pos)).reshape(M, C)
        v_is_x = value_in is None or value_in is x
        id_is_x = identity is x
        val2 = x2 if v_is_x else _f32c(value_in).reshape(-1, C)
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
        packed = MSDA_PACKED_PROJ and b_off is not None and b_aw is not None and n_off % 4 == 0 and C % 4 == 0
        if packed:
            n3 = n_off + n_aw
            wb = torch.empty(n3 * C + n3, dtype=torch.float32, device=x2.device)
            lib.call('rscotr_pack4', ws[0].data_ptr(), n_off * C, ws[1].data_ptr(), n_aw * C, b_off.data_ptr(), n_off,
                     b_aw.data_ptr(), n_aw, wb.data_ptr(), _stream())
            w_cat = wb[:n3 * C].view(n3, C)
            both = gemm(q2, w_cat, M, n3, C, C, C, 0, 0, bias=wb[n3 * C:])
            loc, attn = _msda_prep_fwd_raw(both, both.view(-1)[n_off:], ref, norm, B, Nq, H, L, P, ld_off=n3, ld_logit=n3)
        else:
            w_cat = None
            off = gemm(q2, ws[0], M, n_off, C, C, C, 0, 0, bias=b_off)
            logit = gemm(q2, ws[1], M, n_aw, C, C, C, 0, 0, bias=b_aw)
            loc, attn = _msda_prep_fwd_raw(off, logit, ref, norm, B, Nq, H, L, P)
        out = _msda_fwd_raw(v.view(B, Nk, H, D), spatial_shapes, lsi, loc, attn)
        id2 = x2 if id_is_x else (None if identity is None else _f32c(identity).reshape(M, C))
        y = gemm(out.view(M, C), ws[3], M, C, C, C, C, 0, 0, bias=b_o, resid=id2)
        ctx.save_for_backward(q2, val2, v, loc, attn, ref, norm, out, spatial_shapes, lsi, *ws)
        ctx.kpm = kpm
        ctx.w_cat = w_cat  # (a temporary of this node: not an autograd-tracked tensor)
        ctx.params = (w_off, b_off, w_aw, b_aw, w_v, b_v, w_o, b_o)  # handles for the gradient sink
        ctx.geom = (B, Nq, Nk, C, H, D, L, P)
        ctx.flags = (v_is_x, id_is_x, q_pos is not None, identity is not None)
        ctx.shapes = (x.shape, None if q_pos is None else q_pos.shape, None if value_in is None else value_in.shape,
                      None if identity is None else identity.shape)
        return y.view(B, Nq, C)

    @staticmethod
    def backward(ctx, dy):
        q2, val2, v, loc, attn, ref, norm, out, spatial_shapes, lsi, w_off, w_aw, w_v, w_o = ctx.saved_tensors
        p_off, pb_off, p_aw, pb_aw, p_v, pb_v, p_o, pb_o = ctx.params
        B, Nq, Nk, C, H, D, L, P = ctx.geom
        v_is_x, id_is_x, has_pos, has_id = ctx.flags
        need = ctx.needs_input_grad
        M, Mk = B * Nq, B * Nk
        n_off, n_aw = H * L * P * 2, H * L * P
        g = _f32c(dy).reshape(M, C)
        sinks = []
        # output projection
        gw_o, gb_o, s1, s2 = _linear_param_grad(g, out.view(M, C), C, C, M, p_o, pb_o, 0, need[18], pb_o is not None and need[19])
        sinks += [s1, s2]
        d_out = gemm(g, w_o, M, C, C, C, C, 0, 1)
        # sampling kernel and the location / softmax arithmetic
        gv, gloc, gattn = _msda_bwd_raw(v.view(B, Nk, H, D), spatial_shapes, lsi, loc, attn, d_out.view(B, Nq, C))
        w_cat = ctx.w_cat
        packed = w_cat is not None
        n3 = n_off + n_aw
        goff, glogit = _msda_prep_bwd_raw(gloc, gattn, attn, ref, norm, B, Nq, H, L, P, packed=packed)
        gv = gv.view(Mk, C)
        if packed:
            both = goff                             # (M, 3 n): [d(offsets) | d(logits)]
            goff, glogit, ldg = both.view(-1), both.view(-1)[n_off:], n3   # (flat aliases: column blocks with row stride 3 n)
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
                GRAD_SINK.grad_written(sk_[0])
        # input gradients, merged in the epilogues
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
                pure = gemm(both, w_cat, M, C, n3, n3, C, 0, 1, out2=d_x if two else None, resid=res)
            else:
                pure = gemm(goff, w_off, M, C, n_off, n_off, C, 0, 1)
                gemm(glogit, w_aw, M, C, n_aw, n_aw, C, 0, 1, out=pure, accumulate=True, out2=d_x if two else None, resid=res)
            if two:
                d_pos = pure
            else:
                d_x = pure if want_x else None
                d_pos = pure if want_pos else None
            if v_is_x and want_x:
                gemm(gv, w_v, Mk, C, C, C, C, 0, 1, out=d_x, accumulate=True)
        elif v_is_x and want_x:
            d_x = gemm(gv, w_v, Mk, C, C, C, C, 0, 1, resid=g if merge_id else None)
        if want_val:
            d_val = gemm(gv, w_v, Mk, C, C, C, C, 0, 1).view(ctx.shapes[2])
        d_id = g.view(ctx.shapes[3]) if (has_id and not merge_id and need[3]) else None
        if id_is_x and not merge_id:
            d_id = None
        return (None if d_x is None else d_x.view(ctx.shapes[0]), None if d_pos is None else d_pos.view(ctx.shapes[1]),
                d_val, d_id, None, None, None, None, None, None, None, None,
                gw_off, gb_off, gw_aw, gb_aw, gw_v, gb_v, gw_o, gb_o)


def msda_attention(x, q_pos, value, identity, key_padding_mask, reference_points, spatial_shapes, level_start_index,
                   offset_norm, heads, L, P, w_off, b_off, w_aw, b_aw, w_v, b_v, w_o, b_o):
    """The whole of mmcv MultiScaleDeformableAttention.forward on batch-first tensors (see _MSDAAttn); `value` None or x
    itself = self-attention over the token map (encoder), `identity` None = no residual, x = the usual one."""
    assert not reference_points.requires_grad, 'reference points are detached on this path'
    if reference_points.shape[-1] not in (2, 4):
        raise ValueError(f'Last dim of reference_points must be 2 or 4, got {reference_points.shape[-1]}')
    if q_pos is not None and q_pos.shape != x.shape:
        q_pos = q_pos.expand_as(x)
    return _MSDAAttn.apply(x, q_pos, value, identity, key_padding_mask, reference_points, spatial_shapes,
                           level_start_index, offset_norm, heads, L, P, w_off, b_off, w_aw, b_aw, w_v, b_v, w_o, b_o)


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
    def forward(ctx, x, q_pos, kx, k_pos, vx, in_w, in_b, out_w, out_b, identity, heads, mask, mask_mode):
        B, Lq, C = x.shape
        hd = C // heads
        dev = x.device
        self_attn = kx is None
        x2 = _f32c(x).reshape(B * Lq, C)
        q2 = x2 if q_pos is None else _f32c(torch.add(x, q_pos)).reshape(B * Lq, C)
        # self-attention with one positional embedding for both sides (query + pos feeds q and k): the q and k projections
        # are one GEMM over the first 2C rows of the packed in_proj weight; q / k are then the column halves of one
        # (B*L, 2C) tensor, addressed in place by the batched products (row stride 2C, element offset C for k)
        fused = self_attn and (k_pos is q_pos)
        if self_attn:
            kx2 = x2
            k2 = q2 if fused else (x2 if k_pos is None else _f32c(torch.add(x, k_pos)).reshape(B * Lq, C))
        else:
            kx2 = _f32c(kx).reshape(-1, C)
            k2 = kx2 if k_pos is None else _f32c(torch.add(kx, k_pos)).reshape(-1, C)
        Lk = k2.shape[0] // B
        v_is_kx = vx is None or vx is (x if self_attn else kx)
        v2 = kx2 if v_is_kx else _f32c(vx).reshape(B * Lk, C)
        id_is_x = identity is x
        in_w, in_b = in_w.contiguous(), in_b.contiguous()
        ldq = 2 * C if fused else C
        if fused:
            q = k = gemm(q2, in_w[:2 * C], B * Lq, 2 * C, C, C, C, 0, 0, bias=in_b[:2 * C])
        else:
            q = gemm(q2, in_w[:C], B * Lq, C, C, C, C, 0, 0, bias=in_b[:C])
            k = gemm(k2, in_w[C:2 * C], B * Lk, C, C, C, C, 0, 0, bias=in_b[C:2 * C])
        v = gemm(v2, in_w[2 * C:], B * Lk, C, C, C, C, 0, 0, bias=in_b[2 * C:])
        P = torch.empty((B, heads, Lq, Lk), dtype=torch.float32, device=dev)
        sq, sk, sp = (Lq * C, hd), (Lk * C, hd), (heads * Lq * Lk, Lq * Lk)
        sqp, skp = (Lq * ldq, hd), (Lk * ldq, hd)  # strides of the projected q / k
        gemm_batched(q, k, P, Lq, Lk, hd, ldq, ldq, Lk, 0, 0, B, heads, sqp, skp, sp, offB=C if fused else 0)
        if mask is not None:
            mask = mask.contiguous()
            assert mask.dtype == torch.bool and mask.is_cuda
        lib.call('rscotr_softmax_mask_fwd', P.data_ptr(), _ptr(mask), int(mask_mode) if mask is not None else 0, B, heads,
                 Lq, Lk, float(hd ** -0.5), _stream())
        o = torch.empty((B * Lq, C), dtype=torch.float32, device=dev)
        gemm_batched(P, v, o, Lq, hd, Lk, Lk, C, C, 0, 1, B, heads, sp, sk, sq, ksplit=True)
        id2 = x2 if id_is_x else (None if identity is None else _f32c(identity).reshape(B * Lq, C))
        y = gemm(o, out_w, B * Lq, C, C, C, C, 0, 0, bias=out_b, resid=id2)
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
        do = gemm(g, out_w, B * Lq, C, C, C, C, 0, 1)
        # attention core
        dv = torch.empty((B * Lk, C), dtype=torch.float32, device=dev)
        gemm_batched(P, do, dv, Lk, hd, Lq, Lk, C, C, 1, 1, B, heads, sp, sq, sk)                 # dV = P^T dO
        dP = torch.empty_like(P)
        gemm_batched(do, v, dP, Lq, Lk, hd, C, C, Lk, 0, 0, B, heads, sq, sk, sp)                  # dP = dO V^T
        lib.call('rscotr_softmax_bwd', P.data_ptr(), dP.data_ptr(), B * heads * Lq, Lk, float(hd ** -0.5), _stream())
        ldq = 2 * C if fused else C
        sqp, skp = (Lq * ldq, hd), (Lk * ldq, hd)
        if fused:  # dq | dk as the column halves of one (B*L, 2C) tensor, like q | k
            dq = dk = torch.empty((B * Lq, 2 * C), dtype=torch.float32, device=dev)
        else:
            dq = torch.empty((B * Lq, C), dtype=torch.float32, device=dev)
            dk = torch.empty((B * Lk, C), dtype=torch.float32, device=dev)
        koff = C if fused else 0
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
                GRAD_SINK.grad_written(sk_[0])

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
                    d_qpos = gemm(dq, w_qk, Mq, C, 2 * C, 2 * C, C, 0, 1, out2=d_x, resid=g if merge_id else None)
                elif want_x or want_qpos:
                    pure = gemm(dq, w_qk, Mq, C, 2 * C, 2 * C, C, 0, 1, resid=g if (merge_id and not want_qpos) else None)
                    d_x = pure if want_x else None
                    d_qpos = pure if want_qpos else None
            else:
                dq_in = gemm(dq, w_q, Mq, C, C, C, C, 0, 1) if (want_x or want_qpos) else None
                dk_in = gemm(dk, w_k, Mk, C, C, C, C, 0, 1) if (want_x or want_kpos) else None
                d_qpos = dq_in if want_qpos else None
                d_kpos = dk_in if want_kpos else None
                if want_x:  # (rare on this path: distinct positional embeddings for the two sides)
                    d_x = dq_in + dk_in
                    if merge_id:
                        d_x = d_x + g
            if v_to_x:
                if d_x is None:
                    d_x = gemm(dv, w_v, Mk, C, C, C, C, 0, 1, resid=g if merge_id else None)
                elif d_x is d_qpos or d_x is d_kpos:  # shared with a positional gradient: must not be modified
                    d_x = gemm(dv, w_v, Mk, C, C, C, C, 0, 1, resid=d_x)
                else:
                    gemm(dv, w_v, Mk, C, C, C, C, 0, 1, out=d_x, accumulate=True)
            elif not v_is_kx and need[4]:
                d_vx = gemm(dv, w_v, Mk, C, C, C, C, 0, 1)
        else:
            want_kx, want_kpos = need[2], has_kpos and need[3]
            if want_x or want_qpos:
                if want_qpos and merge_id:
                    d_x = torch.empty((Mq, C), dtype=torch.float32, device=dev)
                    d_qpos = gemm(dq, w_q, Mq, C, C, C, C, 0, 1, out2=d_x, resid=g)
                else:
                    pure = gemm(dq, w_q, Mq, C, C, C, C, 0, 1, resid=g if merge_id else None)
                    d_x = pure if want_x else None
                    d_qpos = pure if want_qpos else None
            v_to_kx = v_is_kx and want_kx
            dv_in = gemm(dv, w_v, Mk, C, C, C, C, 0, 1) if (v_to_kx or (not v_is_kx and need[4])) else None
            if want_kx or want_kpos:
                if want_kpos and v_to_kx:
                    d_kx = torch.empty((Mk, C), dtype=torch.float32, device=dev)
                    d_kpos = gemm(dk, w_k, Mk, C, C, C, C, 0, 1, out2=d_kx, resid=dv_in)
                elif v_to_kx:
                    d_kx = gemm(dk, w_k, Mk, C, C, C, C, 0, 1, out=dv_in, accumulate=True)
                else:
                    pure = gemm(dk, w_k, Mk, C, C, C, C, 0, 1)
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
                gw_in, gb_in, gw_o, gb_o, shaped(d_id, 5), None, None, None)


def mha(x, kx, vx, in_w, in_b, out_w, out_b, heads, attn_mask=None, identity=None, mask_mode=None, q_pos=None, k_pos=None):
    """torch.nn.MultiheadAttention semantics on batch-first tensors (+ the positional adds and the `identity` residual of
    mmcv's wrapper): query = x + q_pos (B,Lq,C), key = kx + k_pos, value = vx (B,Lk,C); kx None or x itself =
    self-attention (k_pos None then means q_pos), vx None = the key content; attn_mask bool, True = blocked: (Lq,Lk)
    shared, (B,Lq,Lk) per image (mask_mode=MASK_PER_IMAGE) or (B*heads,Lq,Lk)."""
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
    return _MHA.apply(x, q_pos, kx, k_pos, vx, in_w, in_b, out_w, out_b, identity, heads, attn_mask, mask_mode or 0)


# ------------------------------------------------------------------------------------------
# classification / detection / segmentation loss pieces
# ------------------------------------------------------------------------------------------
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


def bbox_cxcywh_to_xyxy(b):
    cx, cy, w, h = b.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def bbox_xyxy_to_cxcywh(b):
    x1, y1, x2, y2 = b.unbind(-1)
    return torch.stack([(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1], dim=-1)


def _giou(b1, b2, aligned, eps=1e-6):
    area1 = (b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1])
    area2 = (b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1])
    if aligned:
        lt, rb = torch.max(b1[..., :2], b2[..., :2]), torch.min(b1[..., 2:], b2[..., 2:])
        elt, erb = torch.min(b1[..., :2], b2[..., :2]), torch.max(b1[..., 2:], b2[..., 2:])
        a1, a2 = area1, area2
    else:
        lt = torch.max(b1[..., :, None, :2], b2[..., None, :, :2])
        rb = torch.min(b1[..., :, None, 2:], b2[..., None, :, 2:])
        elt = torch.min(b1[..., :, None, :2], b2[..., None, :, :2])
        erb = torch.max(b1[..., :, None, 2:], b2[..., None, :, 2:])
        a1, a2 = area1[..., None], area2[..., None, :]
    wh = (rb - lt).clamp(min=0)
    overlap = wh[..., 0] * wh[..., 1]
    union = (a1 + a2 - overlap).clamp(min=eps)
    ious = overlap / union
    ewh = (erb - elt).clamp(min=0)
    earea = (ewh[..., 0] * ewh[..., 1]).clamp(min=eps)
    return ious - (earea - union) / earea


def match_cost(cls_score, bbox_pred, gt_bboxes, gt_labels, img_w, img_h, w_cls, w_l1, w_iou, alpha, gamma, eps):
    """mmdet FocalLossCost + BBoxL1Cost(xywh) + IoUCost(giou) for S prediction sets of one image:
    cls_score (S,Q,C), bbox_pred (S,Q,4) cxcywh normalised, gt (G,4) xyxy pixels -> (S,Q,G)."""
    factor = gt_bboxes.new_tensor([img_w, img_h, img_w, img_h])
    p = cls_score.sigmoid()
    neg = -(1 - p + eps).log() * (1 - alpha) * p.pow(gamma)
    pos = -(p + eps).log() * alpha * (1 - p).pow(gamma)
    c_cls = (pos[..., gt_labels] - neg[..., gt_labels]) * w_cls
    gt_c = bbox_xyxy_to_cxcywh(gt_bboxes / factor)
    c_l1 = (bbox_pred[..., :, None, :] - gt_c[None, None, :, :]).abs().sum(-1) * w_l1
    boxes = bbox_cxcywh_to_xyxy(bbox_pred) * factor
    c_iou = -_giou(boxes, gt_bboxes.unsqueeze(0).expand(boxes.shape[0], -1, -1), aligned=False) * w_iou
    return c_cls + c_l1 + c_iou


def match_cost_batched(cls_score, bbox_pred, gt_bboxes, gt_labels, factors, w_cls, w_l1, w_iou, alpha, gamma, eps):
    """mmdet FocalLossCost + BBoxL1Cost(xywh) + IoUCost(giou) for all images at once on padded ground truth, one
    kernel (rscotr_match_cost): cls_score (S,B,Q,C), bbox_pred (S,B,Q,4), gt_bboxes (B,G,4) xyxy pixels, gt_labels
    (B,G), factors (B,4) = (w,h,w,h) -> (S,B,Q,G).  Columns of padding ground truths hold finite garbage; the
    assignment ignores them."""
    cls_score, bbox_pred, gt_bboxes, factors = _f32c(cls_score), _f32c(bbox_pred), _f32c(gt_bboxes), _f32c(factors)
    gt_labels = gt_labels.contiguous()
    _chk(cls_score, bbox_pred, gt_bboxes, gt_labels, factors)
    S, B, Q, C = cls_score.shape
    G = gt_bboxes.shape[1]
    cost = torch.empty((S, B, Q, G), dtype=torch.float32, device=cls_score.device)
    lib.call('rscotr_match_cost', cls_score.data_ptr(), bbox_pred.data_ptr(), gt_bboxes.data_ptr(), gt_labels.data_ptr(),
             factors.data_ptr(), cost.data_ptr(), S, B, Q, C, G, float(w_cls), float(w_l1), float(w_iou), float(alpha),
             float(gamma), float(eps), _stream())
    return cost


def lsap_batch(flat_cost, rows, cols):
    """Solve len(rows) assignment problems whose fp32 costs are concatenated in `flat_cost`
    (device or host).  ONE device->host copy, then the C-ABI solver (rscotr_lsap_batch_f32).
    Returns (row_inds, col_inds): lists of int64 numpy arrays (row_inds ascending, as SciPy)."""
    import numpy as np
    host = flat_cost.detach().to('cpu', torch.float32).contiguous().numpy()  # the step's one sync
    n = len(rows)
    sizes = np.asarray(rows, dtype=np.int64) * np.asarray(cols, dtype=np.int64)
    offsets = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    outs = np.minimum(rows, cols).astype(np.int64)
    out_off = np.concatenate([[0], np.cumsum(outs)[:-1]]).astype(np.int64)
    total = int(outs.sum())
    r = np.zeros(max(total, 1), dtype=np.int64)
    c = np.zeros(max(total, 1), dtype=np.int64)
    rows_a = np.asarray(rows, dtype=np.int32)
    cols_a = np.asarray(cols, dtype=np.int32)
    assert host.size == int(sizes.sum())
    lib.call('rscotr_lsap_batch_f32', host.ctypes.data, offsets.ctypes.data, rows_a.ctypes.data,
             cols_a.ctypes.data, n, out_off.ctypes.data, r.ctypes.data, c.ctypes.data)
    return ([r[out_off[k]:out_off[k] + outs[k]] for k in range(n)],
            [c[out_off[k]:out_off[k] + outs[k]] for k in range(n)])


def lsap_device(cost, gcount):
    """The matcher's assignment problems solved ON THE DEVICE (rscotr_lsap_dev_f32: SciPy's algorithm and
    tie-breaks in fp64, one wavefront per problem, no host round trip).  cost (P, Q, ld) fp32 with the
    first gcount[p] columns of problem p real; gcount (P,) int32 device.  Returns q_for_gt (P, ld) int32:
    the query assigned to each ground truth, -1 for padding columns."""
    cost = _f32c(cost)
    _chk(cost, gcount)
    assert gcount.dtype == torch.int32 and gcount.is_contiguous()
    P, Q, ld = cost.shape
    out = torch.empty((P, ld), dtype=torch.int32, device=cost.device)
    lib.call('rscotr_lsap_dev_f32', cost.data_ptr(), gcount.data_ptr(), P, Q, ld, out.data_ptr(), _stream())
    return out


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


class _RefineBox(Function):
    @staticmethod
    def forward(ctx, delta, ref, eps):
        delta, ref = _f32c(delta), _f32c(ref)
        _chk(delta, ref)
        out = torch.empty_like(delta)
        lib.call('rscotr_refine_box_fwd', delta.data_ptr(), ref.data_ptr(), out.data_ptr(), delta.numel(), float(eps), _stream())
        ctx.save_for_backward(out, ref)
        ctx.eps = float(eps)
        return out

    @staticmethod
    def backward(ctx, g):
        out, ref = ctx.saved_tensors
        g = _f32c(g)
        dd = torch.empty_like(out) if ctx.needs_input_grad[0] else None
        dr = torch.empty_like(out) if ctx.needs_input_grad[1] else None
        lib.call('rscotr_refine_box_bwd', g.data_ptr(), out.data_ptr(), ref.data_ptr(), _ptr(dd), _ptr(dr), out.numel(),
                 ctx.eps, _stream())
        return dd, dr, None


def refine_box(delta, ref, eps=1e-3):
    """sigmoid(delta + inverse_sigmoid(ref, eps)): one kernel per direction (rscotr_refine_box_*)."""
    return _RefineBox.apply(delta, ref, eps)


class _FocalSum(Function):
    @staticmethod
    def forward(ctx, pred, target, gamma, alpha, weight):
        pred = _f32c(pred)
        target = target.contiguous()
        weight = None if weight is None else _f32c(weight)
        _chk(pred, target, weight)
        S, N, C = pred.shape
        sums = torch.empty(S, dtype=torch.float32, device=pred.device)
        dpred = torch.empty_like(pred)
        lib.call('rscotr_focal_sum', pred.data_ptr(), target.data_ptr(), _ptr(weight), sums.data_ptr(), dpred.data_ptr(),
                 S, N, C, float(gamma), float(alpha), _stream())
        ctx.save_for_backward(dpred)
        return sums

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g.view(-1, 1, 1), None, None, None, None


def sigmoid_focal_loss_sum(pred, target, gamma, alpha, weight=None):
    """mmcv sigmoid_focal_loss (CUDA op semantics) summed per set, one kernel that also leaves the gradient
    (rscotr_focal_sum): pred (S,N,C) logits, target (S,N) int64 in [0,C] with C = background, optional per-sample
    weight (S,N) -> (S,)."""
    return _FocalSum.apply(pred, target, gamma, alpha, weight)


class _BoxLoss(Function):
    @staticmethod
    def forward(ctx, pred, target, weight, factors, eps):
        pred, target, weight, factors = _f32c(pred), _f32c(target), _f32c(weight), _f32c(factors)
        _chk(pred, target, weight, factors)
        S, B, Q, _ = pred.shape
        sums = torch.empty((2, S), dtype=torch.float32, device=pred.device)
        d_l1, d_gi = torch.empty_like(pred), torch.empty_like(pred)
        lib.call('rscotr_box_loss', pred.data_ptr(), target.data_ptr(), weight.data_ptr(), factors.data_ptr(),
                 sums.data_ptr(), d_l1.data_ptr(), d_gi.data_ptr(), S, B, Q, float(eps), _stream())
        ctx.save_for_backward(d_l1, d_gi)
        return sums[0], sums[1]

    @staticmethod
    def backward(ctx, g1, g2):
        d_l1, d_gi = ctx.saved_tensors
        return d_l1 * g1.view(-1, 1, 1, 1) + d_gi * g2.view(-1, 1, 1, 1), None, None, None, None


def box_loss_sums(pred, target, weight, factors, eps=1e-6):
    """L1 (cxcywh, normalised) and GIoU (xyxy in pixels: * factors (B,4)) loss sums per prediction set
    (detr_head.py:392-415), one kernel that also leaves both gradients (rscotr_box_loss):
    pred / target / weight (S,B,Q,4) -> (l1 (S,), giou (S,)); the GIoU weight is the mean of the 4 box weights."""
    return _BoxLoss.apply(pred, target, weight, factors, eps)


class _UpsampleCE(Function):
    @staticmethod
    def forward(ctx, logit, label, ignore_index):
        logit = _f32c(logit)
        label = label.contiguous()
        _chk(logit, label)
        B, C, h, w = logit.shape
        H, W = label.shape[-2:]
        lse = torch.empty((B, H, W), dtype=torch.float32, device=logit.device)
        sums = torch.empty(3, dtype=torch.float32, device=logit.device)
        with _Prof('upsample_ce_fwd', 4 * B * C * h * w + 12 * B * H * W):
            nws = lib.rscotr_upsample_ce_workspace()
            lib.call('rscotr_upsample_ce_fwd', logit.data_ptr(), label.data_ptr(), lse.data_ptr(), sums.data_ptr(),
                     B, C, h, w, H, W, int(ignore_index), _WS.get(nws, logit.device).data_ptr(), nws, _stream())
        ctx.save_for_backward(logit, label, lse)
        ctx.ignore = int(ignore_index)
        ctx.mark_non_differentiable(sums)
        npix = float(B * H * W)
        return sums[0] / npix, sums

    @staticmethod
    def backward(ctx, g_loss, g_sums):
        logit, label, lse = ctx.saved_tensors
        B, C, h, w = logit.shape
        H, W = label.shape[-2:]
        gscale = (g_loss / float(B * H * W)).reshape(1).float().contiguous()
        dlogit = torch.empty_like(logit)
        with _Prof('upsample_ce_bwd', 8 * B * C * h * w + 12 * B * H * W):
            lib.call('rscotr_upsample_ce_bwd', logit.data_ptr(), label.data_ptr(), lse.data_ptr(), gscale.data_ptr(),
                     dlogit.data_ptr(), B, C, h, w, H, W, ctx.ignore, _stream())
        return dlogit, None, None


def upsample_ce(seg_logit, label, ignore_index=255):
    """mmseg BaseDecodeHead.losses: bilinear resize (align_corners=False) to the label size, CE with
    ignore_index averaged over ALL pixels, and top-1 accuracy over non-ignored pixels — fused: the
    upsampled logits are never materialised (rscotr_upsample_ce_*).
    seg_logit (B,C,h,w), label (B,H,W) int64 -> (loss_ce 0-d, acc (1,))."""
    loss, sums = _UpsampleCE.apply(seg_logit, label, ignore_index)
    acc = (sums[1] * 100.0 / (sums[2] + torch.finfo(torch.float32).eps)).reshape(1)
    return loss, acc


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


# ------------------------------------------------------------------------------------------
# distributed scalar helpers (packed: one all-reduce per call site group, no host sync)
# ------------------------------------------------------------------------------------------
def dist_world():
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def dist_mean_tensor(t):
    """reduce_mean of a small device vector (one all-reduce); identity in a single process."""
    if dist_world() == 1:
        return t
    import torch.distributed as dist
    t = t / dist.get_world_size()
    dist.all_reduce(t)
    return t


def dist_mean_vec(values, device):
    """mmdet reduce_mean for a list of host scalars in ONE all-reduce.  Single process: returns the
    python floats unchanged.  Distributed: returns 0-d device tensors (no host sync)."""
    if dist_world() == 1:
        return [float(v) for v in values]
    import torch.distributed as dist
    t = torch.tensor([float(v) for v in values], dtype=torch.float32, device=device)
    dist.all_reduce(t.div_(dist.get_world_size()))
    return list(t.unbind(0))


def clamp_min(x, lo):
    return x.clamp(min=lo) if torch.is_tensor(x) else max(x, lo)
