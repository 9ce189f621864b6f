"""Plumbing shared by every module of the operator layer: the raw stream handle, the grow-only workspace, the optional
side stream of the weight-gradient contractions, launch-site profiling (HIP events around sampled launches) and the tensor
checks.  There is NO CPU / eager fallback anywhere in `rscotr_amd.ops`: a missing library, a CPU tensor, or a non-zero
return code raises."""
import os

import torch

from .._lib import lib  # noqa: F401 (re-exported)
from .state import STATE


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
        # one buffer per (device, stream): the weight-gradient contractions run on a side stream (STATE.side) next to
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


_SIDE_OBJ = None  # (STATE.side = the active side stream: set by the runner around a replayed iteration's backward)


def side_enable(on=True):
    global _SIDE_OBJ
    if on and _SIDE_OBJ is None:
        _SIDE_OBJ = _Side()  # one stream (and one split-K workspace) for the life of the process
    STATE.side = _SIDE_OBJ if on else None
    return STATE.side


def side_join():
    if STATE.side is not None:
        STATE.side.join()


def _off_path(fn, *keep):
    """Run a weight-gradient contraction: on the side stream when one is active, else inline."""
    if STATE.side is None or STATE.grad_sink is None:
        fn()
    else:
        STATE.side.run(fn, *keep)


# When STATE.profile is a list (bench.py), every launch of a profiled HIP kernel appends
# dict(kind, bytes, e0, e1): HIP events recorded on the launch stream around the kernel and the
# ALGORITHMIC bytes of that launch.  None = no overhead.  STATE.profile_every: record every n-th launch
# of a kind (HIP events cost host time).
_prof_count = {}


class _Prof:
    def __init__(self, kind, nbytes, name=None, shape=None):
        self.rec = None
        if STATE.profile is not None:
            n = _prof_count.get(kind, 0)
            _prof_count[kind] = n + 1
            if n % STATE.profile_every.get(kind, 1) == 0:
                self.rec = dict(kind=kind, bytes=nbytes, name=name, shape=shape,
                                e0=torch.cuda.Event(enable_timing=True), e1=torch.cuda.Event(enable_timing=True))

    def __enter__(self):
        if self.rec is not None:
            self.rec['e0'].record()
        return self

    def __exit__(self, *exc):
        if self.rec is not None:
            self.rec['e1'].record()
            STATE.profile.append(self.rec)
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


# STATE.grad_sink is set by rscotr_amd.optim.FlatAdamW: object with grad_view(tensor) -> (index, arena view) | None and
# grad_written(index).  When present, backward kernels ADD parameter gradients straight into the flat
# gradient arena (epilogue accumulate) and return None to autograd for them.


def _sink(t):
    return None if STATE.grad_sink is None or t is None else STATE.grad_sink.grad_view(t)


ACT_NONE, ACT_RELU, ACT_GELU, ACT_RELU_GRAD, ACT_GELU_GRAD = 0, 1, 2, 3, 4
ACT_RELU_BITS, ACT_RELU_GRAD_BITS = 5, 6  # the ReLU gate as one bit per element (include/rscotr.h: rscotr_gemm_relu_bits_ok)
_ACT = {None: ACT_NONE, 'relu': ACT_RELU, 'gelu': ACT_GELU}


def _ptr(t):
    return 0 if t is None else t.data_ptr()

