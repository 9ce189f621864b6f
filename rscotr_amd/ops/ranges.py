"""Value ranges of GEMM operands for the fp16 split product (`rscotr_gemm_f32_r`, include/rscotr.h).

The split product scales each operand by a power of two taken from the bit pattern of max |x| over the tensor — a device
word ("slot") that whoever PRODUCED the tensor writes next to it: the epilogue of the product that computed it (`amax_out`),
the LayerNorm kernels, the optimizer for parameters (`FlatAdamW.seg_amax`).  A tensor whose producer writes no slot is
measured by one launch of `rscotr_amax_f32` on first use.

A slot travels WITH the tensor object (attribute `_rs_amax` = (generation, slot address)): a new tensor at a recycled
address has no attribute, and autograd nodes hand slots on explicitly (ctx fields) — nothing is looked up by address.  Any
upper bound of the true maximum is a correct range (the scaled maximum sits 8 x below fp16's largest value, 26 binades above
the point where precision starts to taper), so the slot of a whole tensor also serves its row / column blocks.

Slots are words of ONE device buffer handed out in call order and zeroed by one memset at the start of an iteration
(`RANGES.begin()`, called by `MTL.forward`): the sequence — and with it every address baked into a captured hipGraph — is
the same in every iteration of a task.  `begin()` starts a new generation: a tensor that outlives an iteration (cached
positional encodings, static graph inputs) is measured again in the next one."""
import os

import torch

from .core import _stream, lib
from .state import STATE


class _Ranges:
    PLANES, STRIDE = 32, 1 << 14  # RSCOTR_RANGE_PLANES / RSCOTR_RANGE_STRIDE of include/rscotr.h: a word = 32 sub-words

    def __init__(self):
        # ON by default since the end of round 5 (RSCOTR_GEMM_H3=0 = the six-term bf16 product of rounds 2-4).  Measured on the
        # step (profiles/r5_h3_in_step.txt): the fp16 split product takes 1.2-1.4 ms per round off the GEMM kernels at a lower
        # error (2.5e-7 against 6e-7 of max |C|); the range bookkeeping gave all of it back while ~540 operands per round were
        # measured by launches of their own, and is down to ~60 measuring launches + one grouped launch per backward pass now
        # that the producers (GEMM epilogues, LayerNorm family, deformable-attention backward, AdamW) write the words: the
        # round is as fast as with the bf16 product (34.70 against 34.70-34.90 ms) — and the 800 x 800 det step, whose median
        # distance from the fp64 evaluation was 5 x the fp32 oracle's with the bf16 product, has all 484 gradient tensors
        # within 1e-3 of the fp32 oracle (tests/test_sizes_gpu.py).
        self.enabled = os.environ.get('RSCOTR_GEMM_H3', '1') != '0'
        self.buf = None
        self.base = 0
        self.n = 0
        self.gen = 1
        self.route = {}
        self.route2 = {}  # (... and on the kernel that takes a weight operand from pre-split planes: ops.matmul.HPLANES)
        self.check = os.environ.get('RSCOTR_RANGES_CHECK') == '1'
        self.log = os.environ.get('RSCOTR_RANGES_STATS') == '1'
        self.sites = {}
        self.stats = dict(measured=0, carried=0, params=0)
        # called whenever the slots are zeroed / handed out again (begin(), a wrap of new_slot): whoever holds raw slot
        # addresses outside a tensor attribute (DEFER's pending grouped problems) must forget them
        self.on_invalidate = []

    # ---- slots
    def _ensure(self, device):
        if self.buf is None or self.buf.device != device:
            self.buf = torch.zeros((self.PLANES, self.STRIDE), dtype=torch.int32, device=device)
            self.base = self.buf.data_ptr()
            self.n = 0
            self.dyn = self.STRIDE  # words [dyn, STRIDE) belong to the optimizer (parameters: never zeroed here)

    def param_region(self, n, device):
        """-> address of n consecutive words that begin() leaves alone (FlatAdamW keeps the parameters' ranges there; their
        sub-words 1.. stay zero, sub-word 0 is what the optimizer kernels write)."""
        self._ensure(device)
        assert n < self.STRIDE // 2, 'too many parameter tensors for the range buffer'
        self.dyn = self.STRIDE - n
        self.buf[:, self.dyn:].zero_()
        return self.base + 4 * self.dyn

    def begin(self, device=None):
        """Start of an iteration: every slot free and zero (one memset on the current stream)."""
        if not self.enabled:
            return
        if self.log and any(self.stats.values()):
            import sys
            print(f'[ranges] previous iteration: {self.stats}, slots {self.n}', file=sys.stderr, flush=True)
            for k, v in sorted(self.sites.items(), key=lambda kv: -kv[1]):
                print(f'[ranges]   measured {v:4d} x  {k}', file=sys.stderr, flush=True)
            self.sites = {}
            self.stats = {k: 0 for k in self.stats}
        if self.buf is None:
            if device is None or device.type != 'cuda':
                return
            self._ensure(device)
        elif self.n:
            self._zero()
        self.n = 0
        self.gen += 1
        for cb in self.on_invalidate:
            cb()

    def _zero(self):
        # (one launch: the used words of all planes; at least 1024 so that the shape — and the captured node — is stable)
        self.buf[:, :min(self.dyn, max(1024, (self.n + 1023) // 1024 * 1024))].zero_()

    def new_slot(self, device):
        """Address of a fresh (zero) slot."""
        self._ensure(device)
        if self.n >= self.dyn:  # (only code that never calls begin(): op-level tests, long eager loops)
            self._zero()
            self.n = 0
            self.gen += 1
            for cb in self.on_invalidate:
                cb()
        self.n += 1
        return self.base + 4 * (self.n - 1)

    def out_slot(self, device):
        """A fresh slot for a producer's output, or 0 when ranges are off (producers then skip the bookkeeping)."""
        return self.new_slot(device) if self.enabled else 0

    def index(self, slot):
        return (slot - self.base) // 4

    # ---- tensors
    def tag(self, t, slot):
        """`slot` holds (or will hold, in stream order) an upper bound of max |t|."""
        if slot and t is not None:
            t._rs_amax = (self.gen, slot)
        return t

    def slot_of(self, t):
        a = getattr(t, '_rs_amax', None)
        return a[1] if a is not None and a[0] == self.gen else 0

    def saved(self, t):
        """(generation, slot) of `t` for an autograd ctx (saved tensors come back as new objects without the attribute).  A slot
        must never be re-stamped with the generation current at backward time: a `begin()` between forward and backward (a
        second forward, gradient accumulation, an evaluation in between) has zeroed the word or handed it to another tensor."""
        a = getattr(t, '_rs_amax', None)
        return a if a is not None and a[0] == self.gen else None

    def restore(self, t, saved):
        """Hand a slot saved by `saved()` back to the (unpacked) tensor — only within the generation that wrote it; else the
        tensor stays untagged and is measured where a product needs its range."""
        if saved is not None and saved[0] == self.gen and t is not None:
            t._rs_amax = saved
        return t

    def untag(self, t):
        """`t` is about to be written by something that commits no range word: a bound it carried no longer holds."""
        if t is not None and getattr(t, '_rs_amax', None) is not None:
            del t._rs_amax

    def carry(self, src, dst):
        """dst is a view / reshape / alias of src (same values): hand the slot on."""
        a = getattr(src, '_rs_amax', None)
        if a is not None and dst is not src:
            dst._rs_amax = a
        return dst

    def of(self, t, rows, cols, ld, ptr=None):
        """-> slot address with an upper bound of max |t[r, c]|, r < rows, c < cols, row stride ld (at `ptr`, default the
        tensor's first element): carried by the tensor, kept by the optimizer for parameters, or measured now."""
        s = self.slot_of(t)
        if s:
            self.stats['carried'] += 1
            if self.check:
                self._verify(t, rows, cols, ld, t.data_ptr() if ptr is None else ptr, s, 'carried')
            return s
        p = t.data_ptr() if ptr is None else ptr
        sink = STATE.grad_sink
        if sink is not None and sink.is_param_ptr(p):
            s = sink.amax_slot(p)
            if s:
                self.stats['params'] += 1
                if self.check:
                    self._verify(t, rows, cols, ld, p, s, 'parameter')
                return s
        s = self.new_slot(t.device)
        lib.call('rscotr_amax_f32', p, rows, cols, ld, s, _stream())
        self.stats['measured'] += 1
        if self.log:
            import traceback
            fr = [f for f in traceback.extract_stack(limit=12) if 'ranges.py' not in f.filename and f.name not in ('gemm', '_dw_ranges', '_try_defer_dw')]
            key = f'({rows} x {cols}) ' + ' < '.join(f'{f.name}:{f.lineno}' for f in reversed(fr[-4:]))
            self.sites[key] = self.sites.get(key, 0) + 1
        if ptr is None or ptr == t.data_ptr():
            t._rs_amax = (self.gen, s)
        return s

    def _verify(self, t, rows, cols, ld, p, slot, what):
        """RSCOTR_RANGES_CHECK=1 (debugging aid, synchronises): the slot a tensor carries must bound what it holds NOW.
        (Not while a hipGraph is being captured — a synchronisation there invalidates the capture; the eager warm-up
        iterations in front of every capture run the same code with the check on.)"""
        if torch.cuda.is_current_stream_capturing():
            return
        probe = torch.zeros((self.PLANES, self.STRIDE), dtype=torch.int32, device=t.device)
        lib.call('rscotr_amax_f32', p, rows, cols, ld, probe.data_ptr(), _stream())
        torch.cuda.current_stream().synchronize()
        idx = (slot - self.base) // 4
        assert 0 <= idx < self.STRIDE and self.buf is not None, 'a range word outside the range buffer'
        bound = self.decode(self.buf[:, idx])
        actual = self.decode(probe[:, 0])
        self.stats['checked'] = self.stats.get('checked', 0) + 1
        if actual[0] > bound[0]:
            raise AssertionError(f'stale value range ({what}): the tensor of shape {tuple(t.shape)} holds max |x| in '
                                 f'[{actual[0]}, {actual[1]}) but its slot says [{bound[0]}, {bound[1]})')

    EXP_LO = 80  # kRangeExpLo of csrc/common.h

    @classmethod
    def decode(cls, subwords):
        """The 32 sub-words of a range word (int32 tensor / sequence) -> (lo, hi): 2^e <= max |x| < 2^(e + 1) as floats, (0.0, 0.0) for a word
        nobody marked.  A word is an exponent map (csrc/common.h): byte i of its 128 bytes is set when an element of the tensor has the
        biased fp32 exponent EXP_LO + i."""
        top = -1
        for p, v in enumerate(int(x) & 0xffffffff for x in (subwords.tolist() if hasattr(subwords, 'tolist') else subwords)):
            if v:
                top = max(top, 4 * p + (v.bit_length() - 1) // 8)
        if top < 0:
            return 0.0, 0.0
        e = cls.EXP_LO + top - 127
        return 2.0 ** e, 2.0 ** (e + 1)

    def word(self, slot):
        """(lo, hi) of the word at device address `slot` (synchronises; tests and debugging)."""
        return self.decode(self.buf[:, self.index(slot)])

    # ---- routing
    def wanted(self, M, N, K, lda, ldb, a_kmajor, b_kmajor, act, has_pre, has_rowscale, has_kscale, nws):
        """Does rscotr_gemm_f32 run this product on the split kernels (where the ranges buy the fp16 product)?"""
        if not self.enabled:
            return False
        key = (M, N, K, lda, ldb, a_kmajor, b_kmajor, act, has_pre, has_rowscale, has_kscale, nws, lib.rscotr_gemm_get_precision())
        r = self.route.get(key)
        if r is None:
            r = self.route[key] = bool(lib.rscotr_gemm_f32_split_route(M, N, K, lda, ldb, int(a_kmajor), int(b_kmajor), int(act),
                                                                       int(has_pre), int(has_rowscale), int(has_kscale), nws))
        return r


RANGES = _Ranges()
