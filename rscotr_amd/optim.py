"""Optimizer side of the co-training step.

* `build_param_groups` — mtl/utils/optimizer.py:25-55 (`MTLOptimizerConstructor`) on top of mmcv's
  `DefaultOptimizerConstructor.add_params`: ONE group per parameter, `custom_keys` matched as
  substrings of the parameter name, longest key first (ties alphabetical), first match wins.
* `FlatAdamW` — zero_grad -> (backward) -> clip_grad_norm_ -> AdamW.step of mmcv's OptimizerHook
  (mtl/apis/train.py:66-83; cfg ...potsdam.py:203-213) as two HIP launches over flat arenas.
  torch-1.11 semantics are kept: `zero_grad()` zero-fills, so once a parameter has received its
  first gradient it is updated on EVERY step (momentum decay + weight decay on steps of other
  tasks), and parameters that never receive one (e.g. `backbone.norm0.*`) are never touched.
* `StepLrUpdater` — mmcv `StepLrUpdaterHook` as forced to iteration mode by `IterBasedRunner`.
"""
import math

import numpy as np
import torch

from . import ops
from ._lib import lib

CHUNK = 4096  # elements per workgroup in the optimizer kernels (16 KB per arena)


def build_param_groups(model, optimizer_cfg):
    """-> list of dict(name, param, lr, weight_decay) in `named_parameters()` order."""
    if hasattr(model, 'module'):
        model = model.module
    cfg = dict(optimizer_cfg)
    base_lr = cfg['lr']
    base_wd = cfg.get('weight_decay', None)
    pw = cfg.get('paramwise_cfg') or {}
    custom_keys = pw.get('custom_keys', {})
    sorted_keys = sorted(sorted(custom_keys.keys()), key=len, reverse=True)
    bias_lr_mult = pw.get('bias_lr_mult', 1.)
    bias_decay_mult = pw.get('bias_decay_mult', 1.)
    norm_decay_mult = pw.get('norm_decay_mult', 1.)
    norm_types = (torch.nn.modules.batchnorm._BatchNorm, torch.nn.GroupNorm, torch.nn.LayerNorm,
                  torch.nn.modules.instancenorm._InstanceNorm)
    groups, seen = [], set()

    def add(module, prefix):
        is_norm = isinstance(module, norm_types)
        for name, param in module.named_parameters(recurse=False):
            if id(param) in seen:
                continue
            seen.add(id(param))
            full = f'{prefix}.{name}' if prefix else name
            g = dict(name=full, param=param, lr=base_lr, weight_decay=base_wd if base_wd is not None else 0.0)
            if param.requires_grad:
                for key in sorted_keys:
                    if key in f'{prefix}.{name}':
                        g['lr'] = base_lr * custom_keys[key].get('lr_mult', 1.)
                        if base_wd is not None:
                            g['weight_decay'] = base_wd * custom_keys[key].get('decay_mult', 1.)
                        break
                else:
                    if name == 'bias' and not is_norm:
                        g['lr'] = base_lr * bias_lr_mult
                    if base_wd is not None:
                        if is_norm:
                            g['weight_decay'] = base_wd * norm_decay_mult
                        elif name == 'bias':
                            g['weight_decay'] = base_wd * bias_decay_mult
            groups.append(g)
        for child_name, child in module.named_children():
            add(child, f'{prefix}.{child_name}' if prefix else child_name)

    add(model, '')
    return groups


class StepLrUpdater:
    """lr_config = dict(policy='step', step=[...], gamma=0.1) evaluated per iteration."""

    def __init__(self, step, gamma=0.1, min_lr=None, **kwargs):
        self.step, self.gamma, self.min_lr = step, gamma, min_lr

    def factor(self, cur_iter):
        if isinstance(self.step, int):
            exp = cur_iter // self.step
        else:
            exp = len(self.step)
            for i, s in enumerate(self.step):
                if cur_iter < s:
                    exp = i
                    break
        return self.gamma ** exp


class FlatAdamW:
    def __init__(self, groups, betas=(0.9, 0.999), eps=1e-8, grad_clip=None, order=None):
        """groups: output of build_param_groups. order: optional permutation of group indices that
        fixes the arena layout (used to make each task's parameter subset contiguous)."""
        self.groups = groups
        self.betas, self.eps = betas, eps
        self.max_norm = float(grad_clip['max_norm']) if grad_clip else 0.0
        if grad_clip:
            assert grad_clip.get('norm_type', 2) == 2
        order = list(range(len(groups))) if order is None else list(order)
        params = [g['param'] for g in groups]
        device = params[0].device
        self.device = device
        offs, off = {}, 0
        for gi in order:
            offs[gi] = off
            off += (params[gi].numel() + 3) // 4 * 4  # 16-byte aligned segments
        self.total = off
        self.offsets = [offs[i] for i in range(len(groups))]
        ops.WPLANES.reset()  # (plane sets are keyed by arena addresses)
        self.flat_p = torch.zeros(off, dtype=torch.float32, device=device)
        self.flat_g = torch.zeros(off, dtype=torch.float32, device=device)
        self.flat_m = torch.zeros(off, dtype=torch.float32, device=device)
        self.flat_v = torch.zeros(off, dtype=torch.float32, device=device)
        with torch.no_grad():
            for p, o in zip(params, self.offsets):
                n = p.numel()
                self.flat_p[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[o:o + n].view(p.shape)
                p.grad = self.flat_g[o:o + n].view(p.shape)
        # static chunk tables
        seg, coff, clen = [], [], []
        for si, (p, o) in enumerate(zip(params, self.offsets)):
            n = (p.numel() + 3) // 4 * 4
            for c in range(0, n, CHUNK):
                seg.append(si)
                coff.append(o + c)
                clen.append(min(CHUNK, n - c))
        self.nchunks = len(seg)
        self.chunk_seg = torch.tensor(seg, dtype=torch.int32, device=device)
        self.chunk_off = torch.tensor(coff, dtype=torch.int64, device=device)
        self.chunk_len = torch.tensor(clen, dtype=torch.int32, device=device)
        n = len(groups)
        self.steps = np.zeros(n, dtype=np.int64)
        self.live = np.zeros(n, dtype=bool)
        self.base_lr = np.array([g['lr'] for g in groups], dtype=np.float64)
        self.wd = np.array([g['weight_decay'] for g in groups], dtype=np.float64)
        self.lr_factor = 1.0
        self._dyn_host = torch.zeros((n, 8), dtype=torch.float32, pin_memory=device.type == 'cuda')
        self.seg_dyn = torch.zeros((n, 8), dtype=torch.float32, device=device)
        self._dyn_event = None
        self.sumsq = torch.zeros(1 + 1024, dtype=torch.float32, device=device)  # [0] = total, [1:] = partials
        # value ranges of the parameters (bit pattern of max |w| per tensor) for the fp16 split product of the GEMMs:
        # refreshed by the update kernel for every tensor it steps, recomputed from the arena after any other change
        # (words of the operator layer's range buffer, ops.RANGES: one per parameter tensor, at the top of the buffer)
        self.seg_amax_ptr = ops.RANGES.param_region(max(n, 1), device) if device.type == 'cuda' else 0
        self.chunk_amax = torch.zeros(4 * max(self.nchunks, 1), dtype=torch.int32, device=device)  # (scratch: per-wavefront maxima of a chunk)
        self._seg_order = np.argsort(np.asarray(self.offsets, dtype=np.int64), kind='stable')
        self._seg_starts = np.asarray(self.offsets, dtype=np.int64)[self._seg_order]
        self.amax_dirty = True
        # gradient-ready notifications: autograd's AccumulateGrad (post hook) or the direct-write path
        # of rscotr_amd.ops (GRAD_SINK) both end in _on_ready(i)
        self.ready_callbacks = []
        self.written_callbacks = []  # grad_written(i) observers (GradSync: flush the deferred work bucket by bucket)
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(i))
                       for i, p in enumerate(params) if p.requires_grad]
        self._by_ptr = {p.data_ptr(): i for i, p in enumerate(params) if p.requires_grad and p.numel() > 0}
        ops.STATE.grad_sink = self

    def _make_hook(self, i):
        def hook(param):
            self._on_ready(i)
        return hook

    def _on_ready(self, i):
        self.live[i] = True
        for cb in self.ready_callbacks:
            cb(i)

    # ---- GRAD_SINK protocol (rscotr_amd.ops): backward kernels add a parameter's gradient straight
    # into its slice of the (zero-filled) gradient arena instead of returning a tensor that autograd
    # would add with one more element-wise kernel per parameter.
    def is_param_ptr(self, ptr):
        """Does `ptr` point into the flat parameter arena (a parameter, or a slice of one)?"""
        base = self.flat_p.data_ptr()
        return base <= ptr < base + self.flat_p.numel() * 4

    def params_changed(self):
        """The arena was written by something other than the update kernel (checkpoint load, restore, state-dict copy)."""
        self.amax_dirty = True

    def refresh_amax(self):
        if self.device.type == 'cuda':
            lib.call('rscotr_param_amax', self.flat_p.data_ptr(), self.chunk_seg.data_ptr(), self.chunk_off.data_ptr(),
                     self.chunk_len.data_ptr(), self.nchunks, self.seg_amax_ptr, len(self.groups), self.chunk_amax.data_ptr(),
                     ops._stream())
        self.amax_dirty = False

    def amax_slot(self, ptr):
        """Address of the value-range word of the parameter that holds arena address `ptr` (0: not a parameter's memory)."""
        off = (ptr - self.flat_p.data_ptr()) // 4
        if not 0 <= off < self.total:
            return 0
        if self.amax_dirty:
            self.refresh_amax()
        k = int(np.searchsorted(self._seg_starts, off, side='right')) - 1
        return self.seg_amax_ptr + 4 * int(self._seg_order[k])

    def grad_view(self, tensor):
        """-> (index, view of the gradient arena shaped like `tensor`) if `tensor` IS a registered
        parameter or a contiguous reshaped alias of all of it (same storage start and size), else None."""
        i = self._by_ptr.get(tensor.data_ptr())
        if i is None:
            return None
        p = self.groups[i]['param']
        if p.numel() != tensor.numel() or not tensor.is_contiguous():
            return None
        # (a reshaped alias of the whole parameter — a convolution weight flattened to (O, C * k * k) for the GEMM — is the
        # same memory: its gradient goes to the same rows of the arena)
        return i, (p.grad if p.shape == tensor.shape else p.grad.view(tensor.shape))

    def grad_written(self, i):
        # while split-K weight gradients wait for their deferred combine (ops.DEFER), "written" is not true yet for any
        # of them: the notifications are replayed by ops.flush_deferred()
        if ops.DEFER.pending():
            ops.DEFER.notify.append(i)
        else:
            self._on_ready(i)
        for cb in self.written_callbacks:
            cb(i)

    def close(self):
        if ops.STATE.grad_sink is self:
            ops.STATE.grad_sink = None

    def mark_live(self, names):
        name2i = {g['name']: i for i, g in enumerate(self.groups)}
        for n in names:
            self.live[name2i[n]] = True

    def zero_grad(self):
        """torch 1.11 `Optimizer.zero_grad()`: zero-fill (never set to None) — one memset."""
        if ops.DEFER.pending() or ops.DEFER.notify:  # a backward pass that never reached its flush (exception path)
            ops.DEFER.drop()
        self.flat_g.zero_()

    def set_lr_factor(self, f):
        self.lr_factor = float(f)

    def grad_norm(self):
        """Device 0-d tensor with the pre-clip global L2 norm of the last step (no sync)."""
        ops.flush_deferred()
        return self.sumsq.sqrt()[0]

    def new_host_table(self):
        """A pinned per-segment table of its own for a captured iteration: the captured H2D copy reads it
        asynchronously, so it must not be the table the next (eager) iteration refills."""
        return torch.zeros_like(self._dyn_host).pin_memory() if self.device.type == 'cuda' else torch.zeros_like(self._dyn_host)

    def prepare_step(self, table=None):
        """Host half of a step: advance the per-tensor step counts of the live tensors and fill the
        pinned per-segment table {lr, wd, 1/bc1, 1/sqrt(bc2), live}.  No device work."""
        live = self.live
        self.steps[live] += 1
        b1, b2 = self.betas
        t = np.maximum(self.steps, 1).astype(np.float64)
        if table is None and self._dyn_event is not None:
            # the shared pinned table is read by the previous step's asynchronous upload: do not refill it under that copy
            # (captured iterations own a table each and guard it with their `done` event)
            self._dyn_event.synchronize()
        dyn = (self._dyn_host if table is None else table).numpy()
        dyn[:, 0] = self.base_lr * self.lr_factor
        dyn[:, 1] = self.wd
        dyn[:, 2] = 1.0 / (1.0 - b1 ** t)
        dyn[:, 3] = 1.0 / np.sqrt(1.0 - b2 ** t)
        dyn[:, 4] = live.astype(np.float32)

    def launch_step(self, table=None):
        """Device half: table upload + global-norm pass + fused clip/AdamW pass (capturable)."""
        ops.flush_deferred()  # (no-op unless a caller skipped the flush after backward)
        b1, b2 = self.betas
        self.seg_dyn.copy_(self._dyn_host if table is None else table, non_blocking=True)
        if table is None and self.device.type == 'cuda':
            self._dyn_event = torch.cuda.Event()
            self._dyn_event.record()
        s = ops._stream()
        if self.max_norm > 0:
            lib.call('rscotr_grad_sumsq', self.flat_g.data_ptr(), self.chunk_seg.data_ptr(), self.chunk_off.data_ptr(),
                     self.chunk_len.data_ptr(), self.seg_dyn.data_ptr(), self.nchunks, self.sumsq.data_ptr(), s)
        ops.WPLANES.bump(by_optimizer=True)  # the parameters change: their pre-split planes are stale from here on
        if self.amax_dirty:  # (the words of the tensors this step does not touch must be valid too)
            self.refresh_amax()
        lib.call('rscotr_adamw_clip_step_r', self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(),
                 self.flat_v.data_ptr(), self.chunk_seg.data_ptr(), self.chunk_off.data_ptr(), self.chunk_len.data_ptr(),
                 self.seg_dyn.data_ptr(), self.nchunks, self.sumsq.data_ptr(), self.max_norm, float(b1), float(b2),
                 float(self.eps), self.seg_amax_ptr, len(self.groups), self.chunk_amax.data_ptr(), s)

    def step(self):
        self.prepare_step()
        self.launch_step()

    # checkpoint interop with torch.optim.AdamW's state layout
    def state_dict(self):
        state = {}
        for i, (g, o) in enumerate(zip(self.groups, self.offsets)):
            n = g['param'].numel()
            if self.live[i]:
                state[i] = dict(step=int(self.steps[i]), exp_avg=self.flat_m[o:o + n].view(g['param'].shape).clone(),
                                exp_avg_sq=self.flat_v[o:o + n].view(g['param'].shape).clone())
        # the param_groups of torch.optim.AdamW (torch 1.11) as mmcv leaves them: `initial_lr` is what mmcv's LrUpdaterHook
        # setdefault()s on resume — without it a reference resume past a decay step would take the decayed `lr` as the
        # base and decay it a second time
        return dict(state=state, param_groups=[dict(lr=g['lr'] * self.lr_factor, initial_lr=g['lr'],
                                                     weight_decay=g['weight_decay'], betas=self.betas, eps=self.eps,
                                                     amsgrad=False, maximize=False, params=[i])
                                                for i, g in enumerate(self.groups)])

    def snapshot(self):
        """Weights, moments and step counts (GraphedTask restores them after its warm-up iterations)."""
        return dict(p=self.flat_p.clone(), m=self.flat_m.clone(), v=self.flat_v.clone(), steps=self.steps.copy(),
                    live=self.live.copy())

    def restore(self, snap):
        ops.WPLANES.bump()
        self.amax_dirty = True
        self.flat_p.copy_(snap['p'])
        self.flat_m.copy_(snap['m'])
        self.flat_v.copy_(snap['v'])
        self.steps[:] = snap['steps']
        self.live[:] = snap['live']
        self.refresh_amax()  # (now, not lazily: the next thing may be a hipGraph capture, which would record the refresh)



    def load_state_dict(self, sd):
        """Inverse of state_dict(): also accepts a torch.optim.AdamW state dict saved by the reference (one param group
        per parameter in the same order — mmcv's DefaultOptimizerConstructor builds exactly that, mtl/utils/optimizer.py)."""
        state = sd['state']
        self.steps[:] = 0
        self.live[:] = False
        self.flat_m.zero_()
        self.flat_v.zero_()
        with torch.no_grad():
            for i, st in state.items():
                i = int(i)
                g, o = self.groups[i], self.offsets[i]
                n = g['param'].numel()
                assert tuple(st['exp_avg'].shape) == tuple(g['param'].shape), (g['name'], tuple(st['exp_avg'].shape))
                self.flat_m[o:o + n].copy_(st['exp_avg'].reshape(-1))
                self.flat_v[o:o + n].copy_(st['exp_avg_sq'].reshape(-1))
                self.steps[i] = int(st['step'])
                self.live[i] = True


def task_major_order(groups):
    """Arena order that makes each task's parameter subset a few contiguous ranges:
    backbone | neck | shared_encoder | bbox_head | seg_head | cls_head (cls = backbone+cls_head,
    det = backbone+neck+encoder+bbox_head, seg = backbone+neck+encoder+seg_head)."""
    rank = dict(backbone=0, neck=1, shared_encoder=2, bbox_head=3, seg_head=4, cls_head=5)
    idx = list(range(len(groups)))
    idx.sort(key=lambda i: (rank.get(groups[i]['name'].split('.')[0], 6), i))
    return idx


def build_optimizer(model, optimizer_cfg, optimizer_config=None):
    cfg = dict(optimizer_cfg)
    assert cfg.get('type', 'AdamW') == 'AdamW', 'the MTL configs train with AdamW'
    groups = build_param_groups(model, cfg)
    grad_clip = (optimizer_config or {}).get('grad_clip', None)
    return FlatAdamW(groups, betas=tuple(cfg.get('betas', (0.9, 0.999))), eps=cfg.get('eps', 1e-8),
                     grad_clip=grad_clip, order=task_major_order(groups))
