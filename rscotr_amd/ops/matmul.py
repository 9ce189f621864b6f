"""Matrix products of the step on the C ABI: `gemm` (rscotr_gemm_f32 and its split-product / weight-plane routes),
`gemm_batched` (attention products addressed in place), the deferred split-K / grouped weight-gradient machinery (`DEFER`),
the pre-split weight planes (`WPLANES`), and the Linear / MLP autograd nodes built on them."""
import os

import torch
from torch.autograd import Function

from .core import (ACT_GELU, ACT_GELU_GRAD, ACT_NONE, ACT_RELU, ACT_RELU_BITS, ACT_RELU_GRAD, ACT_RELU_GRAD_BITS, _ACT, _WS, _Prof,
                   _chk, _f32c, _gemm_ws_bytes,
                   _off_path, _ptr, _sink, _stream, lib)
from .ranges import RANGES
from .state import STATE


class _WeightPlanes:
    """bf16 plane sets of the parameters that serve as the B operand of y = x W^T (and dx = dy W): rscotr_gemm_split_weights
    writes them ONCE per optimizer step and rscotr_gemm_f32_wplanes multiplies fp32 activations with them (include/rscotr.h).
    A parameter is recognised by its address inside the optimizer's flat arena (`STATE.grad_sink.is_param_ptr`); the sets a task
    uses are remembered under the task's name (`begin`), and the first product of an iteration that finds them stale
    re-splits ALL of them in one grouped launch (inside the per-task hipGraph when the iteration is replayed).  `bump()` =
    "the parameters have changed" (optimizer step, checkpoint load, snapshot restore)."""

    def __init__(self):
        # (round 5: with the fp16 split product on, the tiled kernels are as fast as the 128-row weight-plane kernel on its own
        # shapes — 10880 x 256 x 2048: 72 us against 70 — and need no plane sets: the route is only taken with RSCOTR_GEMM_H3=0)
        self.enabled = os.environ.get('RSCOTR_WPLANES', '1') != '0' and not RANGES.enabled
        self.version = 1
        self.entries, self.groups, self.tables = {}, {}, {}
        self.shape_ok = {}
        self.current = None

    def begin(self, group):
        self.current = group
        HPLANES.current = group
        FPLANES.current = group

    def reset(self):
        """Forget every plane set (a new optimizer arena: addresses may be reused by other parameters)."""
        self.entries, self.groups, self.tables = {}, {}, {}
        self.version += 1
        HPLANES.reset()
        FPLANES.reset()

    def bump(self, by_optimizer=False):
        """The parameters have changed.  by_optimizer: by the update kernel itself, which also rewrites their range words;
        any other writer (checkpoint / state-dict load, init_weights, a snapshot restore, a manual copy) leaves the words the
        optimizer keeps stale — a stale-small word overflows the fp16 planes — so they are recomputed on next use."""
        self.version += 1
        HPLANES.version += 1
        FPLANES.version += 1
        if not by_optimizer and STATE.grad_sink is not None:
            STATE.grad_sink.params_changed()

    def eligible(self, A, B, M, N, K, lda, ldb, a_kmajor, b_kmajor, gelu=False):
        if not self.enabled or a_kmajor or STATE.grad_sink is None or K % 16 or N < 64:
            return False
        if lda % 4 or A.data_ptr() % 16 or (b_kmajor and ldb % 4):
            return False
        if lib.rscotr_gemm_get_precision() != 3:
            return False
        # the shape: the 128-row weight-plane kernel's domain (the library decides: rscotr_gemm_f32_wplanes_ok)
        key = (M, N, K, bool(gelu))
        ok = self.shape_ok.get(key)
        if ok is None:
            ok = self.shape_ok[key] = bool(lib.rscotr_gemm_f32_wplanes_ok(M, N, K, int(bool(gelu))))
        return ok and STATE.grad_sink.is_param_ptr(B.data_ptr())

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


class _WeightPlanesH:
    """fp16 planes of the weights that serve as B operand of the interior pipelined 64 x 64 fp16 split kernel (round 5,
    rscotr_gemm_split_weights_h3 / rscotr_gemm_f32_rb): y = x W^T takes the planes of W, dx = dy W those of W^T.  A plane set
    carries the scale of the parameter's range word at the time of the split, and that word only changes in the optimizer
    step: the sets follow WPLANES' version (bump / reset / begin are forwarded from there) and the first product of an
    iteration that finds its task's sets stale re-splits ALL of them in one launch (inside the task's hipGraph when the
    iteration is replayed)."""

    def __init__(self):
        self.enabled = os.environ.get('RSCOTR_HPLANES', '1') != '0'
        self.version = 1
        self.entries, self.groups, self.tables = {}, {}, {}
        self.current = None

    def reset(self):
        self.entries, self.groups, self.tables = {}, {}, {}
        self.version += 1

    def eligible(self, B, M, N, K, lda, ldb, a_kmajor, b_kmajor, act, pre, rowscale, kscale, nws):
        sink = STATE.grad_sink
        if not self.enabled or not RANGES.enabled or a_kmajor or sink is None or K % 32 or ldb % 4 or B.data_ptr() % 16:
            return False
        if not sink.is_param_ptr(B.data_ptr()):
            return False
        key = (M, N, K, lda, ldb, int(a_kmajor), int(b_kmajor), int(act), pre is not None, rowscale is not None, kscale is not None,
               nws, lib.rscotr_gemm_get_precision())
        r = RANGES.route2.get(key)
        if r is None:
            r = RANGES.route2[key] = lib.rscotr_gemm_f32_split_route(M, N, K, lda, ldb, int(a_kmajor), int(b_kmajor), int(act),
                                                                     int(pre is not None), int(rowscale is not None),
                                                                     int(kscale is not None), nws) == 2
        return r

    def get(self, B, N, K, ldb, b_kmajor, word):
        """-> (planes pointer, rpad) of the weight behind operand B (N plane rows, reduction K), fresh."""
        key = (B.data_ptr(), N, K, ldb, int(b_kmajor))
        e = self.entries.get(key)
        if e is None:
            rpad = (N + 63) // 64 * 64
            e = self.entries[key] = dict(planes=torch.empty(rpad * K * 2, dtype=torch.int16, device=B.device), rpad=rpad,
                                         version=0, blocks=(rpad * (K // 32) + 255) // 256, word=int(word))
        keys = self.groups.setdefault(self.current, [])
        if key not in keys:
            keys.append(key)
        if e['version'] != self.version:
            self._refresh(keys, B.device)
        return e['planes'].data_ptr(), e['rpad']

    def _refresh(self, keys, dev):
        stale = tuple(k for k in keys if self.entries[k]['version'] != self.version)
        hit = self.tables.get(stale)
        if hit is None:
            import numpy as np
            rows, first = [], 0
            for (ptr, N, K, ldb, tr) in stale:
                e = self.entries[(ptr, N, K, ldb, tr)]
                # {W, planes, rows of W, cols of W, ldw, rpad, first block, transposed, range word}: a row-major operand B (N, K) is
                # W itself; a k-major one is the (K, N) matrix W whose TRANSPOSE is multiplied
                rows.append((ptr, e['planes'].data_ptr(), K if tr else N, N if tr else K, ldb, e['rpad'], first, tr, e['word']))
                first += e['blocks']
            # (a table first needed while a hipGraph is being captured — a parameter set no warm-up iteration touched — goes through
            #  the pinned staging buffers of the deferred-work tables: a pageable host-to-device copy is not capturable)
            hit = (DEFER._upload(np.asarray(rows, dtype=np.int64), dev), len(rows), first)
            if not (dev.type == 'cuda' and torch.cuda.is_current_stream_capturing()):
                self.tables[stale] = hit  # (a table built inside a capture lives in the graph's private pool: not for later eager calls)
        if os.environ.get('RSCOTR_HPLANES_DEBUG'):
            import sys
            print(f'[hplanes] group {self.current}: re-split {hit[1]} of {len(keys)} sets, {hit[2]} blocks (version {self.version})', file=sys.stderr, flush=True)
        lib.call('rscotr_gemm_split_weights_h3', hit[0].data_ptr(), hit[1], hit[2], _stream())
        for k in stale:
            self.entries[k]['version'] = self.version


class _WeightPlanesF(_WeightPlanesH):
    """FRAGMENT-MAJOR fp16 planes of the weights of a fused FFN (round 6, rscotr_gemm_split_weights_frag / rscotr_ffn_h3,
    csrc/ffn.hip): the weight operand of one wavefront's 16 x 16 x 32 MFMA as one contiguous 1 KB record.  Same life cycle as HPLANES
    (scale of the parameter's range word at the time of the split, re-split with all of the task's sets after an optimizer step)."""

    def get(self, W, tr, word):
        """-> planes pointer for the operand Wop = W (tr = 0: plane rows = rows of W, reduction over its columns) or W^T (tr = 1)."""
        wr, wc = W.shape
        return self.get_raw(W.data_ptr(), wr, wc, tr, word, W.device)

    def get_raw(self, ptr, wr, wc, tr, word, device):
        """... of the contiguous (wr, wc) matrix at `ptr` (a parameter or a row block of one)."""
        key = (ptr, wr, wc, wc, int(tr))
        e = self.entries.get(key)
        if e is None:
            rows, red = (wc, wr) if tr else (wr, wc)
            assert rows % 16 == 0 and red % 32 == 0
            e = self.entries[key] = dict(planes=torch.empty(wr * wc * 2, dtype=torch.int16, device=device), version=0,
                                         blocks=(wr * wc // 8 + 255) // 256, word=int(word))
        keys = self.groups.setdefault(self.current, [])
        if key not in keys:
            keys.append(key)
        if e['version'] != self.version:
            self._refresh(keys, device)
        return e['planes'].data_ptr()

    def _refresh(self, keys, dev):
        stale = tuple(k for k in keys if self.entries[k]['version'] != self.version)
        hit = self.tables.get(stale)
        if hit is None:
            import numpy as np
            rows, first = [], 0
            for key in stale:
                ptr, wr, wc, ldw, tr = key
                e = self.entries[key]
                rows.append((ptr, e['planes'].data_ptr(), wr, wc, ldw, 0, first, tr, e['word']))
                first += e['blocks']
            hit = (DEFER._upload(np.asarray(rows, dtype=np.int64), dev), len(rows), first)
            if not (dev.type == 'cuda' and torch.cuda.is_current_stream_capturing()):
                self.tables[stale] = hit
        lib.call('rscotr_gemm_split_weights_frag', hit[0].data_ptr(), hit[1], hit[2], _stream())
        for k in stale:
            self.entries[k]['version'] = self.version


WPLANES = _WeightPlanes()
HPLANES = _WeightPlanesH()
FPLANES = _WeightPlanesF()


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
        self.group_x6 = int(os.environ.get('RSCOTR_DW_GROUP_X6', '1'))  # 0: every member on the fp32 pipe's 64 x 64 tiles
        self.group_edge = int(os.environ.get('RSCOTR_DW_GROUP_EDGE', 48))  # members with min(M, N) >= this on the split product's 128 x 128 edge body
        self.group, self.group_keep, self.group_cache = [], [], {}
        self.group_amax, self.amax_cache = {}, {}  # operands of grouped problems whose value range is measured at the flush
        self.pinned_pool, self.pinned_live = [], []
        self.captured = []  # (cache, signature) of the tables built during the capture in progress
        self.wattn_entries, self.wattn_cache = [], {}

    MAX_TABLES = 64
    GROUP_MAX_OUT = int(os.environ.get('RSCOTR_DW_GROUP_MAX', 160000))     # M * N of a grouped problem
    GROUP_MAX_OUT_SHORT = int(os.environ.get('RSCOTR_DW_GROUP_MAX_SHORT', 2500000))  # ... with a short reduction (K <= GROUP_SHORT_K)
    GROUP_SHORT_K = int(os.environ.get('RSCOTR_DW_GROUP_SHORT_K', 4096))

    def grouped_size(self, M, N, K):
        return M * N <= self.GROUP_MAX_OUT or (K <= self.GROUP_SHORT_K and M * N <= self.GROUP_MAX_OUT_SHORT)
    GROUP_TARGET_WGS = int(os.environ.get('RSCOTR_DW_GROUP_WGS', 4608))    # workgroups a grouped launch aims at

    def _plan_group(self):
        """Slices and slab regions of the pending grouped problems -> ([(device table, problems, workgroups, variant)],
        combine entries).  Interior problems (M, N multiples of 128, aligned operands) go to the bf16x6 128 x 128 variant of
        the grouped kernel, the rest to the fp32 64 x 64 variant: one launch each."""
        import numpy as np
        probs = [p if len(p) == 13 else tuple(p) + (0, 0) for p in self.group]  # (+ the range slots of the two operands | 0)

        def kind(p):
            k = kind6(p)
            if k == 6 and p[11] and p[12] and RANGES.enabled:
                return 7  # the same body as the fp16 split product: both operands carry their value range
            return k

        def kind6(p):
            a, b, _, _, _, M, N, K, lda, ldb, _ = p[:11]
            ok = (self.group_x6 and K % 16 == 0 and K >= 512 and lda % 4 == 0 and ldb % 4 == 0 and a % 16 == 0 and b % 16 == 0
                  and M % 4 == 0 and N % 4 == 0)
            # members with min(M, N) >= group_edge on the split product's 128 x 128 edge body (one launch), the rest on the fp32
            # pipe's 64 x 64 tiles
            return 6 if ok and self.group_edge and min(M, N) >= abs(self.group_edge) else 0
        kinds = [kind(p) for p in probs]
        tiles = [((M + 127) // 128) * ((N + 127) // 128) if k in (6, 7) else ((M + 63) // 64) * ((N + 63) // 64)
                 for k, (_, _, _, _, _, M, N, K, _, _, _, _, _) in zip(kinds, probs)]
        # k-slices of about equal WORK per workgroup (a 128 x 128 tile does four times the work of a 64 x 64 one per k), per
        # LAUNCH: with one target for the whole pass the few fp32 64 x 64 members of a det backward (the 4- and 20-row
        # reg / cls branches over K = 10880) inherited the k-slice of the big bf16x6 launch and ran as 160 workgroups of
        # K = 3632 each: 260 us for 0.1 GFLOP
        dev = self.group_keep[0].device
        launches, ents = [], []
        for variant in (0, 6, 7):
            work = sum(t * p[7] * (4 if k in (6, 7) else 1) for t, k, p in zip(tiles, kinds, probs) if k == variant)
            klen_t = max(256, -(-work // self.GROUP_TARGET_WGS))
            rows = []
            for t, x6, (a, b, out, rs, ks, M, N, K, lda, ldb, kper, sa, sb) in zip(tiles, kinds, probs):
                if x6 != variant:
                    continue
                sp = max(1, -(-K // max(256, klen_t // (4 if x6 in (6, 7) else 1))))
                kq = 32 if x6 in (6, 7) else 16  # (k-slices of whole steps of the body: the one-stage split loop takes 32 k per barrier pair, the fp32 body 16)
                klen = -(-(-(-K // sp)) // kq) * kq
                sp = -(-K // klen)
                if sp == 1:
                    klen = K
                slab = self.reserve(sp * (M * N + M) * 4, dev)
                rs_slab = slab + sp * M * N * 4 if rs else 0
                rng = ((RANGES.index(sa) + 1) << 32 | (RANGES.index(sb) + 1)) if x6 == 7 else 0
                rows.append([a, b, slab, rs_slab, ks, M, N, K, lda, ldb, klen, sp, 0, max(kper, 1), rng, t * sp])
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
                launches.append((self._upload(np.asarray(rows, dtype=np.int64), dev), len(rows), first, variant,
                                 float(sum(2.0 * r[5] * r[6] * r[7] for r in rows))))
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
            assert arr.size <= 4096 * 16 and self.pinned_pool, 'DEFER.prepare_capture() must run before a capture'
            host = self.pinned_pool.pop()
            stage = host.view(-1)[:arr.size].view(arr.shape)  # (tables of any row width share the (4096, 16) staging buffers)
            stage.copy_(torch.from_numpy(arr))
            d = torch.empty(arr.shape, dtype=torch.int64, device=dev)
            d.copy_(stage, non_blocking=True)
            self.pinned_live.append(host)
            return d
        return torch.from_numpy(arr).to(dev)

    def forget_captured(self):
        """A capture was abandoned (a failed capture, or the ranks' agreement to fall back to the split form): the tables that
        were built while it was being recorded were to be filled by the graph's own copy nodes, which will never run —
        their cache entries must not be found by the next capture, whose private pool hands out the same addresses."""
        for cache, sig in self.captured:
            cache.pop(sig, None)
            self.pinned.discard(sig)
        self.captured = []

    def keep_captured(self):
        """The capture is kept: its tables are refilled by every replay."""
        self.captured = []

    def _remember(self, cache, sig, hit):
        if sig not in cache and torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            self.captured.append((cache, sig))
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
        self.group_amax = {}
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

    def group_range(self, t, rows, cols, ld):
        """Range slot of an operand of a grouped problem: what the tensor carries / the optimizer keeps, else a fresh slot
        that ONE launch fills for all such operands right before the grouped product (`_flush_group`).  Such a slot is
        not handed on with the tensor: nothing may read it before the flush."""
        s = RANGES.slot_of(t)
        if s:
            return s
        sink = STATE.grad_sink
        if sink is not None and sink.is_param_ptr(t.data_ptr()):
            return RANGES.of(t, rows, cols, ld)
        key = (t.data_ptr(), rows, cols, ld)
        s = self.group_amax.get(key)
        if s is None:
            s = self.group_amax[key] = RANGES.new_slot(t.device)
            if os.environ.get('RSCOTR_TRACE_CALLS') == '1':
                import sys
                end = t.storage_offset() * 4 + ((rows - 1) * ld + cols) * 4
                print(f'[group_range] {tuple(t.shape)} rows={rows} cols={cols} ld={ld} last byte {end} of storage {t.untyped_storage().nbytes()}'
                      + ('  <-- OUT OF BOUNDS' if end > t.untyped_storage().nbytes() else ''), file=sys.stderr, flush=True)
        return s

    def _measure_group(self):
        sig = tuple(self.group_amax.items())
        hit = self.amax_cache.get(sig)
        if hit is None:
            import numpy as np
            rows_, first = [], 0
            for (ptr, rows, cols, ld), slot in self.group_amax.items():
                rows_.append((ptr, rows, cols, ld, slot, first))
                first += max(1, min(128, rows * cols // 65536))
            hit = (self._upload(np.asarray(rows_, dtype=np.int64), self.group_keep[0].device), len(rows_), first)
        self._remember(self.amax_cache, sig, hit)
        lib.call('rscotr_amax_group', hit[0].data_ptr(), hit[1], hit[2], _stream())
        RANGES.stats['grouped'] = RANGES.stats.get('grouped', 0) + hit[1]
        self.group_amax = {}

    def _flush_group(self):
        if self.group_amax:
            self._measure_group()
        sig = (tuple(self.group), self.cur, self.off)  # (the slab regions continue where this pass's reserves stand)
        hit = self.group_cache.get(sig)
        if hit is None:
            launches, ents = self._plan_group()
            hit = (launches, ents, self.cur, self.off)
        else:
            self.cur, self.off = hit[2], hit[3]
        self._remember(self.group_cache, sig, hit)
        for table, n, total, variant, flops in hit[0]:
            lib.call('rscotr_gemm_dw_group', table.data_ptr(), n, total, variant, flops, RANGES.base if variant == 7 else 0,
                     _stream())
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
                for e in self.entries:  # (e[2] == 0: row-sum partials only — the bias gradient of a split pass)
                    k = max(seen_c.get(e[2], 0) if e[2] else 0, seen_r.get(e[3], 0) if e[3] else 0)
                    if e[2]:
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
                    nbytes = float(sum((e[7] + 2) * (e[4] * e[5] + e[4]) * 4 for e in ents))  # slabs read, destination read + written
                    hit.append((torch.from_numpy(tab).to(dev), torch.from_numpy(np.asarray(wg, dtype=np.int32)).to(dev), len(wg), nbytes))
            self._remember(self.cache, sig, hit)
            for tab, wg, nwg, nbytes in hit:
                lib.call('rscotr_splitk_flush', tab.data_ptr(), wg.data_ptr(), nwg, nbytes, _stream())
        notify, self.notify = self.notify, []
        self.entries = []
        self.cur = self.off = 0
        if STATE.grad_sink is not None:
            for i in notify:
                STATE.grad_sink._on_ready(i)


DEFER = _DeferredCombine()


def _ranges_invalidated():
    """RANGES.begin() / a wrap of the slot buffer while grouped weight gradients are still pending (gradient accumulation, an
    evaluation forward between backward and the flush): the raw slot addresses they hold are zero words or someone else's now.
    The problems fall back to the member kind that needs no ranges (the six-term bf16 body), the to-be-measured list is dropped."""
    if DEFER.group:
        DEFER.group = [tuple(p[:11]) + (0, 0) for p in DEFER.group]
    DEFER.group_amax = {}


RANGES.on_invalidate.append(_ranges_invalidated)


def flush_deferred():
    """Compute the grouped weight gradients and combine the pending split-K weight gradients / LayerNorm parameter
    gradients into the arena (no-op when nothing is pending)."""
    if DEFER.pending() or DEFER.notify:
        DEFER.flush()


def _dw_ranges(A, B, M, N, K, lda, ldb):
    """Slots of the two k-major operands of a weight-gradient contraction (0, 0 when the fp16 product is off)."""
    if not RANGES.enabled:
        return 0, 0
    return RANGES.of(A, K, M, lda), RANGES.of(B, K, N, ldb)


def _try_defer_dw(A, B, out, M, N, K, lda, ldb, rowsum, kscale, krows_per, nws):
    """-> True if the contraction was issued as slabs for the deferred combine."""
    sink = STATE.grad_sink
    if sink is None or not DEFER.enabled or STATE.side is not None or N % 4 or out.data_ptr() % 16:
        return False
    fg = sink.flat_g
    lo = fg.data_ptr()
    if not (lo <= out.data_ptr() < lo + fg.numel() * 4):
        return False
    if DEFER.group_enabled and DEFER.grouped_size(M, N, K) and K >= 16:
        # small output: joins the grouped launch at the end of backward (operands stay alive until then)
        sa = sb = 0
        if (K >= 512 and K % 16 == 0 and M % 4 == 0 and N % 4 == 0 and min(M, N) >= abs(DEFER.group_edge) and DEFER.group_edge
                and DEFER.group_x6):  # (a member of the split-product launch, DEFER._plan_group: it wants the value ranges)
            sa, sb = (DEFER.group_range(A, K, M, lda), DEFER.group_range(B, K, N, ldb)) if RANGES.enabled else (0, 0)
            lo_r, hi_r = RANGES.base, RANGES.base + 4 * RANGES.STRIDE
            if not (lo_r <= sa < hi_r and lo_r <= sb < hi_r):
                sa = sb = 0
        DEFER.group.append((A.data_ptr(), B.data_ptr(), out.data_ptr(), _ptr(rowsum), _ptr(kscale), M, N, K, lda, ldb,
                            int(krows_per), sa, sb))
        DEFER.group_keep.extend(t for t in (A, B, kscale) if t is not None)
        return True
    if nws == 0:
        return False
    import ctypes
    ptr = DEFER.reserve(nws, A.device)
    splits = ctypes.c_int32(1)
    sa = sb = 0
    if RANGES.wanted(M, N, K, lda, ldb, 1, 1, ACT_NONE, False, False, kscale is not None, nws):
        sa, sb = _dw_ranges(A, B, M, N, K, lda, ldb)
    lib.call('rscotr_gemm_f32_dw_slabs_r', A.data_ptr(), B.data_ptr(), out.data_ptr(), M, N, K, lda, ldb, N, _ptr(rowsum),
             _ptr(kscale), int(krows_per), ptr, nws, ctypes.byref(splits), sa, sb, _stream())
    sp = splits.value
    if sp > 1:
        DEFER.entries.append((ptr, ptr + sp * M * N * 4 if rowsum is not None else 0, out.data_ptr(), _ptr(rowsum), M, N, N, sp))
    return True


def _in_arena(t):
    sink = STATE.grad_sink
    if sink is None or t is None:
        return False
    lo = sink.flat_g.data_ptr()
    return lo <= t.data_ptr() < lo + sink.flat_g.numel() * 4


def gemm(A, B, M, N, K, lda, ldb, a_kmajor, b_kmajor, out=None, bias=None, act=ACT_NONE, aux=None, pre=None,
         resid=None, accumulate=False, rowsum=None, rowsum_accumulate=False, rowscale=None, rows_per=0, kscale=None,
         krows_per=0, out2=None, amax_a=0, amax_b=0, amax_out=0, range_out=True):
    """C[m,n] = epilogue(sum_k Aop[m,k] Bop[n,k]) on the fp32 matrix cores (include/rscotr.h,
    rscotr_gemm_f32).  A, B, out are contiguous fp32 device tensors; out (M,N) is allocated here
    unless given.  `rowsum` (M,) (+)= sum_k Aop[m,k] (k-major A only: the bias gradient riding the dW
    contraction).  `out2` (M,N): second output out + resid, `out` itself then stays without the residual.
    Returns out."""
    _chk(A, B, out, bias, aux, pre, resid, rowsum, out2)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    elif not amax_out:
        # a caller's tensor is (re)written: whatever bound it carried no longer holds; a route below that commits a word tags it anew
        RANGES.untag(out)
    if not amax_out:
        RANGES.untag(out2)
    if (rowsum is None and kscale is None and STATE.profile is None
            and WPLANES.eligible(A, B, M, N, K, lda, ldb, a_kmajor, b_kmajor,
                                 gelu=act in (ACT_GELU, ACT_GELU_GRAD) or pre is not None)):
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
            and rowscale is None and (rowsum is None or rowsum_accumulate) and STATE.profile is None
            and _try_defer_dw(A, B, out, M, N, K, lda, ldb, rowsum, kscale, krows_per, nws)):
        return out
    ws = _WS.get(nws, A.device).data_ptr() if nws else 0
    if RANGES.enabled:
        if not (amax_a and amax_b) and RANGES.wanted(M, N, K, lda, ldb, a_kmajor, b_kmajor, act, pre is not None,
                                                     rowscale is not None, kscale is not None, nws):
            # the split kernels take this product: with the value ranges of both operands it runs as the fp16 split product
            amax_a = amax_a or (RANGES.of(A, K, M, lda) if a_kmajor else RANGES.of(A, M, K, lda))
            amax_b = amax_b or (RANGES.of(B, K, N, ldb) if b_kmajor else RANGES.of(B, N, K, ldb))
        if not amax_out and range_out and RANGE_OUT.enabled_for(out):
            # the range of what this product stores rides out of its epilogue: whoever multiplies with it next finds it
            amax_out = RANGES.new_slot(A.device)
            RANGES.tag(out, amax_out)
            if out2 is not None:
                RANGES.tag(out2, amax_out)
    args = (A.data_ptr(), B.data_ptr(), out.data_ptr(), M, N, K, lda, ldb, N, int(a_kmajor), int(b_kmajor),
            _ptr(bias), int(act), _ptr(aux), _ptr(pre), _ptr(resid), int(accumulate), _ptr(rowsum),
            int(rowsum_accumulate), _ptr(rowscale), int(rows_per), _ptr(kscale), int(krows_per), _ptr(out2), ws, nws,
            int(amax_a), int(amax_b), int(amax_out), _stream())
    entry = 'rscotr_gemm_f32_r'
    if (amax_a and amax_b and rowsum is None
            and HPLANES.eligible(B, M, N, K, lda, ldb, a_kmajor, b_kmajor, act, pre, rowscale, kscale, nws)):
        # B is a parameter and the interior pipelined 64 x 64 kernel takes the product: its pre-split fp16 planes ride along
        planes, rpad = HPLANES.get(B, N, K, ldb, b_kmajor, amax_b)
        args = args[:-1] + (planes, rpad, args[-1])
        entry = 'rscotr_gemm_f32_rb'
    if STATE.profile is None:
        lib.call(entry, *args)
    else:
        with _Prof('gemm', 2 * M * N * K, gemm_kernel_name(M, N, K, a_kmajor, b_kmajor),
                   shape=(M, N, K, int(a_kmajor), int(b_kmajor))):
            lib.call(entry, *args)
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


class _RangeOut:
    """Which products leave the range word of their output (amax_out of rscotr_gemm_f32_r).  The commit is one atomic round trip at
    the end of every workgroup's life (~1 us per launch: profiles/r5_range_word_cost.txt), and only an output that a LATER product
    multiplies with needs the word: callers that know their consumer is a norm, an attention core, the sampling kernel or an
    element-wise merge pass `range_out=False` (ops.linear / ops.gemm).  A tensor that does reach a product without a word is
    measured there (rscotr_amax_f32: correct, one launch; RSCOTR_RANGES_STATS=1 lists them).  RSCOTR_RANGE_OUT_ALL=1: every
    product writes its word, as before."""

    def __init__(self):
        self.all = os.environ.get('RSCOTR_RANGE_OUT_ALL', '0') == '1'
        self.skip_next = False  # set by ops.linear(range_out=False) for the forward of the node it creates

    def enabled_for(self, out):
        return not _in_arena(out)

    def want(self, flag):
        return True if self.all else bool(flag)


RANGE_OUT = _RangeOut()


class _ReluBits:
    """The ReLU gate of a wide FFN as one bit per element (include/rscotr.h, rscotr_gemm_relu_bits_ok): where BOTH the forward
    product h = relu(x W1^T + b) and the gated backward product dH = (g W2) * [h > 0] run on the interior 128 x 128
    split-product tiles, the forward leaves M * N / 8 bytes of gate words and the backward reads those instead of h."""

    def __init__(self):
        self.enabled = os.environ.get('RSCOTR_RELU_BITS', '1') != '0'
        self.cache = {}

    def ok(self, M, N, K, N_next):
        if not self.enabled or not RANGES.enabled:
            return False
        key = (M, N, K, N_next, lib.rscotr_gemm_get_precision())
        r = self.cache.get(key)
        if r is None:
            r = self.cache[key] = bool(lib.rscotr_gemm_relu_bits_ok(M, N, K, K, K, 0, 0)
                                       and lib.rscotr_gemm_relu_bits_ok(M, N, N_next, N_next, N, 0, 1))
        return r


RELU_BITS = _ReluBits()


class _FusedFFN:
    """A two-layer MLP block as ONE launch per direction (rscotr_ffn_h3, csrc/ffn.hip): the encoder FFN (Linear - ReLU - Linear,
    256 -> H -> 256) and the MLP of the Swin blocks of stages 1 and 2 (Linear - GELU - Linear with DropPath, C = 96 / 192):
    forward y = act(x W1^T + b1) W2^T + b2 [* out_scale] (+ identity) with the hidden tensor leaving the kernel for the weight
    gradients only; backward dH = (g W2) * act', dX = dH W1 (+ g) with the mirrored call.  Taken where both weights are
    parameters of the optimizer's arena (their planes and range words live there) and the value ranges are on."""

    MIN_ROWS = int(os.environ.get('RSCOTR_FFN_FUSED_MIN_ROWS', 1024))
    MODE = {(ACT_RELU, 0): 0, (ACT_RELU, 1): 1, (ACT_GELU, 0): 2, (ACT_GELU, 1): 3}

    def __init__(self):
        self.enabled = os.environ.get('RSCOTR_FFN_FUSED', '1') != '0'
        self.gelu = os.environ.get('RSCOTR_FFN_FUSED_GELU', '1') != '0'  # (the Swin route on its own switch: A/B runs)
        self.ln = os.environ.get('RSCOTR_FFN_FUSED_LN', '1') != '0'      # (the norm in front of a Swin MLP as the launch's prologue)
        self.calls = 0
        self.ln_calls = 0

    def ok(self, x2, ws, act, out_scale, sum_with):
        if not self.enabled or not RANGES.enabled or len(ws) != 2 or act not in (ACT_RELU, ACT_GELU) or sum_with is not None:
            return False
        if act == ACT_GELU and not self.gelu:
            return False
        sink = STATE.grad_sink
        (H, C), (C2, H2) = ws[0].shape, ws[1].shape
        M = x2.shape[0]
        if sink is None or STATE.profile is not None or C2 != C or H2 != H or M < self.MIN_ROWS or x2.data_ptr() % 16:
            return False
        if not (sink.is_param_ptr(ws[0].data_ptr()) and sink.is_param_ptr(ws[1].data_ptr())):
            return False
        return bool(lib.rscotr_ffn_h3_ok(M, C, H))

    def ln_ok(self, lz, C, act):
        """Can the forward launch take the LayerNorm in front of the block (a pending ops.norm.LazyNorm) as its prologue?"""
        sink = STATE.grad_sink
        return (self.ln and act == ACT_GELU and C in (96, 192, 384) and lz.w is not None and sink is not None
                and sink.is_param_ptr(lz.w.data_ptr()) and (lz.b is None or sink.is_param_ptr(lz.b.data_ptr())))

    def run(self, x2, W1, b1, W2, b2, act, aux, gate, resid, want_y_range, xscale=None, yscale=None, rows_per=0, ln=None):
        """gate = 0: (W1, W2) are the two Linear weights as stored, (out, in); gate = 1: the mirrored products, W1 := W2 and
        W2 := W1 of the forward, both taken transposed.  aux: the gate bits (ReLU) or the pre-activation (GELU), written by the
        forward call and read by the mirrored one.  ln: a pending LazyNorm whose output x2 is — the launch normalises ln.x2's rows
        itself and fills x2 and the norm's statistics (rscotr_ffn_h3_ln).  -> (hid, y)."""
        M, C = x2.shape
        H = W1.shape[1] if gate else W1.shape[0]
        dev = x2.device
        sink = STATE.grad_sink
        s_x = 0 if ln is not None else RANGES.of(x2, M, C, C)
        s_w1, s_w2 = RANGES.of(W1, W1.shape[0], W1.shape[1], W1.shape[1]), RANGES.of(W2, W2.shape[0], W2.shape[1], W2.shape[1])
        s_b1 = sink.amax_slot(b1.data_ptr()) if (b1 is not None and sink.is_param_ptr(b1.data_ptr())) else 0
        if b1 is not None and not s_b1:
            s_b1 = RANGES.of(b1.view(1, -1), 1, H, H)
        w1f, w2f = FPLANES.get(W1, gate, s_w1), FPLANES.get(W2, gate, s_w2)
        hid = torch.empty((M, H), dtype=torch.float32, device=dev)
        y = torch.empty((M, C), dtype=torch.float32, device=dev)
        s_h = RANGES.new_slot(dev)
        RANGES.tag(hid, s_h)
        s_y = 0
        if want_y_range:
            s_y = RANGES.new_slot(dev)
            RANGES.tag(y, s_y)
        relu = act == ACT_RELU
        splits = int(lib.rscotr_ffn_h3_splits(M, C, H))  # (few rows: partial sums over runs of the hidden width, combined by a second launch)
        ws = torch.empty(splits * M * C, dtype=torch.float32, device=dev) if splits > 1 else None
        if ln is not None:
            assert not gate and not relu and xscale is None and ln.y.data_ptr() == x2.data_ptr()
            lib.call('rscotr_ffn_h3_ln', ln.x2.data_ptr(), M, C, H, _ptr(ln.w), _ptr(ln.b), float(ln.eps), x2.data_ptr(),
                     ln.stats[0].data_ptr(), ln.stats[1].data_ptr(), w1f, _ptr(b1), w2f, _ptr(b2), aux.data_ptr(), hid.data_ptr(),
                     _ptr(resid), y.data_ptr(), _ptr(yscale), int(rows_per), sink.amax_slot(ln.w.data_ptr()),
                     0 if ln.b is None else sink.amax_slot(ln.b.data_ptr()), s_w1, s_w2, s_b1, ln.slot, s_h, s_y, _ptr(ws),
                     0 if ws is None else ws.numel() * 4, _stream())
            ln.done = True
            self.calls += 1
            self.ln_calls += 1
            return hid, y
        lib.call('rscotr_ffn_h3', x2.data_ptr(), M, C, H, w1f, _ptr(b1), w2f, _ptr(b2), self.MODE[(act, int(gate))],
                 aux.data_ptr() if relu else 0, 0 if relu else aux.data_ptr(), hid.data_ptr(), _ptr(resid), y.data_ptr(),
                 _ptr(xscale), _ptr(yscale), int(rows_per), s_x, s_w1, s_w2, s_b1, s_h, s_y, _ptr(ws),
                 0 if ws is None else ws.numel() * 4, _stream())
        self.calls += 1
        return hid, y


FFN_FUSED = _FusedFFN()


class _FusedLinear:
    """ONE Linear on the fused MLP kernel's machinery (rscotr_lin_h3, csrc/ffn.hip: the rows' planes staged once per workgroup, the
    weight as fragment-major planes) for the TALL, NARROW products — Swin stages 1 / 2: the qkv / proj Linears of the window attention,
    PatchMerging's reduction, and their input gradients: 32768 x 96 -> 288 and the like, 25-50 MB for ~1 GFLOP, where the tiled
    kernels re-stage the rows once per column tile.  Taken where the weight is a parameter of the optimizer's arena and the value
    ranges are on; the 256-wide 10880-row Linears of the encoder stay on the tiled kernel (measured: profiles/r6_ffn_lab.txt)."""

    MIN_ROWS = int(os.environ.get('RSCOTR_LIN_FUSED_MIN_ROWS', 8192))
    MAX_NARROW = int(os.environ.get('RSCOTR_LIN_FUSED_NARROW', 192))  # the smaller of (N, K) at most this
    FEW_K = tuple(int(k) for k in os.environ.get('RSCOTR_LIN_FUSED_FEW_K', '384,768').split(','))

    def __init__(self):
        self.enabled = os.environ.get('RSCOTR_LIN_FUSED', '1') != '0'
        self.ln = os.environ.get('RSCOTR_LIN_FUSED_LN', '1') != '0'
        self.few = os.environ.get('RSCOTR_LIN_FUSED_FEW', '1') != '0'
        self.calls = 0
        self.ln_calls = 0

    def ok(self, x2, W, N, K):
        M = x2.shape[0]
        sink = STATE.grad_sink
        if not self.enabled or not RANGES.enabled or sink is None or STATE.profile is not None or x2.data_ptr() % 16 or not W.is_contiguous():
            return False
        tall = M >= self.MIN_ROWS and min(N, K) <= self.MAX_NARROW and max(N, K) <= 576
        # FEW rows with a wide reduction (Swin stages 3 / 4: 2048 x 384 -> 1152 / 384, 512 x 768 -> 2304 / 768, the neck's 1x1
        # convolutions on them): one workgroup per (row tile, 256 columns), all of K staged once — 11 us against 17-27 for the tiled
        # kernels (fp32 pipe at 96-192 workgroups).  The decoders' K = 256 products stay where they are (measured: +1.35 ms per round)
        few = self.few and 512 <= M < self.MIN_ROWS and K in self.FEW_K and N <= 3 * K
        if not (tall or few):
            return False
        return sink.is_param_ptr(W.data_ptr()) and bool(lib.rscotr_lin_h3_ok(M, N, K))

    def ln_ok(self, lz, K):
        sink = STATE.grad_sink
        return (self.ln and K in (96, 192, 384) and lz.w is not None and sink is not None and sink.is_param_ptr(lz.w.data_ptr())
                and (lz.b is None or sink.is_param_ptr(lz.b.data_ptr())))

    def run(self, x2, W, bias, tr, resid, want_y_range, xscale=None, yscale=None, rows_per=0, ln=None):
        """tr = 0: y = x W^T (W (N, K) as stored); tr = 1: y = x W (W (K, N): the input gradient of the Linear).  ln: a pending LazyNorm whose
        output x2 is (rscotr_lin_h3_ln).  -> y (M, N)."""
        M, K = x2.shape
        N = W.shape[1] if tr else W.shape[0]
        dev = x2.device
        sink = STATE.grad_sink
        s_x = 0 if ln is not None else RANGES.of(x2, M, K, K)
        s_w = RANGES.of(W, W.shape[0], W.shape[1], W.shape[1])
        wf = FPLANES.get(W, tr, s_w)
        y = torch.empty((M, N), dtype=torch.float32, device=dev)
        s_y = 0
        if want_y_range:
            s_y = RANGES.new_slot(dev)
            RANGES.tag(y, s_y)
        if ln is not None:
            assert xscale is None and ln.y.data_ptr() == x2.data_ptr()
            lib.call('rscotr_lin_h3_ln', ln.x2.data_ptr(), M, N, K, _ptr(ln.w), _ptr(ln.b), float(ln.eps), x2.data_ptr(),
                     ln.stats[0].data_ptr(), ln.stats[1].data_ptr(), wf, _ptr(bias), _ptr(resid), y.data_ptr(), _ptr(yscale),
                     int(rows_per), sink.amax_slot(ln.w.data_ptr()), 0 if ln.b is None else sink.amax_slot(ln.b.data_ptr()), s_w,
                     ln.slot, s_y, _stream())
            ln.done = True
            self.ln_calls += 1
        else:
            lib.call('rscotr_lin_h3', x2.data_ptr(), M, N, K, wf, _ptr(bias), _ptr(resid), y.data_ptr(), _ptr(xscale), _ptr(yscale),
                     int(rows_per), s_x, s_w, s_y, _stream())
        self.calls += 1
        return y


LIN_FUSED = _FusedLinear()


class _MLP(Function):
    """y = L_n(act(L_{n-1}(... act(L_1(x))))) [+ identity], L_i(h) = h W_i^T + b_i: every Linear is
    one MFMA GEMM with bias/activation/residual fused in its epilogue; backward folds act' into the
    epilogue of the dX GEMM of the following layer (no separate element-wise passes).
    `out_scale` (B,) or None: per-sample factor on the last layer's output before the identity is added (the
    DropPath of a Swin block folded into its proj / fc2 Linear): forward rides the epilogue, backward the
    epilogue of dH and the operand staging of dW / db.
    `sum_with` (same shape as the output) or None: a second, non-differentiable output `y + sum_with` leaves the last
    epilogue (the `query + query_pos` of the attention that follows a positional MLP: ops.mha / ops.msda_attention take it as
    `q_sum`); only without identity / out_scale.
    args: x, identity (Tensor | None), act code, out_scale, sum_with, then W_1, b_1, ..., W_n, b_n (b may be None)."""

    @staticmethod
    def forward(ctx, x, identity, act, out_scale, sum_with, *wb):
        n = len(wb) // 2
        ws, bs = wb[0::2], wb[1::2]
        K0 = x.shape[-1]
        x2 = RANGES.carry(x, _f32c(x).reshape(-1, K0))
        M = x2.shape[0]
        id_is_x = identity is x  # mmcv FFN: identity defaults to the input itself
        rows_per = 0
        if out_scale is not None:
            out_scale = _f32c(out_scale)
            assert x.dim() == 3 and out_scale.numel() == x.shape[0]
            rows_per = x.shape[1]
        id2 = None if identity is None else (x2 if id_is_x else _f32c(identity).reshape(M, -1))
        s2 = y2 = None
        if sum_with is not None:
            assert identity is None and out_scale is None
            s2 = _f32c(sum_with).reshape(M, -1)
        hs, auxs = [x2], []
        h = x2
        # the last layer's range word: not for an output that takes a residual (a block output: the next reader is a norm) nor
        # where the caller said so (ops.linear(range_out=False): qkv of a window attention, ...)
        want_last = RANGE_OUT.want(not RANGE_OUT.skip_next and id2 is None)
        RANGE_OUT.skip_next = False
        ctx.fused = FFN_FUSED.ok(x2, ws, act, out_scale, sum_with)
        lz = getattr(x, '_lazy_ln', None)  # (the norm in front has not run yet: ops.layer_norm_fork(lazy=True))
        # ONE Linear, tall and narrow (ops.LIN_FUSED): the rows-resident launch instead of the tiled product
        ctx.lin = (n == 1 and act == ACT_NONE and sum_with is None and LIN_FUSED.ok(x2, ws[0], ws[0].shape[0], K0))
        if lz is not None and not lz.done and not ((ctx.fused and FFN_FUSED.ln_ok(lz, K0, act)) or (ctx.lin and LIN_FUSED.ln_ok(lz, K0))):
            lz.run()
        if lz is not None and lz.done:
            lz = None
        if ctx.lin:
            h = LIN_FUSED.run(x2, ws[0], bs[0], 0, id2, want_last, yscale=out_scale, rows_per=rows_per, ln=lz)
            n = 0
        if ctx.fused:
            W1 = ws[0] if ws[0].is_contiguous() else ws[0].contiguous()
            W2 = ws[1] if ws[1].is_contiguous() else ws[1].contiguous()
            if act == ACT_RELU:  # the gate as one bit per element
                aux = torch.empty(int(lib.rscotr_ffn_h3_bits_words(M, K0, W1.shape[0])), dtype=torch.int32, device=x2.device)
            else:  # GELU: the pre-activation
                aux = torch.empty((M, W1.shape[0]), dtype=torch.float32, device=x2.device)
            hid, h = FFN_FUSED.run(x2, W1, bs[0], W2, bs[1], act, aux, 0, id2, want_last, yscale=out_scale, rows_per=rows_per, ln=lz)
            hs.append(hid)
            auxs.append(aux)
            n = 0  # (the loop below has nothing left to do)
        for i in range(n):
            W = ws[i] if ws[i].is_contiguous() else ws[i].contiguous()
            N, K = W.shape
            last = i == n - 1
            pre, a_i = None, act
            if not last and act == ACT_GELU:
                pre = torch.empty((M, N), dtype=torch.float32, device=x2.device)
            elif not last and act == ACT_RELU and RELU_BITS.ok(M, N, K, ws[i + 1].shape[0]):
                # the gate leaves the forward epilogue as bits: the gated dH product of backward reads 1 / 32 of the bytes
                pre, a_i = torch.empty(M * N // 64, dtype=torch.int64, device=x2.device), ACT_RELU_BITS
            sc = out_scale if last else None
            if last and s2 is not None:  # out2 = y + sum_with, y itself stored without it (epilogue's second output)
                y2 = torch.empty((M, N), dtype=torch.float32, device=x2.device)
                h = gemm(h, W, M, N, K, K, K, 0, 0, bias=bs[i], act=ACT_NONE, resid=s2, out2=y2)
            else:
                h = gemm(h, W, M, N, K, K, K, 0, 0, bias=bs[i], act=ACT_NONE if last else a_i, pre=pre,
                         resid=id2 if last else None, rowscale=sc, rows_per=rows_per if sc is not None else 0,
                         range_out=want_last if last else True)
            if not last:
                hs.append(h)
                auxs.append(pre if pre is not None else h)  # (GELU: the pre-activation; ReLU: the gate bits, or h itself)
        n = len(ws)
        ctx.save_for_backward(*hs, *auxs, *ws)
        ctx.h_slots = [RANGES.saved(t) for t in hs]  # ((generation, slot) of the saved activations: the weight gradients want their ranges)
        ctx.out_scale, ctx.rows_per = out_scale, rows_per
        ctx.n, ctx.act, ctx.has_id, ctx.id_is_x = n, act, identity is not None, id_is_x
        ctx.has_bias = [b is not None for b in bs]
        ctx.biases = bs  # parameter handles only (for the gradient sink); not needed as saved tensors
        ctx.x_shape = x.shape
        ctx.id_shape = None if identity is None else identity.shape
        out = RANGES.carry(h, h.view(*x.shape[:-1], h.shape[-1]))
        if y2 is None:
            return out
        y2 = RANGES.carry(y2, y2.view(out.shape))
        ctx.mark_non_differentiable(y2)
        ctx.set_materialize_grads(False)
        ctx.two_outputs = True
        return out, y2

    @staticmethod
    def backward(ctx, dy, _dsum=None):
        if dy is None:  # (only the non-differentiable sum was used)
            return (None,) * (5 + 2 * ctx.n)
        n, act = ctx.n, ctx.act
        saved = ctx.saved_tensors
        hs, auxs, ws = saved[:n], saved[n:2 * n - 1], saved[2 * n - 1:]
        M = hs[0].shape[0]
        for t, sl in zip(hs, ctx.h_slots):
            RANGES.restore(t, sl)  # (only within the generation that wrote the word: a begin() since the forward voids it)
        g = RANGES.carry(dy, _f32c(dy).reshape(M, -1))
        g_out = g
        d_id = g.view(ctx.id_shape) if ctx.has_id and not ctx.id_is_x and ctx.needs_input_grad[1] else None
        grads_wb = [None] * (2 * n)
        gact = ACT_RELU_GRAD if act == ACT_RELU else ACT_GELU_GRAD
        dx = None
        def param_grads(i, g):
            """dW_i = g^T h_i and db_i (riding the contraction) into the arena / grads_wb; -> (W_i, scr of the layer)"""
            W = ws[i] if ws[i].is_contiguous() else ws[i].contiguous()
            N, K = W.shape
            want_w = ctx.needs_input_grad[5 + 2 * i]
            want_b = ctx.has_bias[i] and ctx.needs_input_grad[6 + 2 * i]
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
                    STATE.grad_sink.grad_written(sk[0])
                else:
                    gemm(g, hs[i], N, K, M, N, K, 1, 1, out=sk[1], accumulate=True, rowsum=rs,
                         rowsum_accumulate=rs_acc, **sck)
                    STATE.grad_sink.grad_written(sk[0])
            elif want_b:
                if sc is not None:
                    raise RuntimeError('out_scale with a bias-only gradient is not supported')
                colsum(g, M, N, out=rs, accumulate=rs_acc)
            if skb is not None:
                STATE.grad_sink.grad_written(skb[0])
            return W, scr

        if getattr(ctx, 'fused', False):
            # the mirrored pair dH = (g W2) * gate, dX = dH W1 (+ dy when the identity is the input) as ONE launch (ops.FFN_FUSED)
            W2, _ = param_grads(1, g)
            W1 = ws[0] if ws[0].is_contiguous() else ws[0].contiguous()
            # (a DropPath'ed block: the upstream gradient of both products is s_b * dy — the rows are scaled while they are staged)
            dH, dx = FFN_FUSED.run(g, W2, None, W1, None, act, auxs[0], 1, g_out if ctx.id_is_x else None, False,
                                   xscale=ctx.out_scale, rows_per=ctx.rows_per)
            param_grads(0, dH)
            dx = RANGES.carry(dx, dx.view(ctx.x_shape)) if ctx.needs_input_grad[0] else None
            return (dx, d_id, None, None, None, *grads_wb)
        for i in range(n - 1, -1, -1):
            W, scr = param_grads(i, g)
            N, K = W.shape
            if i > 0:
                ga = ACT_RELU_GRAD_BITS if (act == ACT_RELU and auxs[i - 1].dtype == torch.int64) else gact
                g = gemm(g, W, M, K, N, N, K, 0, 1, act=ga, aux=auxs[i - 1], **scr)  # dH = (g W) * act'
            elif ctx.needs_input_grad[0]:
                # identity == input: its gradient (dy) rides in this epilogue instead of a separate add
                # (the node's input gradient goes to a norm's backward, an attention backward or a merge: no later product
                #  multiplies with it directly)
                if n == 1 and getattr(ctx, 'lin', False) and LIN_FUSED.ok(g, W, K, N):
                    dx = LIN_FUSED.run(g, W, None, 1, g_out if ctx.id_is_x else None, RANGE_OUT.want(False),
                                       yscale=scr.get('rowscale'), rows_per=scr.get('rows_per', 0))
                else:
                    dx = gemm(g, W, M, K, N, N, K, 0, 1, resid=g_out if ctx.id_is_x else None, range_out=RANGE_OUT.want(False), **scr)
                dx = RANGES.carry(dx, dx.view(ctx.x_shape))
        return (dx, d_id, None, None, None, *grads_wb)


def mlp(x, layers, act='relu', identity=None, out_scale=None, sum_with=None, range_out=True):
    """layers: [(W, b), ...]; activation between layers, none after the last; `identity` (same shape
    as the output) is added in the last epilogue (mmcv FFN add_identity); `out_scale` (B,) multiplies the
    output per sample before that (DropPath).  With `sum_with` (shape of the output, values only): -> (y, y + sum_with), the
    sum without a gradient of its own."""
    flat = []
    for w, b in layers:
        flat += [w, b]
    RANGE_OUT.skip_next = not range_out  # (False: no later product multiplies with the output — RANGE_OUT)
    try:
        return _MLP.apply(x, identity, _ACT[act], out_scale, None if sum_with is None else sum_with.detach(), *flat)
    finally:
        RANGE_OUT.skip_next = False


def linear(x, w, b=None, act=None, resid=None, out_scale=None, range_out=True):
    """F.linear(x, w, b) [* out_scale per sample] (+ resid) on the matrix cores.  Activations belong to `mlp`.
    range_out=False: no later product multiplies with the output (RANGE_OUT)."""
    if act is not None:
        raise RuntimeError('ops.linear has no activation: use ops.mlp')
    RANGE_OUT.skip_next = not range_out
    try:
        return _MLP.apply(x, resid, ACT_NONE, out_scale, None, w, b)
    finally:
        RANGE_OUT.skip_next = False


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
    if STATE.profile is None:
        lib.call('rscotr_gemm_f32_batched', *args)
    else:
        with _Prof('gemm_batched', flops, 'rscotr::gemm_f32_kernel (batched attention products)'):
            lib.call('rscotr_gemm_f32_batched', *args)
    return C


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

