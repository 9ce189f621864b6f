"""Process-wide switches and hooks of the operator layer, in ONE object (`STATE`) instead of module globals.

Every op reads its strategy / hook from here at call time, so a caller that wants another setting changes it through
`STATE.override(...)` (restored on exit) or assigns the field — and `STATE.assert_defaults()` tells whether anything was
left changed (tests/conftest.py checks it before every test: the parity suite must run what ships)."""
import contextlib
import os


class OpsState:
    # field -> (environment variable, parser, default): the A/B switches a user may set before the process starts
    ENV = dict(
        msda_bwd=('RSCOTR_MSDA_BWD', str, 'tiled'),          # 'tiled' | 'sorted' | 'scatter' (ops/deform.py)
        fan_out=('RSCOTR_FAN_OUT', lambda s: s != '0', '1'),  # ops.fan_out sums consumer gradients 8 at a time
        msda_packed=('RSCOTR_MSDA_PACKED', lambda s: s != '0', '1'),  # offsets | weights projections as one product
        pos_sum=('RSCOTR_POS_SUM', lambda s: s != '0', '1'),  # `query + query_pos` leaves the preceding LayerNorm's launch
        merge_norm=('RSCOTR_MERGE_NORM', lambda s: s != '0', '1'),  # PatchMerging's unfold done by its LayerNorm's loads / stores
        msda_fused=('RSCOTR_MSDA_FUSED', lambda s: s != '0', '1'),  # softmax / location prologue (and its backward) inside the MSDA sample-order kernels
        attn_core=('RSCOTR_ATTN_CORE', lambda s: s != '0', '1'),  # dense attention (head dim 32) as one fused pass per direction, no stored scores
    )

    def __init__(self):
        for name, (env, parse, default) in self.ENV.items():
            setattr(self, name, parse(os.environ.get(env, default)))
        # hooks (None = inactive)
        self.grad_sink = None      # rscotr_amd.optim.FlatAdamW: backward kernels add parameter gradients into its arena
        self.side = None           # active side stream of the weight-gradient contractions (ops.side_enable)
        self.profile = None        # list: every profiled launch appends dict(kind, bytes, e0, e1) (bench.py)
        self.profile_every = {'gemm': 8}
        self._defaults = {name: getattr(self, name) for name in self.ENV}

    def defaults(self):
        """The switch values this process started with (environment or built-in)."""
        return dict(self._defaults)

    def changed(self):
        """{field: (current, default)} of every switch that no longer has its start value."""
        return {k: (getattr(self, k), v) for k, v in self._defaults.items() if getattr(self, k) != v}

    def assert_defaults(self):
        ch = self.changed()
        if ch:
            raise AssertionError(f'operator switches differ from the product defaults: {ch}')

    @contextlib.contextmanager
    def override(self, **kw):
        old = {k: getattr(self, k) for k in kw}
        for k in kw:
            if k not in self.ENV and k not in ('profile', 'side', 'grad_sink'):
                raise AttributeError(k)
        try:
            for k, v in kw.items():
                setattr(self, k, v)
            yield self
        finally:
            for k, v in old.items():
                setattr(self, k, v)


STATE = OpsState()
