import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')
    config.addinivalue_line('markers', 'timeout: per-test limit (pytest-timeout)')
    import torch
    # the oracle's CPU ops: a bounded thread pool (the build container has 8 cores and the gloo test spawns 2 more
    # processes; the GPU box has hundreds of logical CPUs and the oracle is what the `-m gpu` suite waits for)
    torch.set_num_threads(min(32 if torch.cuda.is_available() else 4, os.cpu_count() or 1))


@pytest.fixture(scope='session')
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


@pytest.fixture
def gemm_precision():
    """Restores the process-wide precision mode of the tiled GEMM (include/rscotr.h: rscotr_gemm_set_precision)."""
    from rscotr_amd._lib import lib
    old = lib.rscotr_gemm_get_precision()
    yield lambda m: lib.call('rscotr_gemm_set_precision', m)
    lib.call('rscotr_gemm_set_precision', old)


@pytest.fixture
def six_term():
    """The six-term bf16 product of rounds 2-4 for the duration of a test (what RSCOTR_GEMM_H3=0 selects): value ranges off,
    pre-split weight planes on.  Restored on exit."""
    from rscotr_amd import ops
    old = (ops.RANGES.enabled, ops.WPLANES.enabled)
    ops.RANGES.enabled, ops.WPLANES.enabled = False, os.environ.get('RSCOTR_WPLANES', '1') != '0'
    yield
    ops.RANGES.enabled, ops.WPLANES.enabled = old


# ----------------------------------------------------------------------------------------------------------------------
# The parity suite must test what ships (a test that left the MSDA backward strategy changed made
# every later whole-step test run a non-default kernel).  Before EVERY test — ahead of the test's own fixtures, which may
# then select another setting for the test's duration — the process-wide product state must equal what the process started
# with: the operator switches (ops.STATE), the GEMM precision mode, the deferred-work / weight-plane toggles, the RSCOTR_*
# environment.  A leak fails the NEXT test with the name of what leaked instead of silently changing what it measures.
# ----------------------------------------------------------------------------------------------------------------------
_BASE = {}


def _product_state():
    from rscotr_amd import ops
    from rscotr_amd._lib import LIB_PATH, lib
    st = dict(switches=tuple(sorted(ops.STATE.changed().items())),
              hooks=(ops.STATE.side is None, ops.STATE.profile is None),
              defer=(ops.DEFER.enabled, ops.DEFER.group_enabled, ops.DEFER.group_x6, ops.DEFER.pin),
              wplanes=(ops.WPLANES.enabled, ops.HPLANES.enabled, ops.RELU_BITS.enabled, ops.FFN_FUSED.enabled, ops.LIN_FUSED.enabled),
              ranges=(ops.RANGES.enabled, ops.RANGES.check, ops.RANGE_OUT.all, ops.RANGE_OUT.skip_next),
              env=tuple(sorted((k, v) for k, v in os.environ.items() if k.startswith('RSCOTR_'))))
    if os.path.exists(LIB_PATH):
        st['gemm_precision'] = int(lib.rscotr_gemm_get_precision())
    return st


def pytest_runtest_setup(item):
    cur = _product_state()
    if not _BASE:
        assert cur['switches'] == (), cur['switches']
        _BASE.update(cur)
        return
    leaked = {k: (cur[k], _BASE[k]) for k in _BASE if cur.get(k) != _BASE[k]}
    assert not leaked, f'product state changed by an earlier test and not restored (now, at start): {leaked}'


def pytest_sessionfinish(session, exitstatus):
    if _BASE:
        # one line in the log of every run: what the whole-step tests ran with
        from rscotr_amd import ops
        print(f"\n[conftest] product state held for the whole session: switches {ops.STATE.defaults()}, "
              f"gemm precision mode {_BASE.get('gemm_precision')}")


def pytest_terminal_summary(terminalreporter):
    """Which whole-step parity checks needed the fp64 judge, and the product's distance from it (tests/parity.py)."""
    parity = sys.modules.get('parity')
    if parity is None or not parity.PARITY_LOG:
        return
    tr = terminalreporter
    tr.write_sep('-', 'whole-step parity: fp32 tiers / fp64 anchor')
    for r in parity.PARITY_LOG:
        a = r['anchor']
        tr.write_line(f"{r['test']}: {r['tensors'] - r['over_tight']}/{r['tensors']} tensors within 1e-3 of the fp32 oracle; "
                      f"decided_by_fp64_anchor={r['decided_by_fp64_anchor']}"
                      + ('' if a is None else f"; anchor {a['within']}/{a['of']} within bound, worst ratio {a['worst_ratio']} "
                                               f"({a['worst_tensor']}), median ep {a['ep_med']} / eo {a['eo_med']} / amb {a.get('amb_med')}"
                                               + ('' if not a.get('loose_explained') else
                                                  f"; tensors outside the 1e-3 tier explained by the fp64 evaluation / coin-toss "
                                                  f"band: {a['loose_explained'][0]}/{a['loose_explained'][1]}")))
