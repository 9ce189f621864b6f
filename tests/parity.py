"""Product-vs-oracle comparison of one MTL.train_step (forward losses, log keys, gradients,
Hungarian indices).  Used on CPU (HIP ops patched with the oracle: host-logic test) and on the
GPU (real HIP path: parity test)."""
import contextlib

import torch

from oracle import model as OM
from rscotr_amd import synth
from util import rel_err, state_to_oracle

# BASELINE.json north_star: outputs within 1e-3 relative in fp32; Hungarian indices bit-exact
RTOL = 1e-3


@contextlib.contextmanager
def default_dtype(dt):
    old = torch.get_default_dtype()
    torch.set_default_dtype(dt)
    try:
        yield
    finally:
        torch.set_default_dtype(old)


@contextlib.contextmanager
def ranges_checked():
    """Every value-range word a product takes from a tensor, the optimizer or a producer's epilogue is verified against what the
    tensor holds at that moment (ops.RANGES.check: one measuring launch + a synchronisation per operand).  A stale word is
    value-dependent — 8 x of headroom hides it until a seed exceeds it — so the whole-step tests at size run their eager
    iteration under the check, not only the 256^2 one."""
    from rscotr_amd import ops
    old = ops.RANGES.check
    ops.RANGES.check = ops.RANGES.enabled
    ops.RANGES.stats['checked'] = 0
    try:
        yield ops.RANGES
    finally:
        ops.RANGES.check = old


def cast_tree(obj, dt):
    """Floating tensors of a nested batch / rnd structure to dtype dt (everything else untouched)."""
    if torch.is_tensor(obj):
        return obj.to(dt) if obj.dtype.is_floating_point else obj
    if isinstance(obj, dict):
        return {k: cast_tree(v, dt) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(cast_tree(v, dt) for v in obj)
    return obj


def oracle_step_fp64(P, model_cfg, batch_cpu, rnd_cpu, orec32=None, relu_band=None, bilinear_band=None):
    """The oracle's train step evaluated in fp64 on the same weights, batch and draws, under the SAME hard decisions the
    fp32 evaluation took where they are injectable (seg attention masks and det top-k ride in rnd_cpu already; the 7*B
    assignments come from the fp32 record): the reference point that tells rounding error from wrong arithmetic.
    Returns (P64 with .grad, out)."""
    P64 = {k: (v.detach().double().requires_grad_(v.requires_grad) if v.dtype.is_floating_point else v.detach().clone())
           for k, v in P.items()}
    rnd64 = cast_tree(dict(rnd_cpu or {}), torch.float64)
    if orec32 is not None and orec32.get('match'):
        rnd64['det_match'] = orec32['match']
    from oracle import ops as O
    O.RELU_BAND, O.BILINEAR_BAND = relu_band, bilinear_band
    try:
        with default_dtype(torch.float64):
            out = OM.train_step(P64, model_cfg, cast_tree(batch_cpu, torch.float64), rnd64, {})
            out['loss'].backward()
    finally:
        O.RELU_BAND = O.BILINEAR_BAND = None
    return P64, out


def run_step_pair(model, model_cfg, task, size, seed, device='cpu', batch_size=2, P=None, inject_decisions=True,
                  fp64=False, opt=None, **batch_kw):
    """opt: a FlatAdamW over the model's parameters — the step then runs as it does under the runner: parameters and gradients
    in the optimizer's flat arenas, weight gradients written / accumulated straight into the gradient arena (ops.STATE.grad_sink),
    the deferred grouped contractions flushed after backward, and the routes that need a parameter's arena address (pre-split
    weight planes, the fused FFN / Swin MLP launches) ACTIVE.  Without it those routes are off and every gradient goes through
    autograd's AccumulateGrad."""
    batch_cpu = synth.make_batch(task, batch_size, size, seed=seed, **batch_kw)
    rnd_cpu = synth.make_rnd(model, batch_cpu, seed=seed)
    batch_dev = synth.make_batch(task, batch_size, size, seed=seed, device=device, **batch_kw)
    rnd_dev = synth.make_rnd(model, batch_cpu, seed=seed, device=device)
    if P is None:
        P = state_to_oracle(model)
    for p in P.values():
        if p.requires_grad:
            p.grad = None
    if opt is None:
        model.zero_grad(set_to_none=True)
    else:
        opt.zero_grad()
    rec, orec = {}, {}
    out = model.train_step(dict(batch_dev, rnd=rnd_dev, record=rec))
    out['loss'].backward()
    if opt is not None:
        from rscotr_amd import ops
        ops.flush_deferred()
    if task == 'seg' and inject_decisions:
        # hard decisions of the step (the `sigmoid(mask) < 0.5` attention masks) are compared
        # bit-wise in check_step_pair; the continuous part is compared under identical decisions
        rnd_cpu = dict(rnd_cpu or {}, seg_attn_masks=[m.cpu() for m in rec['attn_masks']])
    if task == 'det' and inject_decisions and 'topk_idx' in rec:
        # same for the top-600 proposal selection of the det step
        rnd_cpu = dict(rnd_cpu or {}, det_topk_idx=rec['topk_idx'].cpu())
    oout = OM.train_step(P, model_cfg, batch_cpu, rnd_cpu, orec)
    oout['loss'].backward()
    def fp64_anchor():
        orec['P64'], orec['out64'] = oracle_step_fp64(P, model_cfg, batch_cpu, rnd_cpu, orec)
        # the same step with every ReLU gate within RELU_BAND of zero flipped and every deformable-attention sample within
        # BILINEAR_BAND of a cell boundary taken from the neighbouring cell: how far coin-toss decisions can move a gradient
        orec['P64b'], _ = oracle_step_fp64(P, model_cfg, batch_cpu, rnd_cpu, orec, relu_band=RELU_BAND,
                                           bilinear_band=BILINEAR_BAND)

    if fp64:
        fp64_anchor()
    else:
        # (evaluated by check_step_pair only when the fp32 tiers fail: the fp32 ORACLE can itself sit on the other side of a
        # coin-toss ReLU gate — measured at Swin-B 1024^2 seg: product within 5e-6 of fp64 on every tensor, fp32 oracle
        # 7-14 % away on 559 of 615, the flipped-band evaluation moving by the same 7-14 %; scripts/seg_swinb_anchor.py)
        orec['_fp64_anchor'] = fp64_anchor
    return out, oout, rec, orec, P


def grad_report(model, P):
    """Per-parameter gradient comparison product vs oracle: (name, max_abs_err / tight_tol,
    fraction of elements over tight_tol, relative L2 error), tight_tol = RTOL * max|g_oracle| +
    1e-5 * max over all tensors (tensors whose exact gradient is 0 only hold rounding noise)."""
    gmax = max(float(p.grad.abs().max()) for p in P.values() if p.grad is not None)
    rows = []
    for n, p in model.named_parameters():
        go, g = P[n].grad, p.grad
        if go is None or float(go.abs().max()) == 0.0:
            assert g is None or float(g.abs().max()) <= 1e-5 * gmax, n
            continue
        assert g is not None, f'{n}: oracle has a gradient, product has none'
        e = (g.detach().cpu().double() - go.double()).abs()
        tol = RTOL * float(go.abs().max()) + 1e-5 * gmax
        rows.append((n, float(e.max()) / tol, float((e > tol).double().mean()),
                     float(e.norm() / (go.double().norm() + 1e-30))))
    return rows


RELU_BAND = 3e-6  # gates within 3e-6 of the mean |pre-activation| of zero count as coin tosses (fp32 products: ~1e-6)
BILINEAR_BAND = 2e-5  # sampling coordinates within 2e-5 pixels of a cell boundary (fp32 rounding of loc * W - 0.5 at W ~ 100: ~1e-5)


def anchor_report(model, P, P64, P64b=None):
    """Per parameter tensor: ep / eo = relative L2 distance of the product's / the fp32 oracle's gradient from the fp64
    evaluation of the same step (same decisions); amb = the distance of the band-flipped fp64 evaluation (P64b).  The
    denominator carries a floor (1e-5 of the largest gradient maximum, spread over the tensor) so that tensors whose
    exact gradient is zero or negligible compare as zero."""
    gmax = max(float(p.grad.abs().max()) for p in P64.values() if p.grad is not None)
    rows = []
    for n, p in model.named_parameters():
        g64 = P64[n].grad
        if g64 is None or p.grad is None or P[n].grad is None:
            continue
        den = float(g64.norm()) + 1e-5 * gmax * (g64.numel() ** 0.5)
        rows.append(dict(name=n, ep=float((p.grad.detach().cpu().double() - g64).norm()) / den,
                         eo=float((P[n].grad.double() - g64).norm()) / den,
                         amb=0.0 if P64b is None or P64b[n].grad is None else float((P64b[n].grad - g64).norm()) / den))
    return rows


# A train step contains hard decisions (ReLU gates, `sigmoid(mask) < 0.5` attention masks, top-k
# proposals, arg-max matching).  fp32 rounding can flip one of them between two correct
# implementations — the oracle itself moves single gradient rows by up to 2e-3 of the tensor's
# maximum between its fp32 and fp64 evaluation of the same step — and a flip is a finite change of
# the affected rows, not rounding noise.  The gradient gate is therefore two-tier: nearly every
# tensor must meet the 1e-3 tolerance of the north star element-wise; the few that contain a
# flipped decision must still agree in relative L2 norm.  Expected number of flips: the seg step at 256x256
# evaluates ~4e6 ReLU gates in its decoder alone; with ~1e-6 relative rounding noise between two fp32
# implementations a handful land on the other side of zero, each moving one row of one weight gradient (the
# failing tensors show exactly that signature: <= 0.05 % of their elements off) and, through the residual
# stream, nudging the tensors downstream of it to ~1.1e-3.
TIGHT_FRACTION = 0.97   # share of parameter tensors that must pass at RTOL (measured over 12 cases at 256^2 / 512^2, two
                        # seeds: 97.7-100 %; gpurun_out/r2t2_parity_stats.jsonl)
LOOSE_L2 = 3e-2         # relative L2 bound for the remaining tensors
LOOSE_MAX = 30.0        # and their worst element stays within 30x the tight tolerance

# The fp64 anchor.  With orec['P64'] / orec['P64b'] present (run_step_pair(fp64=True)) every gradient
# tensor of the product is also held against the SAME step evaluated in fp64 under the same injected decisions:
#   ep = |g_product - g_fp64| / |g_fp64|,  eo = the same for the fp32 oracle,  amb = the same for the fp64 evaluation with
#   every ReLU gate within RELU_BAND of zero flipped (what coin-toss gates can do to the tensor).
# Measured (MI355X, 12 cases): ep / max(eo, amb, 2e-7) has median 0.03-1.0, 97th percentile 1.1-2.4, 99th <= 3.7.  Until round 4
# the tail (max 5.7, one case 77, Swin-B 1024^2 det 114) sat on tensors fed by coin-toss decisions the ReLU band does not flip:
# bilinear cell boundaries of the deformable sampling (|frac| within fp32 rounding of 0 moves d/d(location) to the neighbouring
# cell's slope).  Round 5 flips those too (BILINEAR_BAND, oracle/ops.py): the tail is at 5.4 now.
ANCHOR_K = 4.0          # ep <= ANCHOR_K * max(eo, amb, ANCHOR_FLOOR) for at least ANCHOR_FRACTION of the tensors
ANCHOR_FRACTION = 0.97
ANCHOR_K_ALL = 12.0     # ... and within this factor for every tensor (round 5: with the bilinear cell-boundary band in the ambiguity
                        # evaluation the worst measured ratio is 5.4 — Swin-B 1024^2 det, bbox_head.reg_branches.5.4.bias — where the
                        # ReLU band alone left 114 and the bound stood at 150)
ANCHOR_FLOOR = 2e-7
ANCHOR_EP_MEDIAN = 1e-4  # absolute: the median tensor of the product is within 1e-4 of the fp64 evaluation (measured ~5e-6) ...
ANCHOR_K_MED = 1.5       # ... or, where the fp32 ORACLE's own median distance from fp64 or the median coin-toss ambiguity of the step is
                         # larger than that, within 1.5 x the larger of the two (Swin-B 1024^2 det: eo_med 2.4e-4, product 1.1e-4; MlvlClsHead
                         # scheme 7 at 224^2: every tensor within 1.22 x max(eo, amb), product median 2.8e-4 inside the coin-toss band) — the
                         # product is then no farther from fp64 than the reference implementation in fp32, or than its own ReLU coin tosses

# When the fp32 tiers FAIL and the anchor decides (the gate must not loosen exactly where it is consulted), the
# fp32 oracle's own distance eo is no yardstick any more — it is the suspect — so the product is held to the fp64 evaluation
# ABSOLUTELY: >= ANCHOR_FRACTION of the tensors within RTOL (1e-3, the north star's tolerance) of fp64, median within
# ANCHOR_EP_MEDIAN; the remaining tensors within ANCHOR_K_ALL x the coin-toss ambiguity amb of that tensor, or — only where
# the band does NOT explain the oracle's distance (eo > 10 amb: an ill-conditioned sum, e.g. a bias whose exact gradient
# nearly cancels, where both fp32 evaluations carry the same relative noise) — within ANCHOR_K x eo.
ANCHOR_EO_CLEAN = 10.0

# one record per checked step, printed by tests/conftest.py in the terminal summary (so the log of a run shows which tests
# needed the fp64 judge and how far the product was from it)
PARITY_LOG = []


def check_step_pair(model, out, oout, rec, orec, P, grad_rtol=None, loose_max=LOOSE_MAX, median_rel=False):
    """median_rel: the median tensor may be ANCHOR_K_MED x as far from the fp64 evaluation as the fp32 oracle's own median /
    the step's median coin-toss ambiguity where those exceed 1e-4 — only for the cases named in ANCHOR_K_MED's comment
    (everywhere else the absolute 1e-4 holds)."""
    if 'topk_idx' in rec and 'topk_idx' in orec:
        # proposal selection: the product's top-k vs the oracle's own.  Order and membership must agree
        # except where the scores involved are within fp32 rounding of each other.
        tp, to, sc = rec['topk_idx'].cpu(), orec['topk_idx'], orec['topk_scores']
        diff = 0
        for b in range(tp.shape[0]):
            mism = (tp[b] != to[b]).nonzero().flatten()
            diff += int(mism.numel())
            for q in mism.tolist():
                a, o = float(sc[b, tp[b, q]]), float(sc[b, to[b, q]])
                assert abs(a - o) <= 1e-5 * max(abs(o), 1.0), ('top-k differs beyond rounding', b, q, a, o)
        out['topk_positions_differing'] = (diff, tp.numel())
        assert diff <= 0.02 * tp.numel(), (diff, tp.numel())
    assert list(out['log_vars'].keys()) == list(oout['log_vars'].keys())
    assert out['num_samples'] == oout['num_samples']
    for k, v in out['log_vars'].items():
        ref = oout['log_vars'][k]
        assert abs(v - ref) <= RTOL * max(abs(ref), 1e-3), (k, v, ref)
    assert rel_err(out['loss'], oout['loss']) <= RTOL
    rows = grad_report(model, P)
    # tight tier: element-wise within RTOL of the tensor's maximum, or within RTOL in relative L2 norm (a
    # single flipped ReLU gate moves one row of a weight gradient by a finite amount: large as an element,
    # invisible in the norm)
    loose = [r for r in rows if r[1] > 1.0 and r[3] > RTOL]
    out['grad_report'] = dict(tensors=len(rows), over_tight=len(loose),
                              worst=sorted(loose, key=lambda r: -r[1])[:8])
    bad = [r for r in loose if r[3] > LOOSE_L2 or (loose_max is not None and r[1] > loose_max)]
    fp32_tiers_ok = len(rows) - len(loose) >= TIGHT_FRACTION * len(rows) and not bad
    on_gpu = next(model.parameters()).is_cuda  # (the CPU runs patch the HIP ops with the oracle: host-logic tests, old gate)
    if loose and on_gpu and 'P64' not in orec and '_fp64_anchor' in orec:
        #  no tensor enters the loose tier unexplained: the fp64 anchor is evaluated whenever ANY tensor
        # misses the 1e-3 tier, and each such tensor must then be accounted for below (EXPLAINED)
        orec['_fp64_anchor']()
    if not fp32_tiers_ok:
        # product and fp32 oracle disagree beyond the two tiers: decided by the fp64 anchor below (which of the two fp32
        # evaluations is away from the fp64 one under the same decisions, and can a coin-toss ReLU gate explain it?)
        if 'P64' not in orec and '_fp64_anchor' in orec:
            orec['_fp64_anchor']()
        assert 'P64' in orec, (out['grad_report'], bad[:5])
        out['grad_report']['decided_by_fp64_anchor'] = True
    decided = bool(out['grad_report'].get('decided_by_fp64_anchor'))
    if 'P64' in orec:
        rep = anchor_report(model, P, orec['P64'], orec.get('P64b'))
        ep_med = sorted(r['ep'] for r in rep)[len(rep) // 2]
        eo_med = sorted(r['eo'] for r in rep)[len(rep) // 2]
        amb_med = sorted(r['amb'] for r in rep)[len(rep) // 2]
        if decided:
            def allowance(r):
                a = max(RTOL, ANCHOR_K_ALL * max(r['amb'], ANCHOR_FLOOR))
                return max(a, ANCHOR_K * r['eo']) if r['eo'] > ANCHOR_EO_CLEAN * r['amb'] else a
            ratio = sorted(((r['ep'] / allowance(r), r['name']) for r in rep), reverse=True)
            within = sum(1 for r in rep if r['ep'] <= RTOL)
        else:
            ratio = sorted(((r['ep'] / max(r['eo'], r['amb'], ANCHOR_FLOOR), r['name']) for r in rep), reverse=True)
            within = sum(1 for x, _ in ratio if x <= ANCHOR_K)
        out['anchor_report'] = dict(tensors=len(rep), within_k=within, worst=ratio[:5], ep_med=ep_med, eo_med=eo_med,
                                    amb_med=amb_med, decided=decided)
        # every tensor outside the 1e-3 tier of the fp32 oracle needs a REASON: either the product is within 1e-3 of the fp64
        # evaluation (then the fp32 oracle is the one that moved), or the coin-toss ambiguity of that very tensor (the fp64
        # evaluation with the borderline ReLU gates flipped) covers the product's distance, or — an ill-conditioned sum whose
        # two fp32 evaluations carry the same noise (eo > 10 amb) — the product is as close to fp64 as the fp32 oracle is
        by_name = {r['name']: r for r in rep}
        unexplained = []
        for r in loose:
            a = by_name.get(r[0])
            if a is None:
                continue
            ok = (a['ep'] <= RTOL or a['ep'] <= ANCHOR_K_ALL * max(a['amb'], ANCHOR_FLOOR)
                  or (a['eo'] > ANCHOR_EO_CLEAN * a['amb'] and a['ep'] <= ANCHOR_K * a['eo']))
            if not ok:
                unexplained.append((r[0], dict(ep=a['ep'], eo=a['eo'], amb=a['amb'], elementwise=r[1], l2=r[3])))
        out['anchor_report']['loose_explained'] = (len(loose) - len(unexplained), len(loose))
        assert not unexplained, ('tensors outside the 1e-3 tier that neither the fp64 evaluation nor the coin-toss band explains',
                                 unexplained[:5])
        assert within >= ANCHOR_FRACTION * len(rep), out['anchor_report']
        assert ratio[0][0] <= (1.0 if decided else ANCHOR_K_ALL), out['anchor_report']
        assert ep_med <= (max(ANCHOR_EP_MEDIAN, ANCHOR_K_MED * max(eo_med, amb_med)) if median_rel else ANCHOR_EP_MEDIAN), \
            out['anchor_report']
    import os
    PARITY_LOG.append(dict(test=os.environ.get('PYTEST_CURRENT_TEST', '?').split(' ')[0], tensors=len(rows),
                           over_tight=len(loose), decided_by_fp64_anchor=decided,
                           anchor=None if 'anchor_report' not in out else dict(
                               within=out['anchor_report']['within_k'], of=out['anchor_report']['tensors'],
                               loose_explained=out['anchor_report'].get('loose_explained'),
                               worst_ratio=round(out['anchor_report']['worst'][0][0], 3),
                               worst_tensor=out['anchor_report']['worst'][0][1],
                               ep_med=float(f"{out['anchor_report']['ep_med']:.3g}"),
                               eo_med=float(f"{out['anchor_report']['eo_med']:.3g}"),
                               amb_med=float(f"{out['anchor_report']['amb_med']:.3g}"))))
    if 'attn_masks' in rec and 'attn_masks' in orec:
        # masked-attention decisions: the oracle's own masks vs the product's, bit for bit; a logit
        # within fp32 rounding of 0 may land on either side, nothing else may differ
        diff = total = 0
        for mp, mo in zip(rec['attn_masks'], orec['attn_masks']):
            diff += int((mp.cpu() != mo).sum())
            total += mo.numel()
        out['mask_bits_differing'] = (diff, total)
        assert diff <= 2e-5 * total, (diff, total)
    if 'match' in rec:  # bit-exact assignment indices for all 7*B matchings
        n = 0
        for (s, i), (r, c) in rec['match'].items():
            o = orec['match']['interm' if s == 0 else f'dec{s - 1}'][i]
            assert torch.equal(torch.from_numpy(r), o['pos_inds']), (s, i)
            assert torch.equal(torch.from_numpy(c), o['pos_assigned_gt_inds']), (s, i)
            n += 1
        assert n == sum(len(v) for v in orec['match'].values())
