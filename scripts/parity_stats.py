"""Gradient-parity statistics of one train step per (task, size): product (GPU) and oracle-fp32 against the oracle
evaluated in fp64 under the same hard decisions.  Prints one JSON line per case; the thresholds of tests/parity.py are
set from these measurements.  usage: python scripts/parity_stats.py [--sizes 256,512] [--prec 0]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--sizes', default='256,512')
    ap.add_argument('--tasks', default='cls,det,seg')
    ap.add_argument('--prec', type=int, default=None)
    ap.add_argument('--seeds', default='17')
    a = ap.parse_args()
    from parity import anchor_report, grad_report, run_step_pair
    from util import build_model, load_model_cfg
    from rscotr_amd._lib import lib
    if a.prec is not None:
        lib.call('rscotr_gemm_set_precision', a.prec)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    dev = torch.device('cuda:0')
    cfg, mcfg = load_model_cfg(tiny=False)
    for size in [int(x) for x in a.sizes.split(',')]:
        for seed in [int(x) for x in a.seeds.split(',')]:
            model = build_model(mcfg, seed=4).to(dev)
            for task in a.tasks.split(','):
                t0 = time.time()
                out, oout, rec, orec, P = run_step_pair(model, mcfg, task, size, seed=seed, device=dev, fp64=True)
                rows = grad_report(model, P)
                loose = [r for r in rows if r[1] > 1.0 and r[3] > 1e-3]
                rep = anchor_report(model, P, orec['P64'], orec['P64b'])
                ratios = sorted(r['ep'] / max(r['eo'], 1e-30) for r in rep)
                ratios_b = sorted(r['ep'] / max(r['eo'], r['amb'], 1e-7) for r in rep)
                qb = lambda p: ratios_b[min(len(ratios_b) - 1, int(p * len(ratios_b)))]
                q = lambda p: ratios[min(len(ratios) - 1, int(p * len(ratios)))]
                print(json.dumps(dict(task=task, size=size, seed=seed, prec=lib.rscotr_gemm_get_precision(), tensors=len(rows),
                                      over_tight=len(loose), worst_tight=sorted(r[1] for r in rows)[-3:],
                                      worst_l2=sorted(r[3] for r in rows)[-3:],
                                      loss=[float(out['loss']), float(oout['loss']), float(orec['out64']['loss'])],
                                      ep_med=sorted(r['ep'] for r in rep)[len(rep) // 2], ep_max=max(r['ep'] for r in rep),
                                      eo_med=sorted(r['eo'] for r in rep)[len(rep) // 2], eo_max=max(r['eo'] for r in rep),
                                      ratio_q=[q(0.5), q(0.9), q(0.97), q(0.99), ratios[-1]],
                                      ratio_band_q=[qb(0.5), qb(0.9), qb(0.97), qb(0.99), ratios_b[-1]],
                                      amb_med=sorted(r['amb'] for r in rep)[len(rep) // 2],
                                      worst_band=[(r['name'], r['ep'], r['eo'], r['amb']) for r in
                                                  sorted(rep, key=lambda r: -r['ep'] / max(r['eo'], r['amb'], 1e-7))[:4]],
                                      worst_ratio=[(r['name'], r['ep'], r['eo']) for r in sorted(rep, key=lambda r: -r['ep'] / max(r['eo'], 1e-30))[:4]],
                                      seconds=round(time.time() - t0, 1))), flush=True)


if __name__ == '__main__':
    main()
