"""Benchmark of the RSCoTr multi-task co-training step on MI355X.

`python bench.py --gpus N --steps K --warmup W`: N > 1 ranks are either started by a launcher (the driver:
torch.distributed.run, one rank per GPU over RCCL; WORLD_SIZE must equal --gpus) or, when no launcher is in the
environment, by this script itself (spawn_ranks).  A bench "step" is ONE ROUND of the reference's round-robin alternation
(cls batch, det batch, seg batch — SURVEY.md §8d), i.e. 3 train iterations = 3*B images per GPU,
each with forward, loss, backward, gradient exchange, global-norm clip and AdamW.  Prints one JSON
line on rank 0 (contract in the task statement): images/sec whole-job, the roofline of the MSDA
forward kernel measured with HIP events inside the timed region, and the CPU baseline (the
oracle on the host cores, one round of the same workload).
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CFG = os.path.join(ROOT, 'configs', 'multi', 'MTL_slvlcls_swin-t-p4-w7_1x1_resisc&dior&potsdam.py')
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3  # MI355X_MICROARCH.md: fp32-input MFMA (= fp32 vector) peak, dense
MFMA_BF16_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: bf16 MFMA peak, dense (no sparsity)
# the fp16 split product (the default large-product route, RSCOTR_GEMM_H3=0 turns it off: gemm_h3_*_kernel) issues three fp16 MFMAs per fp32-equivalent product
# (price a split product by its MFMA issues; fp16 and bf16 MFMAs share the 2.5 PFLOP/s dense peak)
H3_PEAK_TF = MFMA_BF16_PEAK_TF / 3.0
# gemm_bf16x6_kernel (the route with RSCOTR_GEMM_H3=0 and for operands without a range word; fp32-accurate): six bf16 MFMAs per
# fp32-equivalent product
BF16X6_PEAK_TF = MFMA_BF16_PEAK_TF / 6.0
PROF_EVERY_GEMM = 1  # every GEMM launch of the roofline rounds carries a pair of HIP events (1 in 4 made the choice of the dominant instantiation depend on which launches were drawn)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', choices=sorted(WORKLOADS), default='mtl512',
                    help='mtl512 = BASELINE configs[1] (the metric; default); det800 = configs[3] (DINO det step only, 800x800 '
                         'bs=4); swinb1024 = configs[4] (Swin-B backbone, cls+det+seg round at 1024x1024 bs=1)')
    ap.add_argument('--size', type=int, default=None, help='override the workload\'s image size')
    ap.add_argument('--batch', type=int, default=None, help='override the workload\'s per-task batch size')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true', help='skip the eager profiled rounds after the timed region')
    ap.add_argument('--roofline-rounds', type=int, default=2)
    ap.add_argument('--roofline-hold-ms', type=float, default=60.0,
                    help='stream hold ahead of each eager roofline iteration so that its launches queue up')
    ap.add_argument('--cpu-rounds', type=int, default=3, help='timed CPU-baseline rounds (after one untimed warm-up round)')
    ap.add_argument('--cpu-baseline-worker', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--verbose', action='store_true', help='per-iteration wall times on stderr')
    ap.add_argument('--host-trace', action='store_true', help='per-iteration HOST times (no sync) on stderr')
    ap.add_argument('--watchdog', type=int, default=0,
                    help='dump all Python stacks and exit if the run takes longer than this many seconds')
    ap.add_argument('--launcher', choices=('auto', 'spawn', 'none'), default='auto',
                    help='auto: with --gpus N > 1 and no WORLD_SIZE in the environment, start the N ranks here (torch.distributed.run, '
                         'one process per GPU over RCCL) — tools/train.py:173-182 / mtl/apis/train.py:37-46; spawn: do that for N = 1 '
                         'too; none: never (this process is the only rank)')
    ap.add_argument('--exchange', choices=('inline', 'overlap'), default=None,
                    help='form of the gradient exchange inside the captured iteration (rscotr_amd/dist.py): inline = collectives on '
                         'the compute stream after backward; overlap = bucket all-reduces on RCCL\'s stream under backward '
                         '(torch DDP\'s form). Default: RSCOTR_DIST_INLINE, else inline')
    a = ap.parse_args()
    w = WORKLOADS[a.workload]
    a.size = a.size or w['size']
    a.batch = a.batch or w['batch']
    return a


# BASELINE.json configs -> what one bench "step" runs.  tasks: the iterations of one step, in order; max_gt: ground truths
# per image of the synthetic det batches (C2 U{1..20}, C4 U{1..50}); swin_b: configs[4]'s backbone.
WORKLOADS = {
    'mtl512': dict(size=512, batch=2, tasks=('cls', 'det', 'seg'), max_gt=20, swin_b=False,
                   name='configs/multi MTL_slvlcls swin-t-p4-w7 RESISC45+DIOR+Potsdam',
                   metric='images/sec MTL train step (Swin-T 512^2, bs=2/GPU)'),
    'det800': dict(size=800, batch=4, tasks=('det',), max_gt=50, swin_b=False,
                   name='configs/det DIOR Deformable-DETR (DINO) head only on the MTL trunk (Swin-T + ChannelMapper + shared '
                        'encoder)',
                   metric='images/sec det train step (Swin-T 800^2, bs=4/GPU)'),
    'swinb1024': dict(size=1024, batch=1, tasks=('cls', 'det', 'seg'), max_gt=20, swin_b=True,
                      name='Swin-B backbone MTL RESISC45+DIOR+Potsdam',
                      metric='images/sec MTL train step (Swin-B 1024^2, bs=1/GPU)'),
}


def workload_model_cfg(cfg, workload):
    """The model config of a workload: configs[4] swaps the backbone for Swin-B (embed 128, depths 2-2-18-2, heads
    4-8-16-32; neck / cls head inputs follow) — SURVEY.md 8d C5."""
    import copy
    m = copy.deepcopy(cfg.model)
    if WORKLOADS[workload]['swin_b']:
        m['backbone'].update(embed_dims=128, depths=(2, 2, 18, 2), num_heads=(4, 8, 16, 32))
        m['neck']['in_channels'] = [256, 512, 1024]
        m['cls_head']['in_channels'] = 1024
    return m


CPU_THREADS_CAP = 32  # host threads for the oracle (more only adds scheduling overhead on these op sizes)


def cpu_baseline_worker(size, batch, rounds, workload='mtl512'):
    """Oracle (plain PyTorch fp32 on the host cores) on the same workload: forward, loss, backward,
    clip, AdamW for `rounds` rounds of cls+det+seg.  Runs in its own process (see cpu_baseline)."""
    import copy
    from oracle import model as OM
    from oracle.optim import OracleOptimizer
    from rscotr_amd import synth, Config, MODELS
    threads = min(os.cpu_count() or 1, CPU_THREADS_CAP)
    torch.set_num_threads(threads)
    cfg = Config.fromfile(CFG)
    wl = WORKLOADS[workload]
    mcfg = workload_model_cfg(cfg, workload)
    torch.manual_seed(0)
    model = MODELS.build(copy.deepcopy(mcfg))
    model.init_weights()
    P = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in model.state_dict().items()}
    opt = OracleOptimizer({k: v for k, v in P.items() if v.requires_grad}, cfg.optimizer, max_norm=0.1)

    def one_round(r):
        for task in wl['tasks']:
            b = synth.make_batch(task, batch, size, seed=9000 + r, max_gt=wl['max_gt'])
            rnd = synth.make_rnd(model, b, seed=r)
            out = OM.train_step(P, mcfg, b, rnd)
            opt.zero_grad()
            out['loss'].backward()
            opt.step()

    one_round(-1)  # warm-up round (thread pools, allocator, oneDNN primitive caches): not timed
    t0 = time.time()
    for r in range(rounds):
        one_round(r)
    dt = time.time() - t0
    n_img = rounds * batch * len(wl['tasks'])
    print(json.dumps(dict(value=n_img / dt, unit='images/s', cores=threads, kind='port',
                          sample=f'1 warm-up + {rounds} timed round(s) of {"+".join(wl["tasks"])} at {size}x{size}, '
                                 f'B={batch}/task ({n_img} images timed), fwd+loss+bwd+clip+AdamW, oracle (plain PyTorch fp32) '
                                 f'on {threads} host threads of {os.cpu_count()} logical CPUs, {dt:.1f}s')))


def cpu_baseline(size, batch, rounds, workload='mtl512', limit_s=420):
    """Bounded CPU baseline: the oracle timed in a child process with a hard time limit, so a slow
    host can never hang the benchmark.  kind = "port": the reference itself cannot be imported."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', '--size', str(size), '--batch', str(batch),
           '--cpu-rounds', str(rounds), '--workload', workload]
    env = dict(os.environ, HIP_VISIBLE_DEVICES='', CUDA_VISIBLE_DEVICES='')
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=limit_s, env=env)
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith('{'):
                return json.loads(line)
        return dict(value=None, unit='images/s', cores=None, kind='port', sample=f'worker failed: {r.stderr[-300:]}')
    except subprocess.TimeoutExpired:
        return dict(value=None, unit='images/s', cores=min(os.cpu_count() or 1, CPU_THREADS_CAP), kind='port',
                    sample=f'one round did not finish within {limit_s}s on the host cores')


def spawn_ranks(a):
    """`python bench.py --gpus N` started by hand (no launcher in the environment): start the N ranks — what the reference's
    `init_dist(args.launcher, ...)` expects a launcher to have done (tools/train.py:173-182) — by re-running this command under
    `torch.distributed.run` on 127.0.0.1, one process per GPU.  Rank 0 of the children prints the ONE JSON line on the stdout
    they inherit.  -> exit code of the launcher."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()
    if n_dev < a.gpus:
        raise SystemExit(f'bench.py --gpus {a.gpus}: this node shows {n_dev} GPU(s)')
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={a.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // max(a.gpus, 1))))
    print(f'[bench] starting {a.gpus} rank(s): {" ".join(cmd)}', file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    a = parse()
    if a.cpu_baseline_worker:
        return cpu_baseline_worker(a.size, a.batch, a.cpu_rounds, a.workload)
    if a.exchange is not None:  # read by rscotr_amd.dist at import (below)
        os.environ['RSCOTR_DIST_INLINE'] = '1' if a.exchange == 'inline' else '0'
    launched = 'WORLD_SIZE' in os.environ
    if not launched and a.launcher != 'none' and (a.gpus > 1 or a.launcher == 'spawn'):
        sys.exit(spawn_ranks(a))
    if int(os.environ.get('WORLD_SIZE', '1')) != a.gpus:
        raise SystemExit(f'bench.py: --gpus {a.gpus} but WORLD_SIZE={os.environ.get("WORLD_SIZE", "unset (1 rank)")}: the bench line '
                         'would not be an N-GPU measurement — launch N ranks (or drop --launcher none)')
    if a.watchdog > 0:
        import faulthandler
        faulthandler.dump_traceback_later(a.watchdog, exit=True)
    # stdout carries ONE JSON line and nothing else: libraries write to file descriptor 1 behind Python's back (RCCL prints
    # its version banner there at exit, after the JSON line, whatever NCCL_DEBUG says), so descriptor 1 is pointed at stderr for
    # the life of the process and the JSON line goes to a private duplicate of the real stdout
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise RuntimeError('bench.py needs an MI355X: the product path has no CPU fallback')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if world > 1 or os.environ.get('RSCOTR_DIST_SINGLE') == '1':  # the latter: distributed code path, one rank
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        os.environ.setdefault('NCCL_DEBUG', 'WARN')  # no version banner on stdout next to the JSON line
        if os.environ.get('RSCOTR_DIST_INLINE', '1') == '0':
            # the overlapped exchange is captured only once the RCCL watchdog is known to be idle, which c10d's flight recorder
            # tells (rscotr_amd.runner._wait_watchdog_idle): it must be on when the process group is created
            os.environ.setdefault('TORCH_NCCL_TRACE_BUFFER_SIZE', '2000')
        dist.init_process_group('nccl', device_id=dev)

    import copy
    from rscotr_amd import Config, MODELS, ops
    from rscotr_amd.data import build_synthetic_multidataloader
    from rscotr_amd.runner import build_runner

    cfg = Config.fromfile(CFG)
    wl = WORKLOADS[a.workload]
    ntask = len(wl['tasks'])
    torch.manual_seed(0)  # identical init on all ranks
    np.random.seed(2022)  # identical task order / augment choice on all ranks (tools/train.py:211-215)
    model = MODELS.build(workload_model_cfg(cfg, a.workload))
    model.init_weights()
    model.to(dev).train()
    loader = build_synthetic_multidataloader(cfg, dev, size=a.size, batch_size=a.batch, rank=rank, tasks=wl['tasks'],
                                             max_gt=wl['max_gt'])
    runner = build_runner(model, cfg, loader, logger=lambda m: print(m, file=sys.stderr, flush=True))  # (stdout carries ONE JSON line)

    hold = dict(cycles=0)  # > 0: park the stream this many spin cycles before every iteration (roofline rounds)
    marks = []  # timed region only: (task, HIP event recorded after the iteration) -> per-task step times
    timing = dict(on=False)

    def one_round():
        for _ in range(ntask):
            if hold['cycles']:
                # the eager host path issues ~3000 launches per iteration slower than the GPU drains them; parking
                # the stream first lets the whole iteration queue up, so the kernels (and the events around them)
                # then execute back to back as they do in the replayed graphs instead of each starting on an idle,
                # down-clocked GPU
                with torch.cuda.stream(runner.stream) if runner.stream is not None else contextlib.nullcontext():
                    torch.cuda._sleep(hold['cycles'])
            if a.verbose:
                torch.cuda.synchronize()
            t_it = time.perf_counter()
            runner.train_iter()
            if timing['on']:  # an event record costs the host ~1 us and the stream nothing
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.append((runner.last_task, ev))
            if a.verbose:
                torch.cuda.synchronize()
            if a.verbose or a.host_trace:
                print(f'[bench] iter {runner.iter} {(time.perf_counter() - t_it) * 1e3:.1f} ms'
                      f'{"" if a.verbose else " (host only)"}', file=sys.stderr, flush=True)

    # the whole loop runs with the runner's stream current (runner.on_stream(): no per-iteration stream hand-over; the
    # events below are recorded on that stream)
    _loop_stream = contextlib.ExitStack()
    _loop_stream.enter_context(runner.on_stream())
    # set-up outside the W warm-up steps: first round eager (parameter liveness, workspaces), second
    # round captures the shape-static tasks into hipGraphs (rscotr_amd.runner.GraphedTask)
    for _ in range(2):
        one_round()
    for _ in range(a.warmup):
        one_round()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    from rscotr_amd._lib import lib
    eager_timed = not runner.graphed  # no graphs (RSCOTR_GRAPHS=0): the launch-site events ride in the timed region
    if eager_timed and rank == 0 and not a.no_roofline:
        lib.call('rscotr_prof_enable', PROF_EVERY_GEMM, 1, 1, 8192)
    torch.cuda.synchronize()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev0.record()
    timing['on'] = True
    t0 = time.perf_counter()
    for _ in range(a.steps):
        one_round()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    timing['on'] = False
    per_task, prev = {}, ev0
    for task, ev in marks:  # device time between the ends of consecutive iterations
        per_task.setdefault(task, []).append(prev.elapsed_time(ev))
        prev = ev
    per_task = {t: round(sum(v) / len(v), 3) for t, v in per_task.items()}
    if runner.graphed and not a.no_roofline:  # every rank runs these rounds (they hold collectives); rank 0 records
        # The timed region replays hipGraphs, which cannot carry per-kernel HIP events.  The rooflines are
        # therefore sampled on the same process, model and stream directly after it: the same iterations
        # launched eagerly (identical kernels, arguments and shapes) with the library recording a pair of
        # HIP events on the launch stream around every n-th launch (rscotr_prof_*: events and launch are issued
        # back to back inside the C entry, so host time between them does not leak into the duration).
        # Not part of `value`.  rocprofv3 --kernel-trace of this command sees both phases.
        # calibrate the spin kernel, then hold the stream ~60 ms ahead of every eager iteration
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); torch.cuda._sleep(20_000_000); e1.record(); torch.cuda.synchronize()
        hold['cycles'] = int(20_000_000 * a.roofline_hold_ms / max(e0.elapsed_time(e1), 1e-3))
        runner.force_eager = True
        one_round()  # un-profiled: allocator / workspaces of the eager path
        torch.cuda.synchronize()
        if rank == 0:
            lib.call('rscotr_prof_enable', PROF_EVERY_GEMM, 1, 1, 8192)
        for _ in range(a.roofline_rounds):
            one_round()
        if rank == 0:
            # what the bracketing itself costs: empty event pairs queued behind the same hold (subtracted from every sample below)
            with torch.cuda.stream(runner.stream) if runner.stream is not None else contextlib.nullcontext():
                torch.cuda._sleep(hold['cycles'])
                lib.call('rscotr_prof_empty', 64, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        runner.force_eager = False
        hold['cycles'] = 0
    _loop_stream.close()
    prof = []
    if rank == 0 and not a.no_roofline and (eager_timed or runner.graphed):
        import ctypes
        torch.cuda.synchronize()
        n = lib.rscotr_prof_pause()
        kind, work, ms = ctypes.c_int(), ctypes.c_double(), ctypes.c_float()
        name = ctypes.create_string_buffer(128)
        for i in range(n):
            lib.call('rscotr_prof_get', i, ctypes.byref(kind), ctypes.byref(work), ctypes.byref(ms), name, 128)
            prof.append(dict(kind=('gemm', 'msda_fwd', 'msda_bwd', 'hbm', 'mfma')[kind.value], work=work.value, sec=ms.value * 1e-3,
                             name=name.value.decode()))
        lib.call('rscotr_prof_disable')
        # HIP events around a launch add the processing of the second record's packet to the kernel's own duration (VERDICT r5 item 7:
        # 26.1 us bracketed against 20.15 us for the same kernel in a replayed graph's trace).  Every sample is corrected by the
        # MEDIAN duration of the empty brackets recorded next to them (never below a quarter of the raw sample).
        empty = sorted(p['sec'] for p in prof if p['name'] == 'rscotr::empty_bracket')
        bracket_s = empty[len(empty) // 2] if empty else 0.0
        prof = [p for p in prof if p['name'] != 'rscotr::empty_bracket']
        for p in prof:
            p['raw_sec'] = p['sec']
            p['sec'] = max(p['sec'] - bracket_s, 0.25 * p['sec'])
    else:
        bracket_s = 0.0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        images = ntask * a.batch * world * a.steps
        # rooflines from HIP events recorded around the launches inside the timed region (on the launch
        # stream): algorithmic work of the sampled launches / their summed duration.
        def group(kind):
            g = {}
            for p in prof:
                if p['kind'] == kind:
                    d = g.setdefault(p['name'] if kind == 'gemm' else kind, [0.0, 0.0, 0])
                    d[0] += p['work']
                    d[1] += p['sec']
                    d[2] += 1
            return g

        def hbm(kind, kernel):
            g = group(kind).get(kind)
            if not g:
                return None
            ach = g[0] / g[1] / 1e9
            return dict(bound='hbm', achieved=ach, peak=HBM_PEAK_GBS, unit='GB/s', frac=ach / HBM_PEAK_GBS,
                        traffic=None, kernel=kernel, launches_sampled=g[2], avg_us=g[1] / g[2] * 1e6,
                        bytes_per_launch=g[0] / g[2])

        # dominant kernel of the step = the fp32 MFMA GEMM family; report the instantiation that takes the
        # most time, the whole family next to it
        gg = group('gemm')
        pmc_applies = a.workload == 'mtl512' and a.size == WORKLOADS['mtl512']['size'] and a.batch == WORKLOADS['mtl512']['batch']
        # (the fused attention core reports through the GEMM kind with DENSE-equivalent flops — fully masked tiles are skipped by
        # the kernel — and has its own two lines below: it stays out of the GEMM family's sums)
        ga = {k: v for k, v in gg.items() if 'attn_' in k}
        gg = {k: v for k, v in gg.items() if 'attn_' not in k}
        r_gemm, fam = None, None

        def is_x6(k):  # kernels whose inner product is the six-term bf16 split (priced against 2500 / 6)
            return 'bf16x6' in k or 'wplanes' in k or 'gemm_f32_group_kernel<6>' in k

        def is_h3(k):  # ... the three-term fp16 split (2500 / 3)
            return 'gemm_h3' in k or 'ffn_h3' in k or 'lin_h3' in k
        if gg:
            name, d = max(gg.items(), key=lambda kv: kv[1][1])
            ach = d[0] / d[1] / 1e12
            x6 = is_x6(name)  # (pre-split weight planes, the grouped launch's bf16x6 body: the same six-term product)
            split = is_h3(name) or x6
            peak = BF16X6_PEAK_TF if x6 else (H3_PEAK_TF if split else MFMA_F32_PEAK_TF)
            r_gemm = dict(bound='mfma', achieved=ach, peak=peak, unit='TFLOP/s', frac=ach / peak,
                          traffic=None, kernel=name, launches_sampled=d[2], avg_us=d[1] / d[2] * 1e6,
                          flops_per_launch=d[0] / d[2],
                          note=(('fp32 operands split into three bf16 planes in the kernel (all 24 significand bits), six '
                                 'v_mfma_f32_32x32x16_bf16 per k-step, fp32 accumulate — fp32-FMA-class error: achieved = '
                                 'fp32-equivalent 2MNK flops, peak = dense bf16 MFMA peak / 6; ' if x6 else
                                 'fp32 operands split into two power-of-two-scaled fp16 planes in the kernel (2^-24 relative), three '
                                 'v_mfma_f32_32x32x16_f16 per k-step, fp32 accumulate — fp32-FMA-class error: achieved = fp32-equivalent '
                                 '2MNK flops, peak = dense fp16 MFMA peak / 3; ')
                                if split else 'fp32 in / fp32 accumulate MFMA (v_mfma_f32_32x32x2_f32); ') + '1 launch in '
                               f'{PROF_EVERY_GEMM} sampled (events recorded inside the C entry, on the launch stream; the median duration of an empty event bracket, roofline_bracket_us, is subtracted from every sample); ' + (
                                   f'sampled in {a.roofline_rounds} eager round(s) run right after the timed region '
                                   '(the timed region replays hipGraphs), each iteration queued behind a '
                                   f'{a.roofline_hold_ms:.0f} ms stream hold so that its kernels run back to back'
                                   if runner.graphed else 'sampled inside the timed region'))
            if 'group_kernel' in name:
                r_gemm['note'] = ('ONE launch = every deferred weight gradient dW = A^T B of a backward pass with a small output or a '
                                  'short reduction (~100 problems, cut into 128 x 128 tiles x k-slices; flops_per_launch = sum of '
                                  '2 M N K over the problems of the launch, stated by the host that built the table); ') + r_gemm['note']
            # HBM traffic of that kernel: rocprofv3 PMC passes over this same command (scripts/gpu_pmc.sh), summary
            # committed under profiles/; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950
            try:  # (the committed PMC passes ran the default workload: no traffic figure for the others)
                with open(os.path.join(ROOT, 'profiles', 'pmc_gemm_traffic.json')) as fh:
                    pm = json.load(fh)['kernels'].get(name) if pmc_applies else None
                if pm:
                    r_gemm['traffic'] = (2.0 * pm['fetch_kib_per_launch'] + pm['write_kib_per_launch']) * 1024.0
                    r_gemm['traffic_note'] = ('bytes per launch = 2*FETCH_SIZE + WRITE_SIZE (KiB) averaged over the '
                                              f"{pm['dispatches']} launches of a PMC run of this command "
                                              '(profiles/pmc_gemm_traffic.json); operands + epilogue tensors, fp32')
            except (OSError, KeyError, ValueError):
                pass
            tf, tt = sum(v[0] for v in gg.values()), sum(v[1] for v in gg.values())
            sf3 = sum(v[0] for k, v in gg.items() if is_h3(k))
            sf6 = sum(v[0] for k, v in gg.items() if is_x6(k))
            sf = sf3 + sf6
            st = sum(v[1] for k, v in gg.items() if is_h3(k) or is_x6(k))
            # the family mixes the matrix pipes: its peak is the time the same flops would take at each kernel's own peak
            fam_peak = tf / (sf3 / H3_PEAK_TF + sf6 / BF16X6_PEAK_TF + (tf - sf) / MFMA_F32_PEAK_TF) if tf else MFMA_F32_PEAK_TF
            fam = dict(bound='mfma', achieved=tf / tt / 1e12, peak=fam_peak, unit='TFLOP/s',
                       frac=tf / tt / 1e12 / fam_peak,
                       kernel='rscotr GEMM family: gemm_f32_kernel<*>, gemm_bf16x6_kernel<*>, gemm_wplanes_kernel<*>, gemm_small_kernel<*>, '
                              'gemm_dw_direct_kernel<*>, gemm_f32_group_kernel<*> / gemm_h3_group_kernel (the grouped weight-gradient launch), gemm_h3_kernel<*> / gemm_h3_128_kernel<*>, ffn_h3_kernel<*> (the fused encoder FFN: two products per launch)',
                       launches_sampled=sum(v[2] for v in gg.values()), split_product_flop_share=sf / tf if tf else 0.0,
                       split_product_time_share=st / tt if tt else 0.0,
                       note='fp32-equivalent flops; peak = flop-weighted harmonic mix of 157.3 (fp32 pipe), 2500/6 (bf16x6) and '
                            '2500/3 (fp16 split product)')
        # the fused attention core (csrc/attn_core.hip) reports through the GEMM kind: its own two lines, fp32 matrix pipe
        def attn(name):
            d = ga.get(name)
            if not d:
                return None
            ach = d[0] / d[1] / 1e12
            return dict(bound='mfma', achieved=ach, peak=MFMA_F32_PEAK_TF, unit='TFLOP/s', frac=ach / MFMA_F32_PEAK_TF, traffic=None,
                        kernel=name, launches_sampled=d[2], avg_us=d[1] / d[2] * 1e6, flops_per_launch=d[0] / d[2],
                        note='DENSE-equivalent algorithmic flops (4 Lq Lk 32 per image and head forward, 10 backward; fully masked 32 x 32 '
                             'tiles are skipped by the kernel, so the issued work is lower; the backward recomputes the scores in both '
                             'of its passes: 14 issued per kept tile), v_mfma_f32_32x32x2_f32, all decoder shapes of the round mixed')
        r_af, r_ab = attn('rscotr::attn_fwd_kernel'), attn('rscotr::attn_bwd (dq + dkv kernels)')
        # the other families worth >= 1 ms of a round: LayerNorm, split-K combines, AdamW by their algorithmic
        # bytes against HBM; Swin window attention by its algorithmic flops against the fp32 matrix pipe
        def named(kind, names, bound, peak, unit, scale, work_of=None, note=None):
            w = t = n = 0
            for p in prof:
                if p['kind'] == kind and any(x in p['name'] for x in names):
                    w += p['work'] if work_of is None else work_of
                    t += p['sec']
                    n += 1
            if not n or not t:
                return None
            ach = w / t / scale
            r = dict(bound=bound, achieved=ach, peak=peak, unit=unit, frac=ach / peak, traffic=None, kernel=' + '.join(names),
                     launches_sampled=n, avg_us=t / n * 1e6)
            r['bytes_per_launch' if bound == 'hbm' else 'flops_per_launch'] = w / n
            if note:
                r['note'] = note
            return r
        r_ln = named('hbm', ['layernorm_fwd_kernel', 'layernorm_bwd_kernel'], 'hbm', HBM_PEAK_GBS, 'GB/s', 1e9,
                     note='8 (12 with the second output) bytes per element forward, 12 (16 with the residual gradient) backward')
        r_sk = named('hbm', ['gemm_splitk_reduce_kernel', 'splitk_flush_kernel'], 'hbm', HBM_PEAK_GBS, 'GB/s', 1e9,
                     note='slabs read + destination (and epilogue tensors) read / written')
        opt_ = runner.optimizer
        live_elems = float(sum((g['param'].numel() + 3) // 4 * 4 for g, l in zip(opt_.groups, opt_.live) if l))
        r_ad = named('hbm', ['adamw_clip_kernel'], 'hbm', HBM_PEAK_GBS, 'GB/s', 1e9, work_of=28.0 * live_elems,
                     note='16 bytes read (weight, gradient, two moments) + 12 written per stepped element; '
                          f'{int(live_elems)} elements stepped per iteration (torch 1.11 semantics: every tensor that has ever '
                          'received a gradient)')
        r_wa = named('mfma', ['swin_wattn_fwd_kernel', 'swin_wattn_bwd_kernel'], 'mfma', MFMA_F32_PEAK_TF, 'TFLOP/s', 1e12,
                     note='4 (forward) / 10 (backward) x 49 x 49 x 32 flop per (image, window, head) on the real tokens, '
                          'v_mfma_f32_32x32x2_f32; latency-bound per item (49 x 49 x 32 products on 64-row MFMA tiles)')
        # the fused two-Linear launches (round 6): by total time the largest kernel family of the round after the 64 x 64 split product
        r_ffn = named('gemm', ['ffn_h3_kernel'], 'mfma', H3_PEAK_TF, 'TFLOP/s', 1e12,
                      note='4 M C H fp32-equivalent flops per launch (both products; the few-row launches without their combine), three '
                           'v_mfma_f32_16x16x32_f16 per 32 k, peak = dense fp16 MFMA peak / 3; encoder FFN 256 -> 2048 -> 256 (ReLU), the '
                           'detection decoder\'s, Swin stage 1-3 MLPs (GELU, the norm in front as the prologue), forward and mirrored backward')
        r_f = hbm('msda_fwd', 'rscotr::msda_fwd_kernel<32, 4>')
        r_b = hbm('msda_bwd', 'rscotr_msda_bwd (sample + tile + combine kernels)')
        try:  # HBM bytes per call from the same PMC passes (every msda_* kernel of the backward entry summed)
            with open(os.path.join(ROOT, 'profiles', 'pmc_gemm_traffic.json')) as fh:
                pk = json.load(fh)['kernels'] if pmc_applies else {}
            tb = lambda v: (2.0 * v['fetch_kib_per_launch'] + v['write_kib_per_launch']) * 1024.0
            fw = [v for k, v in pk.items() if 'msda_fwd_kernel' in k]
            if r_f and fw:
                r_f['traffic'] = sum(tb(v) * v['dispatches'] for v in fw) / sum(v['dispatches'] for v in fw)
            bw = [v for k, v in pk.items() if 'msda_' in k and 'msda_fwd_kernel' not in k and 'msda_prep' not in k]
            calls = max([v['dispatches'] for k, v in pk.items() if 'msda_bwd_kernel' in k] or [0])
            if r_b and bw and calls:
                r_b['traffic'] = sum(tb(v) * v['dispatches'] for v in bw) / calls
        except (OSError, KeyError, ValueError):
            pass
        out = dict(metric=wl['metric'], value=images / dt, unit='images/s',
                   n_gpus=world, steps=a.steps, warmup=a.warmup, ms_per_step=dt / a.steps * 1e3,
                   higher_is_better=True, scaling='weak', vs_baseline=None, dtype=('f32' if lib.rscotr_gemm_get_precision() == 0 else
                          'f32 (large products as three fp16 MFMAs on two-plane power-of-two-scaled splits of the fp32 operands, fp32 '
                          'accumulate: fp32-FMA-class error; six bf16 MFMAs on three-plane splits where an operand has no range word)'
                          if ops.RANGES.enabled else
                          'f32 (large products as six bf16 MFMAs on three-plane splits of the fp32 operands, fp32 accumulate: '
                          'fp32-FMA-class error)'),
                   data='synthetic',
                   config=dict(workload=f'{a.workload}: {wl["name"]}, {a.size}x{a.size} bs={a.batch}/task/GPU',
                               step=('one round-robin round = ' + '+'.join(wl['tasks']) + ' train iterations') if ntask > 1
                               else f'one {wl["tasks"][0]} train iteration',
                               images_per_step=ntask * a.batch * world, parallelism=f'dp{world}',
                               optimizer='AdamW+clip0.1 (fused HIP)', precision='fp32',
                               gemm_precision_mode={0: 'fp32', 3: 'bf16x6'}[lib.rscotr_gemm_get_precision()] + ('+h3' if ops.RANGES.enabled else ''),
                               rccl_ranks=dist.get_world_size() if dist.is_initialized() else 0,
                               exchange=(('inline' if os.environ.get('RSCOTR_DIST_INLINE', '1') != '0' else 'overlap')
                                         if dist.is_initialized() else None),
                               hipgraph_tasks=list(runner.graphed.keys())),
                   roofline=r_gemm, roofline_gemm_family=fam, roofline_msda_fwd=r_f, roofline_msda_bwd=r_b,
                   roofline_attn_fwd=r_af, roofline_attn_bwd=r_ab, roofline_swin_wattn=r_wa, roofline_layernorm=r_ln,
                   roofline_splitk=r_sk, roofline_adamw=r_ad, roofline_ffn=r_ffn,
                   roofline_bracket_us=round(bracket_s * 1e6, 3),  # median empty event bracket, subtracted from every roofline sample
                   per_task_ms=per_task)  # rank 0, device time per iteration inside the timed region
        if world == 1 and not a.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(a.size, a.batch, a.cpu_rounds, a.workload)
        else:
            out['cpu_baseline'] = None
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
