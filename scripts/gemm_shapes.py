"""Per-shape GEMM census of one eager co-training round with the in-library launch-site profiler (HIP events recorded
inside the C entry: no host time in the durations): time, TFLOP/s and the shape's own roofline (max of MFMA time at
157.3 TFLOP/s and operand+result bytes at 5 TB/s).  RSCOTR_PROF_SHAPES=1 python scripts/gemm_shapes.py"""
import collections, copy, ctypes, os, sys
os.environ['RSCOTR_PROF_SHAPES'] = '1'
os.environ['RSCOTR_GRAPHS'] = '0'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rscotr_amd import Config, MODELS
from rscotr_amd._lib import lib
from rscotr_amd.data import build_synthetic_multidataloader
from rscotr_amd.runner import build_runner
CFG = os.path.join(ROOT, 'configs', 'multi', 'MTL_slvlcls_swin-t-p4-w7_1x1_resisc&dior&potsdam.py')
dev = torch.device('cuda:0')
cfg = Config.fromfile(CFG)
torch.manual_seed(0); np.random.seed(2022)
model = MODELS.build(copy.deepcopy(cfg.model)); model.init_weights(); model.to(dev).train()
runner = build_runner(model, cfg, build_synthetic_multidataloader(cfg, dev, size=512, batch_size=2, rank=0))
for _ in range(6): runner.train_iter()
torch.cuda.synchronize()
lib.call('rscotr_prof_enable', 1, 0, 0, 16384)
# every eager iteration queues up behind a ~60 ms stream hold (as bench.py's roofline rounds): its kernels then run back to back
# on a busy, clocked-up GPU as they do in the replayed graphs, instead of each starting on an idle one
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(20_000_000); e1.record(); torch.cuda.synchronize()
hold = int(20_000_000 * 60.0 / max(e0.elapsed_time(e1), 1e-3))
for _ in range(6):
    torch.cuda._sleep(hold)
    runner.train_iter()
torch.cuda.synchronize()
n = lib.rscotr_prof_pause()
kind, work, ms = ctypes.c_int(), ctypes.c_double(), ctypes.c_float()
name = ctypes.create_string_buffer(128)
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for i in range(n):
    lib.call('rscotr_prof_get', i, ctypes.byref(kind), ctypes.byref(work), ctypes.byref(ms), name, 128)
    a = agg[name.value.decode()]
    a[0] += 1; a[1] += ms.value * 1e3; a[2] += work.value
lib.call('rscotr_prof_disable')
tot = sum(v[1] for v in agg.values())
print(f'{n} launches in 2 rounds, {tot / 2e3:.2f} ms of GEMM per round (incl. split-K combine)')
rows = []
for k, (c, us, fl) in agg.items():
    parts = dict(p.split('=') for p in k.split() if '=' in p)
    if 'M' not in parts:  # (the grouped weight-gradient launch: many problems, the work is what the host stated)
        rows.append((us / 2, c // 2, us / c, fl / (us * 1e-6) / 1e12, fl / c / 157.3e12 * 1e6, k))
        continue
    M, N, K = int(parts['M']), int(parts['N']), int(parts['K'])
    ideal = max(2.0 * M * N * K / 157.3e12, 4.0 * (M * K + N * K + M * N) / 5e12) * 1e6
    rows.append((us / 2, c // 2, us / c, fl / (us * 1e-6) / 1e12, ideal, k))
rows.sort(reverse=True)
print('  us/round calls  us/call   TF/s  ideal_us  eff   shape')
for us, c, per, tf, ideal, k in rows[:int(os.environ.get('ROWS', 70))]:
    print(f'{us:9.0f} {c:5d} {per:8.1f} {tf:6.1f} {ideal:8.1f} {ideal / per:5.2f}   {k}')
