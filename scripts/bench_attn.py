"""Fused attention core (csrc/attn_core.hip) at the decoders' shapes: us per launch of forward / backward through the C ABI,
next to the unfused chain (batched q k^T -> softmax -> P v and its five-launch backward) timed through ops.mha's core only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rscotr_amd._lib import lib

dev = torch.device('cuda:0')
st = torch.cuda.current_stream().cuda_stream


def t(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


ONLY = sys.argv[1] if len(sys.argv) > 1 else ''
for tag, B, H, Lq, Lk, mode in (('det self-attn', 2, 8, 800, 800, 1), ('seg cross 64x64', 2, 8, 100, 4096, 2),
                                ('seg cross 32x32', 2, 8, 100, 1024, 2), ('seg cross 16x16', 2, 8, 100, 256, 2),
                                ('seg self-attn', 2, 8, 100, 100, 0), ('det800 self-attn', 4, 8, 1100, 1100, 1)):
    if ONLY and ONLY not in tag:
        continue
    C = H * 32
    q, k, v, do = (torch.randn(B, L, C, device=dev) for L in (Lq, Lk, Lk, Lq))
    mask = None if mode == 0 else (torch.rand({1: (Lq, Lk), 2: (B, Lq, Lk)}[mode], device=dev) < 0.3)
    mp = 0 if mask is None else mask.data_ptr()
    out, lse = torch.empty(B, Lq, C, device=dev), torch.empty(B, H, Lq, device=dev)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    nws = lib.rscotr_attn_core_workspace(B, H, Lq, Lk)
    ws = torch.empty(max(nws, 16) // 4, device=dev)
    sc = 32 ** -0.5
    f = lambda: lib.call('rscotr_attn_core_fwd', q.data_ptr(), k.data_ptr(), v.data_ptr(), mp, mode, out.data_ptr(), lse.data_ptr(),
                         B, H, Lq, Lk, 32, C, C, C, C, sc, ws.data_ptr(), nws, st)
    g = lambda: lib.call('rscotr_attn_core_bwd', q.data_ptr(), k.data_ptr(), v.data_ptr(), mp, mode, out.data_ptr(), do.data_ptr(),
                         lse.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), B, H, Lq, Lk, 32, C, C, C, C, C, C, C, sc,
                         ws.data_ptr(), nws, st)
    fl = 4.0 * B * H * Lq * Lk * 32
    tf, tb = t(f), t(g)
    print(f'{tag:18s} B={B} Lq={Lq:5d} Lk={Lk:5d}  fwd {tf:7.1f} us ({fl / tf / 1e6:6.1f} TF/s)   bwd {tb:7.1f} us ({3.5 * fl / tb / 1e6:6.1f} TF/s incl. recompute)',
          flush=True)
