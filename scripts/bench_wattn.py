"""Micro-benchmark of the Swin window-attention kernels (direct C-ABI calls, back-to-back launches)
per stage of Swin-T at 512x512, B=2."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rscotr_amd._lib import lib
dev = torch.device('cuda:0')
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
s = torch.cuda.current_stream().cuda_stream
for (H, heads) in [(128, 3), (64, 6), (32, 12), (16, 24)]:
    C = heads * 32; B = 2
    qkv = torch.randn(B, H * H, 3 * C, device=dev)
    qb = torch.randn(3 * C, device=dev); tb = torch.randn(169, heads, device=dev)
    go = torch.randn(B, H * H, C, device=dev)
    out = torch.empty(B, H * H, C, device=dev); dqkv = torch.empty_like(qkv)
    dqb = torch.zeros_like(qb); dtb = torch.zeros_like(tb)
    nws = lib.rscotr_swin_wattn_bwd_workspace(B, H, H, C, heads); wsb = torch.empty(max(nws, 4), dtype=torch.uint8, device=dev)
    for shift in (0, 3):
        f = t(lambda: lib.call('rscotr_swin_wattn_fwd', qkv.data_ptr(), qb.data_ptr(), tb.data_ptr(), out.data_ptr(), B, H, H, C, heads, 7, shift, 0, s))
        b = t(lambda: lib.call('rscotr_swin_wattn_bwd', qkv.data_ptr(), qb.data_ptr(), tb.data_ptr(), go.data_ptr(), dqkv.data_ptr(), dqb.data_ptr(), dtb.data_ptr(), B, H, H, C, heads, 7, shift, out.data_ptr(), wsb.data_ptr(), nws, 0, s))
        print(json.dumps(dict(H=H, heads=heads, shift=shift, fwd_us=round(f, 1), bwd_us=round(b, 1),
                              phases=os.environ.get('RSCOTR_WATTN_PHASES', '4'))), flush=True)
