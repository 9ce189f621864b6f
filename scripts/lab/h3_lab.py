"""fp16 split product (h3) against the six-term bf16 product and the fp32 pipe: error against fp64 and time per launch
at the step's routed shapes.  `python scripts/lab/h3_lab.py [quick]`"""
import json
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rscotr_amd import ops  # noqa: E402
from rscotr_amd._lib import lib  # noqa: E402

dev = torch.device('cuda:0')
ops.RANGES.enabled = False  # (ranges are passed explicitly below)


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
    return e0.elapsed_time(e1) / n * 1e3  # us


def amax(x):
    slot = ops.RANGES.new_slot(dev)
    lib.call('rscotr_amax_f32', x.data_ptr(), x.shape[0], x.shape[1], x.shape[1], slot, torch.cuda.current_stream().cuda_stream)
    got = ops.RANGES.word(slot)[0]
    assert float(got) == float(x.abs().max()), (float(got), float(x.abs().max()))
    return slot


shapes = [  # (M, N, K, ak, bk, tag)
    (10880, 256, 256, 0, 0, 'enc proj fwd'), (10880, 256, 256, 0, 1, 'enc proj dx'),
    (10880, 2048, 256, 0, 0, 'enc ffn1'), (10880, 256, 2048, 0, 0, 'enc ffn2'), (10880, 2048, 256, 0, 1, 'ffn2 dx'),
    (10880, 256, 2048, 0, 1, 'ffn1 dx'), (256, 2048, 10880, 1, 1, 'ffn2 dW'), (2048, 256, 10880, 1, 1, 'ffn1 dW'),
    (8192, 768, 192, 0, 0, 'swin2 fc1'), (8192, 192, 768, 0, 0, 'swin2 fc2'), (8192, 576, 192, 0, 0, 'swin2 qkv'),
    (2048, 1536, 384, 0, 0, 'swin3 fc1'), (2048, 384, 1536, 0, 0, 'swin3 fc2'), (2048, 384, 1536, 0, 1, 'swin3 fc1 dx'),
    (512, 3072, 768, 0, 0, 'swin4 fc1'), (32768, 384, 96, 0, 0, 'swin1 fc1 (K<192: fp32 pipe)'),
    (4096, 4096, 4096, 0, 0, 'square 4k'),
]
if len(sys.argv) > 1 and sys.argv[1] == 'quick':
    shapes = shapes[:4]
scales = [(1.0, 1.0), (1e-6, 3e-2)] if 'scales' in sys.argv else [(1.0, 1.0)]
for M, N, K, ak, bk, tag in shapes:
    for sa, sb in scales:
        g = torch.Generator().manual_seed(M + N + K)
        A = (torch.randn((K, M) if ak else (M, K), generator=g) * sa).to(dev)
        B = (torch.randn((K, N) if bk else (N, K), generator=g) * sb).to(dev)
        if 'tail' in sys.argv:  # heavy-tailed operand: a few entries 1e4 x the rest
            A.view(-1)[::9973] *= 1e4
        ref = ((A.double().t() if ak else A.double()) @ (B.double() if bk else B.double().t()))
        sA, sB = amax(A), amax(B)
        lda, ldb = A.shape[1], B.shape[1]
        r = dict(tag=tag, M=M, N=N, K=K, ak=ak, bk=bk, sa=sa, sb=sb)

        def err(o):
            return float((o.double() - ref).abs().max() / ref.abs().max())
        o6 = ops.gemm(A, B, M, N, K, lda, ldb, ak, bk)
        oh = ops.gemm(A, B, M, N, K, lda, ldb, ak, bk, amax_a=sA, amax_b=sB)
        prev = lib.rscotr_gemm_get_precision()
        lib.rscotr_gemm_set_precision(0)
        o32 = ops.gemm(A, B, M, N, K, lda, ldb, ak, bk)
        t32 = t(lambda: ops.gemm(A, B, M, N, K, lda, ldb, ak, bk))
        lib.rscotr_gemm_set_precision(prev)
        r['err_x6'], r['err_h3'], r['err_f32'] = err(o6), err(oh), err(o32)
        r['us_x6'] = t(lambda: ops.gemm(A, B, M, N, K, lda, ldb, ak, bk))
        r['us_h3'] = t(lambda: ops.gemm(A, B, M, N, K, lda, ldb, ak, bk, amax_a=sA, amax_b=sB))
        r['us_f32'] = t32
        so = ops.RANGES.new_slot(dev)
        r['us_h3_out'] = t(lambda: ops.gemm(A, B, M, N, K, lda, ldb, ak, bk, amax_a=sA, amax_b=sB, amax_out=so))
        def cold():  # (a fresh, zero slot each time: every wavefront's atomic lands)
            ops.RANGES.buf[:, ops.RANGES.index(so)].zero_()
            ops.gemm(A, B, M, N, K, lda, ldb, ak, bk, amax_a=sA, amax_b=sB, amax_out=so)
        r['us_h3_out_cold'] = t(cold) - t(lambda: ops.RANGES.buf[:, ops.RANGES.index(so)].zero_())
        r['us_amax_a'] = t(lambda: lib.call('rscotr_amax_f32', A.data_ptr(), A.shape[0], A.shape[1], A.shape[1], sA,
                                            torch.cuda.current_stream().cuda_stream))
        r['speedup'] = r['us_x6'] / r['us_h3']
        print(json.dumps({k: (float(f'{v:.3g}') if isinstance(v, float) else v) for k, v in r.items()}), flush=True)
