"""LAB: planes x planes product (both operands pre-split) against the shipped routes.  Needs the lab kernel: `git apply scripts/lab/pplanes_kernel.patch` and rebuild first (the entry rscotr_gemm_f32_pplanes is not in the tree).  python scripts/lab/pplanes_lab.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from rscotr_amd import ops
from rscotr_amd._lib import lib
dev = torch.device('cuda:0')
lib.rscotr_gemm_set_precision(3)
raw = ctypes.CDLL(os.path.join(ROOT, 'rscotr_amd', 'librscotr.so'))
raw.rscotr_gemm_f32_pplanes.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p] + [ctypes.c_int] * 4 + \
    [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
SHAPES = [(10880, 2048, 256), (10880, 256, 256), (10880, 384, 256), (10880, 256, 2048), (32768, 384, 96), (8192, 768, 192), (2048, 1536, 384),
          (2048, 384, 384), (2048, 384, 1536)]
flush = torch.empty(160 * 1024 * 1024, device=dev)
s = torch.cuda.current_stream().cuda_stream


def planes_of(W):
    N, K = W.shape
    npad = (N + 255) // 256 * 256
    pl = torch.empty(npad * K * 3, dtype=torch.int16, device=dev)
    table = torch.from_numpy(np.asarray([[W.data_ptr(), pl.data_ptr(), N, K, K, npad, 0, 0]], dtype=np.int64)).to(dev)
    lib.call('rscotr_gemm_split_weights', table.data_ptr(), 1, (npad * (K // 16) + 255) // 256, s)
    return pl, npad


def timeit(fn, cold, n=12):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for i in range(n):
        if cold:
            flush.fill_(float(i))
        ev[i][0].record(); fn(); ev[i][1].record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev[2:])
    return t[len(t) // 2]


for M, N, K in SHAPES:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05
    bias = torch.randn(N, device=dev); out = torch.empty(M, N, device=dev); out2 = torch.empty(M, N, device=dev)
    pa, mpad = planes_of(A)
    pb, npad = planes_of(W)
    def pp():
        e = raw.rscotr_gemm_f32_pplanes(pa.data_ptr(), mpad, pb.data_ptr(), npad, out2.data_ptr(), M, N, K, N, bias.data_ptr(), 1, None, None,
                                        None, s)
        assert e == 0, e
    ops.WPLANES.enabled = False
    def old():
        ops.gemm(A, W, M, N, K, K, K, 0, 0, out=out, bias=bias, act=1)
    old(); pp()
    torch.cuda.synchronize()
    ref = torch.relu(A.double() @ W.double().t() + bias.double())
    err = float((out2.double() - ref).abs().max() / ref.abs().max())
    same = bool(torch.equal(out, out2))
    r = [timeit(old, False), timeit(pp, False), timeit(old, True), timeit(pp, True)]
    print(f'M={M:6d} N={N:5d} K={K:5d}  hot: in-kernel {r[0]:7.1f} us  planes x planes {r[1]:7.1f} us   cold: {r[2]:7.1f} / {r[3]:7.1f} us   '
          f'err {err:.1e} bit-identical {same}', flush=True)
