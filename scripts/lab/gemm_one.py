"""One GEMM shape, a few launches (for rocprofv3 --pmc runs): python gemm_one.py M N K a_kmajor b_kmajor"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from rscotr_amd import ops
M, N, K, ak, bk = (int(v) for v in sys.argv[1:6])
dev = torch.device('cuda:0')
A = torch.randn((K, M) if ak else (M, K), device=dev)
B = torch.randn((K, N) if bk else (N, K), device=dev)
for _ in range(5):
    ops.gemm(A, B, M, N, K, A.shape[1], B.shape[1], ak, bk)
torch.cuda.synchronize()
