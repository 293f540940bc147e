"""GEMM time against K at fixed M x N (not a test): separates the per-tile fixed cost from the per-k-tile cost.
python tests/tools/gemm_k_sweep.py [M N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from speech_recognition_amd import ops

M, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (7968, 2048)
for K in (32, 64, 128, 256, 512, 1024, 2048, 4096):
    a, b, c = torch.randn(M, K, device="cuda"), torch.randn(K, N, device="cuda"), torch.empty(M, N, device="cuda")
    for _ in range(3):
        ops.gemm(a, b, c)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.gemm(a, b, c)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f"M={M} N={N} K={K:5d}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s")
