"""The dense products of a las_small training step (tests/tools/gemm_shapes.py) under the three evaluations of an f32 product:
f32 MFMA (compute 0), nine bf16 pair products (2), six (3).  python tests/tools/bench_gemm_modes.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench
from speech_recognition_amd import ops

shapes = [("enc input proj NN", 7968, 1024, 512, 0, 0, 1), ("enc dW TN", 512, 1024, 7968, 1, 0, 16), ("enc dx NT", 7968, 512, 1024, 0, 1, 1),
          ("proj NN", 7968, 512, 512, 0, 0, 1), ("proj dW TN", 512, 512, 7968, 1, 0, 31), ("vocab NN", 2048, 16000, 256, 0, 0, 1),
          ("vocab dY NT", 2048, 256, 16000, 0, 1, 15), ("vocab dW TN", 256, 16000, 2048, 1, 0, 2), ("square 4096", 4096, 4096, 4096, 0, 0, 1),
          ("ds2 proj NN", 2688, 384, 256, 0, 0, 1), ("ds2 char NN", 2688, 16000, 256, 0, 0, 1)]
print(f"{'product':22s} {'M':>6s} {'N':>6s} {'K':>6s}   f32 MFMA            nine pairs          six pairs")
for name, M, N, K, ta, tb, sk in shapes:
    a = torch.randn((K, M) if ta else (M, K), device="cuda")
    b = torch.randn((N, K) if tb else (K, N), device="cuda")
    c = torch.zeros(M, N, device="cuda")
    row = f"{name:22s} {M:6d} {N:6d} {K:6d} "
    for compute in (0, 2, 3):
        fn = lambda: ops.gemm(a, b, c, trans_a=bool(ta), trans_b=bool(tb), accumulate=1 if sk > 1 else 0, split_k=sk, compute=compute)
        t = bench.time_kernel(torch.cuda.current_stream(), fn, iters=20)
        row += f"  {t * 1e6:7.1f} us {2.0 * M * N * K / t / 1e12:6.1f} TF"
    print(row)
