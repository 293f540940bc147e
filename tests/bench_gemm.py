"""GPU micro-benchmark of asr_gemm_f32 on the LAS training-step shapes (not a test):
python tests/bench_gemm.py  -> TFLOP/s per shape (HIP events, 20 reps)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from speech_recognition_amd import ops

SHAPES = [  # (name, M, N, K, ta, tb, split_k)
    ("enc L0 input  NN", 7968, 1024, 608, 0, 0, 1), ("enc L1 input  NN", 7968, 1024, 512, 0, 0, 1), ("proj          NN", 7968, 512, 512, 0, 0, 1),
    ("vocab         NN", 2048, 16000, 256, 0, 0, 1), ("keys          NN", 7968, 256, 512, 0, 0, 1),
    ("enc dX        NT", 7968, 608, 1024, 0, 1, 1), ("vocab dY      NT", 2048, 256, 16000, 0, 1, 8),
    ("enc dW        TN", 608, 1024, 7968, 1, 0, 19), ("vocab dW      TN", 256, 16000, 2048, 1, 0, 3), ("proj dW       TN", 512, 512, 7968, 1, 0, 48),
    ("square 4096   NN", 4096, 4096, 4096, 0, 0, 1),
    ("large L1 in   NN", 31936, 4096, 2048, 0, 0, 1), ("large proj    NN", 31936, 2048, 2048, 0, 0, 1), ("large dX      NT", 31936, 2048, 4096, 0, 1, 1),
    ("large dW      TN", 2048, 4096, 31936, 1, 0, 4),
]


def main():
    compute = 1 if "--bf16" in sys.argv else 0
    ops.set_mixed_precision(bool(compute))
    print("compute:", "bf16 operands" if compute else "f32")
    for name, M, N, K, ta, tb, sk in SHAPES:
        a = torch.randn((K, M) if ta else (M, K), device="cuda")
        b = torch.randn((N, K) if tb else (K, N), device="cuda")
        c = torch.zeros(M, N, device="cuda")
        kw = dict(trans_a=bool(ta), trans_b=bool(tb), accumulate=1 if sk > 1 else 0, split_k=sk)
        for _ in range(3):
            ops.gemm(a, b, c, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            ops.gemm(a, b, c, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"{name}  M={M:5d} N={N:5d} K={K:5d} sk={sk:2d}  {ms*1e3:8.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TFLOP/s")


if __name__ == "__main__":
    main()
