"""Micro-benchmark: the tile configurations of asr_gemm_bf16_nt (ASR_G16_CFG numbers) on the las_large product shapes, bf16 images
already made.  Prints ms and TFLOP/s per (shape, configuration, split_k) and the worst error against configuration 0."""
import sys, torch
sys.path.insert(0, ".")
from speech_recognition_amd import ops

def tm(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

cfgs = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 8, 12, 15]
shapes = [("fwd  [31936x2048]x[4096x2048]^T", 31936, 4096, 2048, (1,)),
          ("dX   [31936x4096]x[2048x4096]^T", 31936, 2048, 4096, (1,)),
          ("dW   [2048x31936]x[4096x31936]^T", 2048, 4096, 31936, (1, 2, 4)),
          ("fwd1 [31936x1024]x[4096x1024]^T", 31936, 4096, 1024, (1,)),
          ("voc  [8128x1024]x[16000x1024]^T", 8128, 16000, 1024, (1,)),
          ("sq   [8192x8192]x[8192x8192]^T", 8192, 8192, 8192, (1,)),
          ("sq4k [4096x4096]x[4096x4096]^T", 4096, 4096, 4096, (1,))]
for name, M, N, K, sks in shapes:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    ref = None
    for cfg in cfgs:
        for sk in sks:
            old = ops.lib().asr_gemm_bf16_config(cfg)
            c = torch.zeros(M, N, device="cuda")
            ops.gemm_bf16_nt(a, b, c, accumulate=1 if sk > 1 else 0, split_k=sk)
            if ref is None: ref = c.clone()
            err = float((c - ref).abs().max() / ref.abs().max())
            t = tm(lambda: ops.gemm_bf16_nt(a, b, c, accumulate=1 if sk > 1 else 0, split_k=sk))
            ops.lib().asr_gemm_bf16_config(old)
            print(f"{name:36s} cfg {cfg:3d} split_k {sk}: {t:.3f} ms  {2.0*M*N*K/t/1e9:7.0f} TF   err vs first {err:.1e}", flush=True)
