"""Timing experiments on the eight-phase product (ASR_G16_8P_ABL, set per process): one shape, configuration 15."""
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
ops.lib().asr_gemm_bf16_config(int(sys.argv[1]) if len(sys.argv) > 1 else 15)
for M, N, K in ((8192, 8192, 8192), (4096, 4096, 4096), (31936, 2048, 4096)):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16); b = torch.randn(N, K, device="cuda").to(torch.bfloat16)
    c = torch.zeros(M, N, device="cuda")
    t = tm(lambda: ops.gemm_bf16_nt(a, b, c))
    print(f"  [{M}x{K}]x[{N}x{K}]^T: {t:.3f} ms {2.0*M*N*K/t/1e9:7.0f} TF", flush=True)
