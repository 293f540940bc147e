"""Micro-benchmark: las_large products through the f32-operand kernel (bf16 fragments) vs bf16 images + gemm16."""
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

R, D, G = 31936, 2048, 4096
shapes = [("fwd  x[R,2048] W[2048,4096]", (R, D), (D, G), False, False),
          ("dX   ds[R,4096] W[2048,4096]^T", (R, G), (D, G), False, True),
          ("dW   x[R,2048]^T ds[R,4096]", (R, D), (R, G), True, False),
          ("fwd1 x[R,1024]... W[1024,4096]", (R, 1024), (1024, G), False, False)]
for name, sa, sb, ta, tb in shapes:
    a = torch.randn(*sa, device="cuda"); b = torch.randn(*sb, device="cuda")
    M = sa[1] if ta else sa[0]; N = sb[0] if tb else sb[1]; K = sa[0] if ta else sa[1]
    c = torch.zeros(M, N, device="cuda")
    res = []
    for on in (False, True):
        ops._bf16_images["on"] = on
        res.append(tm(lambda: ops.gemm(a, b, c, trans_a=ta, trans_b=tb, compute=1)))
    fl = 2.0 * M * N * K
    print(f"{name:34s} f32-operand {res[0]:.3f} ms ({fl/res[0]/1e9:.0f} TF)   images+bf16 {res[1]:.3f} ms ({fl/res[1]/1e9:.0f} TF)", flush=True)
# the batch-flattened dU
B, T, H = 64, 499, 1024
hs = torch.randn(B, T, H, device="cuda"); ds = torch.randn(B, T, 4 * H, device="cuda"); gU = torch.zeros(H, 4 * H, device="cuda")
for on in (False, True):
    ops._bf16_images["on"] = on
    t = tm(lambda: ops.gemm(hs[:, :T - 1], ds[:, 1:], gU, trans_a=True, accumulate=1, compute=1))
    print(f"dU batch-flattened images={on}: {t:.3f} ms ({2.0*H*4*H*B*(T-1)/t/1e9:.0f} TF)", flush=True)
