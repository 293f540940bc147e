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
# the pieces alone
ops._bf16_images["on"] = True
x = torch.randn(R, D, device="cuda"); W = torch.randn(D, G, device="cuda"); dsb = torch.randn(R, G, device="cuda")
x16 = torch.zeros(R, D, device="cuda", dtype=torch.bfloat16); w16 = torch.zeros(G, D, device="cuda", dtype=torch.bfloat16)
xt16 = torch.zeros(D, R, device="cuda", dtype=torch.bfloat16); dst16 = torch.zeros(G, R, device="cuda", dtype=torch.bfloat16)
c = torch.zeros(R, G, device="cuda"); gW = torch.zeros(D, G, device="cuda")
print(f"image x straight [R,2048]      {tm(lambda: ops.f32_to_bf16_image(x, x16)):.3f} ms")
print(f"image W transposed [2048,4096] {tm(lambda: ops.f32_to_bf16_image(W, w16, transpose=True)):.3f} ms")
print(f"image x transposed [R,2048]    {tm(lambda: ops.f32_to_bf16_image(x, xt16, transpose=True)):.3f} ms")
print(f"image ds transposed [R,4096]   {tm(lambda: ops.f32_to_bf16_image(dsb, dst16, transpose=True)):.3f} ms")
t = tm(lambda: ops.gemm_bf16_nt(x16, w16, c)); print(f"gemm16 fwd alone  {t:.3f} ms ({2.0*R*D*G/t/1e9:.0f} TF)")
t = tm(lambda: ops.gemm_bf16_nt(xt16, dst16, gW)); print(f"gemm16 dW alone   {t:.3f} ms ({2.0*R*D*G/t/1e9:.0f} TF)")
