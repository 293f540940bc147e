"""Are the hand-scheduled kernels right with another stream's kernels running beside them?  (The row-staged convolution's version 2 was not:
tests/tools/exp/ds2_beside_dbg.py.)  Each kernel's result alone is the reference; then it runs several times while a side stream runs a different
kernel, and the results must be bit-identical (atomically accumulated ones: equal to rounding).
  * gemm16 8-phase (configuration 15, counted vmcnt + global_load_lds) on the las_large shapes, beside another bf16 product and beside an f32 split product
  * the f32 split-product GEMM beside a bf16 product
  * row-staged convolution version 1 (deepspeech conv3 forward) beside a GEMM"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch

from speech_recognition_amd import ops

g = torch.Generator().manual_seed(0)
side = torch.cuda.Stream()
R, D, G = 31936, 2048, 4096
x16 = torch.randn(R, D, generator=g).cuda().bfloat16()
w16 = torch.randn(G, D, generator=g).cuda().bfloat16()
xt16 = torch.randn(D, R, generator=g).cuda().bfloat16()
dst16 = torch.randn(G, R, generator=g).cuda().bfloat16()
a32 = torch.randn(7968, 512, generator=g).cuda()
b32 = torch.randn(512, 1024, generator=g).cuda()


def beside(name, main_fn, side_fn, out, exact=True, reps=6):
    out.zero_()
    main_fn()
    torch.cuda.synchronize()
    ref = out.clone()
    worst = 0.0
    for it in range(reps):
        out.zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            side_fn()
            side_fn()
        main_fn()
        torch.cuda.synchronize()
        d = float((out - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)
        worst = max(worst, d if d == d else float("inf"))
    ok = worst == 0.0 if exact else worst < 1e-5
    print(f"{name:70s} worst relative difference {worst:.3e}  {'ok' if ok else 'DIFFERENT'}", flush=True)


c = torch.zeros(R, G, device="cuda")
c2 = torch.zeros(R, G, device="cuda")
gW = torch.zeros(D, G, device="cuda")
gW2 = torch.zeros(D, G, device="cuda")
c32 = torch.zeros(7968, 1024, device="cuda")
c32b = torch.zeros(7968, 1024, device="cuda")
beside("gemm16 forward product beside the weight-gradient product", lambda: ops.gemm_bf16_nt(x16, w16, c), lambda: ops.gemm_bf16_nt(xt16, dst16, gW2), c)
beside("gemm16 weight-gradient product (split K) beside the forward product", lambda: ops.gemm_bf16_nt(xt16, dst16, gW), lambda: ops.gemm_bf16_nt(x16, w16, c2), gW, exact=False)
beside("gemm16 forward product beside f32 split products", lambda: ops.gemm_bf16_nt(x16, w16, c), lambda: [ops.gemm(a32, b32, c32b) for _ in range(4)], c)
beside("f32 split product beside a gemm16 product", lambda: ops.gemm(a32, b32, c32), lambda: ops.gemm_bf16_nt(x16, w16, c2), c32)
# row-staged convolution, version 1: deepspeech conv3 forward
x2 = torch.randn(16, 355, 25, 32, generator=g).cuda()
w2 = (torch.randn(21, 11, 32, 96, generator=g) * 0.05).cuda()
y3 = torch.zeros(16, 168, 15, 96, device="cuda")
beside("row-staged convolution (version 1, conv3 forward) beside a gemm16 product", lambda: ops.conv2d_fwd(x2, w2, None, (2, 1), y=y3), lambda: ops.gemm_bf16_nt(x16, w16, c2), y3)
x1 = torch.randn(16, 730, 35, 32, generator=g).cuda()
w1 = (torch.randn(21, 11, 32, 32, generator=g) * 0.05).cuda()
y2 = torch.zeros(16, 355, 25, 32, device="cuda")
beside("row-staged convolution (conv2 forward, the default version) beside f32 split products", lambda: ops.conv2d_fwd(x1, w1, None, (2, 1), y=y2),
       lambda: [ops.gemm(a32, b32, c32b) for _ in range(6)], y2)
