"""GPU: the hand-scheduled kernels give the same result with another stream's kernels running beside them.

Round 4 found a kernel that did not (version 2 of the row-staged convolution: single workgroups returned garbage with filter-gradient kernels
beside it, tests/tools/exp/ds2_beside_dbg.py; it is no longer on the training path).  Everything a training step may run beside something
else - its own side stream's products, RCCL's kernels under data parallelism - is checked here: the result alone is the reference, then the
kernel runs several times while a side stream runs a different kernel, and the results must be bit-identical (atomically accumulated ones:
equal to rounding)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _beside(main_fn, side_fn, out, exact=True, reps=5):
    side = torch.cuda.Stream()
    out.zero_()
    main_fn()
    torch.cuda.synchronize()
    ref = out.clone()
    worst = 0.0
    for _ in range(reps):
        out.zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            side_fn()
            side_fn()
        main_fn()
        torch.cuda.synchronize()
        assert bool(torch.isfinite(out).all())
        worst = max(worst, float((out - ref).abs().max()) / max(float(ref.abs().max()), 1e-30))
    assert worst == 0.0 if exact else worst < 1e-5, worst


@pytest.fixture(scope="module")
def operands():
    from speech_recognition_amd import ops
    g = torch.Generator().manual_seed(0)
    R, D, G = 15968, 2048, 4096                       # the las_large products at half the rows
    o = dict(ops=ops, R=R, D=D, G=G)
    o["x16"] = torch.randn(R, D, generator=g).cuda().bfloat16()
    o["w16"] = torch.randn(G, D, generator=g).cuda().bfloat16()
    o["xt16"] = torch.randn(D, R, generator=g).cuda().bfloat16()
    o["dst16"] = torch.randn(G, R, generator=g).cuda().bfloat16()
    o["a32"] = torch.randn(7968, 512, generator=g).cuda()
    o["b32"] = torch.randn(512, 1024, generator=g).cuda()
    o["c"], o["c2"] = torch.zeros(R, G, device="cuda"), torch.zeros(R, G, device="cuda")
    o["gW"], o["gW2"] = torch.zeros(D, G, device="cuda"), torch.zeros(D, G, device="cuda")
    o["c32"], o["c32b"] = torch.zeros(7968, 1024, device="cuda"), torch.zeros(7968, 1024, device="cuda")
    return o


def test_bf16_eight_phase_product_beside_other_products(operands):
    o, ops = operands, operands["ops"]
    _beside(lambda: ops.gemm_bf16_nt(o["x16"], o["w16"], o["c"]), lambda: ops.gemm_bf16_nt(o["xt16"], o["dst16"], o["gW2"]), o["c"])
    _beside(lambda: ops.gemm_bf16_nt(o["xt16"], o["dst16"], o["gW"]), lambda: ops.gemm_bf16_nt(o["x16"], o["w16"], o["c2"]), o["gW"], exact=False)
    _beside(lambda: ops.gemm_bf16_nt(o["x16"], o["w16"], o["c"]), lambda: [ops.gemm(o["a32"], o["b32"], o["c32b"]) for _ in range(4)], o["c"])


def test_f32_split_product_beside_a_bf16_product(operands):
    o, ops = operands, operands["ops"]
    _beside(lambda: ops.gemm(o["a32"], o["b32"], o["c32"]), lambda: ops.gemm_bf16_nt(o["x16"], o["w16"], o["c2"]), o["c32"])


def test_row_staged_convolutions_beside_products(operands):
    """deepspeech conv3 / conv2 forward and conv2's input gradient at batch 16 - the row-staged kernels the training step takes (version 1)."""
    o, ops = operands, operands["ops"]
    g = torch.Generator().manual_seed(1)
    x2 = torch.randn(16, 355, 25, 32, generator=g).cuda()
    w3 = (torch.randn(21, 11, 32, 96, generator=g) * 0.05).cuda()
    y3 = torch.zeros(16, 168, 15, 96, device="cuda")
    x1 = torch.randn(16, 730, 35, 32, generator=g).cuda()
    w2 = (torch.randn(21, 11, 32, 32, generator=g) * 0.05).cuda()
    y2 = torch.zeros(16, 355, 25, 32, device="cuda")
    dx1 = torch.zeros_like(x1)
    gw3 = torch.zeros_like(w3)
    products = lambda: [ops.gemm(o["a32"], o["b32"], o["c32b"]) for _ in range(6)]
    _beside(lambda: ops.conv2d_fwd(x2, w3, None, (2, 1), y=y3), lambda: ops.gemm_bf16_nt(o["x16"], o["w16"], o["c2"]), y3)
    _beside(lambda: ops.conv2d_fwd(x1, w2, None, (2, 1), y=y2), products, y2)
    # the pair that broke version 2: conv2's input gradient beside filter-gradient kernels
    _beside(lambda: ops.conv2d_bwd_data(y2, w2, dx1, (2, 1)), lambda: ops.conv2d_bwd_filter(x2, y3, gw3, (2, 1)), dx1, reps=8)
