"""GPU parity: asr_gemm_f32 vs float64 matmul (exact-f32 MFMA => ~1e-6 relative)."""
import pytest
import torch

from tests.util import assert_close, gpu

pytestmark = pytest.mark.gpu


def _ops():
    from speech_recognition_amd import ops
    return ops


@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (7968 // 8, 512, 608), (33, 77, 19), (1, 1, 1), (300, 32, 27),
                                   (16, 1000, 256), (250, 130, 5)])
def test_gemm_all_layouts(ta, tb, M, N, K):
    ops = _ops()
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + ta * 2 + tb)
    A = torch.randn((K, M) if ta else (M, K), generator=g, dtype=torch.float64)
    B = torch.randn((N, K) if tb else (K, N), generator=g, dtype=torch.float64)
    bias = torch.randn(N, generator=g, dtype=torch.float64)
    ref = (A.T if ta else A) @ (B.T if tb else B) * 0.5 + bias
    a, b, bi = gpu(A), gpu(B), gpu(bias)
    c = torch.full((M, N), 7.0, device="cuda")
    ops.gemm(a, b, c, trans_a=bool(ta), trans_b=bool(tb), alpha=0.5, bias=bi)
    assert_close(c, ref, 2e-6, f"gemm ta={ta} tb={tb} {M}x{N}x{K}")


def test_gemm_strided_views_accumulate_relu_scales():
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    M, N, K, rpg = 96, 80, 52, 12
    Abig = torch.randn(M, K + 9, generator=g, dtype=torch.float64)
    Bbig = torch.randn(K, N + 4, generator=g, dtype=torch.float64)
    A, B = Abig[:, 3:3 + K], Bbig[:, 1:1 + N]
    asc = torch.rand(M // rpg, K, generator=g, dtype=torch.float64)
    csc = torch.rand(M // rpg, N, generator=g, dtype=torch.float64)
    C0 = torch.randn(M, N, generator=g, dtype=torch.float64)
    Asc = A * asc.repeat_interleave(rpg, 0)
    ref = C0 + torch.relu((Asc @ B) * csc.repeat_interleave(rpg, 0))
    ab, bb = gpu(Abig), gpu(Bbig)
    c = gpu(C0)
    # relu applies to the product before accumulation in the epilogue order: alpha, bias, c_scale, relu, then +=
    ops.gemm(ab[:, 3:3 + K], bb[:, 1:1 + N], c, accumulate=1, relu=True, a_scale=gpu(asc), a_rpg=rpg, c_scale=gpu(csc), c_rpg=rpg)
    assert_close(c, ref, 2e-6, "strided/scale/relu/accumulate")


def test_gemm_batched_and_splitk():
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    Z, M, N, K = 5, 70, 48, 36
    A = torch.randn(Z, K, M, generator=g, dtype=torch.float64)   # trans_a layout
    B = torch.randn(Z, K, N, generator=g, dtype=torch.float64)
    ref_b = torch.einsum("zkm,zkn->zmn", A, B)
    a, b = gpu(A), gpu(B)
    c = torch.empty(Z, M, N, device="cuda")
    ops.gemm(a, b, c, trans_a=True)
    assert_close(c, ref_b, 2e-6, "batched TN")
    c2 = torch.zeros(M, N, device="cuda")
    ops.gemm(a, b, c2, trans_a=True, accumulate=1)
    assert_close(c2, ref_b.sum(0), 4e-6, "split-K over batch (atomic)")


def test_gemm_large_tile_path_matches():
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    M, N, K = 2048, 2048, 300
    A = torch.randn(M, K, generator=g)
    B = torch.randn(K, N, generator=g)
    c = torch.empty(M, N, device="cuda")
    ops.gemm(gpu(A), gpu(B), c)
    assert_close(c, A.double() @ B.double(), 3e-6, "128x128 tile path")


def test_gemm_rejects_bad_shapes():
    ops = _ops()
    a = torch.zeros(4, 5, device="cuda")
    b = torch.zeros(6, 7, device="cuda")
    c = torch.zeros(4, 7, device="cuda")
    with pytest.raises(ValueError):
        ops.gemm(a, b, c)
    with pytest.raises(ValueError):
        ops.gemm(a.cpu(), b, c)


# ---------------------------------------------------------------------------------------------- mixed precision
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(256, 384, 512), (7968 // 8, 512, 608), (33, 77, 19), (300, 32, 27), (16, 1000, 256)])
def test_gemm_bf16_operands_f32_accumulate(ta, tb, M, N, K):
    """compute=1: operands rounded to bf16 (round to nearest even) inside the kernel, products and sums in f32 -
    so the result equals the float64 product of the ROUNDED operands to f32 accumulation accuracy, and differs
    from the unrounded product by the bf16 rounding (2^-9 relative per operand)."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + N * 3 + K * 5 + ta * 2 + tb)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    B = torch.randn((N, K) if tb else (K, N), generator=g)
    bias = torch.randn(N, generator=g)
    Ar, Br = A.bfloat16().double(), B.bfloat16().double()         # torch rounds to nearest even as v_cvt_pk_bf16_f32 does
    ref_r = (Ar.T if ta else Ar) @ (Br.T if tb else Br) + bias.double()
    ref_x = (A.double().T if ta else A.double()) @ (B.double().T if tb else B.double()) + bias.double()
    c = torch.full((M, N), 7.0, device="cuda")
    ops.gemm(A.cuda(), B.cuda(), c, trans_a=bool(ta), trans_b=bool(tb), bias=bias.cuda(), compute=1)
    assert_close(c, ref_r, 3e-6, f"bf16 gemm vs rounded operands ta={ta} tb={tb} {M}x{N}x{K}")
    scale = float(ref_x.abs().max())
    assert float((c.cpu().double() - ref_x).abs().max()) < 2e-2 * scale
    c32 = torch.empty(M, N, device="cuda")
    ops.gemm(A.cuda(), B.cuda(), c32, trans_a=bool(ta), trans_b=bool(tb), bias=bias.cuda(), compute=0)
    assert_close(c32, ref_x, 2e-6, "compute=0 stays exact f32")


def test_gemm_bf16_scaled_operand_splitk_and_switch():
    ops = _ops()
    g = torch.Generator().manual_seed(21)
    M, N, K, rpg = 96, 80, 1000, 12
    A, B = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
    asc = torch.rand(K // 8, M, generator=g)                      # a_scale on the stored (K x M) operand, 8 rows per group
    As = (A * asc.repeat_interleave(8, 0)).bfloat16().double()    # the scale is applied in f32 before the rounding
    ref = As.T @ B.bfloat16().double()
    c = torch.zeros(M, N, device="cuda")
    try:
        ops.set_mixed_precision(True)
        assert ops.mixed_precision()
        ops.gemm(A.cuda(), B.cuda(), c, trans_a=True, accumulate=1, split_k=5, a_scale=asc.cuda(), a_rpg=8)
    finally:
        ops.set_mixed_precision(False)
    assert_close(c, ref, 5e-6, "bf16 split-K with a_scale")
    with pytest.raises(ValueError):
        ops.gemm(A.cuda(), B.cuda(), c, trans_a=True, compute=7)


# ---------------------------------------------------------------------------------------------- f32 products on the bf16 matrix pipe
@pytest.mark.parametrize("compute,tol", [(2, 2e-6), (3, 2e-6)])
@pytest.mark.parametrize("ta,tb", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (996, 512, 608), (33, 77, 19), (1, 1, 1), (300, 32, 27), (16, 1000, 256), (250, 130, 5),
                                   (2048, 2048, 300), (32, 512, 100), (700, 24, 90)])
def test_gemm_split_bf16_products_all_layouts(compute, tol, ta, tb, M, N, K):
    """asr_gemm_desc.compute = 2 / 3: f32 operands split exactly into three bf16 parts, products as nine (six) bf16 pair products on
    the bf16 MFMA, f32 accumulation - against the float64 product at the SAME tolerance as the f32 MFMA (2e-6 of the largest entry),
    every layout, ragged tiles, every tile shape (64 / 128 / 256 x 32), bias + alpha epilogue."""
    ops = _ops()
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + ta * 2 + tb)
    A = torch.randn((K, M) if ta else (M, K), generator=g, dtype=torch.float64)
    B = torch.randn((N, K) if tb else (K, N), generator=g, dtype=torch.float64)
    bias = torch.randn(N, generator=g, dtype=torch.float64)
    a, b, bi = gpu(A), gpu(B), gpu(bias)
    ref = (a.double().cpu().T if ta else a.double().cpu()) @ (b.double().cpu().T if tb else b.double().cpu()) * 0.5 + bi.double().cpu()
    c = torch.full((M, N), 7.0, device="cuda")
    ops.gemm(a, b, c, trans_a=bool(ta), trans_b=bool(tb), alpha=0.5, bias=bi, compute=compute)
    assert_close(c, ref, tol, f"split gemm compute={compute} ta={ta} tb={tb} {M}x{N}x{K}")


def test_gemm_split_is_at_least_as_accurate_as_the_f32_mfma():
    """The nine-pair evaluation keeps every product to 2^-32 and differs from the f32 MFMA only in accumulation order: on a long
    contraction with a wide dynamic range (operands spanning 2^+-12) its error against float64 must not exceed the f32 MFMA's by
    more than 1.5x; the six-pair one stays within 3x.  Split-K, scaled A rows and strided views run the same kernels."""
    ops = _ops()
    g = torch.Generator().manual_seed(77)
    M, N, K = 512, 384, 4096
    A = torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-12, 13, (M, K), generator=g).float())
    B = torch.randn(K, N, generator=g) * torch.exp2(torch.randint(-12, 13, (K, N), generator=g).float())
    a, b = gpu(A), gpu(B)
    ref = A.double() @ B.double()
    errs = {}
    for compute in (0, 2, 3):
        c = torch.empty(M, N, device="cuda")
        ops.gemm(a, b, c, compute=compute)
        errs[compute] = float((c.double().cpu() - ref).abs().max()) / float(ref.abs().max())
    print("max-normalised error vs float64: f32 MFMA %.3e, nine pairs %.3e, six pairs %.3e" % (errs[0], errs[2], errs[3]))
    assert errs[2] <= 1.25 * errs[0] + 1e-9 and errs[3] <= 1.25 * errs[0] + 1e-9, errs       # (measured: 1.42e-6 / 8.6e-7 / 8.6e-7)
    # ... and on the training step's own shapes with N(0, 1) operands: root-mean-square error of each evaluation against float64
    for (M2, N2, K2) in ((1024, 512, 512), (256, 384, 7968)):
        A2, B2 = torch.randn(M2, K2, generator=g), torch.randn(K2, N2, generator=g)
        ref3 = A2.double() @ B2.double()
        rms = {}
        for compute in (0, 2, 3):
            c3 = torch.empty(M2, N2, device="cuda")
            ops.gemm(gpu(A2), gpu(B2), c3, compute=compute)
            rms[compute] = float(((c3.double().cpu() - ref3) ** 2).mean().sqrt() / (ref3 ** 2).mean().sqrt())
        print(f"K={K2}: relative rms error vs float64: f32 MFMA {rms[0]:.3e}, nine pairs {rms[2]:.3e}, six pairs {rms[3]:.3e}")
        assert rms[2] <= 1.25 * rms[0] and rms[3] <= 1.25 * rms[0], (K2, rms)
    # row-group scale + split-K + accumulate through the split kernels
    rpg = 64
    asc = torch.rand(M // rpg, K, generator=g)
    C0 = torch.randn(M, N, generator=g)
    c = gpu(C0)
    ops.gemm(a, b, c, accumulate=1, split_k=4, a_scale=gpu(asc), a_rpg=rpg, compute=2)
    ref2 = C0.double() + (A.double() * asc.double().repeat_interleave(rpg, 0)) @ B.double()
    assert_close(c, ref2, 2e-6, "split gemm with row-group scale, split-K, accumulate")
