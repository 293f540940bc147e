"""The bf16-operand product (csrc/gemm16.hip) and the image passes feeding it, against the float64 product of the bf16-rounded
operands: the only difference allowed is f32 accumulation order.  Shapes cover ragged M/N tiles, K that is no multiple of 8
(zero-padded image columns), every operand layout ops.gemm routes (NN, NT, TN, batch-flattened TN), and the epilogue options."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bf(x):
    return x.to(torch.bfloat16).to(torch.float64)


def _check(got, want, a, b):
    # f32 accumulation of K products: error << sqrt(K) * eps_f32 * |a|.|b|; 1e-4 of the output scale is loose for that and tight for
    # a wrong element (a missing k chunk, a misplaced tile)
    scale = float(want.abs().max())
    err = float((got.double() - want).abs().max())
    assert err <= 1e-4 * scale, (err, scale)


@pytest.fixture(params=range(16))
def tile_config(request):
    """Every tile configuration of the kernel (register / direct-to-LDS staging, one / two LDS buffers, 128 / 256-row tiles)."""
    from speech_recognition_amd import ops
    old = ops.lib().asr_gemm_bf16_config(request.param)
    yield request.param
    ops.lib().asr_gemm_bf16_config(old)


@pytest.mark.parametrize("M,N,K", [(256, 256, 512), (300, 520, 1001), (1024, 384, 2048), (130, 4100, 776), (1000, 700, 1088)])
@pytest.mark.parametrize("layout", ["nn", "nt", "tn"])
def test_routed_product_matches_f64_of_rounded_operands(M, N, K, layout, tile_config):
    from speech_recognition_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).cuda()
    Bm = torch.randn(K, N, generator=g).cuda()
    ta, tb = layout[0] == "t", layout[1] == "t"
    a = A.t().contiguous() if ta else A
    b = Bm.t().contiguous() if tb else Bm
    bias = torch.randn(N, generator=g).cuda()
    c = torch.empty(M, N, device="cuda")
    old = ops._bf16_images["min_dim"]
    ops._bf16_images["min_dim"] = 128
    try:
        before = len(ops._bf16_images["scratch"])
        ops.gemm(a, b, c, trans_a=ta, trans_b=tb, bias=bias, compute=1)
        assert len(ops._bf16_images["scratch"]) > before or before > 0, "the bf16-image path must be the one under test"
        want = _bf(A) @ _bf(Bm) + bias.double()
        _check(c, want, A, Bm)
        # accumulate + alpha + relu
        c2 = torch.ones(M, N, device="cuda")
        ops.gemm(a, b, c2, trans_a=ta, trans_b=tb, alpha=0.5, accumulate=1, compute=1)
        _check(c2, 0.5 * (_bf(A) @ _bf(Bm)) + 1.0, A, Bm)
        ops.gemm(a, b, c, trans_a=ta, trans_b=tb, relu=True, compute=1)
        _check(c, torch.relu(_bf(A) @ _bf(Bm)), A, Bm)
    finally:
        ops._bf16_images["min_dim"] = old


def test_row_group_scale_is_folded_into_the_image():
    """Keras input dropout: one multiplier row per sequence (a_rpg = T rows), applied to the f32 values BEFORE the bf16 rounding,
    as the f32-operand kernel does."""
    from speech_recognition_amd import ops
    g = torch.Generator().manual_seed(5)
    Bq, T, D, N = 6, 50, 256, 384
    x = torch.randn(Bq * T, D, generator=g).cuda()
    W = torch.randn(D, N, generator=g).cuda()
    tab = (torch.rand(Bq, D, generator=g) > 0.3).float().cuda() / 0.7
    old = ops._bf16_images["min_dim"]
    ops._bf16_images["min_dim"] = 128
    try:
        c = torch.empty(Bq * T, N, device="cuda")
        ops.gemm(x, W, c, a_scale=tab, a_rpg=T, compute=1)
        xs = (x.view(Bq, T, D) * tab[:, None]).view(Bq * T, D)
        _check(c, _bf(xs) @ _bf(W), xs, W)
        # weight gradient: x^T ds with the same table on the stored rows of x
        ds = torch.randn(Bq * T, N, generator=g).cuda()
        gW = torch.zeros(D, N, device="cuda")
        ops.gemm(x, ds, gW, trans_a=True, accumulate=1, a_scale=tab, a_rpg=T, compute=1)
        _check(gW, _bf(xs).t() @ _bf(ds), xs, ds)
    finally:
        ops._bf16_images["min_dim"] = old


def test_batch_split_flattens_into_one_long_k():
    """dU = sum_b h[b, :T-1]^T ds[b, 1:] (the shifted views of the recurrent-kernel gradient): 3-D operands, 2-D c."""
    from speech_recognition_amd import ops
    g = torch.Generator().manual_seed(7)
    Bq, T, H, N = 5, 41, 256, 512
    hs = torch.randn(Bq, T, H, generator=g).cuda()
    ds = torch.randn(Bq, T, N, generator=g).cuda()
    old = ops._bf16_images["min_dim"]
    ops._bf16_images["min_dim"] = 128
    try:
        gU = torch.zeros(H, N, device="cuda")
        ops.gemm(hs[:, :T - 1], ds[:, 1:], gU, trans_a=True, accumulate=1, compute=1)
        want = torch.einsum("bth,btn->hn", _bf(hs[:, :T - 1]), _bf(ds[:, 1:]))
        _check(gU, want, hs, ds)
        ops.gemm(hs[:, 1:], ds[:, :T - 1, 128:384], gU[:, 128:384], trans_a=True, accumulate=1, compute=1)      # column slices
        want[:, 128:384] += torch.einsum("bth,btn->hn", _bf(hs[:, 1:]), _bf(ds[:, :T - 1, 128:384]))
        _check(gU, want, hs, ds)
    finally:
        ops._bf16_images["min_dim"] = old


def test_small_products_keep_the_f32_operand_kernel():
    from speech_recognition_amd import ops
    a = torch.randn(64, 96).cuda()
    b = torch.randn(96, 80).cuda()
    c = torch.empty(64, 80, device="cuda")
    n = len(ops._bf16_images["scratch"])
    ops.gemm(a, b, c, compute=1)
    assert len(ops._bf16_images["scratch"]) == n
    _check(c, _bf(a) @ _bf(b), a, b)


def test_shifted_transposed_image_lets_the_recurrent_gradient_share_the_ds_image():
    """dU = sum_b sum_t h[b, t - 1]^T ds[b, t]: h is written one column off inside every clip (dst_rows_per_batch / dst_shift), the
    column without a predecessor stays zero, and the product with the FULL transposed ds image equals the shifted-view product."""
    from speech_recognition_amd import ops
    g = torch.Generator().manual_seed(11)
    Bq, T, H, N = 6, 37, 256, 512
    hs = torch.randn(Bq, T, H, generator=g).cuda()
    ds = torch.randn(Bq, T, N, generator=g).cuda()
    K8 = (Bq * T + 7) // 8 * 8
    dsT = torch.zeros(N, K8, device="cuda", dtype=torch.bfloat16)
    ops.f32_to_bf16_image(ds.view(Bq * T, N), dsT, transpose=True)
    for reverse in (False, True):
        hT = torch.zeros(H, K8, device="cuda", dtype=torch.bfloat16)
        if reverse:
            ops.f32_to_bf16_image(hs[:, 1:], hT, transpose=True, dst_rows_per_batch=T, dst_shift=0)
            want = torch.einsum("bth,btn->hn", _bf(hs[:, 1:]), _bf(ds[:, :T - 1]))
        else:
            ops.f32_to_bf16_image(hs[:, :T - 1], hT, transpose=True, dst_rows_per_batch=T, dst_shift=1)
            want = torch.einsum("bth,btn->hn", _bf(hs[:, :T - 1]), _bf(ds[:, 1:]))
        img = hT[:, :Bq * T].float().view(H, Bq, T)
        assert float(img[:, :, T - 1 if reverse else 0].abs().max()) == 0.0
        gU = torch.zeros(H, N, device="cuda")
        ops.gemm_bf16_nt(hT, dsT, gU, accumulate=1)
        _check(gU, want, hs, ds)


def test_scratch_is_not_shared_between_contractions_that_only_agree_on_the_padded_k():
    """the image scratch was keyed by K rounded up to 8 and zeroed once.  K = 1008 leaves values in columns
    [1004, 1008) that a following K = 1004 product (same M, N, same padded size) would read as its zero padding."""
    from speech_recognition_amd import ops
    g = torch.Generator().manual_seed(13)
    M, N = 256, 384
    old = ops._bf16_images["min_dim"]
    ops._bf16_images["min_dim"] = 128
    try:
        for K in (1008, 1004, 1001, 1008):
            A = torch.randn(M, K, generator=g).cuda() + 1.0           # (a non-zero mean makes stale padding products add up)
            Bm = torch.randn(K, N, generator=g).cuda() + 1.0
            for ta, tb in ((False, False), (True, False), (False, True)):
                a = A.t().contiguous() if ta else A
                b = Bm.t().contiguous() if tb else Bm
                c = torch.empty(M, N, device="cuda")
                ops.gemm(a, b, c, trans_a=ta, trans_b=tb, compute=1)
                _check(c, _bf(A) @ _bf(Bm), A, Bm)
    finally:
        ops._bf16_images["min_dim"] = old


def test_birnn_backward_shared_image_path_survives_a_change_of_batch_geometry():
    """the shifted h image's 'no predecessor' column sits at t = 0 (T - 1 reversed) of every clip, so it moves when (B, T)
    changes at constant B * T.  Two BiRNN backward passes through the shared-image path, (B, T) then (2B, T / 2), each against the
    f32-operand-kernel path (bf16 fragments, the same numerics) on the same tensors."""
    from speech_recognition_amd import layers, ops
    from speech_recognition_amd.params import ParamStore
    H, Din = 256, 256
    old = ops._bf16_images["min_dim"]
    ops._bf16_images["min_dim"] = 128
    ops.set_mixed_precision(True)
    try:
        shapes = layers.BiRNN.param_shapes("l/", "lstm", Din, H)
        res = {}
        for images in (True, False):
            ops._bf16_images["on"] = images
            g = torch.Generator().manual_seed(3)
            store = ParamStore(shapes)
            for n in shapes:
                store.p[n].copy_(torch.randn(shapes[n], generator=g) * 0.05)
            store.refresh_bf16()
            rnn = layers.BiRNN(store, "l/", "lstm", Din, H, 0.0, 40)
            rnn.pack()
            for B, T in ((8, 64), (16, 32)):
                x = torch.randn(B, T, Din, generator=g).cuda()
                dy = torch.randn(B, T, 2 * H, generator=g).cuda()
                buf = rnn.alloc(B, T)
                ops.fill(store.grad, 0.0)
                rnn.forward(buf, x, None, None, True, None)
                dcs = [torch.zeros(B, H, device="cuda") for _ in range(2)]
                rnn.backward(buf, dy, [None, None], dcs, None)
                torch.cuda.synchronize()
                res[(images, B, T)] = {k: v.clone() for k, v in store.grads().items()}
        for B, T in ((8, 64), (16, 32)):
            for k, ref in res[(False, B, T)].items():
                got = res[(True, B, T)][k]
                err = float((got - ref).abs().max()) / max(float(ref.abs().max()), 1e-6)
                assert err < 2e-3, (B, T, k, err)
    finally:
        ops._bf16_images["on"] = True
        ops._bf16_images["min_dim"] = old
        ops.set_mixed_precision(False)


def test_time_major_transposed_image_pairs_x_and_shifted_h_with_a_time_major_ds_image():
    """asr_f32_to_bf16_image_tb: dst[c][t * B + b + shift] = bf16(src[b, t, c] * scale[b, c]) - the column order of the transposed ds image the wide
    BPTT sweep writes (round 4).  dW = x^T ds with the dropout table folded in, dU = h_prev^T ds with the shifted h image whose free column
    block takes the initial state, both against float64 sums over the rounded operands; strided batch views."""
    from speech_recognition_amd import ops
    g = torch.Generator().manual_seed(19)
    Bq, T, D, H, N = 6, 21, 96, 64, 128
    K = Bq * T
    K8 = (K + 7) // 8 * 8
    x = torch.randn(Bq, T, D, generator=g).cuda()
    hs_big = torch.randn(Bq, T + 3, H, generator=g).cuda()
    hs = hs_big[:, 2:T + 2]                                           # a strided batch view
    h0 = torch.randn(Bq, H, generator=g).cuda()
    ds = torch.randn(Bq, T, N, generator=g).cuda()
    tab = ((torch.rand(Bq, D, generator=g) > 0.3).float() / 0.7).cuda()
    dsT = torch.zeros(N, K8, device="cuda", dtype=torch.bfloat16)
    dsT[:, :K] = ds.permute(2, 1, 0).reshape(N, K).to(torch.bfloat16)   # [n][t * B + b]
    xT = torch.zeros(D, K8, device="cuda", dtype=torch.bfloat16)
    ops.f32_to_bf16_image_tb(x, xT, scale=tab)
    want = (x * tab[:, None]).to(torch.bfloat16).float().permute(2, 1, 0).reshape(D, K)
    assert torch.equal(xT[:, :K].float(), want) and float(xT[:, K:].float().abs().max() if K8 > K else 0.0) == 0.0
    gW = torch.zeros(D, N, device="cuda")
    ops.gemm_bf16_nt(xT, dsT, gW, accumulate=1)
    _check(gW, torch.einsum("btd,btn->dn", _bf(x * tab[:, None]), _bf(ds)), x, ds)
    for reverse in (False, True):
        hT = torch.zeros(H, K8, device="cuda", dtype=torch.bfloat16)
        if reverse:
            ops.f32_to_bf16_image_tb(hs[:, 1:], hT, dst_shift=0)
            ops.f32_to_bf16_image_tb(h0.unsqueeze(1), hT, dst_shift=(T - 1) * Bq)
            hprev = torch.cat([hs[:, 1:], h0[:, None]], 1)
        else:
            ops.f32_to_bf16_image_tb(hs[:, :T - 1], hT, dst_shift=Bq)
            ops.f32_to_bf16_image_tb(h0.unsqueeze(1), hT, dst_shift=0)
            hprev = torch.cat([h0[:, None], hs[:, :T - 1]], 1)
        gU = torch.zeros(H, N, device="cuda")
        ops.gemm_bf16_nt(hT, dsT, gU, accumulate=1)
        _check(gU, torch.einsum("bth,btn->hn", _bf(hprev), _bf(ds)), hprev, ds)
