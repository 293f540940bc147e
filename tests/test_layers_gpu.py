"""GPU parity of the single-layer kernels (conv, batch-norm, masks, embedding, dropout, attention step,
cross-entropy, Adam) against the float64 oracle and torch.autograd on it."""
import numpy as np
import pytest
import torch

from oracle import layers as L
from oracle import measure as M
from oracle import rng
from tests.util import assert_close, gpu

pytestmark = pytest.mark.gpu


def _ops():
    from speech_recognition_amd import ops
    return ops


def seed_t(v):
    return torch.tensor([v, 0, 0, 0], dtype=torch.int32, device="cuda")


CONV_CASES = [
    ((2, 31, 20, 3), (3, 3, 3, 32), (2, 2)),      # LAS conv1 shape family
    ((2, 15, 9, 32), (3, 3, 32, 32), (2, 2)),     # LAS conv2 shape family
    ((2, 60, 24, 3), (41, 11, 3, 8), (2, 2)),     # DS2 conv1 family
    ((1, 30, 14, 8), (21, 11, 8, 12), (2, 1)),    # DS2 conv2/3 family
    ((3, 9, 7, 5), (2, 3, 5, 7), (1, 2)),         # odd everything (scalar paths)
    ((1, 40, 12, 4), (5, 3, 4, 96), (2, 1)),      # wide output (128x64 tile path)
    # deepspeech conv2 / conv3 filters on a short clip: few tiles and K = 11 * 11 * O >= 2048 -> the data gradient is split into K
    # partitions that add up atomically in a dX the call zeroes first (round 3); 96 channels: 64x64 forward, 128x128 filter-gradient tiles
    ((2, 60, 25, 32), (21, 11, 32, 32), (2, 1)),
    ((2, 44, 25, 32), (21, 11, 32, 96), (2, 1)),
    # the row-staged kernels (csrc/conv_halo.hip: stride 1 along W, 32-channel multiples) off the deepspeech geometry: 40 output channels (a ragged
    # 32-wide tile), stride 3 along H (three classes of the input gradient, one of them with a single kernel row), 12 taps, 64 input channels
    # (two chunks per kernel row), row groups that do not fill the last workgroup
    ((3, 37, 19, 32), (5, 4, 32, 40), (3, 1)),
    ((1, 23, 40, 64), (7, 12, 64, 32), (1, 1)),
    ((2, 9, 30, 32), (2, 2, 32, 32), (4, 1)),
]


@pytest.fixture(params=["mfma", "split9", "split6"])
def f32_product_mode(request):
    """Every evaluation of an f32 product the library offers (asr_gemm_desc.compute 0 / 2 / 3, asr_set_f32_product_mode): the f32 MFMA
    and the nine / six bf16 pair products of exact three-way operand splits - all held to the SAME tolerances against float64."""
    ops = _ops()
    old = ops.set_f32_gemm_mode(request.param)
    yield request.param
    ops.set_f32_gemm_mode(old)


@pytest.fixture(params=["general", "row_staged"])
def conv_route(request):
    """The general implicit-GEMM kernels (csrc/conv.hip) and, where the geometry allows them, the row-staged ones (csrc/conv_halo.hip) with the
    occupancy gate lifted - by default a test-sized problem never fills the 256 workgroups that gate asks for."""
    ops = _ops()
    old = ops.lib().asr_conv2d_halo_force(1 if request.param == "row_staged" else 0)
    yield request.param
    ops.lib().asr_conv2d_halo_force(old)


@pytest.mark.parametrize("xs,ws,st", CONV_CASES)
def test_conv2d_forward_and_gradients(xs, ws, st, f32_product_mode, conv_route):
    ops = _ops()
    if conv_route == "row_staged":
        d = ops.conv_desc(xs, ws, st)
        import ctypes
        routed = [ops.lib().asr_conv2d_halo_workspace(ctypes.byref(d), k) > 0 for k in (0, 1)]
        if f32_product_mode == "mfma" or not any(routed):
            pytest.skip("this case takes the general kernels on both routes")
    g = torch.Generator().manual_seed(sum(xs) + sum(ws))
    x = torch.randn(xs, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(ws, generator=g, dtype=torch.float64, requires_grad=True)
    b = torch.randn(ws[3], generator=g, dtype=torch.float64, requires_grad=True)
    y = L.conv2d_nhwc(x, w, b, st)
    R = torch.randn(y.shape, generator=g, dtype=torch.float64)
    (y * R).sum().backward()
    xg, wg, bg, dy = gpu(x), gpu(w), gpu(b), gpu(R)
    # f32 accumulation over K terms: the rounding error grows like sqrt(K) (the deepspeech filters sum 7392 products per output)
    ks = max(1.0, (ws[0] * ws[1] * ws[2] / 1000.0) ** 0.5)
    yg = ops.conv2d_fwd(xg, wg, bg, st)
    assert tuple(yg.shape) == tuple(y.shape)
    assert_close(yg, y, 3e-6 * ks, "conv fwd")
    dw = torch.zeros(ws, device="cuda")
    ops.conv2d_bwd_filter(xg, dy, dw, st)
    assert_close(dw, w.grad, 5e-6 * ks, "conv dW")
    dx = torch.full(xs, 3.0, device="cuda")
    ops.conv2d_bwd_data(dy, wg, dx, st)
    assert_close(dx, x.grad, 5e-6 * ks, "conv dX")
    db = torch.zeros(ws[3], device="cuda")
    ops.colsum(dy.view(-1, ws[3]), db)
    assert_close(db, b.grad, 5e-6, "conv db (colsum)")


def test_conv2d_fused_dropout_matches_oracle_mask():
    ops = _ops()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 21, 12, 3, generator=g, dtype=torch.float64)
    w = torch.randn(3, 3, 3, 32, generator=g, dtype=torch.float64)
    b = torch.randn(32, generator=g, dtype=torch.float64)
    y = L.conv2d_nhwc(x, w, b, 2)
    y = y * L.dropout_mult(99, 1, y.shape, 0.15)
    yg = ops.conv2d_fwd(gpu(x), gpu(w), gpu(b), 2, seed=seed_t(99), drop_stream=1, drop_rate=0.15)
    assert_close(yg, y, 3e-6, "conv+dropout")
    # gradient side: flat in-place dropout with the same stream reproduces the mask
    r = torch.ones(y.shape, device="cuda")
    ops.dropout_flat(r, seed_t(99), 1, 0.15)
    assert_close(r, L.dropout_mult(99, 1, y.shape, 0.15), 1e-7, "flat dropout mask")
    tab = torch.empty(4, 37, device="cuda")
    ops.dropout_table(tab, seed_t(5), 12, 0.3)
    assert_close(tab, L.dropout_mult(5, 12, (4, 37), 0.3), 1e-7, "dropout table")


@pytest.mark.parametrize("M_,C_,relu", [(300, 70, True), (97, 512, False), (1, 5, True)])
def test_batch_norm_training_forward_backward(M_, C_, relu):
    ops = _ops()
    g = torch.Generator().manual_seed(M_ + C_)
    x = (torch.randn(M_, C_, generator=g, dtype=torch.float64) * 2 + 0.7).requires_grad_(True)
    gamma = (torch.rand(C_, generator=g, dtype=torch.float64) + 0.5).requires_grad_(True)
    beta = torch.randn(C_, generator=g, dtype=torch.float64).requires_grad_(True)
    mm = torch.randn(C_, generator=g, dtype=torch.float64)
    mv = torch.rand(C_, generator=g, dtype=torch.float64) + 0.5
    y, nmm, nmv = L.batch_norm(x, gamma, beta, mm, mv, True)
    if relu:
        y = torch.relu(y)
    R = torch.randn(y.shape, generator=g, dtype=torch.float64)
    (y * R).sum().backward()
    xg, gg, bg, mmg, mvg = gpu(x), gpu(gamma), gpu(beta), gpu(mm), gpu(mv)
    yg, mean, rstd = torch.empty(M_, C_, device="cuda"), torch.empty(C_, device="cuda"), torch.empty(C_, device="cuda")
    ws = torch.empty(2 * C_, dtype=torch.float64, device="cuda")
    ops.bn_fwd(xg, gg, bg, yg, mean, rstd, mmg, mvg, ws, relu=relu, training=True)
    tol = 2e-5 if M_ > 1 else 2e-4
    assert_close(yg, y, tol, "bn y")
    assert_close(mmg, nmm, 1e-6, "moving mean")
    assert_close(mvg, nmv, 1e-6, "moving var")
    dx, dga, dbe = torch.empty(M_, C_, device="cuda"), torch.zeros(C_, device="cuda"), torch.zeros(C_, device="cuda")
    ops.bn_bwd(xg, yg, gpu(R), mean, rstd, gg, dx, dga, dbe, ws, relu=relu)
    if M_ > 1:
        assert_close(dx, x.grad, 5e-5, "bn dx")
        assert_close(dga, gamma.grad, 5e-5, "bn dgamma")
    assert_close(dbe, beta.grad, 5e-5, "bn dbeta")
    # inference mode uses the moving statistics
    yi, _, _ = L.batch_norm(x.detach(), gamma.detach(), beta.detach(), nmm, nmv, False)
    yig = torch.empty(M_, C_, device="cuda")
    ops.bn_fwd(xg, gg, bg, yig, None, None, mmg, mvg, None, relu=False, training=False)
    assert_close(yig, yi, 2e-5, "bn inference")


def test_frame_mask_matches_reference_rule():
    ops = _ops()
    from oracle import las as OLAS
    g = torch.Generator().manual_seed(3)
    x = torch.randn(5, 43, 7, 3, generator=g)
    x[0, 10:] = 0.0
    x[1, ::2] = 0.0
    x[2] = 0.0
    x[3, 5:9] = 0.0
    x[4, 41:] = 0.0
    ref = OLAS.audio_mask(x.double())
    Tout = ref.shape[1]
    out = ops.frame_mask(x.cuda(), 4, Tout)
    assert torch.equal(out.cpu().bool(), ref)
    out1 = ops.frame_mask(x.cuda(), 1, 43)
    assert torch.equal(out1.cpu().bool(), (x.reshape(5, 43, -1) != 0).any(2))


def test_embedding_gather_scatter_with_two_dropout_sites():
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    V, Hd, B, U, Din, rate = 50, 24, 3, 5, 24 + 16, 0.2
    E = torch.randn(V, Hd, generator=g, dtype=torch.float64)
    tok = torch.randint(0, V, (B, U), generator=g, dtype=torch.int32)
    seed = 77
    d1 = ops.rowdrop(1000 + 0, 32, U, Hd, 0, rate)       # embedding dropout: stream 1000+32*i, idx b*Hd+k
    d2 = ops.rowdrop(1000 + 2, 32, U, Din, 0, rate)      # LSTM-0 input dropout: stream 1002+32*i, idx b*Din+k
    ref = torch.empty(B, U, Hd, dtype=torch.float64)
    mult = torch.empty(B, U, Hd, dtype=torch.float64)
    for i in range(U):
        m1 = L.dropout_mult(seed, 1000 + 32 * i, (B, Hd), rate)
        m2 = L.dropout_mult(seed, 1002 + 32 * i, (B, Din), rate)[:, :Hd]
        mult[:, i] = m1 * m2
        ref[:, i] = E[tok[:, i].long()] * mult[:, i]
    out = torch.empty(B, U, Hd, device="cuda")
    ops.embedding_fwd(gpu(E), tok.cuda(), out, seed_t(seed), d1, d2)
    assert_close(out, ref, 1e-6, "embedding fwd")
    dx = torch.randn(B, U, Hd, generator=g, dtype=torch.float64)
    dE_ref = torch.zeros(V, Hd, dtype=torch.float64)
    dE_ref.index_add_(0, tok.reshape(-1).long(), (dx * mult).reshape(-1, Hd))
    dE = torch.zeros(V, Hd, device="cuda")
    ops.embedding_bwd(dE, tok.cuda(), gpu(dx), seed_t(seed), d1, d2)
    assert_close(dE, dE_ref, 1e-6, "embedding bwd")
    # decoder output dropout site
    y = torch.randn(B * U, Hd, generator=g, dtype=torch.float64)
    yd = torch.empty(B * U, Hd, device="cuda")
    ops.dropout_rows(gpu(y), yd, seed_t(seed), ops.rowdrop(1001, 32, U, Hd, 0, rate))
    refd = y.reshape(B, U, Hd).clone()
    for i in range(U):
        refd[:, i] *= L.dropout_mult(seed, 1001 + 32 * i, (B, Hd), rate)
    assert_close(yd, refd.reshape(B * U, Hd), 1e-6, "row dropout")


@pytest.mark.parametrize("B,T,Hd,D", [(5, 13, 128, 128), (43, 33, 256, 512), (3, 111, 16, 32), (1, 1, 1, 1), (4, 249, 256, 512)])
def test_attention_step_forward_backward(B, T, Hd, D):
    """Shapes from reference tests/models/test_las.py:7-18 plus the las_small step."""
    ops = _ops()
    g = torch.Generator().manual_seed(B * T + Hd)
    h = torch.randn(B, Hd, generator=g, dtype=torch.float64, requires_grad=True)
    enc = torch.randn(B, T, D, generator=g, dtype=torch.float64, requires_grad=True)
    mask = torch.randn(B, T, generator=g) > -0.5
    mask[:, 0] = True
    s = 1.0 / np.sqrt(Hd)
    Wq = torch.randn(Hd, Hd, generator=g, dtype=torch.float64) * s
    bq = torch.randn(Hd, generator=g, dtype=torch.float64) * 0.1
    Wk = torch.randn(D, Hd, generator=g, dtype=torch.float64) / np.sqrt(D)
    bk = torch.randn(Hd, generator=g, dtype=torch.float64) * 0.1
    ctx, p = L.attention(h, enc, enc, mask, Wq, bq, Wk, bk)
    R = torch.randn(ctx.shape, generator=g, dtype=torch.float64)
    gh, = torch.autograd.grad((ctx * R).sum(), h, retain_graph=True)
    K = (enc @ Wk + bk).detach()
    Kq, s0 = K @ Wq.T, K @ bq
    e, pg, ctxg = torch.empty(B, T, device="cuda"), torch.empty(B, T, device="cuda"), torch.empty(B, D, device="cuda")
    encg, Kqg = gpu(enc), gpu(Kq)
    ops.attn_step_fwd(gpu(h), Kqg, gpu(s0), mask.to(torch.uint8).cuda(), encg, e, pg, ctxg)
    assert_close(pg, p, 2e-5, "attention probs")
    assert_close(ctxg, ctx, 2e-5, "context")
    dp, ds, dh = torch.empty(B, T, device="cuda"), torch.empty(B, T, device="cuda"), torch.full((B, Hd), 1.0, device="cuda")
    ops.attn_step_bwd(gpu(R), pg, Kqg, encg, dp, ds, dh, accumulate=True)
    assert_close(dh - 1.0, gh, 1e-4, "dh through the scores")


@pytest.mark.parametrize("B,T,Hd,D", [(5, 13, 128, 128), (7, 61, 1024, 2048), (3, 111, 16, 32), (64, 37, 1024, 2048), (48, 29, 512, 1024)])
def test_attention_step_with_bf16_images(B, T, Hd, D):
    """--mixed-precision: Kq and enc are streamed from bf16 images; products and sums in f32.  The results equal the
    float64 attention on the ROUNDED Kq / enc to f32 accuracy."""
    ops = _ops()
    g = torch.Generator().manual_seed(B * T + Hd + 7)
    h = torch.randn(B, Hd, generator=g) * 0.3
    Kq = torch.randn(B, T, Hd, generator=g) * (1.0 / np.sqrt(Hd))
    enc = torch.randn(B, T, D, generator=g)
    s0 = torch.randn(B, T, generator=g) * 0.1
    mask = torch.randn(B, T, generator=g) > -0.5
    mask[:, 0] = True
    R = torch.randn(B, D, generator=g)
    Kq16, enc16 = Kq.bfloat16(), enc.bfloat16()
    Kr, er = Kq16.double(), enc16.double()
    e_ref = torch.einsum("bh,bth->bt", h.double(), Kr) + s0.double() - 1e9 * (1.0 - mask.double())
    p_ref = torch.softmax(e_ref, dim=1)
    ctx_ref = torch.einsum("bt,btd->bd", p_ref, er)
    dp_ref = torch.einsum("bd,btd->bt", R.double(), er)
    ds_ref = p_ref * (dp_ref - (p_ref * dp_ref).sum(1, keepdim=True))
    dh_ref = torch.einsum("bt,bth->bh", ds_ref, Kr)
    e, pg, ctxg = torch.empty(B, T, device="cuda"), torch.empty(B, T, device="cuda"), torch.empty(B, D, device="cuda")
    images = (Kq16.cuda().contiguous(), enc16.cuda().contiguous())
    ops.attn_step_fwd(h.cuda(), Kq.cuda(), s0.cuda(), mask.to(torch.uint8).cuda(), enc.cuda(), e, pg, ctxg, images=images)
    assert_close(pg, p_ref, 2e-5, "attention probs (bf16 streams)")
    assert_close(ctxg, ctx_ref, 2e-5, "context (bf16 streams)")
    dp, ds, dh = torch.empty(B, T, device="cuda"), torch.empty(B, T, device="cuda"), torch.zeros(B, Hd, device="cuda")
    ops.attn_step_bwd(R.cuda(), pg, Kq.cuda(), enc.cuda(), dp, ds, dh, accumulate=False, images=images)
    assert_close(ds, ds_ref, 1e-4, "score gradients (bf16 streams)")
    assert_close(dh, dh_ref, 1e-4, "dh (bf16 streams)")


@pytest.mark.parametrize("B,T,Hd,D", [(5, 13, 128, 128), (43, 33, 256, 512), (3, 111, 16, 32), (2, 2, 4, 4), (4, 249, 256, 512), (32, 499, 64, 128)])
def test_fused_attention_step_matches_oracle_and_two_kernel_path(B, T, Hd, D):
    """One-launch attention steps (chunked softmax + last-arriver combine): same oracle, same tolerances as the
    two-kernel path, several calls in a row on the same ticket words (as a captured decoder loop does), fully
    masked chunks included."""
    ops = _ops()
    assert ops.attn_fused_supported(T, Hd, D)
    g = torch.Generator().manual_seed(B * T + Hd + 1)
    enc = torch.randn(B, T, D, generator=g, dtype=torch.float64)
    mask = torch.randn(B, T, generator=g) > -0.5
    mask[:, 0] = True
    if T > 40:
        mask[0, 35:] = False                                     # whole chunks masked (a padded clip)
    s = 1.0 / np.sqrt(Hd)
    Wq = torch.randn(Hd, Hd, generator=g, dtype=torch.float64) * s
    bq = torch.randn(Hd, generator=g, dtype=torch.float64) * 0.1
    Wk = torch.randn(D, Hd, generator=g, dtype=torch.float64) / np.sqrt(D)
    bk = torch.randn(Hd, generator=g, dtype=torch.float64) * 0.1
    K = enc @ Wk + bk
    Kq, s0 = K @ Wq.T, K @ bq
    encg, Kqg, s0g, mg = gpu(enc), gpu(Kq), gpu(s0), mask.to(torch.uint8).cuda()
    fws = ops.attn_fused_ws(B, Hd, D)
    for rep in range(3):
        h = torch.randn(B, Hd, generator=g, dtype=torch.float64, requires_grad=True)
        ctx, p = L.attention(h, enc, enc, mask, Wq, bq, Wk, bk)
        R = torch.randn(ctx.shape, generator=g, dtype=torch.float64)
        gh, = torch.autograd.grad((ctx * R).sum(), h)
        pg, ctxg = torch.full((B, T), float("nan"), device="cuda"), torch.full((B, D), float("nan"), device="cuda")
        ops.attn_fused_fwd(gpu(h), Kqg, s0g, mg, encg, fws, pg, ctxg)
        assert_close(pg, p, 2e-5, "attention probs (fused)")
        assert_close(ctxg, ctx, 2e-5, "context (fused)")
        e2, p2, c2 = torch.empty(B, T, device="cuda"), torch.empty(B, T, device="cuda"), torch.empty(B, D, device="cuda")
        ops.attn_step_fwd(gpu(h), Kqg, s0g, mg, encg, e2, p2, c2)
        assert_close(ctxg, c2, 5e-6, "context fused vs two-kernel")
        ds, dh = torch.full((B, T), float("nan"), device="cuda"), torch.full((B, Hd), 1.0, device="cuda")
        ops.attn_fused_bwd(gpu(R), pg, Kqg, encg, fws, ds, dh, accumulate=True)
        assert_close(dh - 1.0, gh, 1e-4, "dh through the scores (fused)")
        dp2, ds2, dh2 = torch.empty(B, T, device="cuda"), torch.empty(B, T, device="cuda"), torch.zeros(B, Hd, device="cuda")
        ops.attn_step_bwd(gpu(R), pg, Kqg, encg, dp2, ds2, dh2, accumulate=False)
        assert_close(ds, ds2, 1e-4, "ds fused vs two-kernel")      # p (dp - <p, dp>) cancels: absolute error of a few f32 ulps of dp
    assert int(fws[1].min()) == int(fws[1].max()) == 6 * 8       # 3 forward + 3 backward calls, 8 chunks each


@pytest.mark.parametrize("R_,V", [(10, 3000), (64, 16000), (3, 120), (2, 40000), (7, 3001), (5, 97), (33, 4100)])   # (vectorised rows, odd V: the scalar kernel, V beyond the LDS)
def test_softmax_cross_entropy_loss_accuracy_gradient(R_, V):
    ops = _ops()
    g = torch.Generator().manual_seed(V)
    logits = (torch.randn(R_, V, generator=g, dtype=torch.float64) * 3).requires_grad_(True)
    y = torch.randint(1, V, (R_,), generator=g, dtype=torch.int32)
    y[R_ // 2] = 0
    loss = M.sparse_categorical_crossentropy(y, logits, 0)
    loss.backward()
    correct, count = M.sparse_categorical_accuracy(y, logits.detach(), 0)
    lg = gpu(logits)
    stats = torch.zeros(3, device="cuda")
    ops.softmax_xent(lg, y.cuda(), stats)
    st = stats.cpu().numpy()
    assert abs(st[0] - float(loss)) <= 1e-5 * max(1.0, abs(float(loss)))
    assert st[1] == correct and st[2] == count
    assert_close(lg, logits.grad, 2e-5, "dlogits")
    # all-pad batch: zero loss, zero gradient (Keras _safe_mean)
    lg2 = gpu(logits)
    stats.zero_()
    ops.softmax_xent(lg2, torch.zeros(R_, dtype=torch.int32, device="cuda"), stats)
    assert float(stats[0]) == 0.0 and float(lg2.abs().max()) == 0.0


def test_argmax_rows_lowest_index_on_ties():
    ops = _ops()
    x = torch.zeros(4, 1000)
    x[0, 7] = 5; x[0, 900] = 5
    x[1, 999] = 1
    x[2] = -1.0
    x[3, 3] = float("inf")
    out = torch.empty(4, dtype=torch.int32, device="cuda")
    ops.argmax_rows(x.cuda(), out)
    assert out.cpu().tolist() == [7, 999, 0, 3]


def test_adam_with_fused_lr_schedule_and_state_advance():
    ops = _ops()
    g = torch.Generator().manual_seed(6)
    n = 1003
    p0 = torch.randn(n, generator=g, dtype=torch.float64)
    params, m, v = {"w": p0.clone()}, {"w": torch.zeros(n, dtype=torch.float64)}, {"w": torch.zeros(n, dtype=torch.float64)}
    sched = M.LRScheduler(20, 2e-3, 1e-5, 0.1)
    pg, mg, vg = gpu(p0), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    state = torch.tensor([0, 42, 0, 0], dtype=torch.int32, device="cuda")
    hs = ops.lr_schedule(20, 2e-3, 1e-5, 0.1)
    for it in range(6):
        grad = torch.randn(n, generator=g, dtype=torch.float64)
        M.adam_step(params, {"w": grad}, m, v, it, sched(it))
        ops.adam_step(pg, gpu(grad), mg, vg, state, hs)
        ops.advance_state(state)
    assert_close(pg, params["w"], 2e-6, "adam params")
    assert_close(mg, m["w"], 2e-6, "adam m")
    st = state.cpu().tolist()
    assert st[0] == 6 and st[1] != 42
    # the seed sequence follows the documented rule
    s = 42
    for _ in range(6):
        s = int(rng._fmix32(np.uint64((s + 0x9E3779B9) & 0xFFFFFFFF)))
    assert (st[1] & 0xFFFFFFFF) == s
