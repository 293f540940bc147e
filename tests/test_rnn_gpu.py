"""GPU parity: (Bi)RNN layer kernels (LSTM / GRU / SimpleRNN, arbitrary masks, chained initial
states) vs the oracle's K.rnn restatement; gradients vs torch.autograd on the float64 oracle.
Also the reference's own invariant: appending masked padding must not change the prefix
(reference tests/models/test_las.py:21-44, test_deepspeech2.py:30-56)."""
import pytest
import torch

from oracle import layers as L
from tests.rnn_helpers import NG, HipBiRNN
from tests.util import assert_close

pytestmark = pytest.mark.gpu


def make_params(rt, D, H, g, scale=0.3):
    ng = NG[rt]
    out = []
    for _ in range(2):
        W = torch.randn(D, ng * H, generator=g, dtype=torch.float64) * scale
        U = torch.randn(H, ng * H, generator=g, dtype=torch.float64) * scale
        b = torch.randn((2, ng * H) if rt == "gru" else (ng * H,), generator=g, dtype=torch.float64) * scale
        out.append((W, U, b))
    return out


CASES = [("lstm", 5, 7, 6, 8), ("lstm", 23, 11, 8, 13), ("gru", 18, 9, 5, 20), ("rnn", 3, 5, 4, 6),
         ("gru", 34, 6, 3, 111), ("lstm", 32, 40, 16, 256),
         # wide cells with several batch tiles (the LAS-large regime of the step kernels)
         ("lstm", 40, 5, 8, 512), ("gru", 64, 4, 6, 516)]


@pytest.mark.parametrize("rt,B,T,D,H", CASES)
@pytest.mark.parametrize("masked,with_init", [(False, False), (True, True)])
def test_birnn_forward_backward(rt, B, T, D, H, masked, with_init):
    g = torch.Generator().manual_seed(B * 100 + T * 10 + H)
    fwd, bwd = make_params(rt, D, H, g, 0.3 if H < 100 else 0.08)
    x = torch.randn(B, T, D, generator=g, dtype=torch.float64)
    mask = (torch.randn(B, T, generator=g) > -0.3) if masked else torch.ones(B, T, dtype=torch.bool)
    if masked:
        mask[0] = False           # a fully masked row
        mask[-1, : T // 2] = False
    nst = 2 if rt == "lstm" else 1
    init = [torch.randn(B, H, generator=g, dtype=torch.float64) * 0.5 for _ in range(2 * nst)] if with_init else None

    leaves = [t.clone().requires_grad_(True) for p in (fwd, bwd) for t in p]
    xl = x.clone().requires_grad_(True)
    il = [t.clone().requires_grad_(True) for t in init] if init else None
    out, *states = L.birnn(rt, xl, mask, tuple(leaves[:3]), tuple(leaves[3:]), il)
    R = torch.randn(out.shape, generator=g, dtype=torch.float64)
    S = [torch.randn(s.shape, generator=g, dtype=torch.float64) for s in states]
    loss = (out * R).sum() + sum((s * w).sum() for s, w in zip(states, S))
    loss.backward()

    hip = HipBiRNN(rt, x, mask if masked else None, fwd, bwd, init)
    y, hstates = hip.forward()
    assert_close(y, out, 5e-5, "outputs")
    for i, (a, b_) in enumerate(zip(hstates, states)):
        assert_close(a, b_, 5e-5, f"state {i}")

    grads = hip.backward(R, S)
    dx = grads[0]["dx"] + grads[1]["dx"]
    assert_close(dx, xl.grad, 1e-4, "dx")
    for d in range(2):
        W, U, b = leaves[3 * d: 3 * d + 3]
        assert_close(grads[d]["dW"], W.grad, 1e-4, f"dW[{d}]")
        assert_close(grads[d]["dU"], U.grad, 1e-4, f"dU[{d}]")
        assert_close(grads[d]["db"], b.grad, 1e-4, f"db[{d}]")
        if init:
            assert_close(grads[d]["dh0"], il[d * nst].grad, 1e-4, f"dh0[{d}]")
            if rt == "lstm":
                assert_close(grads[d]["dc0"], il[d * nst + 1].grad, 1e-4, f"dc0[{d}]")


@pytest.mark.parametrize("rt,H,B,T,D,pad", [("rnn", 13, 23, 11, 8, 3), ("lstm", 33, 34, 41, 2, 4), ("gru", 111, 55, 3, 99, 5)])
def test_masked_padding_does_not_change_prefix(rt, H, B, T, D, pad):
    """reference tests/models/test_las.py:21-44 (same parametrisation)."""
    g = torch.Generator().manual_seed(H)
    fwd, bwd = make_params(rt, D, H, g)
    x = torch.randn(B, T, D, generator=g, dtype=torch.float64)
    mask = torch.randn(B, T, generator=g) > 0.1
    y, st = HipBiRNN(rt, x, mask, fwd, bwd).forward()
    xp = torch.cat([x, torch.randn(B, pad, D, generator=g, dtype=torch.float64)], dim=1)
    mp = torch.cat([mask, torch.zeros(B, pad, dtype=torch.bool)], dim=1)
    yp, stp = HipBiRNN(rt, xp, mp, fwd, bwd).forward()
    assert tuple(yp.shape) == (B, T + pad, 2 * H)
    assert torch.equal(y, yp[:, :T])          # the reference asserts exact equality
    assert torch.equal(st[0], stp[0])


def test_invalid_rnn_type_raises_value_error():
    from speech_recognition_amd import ops
    with pytest.raises(ValueError, match="rnn_type: foo is invalid!"):
        ops.rnn_geometry("foo", 8, [8])


PERSIST_CASES = [("lstm", 32, 40, 16, 256), ("gru", 20, 12, 7, 128), ("rnn", 5, 6, 3, 16), ("lstm", 7, 25, 4, 48), ("lstm", 32, 249, 8, 256)]


@pytest.mark.parametrize("rt,B,T,D,H", PERSIST_CASES)
@pytest.mark.parametrize("masked", [False, True])
def test_persistent_forward_matches_oracle_and_step_kernels(rt, B, T, D, H, masked):
    """The one-launch persistent layer kernel (in-kernel hand-offs of h_t) against the oracle and,
    bit for bit, against the per-step kernels (same arithmetic order)."""
    g = torch.Generator().manual_seed(B + T + H)
    fwd, bwd = make_params(rt, D, H, g, 0.3 if H < 100 else 0.08)
    x = torch.randn(B, T, D, generator=g, dtype=torch.float64)
    mask = (torch.randn(B, T, generator=g) > -0.3) if masked else torch.ones(B, T, dtype=torch.bool)
    nst = 2 if rt == "lstm" else 1
    init = [torch.randn(B, H, generator=g, dtype=torch.float64) * 0.5 for _ in range(2 * nst)] if masked else None
    out, *states = L.birnn(rt, x, mask, fwd, bwd, init)
    hip = HipBiRNN(rt, x, mask if masked else None, fwd, bwd, init)
    y, hstates = hip.forward(persistent=True)
    y, hstates = y.clone(), [s.clone() for s in hstates]
    assert_close(y, out, 1e-4, "outputs")
    for i, (a, b_) in enumerate(zip(hstates, states)):
        assert_close(a, b_, 1e-4, f"state {i}")
    hip2 = HipBiRNN(rt, x, mask if masked else None, fwd, bwd, init)
    y2, hstates2 = hip2.forward(persistent=False)
    assert torch.equal(y, y2)
    for a, b_ in zip(hstates, hstates2):
        assert torch.equal(a, b_)
    for da, db in zip(hip.dirs, hip2.dirs):
        assert torch.equal(da["saved"], db["saved"]) and torch.equal(da["hseq"], db["hseq"])


@pytest.mark.parametrize("rt,B,T,D,H", PERSIST_CASES)
@pytest.mark.parametrize("masked", [False, True])
def test_persistent_backward_matches_step_kernels(rt, B, T, D, H, masked):
    """Same algorithm and summation order as the per-step kernels; results agree to fp32 rounding (the
    compiler contracts the gate-gradient expressions differently in the two kernels)."""
    g = torch.Generator().manual_seed(B + 3 * T + H)
    fwd, bwd = make_params(rt, D, H, g, 0.3 if H < 100 else 0.08)
    x = torch.randn(B, T, D, generator=g, dtype=torch.float64)
    mask = (torch.randn(B, T, generator=g) > -0.3) if masked else None
    nst = 2 if rt == "lstm" else 1
    init = [torch.randn(B, H, generator=g, dtype=torch.float64) * 0.5 for _ in range(2 * nst)] if masked else None
    R = torch.randn(B, T, 2 * H, generator=g, dtype=torch.float64)
    S = [torch.randn(B, H, generator=g, dtype=torch.float64) for _ in range(2 * nst)]
    res = []
    for persistent in (False, True):
        hip = HipBiRNN(rt, x, mask, fwd, bwd, init)
        hip.forward(persistent=persistent)
        grads = hip.backward(R, S, persistent=persistent)
        res.append((grads, [(dd["ds"] if persistent else dd["saved"]).clone() for dd in hip.dirs]))
    for d in range(2):
        assert_close(res[1][1][d], res[0][1][d], 2e-5, "ds")
        for k in res[0][0][d]:
            assert_close(res[1][0][d][k], res[0][0][d][k], 2e-5, k)


@pytest.mark.parametrize("rt,B,T,D,H", PERSIST_CASES + [("gru", 18, 30, 5, 128), ("lstm", 33, 21, 4, 64)])
@pytest.mark.parametrize("masked", [False, True])
def test_persistent_sweeps_match_oracle_autograd(rt, B, T, D, H, masked):
    """The one-launch forward AND backward sweeps against the float64 oracle directly (torch.autograd through oracle.layers.birnn):
    outputs, final states, dx, dW, dU, db, dh0, dc0 - arbitrary masks with a fully masked row, chained initial states, gradients
    flowing into both the output sequence and the final states.  Tolerances: f32 recurrences over up to 249 steps against
    float64 - 1e-4 of the largest entry forward, 2e-4 backward (the same bounds the per-step kernels are held to, x2 for the
    longer sequences)."""
    g = torch.Generator().manual_seed(7 * B + T + H)
    fwd, bwd = make_params(rt, D, H, g, 0.3 if H < 100 else 0.08)
    x = torch.randn(B, T, D, generator=g, dtype=torch.float64)
    mask = (torch.randn(B, T, generator=g) > -0.3) if masked else torch.ones(B, T, dtype=torch.bool)
    if masked:
        mask[0] = False
        mask[-1, : T // 2] = False
    nst = 2 if rt == "lstm" else 1
    init = [torch.randn(B, H, generator=g, dtype=torch.float64) * 0.5 for _ in range(2 * nst)] if masked else None
    leaves = [t.clone().requires_grad_(True) for p in (fwd, bwd) for t in p]
    xl = x.clone().requires_grad_(True)
    il = [t.clone().requires_grad_(True) for t in init] if init else None
    out, *states = L.birnn(rt, xl, mask, tuple(leaves[:3]), tuple(leaves[3:]), il)
    R = torch.randn(out.shape, generator=g, dtype=torch.float64)
    S = [torch.randn(s.shape, generator=g, dtype=torch.float64) for s in states]
    ((out * R).sum() + sum((s * w).sum() for s, w in zip(states, S))).backward()

    hip = HipBiRNN(rt, x, mask if masked else None, fwd, bwd, init)
    y, hstates = hip.forward(persistent=True)
    assert_close(y, out, 1e-4, "outputs")
    for i, (a, b_) in enumerate(zip(hstates, states)):
        assert_close(a, b_, 1e-4, f"state {i}")
    grads = hip.backward(R, S, persistent=True)
    assert_close(grads[0]["dx"] + grads[1]["dx"], xl.grad, 2e-4, "dx")
    for d in range(2):
        W, U, b = leaves[3 * d: 3 * d + 3]
        assert_close(grads[d]["dW"], W.grad, 2e-4, f"dW[{d}]")
        assert_close(grads[d]["dU"], U.grad, 2e-4, f"dU[{d}]")
        assert_close(grads[d]["db"], b.grad, 2e-4, f"db[{d}]")
        if init:
            assert_close(grads[d]["dh0"], il[d * nst].grad, 2e-4, f"dh0[{d}]")
            if rt == "lstm":
                assert_close(grads[d]["dc0"], il[d * nst + 1].grad, 2e-4, f"dc0[{d}]")


def test_sweep_timeout_raises_the_error_words_and_drains():
    """spin limit 0: every gather gives up at once - the launches must still terminate, set the per-launch error word and the
    caller's sticky flag, and a following launch with the normal limit must be clean again."""
    from speech_recognition_amd import ops
    rt, B, T, D, H = "lstm", 32, 12, 4, 64
    g = torch.Generator().manual_seed(5)
    fwd, bwd = make_params(rt, D, H, g, 0.1)
    x = torch.randn(B, T, D, generator=g, dtype=torch.float64)
    hip = HipBiRNN(rt, x, None, fwd, bwd, None)
    ws = ops.rnn_persist_ws(B, H, 2)
    flag = torch.zeros(1, device="cuda")
    ops.rnn_sweep_set_spin_limit(0)
    try:
        ops.rnn_seq_fwd_persist(hip.seq, ws, flag)
        torch.cuda.synchronize()
        assert ops.rnn_persist_error(ws) and float(flag[0]) == 1.0
    finally:
        ops.rnn_sweep_set_spin_limit(1 << 20)
    ops.rnn_seq_fwd_persist(hip.seq, ws, flag)
    torch.cuda.synchronize()
    assert not ops.rnn_persist_error(ws)
    assert float(flag[0]) == 1.0, "the sticky flag is only ever cleared by its owner"
    y_ok = hip.y.clone()
    hip2 = HipBiRNN(rt, x, None, fwd, bwd, None)
    hip2.forward(persistent=False)
    assert torch.equal(y_ok, hip2.y)


@pytest.mark.parametrize("rt,B,T,D,H", [("lstm", 40, 5, 8, 512), ("gru", 64, 4, 6, 516)])
def test_wide_step_kernels_mixed_precision(rt, B, T, D, H):
    """--mixed-precision on the wide step kernels (H >= 512, several batch tiles): bf16 weight images and bf16-rounded
    states on the bf16 MFMA, f32 accumulation.  Outputs and gradients stay within bf16 rounding of the f32 kernels
    and are not identical to them (the images are in use)."""
    from speech_recognition_amd import ops
    g = torch.Generator().manual_seed(B + H)
    fwd, bwd = make_params(rt, D, H, g, 0.05)
    x = torch.randn(B, T, D, generator=g, dtype=torch.float64)
    mask = torch.randn(B, T, generator=g) > -0.5
    dy = torch.randn(B, T, 2 * H, generator=g, dtype=torch.float64)
    nst = 2 if rt == "lstm" else 1
    res = {}
    for mode in (False, True):
        ops.set_mixed_precision(mode)
        try:
            hip = HipBiRNN(rt, x, mask, fwd, bwd, None)
            y, _ = hip.forward(persistent=False)
            y = y.clone()
            grads = hip.backward(dy, [None] * (2 * nst), persistent=False)
            res[mode] = (y, {f"{i}/{k}": v.clone() for i, r in enumerate(grads) for k, v in r.items()})
        finally:
            ops.set_mixed_precision(False)
    y0, y1 = res[False][0], res[True][0]
    d = float((y0 - y1).abs().max())
    assert 0.0 < d < 2e-2 * max(1.0, float(y0.abs().max())), d
    for k in res[False][1]:
        a, b = res[False][1][k], res[True][1][k]
        scale = max(1e-3, float(a.abs().max()))
        assert float((a - b).abs().max()) < 5e-2 * scale, (k, float((a - b).abs().max()), scale)


@pytest.mark.parametrize("B,T,H,masked,init", [(40, 14, 512, True, True), (64, 9, 1024, False, False), (18, 11, 768, True, False)])
def test_wide_sweep_equals_wide_step_kernels_mixed_precision(B, T, H, masked, init):
    """The weights-resident bf16 forward sweep of wide LSTM layers (rnn_sweep_wide.hip: las_large) against the wide step kernels
    under mixed precision, same layer and inputs: both round the recurrent kernel and the state operand to bf16 and accumulate
    in f32; the two differ in summation order, which now and then moves an h value across a bf16 rounding boundary of the next
    step's operand (2^-9 relative): outputs, states and saved gate activations within 1e-3 of the largest entry."""
    from speech_recognition_amd import ops
    from tests.rnn_helpers import HipBiRNN
    ops.set_mixed_precision(True)
    try:
        g = torch.Generator().manual_seed(H + B)
        fwd, bwd = make_params("lstm", 24, H, g, 0.03)
        x = torch.randn(B, T, 24, generator=g, dtype=torch.float64)
        mask = None
        if masked:
            mask = torch.ones(B, T, dtype=torch.bool)
            mask[1, T // 2:] = False
            mask[B - 1, 3:5] = False
            mask[B // 2, :2] = False
        init_states = [torch.randn(B, H, generator=g) * 0.3 for _ in range(4)] if init else None
        outs = []
        for wide in (True, False):
            hip = HipBiRNN("lstm", x, mask, fwd, bwd, init_states)
            if wide:
                assert ops.rnn_sweep_wide_supported("lstm", B, T, H, 2)
                ws = ops.rnn_sweep_wide_ws(B, H, 2)
                ops.rnn_sweep_wide_fwd(hip.seq, ws)
                torch.cuda.synchronize()
                assert not ops.rnn_persist_error(ws), "wide sweep: a hand-off timed out"
            else:
                ops.rnn_seq_fwd(hip.seq)
            outs.append(dict(y=hip.y.clone(), **{f"{k}{d}": dd[k].clone() for d, dd in enumerate(hip.dirs) for k in ("hseq", "cseq", "saved")}))
        for k, ref in outs[1].items():
            assert_close(outs[0][k], ref, 1e-3, k)
    finally:
        ops.set_mixed_precision(False)


def test_wide_sweep_timeout_sets_the_error_word_and_drains():
    """spin limit 0 on the wide sweep: the gather waves give up at once, every wave of every workgroup must still reach the
    workgroup barrier of the step and leave (no hang), the error word and the sticky flag are set; the next launch is clean."""
    from speech_recognition_amd import ops
    ops.set_mixed_precision(True)
    try:
        B, T, H = 20, 6, 512
        g = torch.Generator().manual_seed(9)
        fwd, bwd = make_params("lstm", 8, H, g, 0.05)
        x = torch.randn(B, T, 8, generator=g, dtype=torch.float64)
        hip = HipBiRNN("lstm", x, None, fwd, bwd, None)
        ws = ops.rnn_sweep_wide_ws(B, H, 2)
        flag = torch.zeros(1, device="cuda")
        ops.rnn_sweep_set_spin_limit(0)
        try:
            ops.rnn_sweep_wide_fwd(hip.seq, ws, flag)
            torch.cuda.synchronize()
            assert ops.rnn_persist_error(ws) and float(flag[0]) == 1.0
        finally:
            ops.rnn_sweep_set_spin_limit(1 << 20)
        ops.rnn_sweep_wide_fwd(hip.seq, ws, flag)
        torch.cuda.synchronize()
        assert not ops.rnn_persist_error(ws)
    finally:
        ops.set_mixed_precision(False)
