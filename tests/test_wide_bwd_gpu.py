"""The wide layers' one-launch BPTT sweep (csrc/rnn_sweep_wide_bwd.hip; las_large: H = 1024 under mixed precision) against the per-step
staged kernels it replaces, on the same saved activations: same bf16 operands (ds, U), f32 accumulation - the sweep additionally rounds the
32 partial sums of every dh element to bf16 for the exchange, so the comparison is in relative L2 (1e-2 per tensor; measured up to 3e-3 on dh0 after 33 chained steps),
not bit-exact.  Masks (ragged lengths incl. a fully padded tail), a ragged batch (rows >= B never stored), chained final-state
gradients, initial states, both directions."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a - b).norm()) / max(float(b.norm()), 1e-30)


@pytest.mark.parametrize("B,T,masked,states", [(64, 12, False, False), (50, 9, True, True), (7, 33, True, False), (64, 3, False, True), (5, 2, True, True), (1, 7, False, False)])
def test_wide_bwd_sweep_equals_the_staged_step_kernels(B, T, masked, states):
    from speech_recognition_amd import ops
    from tests.rnn_helpers import HipBiRNN
    from tests.test_rnn_gpu import make_params
    D, H = 48, 1024
    g = torch.Generator().manual_seed(B * 100 + T)
    fwd, bwd = make_params("lstm", D, H, g, 0.03)
    x = torch.randn(B, T, D, generator=g, dtype=torch.float64)
    mask = None
    if masked:
        lens = torch.randint(1, T + 1, (B,), generator=g)
        lens[0] = T
        mask = torch.arange(T)[None, :] < lens[:, None]
    init = [torch.randn(B, H, generator=g, dtype=torch.float64) * 0.3 for _ in range(4)] if states else None
    dst = [torch.randn(B, H, generator=g, dtype=torch.float64) * 0.1 for _ in range(4)] if states else [None] * 4
    dy = torch.randn(B, T, 2 * H, generator=g, dtype=torch.float64) * 0.05
    ops.set_mixed_precision(True)
    try:
        outs = []
        for wide in (False, True, "images"):       # "images": ds leaves the sweep as the two bf16 images + bias sums (round 4) instead of f32
            hip = HipBiRNN("lstm", x, mask, fwd, bwd, init)
            hip.forward(persistent=False)
            outs.append(hip.backward(dy, dst, wide=wide))
    finally:
        ops.set_mixed_precision(False)
    for which in (1, 2):
        for d in range(2):
            ref, got = outs[0][d], outs[which][d]
            for key in ("dW", "dU", "db", "dx", "dh0", "dc0"):
                assert torch.isfinite(got[key]).all(), (d, key)
                err = _rel(got[key], ref[key])
                # (the images carry ds rounded to bf16: the reference products here use the unrounded f32 ds of the step kernels)
                assert err < (1e-2 if which == 1 else 1.5e-2), f"{'f32 ds' if which == 1 else 'bf16 images'}: direction {d} {key}: relative L2 {err:.2e}"


def test_wide_bwd_sweep_timeout_is_reported_not_hung():
    """With the spin limit forced to 0 every gather gives up at its first stale probe: the launch must end (all ten barriers of the
    abort path are taken by every wave), raise the error word and the caller's flag, leave a diagnosis record - and the next launch with
    the normal limit must be clean and correct again (the fill kernel re-arms the exchange and the per-launch diagnosis words)."""
    from speech_recognition_amd import ops
    from tests.rnn_helpers import HipBiRNN
    from tests.test_rnn_gpu import make_params
    B, T, D, H = 33, 6, 32, 1024
    g = torch.Generator().manual_seed(3)
    fwd, bwd = make_params("lstm", D, H, g, 0.03)
    x = torch.randn(B, T, D, generator=g, dtype=torch.float64)
    dy = torch.randn(B, T, 2 * H, generator=g).cuda() * 0.05
    ops.set_mixed_precision(True)
    try:
        hip = HipBiRNN("lstm", x, None, fwd, bwd, None)
        hip.forward(persistent=False)
        gds = [dict(direct=torch.zeros(B, H, device="cuda"), dy_carry=torch.zeros(B, H, device="cuda"), dh0=torch.zeros(B, H, device="cuda"),
                    dc=torch.zeros(B, H, device="cuda"), ds=torch.zeros_like(dd["saved"])) for dd in hip.dirs]
        ws = ops.rnn_sweep_wide_bwd_ws(B, H, 2)
        flag = torch.zeros(1, device="cuda")
        ops.rnn_sweep_set_spin_limit(0)
        try:
            ops.rnn_sweep_wide_bwd(hip.seq, dy, gds, ws, flag)
            torch.cuda.synchronize()
        finally:
            ops.rnn_sweep_set_spin_limit(1 << 20)
        assert ops.rnn_persist_error(ws) != 0 and float(flag) == 1.0
        rep = ops.sweep_diagnosis(ws, "rnn_sweep_wide_bwd", clear=True)
        assert rep and rep["expected"] == 256 and rep["verdict"] in ("absent workgroup", "lost hand-off"), rep
        for gd in gds:
            gd["dc"].zero_()
        ops.rnn_sweep_wide_bwd(hip.seq, dy, gds, ws, None)
        torch.cuda.synchronize()
        assert ops.rnn_persist_error(ws) == 0
        ref = HipBiRNN("lstm", x, None, fwd, bwd, None)
        ref.forward(persistent=False)
        want = ref.backward(dy.cpu().double(), [None] * 4)
        for d, dd in enumerate(hip.dirs):
            dsr = ref.dirs[d]["saved"]                                  # (the step kernels leave ds over the saved activations)
            err = float((gds[d]["ds"] - dsr).norm()) / float(dsr.norm())
            assert err < 1e-2, (d, err)
    finally:
        ops.set_mixed_precision(False)
