"""Time the one-launch recurrent sweeps (forward and backward-through-time) of one BiRNN layer, both generations:
ASR_SWEEP=1 sentinel hand-offs (rnn_sweep*.hip) and ASR_SWEEP=0 tagged granules / ds all-gather (rnn_persist*.hip).
Prints microseconds per launch and per dependent step; checks the two generations against each other.

  python tests/tools/bench_sweep.py [--shapes las_small,deepspeech] [--iters 20]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch

SHAPES = {"las_small": ("lstm", 32, 249, 512, 256), "deepspeech": ("gru", 16, 168, 256, 128), "las_b64": ("lstm", 64, 249, 512, 256),
          "small": ("lstm", 16, 64, 32, 64)}


def time_fn(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="las_small,deepspeech")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--masked", action="store_true")
    args = ap.parse_args()
    from speech_recognition_amd import ops
    from tests.rnn_helpers import HipBiRNN
    from tests.test_rnn_gpu import make_params
    for name in args.shapes.split(","):
        rt, B, T, D, H = SHAPES[name]
        g = torch.Generator().manual_seed(1)
        fwd, bwd = make_params(rt, D, H, g, 0.08)
        x = torch.randn(B, T, D, generator=g, dtype=torch.float64)
        mask = (torch.randn(B, T, generator=g) > -0.5) if args.masked else None
        dy = torch.randn(B, T, 2 * H, generator=g).cuda()
        nst = 2 if rt == "lstm" else 1
        results = {}
        for sweep in ((True,) if os.environ.get("ASR_SWEEP_DBG") else (True, False)):
            ops.SWEEP = sweep
            if not ops.rnn_persist_supported(rt, B, T, H, 2) or not ops.rnn_persist_bwd_supported(rt, B, T, H, 2):
                print(f"{name}: generation sweep={sweep} does not take this shape")
                continue
            hip = HipBiRNN(rt, x, mask, fwd, bwd, None)
            mode = os.environ.get("BENCH_SWEEP_OUT", "both")          # what the forward sweep writes for backward: both | saved | coef
            if mode != "both":
                drop = "coef" if mode == "saved" else "saved"
                hip.seq = ops.make_rnn_seq(rt, B, T, H, [dict(dd, **{drop: None}) for dd in hip.dirs], hip.mask, hip.y, [0, H])
            ws = ops.rnn_persist_ws(B, H, 2)
            t_f = time_fn(lambda: ops.rnn_seq_fwd_persist(hip.seq, ws), args.iters)
            assert not ops.rnn_persist_error(ws), "forward hand-off timed out"
            y = hip.y.clone()
            gds = [dict(direct=torch.zeros(B, H, device="cuda"), dy_carry=torch.zeros(B, H, device="cuda"), dh0=torch.zeros(B, H, device="cuda"),
                        dc=torch.zeros(B, H, device="cuda"), ds=torch.empty_like(dd["saved"])) for dd in hip.dirs]
            pws = ops.rnn_persist_bwd_ws(B, H, 2)

            def bwd_once():
                for gd in gds:
                    gd["dc"].zero_()
                ops.rnn_seq_bwd(hip.seq, dy, gds, pws)      # (ds goes to its own buffers: the saved activations stay)

            def copy_only():
                for gd in gds:
                    gd["dc"].zero_()

            t_b = time_fn(bwd_once, args.iters) - time_fn(copy_only, args.iters)
            assert not ops.rnn_persist_error(pws), "backward hand-off timed out"
            results[sweep] = (y, [gd["ds"].clone() for gd in gds], [gd["dh0"].clone() for gd in gds])
            loc = lambda w: tuple(int(v) for v in w[-32:].view(torch.int32)[2:4].tolist())     # (XCD-local workgroups, all)
            print(f"{name} {rt} B={B} T={T} H={H} sweep={int(sweep)}: fwd {t_f:8.1f} us = {t_f / T:5.2f} us/step   "
                  f"bwd {t_b:8.1f} us = {t_b / T:5.2f} us/step   XCD-local workgroups fwd {loc(ws)} bwd {loc(pws)}", flush=True)
        if len(results) == 2:
            ya, sa, ha = results[True]
            yb, sb, hb = results[False]
            print(f"   forward outputs identical: {torch.equal(ya, yb)};  ds max diff {max(float((a - b).abs().max()) for a, b in zip(sa, sb)):.2e} "
                  f"(max |ds| {max(float(b.abs().max()) for b in sb):.2e});  dh0 max diff {max(float((a - b).abs().max()) for a, b in zip(ha, hb)):.2e}")
    ops.SWEEP = True


if __name__ == "__main__":
    main()
