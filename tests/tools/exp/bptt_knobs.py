"""Encoder BPTT sweep (las_small layer) under its polling knobs, in one process:
ASR_SWEEP_BWD_DELAY (10 ns ticks between entering a gather and its first poll); prints the share of gathers whose first poll failed.
  python tests/tools/exp/bptt_knobs.py [ENV_NAME[,ENV_NAME...] values,of,the,first values,of,the,second ...]"""
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch

from speech_recognition_amd import ops
from tests.rnn_helpers import HipBiRNN
from tests.test_rnn_gpu import make_params

rt, B, T, D, H = "lstm", 32, 249, 512, 256
g = torch.Generator().manual_seed(1)
fwd, bwd = make_params(rt, D, H, g, 0.08)
x = torch.randn(B, T, D, generator=g, dtype=torch.float64)
hip = HipBiRNN(rt, x, None, fwd, bwd, None)
hip.forward(persistent=True)
dy = torch.randn(B, T, 2 * H, generator=g).cuda()
gds = [dict(direct=torch.zeros(B, H, device="cuda"), dy_carry=torch.zeros(B, H, device="cuda"), dh0=torch.zeros(B, H, device="cuda"),
            dc=torch.zeros(B, H, device="cuda"), ds=torch.empty_like(dd["saved"])) for dd in hip.dirs]
pws = ops.rnn_persist_bwd_ws(B, H, 2)


def run(iters=20):
    for _ in range(3):
        ops.rnn_seq_bwd(hip.seq, dy, gds, pws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.rnn_seq_bwd(hip.seq, dy, gds, pws)
    e1.record()
    e1.synchronize()
    assert not ops.rnn_persist_error(pws)
    return e0.elapsed_time(e1) * 1e3 / iters


def poll_stats():
    w = pws[-32:].view(torch.int32)
    nfail, waves = int(w[25]), int(w[26])
    w[25:27] = 0
    return nfail / max(waves, 1) / (T - 1)


names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["ASR_SWEEP_BWD_DELAY"]
grids = [[int(v) for v in a.split(",")] for a in sys.argv[2:]] or [[0, 60, 80, 90, 100, 110, 120, 140]]
for combo in itertools.product(*grids):
    for n, v in zip(names, combo):
        os.environ[n] = str(v)
    poll_stats()
    us = run()
    fr = poll_stats()
    print(f"early first polls {fr:5.3f} ", end="")
    print(" ".join(f"{n.replace('ASR_SWEEP_', '')}={v}" for n, v in zip(names, combo)), f"{us:8.1f} us = {us / T:.3f} us/step", flush=True)
