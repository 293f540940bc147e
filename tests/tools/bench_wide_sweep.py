"""Time the wide (bf16 weights-resident) forward sweep against the wide step kernels on the las_large layer (B=64, T'=499, H=1024)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from speech_recognition_amd import ops
from tests.rnn_helpers import HipBiRNN
from tests.test_rnn_gpu import make_params
from tests.tools.bench_sweep import time_fn

B, T, H = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 499, 1024)))
ops.set_mixed_precision(True)
g = torch.Generator().manual_seed(1)
fwd, bwd = make_params("lstm", 32, H, g, 0.02)
x = torch.randn(B, T, 32, generator=g, dtype=torch.float64)
hip = HipBiRNN("lstm", x, None, fwd, bwd, None)
ws = ops.rnn_sweep_wide_ws(B, H, 2)
t_w = time_fn(lambda: ops.rnn_sweep_wide_fwd(hip.seq, ws), 5)
assert not ops.rnn_persist_error(ws) or os.environ.get("ASR_SWEEP_DBG"), "wide sweep timed out"
t_s = time_fn(lambda: ops.rnn_seq_fwd(hip.seq), 3)
w = ws[-32:].view(torch.int32)
print(f"B={B} T={T} H={H}: wide sweep {t_w:9.1f} us = {t_w / T:6.2f} us/step    step kernels {t_s:9.1f} us = {t_s / T:6.2f} us/step"
      f"    whole gathers repeated per wave and step {int(w[25]) / max(int(w[26]), 1) / (T - 1):.3f}")
# the backward sweep (rnn_sweep_wide_bwd.hip) against the staged step kernels
dy = torch.randn(B, T, 2 * H, generator=g).cuda() * 0.05
ops.rnn_seq_fwd(hip.seq)
gds = [dict(direct=torch.zeros(B, H, device="cuda"), dy_carry=torch.zeros(B, H, device="cuda"), dh0=torch.zeros(B, H, device="cuda"),
            dc=torch.zeros(B, H, device="cuda"), ds=torch.empty_like(dd["saved"])) for dd in hip.dirs]
if ops.rnn_sweep_wide_bwd_supported("lstm", B, T, H, 2):
    wb = ops.rnn_sweep_wide_bwd_ws(B, H, 2)
    t_b = time_fn(lambda: ops.rnn_sweep_wide_bwd(hip.seq, dy, gds, wb), 5)
    assert not ops.rnn_persist_error(wb) or os.environ.get("ASR_SWEEP_DBG"), f"wide backward sweep timed out: {ops.sweep_diagnosis(wb, 'rnn_sweep_wide_bwd')}"
    print(f"B={B} T={T} H={H}: wide BPTT sweep {t_b:9.1f} us = {t_b / T:6.2f} us/step", flush=True)
if not os.environ.get("ASR_SWEEP_DBG"):
    t_s = time_fn(lambda: ops.rnn_seq_bwd(hip.seq, dy, gds), 2)
    print(f"B={B} T={T} H={H}: staged step kernels {t_s:9.1f} us = {t_s / T:6.2f} us/step")
