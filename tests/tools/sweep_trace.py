"""Stage timeline of one workgroup of the encoder BPTT sweep (las_small layer): ASR_SWEEP_DBG=128 python tests/tools/sweep_trace.py"""
import ctypes as C
import os
import sys

os.environ["ASR_SWEEP_DBG"] = os.environ.get("ASR_SWEEP_DBG", "128")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
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
for _ in range(3):
    ops.rnn_seq_bwd(hip.seq, dy, gds, pws)
torch.cuda.synchronize()
n = 8 * (T - 1)
buf = (C.c_ulonglong * n)()
assert ops.lib().asr_debug_sweep_trace(buf, n) == 0
t = np.array(buf[:], dtype=np.float64).reshape(-1, 8)[5:, :6] * 0.01        # us
names = ["gather entered", "gather complete", "gate gradients done", "partial block in LDS", "publish wave saw partials", "publish issued"]
step = np.diff(t[:, 0])
print(f"step period {step.mean():.3f} us (min {step.min():.2f}, max {step.max():.2f})")
for k in range(1, 6):
    d = t[:, k] - t[:, k - 1]
    print(f"  {names[k - 1]:28s} -> {names[k]:28s} {d.mean():6.3f} us (median {np.median(d):.3f})")
d = t[1:, 0] - t[:-1, 5]
print(f"  {'publish issued':28s} -> {'next gather entered':28s} {d.mean():6.3f} us   (gather wave: partial written -> next gather entered "
      f"{(t[1:, 0] - t[:-1, 3]).mean():.3f})")
