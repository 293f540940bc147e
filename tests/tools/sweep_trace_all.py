"""Hand-off time of the encoder BPTT sweep measured across workgroups (ASR_SWEEP_DBG bit 256): every workgroup of group 0 stamps
"gather complete" and "publish issued" for 64 steps; the hand-off into workgroup (i, j) at step p + 1 = its gather-complete stamp minus the
LATEST publish stamp of its G senders (column i) at step p.   python tests/tools/sweep_trace_all.py"""
import ctypes as C
import os
import sys

os.environ["ASR_SWEEP_DBG"] = str(int(os.environ.get("ASR_SWEEP_DBG", "0")) | 256)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from speech_recognition_amd import ops
from tests.rnn_helpers import HipBiRNN
from tests.test_rnn_gpu import make_params

rt, B, T, D, H = "lstm", 32, 249, 512, 256
G = 8
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
n = 512 * 8 + 64 * 64 * 2
buf = (C.c_ulonglong * n)()
assert ops.lib().asr_debug_sweep_trace(buf, n) == 0
t = np.array(buf[512 * 8:], dtype=np.float64).reshape(G, G, 64, 2) * 0.01     # [bx / G][bx % G][step][gather complete, publish issued] us
if os.environ.get("ASR_SWEEP_BWD_ROWXCD", "1") != "0":                          # block -> (i = bx % G, j = bx / G): make it [i][j]
    t = t.transpose(1, 0, 2, 3)
gc, pub = t[..., 0], t[..., 1]
local = pub - gc                                                              # gather complete -> publish issued, same step
print(f"local work (gather complete -> publish issued): mean {local.mean():.3f} us, per workgroup min {local.mean(-1).min():.3f} max {local.mean(-1).max():.3f}")
# workgroup (i, j) gathers the blocks published by the workgroups (i', i), i' = 0..G-1
hand_last = np.zeros((G, G, 63))
hand_first = np.zeros((G, G, 63))
for i in range(G):
    for j in range(G):
        senders = pub[:, i, :-1]                                              # [i'][step p]
        hand_last[i, j] = gc[i, j, 1:] - senders.max(0)
        hand_first[i, j] = gc[i, j, 1:] - senders.min(0)
print(f"hand-off (latest sender's publish -> gather complete): mean {hand_last.mean():.3f} us  min {hand_last.min():.3f}  median {np.median(hand_last):.3f}  max {hand_last.max():.3f}")
print(f"          (earliest sender's publish -> gather complete): mean {hand_first.mean():.3f} us")
print(f"spread of the publish stamps of one step over the 64 workgroups: mean {(pub.reshape(64, 64).max(0) - pub.reshape(64, 64).min(0)).mean():.3f} us")
print(f"spread over the 8 senders of a column: mean {(pub.max(0) - pub.min(0)).mean():.3f} us")
period = np.diff(gc, axis=-1)
print(f"step period {period.mean():.3f} us")
