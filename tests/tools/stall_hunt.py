"""Per-launch durations of one sweep, many launches: looks for the rare whole-launch stalls (tens of ms) seen on the shared host.
python tests/tools/stall_hunt.py [shape] [launches] [fwd|bwd]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from speech_recognition_amd import ops
from tests.rnn_helpers import HipBiRNN
from tests.test_rnn_gpu import make_params
from tests.tools.bench_sweep import SHAPES

name = sys.argv[1] if len(sys.argv) > 1 else "deepspeech"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
rt, B, T, D, H = SHAPES[name]
g = torch.Generator().manual_seed(1)
fwd, bwd = make_params(rt, D, H, g, 0.08)
x = torch.randn(B, T, D, generator=g, dtype=torch.float64)
hip = HipBiRNN(rt, x, None, fwd, bwd, None)
ws = ops.rnn_persist_ws(B, H, 2)
for _ in range(5):
    ops.rnn_seq_fwd_persist(hip.seq, ws)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    ops.rnn_seq_fwd_persist(hip.seq, ws)
    ev[i + 1].record()
torch.cuda.synchronize()
dt = torch.tensor([ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n)])
med = float(dt.median())
out = [(i, round(float(v), 1)) for i, v in enumerate(dt) if v > 3 * med]
print(f"{name} forward sweep x {n}: median {med:.1f} us, max {float(dt.max()):.1f} us, launches over 3x median: {len(out)} {out[:12]}")
print("error word", ops.rnn_persist_error(ws))
