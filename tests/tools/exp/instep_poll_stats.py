"""Poll statistics of the encoder BPTT sweeps inside the las_small training step (diagnosis words 25 / 28 of each layer's workspace)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch

import bench

wl = bench.WORKLOADS["las_small"] if hasattr(bench, "WORKLOADS") else None
assert wl is not None
audio, n, toks = bench.synthetic_batch(0, wl)
audio_d, n_d, toks_d = torch.from_numpy(audio).cuda(), torch.from_numpy(n).cuda(), torch.from_numpy(toks).cuda()
trainer, model = bench.build_trainer(wl)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 25
for _ in range(5):
    ws = trainer.step(audio_d, n_d, toks_d, use_teacher_forcing=True)
torch.cuda.synchronize()
for i, lw in enumerate(ws.layers):
    buf = lw.get("rnn") if isinstance(lw, dict) else None
    if buf and "persist_bwd_ws" in buf:
        buf["persist_bwd_ws"][-32:].view(torch.int32)[25:27] = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(trainer.stream)
for _ in range(steps):
    ws = trainer.step(audio_d, n_d, toks_d, use_teacher_forcing=True)
e1.record(trainer.stream)
torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) / steps:.3f} ms per step")
for i, lw in enumerate(ws.layers):
    buf = lw.get("rnn") if isinstance(lw, dict) else None
    if buf and "persist_bwd_ws" in buf:
        w = buf["persist_bwd_ws"][-32:].view(torch.int32)
        T = buf["T"]
        print(f"layer {i}: T {T}  early first polls {int(w[25]) / max(int(w[26]), 1) / (T - 1):.3f}")
