"""Which gradient tensors differ between the overlap scheduler on / off (same model, batch, masks)?
python tests/tools/dbg_overlap.py [B T U He Hd]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from speech_recognition_amd import layers, ops
from speech_recognition_amd.models import LAS

B, T, U, He, Hd = (int(v) for v in sys.argv[1:6]) if len(sys.argv) > 5 else (32, 999, 12, 256, 256)
V = 97
g = torch.Generator().manual_seed(B + T + U)
audio = torch.randn(B, T, 20, 3, generator=g)
audio[1, T // 2:] = 0.0
tokens = torch.randint(1, V, (B, U + 1), generator=g, dtype=torch.int32)
tokens[1, U // 2:] = 0
outs = {}
for rep in range(3):
    for on in (False, True):
        layers.Overlap.enabled = on
        m = LAS("lstm", V, He, Hd, int(os.environ.get("LE", "1")), 2, 0.15, 0.99, 0, seed=3).build(20, 3)
        assert m._ov.on == on
        m.state[1] = 77
        ws, labels = m.train_workspace(B, T, U + 1)
        m.set_targets(ws, tokens.cuda(), labels)
        ops.fill(m.store.grad, 0.0)
        ag = audio.cuda()
        m.forward_ws(ws, ag, True, True)
        m.loss_and_grad(ws, labels)
        m.backward_ws(ws, ag)
        torch.cuda.synchronize()
        outs[on] = {k: v.clone() for k, v in m.store.grads().items()}
    bad = 0
    for k, ref in outs[False].items():
        d = float((outs[True][k] - ref).abs().max())
        s = float(ref.abs().max())
        if d > 3e-5 * max(s, 1e-6):
            bad += 1
            print(f"rep {rep}: {k}: max diff {d:.3e} (max |ref| {s:.3e}) finite={bool(torch.isfinite(outs[True][k]).all())}")
    print(f"rep {rep}: {bad} tensors differ")
