"""Where does the garbage come from when DeepSpeech2's upper convolutions' filter gradients run on the side stream (ASR_DS2_CONV_BESIDE=6)?
Runs the backward pass of deepspeech.yml at B = 16 twice in one process - chain on one stream, then side by side - and compares every buffer of the
convolutions' backward pass."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from speech_recognition_amd import ops
from speech_recognition_amd.configs import get_model_config

B = 16
g = torch.Generator().manual_seed(5)
feats = torch.randn(B, 1499, 80, 3, generator=g).cuda()
toks = torch.randint(15, 29, (B, 96), generator=g, dtype=torch.int32)
model = get_model_config(os.path.join(ROOT, "resources", "configs", "deepspeech.yml")).create_model(seed=3)
model.build(80, 3)
ws, labels = model.train_workspace(B, feats.shape[1], toks.shape[1])
model.set_targets(ws, toks.cuda(), labels)


def run(mask):
    os.environ["ASR_DS2_CONV_BESIDE"] = str(mask)
    model.state[1] = 777
    ops.fill(model.store.grad, 0.0)
    for d in ws.dconv:
        d.fill_(float("nan"))
    torch.cuda.synchronize()
    model.forward_ws(ws, feats, True)
    model.loss_and_grad(ws, labels)
    model.backward_ws(ws, feats)
    torch.cuda.synchronize()
    out = {f"dconv[{i}]": d.clone() for i, d in enumerate(ws.dconv)}
    out["dx0"] = ws.dx0.clone()
    for n, t in model.store.grads().items():
        if n.startswith("convolution"):
            out[n] = t.clone()
    return out


ref = run(0)
for trial in range(3):
    got = run(6)
    for k in ref:
        a, b = ref[k].double(), got[k].double()
        bad = ~torch.isfinite(b) | ((a - b).abs() > 1e-3 * a.abs().max())
        nbad = int(bad.sum())
        print(f"trial {trial} {k:38s} max |ref| {float(a.abs().max()):.3e}  differing {nbad}/{a.numel()}" +
              ("" if not nbad else f"  largest {float(b[bad].abs().max()):.3e}  first index {np.unravel_index(int(bad.flatten().nonzero()[0]), tuple(a.shape))}"))
