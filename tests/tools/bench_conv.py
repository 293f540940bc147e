"""GPU micro-benchmark of the conv kernels on the LAS / DeepSpeech2 shapes (not a test)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speech_recognition_amd import ops

SHAPES = [  # name, B, H, W, C, kh, kw, sh, sw, O
    ("las conv1", 32, 999, 80, 3, 3, 3, 2, 2, 32), ("las conv2", 32, 499, 39, 32, 3, 3, 2, 2, 32),
    ("ds2 conv1", 16, 1499, 80, 3, 41, 11, 2, 2, 32), ("ds2 conv2", 16, 730, 35, 32, 21, 11, 2, 1, 32), ("ds2 conv3", 16, 355, 25, 32, 21, 11, 2, 1, 96),
]

def t(fn, reps=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for name, B, H, W, C, kh, kw, sh, sw, O in SHAPES:
    Ho, Wo = (H - kh) // sh + 1, (W - kw) // sw + 1
    x = torch.randn(B, H, W, C, device="cuda"); w = torch.randn(kh, kw, C, O, device="cuda") * 0.05
    b = torch.zeros(O, device="cuda"); y = torch.empty(B, Ho, Wo, O, device="cuda"); dy = torch.randn_like(y)
    dw = torch.zeros_like(w); dx = torch.empty_like(x)
    fl = 2.0 * B * Ho * Wo * O * kh * kw * C
    a = t(lambda: ops.conv2d_fwd(x, w, b, (sh, sw) if sh != sw else sh, y))
    c = t(lambda: ops.conv2d_bwd_filter(x, dy, dw, (sh, sw) if sh != sw else sh))
    d = t(lambda: ops.conv2d_bwd_data(dy, w, dx, (sh, sw) if sh != sw else sh))
    print(f"{name}: fwd {a:8.1f} us {fl/a/1e6:6.1f} TF | bwd_filter {c:8.1f} us {fl/c/1e6:6.1f} TF | bwd_data {d:8.1f} us {fl/d/1e6:6.1f} TF (useful)")
