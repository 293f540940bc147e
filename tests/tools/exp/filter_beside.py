"""Does a filter-gradient convolution give the same result with ANOTHER filter-gradient kernel running beside it?  DeepSpeech2 conv1 (3 input
channels: the scalar im2col loader) at B = 16, 15 s, alone vs beside conv2's filter gradient - the pair that was co-resident when the
training step produced garbage (tests/tools/exp/halo_beside.py cleared the row-staged input-gradient kernel)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
import yaml

from speech_recognition_amd import ops

B = 16
g = torch.Generator().manual_seed(0)
cfg = yaml.safe_load(open(os.path.join(ROOT, "resources", "configs", "deepspeech.yml")))
ks, st, ch = cfg["kernel_sizes"], cfg["strides"], cfg["channels"]
shape = [B, 1499, 80, 3]
shapes = [tuple(shape)]
for c, (kt, kf), (s0, s1) in zip(ch, ks, st):
    shape = [B, (shape[1] - kt) // s0 + 1, (shape[2] - kf) // s1 + 1, c]
    shapes.append(tuple(shape))
x0, x1, x2, x3 = (torch.randn(*s, generator=g).cuda() for s in shapes)
dy0 = torch.randn(*shapes[1], generator=g).cuda()
dy1 = torch.randn(*shapes[2], generator=g).cuda()
w0 = torch.zeros(ks[0][0], ks[0][1], 3, ch[0]).cuda()
w1 = torch.zeros(ks[1][0], ks[1][1], ch[0], ch[1]).cuda()
ref = torch.zeros_like(w0)
ops.conv2d_bwd_filter(x0, dy0, ref, tuple(st[0]))
torch.cuda.synchronize()
side = torch.cuda.Stream()
for it in range(8):
    gw0, gw1 = torch.zeros_like(w0), torch.zeros_like(w1)
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        ops.conv2d_bwd_filter(x1, dy1, gw1, tuple(st[1]))
        if it % 2:
            ops.conv2d_bwd_filter(x1, dy1, gw1, tuple(st[1]))
    ops.conv2d_bwd_filter(x0, dy0, gw0, tuple(st[0]))
    torch.cuda.synchronize()
    d = float((gw0 - ref).abs().max()) / float(ref.abs().max())
    print(f"run {it}: max |dw - dw alone| / max |dw| = {d:.3e}   finite {bool(torch.isfinite(gw0).all())}")
