"""Does the row-staged input-gradient convolution (asr_conv2d_bwd_data_halo) give the same result with another kernel running beside it?
DeepSpeech2 conv2 geometry at B = 16, 15 s: dx alone vs dx while a side stream runs conv3's filter gradient (the combination that produced
garbage inside the training step)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch

from speech_recognition_amd import ops

B = 16
g = torch.Generator().manual_seed(0)
# conv stack of deepspeech.yml on [B, 1499, 80, 3]: kernels (41,11) (21,11) (21,11), strides (2,2) (2,1) (2,1)? read from the yml
import yaml
cfg = yaml.safe_load(open(os.path.join(ROOT, "resources", "configs", "deepspeech.yml")))
ks, st, ch = cfg["kernel_sizes"], cfg["strides"], cfg["channels"]
shape = [B, 1499, 80, 3]
shapes = [tuple(shape)]
for c, (kt, kf), (s0, s1) in zip(ch, ks, st):
    shape = [B, (shape[1] - kt) // s0 + 1, (shape[2] - kf) // s1 + 1, c]
    shapes.append(tuple(shape))
print("activations", shapes)
x0, x1, x2, x3 = (torch.randn(*s, generator=g).cuda() for s in shapes)
w1 = (torch.randn(ks[1][0], ks[1][1], ch[0], ch[1], generator=g) * 0.05).cuda()
w2 = (torch.randn(ks[2][0], ks[2][1], ch[1], ch[2], generator=g) * 0.05).cuda()
dy1 = torch.randn(*shapes[2], generator=g).cuda()        # gradient wrt conv2's output
dy2 = torch.randn(*shapes[3], generator=g).cuda()
dx_ref = torch.empty_like(x1)
ops.conv2d_bwd_data(dy1, w1, dx_ref, tuple(st[1]))
torch.cuda.synchronize()
print("row-staged route taken:", ops._conv_halo_ws(ops.conv_desc(dx_ref.shape, w1.shape, tuple(st[1])), 1) is not None)
side = torch.cuda.Stream()
gw2, gw1 = torch.zeros_like(w2), torch.zeros_like(w1)
worst = 0.0
for it in range(8):
    dx = torch.full_like(x1, float("nan"))
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        if it < 4:
            ops.conv2d_bwd_filter(x2, dy2, gw2, tuple(st[2]))       # conv3's filter gradient beside it
        else:
            ops.conv2d_bwd_filter(x1, dy1, gw1, tuple(st[1]))       # conv2's own filter gradient (same dy) beside it: the pair of the failing step
    ops.conv2d_bwd_data(dy1, w1, dx, tuple(st[1]))
    torch.cuda.synchronize()
    d = float((dx - dx_ref).abs().max())
    worst = max(worst, d if d == d else float("inf"))
    print(f"run {it}: max |dx - dx alone| = {d:.3e}   finite {bool(torch.isfinite(dx).all())}")
print("worst", worst)
