"""Do the filter-gradient convolutions, the input-gradient convolutions or colsum write outside their outputs?  Every output sits in the middle of
a sentinel-filled arena (DeepSpeech2 geometry, B = 16, 15 s); the guard regions are checked after each call."""
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
xs = [torch.randn(*s, generator=g).cuda() for s in shapes]
cin = [3] + ch[:-1]
GUARD = 1 << 20                                             # floats either side
SENT = 12345.0


def guarded(shape):
    n = 1
    for v in shape:
        n *= v
    arena = torch.full((n + 2 * GUARD,), SENT, device="cuda")
    out = arena[GUARD:GUARD + n].view(*shape)
    return arena, out, n


def check(name, arena, n):
    torch.cuda.synchronize()
    lo, hi = arena[:GUARD], arena[GUARD + n:]
    bad = int((lo != SENT).sum()) + int((hi != SENT).sum())
    print(f"{name:38s} guard words changed: {bad}" + ("" if not bad else f"   (below: {int((lo != SENT).sum())}, above: {int((hi != SENT).sum())}; first above at +{int((hi != SENT).nonzero()[0]) if int((hi != SENT).sum()) else -1})"))


for k in range(3):
    arena, dw, n = guarded((ks[k][0], ks[k][1], cin[k], ch[k]))
    dw.zero_()
    ops.conv2d_bwd_filter(xs[k], xs[k + 1], dw, tuple(st[k]))
    check(f"conv{k + 1} filter gradient", arena, n)
    arena, db, n = guarded((ch[k],))
    db.zero_()
    ops.colsum(xs[k + 1].view(-1, ch[k]), db)
    check(f"conv{k + 1} bias gradient (colsum)", arena, n)
    if k > 0:
        w = (torch.randn(ks[k][0], ks[k][1], cin[k], ch[k], generator=g) * 0.05).cuda()
        arena, dx, n = guarded(shapes[k])
        ops.conv2d_bwd_data(xs[k + 1], w, dx, tuple(st[k]))
        check(f"conv{k + 1} input gradient", arena, n)
        arena, y, n = guarded(shapes[k + 1])
        ops.conv2d_fwd(xs[k], w, None, tuple(st[k]), y=y)
        check(f"conv{k + 1} forward", arena, n)
