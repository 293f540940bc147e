"""Row-staged version 2 (ASR_CONV_HALO_V=2, gate lifted) against the general kernels: forward with an odd and an even number of kernel rows, and the input gradient."""
import os, sys
os.environ["ASR_CONV_HALO_V"] = "2"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import torch
from speech_recognition_amd import ops
g = torch.Generator().manual_seed(0)
for name, xs, wsz, st in (("kh 21", (2, 60, 25, 32), (21, 11, 32, 32), (2, 1)), ("kh 20", (2, 60, 25, 32), (20, 11, 32, 32), (2, 1)), ("kh 10 stride 1", (2, 40, 25, 32), (10, 11, 32, 32), (1, 1))):
    x = torch.randn(*xs, generator=g).cuda(); w = (torch.randn(*wsz, generator=g) * 0.05).cuda()
    Ho, Wo = (xs[1] - wsz[0]) // st[0] + 1, xs[2] - wsz[1] + 1
    res = {}
    for force in (0, 1):
        ops.lib().asr_conv2d_halo_force(force)
        y = torch.zeros(xs[0], Ho, Wo, wsz[3], device="cuda"); ops.conv2d_fwd(x, w, None, st, y=y)
        dy = torch.randn(xs[0], Ho, Wo, wsz[3], generator=torch.Generator().manual_seed(1)).cuda()
        dx = torch.zeros_like(x); ops.conv2d_bwd_data(dy, w, dx, st)
        torch.cuda.synchronize(); res[force] = (y.clone(), dx.clone())
    ops.lib().asr_conv2d_halo_force(0)
    ey = float((res[0][0] - res[1][0]).abs().max() / res[0][0].abs().max()); d = (res[0][1] - res[1][1]).abs()
    edx = float(d.max() / res[0][1].abs().max())
    bad_rows = sorted(set((d.amax(dim=(0, 2, 3)) > 1e-4 * float(res[0][1].abs().max())).nonzero().flatten().tolist()))
    print(f"{name:16s} forward rel err {ey:.2e}   input gradient rel err {edx:.2e}   bad input rows {bad_rows[:24]}{'...' if len(bad_rows) > 24 else ''}")
