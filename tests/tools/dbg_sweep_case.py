"""Run one backward-sweep case and print the per-launch error word (diagnosis code | step << 8)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from speech_recognition_amd import ops
from tests.rnn_helpers import HipBiRNN
from tests.test_rnn_gpu import make_params

rt, B, T, D, H = sys.argv[1], *map(int, sys.argv[2:6])
g = torch.Generator().manual_seed(1)
fwd, bwd = make_params(rt, D, H, g, 0.1)
x = torch.randn(B, T, D, generator=g, dtype=torch.float64)
hip = HipBiRNN(rt, x, None, fwd, bwd, None)
hip.forward(persistent=True)
dy = torch.randn(B, T, 2 * H, generator=g).cuda()
gds = [dict(direct=torch.zeros(B, H, device="cuda"), dy_carry=torch.zeros(B, H, device="cuda"), dh0=torch.zeros(B, H, device="cuda"),
            dc=torch.zeros(B, H, device="cuda"), ds=torch.empty_like(dd["saved"])) for dd in hip.dirs]
pws = ops.rnn_persist_bwd_ws(B, H, 2)
ops.rnn_seq_bwd(hip.seq, dy, gds, pws)
torch.cuda.synchronize()
w = int(pws[-32:].view(torch.int32)[0].item())
print(f"{rt} B={B} T={T} H={H}: err word code {w & 255} step {w >> 8}")
