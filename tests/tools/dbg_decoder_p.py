import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from speech_recognition_amd.models import LAS
from speech_recognition_amd.models import las as las_mod
B, T, U, He, Hd = map(int, sys.argv[1:6]); dropout = float(sys.argv[6])
V = 97
g = torch.Generator().manual_seed(B + T + U)
audio = torch.randn(B, T, 20, 3, generator=g)
audio[1, T // 2:] = 0.0
audio[B - 1, 3 * T // 4:] = 0.0
tokens = torch.randint(1, V, (B, U), generator=g, dtype=torch.int32)
tokens[1, U // 2:] = 0
outs = {}
for sweep in (True, False):
    las_mod.DECODER_SWEEP = sweep
    m = LAS("lstm", V, He, Hd, 1, 2, dropout, 0.99, 0, seed=3).build(20, 3)
    m.state[1] = 77
    ws = m._workspace(B, T, U)
    ws.toks_T[:U].copy_(tokens.t().cuda())
    m.forward_ws(ws, audio.cuda(), True, True)
    torch.cuda.synchronize()
    outs[sweep] = dict(p=ws.p.clone().cpu(), ctx=ws.ctx.clone().cpu(), mask=ws.mask.clone().cpu(), hin=ws.hin.clone().cpu())
a, b = outs[True], outs[False]
bad = [(i, r, round((a["p"][i, r] - b["p"][i, r]).abs().max().item(), 5)) for i in range(U) for r in range(B) if (a["p"][i, r] - b["p"][i, r]).abs().max().item() > 1e-5]
print("p mismatches (step, row, max diff):", bad[:12], "of", len(bad))
if bad:
    i, r, _ = bad[0]
    d = (a["p"][i, r] - b["p"][i, r]).abs()
    print("  first: frames off", [t for t in range(d.numel()) if d[t] > 1e-5][:20], "sum sweep", a["p"][i, r].sum().item(), "sum ref", b["p"][i, r].sum().item())
print("ctx diff", (a["ctx"] - b["ctx"]).abs().max().item(), "hin diff", (a["hin"] - b["hin"]).abs().max().item())
