"""Run the decoder sweep on one shape and print the error words (last code, earliest code)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from speech_recognition_amd.models import LAS
B, T, U, He, Hd = map(int, sys.argv[1:6])
dropout = float(sys.argv[6]) if len(sys.argv) > 6 else 0.0
g = torch.Generator().manual_seed(1)
audio = torch.randn(B, T, 20, 3, generator=g)
tokens = torch.randint(1, 97, (B, U), generator=g, dtype=torch.int32)
m = LAS("lstm", 97, He, Hd, 1, 2, dropout, 0.99, 0, seed=3).build(20, 3)
ws = m._workspace(B, T, U)
ws.toks_T[:U].copy_(tokens.t().cuda())
m.forward_ws(ws, audio.cuda(), True, True)
torch.cuda.synchronize()
w = ws.dsweep_ws[-288:-256].view(torch.int32).cpu()
per = ws.dsweep_ws[-256:].view(torch.int32).cpu().tolist()
last, first = int(w[0]), (0x7fffffff - int(w[1])) if int(w[1]) else 0
print(f"B={B} T={T} (T2={ws.T2}) U={U} He={He} Hd={Hd}: last code {last & 255} step {last >> 8}; earliest code {first & 255} step {first >> 8}")

from collections import Counter
cnt = Counter((c & 255, (c >> 8) & 255) for c in per if c)
print("  aborts by (code, step):", dict(cnt))
first_key = min(cnt, key=lambda k: (k[1], k[0])) if cnt else None
if first_key:
    print("  workgroups with the earliest abort", first_key, ":", [i for i, c in enumerate(per) if c and (c & 255, (c >> 8) & 255) == first_key][:40], 'piece/lane:', [((c >> 16) & 15, c >> 20) for c in per if c and (c & 255, (c >> 8) & 255) == first_key][:40])

# which blocks of the four state exchanges are still the sentinel, per slot
import numpy as np
Q = Hd // 4
hsz = 2 * Hd * 16
D = 2 * He
slot = 4 * hsz + 32 * 8 * (4 + D) + 2 * D * 16
raw = ws.dsweep_ws[: 4 * slot].view(torch.int32).cpu().numpy().reshape(4, slot)
SENT = np.int32(0x7FC0DEAD)
for sl_ in range(4):
    for name, off in (("h1", 0), ("c1", hsz), ("h0", 2 * hsz), ("c0", 3 * hsz)):
        blk = raw[sl_, off:off + hsz].reshape(2, Q, 64)
        missing = [(t_, q_) for t_ in range(2) for q_ in range(Q) if (blk[t_, q_] == SENT).any()]
        print(f"  slot {sl_} {name}: blocks holding a sentinel: {missing}")
