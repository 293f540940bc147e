"""Time the one-launch decoder sweep on the las_small geometry (B=32, T'=249, U=64, Hd=256, D=512)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from speech_recognition_amd.models import LAS
B, T, U = 32, 999, 64
g = torch.Generator().manual_seed(1)
audio = torch.randn(B, T, 80, 3, generator=g).cuda()
tokens = torch.randint(1, 16000, (B, U), generator=g, dtype=torch.int32)
m = LAS("lstm", 16000, 256, 256, 3, 2, 0.15, 0.99, 0, seed=3).build(80, 3)
ws = m._workspace(B, T, U)
ws.toks_T[:U].copy_(tokens.t().cuda())
m.forward_ws(ws, audio, True, True)
torch.cuda.synchronize()
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
us = t(lambda: m._decoder_sweep(ws, True))
print(f"decoder sweep: {us:.1f} us = {us / U:.2f} us/step (delay {os.environ.get('ASR_DECODER_SWEEP_DELAY', 'default')})")
# backward sweep: needs the loss gradient in ws.dyd (any values do for timing)
ws.dyd.normal_(0, 1e-3)
m._decoder_sweep_bwd(ws)
torch.cuda.synchronize()
from speech_recognition_amd import ops
assert not ops.decoder_sweep_error(ws.dsweep_bwd_ws), "backward sweep timed out"
us = t(lambda: m._decoder_sweep_bwd(ws))
print(f"decoder backward sweep: {us:.1f} us = {us / U:.2f} us/step (delay {os.environ.get('ASR_DECODER_SWEEP_BWD_DELAY', 'default')})")
