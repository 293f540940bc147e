"""Stage timeline of the forward decoder sweep at the headline geometry (las_small: B = 32, T' = 249, U = 64, Hd = 256, D = 512):
   ASR_DECODER_SWEEP_TRACE=1 python tests/tools/decoder_trace.py
Two workgroups are stamped (decoder_sweep.hip): block 0 = attention chunk (row 0, chunk 0) + a layer-0 cell, block 128 = attention chunk
(row 16, chunk 0) + a layer-1 cell.  Prints, per workgroup, the mean time between consecutive stamps over the steady-state steps."""
import ctypes as C
import os
import sys

os.environ["ASR_DECODER_SWEEP_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from speech_recognition_amd import ops
from speech_recognition_amd.configs import get_model_config

B, T, U = 32, 999, 64
model = get_model_config(os.path.join(ROOT, "resources", "configs", "las_small.yml")).create_model(seed=3)
model.build(80, 3)
g = torch.Generator().manual_seed(0)
feats = torch.randn(B, T, 80, 3, generator=g).cuda()
toks = torch.randint(17, 16000, (B, U + 1), generator=g, dtype=torch.int32)
ws, labels = model.train_workspace(B, T, U + 1)
model.set_targets(ws, toks.cuda(), labels)
model.pack_weights()
for _ in range(3):
    model.forward_ws(ws, feats, True, True)
torch.cuda.synchronize()
assert getattr(ws, "_sweep_ok", False) and not ops.decoder_sweep_error(ws.dsweep_ws)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    model._decoder_sweep(ws, True)
e1.record()
torch.cuda.synchronize()
print(f"decoder_sweep_fwd: {e0.elapsed_time(e1) / 5 * 1e3 / U:.2f} us per decoder step (traced instance)")
n = 2 * 128 * 16
buf = (C.c_ulonglong * n)()
assert ops.lib().asr_debug_decoder_trace(buf, n) == 0
t = np.array(buf[:], dtype=np.float64).reshape(2, 128, 16)[:, 4:U - 1] * 0.01     # us; skip the first steps
gnames = ["step entered", "h1 gathered (+L0 state product)", "chunk scores visible (LDS hand-over)", "partial ctx handed to publish wave",
          "row's partials gathered", "slice combined", "cell operands gathered", "partial sums in LDS"]
pnames = ["partial published", "slice published", "cell state published", "saved tensors stored"]
for wg, role in ((0, "block 0: attention chunk + LAYER-0 cell"), (1, "block 128: attention chunk + LAYER-1 cell")):
    x = t[wg]
    print(f"---- {role}: step period {np.diff(x[:, 0]).mean():.2f} us")
    for k in range(1, 8):
        d = x[:, k] - x[:, k - 1]
        print(f"  gather wave  {gnames[k - 1]:40s} -> {gnames[k]:40s} {d.mean():6.2f} us (median {np.median(d):.2f})")
    print(f"  gather wave  {gnames[7]:40s} -> next step entered {'':22s} {(x[1:, 0] - x[:-1, 7]).mean():6.2f} us")
    print(f"  publish wave {'partial handed over (stamp 3)':40s} -> {pnames[0]:40s} {(x[:, 8] - x[:, 3]).mean():6.2f} us")
    print(f"  publish wave {'slice combined (stamp 5)':40s} -> {pnames[1]:40s} {(x[:, 9] - x[:, 5]).mean():6.2f} us")
    print(f"  publish wave {'partial sums in LDS (stamp 7)':40s} -> {pnames[2]:40s} {(x[:, 10] - x[:, 7]).mean():6.2f} us")
    print(f"  publish wave {pnames[2]:40s} -> {pnames[3]:40s} {(x[:, 11] - x[:, 10]).mean():6.2f} us")
# cross-workgroup: the step's chain h1 -> A -> S -> L0 -> L1
a0, a1 = t[0], t[1]
print("---- chain across the two workgroups (clocks are the device-wide real-time counter):")
print(f"  L1 cell published (blk 128, step i-1) -> h1 gathered (blk 0, step i)      {(a0[1:, 1] - a1[:-1, 10]).mean():6.2f} us   [hand-off 1: h1]")
print(f"  partial published (blk 0)             -> partials gathered (blk 0)        {(a0[:, 4] - a0[:, 8]).mean():6.2f} us   [hand-off 2: partials, incl. the slowest of 8 chunks]")
print(f"  slice published (blk 0)               -> ctx gathered by L0 cell (blk 0)  {(a0[:, 6] - a0[:, 9]).mean():6.2f} us   [hand-off 3: context]")
print(f"  L0 cell published (blk 0)             -> h0 gathered by L1 cell (blk 128) {(a1[:, 6] - a0[:, 10]).mean():6.2f} us   [hand-off 4: h0]")
