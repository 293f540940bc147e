"""Run one eager step of a bench workload and print the error words of every one-launch sweep workspace."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench
from speech_recognition_amd import ops

name = sys.argv[1] if len(sys.argv) > 1 else "deepspeech"
wl = bench.WORKLOADS[name]
trainer, model = bench.build_trainer(wl, None, use_graph=len(sys.argv) > 2 and sys.argv[2] == "graph")
audio, n, toks = bench.synthetic_batch(0, wl)
a, n_d, t = torch.from_numpy(audio).cuda(), torch.from_numpy(n).cuda(), torch.from_numpy(toks).cuda()
for step in range(int(sys.argv[3]) if len(sys.argv) > 3 else 3):
    ws = trainer.step(a, n_d, t, use_teacher_forcing=True)
    if os.environ.get("NOSYNC") and step % 35 != 34:
        continue
    torch.cuda.synchronize()
    for i, lw in enumerate(ws.layers):
        b = lw["rnn"]
        for k in ("persist_ws", "persist_bwd_ws"):
            if k in b:
                e = ops.rnn_persist_error(b[k])
                if e:
                    print(f"step {step} layer {i} {k}: error word {e:#x} (code {e & 255}, step {e >> 8})  B={b['B']} T={b['T']} mask={b['mask'] is not None}")
    st = model.state.cpu().tolist()
    if st[2] or step % 10 == 0:
        print("step", step, "state", st, "err_flag", float(model.store.err_flag))
