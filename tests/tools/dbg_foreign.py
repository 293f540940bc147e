"""Debug aid for tests/test_trainstep_gpu.py::test_sweeps_survive_foreign_kernels_holding_compute_units: repeats the test's scenario
(two trainers built first; one runs two steps beside a foreign kernel that holds 192 compute units, the other quietly) and reports
every gradient tensor whose entries differ by more than 1e-5 of the tensor's largest entry."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from speech_recognition_amd import ops
from tests.test_trainstep_gpu import _las_batch, _las_trainer

print("ASR_CONCURRENT", os.environ.get("ASR_CONCURRENT"), "mode", ops.f32_gemm_mode())
batch = _las_batch(B=32, T=126, L=6, seed=2)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for rep in range(reps):
    tr, model = _las_trainer(He=64, seed=5, use_graph=False)
    tr2, model2 = _las_trainer(He=64, seed=5, use_graph=False)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        ops.debug_occupy(192, 1024, 400000)
    ga, gb, la, lb = [], [], [], []
    for _ in range(2):
        ws = tr.step(*batch, use_teacher_forcing=True)
        la.append(tr.read_stats(ws)[0])
        ga.append(model.store.grads())
    side.synchronize()
    for _ in range(2):
        ws2 = tr2.step(*batch, use_teacher_forcing=True)
        lb.append(tr2.read_stats(ws2)[0])
        gb.append(model2.store.grads())
    bad = []
    for step in range(2):
        for k in ga[step]:
            d = float((ga[step][k] - gb[step][k]).abs().max())
            s = float(gb[step][k].abs().max())
            if s > 1e-6 and d > 1e-5 * s:
                idx = int((ga[step][k] - gb[step][k]).abs().flatten().argmax())
                bad.append(f"step {step} {k}: max diff {d:.3e} of {s:.3e} at flat index {idx} ({ga[step][k].flatten()[idx]:.4e} vs {gb[step][k].flatten()[idx]:.4e}), "
                           f"{int(((ga[step][k] - gb[step][k]).abs() > 1e-5 * s).sum())} of {ga[step][k].numel()} entries")
    pd = float((model.store.flat - model2.store.flat).abs().max())
    print(f"rep {rep}: losses {la} vs {lb}; parameters differ by at most {pd:.3e}; gradient tensors beyond 1e-5: {len(bad)}")
    for b in bad:
        print("    ", b)
    del tr, tr2, model, model2
    torch.cuda.empty_cache()
