"""Which bf16 images does a mixed-precision training step make?  Logs every ops.f32_to_bf16_image / f32_to_bf16_image_tb call of one eager
step of a bench workload (default las_large): source shape, transposed or straight, and the bytes moved."""
import collections, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
from speech_recognition_amd import ops

wname = sys.argv[1] if len(sys.argv) > 1 else "las_large"
wl = bench.WORKLOADS[wname]
ops.set_mixed_precision(wl.get("precision", "f32") == "bf16")
audio, n, toks = bench.synthetic_batch(0, wl)
audio_d, n_d, toks_d = torch.from_numpy(audio).cuda(), torch.from_numpy(n).cuda(), torch.from_numpy(toks).cuda()
trainer, model = bench.build_trainer(wl, None, use_graph=False)
trainer.step(audio_d, n_d, toks_d, use_teacher_forcing=True)
torch.cuda.synchronize()
calls = collections.OrderedDict()
o1, o2, o3 = ops.f32_to_bf16_image, ops.f32_to_bf16_image_tb, ops.gemm_bf16_nt
def r1(src, dst, transpose=False, **kw):
    k = ("image_T" if transpose else "image", tuple(src.shape), kw.get("scale") is not None)
    calls[k] = calls.get(k, 0) + 1
    return o1(src, dst, transpose=transpose, **kw)
def r2(src, dst, **kw):
    k = ("image_tb", tuple(src.shape), kw.get("scale") is not None)
    calls[k] = calls.get(k, 0) + 1
    return o2(src, dst, **kw)
def r3(a, b, c, **kw):
    k = ("gemm16", (a.shape[0], b.shape[0], int(kw.get("K") or a.shape[1])), int(kw.get("accumulate", 0)))
    calls[k] = calls.get(k, 0) + 1
    return o3(a, b, c, **kw)
ops.f32_to_bf16_image, ops.f32_to_bf16_image_tb, ops.gemm_bf16_nt = r1, r2, r3
import speech_recognition_amd.layers as L
trainer.step(audio_d, n_d, toks_d, use_teacher_forcing=True)
torch.cuda.synchronize()
for k, v in calls.items():
    print(f"{v:4d} x {k}")
