"""Which dense products does a training step launch, and how far from the MFMA peak is each one alone?  Records every ops.gemm call of
one eager step of a bench workload, then times each distinct call signature by itself (fresh operands of the same shapes / strides).
  python tests/tools/gemm_shapes.py [las_small|deepspeech|las_large]"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench
from speech_recognition_amd import ops

wname = sys.argv[1] if len(sys.argv) > 1 else "las_small"
wl = bench.WORKLOADS[wname]
precision = wl.get("precision", "f32")
ops.set_mixed_precision(precision == "bf16")
audio, n, toks = bench.synthetic_batch(0, wl)
audio_d, n_d, toks_d = torch.from_numpy(audio).cuda(), torch.from_numpy(n).cuda(), torch.from_numpy(toks).cuda()
trainer, model = bench.build_trainer(wl, None, use_graph=False)
trainer.step(audio_d, n_d, toks_d, use_teacher_forcing=True)
torch.cuda.synchronize()

calls = collections.OrderedDict()
orig = ops.gemm


def rec(a, b, c, **kw):
    key = (tuple(a.shape), tuple(a.stride()), tuple(b.shape), tuple(b.stride()), tuple(c.shape), tuple(c.stride()), bool(kw.get("trans_a")), bool(kw.get("trans_b")),
           int(kw.get("accumulate", 0)), int(kw.get("split_k", 1)), kw.get("bias") is not None, kw.get("a_scale") is not None, int(kw.get("a_rpg", 0)),
           kw.get("c_scale") is not None, int(kw.get("c_rpg", 0)), int(kw.get("a_scale_stride", 0)), bool(kw.get("relu")))
    calls.setdefault(key, [0, (a, b, c, kw)])[0] += 1
    return orig(a, b, c, **kw)


ops.gemm = rec
import speech_recognition_amd.layers as L
trainer.step(audio_d, n_d, toks_d, use_teacher_forcing=True)
torch.cuda.synchronize()
ops.gemm = orig
peak = bench.PEAK_BF16_MFMA if precision == "bf16" else bench.PEAK_F32_MFMA
rows = []
for key, (cnt, (a, b, c, kw)) in calls.items():
    ta, tb = key[6], key[7]
    batch = a.shape[0] if a.dim() == 3 else 1
    M, K = (a.shape[-1], a.shape[-2]) if ta else (a.shape[-2], a.shape[-1])
    N = b.shape[-2] if tb else b.shape[-1]
    c2 = c.clone()
    t = bench.time_kernel(torch.cuda.current_stream(), lambda: orig(a, b, c2, **kw), iters=10)
    fl = 2.0 * M * N * K * batch
    rows.append((t * cnt, cnt, t, fl, M, N, K, batch, ta, tb, key[8], key[9], key[11]))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"{wname}: {sum(r[1] for r in rows)} products per step, {tot * 1e3:.3f} ms when each runs alone")
for tt, cnt, t, fl, M, N, K, batch, ta, tb, acc, sk, sc in rows[:40]:
    print(f"  x{cnt:3d} {t * 1e6:8.1f} us {fl / t / 1e12:7.1f} TF ({fl / t / peak:5.2f})  M={M:6d} N={N:6d} K={K:6d} batch={batch:3d} "
          f"{'T' if ta else 'N'}{'T' if tb else 'N'} acc={acc} split_k={sk} a_scale={int(sc)}   total {tt * 1e3:6.3f} ms")
