"""Cost of one dependent kernel boundary inside a replayed hipGraph (not a test): a chain of N tiny fill kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speech_recognition_amd import ops
x = torch.zeros(64, device="cuda")
s = torch.cuda.Stream()
N = 500
with torch.cuda.stream(s):
    for _ in range(3): ops.fill(x, 1.0)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for i in range(N): ops.fill(x, float(i))
    for _ in range(3): g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(10): g.replay()
    e1.record(s)
e1.synchronize()
print(f"graph chain of {N} tiny kernels: {e0.elapsed_time(e1) / 10 / N * 1e3:.2f} us per kernel")
with torch.cuda.stream(s):
    e0.record(s)
    for _ in range(10):
        for i in range(N): ops.fill(x, float(i))
    e1.record(s)
e1.synchronize()
print(f"eager chain: {e0.elapsed_time(e1) / 10 / N * 1e3:.2f} us per kernel")
