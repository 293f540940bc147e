"""ADVICE (round 1, low): is a hipMemsetAsync inside a captured graph ordered before the kernel that follows it on the same stream?
Captures  [memset(buf) -> kernel that reads buf]  on the capture stream, with a forked side stream as the training step has, dumps
the graph with hipGraphDebugDotPrint and prints the nodes and edges, then replays it many times next to a second process-like
load (another stream) and counts stale reads.   python tests/tools/dbg_memset_graph.py"""
import ctypes as C
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from speech_recognition_amd import ops

hip = C.CDLL("libamdhip64.so")
n = 1 << 20
buf = torch.ones(n, device="cuda")
out = torch.zeros(n, device="cuda")
side_a, side_b = torch.randn(1 << 22, device="cuda"), torch.zeros(1 << 22, device="cuda")
st = torch.cuda.Stream()
side = torch.cuda.Stream()


def body():
    # buf <- 0 by a memset node, then out += buf + 1 by one of the library's kernels; if the memset is not ordered before it, out sees 1s
    rc = hip.hipMemsetAsync(C.c_void_p(buf.data_ptr()), C.c_int(0), C.c_size_t(n * 4), C.c_void_p(st.cuda_stream))   # a MEMSET node
    assert rc == 0, rc
    ev = torch.cuda.Event()
    ev.record(st)
    with torch.cuda.stream(side):
        side.wait_event(ev)
        side_b.copy_(side_a)                      # independent work on a forked stream (a parallel branch of the graph)
        ev2 = torch.cuda.Event()
        ev2.record(side)
    out.add_(buf)                                 # reads buf
    buf.fill_(1.0)                                # makes the next replay's stale read visible
    st.wait_event(ev2)


with torch.cuda.stream(st):
    body()
    torch.cuda.synchronize()
    try:
        g = torch.cuda.CUDAGraph(keep_graph=True)
    except TypeError:
        g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
        body()
    raw = None
    if hasattr(g, "raw_cuda_graph"):
        try:
            raw = g.raw_cuda_graph()
        except Exception as e:  # noqa: BLE001
            print("raw_cuda_graph():", e)
    if raw:
        path = os.path.join(ROOT, "gpurun_out", "memset_graph.dot")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        rc = hip.hipGraphDebugDotPrint(C.c_void_p(raw), path.encode(), C.c_uint(1))
        print("hipGraphDebugDotPrint rc", rc)
        if rc == 0:
            txt = open(path).read()
            nodes = re.findall(r'"?(\w+)"?\s*\[.*?label="([^"]*)"', txt, re.S)
            edges = re.findall(r'"?(\w+)"?\s*->\s*"?(\w+)"?', txt)
            for k, lab in nodes:
                print("node", k, lab.replace("\\n", " | ")[:110])
            print("edges", edges)
    if hasattr(g, "instantiate") and raw:
        g.instantiate()
    out.zero_()
    buf.fill_(1.0)
    torch.cuda.synchronize()
    reps = 2000
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    stale = float(out.max())
    print(f"{reps} replays: out max = {stale} (0 = the memset always ran before the reader; >0 = stale reads)")
