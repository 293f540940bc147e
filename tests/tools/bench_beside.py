"""How fast does a GEMM run BESIDE a resident sweep?  Launches the encoder BPTT sweep (las_small layer) or a plain occupier kernel
on one stream and, gated on its start, a weight-gradient-shaped GEMM on another; prints the GEMM's duration alone and beside.
python tests/tools/bench_beside.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from speech_recognition_amd import ops
from tests.rnn_helpers import HipBiRNN
from tests.test_rnn_gpu import make_params

rt, B, T, D, H = "lstm", 32, 249, 512, 256
g = torch.Generator().manual_seed(1)
fwd, bwd = make_params(rt, D, H, g, 0.08)
x = torch.randn(B, T, D, generator=g, dtype=torch.float64)
hip = HipBiRNN(rt, x, None, fwd, bwd, None)
hip.forward(persistent=True)
dy = torch.randn(B, T, 2 * H, generator=g).cuda()
gds = [dict(direct=torch.zeros(B, H, device="cuda"), dy_carry=torch.zeros(B, H, device="cuda"), dh0=torch.zeros(B, H, device="cuda"),
            dc=torch.zeros(B, H, device="cuda"), ds=torch.empty_like(dd["saved"])) for dd in hip.dirs]
pws = ops.rnn_persist_bwd_ws(B, H, 2)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

M, K, N = 512, B * T, 1024
xa = torch.randn(K, M, device="cuda")
dsb = torch.randn(K, N, device="cuda")
gw = torch.zeros(M, N, device="cuda")
a2 = torch.randn(K, 512, device="cuda")
w2 = torch.randn(512, 1024, device="cuda")
y2 = torch.empty(K, 1024, device="cuda")
GEMMS = {"dW TA=1 split16 [512x7968]x[7968x1024]": lambda: ops.gemm(xa, dsb, gw, trans_a=True, accumulate=1, split_k=16),
         "fwd [7968x512]x[512x1024]": lambda: ops.gemm(a2, w2, y2)}


def sweep():
    for gd in gds:
        gd["dc"].zero_()
    ops.rnn_seq_bwd(hip.seq, dy, gds, pws)


def occupy():
    ops.debug_occupy(256, 384, 1000)


def timed(fn, stream, n=5):
    with torch.cuda.stream(stream):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(n):
            fn()
        e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for name, gemm in GEMMS.items():
    alone = timed(gemm, s2)
    print(f"{name}: alone {alone:.1f} us", flush=True)
    for bname, bg in (("BPTT sweep", sweep), ("occupier 256 x 384 threads", occupy)):
        torch.cuda.synchronize()
        t_bg = timed(bg, s1, 3)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(s1):
            b0.record(s1)
            bg()
            b1.record(s1)
        with torch.cuda.stream(s2):
            if bg is sweep:
                ops.sweep_gate(ops.sweep_diag_words(pws), 300)
            else:
                torch.cuda._sleep(20000)
            ev0.record(s2)
            gemm()
            ev1.record(s2)
        torch.cuda.synchronize()
        print(f"   beside {bname}: gemm {ev0.elapsed_time(ev1) * 1e3:.1f} us; background {b0.elapsed_time(b1) * 1e3:.1f} us (alone {t_bg:.1f})", flush=True)
