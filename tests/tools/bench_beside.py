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


# ---------------------------------------------------------------------------------------------- CU-masked streams
# The sweep on a stream confined to one set of compute units, the GEMM on a stream confined to the complement: no CU is shared, what
# is left of the interference is L2 / fabric / HBM.  Masks are tried two ways because the bit -> (XCD, CU) map is the runtime's business.
import ctypes

_hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(words):
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = _hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), arr)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return torch.cuda.ExternalStream(s.value)


MASKS = {"low/high halves": ([0xFFFFFFFF] * 4 + [0] * 4, [0] * 4 + [0xFFFFFFFF] * 4),
         "even/odd bits": ([0x55555555] * 8, [0xAAAAAAAA] * 8),
         "even/odd bytes": ([0x00FF00FF] * 8, [0xFF00FF00] * 8),
         "192 / 64": ([0xFFFFFFFF] * 6 + [0] * 2, [0] * 6 + [0xFFFFFFFF] * 2)}
for mname, (ma, mb) in MASKS.items():
    sa, sb = masked_stream(ma), masked_stream(mb)
    torch.cuda.synchronize()
    t_sw = timed(sweep, sa, 3)
    print(f"masks {mname}: sweep alone on its CUs {t_sw:.1f} us", flush=True)
    for name, gemm in GEMMS.items():
        t_g = timed(gemm, sb)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(sa):
            b0.record(sa)
            sweep()
            b1.record(sa)
        with torch.cuda.stream(sb):
            ops.sweep_gate(ops.sweep_diag_words(pws), 300)
            ev0.record(sb)
            for _ in range(6):
                gemm()
            ev1.record(sb)
        torch.cuda.synchronize()
        print(f"   {name}: alone on its CUs {t_g:.1f} us; 6 beside the sweep {ev0.elapsed_time(ev1) * 1e3 / 6:.1f} us each, sweep {b0.elapsed_time(b1) * 1e3:.1f} us",
              flush=True)
    diag = ops.sweep_diagnosis(pws, "rnn_sweep_bwd", clear=True) if hasattr(ops, "sweep_diagnosis") else None
    print(f"   diagnosis: {diag}", flush=True)
