"""Flat parameter storage and Keras-default initialisation.

All trainable variables of a model live in ONE contiguous fp32 device buffer (same for gradients
and the two Adam moments), each variable 16-byte aligned inside it.  That makes the optimizer a
single HBM-bound kernel and the data-parallel exchange a couple of large RCCL all-reduces on views
of the gradient buffer (no per-tensor launches, no packing copies).  Variable names are the TF
checkpoint keys of the reference models (e.g. ``listener/conv1/kernel``) and shapes are the Keras
layouts, so a tensor-bundle reader is a straight copy.
"""
import math
from collections import OrderedDict
from typing import Dict, Iterable, List, Tuple

import torch


def _align(n, a=4):
    return (n + a - 1) // a * a


class ParamStore:
    def __init__(self, shapes: "OrderedDict[str, tuple]", buckets: List[List[str]] = None, device="cuda"):
        """shapes: ordered name -> shape of the TRAINABLE variables.  buckets: optional partition of
        the names into contiguous all-reduce buckets (order = order gradients become ready)."""
        if buckets is None:
            buckets = [list(shapes)]
        order = [n for b in buckets for n in b]
        assert sorted(order) == sorted(shapes), "buckets must cover every trainable variable exactly once"
        self.shapes = OrderedDict((n, tuple(shapes[n])) for n in order)
        self.offsets: Dict[str, int] = {}
        off = 0
        self.bucket_ranges: List[Tuple[int, int]] = []
        for b in buckets:
            start = off
            for n in b:
                self.offsets[n] = off
                off += _align(math.prod(self.shapes[n]))
            self.bucket_ranges.append((start, off))
        # one extra cell at the end of the LAST bucket: the step's error flag.  The one-launch recurrent sweeps write 1.0 into
        # the gradient buffer's copy when a hand-off times out; it is zeroed with the gradients, summed across ranks with the last
        # bucket (so every replica sees it), and Adam / advance_state skip the step when it is non-zero.
        self.err_off = off
        off += 4
        self.bucket_ranges[-1] = (self.bucket_ranges[-1][0], off)
        self.numel = off
        self.device = device
        self.flat = torch.zeros(off, device=device, dtype=torch.float32)
        self.grad = torch.zeros(off, device=device, dtype=torch.float32)
        self.adam_m = torch.zeros(off, device=device, dtype=torch.float32)
        self.adam_v = torch.zeros(off, device=device, dtype=torch.float32)
        self.err_flag = self.grad[self.err_off:self.err_off + 1]
        self.p = {n: self._view(self.flat, n) for n in self.shapes}
        self.g = {n: self._view(self.grad, n) for n in self.shapes}
        self.flat16 = None          # bf16 image of `flat` (mixed precision), refreshed once per optimizer step

    def refresh_bf16(self):
        """Bring the bf16 image of the parameters up to date (one pass over the flat buffer) and make it known to ops."""
        from . import ops
        if self.flat16 is None:
            self.flat16 = torch.empty(self.numel, device=self.flat.device, dtype=torch.bfloat16)
            ops.register_bf16_mirror(self.flat, self.flat16)
        ops.f32_to_bf16(self.flat, self.flat16)

    def _view(self, buf, n):
        o = self.offsets[n]
        return buf[o:o + math.prod(self.shapes[n])].view(self.shapes[n])

    def num_trainable(self):
        return sum(math.prod(s) for s in self.shapes.values())

    def load(self, values: Dict[str, torch.Tensor]):
        for n, v in values.items():
            if n in self.p:
                self.p[n].copy_(torch.as_tensor(v).to(torch.float32).reshape(self.shapes[n]))

    def state_dict(self):
        return {n: v.detach().cpu().clone() for n, v in self.p.items()}

    def grads(self):
        return {n: v.detach().cpu().clone() for n, v in self.g.items()}

    def bucket_views(self, buf=None):
        buf = self.grad if buf is None else buf
        return [buf[a:b] for a, b in self.bucket_ranges]


# ------------------------------------------------------------------------------------------------
# Keras default initialisers ([TF-sem]): Dense/Conv kernel glorot_uniform, bias zeros; RNN kernel
# glorot_uniform, recurrent_kernel orthogonal, bias zeros with LSTM unit_forget_bias; Embedding
# uniform(-0.05, 0.05); BatchNormalization gamma ones, beta zeros.
# ------------------------------------------------------------------------------------------------
def glorot_uniform(shape, gen):
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:  # conv HWIO
        rf = math.prod(shape[:-2])
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(shape, generator=gen) * 2 - 1) * lim


def orthogonal(shape, gen):
    rows, cols = shape
    a = torch.randn(max(rows, cols), min(rows, cols), generator=gen)
    q, r = torch.linalg.qr(a)
    q = q * torch.sign(torch.diagonal(r))
    if rows < cols:
        q = q.T
    return q[:rows, :cols].contiguous()


def init_value(name: str, shape, gen, rnn_type="lstm"):
    leaf = name.split("/")[-1]
    if leaf in ("bias", "beta", "moving_mean"):
        v = torch.zeros(shape)
        if leaf == "bias" and "/cell/" in name and rnn_type == "lstm":
            H = shape[-1] // 4
            v[..., H:2 * H] = 1.0  # unit_forget_bias
        return v
    if leaf in ("gamma", "moving_variance"):
        return torch.ones(shape)
    if leaf == "embeddings":
        return torch.rand(shape, generator=gen) * 0.1 - 0.05
    if leaf == "recurrent_kernel":
        return orthogonal(shape, gen)
    return glorot_uniform(shape, gen)
