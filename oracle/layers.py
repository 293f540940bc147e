"""Oracle (TEST INFRASTRUCTURE): Keras layer semantics used by the reference models, in
torch-CPU (float64 unless the inputs say otherwise; differentiable, so torch.autograd on
this file yields the reference gradients).

[TF-sem] = TensorFlow/Keras 2.4-2.5 behaviour that the reference relies on but that is not
in its tree (TF is not installed here).  Reference call sites are cited per function.
"""
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import rng


def _t(x, like=None, dtype=None):
    if isinstance(x, torch.Tensor):
        return x
    return torch.as_tensor(np.asarray(x), dtype=dtype or (like.dtype if like is not None else torch.float64))


def dropout_mult(seed, stream, shape, rate, dtype=torch.float64):
    """Inverted-dropout multiplier from the build's stateless RNG (oracle/rng.py)."""
    return torch.as_tensor(rng.dropout_mask(seed, stream, tuple(shape), rate), dtype=dtype)


# --------------------------------------------------------------------------------------
# Conv2D  (las.py:163-164, deepspeech2.py:47-50)  [TF-sem] padding VALID, NHWC, HWIO, linear
# --------------------------------------------------------------------------------------
def conv2d_nhwc(x, kernel, bias, strides):
    sh, sw = (strides, strides) if isinstance(strides, int) else strides
    B, H, W, C = x.shape
    kh, kw, _, O = kernel.shape
    Ho, Wo = (H - kh) // sh + 1, (W - kw) // sw + 1
    out = torch.zeros(B, Ho, Wo, O, dtype=x.dtype)
    for r in range(kh):
        for s in range(kw):
            patch = x[:, r: r + sh * (Ho - 1) + 1: sh, s: s + sw * (Wo - 1) + 1: sw, :]
            out = out + patch @ kernel[r, s]
    return out + bias


# --------------------------------------------------------------------------------------
# Recurrent cells and K.rnn masking  (las.py:62-126)  [TF-sem]
# --------------------------------------------------------------------------------------
def lstm_cell(x, h, c, W, U, b):
    """Keras LSTMCell: gates i,f,c~,o; c' = f c + i tanh(.); h' = o tanh(c')."""
    z = x @ W + h @ U + b
    H = h.shape[-1]
    i, f, g, o = z[:, :H], z[:, H:2 * H], z[:, 2 * H:3 * H], z[:, 3 * H:]
    i, f, o = torch.sigmoid(i), torch.sigmoid(f), torch.sigmoid(o)
    c2 = f * c + i * torch.tanh(g)
    h2 = o * torch.tanh(c2)
    return h2, [h2, c2]


def gru_cell(x, h, W, U, b):
    """Keras GRUCell(reset_after=True): bias [2, 3H] = (input bias, recurrent bias); order z,r,h~."""
    H = h.shape[-1]
    mx = x @ W + b[0]
    mh = h @ U + b[1]
    z = torch.sigmoid(mx[:, :H] + mh[:, :H])
    r = torch.sigmoid(mx[:, H:2 * H] + mh[:, H:2 * H])
    hh = torch.tanh(mx[:, 2 * H:] + r * mh[:, 2 * H:])
    h2 = z * h + (1.0 - z) * hh
    return h2, [h2]


def lstm_cell_impl1(x, h, c, W, U, b, im, rm):
    """Keras LSTMCell implementation 1 - what tf.keras uses whenever recurrent_dropout != 0 ([TF-sem] LSTMCell.__init__ forces it): one
    dropout mask PER GATE on the input (im[g], or None) and on h_tm1 (rm[g], or None); each gate multiplies its own masked operands."""
    H = h.shape[-1]
    zs = []
    for g in range(4):
        xg = x if im is None else x * im[g]
        hg = h if rm is None else h * rm[g]
        zs.append(xg @ W[:, g * H:(g + 1) * H] + hg @ U[:, g * H:(g + 1) * H] + b[g * H:(g + 1) * H])
    i, f, o = torch.sigmoid(zs[0]), torch.sigmoid(zs[1]), torch.sigmoid(zs[3])
    c2 = f * c + i * torch.tanh(zs[2])
    h2 = o * torch.tanh(c2)
    return h2, [h2, c2]


def gru_cell_impl1(x, h, W, U, b, im, rm):
    """Keras GRUCell(reset_after=True) implementation 1 (forced by recurrent_dropout != 0): masks per gate z, r, h; the carry
    z * h_tm1 uses the UNMASKED h_tm1."""
    H = h.shape[-1]
    mx, mh = [], []
    for g in range(3):
        xg = x if im is None else x * im[g]
        hg = h if rm is None else h * rm[g]
        mx.append(xg @ W[:, g * H:(g + 1) * H] + b[0][g * H:(g + 1) * H])
        mh.append(hg @ U[:, g * H:(g + 1) * H] + b[1][g * H:(g + 1) * H])
    z = torch.sigmoid(mx[0] + mh[0])
    r = torch.sigmoid(mx[1] + mh[1])
    hh = torch.tanh(mx[2] + r * mh[2])
    h2 = z * h + (1.0 - z) * hh
    return h2, [h2]


def simple_rnn_cell(x, h, W, U, b):
    h2 = torch.tanh(x @ W + b + h @ U)
    return h2, [h2]


def num_states(rnn_type):
    return 2 if rnn_type == "lstm" else 1


def rnn_layer(rnn_type, x, mask, W, U, b, initial_state=None, go_backwards=False,
              in_mult=None, rec_mult=None):
    """Keras RNN layer with return_sequences=True, return_state=True and a mask.

    [TF-sem] K.rnn: at a masked step the states are carried unchanged and the emitted output
    is the previous emitted output (zeros before the first unmasked step).  go_backwards
    consumes input and mask reversed in time; outputs stay in *processing* order (the caller,
    BiRNN, re-reverses them: las.py:125).  Dropout multipliers in_mult/rec_mult are [B, D] /
    [B, H], constant over time (Keras DropoutRNNCellMixin), applied to the input / to h.  A LIST of multipliers (one per gate:
    4 for LSTM, 3 for GRU) selects Keras' implementation 1, which LSTM / GRU cells switch to whenever recurrent_dropout != 0
    (per-gate masks on both operands, GRU carry on the unmasked state); a single tensor is implementation 2 (input dropout alone).
    Returns (outputs [B,T,H], [states...])."""
    if rnn_type not in ("rnn", "lstm", "gru"):
        raise ValueError(f"rnn_type: {rnn_type} is invalid!")
    B, T, _ = x.shape
    H = U.shape[0]
    if initial_state is None:
        initial_state = [torch.zeros(B, H, dtype=x.dtype) for _ in range(num_states(rnn_type))]
    states = list(initial_state)
    mask = mask.to(torch.bool)
    order = range(T - 1, -1, -1) if go_backwards else range(T)
    prev_out = torch.zeros(B, H, dtype=x.dtype)
    outs = []
    impl1 = isinstance(in_mult, (list, tuple)) or isinstance(rec_mult, (list, tuple))
    if impl1 and rnn_type == "rnn":                     # SimpleRNN has one gate: one mask each
        in_mult = in_mult[0] if isinstance(in_mult, (list, tuple)) else in_mult
        rec_mult = rec_mult[0] if isinstance(rec_mult, (list, tuple)) else rec_mult
        impl1 = False
    for t in order:
        if impl1:
            if rnn_type == "lstm":
                out, new = lstm_cell_impl1(x[:, t], states[0], states[1], W, U, b, in_mult, rec_mult)
            else:
                out, new = gru_cell_impl1(x[:, t], states[0], W, U, b, in_mult, rec_mult)
        else:
            xt = x[:, t] if in_mult is None else x[:, t] * in_mult
            h = states[0] if rec_mult is None else states[0] * rec_mult
            if rnn_type == "lstm":
                out, new = lstm_cell(xt, h, states[1], W, U, b)
            elif rnn_type == "gru":
                out, new = gru_cell(xt, h, W, U, b)
            else:
                out, new = simple_rnn_cell(xt, h, W, U, b)
        m = mask[:, t][:, None]
        out = torch.where(m, out, prev_out)
        states = [torch.where(m, n, s) for n, s in zip(new, states)]
        prev_out = out
        outs.append(out)
    return torch.stack(outs, dim=1), states


def birnn(rnn_type, x, mask, fwd, bwd, initial_state=None, in_mult_f=None, in_mult_b=None,
          rec_mult_f=None, rec_mult_b=None):
    """BiRNN.call (las.py:108-126).  fwd/bwd: (kernel, recurrent_kernel, bias).
    Returns [output [B,T,2H]] + forward_states + backward_states."""
    if initial_state is None:
        fs = bs = None
    else:
        n = len(initial_state) // 2
        fs, bs = list(initial_state[:n]), list(initial_state[n:])
    fo, fstates = rnn_layer(rnn_type, x, mask, *fwd, initial_state=fs, in_mult=in_mult_f, rec_mult=rec_mult_f)
    bo, bstates = rnn_layer(rnn_type, x, mask, *bwd, initial_state=bs, go_backwards=True,
                            in_mult=in_mult_b, rec_mult=rec_mult_b)
    out = torch.cat([fo, torch.flip(bo, dims=[1])], dim=-1)
    return [out] + fstates + bstates


# --------------------------------------------------------------------------------------
# BatchNormalization  (las.py:170,193; deepspeech2.py:112,118)  [TF-sem] axis -1, eps 1e-3,
# momentum 0.99; training: biased batch statistics over every other axis (padding included).
# --------------------------------------------------------------------------------------
def batch_norm(x, gamma, beta, moving_mean, moving_var, training, eps=1e-3, momentum=0.99):
    if training:
        dims = tuple(range(x.dim() - 1))
        mean = x.mean(dim=dims)
        var = ((x - mean) ** 2).mean(dim=dims)
        new_mm = moving_mean * momentum + mean.detach() * (1.0 - momentum)
        new_mv = moving_var * momentum + var.detach() * (1.0 - momentum)
    else:
        mean, var, new_mm, new_mv = moving_mean, moving_var, moving_mean, moving_var
    y = (x - mean) / torch.sqrt(var + eps) * gamma + beta
    return y, new_mm, new_mv


# --------------------------------------------------------------------------------------
# "AdditiveAttention" (las.py:46-59): projected dot-product attention, no tanh / v.
# --------------------------------------------------------------------------------------
def attention(query, key, value, attention_mask, Wq, bq, Wk, bk, return_scores=False):
    q = (query @ Wq + bq)[:, None, :]                       # [B,1,H]
    k = (key @ Wk + bk).transpose(1, 2)                     # [B,H,T]
    w = q @ k                                               # [B,1,T]
    w = w - 1e9 * (1.0 - attention_mask[:, None, :].to(w.dtype))
    p = torch.softmax(w, dim=-1)
    ctx = (p @ value)[:, 0, :]
    if return_scores:                                       # (tests: the masked scores, to read their gradient after backward())
        return ctx, p[:, 0, :], w
    return ctx, p[:, 0, :]
