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
# --mixed-precision (run/train.py:62-66) as the MI355X build implements it: bf16 OPERANDS.  The reference switches the Keras
# policy to mixed_float16; the build (north_star: bf16) keeps every tensor and every accumulation in f32 and rounds the two
# operands of its contractions to bf16 (round to nearest even) on the way into the matrix cores.  `bf16_operands()` makes this
# restatement round at exactly the same places, forward AND backward, so that mixed-precision runs can be checked to rounding
# noise instead of to "bf16-sized" tolerances:
#   * every dense contraction (ops.gemm: input projections of the recurrent layers, Dense layers, attention keys, vocabulary
#     layer) rounds both operands, and so do the products of its backward pass (dX = dY W^T, dW = X^T dY);
#   * a recurrent cell's in-kernel products (h U, and for decoder cells the input product) are rounded only by the WIDE kernels
#     (H >= wide_h, more than 16 batch rows); the narrow cells multiply in exact f32 - but their weight gradients are dense
#     products after the sweep and round (dU = h^T ds, dW = x^T ds), while dh = ds U^T inside the sweep does not;
#   * with Hd >= attn_hd the decoder's attention steps read bf16 images of Kq = (enc Wk + bk) Wq^T and of enc (the build hoists
#     the key projection out of the step loop - the same sums as las.py:46-59 in a different association, which bf16_operands()
#     therefore adopts; with rounding switched off it equals the reference form to 1e-12, tests/test_oracle.py).
# Convolutions, gate math, softmax, CTC, batch norm and the optimizer are f32 in the build: untouched here.
# --------------------------------------------------------------------------------------
class _Bf16:
    on = False
    rounding = True          # False: the mode's algebra (hoisted attention) without any rounding - for the CPU cross-check
    wide_h = 512
    attn_hd = 512


class bf16_operands:
    """Context manager: the restatement rounds contraction operands to bf16 where the build's kernels do (see above)."""

    def __init__(self, wide_h=512, attn_hd=512, rounding=True):
        self.new = (True, rounding, wide_h, attn_hd)

    def __enter__(self):
        self.old = (_Bf16.on, _Bf16.rounding, _Bf16.wide_h, _Bf16.attn_hd)
        _Bf16.on, _Bf16.rounding, _Bf16.wide_h, _Bf16.attn_hd = self.new
        return self

    def __exit__(self, *exc):
        _Bf16.on, _Bf16.rounding, _Bf16.wide_h, _Bf16.attn_hd = self.old
        return False


def bf16_round(t, rounding=None):
    """Round to bf16 (nearest even) and back to t's dtype (no gradient: callers route gradients themselves).  `rounding`: the
    mode's switch as it stood when the forward product ran (backward passes run outside the `with` block)."""
    if not (_Bf16.rounding if rounding is None else rounding):
        return t.detach()
    return t.detach().to(torch.float32).to(torch.bfloat16).to(t.dtype)


class _MM(torch.autograd.Function):
    """c = a @ b over the last axis of a ([..., K] @ [K, N]) with per-product rounding flags: rf - round both operands in the
    forward product; rda - in dA = dC B^T; rdb - in dB = A^T dC."""

    @staticmethod
    def forward(ctx, a, b, rf, rda, rdb):
        ctx.save_for_backward(a, b)
        ctx.flags = (rda, rdb, _Bf16.rounding)
        return (bf16_round(a) @ bf16_round(b)) if rf else (a.detach() @ b.detach())

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        rda, rdb, rnd = ctx.flags
        da = db = None
        if ctx.needs_input_grad[0]:
            da = (bf16_round(dc, rnd) @ bf16_round(b, rnd).t()) if rda else (dc @ b.t())
        if ctx.needs_input_grad[1]:
            a2, d2 = a.reshape(-1, a.shape[-1]), dc.reshape(-1, dc.shape[-1])
            db = (bf16_round(a2, rnd).t() @ bf16_round(d2, rnd)) if rdb else (a2.t() @ d2)
        return da, db, None, None, None


def mm_dense(a, b):
    """A product the build runs as a dense GEMM (ops.gemm): both operands rounded, forward and backward, under bf16_operands()."""
    return _MM.apply(a, b, True, True, True) if _Bf16.on else a @ b


def mm_cell(a, b, H):
    """A product inside a recurrent cell kernel (state x recurrent kernel; decoder input x kernel): rounded like a dense product
    by the wide kernels (H >= wide_h and more than 16 rows); exact in the narrow ones, whose WEIGHT gradient alone is a dense
    (rounded) product after the loop."""
    if not _Bf16.on:
        return a @ b
    wide = H >= _Bf16.wide_h and a.shape[0] > 16
    return _MM.apply(a, b, wide, wide, True)


def _xw(x, W, x_mode, H):
    """Input product of a cell.  x_mode: "dense" - batched over time outside the cell kernels (encoder layers); "cell" - inside
    the step kernel (decoder layer >= 1); ("split", k) - the first k input features through a dense product, the rest inside the
    step kernel (decoder layer 0: the embedding half is batched over all steps under teacher forcing)."""
    if not _Bf16.on:
        return x @ W
    if x_mode == "dense":
        return mm_dense(x, W)
    if x_mode == "cell":
        return mm_cell(x, W, H)
    k = x_mode[1]
    return mm_dense(x[:, :k], W[:k]) + mm_cell(x[:, k:], W[k:], H)


class _Scores(torch.autograd.Function):
    """scores[b, t] = h[b] . Kq[b, t] of one decoder step (attention.hip / decoder_sweep.hip): h in f32; Kq from its bf16 image when
    `img`.  Backward as the build does it: dh = de Kq (f32 de, the same Kq), dKq = de^T h as a dense product (both rounded)."""

    @staticmethod
    def forward(ctx, h, Kq, img):
        Kq_ = bf16_round(Kq) if img else Kq.detach()
        ctx.save_for_backward(h, Kq_)
        ctx.rnd = _Bf16.rounding
        return torch.einsum("bh,bth->bt", h.detach(), Kq_)

    @staticmethod
    def backward(ctx, de):
        h, Kq_ = ctx.saved_tensors
        dh = torch.einsum("bt,bth->bh", de, Kq_)
        dKq = torch.einsum("bt,bh->bth", bf16_round(de, ctx.rnd), bf16_round(h, ctx.rnd))
        return dh, dKq, None


class _Context(torch.autograd.Function):
    """ctx[b] = p[b] . enc[b] of one decoder step: p in f32, enc from its bf16 image when `img`; backward: dp = dctx enc^T (f32 dctx),
    d enc = p^T dctx as a dense product (both rounded)."""

    @staticmethod
    def forward(ctx, p, enc, img):
        enc_ = bf16_round(enc) if img else enc.detach()
        ctx.save_for_backward(p, enc_)
        ctx.rnd = _Bf16.rounding
        return torch.einsum("bt,btd->bd", p.detach(), enc_)

    @staticmethod
    def backward(ctx, dctx):
        p, enc_ = ctx.saved_tensors
        dp = torch.einsum("bd,btd->bt", dctx, enc_)
        denc = torch.einsum("bt,bd->btd", bf16_round(p, ctx.rnd), bf16_round(dctx, ctx.rnd))
        return dp, denc, None


def attention_keys_hoisted(key, Wq, bq, Wk, bk):
    """The step-invariant part of las.py:46-54 as the build computes it once per batch: K = key Wk + bk, Kq = K Wq^T, s0 = K bq
    (so that scores = h Kq^T + s0 = (h Wq + bq) K^T).  Dense products; s0 is an exact f32 row dot in the build."""
    K = mm_dense(key, Wk) + bk
    Kq = mm_dense(K, Wq.t())
    s0 = K @ bq
    return Kq, s0


def attention_hoisted(query, keys, value, attention_mask, return_scores=False):
    """One step of the hoisted attention: keys = attention_keys_hoisted(...)."""
    Kq, s0 = keys
    img = Kq.shape[-1] >= _Bf16.attn_hd
    w = _Scores.apply(query, Kq, img) + s0
    w = (w - 1e9 * (1.0 - attention_mask.to(w.dtype)))[:, None, :]
    p = torch.softmax(w, dim=-1)
    ctx = _Context.apply(p[:, 0, :], value, img)
    if return_scores:
        return ctx, p[:, 0, :], w
    return ctx, p[:, 0, :]


# --------------------------------------------------------------------------------------
# Conv2D  (las.py:163-164, deepspeech2.py:47-50)  [TF-sem] padding VALID, NHWC, HWIO, linear
# --------------------------------------------------------------------------------------
def conv2d_nhwc_taps(x, kernel, bias, strides):
    """The definition, tap by tap (kept as the cross-check of conv2d_nhwc in tests/test_oracle.py)."""
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


class _Conv2dIm2col(torch.autograd.Function):
    """The same sums as conv2d_nhwc_taps, gathered per clip into one [Ho Wo, kh kw C] matrix (a strided window view of the clip,
    copied once) and multiplied in ONE product, with the gradients written out by hand (filter: cols^T dy; input: dy W^T scattered
    back tap by tap).  The float64 BLAS then sees a few fat products instead of kh * kw skinny ones and autograd keeps no per-tap
    intermediates: DeepSpeech2's 41 x 11 / 21 x 11 filters at batch 16 take tens of seconds instead of minutes."""

    @staticmethod
    def _cols(xb, kh, kw, sh, sw, Ho, Wo):
        H, W, C = xb.shape
        xb = xb.contiguous()
        win = xb.as_strided((Ho, Wo, kh, kw, C), (sh * W * C, sw * C, W * C, C, 1))
        return win.reshape(Ho * Wo, kh * kw * C)

    @staticmethod
    def forward(ctx, x, kernel, sh, sw):
        B, H, W, C = x.shape
        kh, kw, _, O = kernel.shape
        Ho, Wo = (H - kh) // sh + 1, (W - kw) // sw + 1
        wm = kernel.reshape(kh * kw * C, O)
        out = torch.empty(B, Ho, Wo, O, dtype=x.dtype)
        for b in range(B):
            out[b] = (_Conv2dIm2col._cols(x[b], kh, kw, sh, sw, Ho, Wo) @ wm).view(Ho, Wo, O)
        ctx.save_for_backward(x, kernel)
        ctx.geom = (sh, sw, Ho, Wo)
        return out

    @staticmethod
    def backward(ctx, dy):
        x, kernel = ctx.saved_tensors
        sh, sw, Ho, Wo = ctx.geom
        B, H, W, C = x.shape
        kh, kw, _, O = kernel.shape
        wm = kernel.reshape(kh * kw * C, O)
        dw = torch.zeros_like(wm) if ctx.needs_input_grad[1] else None
        dx = torch.zeros_like(x) if ctx.needs_input_grad[0] else None
        for b in range(B):
            dyb = dy[b].reshape(Ho * Wo, O)
            if dw is not None:
                dw += _Conv2dIm2col._cols(x[b], kh, kw, sh, sw, Ho, Wo).t() @ dyb
            if dx is not None:
                dcols = (dyb @ wm.t()).view(Ho, Wo, kh, kw, C)
                for r in range(kh):
                    for s in range(kw):
                        dx[b, r: r + sh * (Ho - 1) + 1: sh, s: s + sw * (Wo - 1) + 1: sw, :] += dcols[:, :, r, s, :]
        return dx, (dw.view_as(kernel) if dw is not None else None), None, None


def conv2d_nhwc(x, kernel, bias, strides):
    sh, sw = (strides, strides) if isinstance(strides, int) else strides
    return _Conv2dIm2col.apply(x, kernel, sh, sw) + bias


# --------------------------------------------------------------------------------------
# Recurrent cells and K.rnn masking  (las.py:62-126)  [TF-sem]
# --------------------------------------------------------------------------------------
def lstm_cell(x, h, c, W, U, b, x_mode="dense"):
    """Keras LSTMCell: gates i,f,c~,o; c' = f c + i tanh(.); h' = o tanh(c')."""
    H = h.shape[-1]
    z = _xw(x, W, x_mode, H) + mm_cell(h, U, H) + b
    i, f, g, o = z[:, :H], z[:, H:2 * H], z[:, 2 * H:3 * H], z[:, 3 * H:]
    i, f, o = torch.sigmoid(i), torch.sigmoid(f), torch.sigmoid(o)
    c2 = f * c + i * torch.tanh(g)
    h2 = o * torch.tanh(c2)
    return h2, [h2, c2]


def gru_cell(x, h, W, U, b, x_mode="dense"):
    """Keras GRUCell(reset_after=True): bias [2, 3H] = (input bias, recurrent bias); order z,r,h~."""
    H = h.shape[-1]
    mx = _xw(x, W, x_mode, H) + b[0]
    mh = mm_cell(h, U, H) + b[1]
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


def simple_rnn_cell(x, h, W, U, b, x_mode="dense"):
    H = h.shape[-1]
    h2 = torch.tanh(_xw(x, W, x_mode, H) + b + mm_cell(h, U, H))
    return h2, [h2]


def num_states(rnn_type):
    return 2 if rnn_type == "lstm" else 1


def rnn_layer(rnn_type, x, mask, W, U, b, initial_state=None, go_backwards=False,
              in_mult=None, rec_mult=None, x_mode="dense"):
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
                out, new = lstm_cell(xt, h, states[1], W, U, b, x_mode)
            elif rnn_type == "gru":
                out, new = gru_cell(xt, h, W, U, b, x_mode)
            else:
                out, new = simple_rnn_cell(xt, h, W, U, b, x_mode)
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
