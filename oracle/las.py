"""Oracle (TEST INFRASTRUCTURE): Listen-Attend-Spell forward, restating
/root/reference/speech_recognition/models/las.py in torch-CPU (differentiable).

Parameter names are the TF checkpoint keys of the reference model (tests/data/model-checkpoints/
las.ckpt.index), e.g. ``listener/encoder_layers/0/forward_rnn/cell/kernel``; shapes are the Keras
layouts (kernel [Din, 4H], recurrent_kernel [H, 4H], conv kernels HWIO, Dense kernels [in, out]).
"""
from typing import Dict, List, Optional

import numpy as np
import torch

from . import layers as L

# RNG stream ids (must equal speech_recognition_amd.rng; checked in tests/test_oracle.py)
STREAM_CONV1_DROP = 1
STREAM_CONV2_DROP = 2
STREAM_SPECAUG = 3
STREAM_TEACHER = 4
STREAM_ENC_IN = 10          # + 2*layer + direction
STREAM_ENC_REC = 60         # + 2*layer + direction (recurrent dropout, DS2 only)
STREAM_DEC = 1000           # + 32*step + {0: embedding, 1: output, 2+j: decoder layer j input}


def param_shapes(cfg, freq_dim=80, feat_dim=3) -> Dict[str, tuple]:
    """Shapes of every LAS variable for a model config dict (keys of las_small.yml)."""
    rt, V = cfg["rnn_type"], cfg["vocab_size"]
    He, Hd = cfg["encoder_hidden_dim"], cfg["decoder_hidden_dim"]
    g = {"lstm": 4, "gru": 3, "rnn": 1}[rt]
    f1 = (freq_dim - 3) // 2 + 1
    f2 = (f1 - 3) // 2 + 1
    s = {"listener/conv1/kernel": (3, 3, feat_dim, 32), "listener/conv1/bias": (32,),
         "listener/conv2/kernel": (3, 3, 32, 32), "listener/conv2/bias": (32,)}
    din = f2 * 32
    for i in range(cfg["num_encoder_layers"]):
        for d in ("forward_rnn", "backward_rnn"):
            pre = f"listener/encoder_layers/{i}/{d}/cell/"
            s[pre + "kernel"] = (din, g * He)
            s[pre + "recurrent_kernel"] = (He, g * He)
            s[pre + "bias"] = (2, g * He) if rt == "gru" else (g * He,)
        s[f"listener/projection/{i}/kernel"] = (2 * He, 2 * He)
        s[f"listener/projection/{i}/bias"] = (2 * He,)
        for n in ("gamma", "beta", "moving_mean", "moving_variance"):
            s[f"listener/batch_norm/{i}/{n}"] = (2 * He,)
        din = 2 * He
    s["listener/hidden_states_proj/kernel"] = (2 * He, Hd)
    s["listener/hidden_states_proj/bias"] = (Hd,)
    if rt == "lstm":
        s["listener/cell_states_proj/kernel"] = (2 * He, Hd)
        s["listener/cell_states_proj/bias"] = (Hd,)
    s["attend_and_speller/embedding/embeddings"] = (V, Hd)
    din = Hd + 2 * He
    for j in range(cfg["num_decoder_layers"]):
        pre = f"attend_and_speller/decoder_layers/{j}/cell/"
        s[pre + "kernel"] = (din, g * Hd)
        s[pre + "recurrent_kernel"] = (Hd, g * Hd)
        s[pre + "bias"] = (2, g * Hd) if rt == "gru" else (g * Hd,)
        din = Hd
    s["attend_and_speller/attention/query_weight/kernel"] = (Hd, Hd)
    s["attend_and_speller/attention/query_weight/bias"] = (Hd,)
    s["attend_and_speller/attention/key_weight/kernel"] = (2 * He, Hd)
    s["attend_and_speller/attention/key_weight/bias"] = (Hd,)
    s["attend_and_speller/feedforward/kernel"] = (Hd, V)
    s["attend_and_speller/feedforward/bias"] = (V,)
    return s


def audio_mask(audio, kernel=3, stride=2):
    """Listener._audio_mask (las.py:205-217)."""
    B, T = audio.shape[:2]
    m = (audio.reshape(B, T, -1) != 0.0).any(dim=2)
    L_ = T
    L_ -= kernel - stride
    L_ //= stride
    L_ -= kernel - stride
    L_ //= stride
    n = L_ * stride ** 2
    return m[:, :n].reshape(B, -1, stride ** 2).any(dim=2)


def listener(p, cfg, audio, training, seed=0):
    """Listener.call (las.py:177-203). Returns (enc, mask, states, bn_updates)."""
    rt, rate, nl = cfg["rnn_type"], float(cfg["dropout"]), cfg["num_encoder_layers"]
    dt = audio.dtype
    mask = audio_mask(audio)
    x = L.conv2d_nhwc(audio, p["listener/conv1/kernel"], p["listener/conv1/bias"], 2)
    if training and rate > 0:
        x = x * L.dropout_mult(seed, STREAM_CONV1_DROP, x.shape, rate, dt)
    x = L.conv2d_nhwc(x, p["listener/conv2/kernel"], p["listener/conv2/bias"], 2)
    if training and rate > 0:
        x = x * L.dropout_mult(seed, STREAM_CONV2_DROP, x.shape, rate, dt)
    B = x.shape[0]
    x = x.reshape(B, x.shape[1], x.shape[2] * x.shape[3])
    states = None
    bn_updates = {}
    for i in range(nl):
        pre = f"listener/encoder_layers/{i}/"
        fwd = tuple(p[pre + "forward_rnn/cell/" + n] for n in ("kernel", "recurrent_kernel", "bias"))
        bwd = tuple(p[pre + "backward_rnn/cell/" + n] for n in ("kernel", "recurrent_kernel", "bias"))
        mf = mb = None
        if training and rate > 0:
            mf = L.dropout_mult(seed, STREAM_ENC_IN + 2 * i, (B, x.shape[2]), rate, dt)
            mb = L.dropout_mult(seed, STREAM_ENC_IN + 2 * i + 1, (B, x.shape[2]), rate, dt)
        x, *states = L.birnn(rt, x, mask, fwd, bwd, states, mf, mb)
        x = L.mm_dense(x, p[f"listener/projection/{i}/kernel"]) + p[f"listener/projection/{i}/bias"]
        bn = f"listener/batch_norm/{i}/"
        x, mm, mv = L.batch_norm(x, p[bn + "gamma"], p[bn + "beta"], p[bn + "moving_mean"],
                                 p[bn + "moving_variance"], training)
        bn_updates[bn + "moving_mean"], bn_updates[bn + "moving_variance"] = mm, mv
        x = torch.relu(x)
    if len(states) == 2:
        states = [L.mm_dense(torch.cat(states, dim=-1), p["listener/hidden_states_proj/kernel"])
                  + p["listener/hidden_states_proj/bias"]]
    else:
        states = [
            L.mm_dense(torch.cat(states[::2], dim=-1), p["listener/hidden_states_proj/kernel"])
            + p["listener/hidden_states_proj/bias"],
            L.mm_dense(torch.cat(states[1::2], dim=-1), p["listener/cell_states_proj/kernel"])
            + p["listener/cell_states_proj/bias"],
        ]
    return x, mask, states, bn_updates


def attend_and_speller(p, cfg, enc, tok, attention_mask, states, training, seed=0, step=0, trace=None, keys=None):
    """AttendAndSpeller.call (las.py:267-292) for one decoder step. Returns (logits, states, probs).
    trace: optional dict (tests) that receives this step's context, masked attention scores and per-layer cell states with
    retain_grad(), so that the gradients the HIP backward sweep writes out can be read after backward()."""
    rt, rate, nd = cfg["rnn_type"], float(cfg["dropout"]), cfg["num_decoder_layers"]
    pad = cfg.get("pad_id", 0)
    dt = enc.dtype
    B = enc.shape[0]
    m = (tok != pad)[:, None]                                    # [B,1] mask of the 1-step sequence
    x = p["attend_and_speller/embedding/embeddings"][tok.long()]
    base = STREAM_DEC + 32 * step
    if training and rate > 0:
        x = x * L.dropout_mult(seed, base + 0, x.shape, rate, dt)
    a = "attend_and_speller/attention/"
    if keys is not None:       # L.bf16_operands(): the build's hoisted association of the same sums (rounding points follow it)
        ctx, probs, scores = L.attention_hoisted(states[0], keys, enc, attention_mask, return_scores=True)
    else:
        ctx, probs, scores = L.attention(states[0], enc, enc, attention_mask, p[a + "query_weight/kernel"],
                                         p[a + "query_weight/bias"], p[a + "key_weight/kernel"], p[a + "key_weight/bias"], return_scores=True)
    if trace is not None:
        for t_ in (ctx, scores):
            if t_.requires_grad:
                t_.retain_grad()
        trace.setdefault("ctx", []).append(ctx)
        trace.setdefault("scores", []).append(scores)
    x = torch.cat([x, ctx], dim=-1)
    for j in range(nd):
        pre = f"attend_and_speller/decoder_layers/{j}/cell/"
        im = None
        if training and rate > 0:
            im = L.dropout_mult(seed, base + 2 + j, (B, x.shape[1]), rate, dt)
        # (x_mode: where the build multiplies this cell's input - L._xw; without L.bf16_operands() it changes nothing)
        Hd = cfg["decoder_hidden_dim"]
        out, states = L.rnn_layer(rt, x[:, None, :], m, p[pre + "kernel"], p[pre + "recurrent_kernel"],
                                  p[pre + "bias"], initial_state=states, in_mult=im, x_mode=("split", Hd) if j == 0 else "cell")
        x = out[:, -1]                                           # return_sequences=False: last output
        if trace is not None:
            trace.setdefault(f"y{j}", []).append(x)
            trace.setdefault(f"h{j}", []).append(states[0])
            if len(states) > 1:
                trace.setdefault(f"c{j}", []).append(states[1])
    if training and rate > 0:
        x = x * L.dropout_mult(seed, base + 1, x.shape, rate, dt)
    logits = L.mm_dense(x, p["attend_and_speller/feedforward/kernel"]) + p["attend_and_speller/feedforward/bias"]
    return logits, states, probs


def las_forward(p, cfg, audio, tokens, training=False, seed=0, use_teacher_forcing=True, return_aux=False):
    """LAS.call (las.py:349-380). `use_teacher_forcing` is the one-per-batch coin of las.py:366
    (also drawn at eval in the reference); the caller decides it."""
    enc, mask, states, bn_updates = listener(p, cfg, audio, training, seed)
    U = tokens.shape[1]
    outs, probs_all = [], []
    logits = None
    trace = {} if return_aux else None
    init_states = list(states)
    if return_aux:
        for t_ in init_states:
            if t_.requires_grad:
                t_.retain_grad()
    keys = None
    if L._Bf16.on:
        a = "attend_and_speller/attention/"
        keys = L.attention_keys_hoisted(enc, p[a + "query_weight/kernel"], p[a + "query_weight/bias"], p[a + "key_weight/kernel"], p[a + "key_weight/bias"])
    for i in range(U):
        if use_teacher_forcing or i == 0:
            tok = tokens[:, i]
        else:
            tok = logits.argmax(dim=-1)
        logits, states, probs = attend_and_speller(p, cfg, enc, tok, mask, states, training, seed, i, trace, keys)
        outs.append(logits)
        probs_all.append(probs)
    out = torch.stack(outs, dim=1)
    if return_aux:
        return out, {"enc": enc, "mask": mask, "bn_updates": bn_updates, "probs": torch.stack(probs_all, 1),
                     "states": states, "init_states": init_states, "trace": trace}
    return out
