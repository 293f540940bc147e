"""Oracle (TEST INFRASTRUCTURE): DeepSpeech2 forward, restating
/root/reference/speech_recognition/models/deepspeech2.py in torch-CPU (differentiable).
Parameter names = TF checkpoint keys (tests/data/model-checkpoints/ds.ckpt.index).
"""
from typing import Dict

import torch

from . import layers as L
from .las import STREAM_ENC_IN, STREAM_ENC_REC


def conv_out_dims(T, Fq, kernel_sizes, strides):
    for (kt, kf), (st, sf) in zip(kernel_sizes, strides):
        T = (T - kt) // st + 1
        Fq = (Fq - kf) // sf + 1
    return T, Fq


def param_shapes(cfg, freq_dim=80, feat_dim=3) -> Dict[str, tuple]:
    rt, V, H = cfg["rnn_type"], cfg["vocab_size"], cfg["hidden_dim"]
    g = {"lstm": 4, "gru": 3, "rnn": 1}[rt]
    s = {}
    cin, Fq = feat_dim, freq_dim
    for i, (ch, (kt, kf), (st, sf)) in enumerate(zip(cfg["channels"], cfg["kernel_sizes"], cfg["strides"])):
        s[f"convolution/conv_layers/{i}/kernel"] = (kt, kf, cin, ch)
        s[f"convolution/conv_layers/{i}/bias"] = (ch,)
        cin = ch
        Fq = (Fq - kf) // sf + 1
    din = Fq * cin
    for i in range(cfg["num_reccurent_layers"]):
        for d in ("forward_rnn", "backward_rnn"):
            pre = f"recurrent/rnn_layers/{i}/{d}/cell/"
            s[pre + "kernel"] = (din, g * H)
            s[pre + "recurrent_kernel"] = (H, g * H)
            s[pre + "bias"] = (2, g * H) if rt == "gru" else (g * H,)
        for n in ("gamma", "beta", "moving_mean", "moving_variance"):
            s[f"recurrent/batch_norm/{i}/{n}"] = (2 * H,)
        din = 2 * H
    s["fully_connected/kernel"] = (2 * H, V)
    s["fully_connected/bias"] = (V,)
    return s


def audio_mask(audio, kernel_sizes, strides, mode="intended"):
    """Convolution._audio_mask (deepspeech2.py:68-78).

    mode="reference_compat": line 74 as written, `tf.reduce_prod([time_stride, _ in self.strides])`,
    is the product of the last time stride and the boolean `_ in self.strides` (an int tested for
    membership in a list of lists -> False), i.e. 0: the mask is sliced to zero width and
    `reduce_any` over the empty axis gives all-False (unconfirmed by execution).
    mode="intended": product of the time strides."""
    B, T = audio.shape[:2]
    m = (audio.reshape(B, T, -1) != 0.0).any(dim=2)
    L_ = T
    for (kt, _), (st, _) in zip(kernel_sizes, strides):
        L_ -= kt - st
        L_ //= st
    if mode == "reference_compat":
        return torch.zeros(B, L_, dtype=torch.bool)
    sc = 1
    for st, _ in strides:
        sc *= st
    return m[:, : L_ * sc].reshape(B, L_, sc).any(dim=2)


def ds2_forward(p, cfg, audio, training=False, seed=0, mask_mode="intended", return_aux=False):
    """DeepSpeech2.call (deepspeech2.py:174-178): conv -> recurrent -> * mask -> Dense(V)."""
    rt = cfg["rnn_type"]
    rate, rrate = float(cfg["dropout"]), float(cfg.get("recurrent_dropout", 0.0))
    dt = audio.dtype
    mask = audio_mask(audio, cfg["kernel_sizes"], cfg["strides"], mask_mode)
    x = audio
    for i in range(cfg["num_conv_layers"]):
        x = L.conv2d_nhwc(x, p[f"convolution/conv_layers/{i}/kernel"], p[f"convolution/conv_layers/{i}/bias"],
                          tuple(cfg["strides"][i]))
    B = x.shape[0]
    x = x.reshape(B, x.shape[1], x.shape[2] * x.shape[3])
    states = None
    bn_updates = {}
    H = cfg["hidden_dim"]
    for i in range(cfg["num_reccurent_layers"]):
        pre = f"recurrent/rnn_layers/{i}/"
        fwd = tuple(p[pre + "forward_rnn/cell/" + n] for n in ("kernel", "recurrent_kernel", "bias"))
        bwd = tuple(p[pre + "backward_rnn/cell/" + n] for n in ("kernel", "recurrent_kernel", "bias"))
        mf = mb = rf = rb = None
        # [TF-sem] recurrent_dropout != 0 switches the Keras cells to implementation 1: one mask per gate on BOTH operands (streams of
        # gate g: + 128 g); without it, implementation 2: a single input mask
        ng = {"lstm": 4, "gru": 3, "rnn": 1}[rt]
        per_gate = training and rrate > 0
        if training and rate > 0:
            if per_gate:
                mf = [L.dropout_mult(seed, STREAM_ENC_IN + 2 * i + 128 * g, (B, x.shape[2]), rate, dt) for g in range(ng)]
                mb = [L.dropout_mult(seed, STREAM_ENC_IN + 2 * i + 1 + 128 * g, (B, x.shape[2]), rate, dt) for g in range(ng)]
            else:
                mf = L.dropout_mult(seed, STREAM_ENC_IN + 2 * i, (B, x.shape[2]), rate, dt)
                mb = L.dropout_mult(seed, STREAM_ENC_IN + 2 * i + 1, (B, x.shape[2]), rate, dt)
        if per_gate:
            rf = [L.dropout_mult(seed, STREAM_ENC_REC + 2 * i + 128 * g, (B, H), rrate, dt) for g in range(ng)]
            rb = [L.dropout_mult(seed, STREAM_ENC_REC + 2 * i + 1 + 128 * g, (B, H), rrate, dt) for g in range(ng)]
        x, *states = L.birnn(rt, x, mask, fwd, bwd, states, mf, mb, rf, rb)
        bn = f"recurrent/batch_norm/{i}/"
        x, mm, mv = L.batch_norm(x, p[bn + "gamma"], p[bn + "beta"], p[bn + "moving_mean"],
                                 p[bn + "moving_variance"], training)
        bn_updates[bn + "moving_mean"], bn_updates[bn + "moving_variance"] = mm, mv
    x = x * mask[:, :, None].to(dt)
    logits = L.mm_dense(x, p["fully_connected/kernel"]) + p["fully_connected/bias"]
    if return_aux:
        return logits, {"mask": mask, "bn_updates": bn_updates}
    return logits
