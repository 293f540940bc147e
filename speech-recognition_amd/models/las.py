"""Listen, Attend and Spell on MI355X (mirror of speech_recognition/models/las.py).

Same constructor, call signature, sub-objects (``listener``, ``attend_and_speller``), loss/metric
factories and batching helpers as the reference; every tensor operation is a kernel of
libasr_mi355x.so.  Differences that are deliberate and do not change results beyond rounding:

  * the attention key projection, which the reference recomputes at every decoder step
    (las.py:50 inside the step called at las.py:282), is hoisted: K = enc Wk + bk, Kq = K Wq^T and
    s0 = K bq are computed once per batch (53 % of the reference's forward flops disappear);
  * under teacher forcing the embedding lookup, the embedding half of the first decoder LSTM's input
    projection, the vocabulary projection and the loss are batched over all decoder steps;
  * dropout masks come from a stateless hash (rng.py) instead of TF's stateful RNG.

Decoder-side sequences are stored step-major ([U, B, ...]) so that a step's slice is contiguous;
encoder-side tensors are batch-major ([B, T', ...]) like the reference.
"""
import math
import os
import random
from collections import OrderedDict
from typing import List, Optional, Tuple

import torch

from .. import _lib, ops
from .. import rng as R
from ..layers import NG, NS, BiRNN, Overlap, cell_input_grad, cell_param_grads, dense_bwd
from ..measure import SparseCategoricalAccuracy, SparseCategoricalCrossentropy
from ..params import ParamStore, init_value
from .model_proto import ModelProto


# ASR_DECODER_SWEEP=0 forces one launch per decoder kernel and step (debugging / A-B timing)
DECODER_SWEEP = os.environ.get("ASR_DECODER_SWEEP", "1") != "0"
DECODER_SWEEP_BWD = os.environ.get("ASR_DECODER_SWEEP_BWD", "1") != "0"     # the backward loop alone


def get_rnn_cls(rnn_type: str) -> str:
    """las.py:10-17: validates the type (the 'class' here is just the kernel family name)."""
    if rnn_type in ("rnn", "lstm", "gru"):
        return rnn_type
    raise ValueError(f"rnn_type: {rnn_type} is invalid!")


def _f(*shape, device="cuda"):
    return torch.empty(*shape, device=device, dtype=torch.float32)


class _Workspace:
    """All activations / gradients of one (B, T, U) shape, allocated once (graph-capturable)."""


class Listener:
    """Callable view on the encoder half (las.py:129-217): ``listener(audio, training) ->
    [enc [B,T',2H], mask bool[B,T'], h0, (c0)]``."""

    def __init__(self, model):
        self.m = model

    def __call__(self, audio, training: Optional[bool] = None):
        m = self.m
        m._ensure_built(audio.shape[2], audio.shape[3])
        B, T = audio.shape[:2]
        ws = m._workspace(B, T, 1)
        m._encode(ws, audio.contiguous(), bool(training))
        states = [ws.hin[0].clone()] + ([ws.cin[0].clone()] if m.rt == "lstm" else [])
        return [ws.enc.view(B, ws.T2, 2 * m.He).clone(), ws.mask.bool().clone()] + states


class AttendAndSpeller:
    """Callable view on one decoder step (las.py:220-292): ``attend_and_speller(enc, tok, mask,
    states, training) -> [logits [B,V], *states]``.  Used by greedy/beam search in the reference
    (search.py:42,59,128,183); the training path uses the batched form inside LAS.call."""

    def __init__(self, model):
        self.m = model
        self._step = 0

    def __call__(self, audio_output, decoder_input, attention_mask, states, training: Optional[bool] = None):
        m = self.m
        B, T2 = audio_output.shape[:2]
        ws = m._step_workspace(B, T2)
        ws.enc.copy_(audio_output.reshape(B * T2, -1))
        ws.mask.copy_(attention_mask.to(torch.uint8))
        m._attention_keys(ws)
        ws.hin[0].copy_(states[0])
        if m.rt == "lstm":
            ws.cin[0].copy_(states[1])
        ws.toks_T[0].copy_(decoder_input.to(torch.int32))
        m._embed(ws, 0, 1, bool(training), step_offset=self._step)
        m._decoder_step(ws, 0, bool(training), step_offset=self._step)
        m._vocab(ws, 0, 1, bool(training), step_offset=self._step)
        out = [ws.logits.view(1, B, m.V)[0].clone(), ws.hin[1].clone()]
        if m.rt == "lstm":
            out.append(ws.cin[1].clone())
        return out


class LAS(ModelProto):
    """Listen, Attend and Spell (las.py:295-406).  Arguments as in the reference."""

    model_checkpoint_path = "model-{epoch}epoch-{val_loss:.4f}loss_{val_accuracy:.4f}acc.ckpt"

    def __init__(self, rnn_type: str, vocab_size: int, encoder_hidden_dim: int, decoder_hidden_dim: int,
                 num_encoder_layers: int, num_decoder_layers: int, dropout: float, teacher_forcing_rate: float,
                 pad_id: int = 0, device: str = "cuda", seed: Optional[int] = None, **kwargs):
        super().__init__(**kwargs)
        self.rt = get_rnn_cls(rnn_type)
        if num_decoder_layers > R.MAX_DECODER_LAYERS:
            raise ValueError(f"num_decoder_layers > {R.MAX_DECODER_LAYERS} is not supported")
        self.vocab_size = self.V = vocab_size
        self.pad_id = pad_id
        self.teacher_forcing_rate = teacher_forcing_rate
        self.He, self.Hd = encoder_hidden_dim, decoder_hidden_dim
        self.Le, self.Ld = num_encoder_layers, num_decoder_layers
        self.dropout = float(dropout)
        self.device = device
        self.init_seed = seed
        self._py_rng = random.Random(seed)
        self.kernel_sizes, self.strides = (3, 3), 2
        self.listener = Listener(self)
        self.attend_and_speller = AttendAndSpeller(self)
        self._ws = {}
        self._ov = Overlap("las")            # weight gradients released beside the next backward sweep (layers.Overlap)
        self._packed_version = -1
        self._version = 0
        # device state: [0] optimizer iterations, [1] dropout seed (advanced by the trainer each step)
        self.state = torch.tensor([0, (seed or 0) & 0x7FFFFFFF, 0, 0], dtype=torch.int32, device=device)

    # ------------------------------------------------------------------------------------------ build
    def param_shapes(self, freq_dim, feat_dim):
        rt, g = self.rt, NG[self.rt]
        He, Hd, V = self.He, self.Hd, self.V
        f1 = (freq_dim - 3) // 2 + 1
        f2 = (f1 - 3) // 2 + 1
        enc, dec, bufs = OrderedDict(), OrderedDict(), OrderedDict()
        enc["listener/conv1/kernel"] = (3, 3, feat_dim, 32)
        enc["listener/conv1/bias"] = (32,)
        enc["listener/conv2/kernel"] = (3, 3, 32, 32)
        enc["listener/conv2/bias"] = (32,)
        din = f2 * 32
        for i in range(self.Le):
            enc.update(BiRNN.param_shapes(f"listener/encoder_layers/{i}/", rt, din, He))
            enc[f"listener/projection/{i}/kernel"] = (2 * He, 2 * He)
            enc[f"listener/projection/{i}/bias"] = (2 * He,)
            enc[f"listener/batch_norm/{i}/gamma"] = (2 * He,)
            enc[f"listener/batch_norm/{i}/beta"] = (2 * He,)
            bufs[f"listener/batch_norm/{i}/moving_mean"] = (2 * He,)
            bufs[f"listener/batch_norm/{i}/moving_variance"] = (2 * He,)
            din = 2 * He
        dec["listener/hidden_states_proj/kernel"] = (2 * He, Hd)
        dec["listener/hidden_states_proj/bias"] = (Hd,)
        if rt == "lstm":
            dec["listener/cell_states_proj/kernel"] = (2 * He, Hd)
            dec["listener/cell_states_proj/bias"] = (Hd,)
        dec["attend_and_speller/embedding/embeddings"] = (V, Hd)
        din = Hd + 2 * He
        for j in range(self.Ld):
            pre = f"attend_and_speller/decoder_layers/{j}/cell/"
            dec[pre + "kernel"] = (din, g * Hd)
            dec[pre + "recurrent_kernel"] = (Hd, g * Hd)
            dec[pre + "bias"] = (2, g * Hd) if rt == "gru" else (g * Hd,)
            din = Hd
        for n in ("query_weight", "key_weight"):
            dec[f"attend_and_speller/attention/{n}/kernel"] = (Hd if n == "query_weight" else 2 * He, Hd)
            dec[f"attend_and_speller/attention/{n}/bias"] = (Hd,)
        dec["attend_and_speller/feedforward/kernel"] = (Hd, V)
        dec["attend_and_speller/feedforward/bias"] = (V,)
        return enc, dec, bufs

    def build(self, frequency_dim: int, feature_dim: int):
        """Allocate and initialise the variables (Keras builds on first call; so do we)."""
        self.F, self.C = frequency_dim, feature_dim
        self.F1 = (frequency_dim - 3) // 2 + 1
        self.F2 = (self.F1 - 3) // 2 + 1
        self.D0 = self.F2 * 32
        enc, dec, bufs = self.param_shapes(frequency_dim, feature_dim)
        shapes = OrderedDict(list(dec.items()) + list(enc.items()))
        # all-reduce buckets in the order the backward pass completes them: the vocabulary layer (its gradient is ready
        # before the decoder chain starts: 4.1 M of las_small's 16 M), the rest of the decoder side (embedding, cells, attention,
        # state projections - complete after the chain, in flight during the encoder sweeps), then one bucket per encoder layer from
        # the top down (layer i's bucket travels while layer i-1 is swept), the convolutions with the bottom layer
        vocab = [n for n in dec if n.startswith("attend_and_speller/feedforward/")]
        rest = [n for n in dec if n not in vocab]
        per_layer = []
        for i in range(self.Le - 1, -1, -1):
            names = [n for n in enc if n.startswith((f"listener/encoder_layers/{i}/", f"listener/projection/{i}/", f"listener/batch_norm/{i}/"))]
            if i == 0:
                names += [n for n in enc if n.startswith("listener/conv")]
            per_layer.append(names)
        self.store = ParamStore(shapes, [vocab, rest] + per_layer, self.device)
        gen = torch.Generator().manual_seed(self.init_seed if self.init_seed is not None else random.randrange(2 ** 31))
        self.store.load({n: init_value(n, s, gen, self.rt) for n, s in shapes.items()})
        self.buffers = {n: init_value(n, s, gen).to(self.device) for n, s in bufs.items()}
        self.enc_layers = []
        din = self.D0
        for i in range(self.Le):
            self.enc_layers.append(BiRNN(self.store, f"listener/encoder_layers/{i}/", self.rt, din, self.He, self.dropout,
                                         R.STREAM_ENC_IN + 2 * i, self.device))
            din = 2 * self.He
        self.dec_cells = [ops.PackedCell(self.rt, self.Hd, [2 * self.He if j == 0 else self.Hd, self.Hd], self.device)
                          for j in range(self.Ld)]
        self.built = True
        self.weights_changed()
        return self

    def _ensure_built(self, freq_dim, feat_dim):
        if not self.built:
            self.build(freq_dim, feat_dim)
        elif (freq_dim, feat_dim) != (self.F, self.C):
            raise ValueError(f"model was built for [T,{self.F},{self.C}] features, got [T,{freq_dim},{feat_dim}]")

    def weights_changed(self):
        self._version += 1

    def pack_weights(self):
        """Refresh the MFMA-fragment images of every recurrent weight (cheap; once per optimizer step)."""
        p = self.store.p
        if ops.mixed_precision():
            self.store.refresh_bf16()
        cells = [cw for l in self.enc_layers for cw in l.pack_list()]
        for j, cell in enumerate(self.dec_cells):
            pre = f"attend_and_speller/decoder_layers/{j}/cell/"
            W = p[pre + "kernel"]
            cells.append((cell, [(W[self.Hd:] if j == 0 else W, False), (p[pre + "recurrent_kernel"], True)]))
        ops.pack_cells(cells)                              # every cell of the model in one launch
        self._packed_version = self._version

    # ------------------------------------------------------------------------------------------ workspaces
    def out_frames(self, T):
        T1 = (T - 3) // 2 + 1
        return T1, (T1 - 3) // 2 + 1

    def _workspace(self, B, T, U):
        key = (B, T, U)
        if key in self._ws:
            return self._ws[key]
        ws = _Workspace()
        dev, He, Hd, V, rt = self.device, self.He, self.Hd, self.V, self.rt
        T1, T2 = self.out_frames(T)
        if T2 < 1:
            raise ValueError(f"audio with {T} frames is too short for the two stride-2 convolutions")
        ws.B, ws.T, ws.U, ws.T1, ws.T2 = B, T, U, T1, T2
        f = lambda *s: _f(*s, device=dev)
        ws.mask = torch.empty(B, T2, dtype=torch.uint8, device=dev)
        ws.c1, ws.c2 = f(B, T1, self.F1, 32), f(B, T2, self.F2, 32)
        ws.layers = []
        for l in self.enc_layers:
            ws.layers.append(dict(rnn=l.alloc(B, T2, dev), z=f(B * T2, 2 * He), a=f(B * T2, 2 * He), mean=f(2 * He), rstd=f(2 * He)))
        ws.bn_ws = torch.empty(4 * He, dtype=torch.float64, device=dev)
        ws.enc = ws.layers[-1]["a"]
        ws.K, ws.Kq, ws.s0 = f(B * T2, Hd), f(B * T2, Hd), f(B * T2, 1)
        ws.hin, ws.cin = f(U + 1, B, Hd), f(U + 1, B, Hd)
        ws.toks_T = torch.zeros(U + 1, B, dtype=torch.int32, device=dev)
        ws.tokmask = torch.empty(U, B, dtype=torch.uint8, device=dev)
        ws.emb, ws.pre0 = f(U, B, Hd), f(U, B, NG[rt] * Hd)
        ws.dec = []
        for j, cell in enumerate(self.dec_cells):
            d = dict(y=f(U, B, Hd), saved=f(U, B, NS[rt] * Hd))
            if j < self.Ld - 1:
                d["h"], d["c"] = f(U, B, Hd), f(U, B, Hd)
            ws.dec.append(d)
        ws.e, ws.p, ws.ctx = f(B, T2), f(U, B, T2), f(U, B, 2 * He)
        # ASR_FUSED_ATTENTION=1 selects the one-launch attention steps (attention_fused.hip).  Off by default:
        # measured on las_small they take 17.5 us per step against 5.2 + 9.0 us for the two-kernel steps - the
        # in-launch combine (agent release/acquire + a serial last-arriver pass) costs more than the launch it saves
        fused = os.environ.get("ASR_FUSED_ATTENTION", "0") == "1" and ops.attn_fused_supported(T2, Hd, 2 * He)
        ws.attn_fused = ops.attn_fused_ws(B, Hd, 2 * He, dev) if fused else None
        ws.yd, ws.logits = f(U * B, Hd), f(U * B, V)
        ws.stats = torch.zeros(4, device=dev)
        # backward
        ws.dyd, ws.dctx, ws.dp, ws.ds = f(U * B, Hd), f(U, B, 2 * He), f(B, T2), f(U, B, T2)
        ws.dh_attn, ws.dc_dec, ws.ddirect = f(B, Hd), f(B, Hd), f(B, Hd)
        ws.xdrop = f(U * B, max(Hd, 2 * He))
        ws.demb = f(U * B, Hd)
        ws.denc, ws.dKq, ws.dK, ws.ds0 = f(B * T2, 2 * He), f(B * T2, Hd), f(B * T2, Hd), f(B * T2, 1)
        ws.dhs, ws.dcs = f(B, Hd), f(B, Hd)
        ws.dz = [f(B * T2, 2 * He) for _ in self.enc_layers]      # one per layer: layer i's projection gradient may still read its dz beside
        ws.dy = f(B, T2, 2 * He)                                  # the sweep while layer i-1's BatchNorm gradient is being written
        ws.dx = [f(B * T2, 2 * He), f(B * T2, 2 * He)]
        ws.dfin_h = [f(B, He), f(B, He)]
        ws.dc_enc = [f(B, He), f(B, He)]
        ws.dx0, ws.dc1 = f(B, T2, self.D0), f(B, T1, self.F1, 32)
        ws.ones_u = torch.ones(U, 1, device=dev)
        self._ws[key] = ws
        return ws

    def _step_workspace(self, B, T2):
        """Workspace for single decoder steps on a caller-supplied encoder output (search API)."""
        key = ("step", B, T2)
        if key in self._ws:
            return self._ws[key]
        T = 4 * T2 + 3
        ws = self._workspace(B, T, 1)
        assert ws.T2 == T2
        self._ws[key] = ws
        return ws

    # ------------------------------------------------------------------------------------------ forward pieces
    @property
    def seed(self):
        return self.state[1:2]

    def _encode(self, ws, audio, training):
        """Listener.call (las.py:177-203)."""
        if self._packed_version != self._version:
            self.pack_weights()
        p, B, T2, He, rt = self.store.p, ws.B, ws.T2, self.He, self.rt
        rate = self.dropout if training else 0.0
        ops.frame_mask(audio, 4, T2, ws.mask)                                            # las.py:205-217
        ops.conv2d_fwd(audio, p["listener/conv1/kernel"], p["listener/conv1/bias"], 2, ws.c1, self.seed, R.STREAM_CONV1_DROP, rate)
        ops.conv2d_fwd(ws.c1, p["listener/conv2/kernel"], p["listener/conv2/bias"], 2, ws.c2, self.seed, R.STREAM_CONV2_DROP, rate)
        x3 = ws.c2.view(B, T2, self.D0)
        states = None
        ops.dropout_tables([t for l, lw in zip(self.enc_layers, ws.layers) for t in l.dropout_table_list(lw["rnn"], training)], self.seed)
        for i, (l, lw) in enumerate(zip(self.enc_layers, ws.layers)):
            y = l.forward(lw["rnn"], x3, ws.mask, states, training, self.seed, tables_ready=True)
            states = l.final_states(lw["rnn"])
            ops.gemm(y.view(B * T2, 2 * He), p[f"listener/projection/{i}/kernel"], lw["z"], bias=p[f"listener/projection/{i}/bias"])
            bn = f"listener/batch_norm/{i}/"
            ops.bn_fwd(lw["z"], p[bn + "gamma"], p[bn + "beta"], lw["a"], lw["mean"], lw["rstd"], self.buffers[bn + "moving_mean"],
                       self.buffers[bn + "moving_variance"], ws.bn_ws, relu=True, training=training)
            x3 = lw["a"].view(B, T2, 2 * He)
        # las.py:196-202: project the concatenated final states of the two directions
        nst = 2 if rt == "lstm" else 1
        for k, (name, dst) in enumerate((("hidden_states_proj", ws.hin[0]), ("cell_states_proj", ws.cin[0]))[:nst]):
            W, b = p[f"listener/{name}/kernel"], p[f"listener/{name}/bias"]
            ops.gemm(states[k], W[:He], dst, bias=b)
            ops.gemm(states[nst + k], W[He:], dst, accumulate=1)
        ws.final_states = states

    def _attention_keys(self, ws):
        """Loop-invariant part of las.py:46-54: K = enc Wk + bk ; Kq = K Wq^T ; s0 = K bq."""
        p = self.store.p
        a = "attend_and_speller/attention/"
        ops.gemm(ws.enc, p[a + "key_weight/kernel"], ws.K, bias=p[a + "key_weight/bias"])
        ops.gemm(ws.K, p[a + "query_weight/kernel"], ws.Kq, trans_b=True)
        ops.rowdot(ws.K, p[a + "query_weight/bias"], ws.s0.view(-1))                         # s0 = K bq
        # mixed precision: the decoder steps stream Kq and enc once each per step - give them bf16 images (wide models only:
        # on las_small the streams sit in L2 / Infinity Cache and the steps are latency-bound either way)
        ws.attn_images = None
        if ops.mixed_precision() and self.Hd % 8 == 0 and (2 * self.He) % 8 == 0 and self.Hd >= int(os.environ.get("ASR_ATTN_IMAGE_MIN_HD", "512")):
            if getattr(ws, "Kq16", None) is None:
                ws.Kq16 = torch.empty(ws.Kq.numel(), device=ws.Kq.device, dtype=torch.bfloat16)
                ws.enc16 = torch.empty(ws.enc.numel(), device=ws.enc.device, dtype=torch.bfloat16)
            ops.f32_to_bf16(ws.Kq.view(-1), ws.Kq16)
            ops.f32_to_bf16(ws.enc.view(-1), ws.enc16)
            ws.attn_images = (ws.Kq16, ws.enc16)

    def _drops(self, ws, i0, n, training, step_offset=0):
        """Row-dropout descriptors of the decoder sites for step-major rows [i0*B, (i0+n)*B)."""
        rate = self.dropout if training else 0.0
        base = R.STREAM_DEC + R.DEC_STREAMS_PER_STEP * (i0 + step_offset)
        mk = lambda k, ld, off=0: ops.rowdrop(base + k, R.DEC_STREAMS_PER_STEP, -ws.B, ld, off, rate)
        return mk, rate

    def _embed(self, ws, i0, n, training, step_offset=0):
        """Embedding + its dropout + the first decoder layer's input dropout (las.py:276-278, 283-288),
        then the embedding half of layer 0's input projection, for steps [i0, i0+n)."""
        p, B, Hd = self.store.p, ws.B, self.Hd
        mk, rate = self._drops(ws, i0, n, training, step_offset)
        tok = ws.toks_T[i0:i0 + n]
        ops.token_mask(tok, self.pad_id, ws.tokmask[i0:i0 + n])
        emb = ws.emb[i0:i0 + n].view(n * B, Hd)
        ops.embedding_fwd(p["attend_and_speller/embedding/embeddings"], tok, emb, self.seed if rate > 0 else None,
                          mk(0, Hd), mk(2, Hd + 2 * self.He))
        pre = "attend_and_speller/decoder_layers/0/cell/"
        b = p[pre + "bias"]
        ops.gemm(emb, p[pre + "kernel"][:Hd], ws.pre0[i0:i0 + n].view(n * B, -1), bias=(b[0] if self.rt == "gru" else b))

    def _vocab(self, ws, i0, n, training, step_offset=0):
        """Output dropout + Dense(V) (las.py:291) for steps [i0, i0+n)."""
        p, B, Hd = self.store.p, ws.B, self.Hd
        mk, rate = self._drops(ws, i0, n, training, step_offset)
        y = ws.dec[-1]["y"][i0:i0 + n].view(n * B, Hd)
        yd = ws.yd[i0 * B:(i0 + n) * B]
        if rate > 0:
            ops.dropout_rows(y, yd, self.seed, mk(1, Hd))
            src = yd
        else:
            src = y
        ops.gemm(src, p["attend_and_speller/feedforward/kernel"], ws.logits[i0 * B:(i0 + n) * B], bias=p["attend_and_speller/feedforward/bias"])

    def _cell_states(self, ws, j, i):
        """(h_in, c_in, h_out, c_out) buffers of decoder layer j at step i (las.py:285-288 state threading)."""
        last = self.Ld - 1
        h_in = ws.hin[i] if j == 0 else ws.dec[j - 1]["h"][i]
        c_in = ws.cin[i] if j == 0 else ws.dec[j - 1]["c"][i]
        h_out = ws.hin[i + 1] if j == last else ws.dec[j]["h"][i]
        c_out = ws.cin[i + 1] if j == last else ws.dec[j]["c"][i]
        return h_in, c_in, h_out, c_out

    def _decoder_step(self, ws, i, training, step_offset=0):
        """One AttendAndSpeller step minus embedding/vocab (las.py:282-288)."""
        p, B, Hd, He, rt = self.store.p, ws.B, self.Hd, self.He, self.rt
        rate = self.dropout if training else 0.0
        base = R.STREAM_DEC + R.DEC_STREAMS_PER_STEP * (i + step_offset)
        if ws.attn_fused is not None:      # one launch per step
            ops.attn_fused_fwd(ws.hin[i], ws.Kq.view(B, ws.T2, Hd), ws.s0, ws.mask, ws.enc.view(B, ws.T2, 2 * He), ws.attn_fused, ws.p[i], ws.ctx[i])
        else:
            ops.attn_step_fwd(ws.hin[i], ws.Kq.view(B, ws.T2, Hd), ws.s0, ws.mask, ws.enc.view(B, ws.T2, 2 * He), ws.e, ws.p[i], ws.ctx[i],
                              images=getattr(ws, "attn_images", None))
        for j, cell in enumerate(self.dec_cells):
            h_in, c_in, h_out, c_out = self._cell_states(ws, j, i)
            pre = f"attend_and_speller/decoder_layers/{j}/cell/"
            b = p[pre + "bias"]
            st = _lib.RnnStepFwd()
            g = cell.geom
            st.nseg, st.KSt, st.Wp, st.Wp16 = 2, g.KSt, cell.Wp.data_ptr(), cell.wp16_ptr()
            x = ws.ctx[i] if j == 0 else ws.dec[j - 1]["y"][i]
            Kx = 2 * He if j == 0 else Hd
            st.seg_x[0], st.seg_ld[0], st.seg_K[0], st.seg_ks0[0] = x.data_ptr(), x.stride(0), Kx, g.ks0[0]
            st.seg_drop_rate[0], st.seg_drop_stream[0] = rate, base + 2 + j
            st.seg_drop_ld[0], st.seg_drop_off[0] = (Hd + 2 * He, Hd) if j == 0 else (Hd, 0)
            st.seg_x[1], st.seg_ld[1], st.seg_K[1], st.seg_ks0[1] = h_in.data_ptr(), h_in.stride(0), Hd, g.ks0[1]
            if j == 0:
                st.pre, st.pre_ld = ws.pre0[i].data_ptr(), ws.pre0[i].stride(0)
            else:
                st.bias = (b[0] if rt == "gru" else b).data_ptr()
            if rt == "gru":
                st.bias_rec = b[1].data_ptr()
            st.h_prev, st.h_prev_ld = h_in.data_ptr(), h_in.stride(0)
            if rt == "lstm":
                st.c_prev, st.c_prev_ld = c_in.data_ptr(), c_in.stride(0)
                st.c_out, st.c_out_ld = c_out.data_ptr(), c_out.stride(0)
            st.mask, st.mask_ld = ws.tokmask[i].data_ptr(), 1
            st.h_out, st.h_out_ld = h_out.data_ptr(), h_out.stride(0)
            y = ws.dec[j]["y"][i]
            st.y_out, st.y_out_ld = y.data_ptr(), y.stride(0)
            if training:
                sv = ws.dec[j]["saved"][i]
                st.saved, st.saved_ld = sv.data_ptr(), sv.stride(0)
            ops.rnn_cell_fwd(rt, B, Hd, [st], self.seed if rate > 0 else None)

    def _decoder_sweep_ok(self, ws):
        ok = getattr(ws, "_sweep_ok", None)
        if ok is None:
            ok = ws._sweep_ok = (ws.attn_fused is None and ops.device_exclusive()
                                 and ops.decoder_sweep_supported(self.rt, self.Ld, ws.B, ws.U, ws.T2, self.Hd, 2 * self.He)
                                 and torch.cuda.get_device_properties(ws.enc.device).multi_processor_count >= 256)
        return ok

    def _decoder_sweep(self, ws, training):
        """Every AttendAndSpeller step of a teacher-forced pass (las.py:368-377) in ONE launch (decoder_sweep.hip): same operands
        and the same saved tensors as U x _decoder_step."""
        p, B, U, Hd, He = self.store.p, ws.B, ws.U, self.Hd, self.He
        if getattr(ws, "dsweep_ws", None) is None:
            ws.dsweep_ws = ops.decoder_sweep_ws(Hd, 2 * He, ws.enc.device)
        rate = self.dropout if training else 0.0
        d = _lib.DecoderSweep()
        d.B, d.U, d.T2, d.Hd, d.D = B, U, ws.T2, Hd, 2 * He
        d.Kq, d.enc, d.s0, d.mask = ws.Kq.data_ptr(), ws.enc.data_ptr(), ws.s0.data_ptr(), ws.mask.data_ptr()
        d.h_init, d.c_init = ws.hin[0].data_ptr(), ws.cin[0].data_ptr()
        c0, c1 = self.dec_cells
        d.Wp0, d.KSt0, d.ks0_ctx, d.ks0_h = c0.Wp.data_ptr(), c0.geom.KSt, c0.geom.ks0[0], c0.geom.ks0[1]
        d.Wp1, d.KSt1, d.ks1_x, d.ks1_h = c1.Wp.data_ptr(), c1.geom.KSt, c1.geom.ks0[0], c1.geom.ks0[1]
        d.pre0 = ws.pre0.data_ptr()
        d.bias1 = p["attend_and_speller/decoder_layers/1/cell/bias"].data_ptr()
        d.tokmask = ws.tokmask.data_ptr()
        d.seed = self.seed.data_ptr() if rate > 0 else None
        d.drop_rate, d.drop_stream0, d.drop_stream_step = rate, R.STREAM_DEC, R.DEC_STREAMS_PER_STEP
        d.p, d.ctx, d.hin, d.cin = ws.p.data_ptr(), ws.ctx.data_ptr(), ws.hin.data_ptr(), ws.cin.data_ptr()
        d.y0, d.saved0, d.h0, d.c0 = ws.dec[0]["y"].data_ptr(), ws.dec[0]["saved"].data_ptr(), ws.dec[0]["h"].data_ptr(), ws.dec[0]["c"].data_ptr()
        d.y1, d.saved1 = ws.dec[1]["y"].data_ptr(), ws.dec[1]["saved"].data_ptr()
        ops.decoder_sweep_fwd(d, ws.dsweep_ws, getattr(self.store, "err_flag", None))

    def _decoder_sweep_bwd_ok(self, ws):
        ok = getattr(ws, "_sweep_bwd_ok", None)
        if ok is None:
            ok = ws._sweep_bwd_ok = (DECODER_SWEEP and DECODER_SWEEP_BWD and ws.attn_fused is None and ops.device_exclusive() and
                                     ops.decoder_sweep_bwd_supported(self.rt, self.Ld, ws.B, ws.U, ws.T2, self.Hd, 2 * self.He)
                                     and torch.cuda.get_device_properties(ws.enc.device).multi_processor_count >= 256)
        return ok

    def _decoder_sweep_bwd_buffers(self, ws):
        if getattr(ws, "dsweep_bwd_ws", None) is None:
            ws.dsweep_bwd_ws = ops.decoder_sweep_bwd_ws(self.Hd, 2 * self.He, ws.enc.device)
            for j in range(self.Ld):
                ws.dec[j]["ds"] = torch.empty_like(ws.dec[j]["saved"])
        return ws.dsweep_bwd_ws

    def _decoder_sweep_bwd(self, ws):
        """The decoder loop of backward_decoder (las.py:282-288 differentiated, U-1 .. 0) in ONE launch: gate-sum gradients of both
        layers (out of place, ws.dec[j]["ds"]), score gradients ws.ds, context gradients ws.dctx, initial-state gradients
        ws.dhs / ws.dc_dec - what U x {cell backward x 2, context gradient, attention backward} leave behind."""
        p, B, U, Hd, He = self.store.p, ws.B, ws.U, self.Hd, self.He
        self._decoder_sweep_bwd_buffers(ws)
        rate = self.dropout
        dk = "attend_and_speller/decoder_layers/{}/cell/"
        d = _lib.DecoderSweepGrad()
        d.B, d.U, d.T2, d.Hd, d.D = B, U, ws.T2, Hd, 2 * He
        d.Kq, d.enc, d.p, d.ctx = ws.Kq.data_ptr(), ws.enc.data_ptr(), ws.p.data_ptr(), ws.ctx.data_ptr()
        d.saved0, d.saved1 = ws.dec[0]["saved"].data_ptr(), ws.dec[1]["saved"].data_ptr()
        d.cin, d.c0, d.tokmask = ws.cin.data_ptr(), ws.dec[0]["c"].data_ptr(), ws.tokmask.data_ptr()
        d.dy1, d.dy1_ld = ws.dyd.data_ptr(), ws.dyd.stride(0)
        d.U1, d.W1 = p[dk.format(1) + "recurrent_kernel"].data_ptr(), p[dk.format(1) + "kernel"].data_ptr()
        d.U0, d.W0 = p[dk.format(0) + "recurrent_kernel"].data_ptr(), p[dk.format(0) + "kernel"].data_ptr()
        d.seed = self.seed.data_ptr() if rate > 0 else None
        d.drop_rate, d.drop_stream0, d.drop_stream_step = rate, R.STREAM_DEC, R.DEC_STREAMS_PER_STEP
        d.ds0, d.ds1 = ws.dec[0]["ds"].data_ptr(), ws.dec[1]["ds"].data_ptr()
        d.de, d.dctx, d.dh_init, d.dc_init = ws.ds.data_ptr(), ws.dctx.data_ptr(), ws.dhs.data_ptr(), ws.dc_dec.data_ptr()
        d.de_sum = ws.ds0.data_ptr()                # sum_i de[i] = the gradient wrt s0, for free (otherwise a product with a column of ones)
        ops.decoder_sweep_bwd(d, ws.dsweep_bwd_ws, getattr(self.store, "err_flag", None))

    # ------------------------------------------------------------------------------------------ forward
    def draw_teacher_forcing(self) -> bool:
        """las.py:366: one coin per batch, also at eval."""
        return self._py_rng.random() < self.teacher_forcing_rate

    def forward(self, audio, tokens, training=False, use_teacher_forcing: Optional[bool] = None):
        """audio f32 [B,T,F,C], tokens i32 [B,U] -> logits [B,U,V] (a view of the step-major buffer)."""
        ops._dev(audio, name="audio")
        ops._dev(tokens, torch.int32, "tokens")
        self._ensure_built(audio.shape[2], audio.shape[3])
        B, T = audio.shape[:2]
        U = tokens.shape[1]
        ws = self._workspace(B, T, U)
        ws.toks_T[:U].copy_(tokens.t())
        if use_teacher_forcing is None:
            use_teacher_forcing = self.draw_teacher_forcing()
        self.forward_ws(ws, audio.contiguous(), training, use_teacher_forcing)
        return ws.logits.view(U, B, self.V).permute(1, 0, 2)

    def forward_ws(self, ws, audio, training, use_teacher_forcing):
        """LAS.call (las.py:349-380) on a workspace whose ws.toks_T[:U] holds the step-major tokens."""
        U, B = ws.U, ws.B
        self._encode(ws, audio, training)
        self._attention_keys(ws)
        ws.training, ws.teacher = training, use_teacher_forcing
        if use_teacher_forcing:
            self._embed(ws, 0, U, training)
            if DECODER_SWEEP and self._decoder_sweep_ok(ws):
                self._decoder_sweep(ws, training)                     # all U steps in one launch
            else:
                for i in range(U):
                    self._decoder_step(ws, i, training)
            self._vocab(ws, 0, U, training)
        else:
            for i in range(U):
                if i > 0:   # las.py:372: feed back the arg-max of the previous step's logits
                    ops.argmax_rows(ws.logits[(i - 1) * B:i * B], ws.toks_T[i])
                self._embed(ws, i, 1, training)
                self._decoder_step(ws, i, training)
                self._vocab(ws, i, 1, training)
        return ws.logits

    def call(self, inputs: Tuple[torch.Tensor, torch.Tensor], training: Optional[bool] = None):
        audio, tokens = inputs
        return self.forward(audio, tokens.to(torch.int32), bool(training))

    # ------------------------------------------------------------------------------------------ training hooks
    def train_workspace(self, B, T, L):
        """Buffers for token rows of length L (= BOS ... EOS): decoder input L-1 steps (make_example)."""
        U = L - 1
        return self._workspace(B, T, U), torch.empty(U, B, dtype=torch.int32, device=self.device)

    def set_targets(self, ws, tokens, labels_T):
        """make_example (las.py:396-406): decoder input = tokens[:, :-1], target = tokens[:, 1:], stored step-major."""
        ws.toks_T[:ws.U].copy_(tokens[:, :-1].t())
        labels_T.copy_(tokens[:, 1:].t())

    # ------------------------------------------------------------------------------------------ backward
    def loss_and_grad(self, ws, labels_T, grad_scale=1.0):
        """Masked cross-entropy (measure.py:4-21) over ws.logits (overwritten with its gradient).
        labels_T: i32 [U, B] step-major.  ws.stats <- [loss, #correct, #kept]."""
        ops.fill(ws.stats, 0.0)
        ops.softmax_xent(ws.logits, labels_T.reshape(-1), ws.stats, self.pad_id, True, grad_scale)

    def backward_ws(self, ws, audio):
        """Back-propagate ws.logits (holding d loss / d logits) through the whole model, accumulating
        into store.grad (which the caller zeroes)."""
        for seg in self.backward_segments(ws, audio):
            seg()

    def backward_segments(self, ws, audio):
        """The backward pass as one callable per gradient bucket of the ParamStore, in the order the buckets complete (vocabulary
        layer, rest of the decoder side, encoder layers from the top down): the data-parallel step all-reduces bucket k while
        segment k+1 runs."""
        segs = [lambda: self.backward_vocab(ws), lambda: self.backward_decoder(ws)]
        for i in range(self.Le - 1, -1, -1):
            segs.append(lambda i=i: self.backward_encoder_layer(ws, audio, i))
        return segs

    def bucket_schedule(self):
        """Per backward segment, the gradient buckets that are complete when it ends.  With the overlap scheduler a stage's weight
        gradients run beside the NEXT stage's sweep, so every bucket completes one segment later and the last two together."""
        n = 2 + self.Le
        if self._ov.late_buckets:
            return [[]] + [[k] for k in range(n - 2)] + [[n - 2, n - 1]]
        # The vocabulary bucket (complete after segment 0) is held back until the decoder segment has been enqueued: the decoder's
        # backward sweep is one workgroup per compute unit at the register limit - an RCCL kernel that is resident when it starts keeps
        # some of its workgroups off the chip until the collective has finished (the start handshake waits, sweep_common.h: safe, but
        # serialised).  Behind the decoder segment both decoder-side buckets travel while the encoder sweeps run, which keep a
        # quarter of the chip free (asr_sweep_capacity)
        return [[], [0, 1]] + [[k] for k in range(2, n)]

    def backward_vocab(self, ws):
        """Vocabulary projection (las.py:291): its weight gradient and the gradient flowing into the decoder chain."""
        assert ws.training, "backward needs a training-mode forward"
        p, g = self.store.p, self.store.g
        B, U, Hd = ws.B, ws.U, self.Hd
        rate = self.dropout
        mk, _ = self._drops(ws, 0, U, True)
        y_last = ws.dec[-1]["y"].view(U * B, Hd)
        src = ws.yd if rate > 0 else y_last
        # (the weight gradient - 16 M outputs nobody downstream reads - is released beside the decoder's backward sweep)
        self._ov.defer(lambda: dense_bwd(src, None, ws.logits, g["attend_and_speller/feedforward/kernel"], g["attend_and_speller/feedforward/bias"]))
        dense_bwd(src, p["attend_and_speller/feedforward/kernel"], ws.logits, None, None, ws.dyd)
        if rate > 0:
            ops.dropout_rows(ws.dyd, ws.dyd, self.seed, mk(1, Hd))
        if getattr(self, "bucket_sync", False):
            self._ov.join_all()

    def backward_decoder(self, ws):
        p, g = self.store.p, self.store.g
        B, U, T2, He, Hd, rt, V = ws.B, ws.U, ws.T2, self.He, self.Hd, self.rt, self.V
        rate = self.dropout
        mk, _ = self._drops(ws, 0, U, True)
        seed = self.seed if rate > 0 else None
        # ---- decoder steps in reverse (las.py:282-288).  Each cell hands ds (gradient wrt its gate sums,
        # written over its saved activations) to the cells that fed it; `ws.ddirect` carries the part of
        # dh that bypasses the gates (pad-token rows, GRU z*dh) along the single state chain.
        swept = self._decoder_sweep_bwd_ok(ws)
        ov = self._ov
        if swept:                                                     # all U steps in one launch (decoder_sweep_bwd.hip), the vocabulary
            gate = ops.sweep_diag_words(self._decoder_sweep_bwd_buffers(ws), decoder=True)
            ov.beside(lambda: self._decoder_sweep_bwd(ws), gate)      # layer's weight gradient beside it
        else:
            ov.flush(join=False)
            if rt == "lstm":
                ops.fill(ws.dc_dec, 0.0)
            ops.fill(ws.ddirect, 0.0)
            last = self.Ld - 1
            enc3, Kq3 = ws.enc.view(B, T2, 2 * He), ws.Kq.view(B, T2, Hd)
            dk = "attend_and_speller/decoder_layers/{}/cell/"
            W0 = p[dk.format(0) + "kernel"]
            for i in range(U - 1, -1, -1):
                base = R.STREAM_DEC + R.DEC_STREAMS_PER_STEP * i
                for j in range(last, -1, -1):
                    h_in, c_in, h_out, c_out = self._cell_states(ws, j, i)
                    st = _lib.RnnStepBwd()
                    st.n_units = Hd
                    if j < last:      # state and output both feed layer j+1 of the same step
                        dn = ws.dec[j + 1]["saved"][i]
                        st.srcA = ops.back_src(dn, p[dk.format(j + 1) + "recurrent_kernel"], rt, Hd, "rec")
                        st.srcB = ops.back_src(dn, p[dk.format(j + 1) + "kernel"], rt, Hd, "input", (rate, base + 2 + j + 1, Hd, 0))
                    else:             # last layer: state feeds layer 0 + attention of step i+1; output feeds Dense(V)
                        if i < U - 1:
                            st.srcA = ops.back_src(ws.dec[0]["saved"][i + 1], p[dk.format(0) + "recurrent_kernel"], rt, Hd, "rec")
                            st.addA, st.addA_ld = ws.dh_attn.data_ptr(), ws.dh_attn.stride(0)
                        dyl = ws.dyd[i * B:(i + 1) * B]
                        st.addB, st.addB_ld = dyl.data_ptr(), dyl.stride(0)
                    st.direct, st.direct_ld = ws.ddirect.data_ptr(), ws.ddirect.stride(0)
                    if rt == "lstm":
                        st.dc, st.dc_ld = ws.dc_dec.data_ptr(), ws.dc_dec.stride(0)
                        st.c_prev, st.c_prev_ld = c_in.data_ptr(), c_in.stride(0)
                        st.c_out, st.c_out_ld = c_out.data_ptr(), c_out.stride(0)
                    st.mask, st.mask_ld = ws.tokmask[i].data_ptr(), 1
                    sv = ws.dec[j]["saved"][i]
                    st.saved, st.saved_ld = sv.data_ptr(), sv.stride(0)
                    st.dslots, st.dslots_ld = sv.data_ptr(), sv.stride(0)
                    st.h_prev, st.h_prev_ld = h_in.data_ptr(), h_in.stride(0)
                    ops.rnn_cell_bwd(rt, B, [st], seed)
                # context gradient = layer 0's input gradient over the context rows of its kernel, through its input dropout
                lin = _lib.RnnStepBwd()
                lin.n_units = 2 * He
                lin.srcB = ops.back_src(ws.dec[0]["saved"][i], W0[Hd:], rt, Hd, "input", (rate, base + 2, Hd + 2 * He, Hd))
                lin.out, lin.out_ld = ws.dctx[i].data_ptr(), ws.dctx[i].stride(0)
                ops.rnn_cell_bwd(rt, B, [lin], seed)
                if ws.attn_fused is not None:
                    ops.attn_fused_bwd(ws.dctx[i], ws.p[i], Kq3, enc3, ws.attn_fused, ws.ds[i], ws.dh_attn, accumulate=False)
                else:
                    ops.attn_step_bwd(ws.dctx[i], ws.p[i], Kq3, enc3, ws.dp, ws.ds[i], ws.dh_attn, accumulate=False,
                                      images=getattr(ws, "attn_images", None))
            # gradient wrt the decoder's initial states (= listener state projections)
            lin = _lib.RnnStepBwd()
            lin.n_units = Hd
            lin.srcA = ops.back_src(ws.dec[0]["saved"][0], p[dk.format(0) + "recurrent_kernel"], rt, Hd, "rec")
            lin.addA, lin.addA_ld = ws.dh_attn.data_ptr(), ws.dh_attn.stride(0)
            lin.direct, lin.direct_ld = ws.ddirect.data_ptr(), ws.ddirect.stride(0)
            lin.out, lin.out_ld = ws.dhs.data_ptr(), ws.dhs.stride(0)
            ops.rnn_cell_bwd(rt, B, [lin], None)
        # ---- decoder weight gradients, batched over steps: nothing downstream reads them before Adam - released beside the top
        # encoder layer's backward sweep (layers.Overlap)
        def decoder_weight_grads():
            for j in range(self.Ld):
                pre = f"attend_and_speller/decoder_layers/{j}/cell/"
                ds2 = (ws.dec[j]["ds"] if swept else ws.dec[j]["saved"]).view(U * B, -1)
                gW, gU, gb = g[pre + "kernel"], g[pre + "recurrent_kernel"], g[pre + "bias"]
                if j == 0:
                    hprev = ws.hin[:U].view(U * B, Hd)
                    emb = ws.emb.view(U * B, Hd)
                    xin = ws.ctx.view(U * B, 2 * He)
                    if rate > 0:
                        xd = ws.xdrop[:, :2 * He]
                        ops.dropout_rows(xin, xd, self.seed, mk(2, Hd + 2 * He, Hd))
                        xin = xd
                    cell_param_grads(rt, Hd, emb, hprev, ds2, gW[:Hd], gU, gb)
                    cell_param_grads(rt, Hd, xin, None, ds2, gW[Hd:], None, None)
                    cell_input_grad(rt, Hd, ds2, p[pre + "kernel"][:Hd], ws.demb)
                    ops.embedding_bwd(g["attend_and_speller/embedding/embeddings"], ws.toks_T[:U], ws.demb, seed, mk(0, Hd), mk(2, Hd + 2 * He))
                else:
                    hprev = ws.dec[j - 1]["h"].view(U * B, Hd)
                    xin = ws.dec[j - 1]["y"].view(U * B, Hd)
                    if rate > 0:
                        xd = ws.xdrop[:, :Hd]
                        ops.dropout_rows(xin, xd, self.seed, mk(2 + j, Hd))
                        xin = xd
                    cell_param_grads(rt, Hd, xin, hprev, ds2, gW, gU, gb)
        ov.defer(decoder_weight_grads)
        # ---- attention: batched key-side gradients (las.py:46-59, hoisted form); the products that lead to d enc first
        a = "attend_and_speller/attention/"
        Wq, bq, Wk = p[a + "query_weight/kernel"], p[a + "query_weight/bias"], p[a + "key_weight/kernel"]
        ds_b = ws.ds.permute(1, 0, 2)                      # [B, U, T2] views of the step-major buffers
        hin_b = ws.hin[:U].permute(1, 0, 2)
        ops.gemm(ds_b, hin_b, ws.dKq.view(B, T2, Hd), trans_a=True)                       # dKq[b] = ds[b]^T hin[b]
        if not swept:                                      # (the backward sweep leaves ds0[b,t] = sum_i ds[b,i,t] behind)
            ops.gemm(ds_b, ws.ones_u.expand(B, U, 1), ws.ds0.view(B, T2, 1), trans_a=True)
        ops.gemm(ws.p.permute(1, 0, 2), ws.dctx.permute(1, 0, 2), ws.denc.view(B, T2, 2 * He), trans_a=True)  # p^T dctx
        ops.gemm(ws.dKq, Wq, ws.dK)                                                       # dK = dKq Wq
        ops.rank1_add(ws.dK, ws.ds0.view(-1), bq)                                         #    + ds0 (x) bq
        dense_bwd(ws.enc, Wk, ws.dK, None, None, ws.denc, dx_accumulate=True)             # d enc += dK Wk^T

        def attention_weight_grads():
            ops.gemm(ws.dKq, ws.K, g[a + "query_weight/kernel"], trans_a=True, accumulate=1, split_k=max(1, (B * T2) // 256))
            ops.colsum_weighted(ws.K, ws.ds0.view(-1), g[a + "query_weight/bias"])       # d bq += K^T ds0
            dense_bwd(ws.enc, None, ws.dK, g[a + "key_weight/kernel"], g[a + "key_weight/bias"])
        ov.defer(attention_weight_grads)
        # ---- listener state projections (las.py:196-202): the gradients wrt the encoder's final states first
        nst = 2 if rt == "lstm" else 1
        fin = ws.final_states
        projs = (("hidden_states_proj", ws.dhs), ("cell_states_proj", ws.dc_dec))[:nst]
        for k, (name, dsrc) in enumerate(projs):
            W = p[f"listener/{name}/kernel"]
            for d in range(2):
                dst = ws.dfin_h[d] if k == 0 else ws.dc_enc[d]
                ops.gemm(dsrc, W[d * He:(d + 1) * He], dst, trans_b=True)

        def state_proj_weight_grads():
            for k, (name, dsrc) in enumerate(projs):
                gW, gb = g[f"listener/{name}/kernel"], g[f"listener/{name}/bias"]
                ops.colsum(dsrc, gb)
                for d in range(2):
                    ops.gemm(fin[d * nst + k], dsrc, gW[d * He:(d + 1) * He], trans_a=True, accumulate=1)
        ov.defer(state_proj_weight_grads)
        if getattr(self, "bucket_sync", False):
            ov.join_all()

    def backward_encoder_layer(self, ws, audio, i):
        """Encoder layer i backwards (las.py:190-193): BatchNorm+ReLU, projection, BiRNN; the convolutions follow layer 0."""
        p, g = self.store.p, self.store.g
        B, T2, He = ws.B, ws.T2, self.He
        rate = self.dropout
        top = i == self.Le - 1
        da = ws.denc if top else (ws.dx[(i + 1) & 1]).view(B * T2, -1)
        dfin = list(ws.dfin_h) if top else ws.dfin_next
        l, lw = self.enc_layers[i], ws.layers[i]
        bn = f"listener/batch_norm/{i}/"
        ov, dz = self._ov, ws.dz[i]
        ops.bn_bwd(lw["z"], lw["a"], da, lw["mean"], lw["rstd"], p[bn + "gamma"], dz, g[bn + "gamma"], g[bn + "beta"], ws.bn_ws, relu=True)
        y2 = lw["rnn"]["y"].view(B * T2, 2 * He)
        # the projection's weight / bias gradients wait for this layer's sweep (beside it, with whatever the stage before left pending:
        # the decoder's or the layer above's weight gradients); only its input gradient is on the critical path
        ov.defer(lambda: dense_bwd(y2, None, dz, g[f"listener/projection/{i}/kernel"], g[f"listener/projection/{i}/bias"]))
        dense_bwd(y2, p[f"listener/projection/{i}/kernel"], dz, None, None, ws.dy.view(B * T2, 2 * He))
        dx = ws.dx0 if i == 0 else ws.dx[i & 1].view(B, T2, 2 * He)
        ws.dfin_next = l.backward(lw["rnn"], ws.dy, dfin, ws.dc_enc, dx, overlap=ov)   # this layer's dW / dU / db: beside the next sweep
        if i > 0:
            if getattr(self, "bucket_sync", False):
                ov.join_all()             # data parallel: what ran beside this layer's sweep is a complete bucket when the segment ends
            return
        ov.flush(join=False)               # no sweep left: the bottom layer's weight gradients run beside the convolutions' backward pass
        # ---- convolutions (las.py:183-184)
        if rate > 0:
            ops.dropout_flat(ws.dx0, self.seed, R.STREAM_CONV2_DROP, rate)
        dy2 = ws.dx0.view(B, T2, self.F2, 32)
        # the chain the update waits for is dropout -> conv2's input gradient -> dropout -> conv1's filter gradient; conv2's filter /
        # bias gradients and conv1's bias gradient run beside it (round 4, from the step's timeline: the seven kernels were one serial
        # run of 400 us at the end of the step while the other stream sat idle)
        ov.defer(lambda: (ops.conv2d_bwd_filter(ws.c1, dy2, g["listener/conv2/kernel"], 2), ops.colsum(dy2.view(-1, 32), g["listener/conv2/bias"])))
        ops.conv2d_bwd_data(dy2, p["listener/conv2/kernel"], ws.dc1, 2)
        if rate > 0:
            ops.dropout_flat(ws.dc1, self.seed, R.STREAM_CONV1_DROP, rate)
        ov.defer(lambda: ops.colsum(ws.dc1.view(-1, 32), g["listener/conv1/bias"]))
        ops.conv2d_bwd_filter(audio, ws.dc1, g["listener/conv1/kernel"], 2)
        ov.flush()                         # (flush, not join_all: in "beside" mode defer() only queues, and nothing is left to run beside)

    # ------------------------------------------------------------------------------------------ reference API
    def get_loss_fn(self):
        return SparseCategoricalCrossentropy(self.pad_id)

    def get_metrics(self):
        return [SparseCategoricalAccuracy(self.pad_id)]

    @staticmethod
    def get_batching_shape(audio_pad_length: Optional[int], token_pad_length: Optional[int], frequency_dim: int,
                           feature_dim: int):
        if token_pad_length is not None:
            token_pad_length = token_pad_length - 1
        return (([audio_pad_length, frequency_dim, feature_dim], [token_pad_length]), [token_pad_length])

    @staticmethod
    def make_example(audio, tokens):
        """las.py:396-406: ((audio, tokens[:-1]), tokens[1:])."""
        return (audio, tokens[:-1]), tokens[1:]
