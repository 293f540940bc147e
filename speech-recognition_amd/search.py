"""Greedy decoding of speech_recognition/search.py on the HIP kernels.

`LAS_Searcher.greedy_search` (search.py:23-81) and `DeepSpeechSearcher.greedy_search` (search.py:223-252)
keep the reference's constructor arguments and return values.  The LAS loop runs the encoder and the
loop-invariant attention keys once, then one embedding / attention / LSTM / vocabulary step per token with
the arg-max, end-of-sentence bookkeeping and log-perplexity accumulated on the device
(asr_greedy_update); the host looks at the `ended` flags only every `check_every` steps.

`LAS_Searcher.beam_search` (search.py:83-209) runs the same per-token step on B * beam rows with the top-k,
length penalty, stable selection and history gather on the device (asr_beam_topk / asr_beam_select);
`DeepSpeechSearcher.beam_search` (search.py:254-285) takes the masked log_softmax on the device
(asr_ctc_log_softmax) and walks the CTC prefix tree on the host (asr_ctc_beam_search) - TensorFlow's decoder op
is a CPU op too.
"""
from typing import Tuple

import ctypes as C

import torch

from . import ops
from ._lib import check, load


def _p(t):
    return C.c_void_p(t.data_ptr())


class LAS_Searcher:
    """Provide search functions for LAS model (search.py:6-21)."""

    def __init__(self, model, max_token_length: int, bos_id: int, eos_id: int, pad_id: int = 0, check_every: int = 8):
        self.model = model
        self.max_token_length = max_token_length
        self.bos_id, self.eos_id, self.pad_id = bos_id, eos_id, pad_id
        self.check_every = max(1, int(check_every))

    def greedy_search(self, audio_input: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """audio_input f32 [B, T, F, C] -> (tokens i32 [B, L] starting with BOS, perplexity f32 [B])."""
        m = self.model
        ops._dev(audio_input, name="audio_input")
        m._ensure_built(audio_input.shape[2], audio_input.shape[3])
        B, T = audio_input.shape[:2]
        steps = self.max_token_length - 1                      # search.py:37: loop while len(decoder_input) < max
        dev = audio_input.device
        log_ppl = torch.zeros(B, device=dev)
        seq_len = torch.full((B,), self.max_token_length, dtype=torch.int32, device=dev)
        ended = torch.zeros(B, dtype=torch.uint8, device=dev)
        if steps <= 0:
            return torch.full((B, 1), self.bos_id, dtype=torch.int32, device=dev), torch.ones(B, device=dev)
        ws = m._workspace(B, T, steps)
        m._encode(ws, audio_input.contiguous(), False)
        m._attention_keys(ws)
        ws.training, ws.teacher = False, False
        ws.toks_T[0].fill_(self.bos_id)
        lib, stream = load(), ops._stream()
        for i in range(steps):
            m._embed(ws, i, 1, False)
            m._decoder_step(ws, i, False)
            m._vocab(ws, i, 1, False)
            check(lib.asr_greedy_update(_p(ws.logits[i * B:(i + 1) * B]), ws.logits.stride(0), B, m.V, i + 1, self.eos_id, self.pad_id,
                                        _p(ws.toks_T[i + 1]), _p(ended), _p(log_ppl), _p(seq_len), stream))
            if (i + 1) % self.check_every == 0 and i + 1 < steps and bool(ended.all().item()):
                break
        # the reference leaves its loop as soon as every row has ended: keep the tokens up to the last EOS
        done = int(seq_len.max().item()) - 1 if bool(ended.all().item()) else steps
        m.raise_on_sweep_timeout()                             # (the encoder ran as one-launch sweeps; the host has just synchronised)
        tokens = ws.toks_T[:done + 1].t().contiguous()
        perplexity = torch.exp(-log_ppl / seq_len.to(torch.float32))      # pow(exp(log_ppl), -1 / sequence_lengths)
        return tokens, perplexity

    def beam_search(self, audio_input: torch.Tensor, beam_size: int, alpha: float = 1, beta: int = 32,
                    reorder_states: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        """audio_input f32 [B, T, F, C] -> (tokens i32 [B, beam, L] padded with pad_id after the first EOS,
        perplexity f32 [B, beam]) (search.py:83-209).

        Faithful to the reference, including two things a reader may not expect: a hypothesis that has ended
        spawns `beam_size` equal-scored children (so it can fill the beam with copies of itself), and the decoder
        states are NOT re-ordered by the chosen parents (search.py:169 returns the rows' states as they are).
        `reorder_states=True` gathers the recurrent state by parent instead - not what the reference computes."""
        m = self.model
        ops._dev(audio_input, name="audio_input")
        m._ensure_built(audio_input.shape[2], audio_input.shape[3])
        B, T = audio_input.shape[:2]
        k = int(beam_size)
        if not 1 <= k <= 32:
            raise ValueError(f"beam_size must be in [1, 32], got {beam_size}")
        R, Lmax = B * k, self.max_token_length
        steps = Lmax - 1
        dev = audio_input.device
        if steps <= 0:
            return torch.full((B, 1, 1), self.bos_id, dtype=torch.int32, device=dev), torch.ones(B, 1, device=dev)
        ws0 = m._workspace(B, T, 1)
        m._encode(ws0, audio_input.contiguous(), False)
        ws = m._workspace(R, T, steps)
        T2 = ws.T2
        # search.py:142-147: every hypothesis of an utterance attends over the same encoder output
        enc = ws0.enc.view(B, T2, -1).repeat_interleave(k, 0)
        mask, h0, c0 = ws0.mask.repeat_interleave(k, 0), ws0.hin[0].repeat_interleave(k, 0), ws0.cin[0].repeat_interleave(k, 0)
        ws.enc.view(R, T2, -1).copy_(enc)
        ws.mask.copy_(mask)
        ws.hin[0].copy_(h0)
        ws.cin[0].copy_(c0)
        m._attention_keys(ws)
        ws.training, ws.teacher = False, False
        ws.toks_T.zero_()
        ws.toks_T[0].fill_(self.bos_id)
        hist = torch.zeros(2, R, Lmax, dtype=torch.int32, device=dev)
        hist[0, :, 0] = self.bos_id
        ppl = torch.zeros(2, R, device=dev)
        ended = torch.zeros(2, R, dtype=torch.uint8, device=dev)
        if self.bos_id == self.eos_id:
            ended[0].fill_(1)
        slen = torch.ones(2, R, dtype=torch.int32, device=dev)
        lp, tok = torch.empty(R, k, device=dev), torch.empty(R, k, dtype=torch.int32, device=dev)
        parent = torch.empty(R, dtype=torch.int32, device=dev)
        final_len = torch.ones(1, dtype=torch.int32, device=dev)
        lib, stream = load(), ops._stream()
        cur = 0
        for i in range(steps):
            m._embed(ws, i, 1, False)
            m._decoder_step(ws, i, False)
            m._vocab(ws, i, 1, False)
            logits = ws.logits[i * R:(i + 1) * R]
            check(lib.asr_beam_topk(_p(logits), ws.logits.stride(0), R, m.V, k, _p(lp), _p(tok), stream))
            nxt = 1 - cur
            check(lib.asr_beam_select(_p(lp), _p(tok), B, k, i + 1, Lmax, self.eos_id, float(alpha), float(beta), _p(hist[cur]), _p(ppl[cur]),
                                      _p(ended[cur]), _p(slen[cur]), _p(hist[nxt]), _p(ppl[nxt]), _p(ended[nxt]), _p(slen[nxt]),
                                      _p(ws.toks_T[i + 1]), _p(parent), _p(final_len), stream))
            cur = nxt
            if reorder_states and i > 0:
                rows = parent.long()
                ws.hin[i + 1].copy_(ws.hin[i + 1].index_select(0, rows))
                ws.cin[i + 1].copy_(ws.cin[i + 1].index_select(0, rows))
            if (i + 1) % self.check_every == 0 and i + 1 < steps and bool(ended[cur].all().item()):
                break
        L = int(final_len.item())
        m.raise_on_sweep_timeout()
        tokens = hist[cur, :, :L].reshape(B, k, L)
        lengths = torch.where(ended[cur].bool(), slen[cur], torch.full_like(slen[cur], L)).view(B, k)
        keep = torch.arange(L, device=dev)[None, None, :] < lengths[..., None]
        tokens = torch.where(keep, tokens, torch.full_like(tokens, self.pad_id))
        # search.py:207: pow(exp(log_perplexity), -1 / sequence_lengths), the exponent formed in float64
        perplexity = torch.pow(torch.exp(ppl[cur].view(B, k)), (-1.0 / lengths.double()).float())
        return tokens, perplexity


class DeepSpeechSearcher:
    """Provide search functions for DeepSpeech2 model (search.py:212-221)."""

    def __init__(self, model, blank_index: int):
        self.model = model
        self.blank_index = blank_index

    def greedy_search(self, audio_input: torch.Tensor, return_alignment: bool = False):
        """audio_input f32 [B, T, F, C] -> (tokens i32 [B, max decoded length] zero padded, probability f32 [B]).
        With return_alignment also the per-frame best path i32 [B, T'] (class V = blank) - the "CTC alignment"."""
        m = self.model
        logits = m(audio_input, training=False)                 # [B, T', V]
        B, T2, V = logits.shape
        dev = logits.device
        best = torch.empty(B * T2, dtype=torch.int32, device=dev)
        best_lp = torch.empty(B * T2, dtype=torch.float32, device=dev)
        tokens = torch.empty(B, T2, dtype=torch.int32, device=dev)
        lengths = torch.empty(B, dtype=torch.int32, device=dev)
        neg_sum = torch.empty(B, dtype=torch.float32, device=dev)
        flat = logits.reshape(B * T2, V)
        check(load().asr_ctc_greedy(_p(flat), flat.stride(0), B, T2, V, self.blank_index, _p(best), _p(best_lp), _p(tokens), _p(lengths),
                                    _p(neg_sum), ops._stream()))
        width = max(int(lengths.max().item()), 0)
        m.raise_on_sweep_timeout()
        out = tokens[:, :width].contiguous()
        probability = torch.exp(-neg_sum)
        if return_alignment:
            return out, probability, best.view(B, T2)
        return out, probability

    def beam_search(self, audio_input: torch.Tensor, beam_size: int, top_paths: int = 1, threads: int = 0):
        """audio_input f32 [B, T, F, C] -> (tokens i32 [B, top_paths, L] zero padded, probability f32 [B, top_paths])
        (search.py:254-285; the reference leaves top_paths at TensorFlow's default of 1)."""
        import os
        m = self.model
        logits = m(audio_input, training=False)                 # [B, T', V]
        B, T2, V = logits.shape
        dev = logits.device
        flat = logits.reshape(B * T2, V)
        lsm = torch.empty(B * T2, V + 1, dtype=torch.float32, device=dev)
        check(load().asr_ctc_log_softmax(_p(flat), flat.stride(0), B * T2, V, self.blank_index, _p(lsm), ops._stream()))
        host = lsm.cpu()                                        # the prefix tree is walked on the host
        m.raise_on_sweep_timeout()
        tokens = torch.empty(B, top_paths, T2, dtype=torch.int32)
        lengths = torch.empty(B, top_paths, dtype=torch.int32)
        log_prob = torch.empty(B, top_paths, dtype=torch.float32)
        if threads <= 0:
            threads = min(16, len(os.sched_getaffinity(0)))
        check(load().asr_ctc_beam_search(_p(host), B, T2, V + 1, None, int(beam_size), int(top_paths), _p(tokens), _p(lengths), _p(log_prob),
                                         int(threads)))
        width = max(int(lengths.max().item()), 0)
        return tokens[:, :, :width].contiguous().to(dev), torch.exp(log_prob).to(dev)
