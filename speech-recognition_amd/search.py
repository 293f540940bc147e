"""Greedy decoding of speech_recognition/search.py on the HIP kernels.

`LAS_Searcher.greedy_search` (search.py:23-81) and `DeepSpeechSearcher.greedy_search` (search.py:223-252)
keep the reference's constructor arguments and return values.  The LAS loop runs the encoder and the
loop-invariant attention keys once, then one embedding / attention / LSTM / vocabulary step per token with
the arg-max, end-of-sentence bookkeeping and log-perplexity accumulated on the device
(asr_greedy_update); the host looks at the `ended` flags only every `check_every` steps.  Beam search
(search.py:83-209, 254-285) is not part of this build.
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
        tokens = ws.toks_T[:done + 1].t().contiguous()
        perplexity = torch.exp(-log_ppl / seq_len.to(torch.float32))      # pow(exp(log_ppl), -1 / sequence_lengths)
        return tokens, perplexity

    def beam_search(self, *args, **kwargs):
        raise NotImplementedError("beam search is outside this build's scope (SURVEY.md section 8 f4)")


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
        out = tokens[:, :width].contiguous()
        probability = torch.exp(-neg_sum)
        if return_alignment:
            return out, probability, best.view(B, T2)
        return out, probability

    def beam_search(self, *args, **kwargs):
        raise NotImplementedError("beam search is outside this build's scope (SURVEY.md section 8 f4)")
