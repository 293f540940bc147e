"""Oracle (TEST INFRASTRUCTURE - never imported by the product): greedy decoding of
speech_recognition/search.py restated on the float64 oracle models.

greedy_las    follows LAS_Searcher.greedy_search (search.py:23-81) line by line (while-loop, top_k of
              log_softmax, is_ended / sequence_lengths / log_perplexity updates, final pow).
greedy_ds2    follows DeepSpeechSearcher.greedy_search (search.py:223-252): blank moved to the last class
              and masked at its old index, log_softmax, [TF-sem] tf.nn.ctc_greedy_decoder with
              merge_repeated=True (emit class c at frame t iff c != blank and c != class at t-1),
              tf.sparse.to_dense zero padding, probability = exp(sum_t max log-prob).
Parity status: unpinned (no TensorFlow here); the reference's tests/test_search.py checks shapes only.
"""
import numpy as np
import torch

from . import deepspeech2 as DS
from . import las as LAS


def greedy_las(p, cfg, audio, max_token_length, bos_id, eos_id, pad_id=0):
    enc, mask, states, _ = LAS.listener(p, cfg, audio, False)
    B = audio.shape[0]
    decoder_input = torch.full((B, 1), bos_id, dtype=torch.int64)
    log_perplexity = torch.zeros(B, 1, dtype=enc.dtype)
    sequence_lengths = torch.full((B, 1), max_token_length, dtype=torch.int64)
    is_ended = torch.zeros(B, 1, dtype=torch.bool)
    step = 0
    while decoder_input.shape[1] < max_token_length and not bool(is_ended.all()):
        output, states, _ = LAS.attend_and_speller(p, cfg, enc, decoder_input[:, -1], mask, states, False, step=step)
        output = torch.log_softmax(output, dim=1)
        log_probs, new_tokens = output.max(dim=1, keepdim=True)      # top_k(k=1); ties -> lowest index
        new_tokens = output.argmax(dim=1, keepdim=True)
        log_perplexity = torch.where(is_ended, log_perplexity, log_perplexity + log_probs)
        new_tokens = torch.where(is_ended, torch.full_like(new_tokens, pad_id), new_tokens)
        is_ended = is_ended | (new_tokens == eos_id)
        sequence_lengths = torch.where(new_tokens == eos_id, torch.full_like(sequence_lengths, decoder_input.shape[1] + 1), sequence_lengths)
        decoder_input = torch.cat([decoder_input, new_tokens], dim=1)
        step += 1
    perplexity = torch.pow(torch.exp(log_perplexity), -1.0 / sequence_lengths.to(enc.dtype))[:, 0]
    return decoder_input, perplexity


def greedy_ds2(p, cfg, audio, blank_index, mask_mode="intended"):
    output = DS.ds2_forward(p, cfg, audio, training=False, mask_mode=mask_mode)          # [B, T', V]
    B, T, V = output.shape
    output = torch.cat([output, output[:, :, blank_index:blank_index + 1]], dim=2)
    m = torch.zeros(V + 1, dtype=output.dtype)
    m[blank_index] = -1e9
    output = torch.log_softmax(output + m, dim=2)
    best = output.argmax(dim=2)                                                           # [B, T'], V = blank
    neg_sum = -output.max(dim=2).values.sum(dim=1)
    rows = []
    for b in range(B):
        prev, row = -1, []
        for t in range(T):
            c = int(best[b, t])
            if c != V and c != prev:
                row.append(c)
            prev = c
        rows.append(row)
    width = max((len(r) for r in rows), default=0)
    tokens = np.zeros((B, width), np.int32)
    for b, r in enumerate(rows):
        tokens[b, :len(r)] = r
    return tokens, torch.exp(-neg_sum), best
