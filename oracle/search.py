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


def _first_eos_length(row, eos_id):
    """search.py:107-111 `_to_sequence_lengths`: index of the first EOS + 1, else the row's size."""
    idx = (row == eos_id).nonzero()
    return int(row.numel()) if idx.numel() == 0 else int(idx.min()) + 1


def _stable_top_k(values, k):
    """tf.math.top_k ordering: descending value, ties -> lowest index."""
    order = sorted(range(values.numel()), key=lambda i: (-float(values[i]), i))[:k]
    return torch.tensor(order, dtype=torch.int64)


def beam_las(p, cfg, audio, max_token_length, bos_id, eos_id, pad_id, beam_size, alpha=1, beta=32):
    """LAS_Searcher.beam_search (search.py:83-209) line by line, including what the reference does NOT do:
    the decoder states are never gathered by the chosen parents (the loop returns the states of the rows as
    they were, search.py:169) and an ended hypothesis spawns beam_size equal-scored children."""
    enc, mask, states, _ = LAS.listener(p, cfg, audio, False)
    B = audio.shape[0]
    decoder_input = torch.full((B, 1), bos_id, dtype=torch.int64)
    log_perplexity = torch.zeros(B, 1, dtype=torch.float32)

    def has_eos(d):
        return (d == eos_id).any(dim=-1)

    step = 0
    while decoder_input.shape[1] < max_token_length and bool((~has_eos(decoder_input)).any()):
        output, states, _ = LAS.attend_and_speller(p, cfg, enc, decoder_input[:, -1], mask, states, False, step=step)
        step += 1
        output = torch.log_softmax(output, dim=1).to(torch.float32)
        rows = output.shape[0]
        idx = torch.stack([_stable_top_k(output[r], beam_size) for r in range(rows)])        # [rows, beam]
        log_probs = torch.gather(output, 1, idx).reshape(B, -1)
        new_tokens = idx.reshape(-1, 1)
        is_end = has_eos(decoder_input).repeat_interleave(beam_size, dim=0).reshape(B, -1)
        log_probs = torch.where(is_end, torch.zeros_like(log_probs), log_probs)
        log_probs = log_probs + log_perplexity.repeat_interleave(beam_size, dim=1)
        if decoder_input.shape[1] == 1:
            enc = enc.repeat_interleave(beam_size, dim=0)
            mask = mask.repeat_interleave(beam_size, dim=0)
            states = tuple(s.repeat_interleave(beam_size, dim=0) for s in states)
            decoder_input = torch.cat([torch.full((B * beam_size, 1), bos_id, dtype=torch.int64), new_tokens], dim=1)
            log_perplexity = log_probs
            continue
        cand = torch.cat([decoder_input.repeat_interleave(beam_size, dim=0), new_tokens], dim=1)
        cand = cand.reshape(B, beam_size * beam_size, -1)
        lengths = torch.tensor([[_first_eos_length(cand[b, c], eos_id) for c in range(cand.shape[1])] for b in range(B)])
        penalty = torch.pow((1 + lengths).double() / (1 + beta), float(alpha)).to(torch.float32)   # int / int -> float64 in TF
        score = log_probs * penalty
        top = torch.stack([_stable_top_k(score[b], beam_size) for b in range(B)])              # [B, beam]
        decoder_input = torch.stack([cand[b, top[b]] for b in range(B)]).reshape(B * beam_size, -1)
        log_perplexity = torch.gather(log_probs, 1, top)
    decoder_input = decoder_input.reshape(B, -1, decoder_input.shape[-1])
    nb = decoder_input.shape[1]
    lengths = torch.tensor([[_first_eos_length(decoder_input[b, j], eos_id) for j in range(nb)] for b in range(B)])
    keep = torch.arange(decoder_input.shape[2])[None, None, :] < lengths[..., None]
    decoder_input = torch.where(keep, decoder_input, torch.full_like(decoder_input, pad_id))
    perplexity = torch.pow(torch.exp(log_perplexity), (-1.0 / lengths.double()).to(torch.float32))
    return decoder_input, perplexity


_LOG_ZERO = float("-inf")


def _lse(a, b):
    if a == _LOG_ZERO:
        return b
    if b == _LOG_ZERO:
        return a
    m = max(a, b)
    return m + float(np.log1p(np.exp(-abs(a - b))))


class _Beam:
    __slots__ = ("parent", "label", "children", "old", "new")

    def __init__(self, parent, label):
        self.parent, self.label, self.children = parent, label, None
        self.old = [_LOG_ZERO, _LOG_ZERO, _LOG_ZERO]             # total, blank, label
        self.new = [_LOG_ZERO, _LOG_ZERO, _LOG_ZERO]

    def active(self):
        return self.new[0] != _LOG_ZERO


def ctc_beam_search(log_probs, beam_width, top_paths=1):
    """[TF-sem] tf.nn.ctc_beam_search_decoder for one utterance: the prefix beam search of TensorFlow's
    CTCBeamSearchDecoder (core/util/ctc/ctc_beam_search.h `Step` / `TopPaths`, TF 2.x; merge_repeated=False, no label
    selection, the default scorer): inputs [T, C] with the blank as class C-1.  Per frame: every beam entry keeps
    (p_blank, p_label, p_total) in log space; surviving entries are extended in place
    (label: repeat needs the parent's blank mass, otherwise the parent's total), then every still-competitive
    entry grows its C-1 children (a child repeating the entry's label starts from the entry's blank mass) and
    the best beam_width entries by p_total survive.  Returns ([paths], [log p_total])."""
    x = np.asarray(log_probs, np.float64)
    T, C = x.shape
    blank = C - 1
    root = _Beam(None, -1)
    root.new = [0.0, 0.0, _LOG_ZERO]
    leaves = [root]
    for t in range(T):
        row = x[t]
        mx = float(row.max())
        norm = mx + float(np.log(np.exp(row - mx).sum()))
        branches = sorted(leaves, key=lambda e: -e.new[0])
        leaves = []
        for b in branches:
            b.old = list(b.new)
        for b in branches:
            if b.parent is not None:
                if b.parent.active():
                    prev = b.parent.old[1] if b.label == b.parent.label else b.parent.old[0]
                    b.new[2] = _lse(b.new[2], prev)
                b.new[2] += row[b.label] - norm
            b.new[1] = b.old[0] + row[blank] - norm
            b.new[0] = _lse(b.new[1], b.new[2])
            leaves.append(b)

        def bottom():
            return min(leaves, key=lambda e: e.new[0])

        def is_candidate(total):
            return total > _LOG_ZERO and (len(leaves) < beam_width or total > bottom().new[0])

        for b in branches:
            if not is_candidate(b.old[0]):
                continue
            if b.children is None:
                b.children = [_Beam(b, c) for c in range(C - 1)]
            for c in b.children:
                if c.active():
                    continue
                c.new[1] = _LOG_ZERO
                prev = b.old[1] if c.label == b.label else b.old[0]
                c.new[2] = row[c.label] - norm + prev
                c.new[0] = c.new[2]
                if is_candidate(c.new[0]):
                    if len(leaves) == beam_width:
                        worst = bottom()
                        worst.new = [_LOG_ZERO, _LOG_ZERO, _LOG_ZERO]
                        leaves.remove(worst)
                    leaves.append(c)
                else:
                    c.old = [_LOG_ZERO, _LOG_ZERO, _LOG_ZERO]
                    c.new = [_LOG_ZERO, _LOG_ZERO, _LOG_ZERO]
    best = sorted(leaves, key=lambda e: -e.new[0])[:top_paths]
    paths, scores = [], []
    for e in best:
        seq = []
        n = e
        while n.parent is not None:
            seq.append(n.label)
            n = n.parent
        paths.append(seq[::-1])
        scores.append(e.new[0])
    return paths, scores


def beam_ds2(p, cfg, audio, blank_index, beam_size, mask_mode="intended"):
    """DeepSpeechSearcher.beam_search (search.py:254-285): tokens [B, 1, L] (top_paths defaults to 1), probability [B, 1]."""
    output = DS.ds2_forward(p, cfg, audio, training=False, mask_mode=mask_mode)
    B, T, V = output.shape
    output = torch.cat([output, output[:, :, blank_index:blank_index + 1]], dim=2)
    m = torch.zeros(V + 1, dtype=output.dtype)
    m[blank_index] = -1e9
    output = torch.log_softmax(output + m, dim=2).to(torch.float32)
    rows, logp = [], []
    for b in range(B):
        paths, scores = ctc_beam_search(output[b].numpy(), beam_size, 1)
        rows.append(paths[0])
        logp.append(scores[0])
    width = max((len(r) for r in rows), default=0)
    tokens = np.zeros((B, 1, width), np.int32)
    for b, r in enumerate(rows):
        tokens[b, 0, :len(r)] = r
    return tokens, np.exp(np.asarray(logp))[:, None]
