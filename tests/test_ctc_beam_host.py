"""The host CTC prefix beam search (asr_ctc_beam_search, reference search.py:254-285 via tf.nn.ctc_beam_search_decoder)
against the oracle restatement and, with a beam wide enough to be exhaustive, against brute-force enumeration of
every alignment.  No GPU: the function runs on the host, as TensorFlow's op does."""
import itertools

import numpy as np
import pytest

from oracle import search as OS


def _run(x, width, top_paths, threads=1, seq_len=None):
    from speech_recognition_amd._lib import check, load
    B, T, C = x.shape
    x = np.ascontiguousarray(x, np.float32)
    tokens = np.full((B, top_paths, T), -7, np.int32)
    lengths = np.zeros((B, top_paths), np.int32)
    lp = np.zeros((B, top_paths), np.float32)
    sl = None if seq_len is None else np.ascontiguousarray(seq_len, np.int32)
    check(load().asr_ctc_beam_search(x.ctypes.data, B, T, C, None if sl is None else sl.ctypes.data, width, top_paths, tokens.ctypes.data,
                                     lengths.ctypes.data, lp.ctypes.data, threads))
    return tokens, lengths, lp


def _log_softmax(x):
    return x - np.log(np.exp(x).sum(-1, keepdims=True))


def test_wide_beam_is_exact():
    rng = np.random.default_rng(0)
    T, C = 5, 4
    x = _log_softmax(rng.normal(size=(1, T, C)) * 1.5).astype(np.float32)
    total = {}
    for path in itertools.product(range(C), repeat=T):
        lab, prev = [], -1
        for c in path:
            if c != C - 1 and c != prev:
                lab.append(c)
            prev = c
        total[tuple(lab)] = np.logaddexp(total.get(tuple(lab), -np.inf), sum(float(x[0, t, c]) for t, c in enumerate(path)))
    best = sorted(total.items(), key=lambda kv: -kv[1])[:4]
    tokens, lengths, lp = _run(x, 1000, 4)
    for i, (lab, score) in enumerate(best):
        assert tokens[0, i, :lengths[0, i]].tolist() == list(lab)
        assert abs(lp[0, i] - score) < 1e-4
        assert (tokens[0, i, lengths[0, i]:] == 0).all()


@pytest.mark.parametrize("B,T,C,W,tp", [(3, 20, 6, 4, 2), (2, 50, 30, 8, 3), (1, 7, 4, 100, 5), (2, 40, 200, 3, 1), (4, 30, 12, 1, 1)])
def test_matches_oracle(B, T, C, W, tp):
    rng = np.random.default_rng(B * 1000 + T)
    x = _log_softmax(rng.normal(size=(B, T, C)) * 2).astype(np.float32)
    tokens, lengths, lp = _run(x, W, tp, threads=2)
    for b in range(B):
        paths, scores = OS.ctc_beam_search(x[b], W, tp)
        for i in range(len(paths)):
            assert tokens[b, i, :lengths[b, i]].tolist() == paths[i]
            assert abs(lp[b, i] - scores[i]) < 2e-3


def test_pruned_beams_follow_tensorflow_order_rules():
    """Many small random problems with narrow beams: pruning, re-entry of evicted prefixes and the deactivation
    corner all show up; the survivors and their order must match the oracle exactly."""
    rng = np.random.default_rng(5)
    for _ in range(600):
        T, C, W = int(rng.integers(2, 9)), int(rng.integers(2, 6)), int(rng.integers(1, 5))
        x = _log_softmax(rng.normal(size=(1, T, C)) * 2).astype(np.float32)
        tokens, lengths, lp = _run(x, W, W)
        paths, scores = OS.ctc_beam_search(x[0], W, W)
        got = [tokens[0, i, :lengths[0, i]].tolist() for i in range(len(paths))]
        assert got == paths
        assert np.abs(lp[0, :len(scores)] - np.asarray(scores)).max() < 1e-3


def test_beam_one_on_peaked_input_equals_best_path_and_seq_len_is_honoured():
    rng = np.random.default_rng(2)
    B, T, C = 2, 25, 9
    x = rng.normal(size=(B, T, C))
    cls = rng.integers(0, C, size=(B, T))
    for b in range(B):
        x[b, np.arange(T), cls[b]] += 12.0
    x = _log_softmax(x).astype(np.float32)
    seq_len = np.array([T, 11], np.int32)
    tokens, lengths, lp = _run(x, 1, 1, seq_len=seq_len)
    for b in range(B):
        row, prev = [], -1
        for t in range(seq_len[b]):
            c = int(cls[b, t])
            if c != C - 1 and c != prev:
                row.append(c)
            prev = c
        assert tokens[b, 0, :lengths[b, 0]].tolist() == row


def test_bad_arguments_are_rejected():
    x = np.zeros((1, 3, 4), np.float32)
    with pytest.raises(ValueError):
        _run(x, 2, 3)                                            # top_paths > beam_width
    with pytest.raises(ValueError):
        _run(x, 2, 1, seq_len=np.array([5], np.int32))           # longer than T
