"""GPU parity of greedy decoding (reference search.py:23-81, 223-252) against the oracle restatement,
plus the reference's own shape-level checks (tests/test_search.py)."""
import numpy as np
import pytest
import torch

from oracle import search as OS
from tests import test_ds2_gpu as TD
from tests import test_las_gpu as TL
from tests.util import assert_close

pytestmark = pytest.mark.gpu


def _p(t):
    import ctypes as C
    return C.c_void_p(t.data_ptr())


# ---------------------------------------------------------------------------------------------- kernels
@pytest.mark.parametrize("B,V", [(5, 53), (3, 16000), (2, 257)])
def test_greedy_update_kernel_matches_reference_update_rules(B, V):
    from speech_recognition_amd import ops
    from speech_recognition_amd._lib import check, load
    g = torch.Generator().manual_seed(B + V)
    eos, pad = 3, 0
    ended = torch.zeros(B, dtype=torch.bool)
    log_ppl = torch.zeros(B, dtype=torch.float64)
    seq_len = torch.full((B,), 99, dtype=torch.int64)
    d_ended, d_ppl = torch.zeros(B, dtype=torch.uint8).cuda(), torch.zeros(B).cuda()
    d_len, d_tok = torch.full((B,), 99, dtype=torch.int32).cuda(), torch.zeros(B, dtype=torch.int32).cuda()
    for step in range(6):
        logits = torch.randn(B, V, generator=g) * 3
        if step == 2:
            logits[0, eos] = 50.0                                # row 0 ends here
        if step == 4:
            logits[1, eos] = 50.0
            logits[B - 1, 7] = logits[B - 1, 11] = 60.0          # exact tie -> lowest index
        lsm = torch.log_softmax(logits.double(), dim=1)
        lp, tok = lsm.max(dim=1)
        tok = lsm.argmax(dim=1)
        log_ppl = torch.where(ended, log_ppl, log_ppl + lp)
        tok = torch.where(ended, torch.full_like(tok, pad), tok)
        ended = ended | (tok == eos)
        seq_len = torch.where(tok == eos, torch.full_like(seq_len, step + 2), seq_len)
        dl = logits.cuda()
        check(load().asr_greedy_update(_p(dl), V, B, V, step + 1, eos, pad, _p(d_tok), _p(d_ended), _p(d_ppl), _p(d_len), ops._stream()))
        assert d_tok.cpu().tolist() == tok.tolist()
        assert d_ended.cpu().bool().tolist() == ended.tolist()
        assert d_len.cpu().tolist() == seq_len.tolist()
        assert_close(d_ppl, log_ppl, 1e-5, "log perplexity")
    assert ended[0] and d_tok[0].item() == pad


@pytest.mark.parametrize("B,T,V,blank", [(3, 50, 17, 3), (2, 333, 16000, 14), (4, 64, 30, 0), (1, 1, 5, 4)])
def test_ctc_greedy_kernel_matches_tf_semantics(B, T, V, blank):
    from speech_recognition_amd import ops
    from speech_recognition_amd._lib import check, load
    g = torch.Generator().manual_seed(T + V)
    logits = torch.randn(B, T, V, generator=g) * 2
    # long runs of one class, blanks between repeats, and exact ties between blank and a token
    cls = torch.randint(0, V, (B, (T + 2) // 3), generator=g).repeat_interleave(3, dim=1)[:, :T].contiguous()   # runs of 3
    cls[:, 4::7] = blank
    logits.scatter_(2, cls[..., None], 9.0)                      # a clear winner whose log-probability is not ~0
    if T > 10:
        logits[0, 5, blank] = 9.0
        logits[0, 5, cls[0, 5]] = 9.0                            # tie: the non-blank class wins (blank is last)
    out = torch.cat([logits.double(), logits.double()[:, :, blank:blank + 1]], dim=2)
    m = torch.zeros(V + 1, dtype=torch.float64)
    m[blank] = -1e9
    lsm = torch.log_softmax(out + m, dim=2)
    best_ref = lsm.argmax(dim=2)
    neg_ref = -lsm.max(dim=2).values.sum(dim=1)
    dl = logits.cuda().reshape(B * T, V)
    best, best_lp = torch.empty(B * T, dtype=torch.int32).cuda(), torch.empty(B * T).cuda()
    tokens, lengths, neg = torch.empty(B, T, dtype=torch.int32).cuda(), torch.empty(B, dtype=torch.int32).cuda(), torch.empty(B).cuda()
    check(load().asr_ctc_greedy(_p(dl), V, B, T, V, blank, _p(best), _p(best_lp), _p(tokens), _p(lengths), _p(neg), ops._stream()))
    assert torch.equal(best.view(B, T).cpu().long(), best_ref)
    assert_close(neg, neg_ref, 1e-3 if V < 100 else 1e-5, "neg_sum_logits")   # small V: log-probs near 0, f32 absolute error
    for b in range(B):
        prev, row = -1, []
        for t in range(T):
            c = int(best_ref[b, t])
            if c != V and c != prev:
                row.append(c)
            prev = c
        assert lengths[b].item() == len(row)
        assert tokens[b, :len(row)].cpu().tolist() == row
        assert (tokens[b, len(row):] == 0).all()


# ---------------------------------------------------------------------------------------------- searchers
@pytest.mark.parametrize("rt", ["lstm", "gru"])
def test_las_greedy_search_matches_oracle(rt):
    from speech_recognition_amd.search import LAS_Searcher
    cfg = TL.mk_cfg(rt)
    m, vals = TL.build(cfg)
    audio, _, _ = TL.inputs(B=4, T=38)
    bos, eos, max_len = 2, 3, 9
    ref_tok, ref_ppl = OS.greedy_las(vals, cfg, audio.double(), max_len, bos, eos, 0)
    tok, ppl = LAS_Searcher(m, max_len, bos, eos, check_every=2).greedy_search(audio.cuda())
    assert tuple(tok.shape) == tuple(ref_tok.shape) and tok.dtype == torch.int32
    assert torch.equal(tok.cpu().long(), ref_tok)
    assert_close(ppl, ref_ppl, 2e-4, "perplexity")
    assert (tok[:, 0] == bos).all()


def test_las_greedy_search_stops_at_eos_and_pads():
    """Bias the vocabulary layer so that EOS wins at once: every row ends at step 1, the result is [BOS, EOS]
    (search.py:37 leaves the loop when all rows ended) and the perplexity uses sequence length 2."""
    from speech_recognition_amd.search import LAS_Searcher
    cfg = TL.mk_cfg("lstm")
    m, vals = TL.build(cfg)
    vals = dict(vals)
    bias = vals["attend_and_speller/feedforward/bias"].clone()
    bias[3] = 80.0
    vals["attend_and_speller/feedforward/bias"] = bias
    m.load_state_dict({k: v.float() for k, v in vals.items()})
    audio, _, _ = TL.inputs(B=3, T=38)
    ref_tok, ref_ppl = OS.greedy_las(vals, cfg, audio.double(), 12, 2, 3, 0)
    tok, ppl = LAS_Searcher(m, 12, 2, 3).greedy_search(audio.cuda())
    assert ref_tok.tolist() == [[2, 3]] * 3 and tok.cpu().tolist() == [[2, 3]] * 3
    assert_close(ppl, ref_ppl, 1e-4, "perplexity")


@pytest.mark.parametrize("mask_mode", ["intended", "reference_compat"])
def test_ds2_greedy_search_matches_oracle(mask_mode):
    from speech_recognition_amd.search import DeepSpeechSearcher
    cfg = TD.mk_cfg("gru")
    m, vals = TD.build(cfg, mask_mode)
    audio, _ = TD.inputs(B=3, T=61)
    ref_tok, ref_prob, ref_best = OS.greedy_ds2(vals, cfg, audio.double(), cfg["blank_index"], mask_mode)
    tok, prob, best = DeepSpeechSearcher(m, cfg["blank_index"]).greedy_search(audio.cuda(), return_alignment=True)
    assert torch.equal(best.cpu().long(), ref_best)              # the frame-level CTC alignment
    assert tok.cpu().numpy().tolist() == ref_tok.tolist()
    assert_close(prob, ref_prob, 5e-4, "sequence probability")


# ---------------------------------------------------------------------------------------------- beam search
@pytest.mark.parametrize("R,V,k", [(6, 53, 4), (3, 16000, 8), (2, 40, 32)])
def test_beam_topk_kernel_is_log_softmax_top_k_with_tf_tie_order(R, V, k):
    from speech_recognition_amd import ops
    from speech_recognition_amd._lib import check, load
    g = torch.Generator().manual_seed(R * V + k)
    logits = torch.randn(R, V, generator=g) * 3
    logits[0, 5] = logits[0, 9] = logits[0, 2] = 20.0            # three-way tie at the top: index order 2, 5, 9
    logits[1, 7] = logits[1, 3]                                   # a tie somewhere below
    d = logits.cuda()
    lp, tok = torch.empty(R, k).cuda(), torch.empty(R, k, dtype=torch.int32).cuda()
    check(load().asr_beam_topk(_p(d), V, R, V, k, _p(lp), _p(tok), ops._stream()))
    lsm = torch.log_softmax(logits.double(), dim=1)
    for r in range(R):
        order = sorted(range(V), key=lambda i: (-float(logits[r, i]), i))[:k]
        assert tok[r].cpu().tolist() == order
        assert_close(lp[r], lsm[r, order], 1e-5, "top-k log-probabilities")
    assert tok[0, :3].cpu().tolist() == [2, 5, 9]


@pytest.mark.parametrize("rt,beam,max_len", [("lstm", 3, 9), ("gru", 2, 7), ("lstm", 1, 6)])
def test_las_beam_search_matches_oracle(rt, beam, max_len):
    from speech_recognition_amd.search import LAS_Searcher
    cfg = TL.mk_cfg(rt)
    m, vals = TL.build(cfg)
    audio, _, _ = TL.inputs(B=3, T=38)
    bos, eos = 2, 3
    ref_tok, ref_ppl = OS.beam_las(vals, cfg, audio.double(), max_len, bos, eos, 0, beam)
    tok, ppl = LAS_Searcher(m, max_len, bos, eos, check_every=2).beam_search(audio.cuda(), beam)
    assert tuple(tok.shape) == tuple(ref_tok.shape) and tok.dtype == torch.int32
    assert torch.equal(tok.cpu().long(), ref_tok)
    assert_close(ppl, ref_ppl, 5e-4, "perplexity")


def test_las_beam_search_with_early_eos_matches_oracle():
    """EOS made likely: hypotheses end at different steps, ended rows spawn equal-scored children (stable top_k
    ties), the length penalty separates short from long rows and the loop stops once every row holds an EOS."""
    from speech_recognition_amd.search import LAS_Searcher
    cfg = TL.mk_cfg("lstm")
    m, vals = TL.build(cfg)
    vals = dict(vals)
    bias = vals["attend_and_speller/feedforward/bias"].clone()
    bias[3] = 1.5
    vals["attend_and_speller/feedforward/bias"] = bias
    m.load_state_dict({k: v.float() for k, v in vals.items()})
    audio, _, _ = TL.inputs(B=4, T=38)
    for beam, alpha, beta in ((3, 1, 32), (4, 0.7, 2)):
        ref_tok, ref_ppl = OS.beam_las(vals, cfg, audio.double(), 14, 2, 3, 0, beam, alpha, beta)
        tok, ppl = LAS_Searcher(m, 14, 2, 3, check_every=3).beam_search(audio.cuda(), beam, alpha, beta)
        assert (ref_tok == 3).any()                               # the case is not vacuous
        assert tuple(tok.shape) == tuple(ref_tok.shape)
        assert torch.equal(tok.cpu().long(), ref_tok)
        assert_close(ppl, ref_ppl, 5e-4, "perplexity")


def test_beam_one_equals_greedy():
    """tests/test_search.py:19-25 and 58-64: beam_search(…, 1)[:, 0] equals greedy_search for both models."""
    from speech_recognition_amd.search import DeepSpeechSearcher, LAS_Searcher
    cfg = TL.mk_cfg("lstm")
    m, _ = TL.build(cfg)
    audio, _, _ = TL.inputs(B=4, T=38)
    s = LAS_Searcher(m, 10, 2, 3)
    g_tok, g_ppl = s.greedy_search(audio.cuda())
    b_tok, b_ppl = s.beam_search(audio.cuda(), 1)
    assert torch.equal(b_tok[:, 0, :], g_tok)
    assert_close(b_ppl[:, 0], g_ppl, 1e-4, "perplexity")
    cfg = TD.mk_cfg("gru")
    m, _ = TD.build(cfg, "intended")
    audio, _ = TD.inputs(B=3, T=61)
    s = DeepSpeechSearcher(m, cfg["blank_index"])
    g_tok, g_prob = s.greedy_search(audio.cuda())
    b_tok, b_prob = s.beam_search(audio.cuda(), 1)
    assert b_tok.shape[1] == 1 and b_prob.shape == (3, 1)


@pytest.mark.parametrize("beam", [1, 4, 16])
def test_ds2_beam_search_matches_oracle(beam):
    from speech_recognition_amd.search import DeepSpeechSearcher
    cfg = TD.mk_cfg("gru")
    m, vals = TD.build(cfg, "intended")
    audio, _ = TD.inputs(B=3, T=61)
    ref_tok, ref_prob = OS.beam_ds2(vals, cfg, audio.double(), cfg["blank_index"], beam)
    tok, prob = DeepSpeechSearcher(m, cfg["blank_index"]).beam_search(audio.cuda(), beam)
    assert tok.dtype == torch.int32 and tuple(tok.shape) == tuple(ref_tok.shape)
    assert tok.cpu().numpy().tolist() == ref_tok.tolist()
    assert_close(prob, torch.from_numpy(ref_prob), 2e-3, "sequence probability")


def test_ctc_log_softmax_kernel():
    from speech_recognition_amd import ops
    from speech_recognition_amd._lib import check, load
    g = torch.Generator().manual_seed(3)
    R, V, blank = 37, 301, 14
    logits = torch.randn(R, V, generator=g) * 2
    out = torch.empty(R, V + 1).cuda()
    check(load().asr_ctc_log_softmax(_p(logits.cuda()), V, R, V, blank, _p(out), ops._stream()))
    x = torch.cat([logits.double(), logits.double()[:, blank:blank + 1]], dim=1)
    x[:, blank] += -1e9
    ref = torch.log_softmax(x, dim=1)
    keep = [c for c in range(V + 1) if c != blank]
    assert_close(out[:, keep], ref[:, keep], 1e-5, "log_softmax with the blank last")
    assert (out[:, blank] < -1e8).all()


def test_reference_search_smoke_shapes():
    """tests/test_search.py:7-62 minus the beam half: model sizes and input ranges of the reference tests."""
    from speech_recognition_amd.models import LAS, DeepSpeech2
    from speech_recognition_amd.search import DeepSpeechSearcher, LAS_Searcher
    g = torch.Generator().manual_seed(0)
    ds = DeepSpeech2(1, [32], [[41, 11]], [[2, 2]], "lstm", 1, 240, 0.1, 0.0, 111, 33, 1)
    x = (torch.rand(8, 300, 123, 3, generator=g) * 100).cuda()
    tok, prob = DeepSpeechSearcher(ds, 33).greedy_search(x)
    assert tok.shape[0] == 8 and tok.dtype == torch.int32 and tuple(prob.shape) == (8,)
    assert int(tok.max()) < 111 and (tok != 33).all() and torch.isfinite(prob).all()
    las = LAS(rnn_type="lstm", vocab_size=100, encoder_hidden_dim=32, decoder_hidden_dim=32, num_encoder_layers=1,
              num_decoder_layers=1, dropout=0.1, teacher_forcing_rate=0.99)
    x = (torch.rand(8, 10, 123, 3, generator=g) * 100).cuda()
    tok, ppl = LAS_Searcher(las, 17, 2, 3).greedy_search(x)
    assert tok.shape[0] == 8 and 2 <= tok.shape[1] <= 17 and tuple(ppl.shape) == (8,)
    assert (tok[:, 0] == 2).all() and torch.isfinite(ppl).all() and (ppl >= 1.0).all()
