"""CPU: pin the oracle.  (1) against every number the reference's own tests hold for this path
(the silence log-mel fixture, frame-count formula, SpecAugment bounds, mask-padding invariance,
LR schedule end point); (2) against an independent second route (torch.nn.LSTM/GRU/RNN,
F.conv2d, torch.stft, F.ctc_loss) because no reference test pins model/loss numerics."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as Fn

from oracle import deepspeech2 as ODS
from oracle import features as F
from oracle import las as OLAS
from oracle import layers as L
from oracle import measure as M
from oracle import rng
from oracle.tfrecord import read_tfrecord

LIBRI = dict(sample_rate=16000, frame_length=320, frame_step=160, fft_length=320, num_mel_bins=80,
             lower_edge_hertz=80.0, upper_edge_hertz=7600.0)


# ---------------------------------------------------------------- reference-held golden facts
def test_silence_logmel_equals_reference_tfrecord_fixture(golden_dir):
    """reference tests/test_data.py:53-57: fresh log-mel of the (silent) wav == TFRecord fixture."""
    recs = list(read_tfrecord(os.path.join(golden_dir, "reference_fixtures", "wav_dataset.tfrecord")))
    assert len(recs) == 2
    x = F.log_mel_spectrogram(np.zeros(66150), 16000, 320, 160, 320, 80, 80.0, 7600.0)
    for feat, tok in recs:
        assert feat.shape == (412, 80, 1) and feat.dtype == np.float32
        np.testing.assert_array_equal(x.astype(np.float32), feat)
    assert recs[0][1].tolist() == [ord(c) for c in "Hello World Good night"]   # PseudoTokenizer ids


@pytest.mark.parametrize("n,fl,fs", [(66150, 320, 160), (1000, 400, 160), (16000, 1024, 256), (319, 320, 160)])
def test_frame_count_formula(n, fl, fs):
    """reference tests/test_data.py:60-145: (N - frame_length + frame_step) // frame_step."""
    assert F.num_frames(n, fl, fs) == max(0, (n - fl + fs) // fs)


@pytest.mark.parametrize("Fm,m_F,T,p,m_T", [(27, 1, 100, 1.0, 1), (15, 2, 70, 0.2, 2)])
def test_spec_augment_bounds(Fm, m_F, T, p, m_T):
    """reference tests/test_data.py:148-163 (time warp excluded)."""
    num_time, v = 234, 80
    g = np.random.default_rng(0)
    data = g.uniform(0.1, 1.0, (num_time, v, 1))
    changed = False
    for seed in range(20):
        fr, tm = F.spec_augment_params(seed, 0, num_time, v, Fm, m_F, T, p, m_T)
        aug = F.spec_augment(data, fr, tm)
        zero = (aug == 0).all(axis=2)
        assert zero.all(axis=0).sum() <= Fm * m_F
        assert zero.all(axis=1).sum() <= T * m_T
        assert zero.all(axis=1).sum() <= int(np.float32(num_time) * np.float32(p)) + 0
        changed |= bool((aug != data).any())
        for f0, f in fr:
            assert 0 <= f < Fm and 0 <= f0 and f0 + f <= v
        for t0, t in tm:
            assert 0 <= t < T and 0 <= t0 and t0 + t <= num_time
    assert changed


@pytest.mark.parametrize("num_epoch,lr,min_lr,warm", [(10, 1.1, 0.0, 0.3), (33, 1e-5, 1e-7, 0.1), (100, 100, 0, 0.5)])
def test_lr_scheduler_reaches_min(num_epoch, lr, min_lr, warm):
    """reference tests/test_utils.py:7-16 (strengthened: the reference forgot the assert)."""
    fn = M.LRScheduler(num_epoch, lr, min_lr, warm)
    vals = [fn(i) for i in range(num_epoch + 1)]
    assert math.isclose(vals[-1], min_lr, rel_tol=1e-9, abs_tol=1e-12)
    assert max(vals) <= lr * (1 + 1e-12)
    assert all(v >= min_lr for v in vals)


@pytest.mark.parametrize("rt,units,B,T,D,pad", [("rnn", 13, 23, 11, 8, 3), ("lstm", 33, 34, 41, 2, 4), ("gru", 111, 55, 3, 99, 5)])
def test_birnn_mask_padding_invariance(rt, units, B, T, D, pad):
    """reference tests/models/test_las.py:21-44."""
    g = torch.Generator().manual_seed(0)
    ng = {"lstm": 4, "gru": 3, "rnn": 1}[rt]
    mk = lambda: (torch.randn(D, ng * units, generator=g, dtype=torch.float64) * .3,
                  torch.randn(units, ng * units, generator=g, dtype=torch.float64) * .3,
                  torch.randn((2, ng * units) if rt == "gru" else (ng * units,), generator=g, dtype=torch.float64) * .3)
    fwd, bwd = mk(), mk()
    x = torch.randn(B, T, D, generator=g, dtype=torch.float64)
    mask = torch.randn(B, T, generator=g) > 0.1
    out, *st = L.birnn(rt, x, mask, fwd, bwd)
    assert out.shape == (B, T, 2 * units) and st[0].shape == (B, units)
    xp = torch.cat([x, torch.randn(B, pad, D, generator=g, dtype=torch.float64)], 1)
    mp = torch.cat([mask, torch.zeros(B, pad, dtype=torch.bool)], 1)
    outp, *stp = L.birnn(rt, xp, mp, fwd, bwd)
    assert torch.equal(out, outp[:, :-pad])


def test_checkpoint_shapes_match_oracle_param_tables(golden_dir):
    """Keras weight layouts: every variable name/shape of the reference's test checkpoints
    (tests/data/model-checkpoints/*.ckpt.index) appears in the oracle's parameter tables."""
    import yaml
    fx = os.path.join(golden_dir, "reference_fixtures")
    for index, cfgf, shapes_fn in (("las.ckpt.index", "las_mini_for_test.yml", OLAS.param_shapes),
                                   ("ds.ckpt.index", "deepspeech_mini_for_test.yml", ODS.param_shapes)):
        raw = open(os.path.join(fx, index), "rb").read()
        cfg = yaml.safe_load(open(os.path.join(fx, cfgf)))
        names = shapes_fn(cfg)
        # the index is prefix-compressed; look for the distinctive path fragments instead
        frags = [n.split("/")[-2] + "/" + n.split("/")[-1] for n in names]
        hits = sum(1 for f in set(frags) if f.replace("moving_variance", "variance").split("/")[-1].encode() in raw)
        assert hits == len(set(frags))


# ---------------------------------------------------------------- second-route cross checks
def test_stft_power_against_torch_stft():
    g = np.random.default_rng(1)
    x = g.standard_normal(4000)
    ref = torch.stft(torch.from_numpy(x), 320, 160, 320, window=torch.hann_window(320, periodic=True, dtype=torch.float64),
                     center=False, return_complex=True).abs().pow(2).T.numpy()
    T = F.num_frames(4000, 320, 160)
    mel = F.mel_weight_matrix(80, 161, 16000, 80., 7600.).astype(np.float32).astype(np.float64)
    want = np.log(ref @ mel + 1e-12)
    got = F.log_mel_spectrogram(x, 16000, 320, 160, 320)[:, :, 0]
    assert got.shape == (T, 80)
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9)


def test_mel_matrix_properties():
    m = F.mel_weight_matrix(80, 161, 16000, 80.0, 7600.0)
    assert m.shape == (161, 80) and (m[0] == 0).all() and (m >= 0).all() and m.max() <= 1.0
    assert ((m > 0).sum(axis=0) >= 1).all()          # every filter has support
    centers = m.argmax(axis=0)
    assert (np.diff(centers) >= 0).all()


def test_delta_matches_direct_formula():
    g = np.random.default_rng(2)
    x = g.standard_normal((9, 5, 1))
    y = F.delta_accelerate(x)
    assert y.shape == (9, 5, 3)
    np.testing.assert_allclose(y[0, :, 1], x[0, :, 0]); np.testing.assert_allclose(y[0, :, 2], x[0, :, 0])
    np.testing.assert_allclose(y[1, :, 2], x[1, :, 0] - 2 * x[0, :, 0])
    np.testing.assert_allclose(y[5, :, 2], x[5, :, 0] - 2 * x[4, :, 0] + x[3, :, 0])


def test_conv2d_against_torch():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 50, 20, 3, generator=g, dtype=torch.float64)
    for (kh, kw), (sh, sw), O in (((3, 3), (2, 2), 8), ((11, 5), (2, 1), 4)):
        k = torch.randn(kh, kw, 3, O, generator=g, dtype=torch.float64)
        b = torch.randn(O, generator=g, dtype=torch.float64)
        got = L.conv2d_nhwc(x, k, b, (sh, sw))
        ref = Fn.conv2d(x.permute(0, 3, 1, 2), k.permute(3, 2, 0, 1), b, stride=(sh, sw)).permute(0, 2, 3, 1)
        torch.testing.assert_close(got, ref, rtol=1e-10, atol=1e-10)


def test_conv2d_hand_written_gradients_against_autograd_of_the_tap_by_tap_definition():
    """conv2d_nhwc gathers each clip into one matrix and carries hand-written gradients (oracle/layers.py _Conv2dIm2col): forward
    and both gradients against autograd on the tap-by-tap definition and against F.conv2d, strides (2,2) / (2,1), ragged extents."""
    g = torch.Generator().manual_seed(31)
    for (B, H, W, Cc), (kh, kw), (sh, sw), O in (((2, 50, 20, 3), (3, 3), (2, 2), 8), ((3, 47, 23, 5), (11, 5), (2, 1), 4),
                                                  ((1, 64, 30, 2), (21, 11), (2, 1), 6)):
        x = torch.randn(B, H, W, Cc, generator=g, dtype=torch.float64)
        k = torch.randn(kh, kw, Cc, O, generator=g, dtype=torch.float64)
        b = torch.randn(O, generator=g, dtype=torch.float64)
        res = []
        for fn in (L.conv2d_nhwc, L.conv2d_nhwc_taps,
                   lambda x_, k_, b_, st: Fn.conv2d(x_.permute(0, 3, 1, 2), k_.permute(3, 2, 0, 1), b_, stride=st).permute(0, 2, 3, 1)):
            xl, kl, bl = (t.clone().requires_grad_(True) for t in (x, k, b))
            y = fn(xl, kl, bl, (sh, sw))
            dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
            (y * dy).sum().backward()
            res.append((y.detach(), xl.grad, kl.grad, bl.grad))
        for other in res[1:]:
            for a, r in zip(res[0], other):
                torch.testing.assert_close(a, r, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("rt", ["lstm", "gru", "rnn"])
def test_rnn_layer_against_torch_nn(rt):
    g = torch.Generator().manual_seed(4)
    B, T, D, H = 3, 6, 5, 7
    ng = {"lstm": 4, "gru": 3, "rnn": 1}[rt]
    W = torch.randn(D, ng * H, generator=g, dtype=torch.float64) * .4
    U = torch.randn(H, ng * H, generator=g, dtype=torch.float64) * .4
    b = torch.randn((2, ng * H) if rt == "gru" else (ng * H,), generator=g, dtype=torch.float64) * .4
    x = torch.randn(B, T, D, generator=g, dtype=torch.float64)
    mask = torch.ones(B, T, dtype=torch.bool)
    out, st = L.rnn_layer(rt, x, mask, W, U, b)
    mod = {"lstm": torch.nn.LSTM, "gru": torch.nn.GRU, "rnn": torch.nn.RNN}[rt](D, H, batch_first=True).double()
    with torch.no_grad():
        if rt == "gru":   # torch gate order r,z,n ; Keras z,r,h
            perm = torch.cat([torch.arange(H, 2 * H), torch.arange(0, H), torch.arange(2 * H, 3 * H)])
            mod.weight_ih_l0.copy_(W.T[perm]); mod.weight_hh_l0.copy_(U.T[perm])
            mod.bias_ih_l0.copy_(b[0][perm]); mod.bias_hh_l0.copy_(b[1][perm])
        else:
            mod.weight_ih_l0.copy_(W.T); mod.weight_hh_l0.copy_(U.T)
            mod.bias_ih_l0.copy_(b); mod.bias_hh_l0.zero_()
        ref, _ = mod(x)
    torch.testing.assert_close(out, ref, rtol=1e-10, atol=1e-10)
    # go_backwards == run on the time-reversed input
    outb, _ = L.rnn_layer(rt, x, mask, W, U, b, go_backwards=True)
    torch.testing.assert_close(outb, mod(torch.flip(x, [1]))[0], rtol=1e-10, atol=1e-10)


def test_ctc_against_torch_ctc_loss():
    g = torch.Generator().manual_seed(5)
    B, T, V, Lmax, blank = 3, 12, 9, 5, 4
    logits = torch.randn(B, T, V, generator=g, dtype=torch.float64, requires_grad=True)
    labels = torch.tensor([[1, 2, 2, 3, 0], [5, 6, 0, 0, 0], [7, 7, 7, 0, 0]], dtype=torch.int64)
    loss, per = M.ctc_loss(labels, logits, blank_index=blank)
    lens = (labels != 0).sum(1)
    ref = Fn.ctc_loss(torch.log_softmax(logits, -1).transpose(0, 1), labels, torch.full((B,), T), lens, blank=blank,
                      reduction="none", zero_infinity=False)
    torch.testing.assert_close(per, ref / lens, rtol=1e-10, atol=1e-10)
    g1, = torch.autograd.grad(loss, logits, retain_graph=True)
    g2, = torch.autograd.grad((ref / lens).mean(), logits)
    torch.testing.assert_close(g1, g2, rtol=1e-8, atol=1e-10)


def test_cross_entropy_and_accuracy_ignore_pad():
    g = torch.Generator().manual_seed(6)
    logits = torch.randn(2, 5, 11, generator=g, dtype=torch.float64)
    y = torch.tensor([[3, 4, 0, 0, 0], [1, 0, 2, 5, 0]])
    got = M.sparse_categorical_crossentropy(y, logits, 0)
    ref = Fn.cross_entropy(logits.reshape(-1, 11), y.reshape(-1), ignore_index=0, reduction="mean")
    torch.testing.assert_close(got, ref)
    s, n = M.sparse_categorical_accuracy(y, logits, 0)
    assert n == 5 and 0 <= s <= 5


def test_adam_first_step_closed_form_with_tf_epsilon_placement():
    g = torch.Generator().manual_seed(7)
    p0 = torch.randn(10, generator=g, dtype=torch.float64)
    params, m, v = {"w": p0.clone()}, {"w": torch.zeros(10, dtype=torch.float64)}, {"w": torch.zeros(10, dtype=torch.float64)}
    for it in range(3):
        grad = torch.randn(10, generator=g, dtype=torch.float64)
        M.adam_step(params, {"w": grad}, m, v, it, 1e-2)
    # closed form check of the first-step magnitude: |delta| ~= lr for eps << |g|
    params2, m2, v2 = {"w": p0.clone()}, {"w": torch.zeros(10, dtype=torch.float64)}, {"w": torch.zeros(10, dtype=torch.float64)}
    grad = torch.ones(10, dtype=torch.float64) * 0.5
    M.adam_step(params2, {"w": grad}, m2, v2, 0, 1e-2)
    torch.testing.assert_close(p0 - params2["w"], torch.full((10,), 1e-2 * 0.5 / (0.5 + 1e-7 / math.sqrt(1 - 0.999)), dtype=torch.float64), rtol=1e-9, atol=0)


# ---------------------------------------------------------------- RNG spec + end-to-end smoke of the restatement
def test_rng_spec_known_answers():
    """Known answers of the stateless RNG: the HIP kernels implement the same spec (checked on GPU
    through dropout / SpecAugment parity)."""
    assert int(rng._fmix32(np.uint64(1))) == 0x514E28B7
    r = rng.rand_u32(123, 7, np.arange(4))
    assert [int(v) for v in r] == [int(v) for v in rng.rand_u32(123, 7, np.arange(4))]
    assert len({int(v) for v in rng.rand_u32(123, 7, np.arange(1000))}) == 1000
    m = rng.dropout_mask(5, 1, (1000,), 0.15)
    keep = (m != 0).mean()
    assert 0.80 < keep < 0.90 and np.isclose(m.max(), 1 / (1 - np.float32(0.15)))
    assert 0 <= rng.uniform_int(1, 2, 3, 27) < 27 and rng.uniform_int(1, 2, 3, 0) == 0


def _rand_params(shapes, g, dtype=torch.float64):
    p = {}
    for k, s in shapes.items():
        if k.endswith("gamma") or k.endswith("moving_variance"):
            p[k] = torch.rand(s, generator=g, dtype=dtype) + 0.5
        else:
            p[k] = torch.randn(s, generator=g, dtype=dtype) * 0.2
    return p


@pytest.mark.parametrize("rt", ["lstm", "gru", "rnn"])
def test_las_forward_shapes_and_padding_rows(rt):
    """reference tests/models/test_las.py:47-82 (shape contract) on the restatement."""
    cfg = dict(rnn_type=rt, vocab_size=37, encoder_hidden_dim=6, decoder_hidden_dim=5, num_encoder_layers=2,
               num_decoder_layers=2, dropout=0.1, teacher_forcing_rate=0.9, pad_id=0)
    g = torch.Generator().manual_seed(8)
    p = _rand_params(OLAS.param_shapes(cfg, freq_dim=12, feat_dim=3), g)
    audio = torch.randn(3, 30, 12, 3, generator=g, dtype=torch.float64)
    audio[1, 20:] = 0.0
    tokens = torch.randint(1, 37, (3, 4), generator=g)
    tokens[2, 2:] = 0
    for tf_ in (True, False):
        for training in (False, True):
            out = OLAS.las_forward(p, cfg, audio, tokens, training=training, seed=3, use_teacher_forcing=tf_)
            assert out.shape == (3, 4, 37) and torch.isfinite(out).all()


def test_ds2_forward_shapes_and_mask_modes():
    cfg = dict(num_conv_layers=2, channels=[4, 6], kernel_sizes=[[11, 5], [5, 3]], strides=[[2, 2], [2, 1]], rnn_type="gru",
               num_reccurent_layers=2, hidden_dim=5, dropout=0.1, recurrent_dropout=0.0, vocab_size=17, blank_index=3, pad_index=0)
    g = torch.Generator().manual_seed(9)
    p = _rand_params(ODS.param_shapes(cfg, freq_dim=20, feat_dim=3), g)
    audio = torch.randn(2, 61, 20, 3, generator=g, dtype=torch.float64)
    T2, _ = ODS.conv_out_dims(61, 20, cfg["kernel_sizes"], cfg["strides"])
    out = ODS.ds2_forward(p, cfg, audio, mask_mode="intended")
    assert out.shape == (2, T2, 17)
    compat = ODS.ds2_forward(p, cfg, audio, mask_mode="reference_compat")
    # deepspeech2.py:74 as written masks every frame: logits collapse to the FC bias
    torch.testing.assert_close(compat, p["fully_connected/bias"].expand(2, T2, 17))


# ------------------------------------------------------------------------------------------ the bf16-operand mode of the restatement
def _tiny_las(rt="lstm", He=8, Hd=8, V=13):
    from oracle import las as OLAS
    cfg = dict(rnn_type=rt, vocab_size=V, encoder_hidden_dim=He, decoder_hidden_dim=Hd, num_encoder_layers=2, num_decoder_layers=2,
               dropout=0.1, teacher_forcing_rate=0.99, pad_id=0)
    g = torch.Generator().manual_seed(12)
    p = {}
    for k, s in OLAS.param_shapes(cfg, 12, 3).items():
        p[k] = (torch.rand(s, generator=g, dtype=torch.float64) + 0.5) if k.endswith(("gamma", "moving_variance")) else \
            torch.randn(s, generator=g, dtype=torch.float64) * 0.3
    leaves = {k: v.clone().requires_grad_(not k.endswith(("moving_mean", "moving_variance"))) for k, v in p.items()}
    audio = torch.randn(19, 30, 12, 3, generator=g, dtype=torch.float64)
    audio[2, 20:] = 0.0
    toks = torch.randint(1, V, (19, 5), generator=g)
    toks[4, 3:] = 0
    return OLAS, cfg, leaves, audio, toks


@pytest.mark.parametrize("rt", ["lstm", "gru"])
def test_bf16_operand_mode_without_rounding_is_the_reference_form(rt):
    """L.bf16_operands() re-associates the attention the way the build does (key projection hoisted out of the decoder loop,
    scores = h Kq^T + s0) and routes every contraction through hand-written gradients: with the rounding itself switched off the
    logits and EVERY gradient must equal the reference-form restatement to float64 rounding."""
    OLAS, cfg, leaves, audio, toks = _tiny_las(rt)
    res = []
    for mode in (None, L.bf16_operands(rounding=False, wide_h=4, attn_hd=4), L.bf16_operands(rounding=False)):
        lv = {k: v.detach().clone().requires_grad_(v.requires_grad) for k, v in leaves.items()}
        if mode is None:
            out = OLAS.las_forward(lv, cfg, audio, toks[:, :-1], training=True, seed=3)
        else:
            with mode:
                out = OLAS.las_forward(lv, cfg, audio, toks[:, :-1], training=True, seed=3)
        loss = M.sparse_categorical_crossentropy(toks[:, 1:], out, 0)
        loss.backward()
        res.append((out.detach(), {k: v.grad for k, v in lv.items() if v.requires_grad}))
    for out, grads in res[1:]:
        torch.testing.assert_close(out, res[0][0], rtol=1e-10, atol=1e-11)
        for k, gr in grads.items():
            torch.testing.assert_close(gr, res[0][1][k], rtol=1e-9, atol=1e-11, msg=k)


def test_bf16_operand_mode_rounds_where_it_says():
    """mm_dense rounds both operands in all three products; mm_cell is exact below the wide threshold except for its weight
    gradient; bf16 rounding is to nearest even on 8 significant bits."""
    g = torch.Generator().manual_seed(2)
    a = torch.randn(20, 6, generator=g, dtype=torch.float64)
    b = torch.randn(6, 5, generator=g, dtype=torch.float64)
    dc = torch.randn(20, 5, generator=g, dtype=torch.float64)
    bf = lambda t: t.float().to(torch.bfloat16).double()
    assert float(bf(torch.tensor([1.0 + 2 ** -8])).item()) == 1.0 and float(bf(torch.tensor([1.0 + 3 * 2 ** -8])).item()) == 1.0 + 2 ** -6
    with L.bf16_operands(wide_h=8):
        for fn, rf, rda in ((L.mm_dense, True, True), (lambda x, y: L.mm_cell(x, y, 4), False, False), (lambda x, y: L.mm_cell(x, y, 8), True, True)):
            al, bl = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
            c = fn(al, bl)
            (c * dc).sum().backward()
            torch.testing.assert_close(c.detach(), bf(a) @ bf(b) if rf else a @ b, rtol=0, atol=1e-14)
            torch.testing.assert_close(al.grad, bf(dc) @ bf(b).t() if rda else dc @ b.t(), rtol=0, atol=1e-14)
            torch.testing.assert_close(bl.grad, bf(a).t() @ bf(dc), rtol=0, atol=1e-14)
    c = L.mm_dense(a, b)
    torch.testing.assert_close(c, a @ b, rtol=0, atol=0)              # outside the context: the plain product
