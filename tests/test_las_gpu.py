"""GPU parity of the whole LAS model (reference models/las.py) against the float64 oracle:
logits in inference and training mode (same stateless-RNG dropout masks), loss / accuracy, every
parameter gradient (torch.autograd on the oracle), BatchNorm moving statistics, with and without
teacher forcing, for all three rnn types, padded clips and padded token rows included."""
import numpy as np
import pytest
import torch

from oracle import las as OLAS
from oracle import measure as OM
from tests.util import assert_close, gpu

pytestmark = pytest.mark.gpu


def mk_cfg(rt, V=53, He=12, Hd=16, Le=2, Ld=2, dropout=0.15):
    return dict(rnn_type=rt, vocab_size=V, encoder_hidden_dim=He, decoder_hidden_dim=Hd, num_encoder_layers=Le,
                num_decoder_layers=Ld, dropout=dropout, teacher_forcing_rate=0.99, pad_id=0)


def build(cfg, F_=20, C_=3, seed=5):
    from speech_recognition_amd.models import LAS
    m = LAS(cfg["rnn_type"], cfg["vocab_size"], cfg["encoder_hidden_dim"], cfg["decoder_hidden_dim"], cfg["num_encoder_layers"],
            cfg["num_decoder_layers"], cfg["dropout"], cfg["teacher_forcing_rate"], cfg["pad_id"], seed=seed)
    m.build(F_, C_)
    g = torch.Generator().manual_seed(seed)
    vals = {}
    for n, s in m.store.shapes.items():
        if n.endswith("gamma"):
            vals[n] = torch.rand(s, generator=g) + 0.5
        else:
            vals[n] = torch.randn(s, generator=g) * (0.5 if "embedding" in n else 0.25)
    for n, v in m.buffers.items():
        vals[n] = torch.rand(v.shape, generator=g) + 0.5 if n.endswith("variance") else torch.randn(v.shape, generator=g) * 0.1
    m.load_state_dict(vals)
    return m, {k: v.double() for k, v in vals.items()}


def inputs(B=3, T=38, F_=20, U=5, V=53, seed=1):
    g = torch.Generator().manual_seed(seed)
    audio = torch.randn(B, T, F_, 3, generator=g)
    audio[1, 25:] = 0.0                # right-padded clip (padded_batch zeros)
    audio[2, 8:12] = 0.0               # interior all-zero frames (SpecAugment-like) -> non-contiguous mask
    tokens = torch.randint(1, V, (B, U), generator=g, dtype=torch.int32)
    tokens[1, 3:] = 0                  # padded token row
    labels = torch.randint(1, V, (B, U), generator=g, dtype=torch.int32)
    labels[1, 3:] = 0
    return audio, tokens, labels


@pytest.mark.parametrize("rt", ["lstm", "gru", "rnn"])
@pytest.mark.parametrize("teacher", [True, False])
def test_inference_logits(rt, teacher):
    cfg = mk_cfg(rt)
    m, vals = build(cfg)
    audio, tokens, _ = inputs()
    ref = OLAS.las_forward(vals, cfg, audio.double(), tokens, training=False, use_teacher_forcing=teacher)
    out = m.forward(audio.cuda(), tokens.cuda(), training=False, use_teacher_forcing=teacher)
    assert tuple(out.shape) == tuple(ref.shape)
    assert_close(out, ref, 2e-4, f"logits {rt} teacher={teacher}")


def test_listener_and_step_api_match_oracle():
    """search.py's entry points: listener(audio) and attend_and_speller(enc, tok, mask, states)."""
    cfg = mk_cfg("lstm")
    m, vals = build(cfg)
    audio, tokens, _ = inputs()
    enc_r, mask_r, st_r, _ = OLAS.listener(vals, cfg, audio.double(), False)
    enc, mask, *st = m.listener(audio.cuda(), training=False)
    assert torch.equal(mask.cpu(), mask_r)
    assert_close(enc, enc_r, 2e-4, "enc")
    for a, b in zip(st, st_r):
        assert_close(a, b, 2e-4, "state")
    lo_r, st2_r, _ = OLAS.attend_and_speller(vals, cfg, enc_r, tokens[:, 0], mask_r, st_r, False)
    lo, *st2 = m.attend_and_speller(enc, tokens[:, 0].cuda(), mask, st, training=False)
    assert_close(lo, lo_r, 2e-4, "step logits")
    for a, b in zip(st2, st2_r):
        assert_close(a, b, 2e-4, "step state")


@pytest.mark.parametrize("rt,teacher,dropout", [("lstm", True, 0.15), ("lstm", False, 0.15), ("gru", True, 0.15), ("rnn", True, 0.0),
                                                ("lstm", True, 0.0)])
def test_training_step_loss_and_every_gradient(rt, teacher, dropout):
    from speech_recognition_amd import ops
    cfg = mk_cfg(rt, dropout=dropout)
    m, vals = build(cfg)
    audio, tokens, labels = inputs()
    seedv = 4242
    m.state[1] = seedv
    leaves = {k: v.clone().requires_grad_(not k.startswith("listener/batch_norm") or k.endswith(("gamma", "beta"))) for k, v in vals.items()}
    logits_r, aux = OLAS.las_forward(leaves, cfg, audio.double(), tokens, training=True, seed=seedv, use_teacher_forcing=teacher,
                                     return_aux=True)
    loss_r = OM.sparse_categorical_crossentropy(labels, logits_r, 0)
    loss_r.backward()
    correct_r, count_r = OM.sparse_categorical_accuracy(labels, logits_r.detach(), 0)

    B, U = tokens.shape
    ws = m._workspace(B, audio.shape[1], U)
    ws.toks_T[:U].copy_(tokens.t().cuda())
    ag = audio.cuda()
    m.forward_ws(ws, ag, True, teacher)
    out = ws.logits.view(U, B, -1).permute(1, 0, 2).clone()
    assert_close(out, logits_r, 3e-4, "training logits")
    ops.fill(m.store.grad, 0.0)
    m.loss_and_grad(ws, labels.t().contiguous().cuda())
    st = ws.stats.cpu().numpy()
    assert abs(st[0] - float(loss_r.detach())) < 1e-3, (st[0], float(loss_r))     # north-star tolerance on the loss
    assert st[1] == correct_r and st[2] == count_r
    m.backward_ws(ws, ag)
    grads = m.store.grads()
    worst = 0.0
    for n, gten in grads.items():
        ref = leaves[n].grad
        assert ref is not None, n
        scale = max(float(ref.abs().max()), 1e-4)   # d/d(key bias) is analytically 0 (softmax shift invariance)
        err = float((gten.double() - ref).abs().max()) / scale
        worst = max(worst, err)
        assert err < 2e-3, f"gradient {n}: normalised error {err:.2e}"
    for n, v in m.buffers.items():
        assert_close(v, aux["bn_updates"][n], 1e-4, n)


def test_padded_rows_do_not_change_valid_rows():
    """Extra zero-padded frames and pad tokens leave the logits of the valid positions unchanged up to
    BatchNorm statistics - so compare in inference mode (moving statistics)."""
    cfg = mk_cfg("lstm")
    m, vals = build(cfg)
    audio, tokens, _ = inputs()
    audio[:, 27:] = 0.0     # the clips end early enough that no new conv window sees real frames
    out = m.forward(audio.cuda(), tokens.cuda(), training=False, use_teacher_forcing=True).clone()
    audio2 = torch.cat([audio, torch.zeros(3, 8, 20, 3)], dim=1)
    tokens2 = torch.cat([tokens, torch.zeros(3, 2, dtype=torch.int32)], dim=1)
    out2 = m.forward(audio2.cuda(), tokens2.cuda(), training=False, use_teacher_forcing=True)
    assert_close(out2[:, :5], out, 2e-5, "prefix logits")


def test_reference_api_surface():
    from speech_recognition_amd.models import LAS
    from speech_recognition_amd.models.las import get_rnn_cls
    with pytest.raises(ValueError, match="rnn_type: foo is invalid!"):
        get_rnn_cls("foo")
    assert LAS.get_batching_shape(None, None, 80, 3) == (([None, 80, 3], [None]), [None])
    assert LAS.get_batching_shape(2048, 128, 80, 3) == (([2048, 80, 3], [127]), [127])
    tok = torch.arange(6)
    (a, t_in), t_out = LAS.make_example("audio", tok)
    assert a == "audio" and t_in.tolist() == [0, 1, 2, 3, 4] and t_out.tolist() == [1, 2, 3, 4, 5]
    assert "{val_accuracy" in LAS.model_checkpoint_path
