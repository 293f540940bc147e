"""GPU parity of the data.py feature functions (make_log_mel_spectrogram / spec_augment /
delta_accelerate factories and the stored-feature front end) against the float64 oracle, and the
reference's own data tests (tests/test_data.py:53-163) restated on its fixtures."""
import os

import numpy as np
import pytest
import torch

from oracle import features as F
from tests.util import gpu

pytestmark = pytest.mark.gpu

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_fixtures")


class PseudoTokenizer:
    @staticmethod
    def tokenize(sentence):
        return [ord(c) for c in sentence]


def _wav_dataset():
    from speech_recognition_amd.data import get_dataset
    return get_dataset(os.path.join(FIX, "wav_dataset.tsv"), "wav", 22050, PseudoTokenizer, False)


def test_make_tfrecord_dataset_equals_reference_fixture():
    """tests/test_data.py:53-57: log-mel of the wav dataset == the stored TFRecord features, exactly."""
    from speech_recognition_amd.data import get_tfrecord_dataset, make_log_mel_spectrogram
    ds = _wav_dataset().map(make_log_mel_spectrogram(16000, 320, 160, 320, 80, 80.0, 7600.0))
    stored = get_tfrecord_dataset(os.path.join(FIX, "wav_dataset.tfrecord"))
    n = 0
    for (feat, tok), (ref_feat, ref_tok) in zip(ds, stored):
        np.testing.assert_array_equal(feat.cpu().numpy(), ref_feat)
        np.testing.assert_array_equal(tok, ref_tok)
        n += 1
    assert n == 2


@pytest.mark.parametrize("sample_rate,frame_length,frame_step,fft_length,num_mel_bins,lower_edge_hertz,upper_edge_hertz", [
    (22050, 1024, 1024, 1024, 80, 10, 10000),
    (16000, 128, 64, 256, 123, 12, 88),
    (32000, 128, 80, 128, 321, 32, 16000),
    (44100, 512, 512, 256, 333, 333, 3333),
])
def test_make_log_mel_spectrogram_shapes_and_values(sample_rate, frame_length, frame_step, fft_length, num_mel_bins,
                                                    lower_edge_hertz, upper_edge_hertz):
    """The reference's parameter grid (tests/test_data.py:82-113): shape on the silent fixture, and values against
    the oracle on a noisy clip (long frames take the kernel's fewer-frames-per-workgroup variants, fft_length <
    frame_length crops the frame like tf.signal.stft)."""
    from speech_recognition_amd.data import make_log_mel_spectrogram
    fn = make_log_mel_spectrogram(sample_rate, frame_length, frame_step, fft_length, num_mel_bins, lower_edge_hertz, upper_edge_hertz)
    audio = next(iter(_wav_dataset()))[0]
    n = audio.shape[0]
    out = fn(audio)
    assert tuple(out.shape) == ((n - frame_length + frame_step) // frame_step, num_mel_bins, 1)
    g = np.random.default_rng(0)
    x = np.clip(0.1 * g.standard_normal(20000) + 0.2 * np.sin(np.arange(20000) * 0.05), -1, 1).astype(np.float32)
    got = fn(x).cpu().numpy().astype(np.float64)
    ref = F.log_mel_spectrogram(x.astype(np.float64), sample_rate, frame_length, frame_step, fft_length, num_mel_bins,
                                lower_edge_hertz, upper_edge_hertz)
    assert got.shape == ref.shape
    err = np.abs(got - ref)
    # empty mel filters give log(1e-12) exactly on both sides; elsewhere f32 DFT error on O(1..10) values
    assert err.max() < 5e-3 and np.median(err) < 5e-5, (err.max(), np.median(err))


@pytest.mark.parametrize("W,F_,m_F,T,p,m_T", [(None, 27, 1, 100, 1.0, 1), (None, 15, 2, 70, 0.2, 2)])
def test_spec_augment_reference_properties(W, F_, m_F, T, p, m_T):
    """tests/test_data.py:148-163 (W dropped: time warping is unsupported, see test below)."""
    from speech_recognition_amd.data import spec_augment
    num_time, num_frequency = 234, 80
    fn = spec_augment(num_frequency, W, F_, m_F, T, p, m_T)
    g = np.random.default_rng(1)
    data = g.uniform(0.1, 1.0, (num_time, num_frequency, 1)).astype(np.float32)
    changed = False
    for seed in range(1, 6):
        aug = fn(data, seed=seed).cpu().numpy()
        assert aug.shape == data.shape
        is_zero = (aug == 0.0).all(axis=2)
        assert is_zero.all(axis=0).sum() <= F_ * m_F
        assert is_zero.all(axis=1).sum() <= T * m_T
        assert ((aug == data) | (aug == 0)).all()                   # mask value is 0.0, nothing else changes
        fr, tm = F.spec_augment_params(seed, 0, num_time, num_frequency, F_, m_F, T, p, m_T)
        np.testing.assert_array_equal(aug, F.spec_augment(data, fr, tm).astype(np.float32))
        changed |= bool((aug != data).any())
    assert changed


def test_spec_augment_batch_uses_per_clip_lengths_and_matches_fused_kernel():
    from speech_recognition_amd import ops
    from speech_recognition_amd.data import delta_accelerate, make_log_mel_spectrogram, spec_augment
    sa = dict(F=27, m_F=2, T=100, p=1.0, m_T=2)
    B, n = 3, 16000 * 3
    g = np.random.default_rng(2)
    audio = np.clip(0.1 * g.standard_normal((B, n)), -1, 1).astype(np.float32)
    ns = np.array([n, n - 7000, 25000], np.int32)
    cfg = dict(sample_rate=16000, frame_length=320, frame_step=160, fft_length=320, num_mel_bins=80, lower_edge_hertz=80.0,
               upper_edge_hertz=7600.0)
    # route 1: three stand-alone calls (the reference's map chain, run/train.py:88-116)
    mel = make_log_mel_spectrogram(**cfg)(audio, n_samples=ns)
    nf = np.array([F.num_frames(int(v), 320, 160) for v in ns], np.int32)
    aug = spec_augment(80, None, **sa)(mel, n_frames=nf, seed=99)
    feats = delta_accelerate(aug, n_frames=nf)
    # route 2: the fused kernel of the training step
    plan = ops.LogmelPlan(**cfg, spec_augment=dict(enable=True, W=None, **sa))
    fused = plan(gpu(audio), torch.from_numpy(ns).cuda(), plan.num_frames(n), seed=torch.tensor([99], dtype=torch.int32, device="cuda"))
    torch.testing.assert_close(feats, fused, rtol=0, atol=0)         # same arithmetic, same draws: bit-identical
    ref = F.batch_features(audio.astype(np.float64), ns, cfg, seed=99, spec_aug=sa, T_out=plan.num_frames(n))
    assert np.abs(feats.cpu().numpy() - ref).max() < 2e-3
    for b in range(B):
        assert (feats[b, nf[b]:] == 0).all()


def test_delta_accelerate_matches_oracle_bitwise():
    from speech_recognition_amd.data import delta_accelerate
    g = np.random.default_rng(3)
    x = g.standard_normal((57, 80, 1)).astype(np.float32)
    out = delta_accelerate(x)
    assert tuple(out.shape) == (57, 80, 3)
    np.testing.assert_array_equal(out.cpu().numpy(), F.delta_accelerate(x))       # f32 subtractions in the same order
    out2, text = delta_accelerate(x, np.array([1, 2], np.int32))                   # (audio, text) form (data.py:326-328)
    assert text.tolist() == [1, 2] and torch.equal(out2, out)
    xb = np.zeros((2, 60, 80, 1), np.float32)
    xb[0, :57], xb[1, :20] = x, x[:20]
    ob = delta_accelerate(xb, n_frames=np.array([57, 20], np.int32)).cpu().numpy()
    np.testing.assert_array_equal(ob[0, :57], F.delta_accelerate(x))
    np.testing.assert_array_equal(ob[1, :20], F.delta_accelerate(x[:20]))
    assert (ob[0, 57:] == 0).all() and (ob[1, 20:] == 0).all()                     # padding stays padding


def test_stored_feature_plan_equals_stand_alone_calls():
    from speech_recognition_amd import ops
    from speech_recognition_amd.data import delta_accelerate, spec_augment
    sa = dict(enable=True, W=None, F=27, m_F=2, T=100, p=1.0, m_T=2)
    g = np.random.default_rng(4)
    feats = g.standard_normal((2, 300, 80, 1)).astype(np.float32)
    nf = np.array([300, 180], np.int32)
    feats[1, 180:] = 0
    seed = torch.tensor([5], dtype=torch.int32, device="cuda")
    plan = ops.StoredFeaturePlan(80, True, sa)
    out = plan(gpu(feats), torch.from_numpy(nf).cuda(), 300, seed=seed)
    ref = delta_accelerate(spec_augment(80, None, 27, 2, 100, 1.0, 2)(feats, n_frames=nf, seed=5), n_frames=nf)
    assert torch.equal(out, ref)
    plain = ops.StoredFeaturePlan(80, False, None)(gpu(feats), torch.from_numpy(nf).cuda(), 300)
    assert torch.equal(plain, gpu(feats))
