"""Host side of the input pipeline (no GPU): file decoders in the native library, TFRecord framing,
the Dataset verbs run/train.py uses.  Mirrors the reference's tests/test_data.py:31-57 on its own data
fixtures (copied under tests/golden/reference_fixtures as data, not code)."""
import ctypes as C
import gzip
import os

import numpy as np
import pytest

from speech_recognition_amd import _lib, tfrecord
from speech_recognition_amd.data import (Dataset, SentencePieceTokenizer, filter_example, get_dataset, get_tfrecord_dataset,
                                         load_audio_file, slice_example)
from tests import flac_writer as FW

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_fixtures")


class PseudoTokenizer:                       # tests/test_data.py:18-21: unicode code points
    @staticmethod
    def tokenize(sentence):
        return [ord(c) for c in sentence]


def _decode_blob(blob, fmt):
    lib = _lib.load()
    info = _lib.AudioInfo()
    _lib.check(lib.asr_audio_info(blob, len(blob), fmt, C.byref(info)))
    out = np.empty(max(info.frames, 1), np.float32)
    n = C.c_long()
    _lib.check(lib.asr_audio_decode(blob, len(blob), fmt, out.ctypes.data_as(C.c_void_p), out.size, C.byref(n)))
    return info, out[:n.value]


# ---------------------------------------------------------------------------------------------- reference fixtures
def test_get_dataset_wav_and_flac_fixture():
    data = list(get_dataset(os.path.join(FIX, "wav_dataset.tsv"), "wav", 22050, PseudoTokenizer, False))
    assert len(data) == 2 and len(data[0]) == 2
    audio, tokens = data[0]
    assert audio.shape == (66150,) and audio.dtype == np.float32          # tests/test_data.py:39
    assert tokens.shape == (22,) and tokens.dtype == np.int32             # tests/test_data.py:40
    np.testing.assert_array_equal(data[0][0], data[1][0])                 # test.wav == test.flac (tests/test_data.py:41)
    assert data[1][1].tolist() == [ord(c) for c in "gOddy bye"]


def test_get_dataset_pcm_fixture():
    audio, tokens = next(iter(get_dataset(os.path.join(FIX, "pcm_dataset.tsv"), "pcm", 22050, PseudoTokenizer, False)))
    raw = open(os.path.join(FIX, "audio_files", "test.pcm"), "rb").read()
    ref = np.frombuffer(raw + (b"\0" if len(raw) % 2 else b""), "<i2").astype(np.float32) / 32768.0   # data.py:101-105
    np.testing.assert_array_equal(audio, ref)
    assert tokens.shape == (22,)


def test_flac_fixture_header_matches_wav():
    wav = open(os.path.join(FIX, "audio_files", "test.wav"), "rb").read()
    flac = open(os.path.join(FIX, "audio_files", "test.flac"), "rb").read()
    iw, aw = _decode_blob(wav, 0)
    iff, af = _decode_blob(flac, 1)
    assert (iw.sample_rate, iw.channels, iw.bits_per_sample, iw.frames) == (22050, 1, 16, 66150)
    assert (iff.sample_rate, iff.channels, iff.bits_per_sample, iff.frames) == (22050, 1, 16, 66150)
    np.testing.assert_array_equal(aw, af)


def test_invalid_format_and_corrupt_files():
    with pytest.raises(ValueError):
        load_audio_file(16000, "ogg")                                     # data.py:109
    flac = bytearray(open(os.path.join(FIX, "audio_files", "test.flac"), "rb").read())
    flac[-3] ^= 0x55                                                      # breaks the last frame's CRC-16
    info = _lib.AudioInfo()
    lib = _lib.load()
    assert lib.asr_audio_info(bytes(flac), len(flac), 1, C.byref(info)) == 0
    out = np.empty(info.frames, np.float32)
    n = C.c_long()
    assert lib.asr_audio_decode(bytes(flac), len(flac), 1, out.ctypes.data_as(C.c_void_p), out.size, C.byref(n)) != 0
    assert b"CRC" in lib.asr_last_error()
    assert lib.asr_audio_info(b"RIFFxxxxWAVE", 12, 0, C.byref(info)) != 0  # no chunks
    assert lib.asr_audio_info(b"nope", 4, 1, C.byref(info)) != 0


def test_tfrecord_fixture_and_writer_reproduces_it(tmp_path):
    path = os.path.join(FIX, "wav_dataset.tfrecord")
    data = list(get_tfrecord_dataset(path))
    assert len(data) == 2 and len(data[0]) == 2
    assert data[0][0].shape == (412, 80, 1) and data[0][0].dtype == np.float32   # tests/test_data.py:50-51
    assert data[0][1].shape == (22,) and data[0][1].dtype == np.int32
    assert list(tfrecord.read_examples(path, check_crc=True))[1][1].tolist() == [ord(c) for c in "gOddy bye"]
    out = str(tmp_path / "copy.tfrecord")
    with tfrecord.TFRecordWriter(out) as w:
        for a, t in data:
            w.write(a, t)
    assert gzip.open(out).read() == gzip.open(path).read()               # byte-identical records incl. both CRCs


def test_crc32c_known_answers():
    assert tfrecord.crc32c(b"123456789") == 0xE3069283                    # the CRC-32C check value
    assert tfrecord.crc32c(b"") == 0
    blob = bytes(range(256)) * 5 + b"xyz"
    assert tfrecord.crc32c(blob[100:], tfrecord.crc32c(blob[:100])) == tfrecord.crc32c(blob)
    assert tfrecord.crc32c(bytes(32)) == 0x8A9136AA                       # RFC 3720 B.4: 32 zero bytes


def test_sentencepiece_tokenizer_adds_bos_eos():
    tok = SentencePieceTokenizer(os.path.join(FIX, "sp_model_unigram_16K_libri.model"))
    ids = tok.tokenize("HELLO WORLD GOOD NIGHT")
    assert ids.dtype == np.int32 and ids[0] == tok.sp.bos_id() and ids[-1] == tok.sp.eos_id()
    assert tok.detokenize(ids[1:-1]) == "HELLO WORLD GOOD NIGHT"
    assert int(ids.max()) < 16000


# ---------------------------------------------------------------------------------------------- FLAC decoder branches
def _pcm(n, seed, nch=1, amp=9000):
    g = np.random.default_rng(seed)
    t = np.arange(n)
    return [np.clip(amp * np.sin(2 * np.pi * (0.01 + 0.013 * c) * t) + g.normal(0, 300, n), -32768, 32767).astype(np.int64)
            for c in range(nch)]


def _expect(chans):
    x = np.stack(chans).astype(np.float32) / np.float32(32768.0)
    return x[0] if len(chans) == 1 else x.sum(0) / np.float32(len(chans))


@pytest.mark.parametrize("spec", [
    dict(type="verbatim"),
    dict(type="fixed", order=0, porder=0, method=0, params=[12]),
    dict(type="fixed", order=1, porder=1, method=0, params=[9, 10]),
    dict(type="fixed", order=2, porder=2, method=1, params=[8, 9, 17, 8]),
    dict(type="fixed", order=3, porder=0, method=0, params=[9]),
    dict(type="fixed", order=4, porder=3, method=0, params=[10, 9, ("esc", 18), 9, 10, 11, 9, 10]),
    dict(type="lpc", order=2, precision=12, shift=10, coefs=[1900, -950], porder=1, method=0, params=[8, 8]),
    dict(type="lpc", order=8, precision=14, shift=12, coefs=[5000, -2100, 900, -400, 150, -60, 20, -5], porder=2, method=1,
         params=[9, 9, 9, 9]),
])
def test_flac_mono_subframe_types(spec):
    chans = _pcm(256 * 3 + 77, 1)                                          # last block is short
    blob = FW.encode(chans, 16000, 16, 256, "independent", [spec])
    info, audio = _decode_blob(blob, 1)
    assert (info.sample_rate, info.channels, info.frames) == (16000, 1, len(chans[0]))
    np.testing.assert_array_equal(audio, _expect(chans))


@pytest.mark.parametrize("assignment", ["independent", "left_side", "right_side", "mid_side"])
def test_flac_stereo_assignments_average_channels(assignment):
    chans = _pcm(1024, 2, nch=2)
    spec = dict(type="fixed", order=2, porder=1, method=0, params=[11, 11])
    blob = FW.encode(chans, 22050, 16, 512, assignment, [spec, spec])
    info, audio = _decode_blob(blob, 1)
    assert info.channels == 2
    np.testing.assert_array_equal(audio, _expect(chans))                   # data.py:116 reduce_mean over channels


def test_flac_constant_wasted_bits_and_unknown_length():
    x = (_pcm(512, 3)[0] >> 3) << 3                                         # 3 wasted bits
    blob = FW.encode([x], 16000, 16, 256, "independent", [dict(type="fixed", order=1, porder=0, method=0, params=[7], wasted=3)],
                     total_known=False)
    info, audio = _decode_blob(blob, 1)
    assert info.frames == 512                                               # counted by decoding
    np.testing.assert_array_equal(audio, _expect([x]))
    c = np.full(300, -1234, np.int64)
    _, audio = _decode_blob(FW.encode([c], 16000, 16, 128, "independent", [dict(type="constant")]), 1)
    np.testing.assert_array_equal(audio, _expect([c]))


def test_wav_stereo_and_extra_chunks():
    import struct
    g = np.random.default_rng(5)
    pcm = g.integers(-32768, 32767, (1000, 2)).astype("<i2")
    fmt = struct.pack("<HHIIHH", 1, 2, 8000, 8000 * 4, 4, 16)
    body = b"WAVE" + b"LIST" + struct.pack("<I", 3) + b"abc\0" + b"fmt " + struct.pack("<I", 16) + fmt + b"data" + \
        struct.pack("<I", pcm.nbytes) + pcm.tobytes()
    blob = b"RIFF" + struct.pack("<I", len(body)) + body
    info, audio = _decode_blob(blob, 0)
    assert (info.sample_rate, info.channels, info.frames) == (8000, 2, 1000)
    ref = (pcm.astype(np.float32) / np.float32(32768.0)).sum(1) / np.float32(2)
    np.testing.assert_array_equal(audio, ref)


# ---------------------------------------------------------------------------------------------- Dataset verbs
def _toy(n=10):
    return Dataset.from_iterable([(np.arange(i + 1, dtype=np.float32), np.arange(i % 3 + 1, dtype=np.int32)) for i in range(n)])


def test_dataset_map_filter_skip_take_repeat():
    ds = _toy(6)
    assert [len(a) for a, _ in ds.map(lambda a, t: (a * 2, t))] == [1, 2, 3, 4, 5, 6]
    assert [len(a) for a, _ in ds.filter(lambda a, t: len(a) % 2 == 0)] == [2, 4, 6]
    assert [len(a) for a, _ in ds.skip(4)] == [5, 6]
    assert [len(a) for a, _ in ds.repeat().skip(5).take(3)] == [6, 1, 2]
    assert len(list(ds.repeat(2))) == 12
    assert list(Dataset.from_iterable([]).repeat()) == []                  # an empty dataset does not spin


def test_dataset_shuffle_is_a_permutation_and_buffered():
    ds = _toy(50)
    out = [len(a) for a, _ in ds.shuffle(8, seed=3)]
    assert sorted(out) == list(range(1, 51)) and out != list(range(1, 51))
    assert max(i - (v - 1) for i, v in enumerate(out)) <= 50                # sanity
    assert [len(a) for a, _ in ds.shuffle(1, seed=3)] == list(range(1, 51))  # buffer 1 keeps the order (train.py:87)
    first = out[0]
    assert first <= 9                                                      # the first output comes from the first 9 inputs


def test_padded_batch_pads_with_zeros_and_reports_lengths():
    ex = [((np.ones((3, 2), np.float32), np.array([5, 6], np.int32)), np.array([6, 7], np.int32)),
          ((np.ones((5, 2), np.float32), np.array([5], np.int32)), np.array([7], np.int32)),
          ((np.ones((1, 2), np.float32), np.array([1, 2, 3], np.int32)), np.array([2, 3, 4], np.int32))]
    batches = list(Dataset.from_iterable(ex).padded_batch(2, with_lengths=True))
    assert len(batches) == 2
    ((audio, tin), tout), ((la, lti), lto) = batches[0]
    assert audio.shape == (2, 5, 2) and tin.shape == (2, 2) and tout.shape == (2, 2)
    assert (audio[0, 3:] == 0).all() and tin[1].tolist() == [5, 0] and tout[1].tolist() == [7, 0]
    assert la.tolist() == [3, 5] and lti.tolist() == [2, 1] and lto.dtype == np.int32
    fixed = list(Dataset.from_iterable(ex).padded_batch(3, (([8, 2], [4]), [4])))[0]
    assert fixed[0][0].shape == (3, 8, 2) and fixed[0][1].shape == (3, 4) and fixed[1].shape == (3, 4)
    with pytest.raises(ValueError):
        list(Dataset.from_iterable(ex).padded_batch(3, (([4, 2], [4]), [4])))
    assert len(list(Dataset.from_iterable(ex).padded_batch(2, drop_remainder=True))) == 1


def test_parallel_map_is_ordered_bounded_and_propagates_errors():
    import threading
    import time
    from speech_recognition_amd.data import AUTOTUNE
    seen = []
    lock = threading.Lock()

    def slow(a, t):
        time.sleep(0.002 * (len(a) % 3))                                   # finish out of order
        with lock:
            seen.append(len(a))
        return a * 2, t
    out = [len(a) for a, _ in _toy(100).map(slow, num_parallel_calls=4)]
    assert out == list(range(1, 101)) and sorted(seen) == out             # ordered results, every element once
    assert [len(a) for a, _ in _toy(7).map(slow, num_parallel_calls=AUTOTUNE)] == list(range(1, 8))

    def boom(a, t):
        if len(a) == 5:
            raise ValueError("bad clip")
        return a, t
    with pytest.raises(ValueError, match="bad clip"):
        list(_toy(10).map(boom, num_parallel_calls=3))


def test_prefetch_preserves_order_and_propagates_errors():
    assert [len(a) for a, _ in _toy(20).prefetch(2)] == list(range(1, 21))

    def boom():
        yield 1
        raise RuntimeError("decode failed")
    with pytest.raises(RuntimeError, match="decode failed"):
        list(Dataset(boom).prefetch(2))


def test_levenshtein_distance_reference_cases():
    """The reference's known answers, tests/test_utils.py:18-34."""
    from speech_recognition_amd.utils import levenshtein_distance
    cases = [([], [], 0, False), (list("abc"), [], 3, False), ("hello", "hello", 0, False), (list("kitten"), list("sitten"), 1, False),
             ("sunday", "saturday", 3, False), ([1, 2, 3], [4, 4, 4, 5], 4, False), (list("안녕하세요"), list("안녕? 해..?"), 6, False),
             ("hi", "hello", 2.0, True), ("byebye", "yes", 2 / 3, True)]
    for truth, hyp, dist, norm in cases:
        assert levenshtein_distance(truth, hyp, norm) == dist


def test_filter_and_slice_example():
    ds = Dataset.from_iterable([(np.zeros((n, 80, 1), np.float32), np.zeros(u, np.int32)) for n, u in [(10, 3), (30, 3), (10, 9)]])
    assert len(list(ds.apply(filter_example(20, 5)))) == 1                 # data.py:331-341
    sliced = list(ds.apply(slice_example(20, 5)))                          # data.py:344-354
    assert [(a.shape[0], t.shape[0]) for a, t in sliced] == [(10, 3), (20, 3), (10, 5)]
