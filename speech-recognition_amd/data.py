"""Input pipeline of speech_recognition/data.py for the MI355X build.

Two halves:

* **Host side** - `Dataset`, a small lazy pipeline with the tf.data verbs run/train.py uses (map, filter,
  apply, repeat, skip, shuffle, padded_batch, prefetch), the TSV / TFRecord sources (`get_dataset`,
  `get_tfrecord_dataset`, data.py:12-79), audio file decoding in the native library (`load_audio_file`,
  data.py:82-119) and the max-length policies (`filter_example`, `slice_example`, data.py:331-354).
  Examples are numpy arrays; nothing here touches the GPU.
* **Device side** - the feature functions with the reference's factory signatures
  (`make_log_mel_spectrogram`, `spec_augment`, `delta_accelerate`, data.py:145-328).  They accept one
  example ([N] audio / [T, v, C] features, as the reference's per-example map does) or a padded batch
  with lengths, and run the HIP kernels of libasr_mi355x.so.  The training step itself does not call
  them one by one: it uses the fused kernel (`DataConfig.logmel_plan`), which computes the same values
  in one pass.
"""
import csv
import ctypes as C
import glob
import os
import queue
import random
import threading
from typing import Callable, Iterable, Iterator, Optional

import numpy as np
import torch

from . import _lib, tfrecord


# =========================================================================================== Dataset
def _map_structure(fn, *structs):
    """Apply fn leaf-wise over examples that are nested tuples of arrays."""
    s0 = structs[0]
    if isinstance(s0, tuple):
        return tuple(_map_structure(fn, *[s[i] for s in structs]) for i in range(len(s0)))
    return fn(*structs)


def _call(fn, element):
    return fn(*element) if isinstance(element, tuple) else fn(element)


AUTOTUNE = -1     # tf.data.experimental.AUTOTUNE stand-in for map(num_parallel_calls=...)


def _autotune_workers() -> int:
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


class Dataset:
    """Lazy, re-iterable sequence of examples (nested tuples of numpy arrays)."""

    def __init__(self, make_iter: Callable[[], Iterator]):
        self._make_iter = make_iter

    def __iter__(self):
        return iter(self._make_iter())

    @staticmethod
    def from_iterable(items: Iterable) -> "Dataset":
        items = list(items)
        return Dataset(lambda: iter(items))

    def map(self, fn, num_parallel_calls=None) -> "Dataset":
        """Apply fn to every example.  num_parallel_calls > 1 (or AUTOTUNE) runs fn on a thread pool with a
        bounded window of examples in flight, results in input order (tf.data's deterministic parallel map).
        Worth it for functions that release the GIL - the native audio decoders do."""
        workers = _autotune_workers() if num_parallel_calls == AUTOTUNE else int(num_parallel_calls or 0)
        if workers <= 1:
            return Dataset(lambda: (_call(fn, e) for e in self))

        def gen():
            from collections import deque
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=workers) as pool:
                window = deque()
                for e in self:
                    window.append(pool.submit(_call, fn, e))
                    if len(window) >= 4 * workers:
                        yield window.popleft().result()
                while window:
                    yield window.popleft().result()
        return Dataset(gen)

    def filter(self, fn) -> "Dataset":
        return Dataset(lambda: (e for e in self if _call(fn, e)))

    def apply(self, transformation) -> "Dataset":
        return transformation(self)

    def repeat(self, count: Optional[int] = None) -> "Dataset":
        def gen():
            n = 0
            while count is None or n < count:
                empty = True
                for e in self:
                    empty = False
                    yield e
                if empty:
                    return
                n += 1
        return Dataset(gen)

    def skip(self, count: int) -> "Dataset":
        def gen():
            it = iter(self)
            for _ in range(count):
                if next(it, _END) is _END:
                    return
            yield from it
        return Dataset(gen)

    def take(self, count: int) -> "Dataset":
        def gen():
            for i, e in enumerate(self):
                if i >= count:
                    return
                yield e
        return Dataset(gen)

    def shuffle(self, buffer_size: int, seed: Optional[int] = None) -> "Dataset":
        """tf.data semantics: a buffer of `buffer_size` examples, a uniformly drawn one leaves as each new
        one enters; reshuffled on every iteration."""
        epoch = [0]

        def gen():
            rng = random.Random(None if seed is None else seed + epoch[0])
            epoch[0] += 1
            buf = []
            for e in self:
                if len(buf) < max(1, buffer_size):
                    buf.append(e)
                    continue
                i = rng.randrange(len(buf))
                out, buf[i] = buf[i], e
                yield out
            rng.shuffle(buf)
            yield from buf
        return Dataset(gen)

    def padded_batch(self, batch_size: int, padded_shapes=None, drop_remainder: bool = False, with_lengths: bool = False) -> "Dataset":
        """Stack `batch_size` consecutive examples, zero-padding every axis to the batch maximum (or to the
        fixed size given in padded_shapes: a nested structure of shape lists, None = dynamic).  With
        with_lengths the dataset yields (batch, lengths): lengths has the same structure, each leaf the
        int32 [B] sizes of axis 0 before padding - what the device kernels need to tell padding from data
        (a silent clip is all zeros too)."""
        def pad_leaf(shape, *arrays):
            arrays = [np.asarray(a) for a in arrays]
            nd = arrays[0].ndim
            dims = [max(a.shape[k] for a in arrays) for k in range(nd)]
            if shape is not None:
                for k, want in enumerate(shape):
                    if want is not None:
                        if dims[k] > want:
                            raise ValueError(f"padded_batch: dimension {k} is {dims[k]}, larger than the padded shape {want}")
                        dims[k] = want
            out = np.zeros([len(arrays)] + dims, arrays[0].dtype)
            for i, a in enumerate(arrays):
                out[(i,) + tuple(slice(0, n) for n in a.shape)] = a
            return out

        def len_leaf(*arrays):
            return np.asarray([np.asarray(a).shape[0] if np.asarray(a).ndim else 1 for a in arrays], np.int32)

        def emit(chunk):
            if padded_shapes is None:
                batch = _map_structure(lambda *xs: pad_leaf(None, *xs), *chunk)
            else:
                batch = _pad_with_shapes(pad_leaf, padded_shapes, chunk)
            return (batch, _map_structure(len_leaf, *chunk)) if with_lengths else batch

        def gen():
            chunk = []
            for e in self:
                chunk.append(e)
                if len(chunk) == batch_size:
                    yield emit(chunk)
                    chunk = []
            if chunk and not drop_remainder:
                yield emit(chunk)
        return Dataset(gen)

    def prefetch(self, buffer_size: Optional[int] = None) -> "Dataset":
        """Produce the next elements on a background thread (decode + batching overlap the GPU step)."""
        depth = buffer_size if buffer_size and buffer_size > 0 else 4

        def gen():
            q: "queue.Queue" = queue.Queue(maxsize=depth)
            stop = threading.Event()

            def work():
                try:
                    for e in self:
                        while not stop.is_set():
                            try:
                                q.put((0, e), timeout=0.1)
                                break
                            except queue.Full:
                                continue
                        if stop.is_set():
                            return
                    q.put((1, None))
                except BaseException as exc:   # surfaced on the consumer side
                    q.put((2, exc))

            th = threading.Thread(target=work, daemon=True)
            th.start()
            try:
                while True:
                    kind, val = q.get()
                    if kind == 0:
                        yield val
                    elif kind == 1:
                        return
                    else:
                        raise val
            finally:
                stop.set()
        return Dataset(gen)


_END = object()


def _is_shape(x):
    return isinstance(x, (list, tuple)) and all(isinstance(v, (int, type(None))) for v in x)


def _pad_with_shapes(pad_leaf, shapes, chunk):
    if _is_shape(shapes):
        return pad_leaf(list(shapes), *chunk)
    return tuple(_pad_with_shapes(pad_leaf, shapes[i], [c[i] for c in chunk]) for i in range(len(shapes)))


# =========================================================================================== sources
def _decode(path: str, file_format: str) -> np.ndarray:
    if file_format not in _lib.AUDIO_FORMATS:
        if file_format == "mp3":
            raise NotImplementedError("mp3 decoding is not part of this build (no shipped data config uses it)")
        raise ValueError(f"File Format: {file_format} is not valid!")      # data.py:109
    with open(path, "rb") as f:
        blob = f.read()
    if file_format in ("wav", "flac"):                     # tfio.audio.AudioIOTensor sniffs the container (data.py:97-98)
        file_format = "flac" if blob[:4] == b"fLaC" else ("wav" if blob[:4] == b"RIFF" else file_format)
    lib, fmt = _lib.load(), _lib.AUDIO_FORMATS[file_format]
    info = _lib.AudioInfo()
    _lib.check(lib.asr_audio_info(blob, len(blob), fmt, C.byref(info)))
    out = np.empty(max(int(info.frames), 1), np.float32)
    n = C.c_long()
    _lib.check(lib.asr_audio_decode(blob, len(blob), fmt, out.ctypes.data_as(C.c_void_p), out.size, C.byref(n)))
    return out[:n.value]


def resample_linear(audio: np.ndarray, rate_in: int, rate_out: int) -> np.ndarray:
    """Stand-in for tfio.audio.resample (data.py:113): linear interpolation on the host.  The reference's
    resampler is a windowed-sinc filter inside tensorflow-io; `resample` is None in run/train.py, so this
    is off the training path and not claimed to match it."""
    if rate_in == rate_out:
        return audio
    n_out = int(round(len(audio) * rate_out / rate_in))
    pos = np.arange(n_out, dtype=np.float64) * (rate_in / rate_out)
    return np.interp(pos, np.arange(len(audio)), audio).astype(np.float32)


def load_audio_file(sample_rate: int, file_format: str, resample: Optional[float] = None) -> Callable[[str], np.ndarray]:
    """data.py:82-119: path -> float32 [TimeStep] in [-1, 1), channels averaged."""
    if file_format not in ("flac", "wav", "pcm", "mp3"):
        raise ValueError(f"File Format: {file_format} is not valid!")

    def _wrapper(audio_file_path) -> np.ndarray:
        audio = _decode(os.fspath(audio_file_path), file_format)
        if resample is not None:
            audio = resample_linear(audio, sample_rate, int(resample))
        return audio

    return _wrapper


def get_dataset(dataset_paths: str, file_format: str, sample_rate: int, tokenizer, shuffle: bool = False,
                resample: Optional[int] = None) -> Dataset:
    """data.py:12-61: TSV files (header line; column 0 = audio path relative to the TSV, column 1 = text)
    -> (audio f32 [N], tokens i32 [U]).  tokenizer: object with ``tokenize(str) -> sequence of int``."""
    dataset_list = sorted(glob.glob(dataset_paths))
    if shuffle:
        random.shuffle(dataset_list)
    load = load_audio_file(sample_rate, file_format, resample)

    def rows():
        for tsv in dataset_list:
            base = os.path.dirname(os.path.abspath(tsv))
            with open(tsv, newline="", encoding="utf-8") as f:
                reader = csv.reader(f, delimiter="\t", quoting=csv.QUOTE_NONE)
                next(reader, None)                                 # header
                for row in reader:
                    if len(row) >= 2:
                        yield os.path.join(base, row[0]), row[1]

    # file read + decode (native code, GIL released) + tokenisation on a thread pool, order preserved
    # (the reference maps load_example with AUTOTUNE parallelism inside each interleaved file, data.py:52-61)
    return Dataset(rows).map(lambda path, text: (load(path), np.asarray(tokenizer.tokenize(text), np.int32)), num_parallel_calls=AUTOTUNE)


def get_tfrecord_dataset(dataset_paths: str) -> Dataset:
    """data.py:64-79: GZIP TFRecord files of (feature tensor f32 [T, F, 1], tokens i32 [U])."""
    dataset_list = sorted(glob.glob(dataset_paths))

    def gen():
        for path in dataset_list:
            yield from tfrecord.read_examples(path)
    return Dataset(gen)


class SentencePieceTokenizer:
    """text.SentencepieceTokenizer(model, add_bos=True, add_eos=True) of run/train.py:78-79."""

    def __init__(self, model_path: str, add_bos: bool = True, add_eos: bool = True):
        import sentencepiece as spm
        self.sp = spm.SentencePieceProcessor()
        self.sp.Load(model_path)
        self.add_bos, self.add_eos = add_bos, add_eos

    def tokenize(self, sentence: str):
        ids = list(self.sp.EncodeAsIds(sentence))
        if self.add_bos:
            ids = [self.sp.bos_id()] + ids
        if self.add_eos:
            ids = ids + [self.sp.eos_id()]
        return np.asarray(ids, np.int32)

    def detokenize(self, ids):
        return self.sp.DecodeIds([int(i) for i in ids])


def filter_example(max_audio_length, max_token_length):
    """data.py:331-341: drop examples whose audio (axis 0) or token count exceeds the maximum."""
    def _wrapper(dataset: Dataset) -> Dataset:
        return dataset.filter(lambda audio, text: np.shape(audio)[0] <= max_audio_length and np.size(text) <= max_token_length)
    return _wrapper


def slice_example(max_audio_length, max_token_length):
    """data.py:344-354: cut audio (axis 0) and tokens to the maximum."""
    def _wrapper(dataset: Dataset) -> Dataset:
        return dataset.map(lambda audio, text: (audio[:max_audio_length], text[:max_token_length]))
    return _wrapper


# =========================================================================================== device features
def _to_device(x, dtype=torch.float32):
    if isinstance(x, torch.Tensor):
        return x.to(device="cuda", dtype=dtype).contiguous()
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).cuda()


def _lengths(n, B, full, device):
    if n is None:
        return torch.full((B,), full, dtype=torch.int32, device=device)
    return _to_device(n, torch.int32)


def _feature_fn(plan_kwargs):
    """Shared body of the three make_* wrappers: build the fused front-end plan on first use, run it."""
    from . import ops
    plan = [None]

    def _wrapper(audio, text=None, n_samples=None):
        if plan[0] is None:
            plan[0] = ops.LogmelPlan(use_delta=False, **plan_kwargs)
        x = _to_device(audio)
        single = x.dim() == 1
        if single:
            x = x[None]
        B, N = x.shape
        ns = _lengths(n_samples, B, N, x.device)
        out = plan[0](x, ns, plan[0].num_frames(N))
        out = out[0] if single else out
        return out if text is None else (out, text)

    return _wrapper


def make_spectrogram(frame_length: int, frame_step: int, fft_length: Optional[int] = None):
    """data.py:122-142: |STFT|.  audio [N] -> [NumFrame, fft_length // 2 + 1, 1] (a padded batch [B, N] with n_samples
    likewise).  fft_length None = the smallest power of two enclosing frame_length ([TF-sem] tf.signal.stft)."""
    if fft_length is None:
        fft_length = 1 << max(int(frame_length) - 1, 0).bit_length()
    return _feature_fn(dict(sample_rate=0, frame_length=frame_length, frame_step=frame_step, fft_length=fft_length, num_mel_bins=0,
                            lower_edge_hertz=0.0, upper_edge_hertz=0.0, feature_type="spectrogram"))


def make_log_mel_spectrogram(sample_rate: int, frame_length: int, frame_step: int, fft_length: int, num_mel_bins: int = 80,
                             lower_edge_hertz: float = 80.0, upper_edge_hertz: float = 7600.0, epsilon: float = 1e-12):
    """data.py:145-189.  Returned callable: (audio[, text][, n_samples]) -> log-mel (, text).
    audio [N] -> [NumFrame, num_mel_bins, 1]; a padded batch [B, N] with n_samples [B] -> [B, NumFrame, mel, 1]
    (frames past each clip's end are exact zeros, i.e. already padded_batch'ed)."""
    return _feature_fn(dict(sample_rate=sample_rate, frame_length=frame_length, frame_step=frame_step, fft_length=fft_length,
                            num_mel_bins=num_mel_bins, lower_edge_hertz=lower_edge_hertz, upper_edge_hertz=upper_edge_hertz,
                            epsilon=epsilon))


def make_mfcc(sample_rate: int, frame_length: int, frame_step: int, fft_length: int, num_mel_bins: int = 80, num_mfcc: int = 40,
              lower_edge_hertz: float = 80.0, upper_edge_hertz: float = 7600.0, epsilon: float = 1e-12):
    """data.py:192-241: DCT-II of the log-mel spectrogram ([TF-sem] tf.signal.mfccs_from_log_mel_spectrograms), first
    num_mfcc coefficients.  audio [N] -> [NumFrame, num_mfcc, 1]."""
    return _feature_fn(dict(sample_rate=sample_rate, frame_length=frame_length, frame_step=frame_step, fft_length=fft_length,
                            num_mel_bins=num_mel_bins, lower_edge_hertz=lower_edge_hertz, upper_edge_hertz=upper_edge_hertz,
                            epsilon=epsilon, feature_type="mfcc", num_mfcc=num_mfcc))


_sa_calls = [0]


def spec_augment(v: int, W: Optional[int] = None, F: Optional[int] = None, m_F: Optional[int] = None, T: Optional[int] = None,
                 p: Optional[float] = None, m_T: Optional[int] = None):
    """data.py:244-307: time warping (W), frequency masking (F, m_F) and time masking (T, p, m_T), in that order.
    Returned callable: (features[, text][, n_frames][, seed]) -> masked features (a new tensor).
    features [T, v, C] or a padded batch [B, T, v, C] with n_frames [B].  seed: int; by default a new one
    per call drawn from Python's `random` (so utils.set_random_seed makes runs repeatable)."""
    from . import ops
    cfg = ops.spec_augment_cfg(v, F, m_F, T, p, m_T)

    def _wrapper(audio, text=None, n_frames=None, seed=None):
        x = _to_device(audio).clone()
        single = x.dim() == 3
        if single:
            x = x[None]
        B, Tn = x.shape[:2]
        if seed is None:
            seed = random.getrandbits(31)
        _sa_calls[0] += 1
        seed_dev = torch.tensor([int(seed) & 0x7FFFFFFF], dtype=torch.int32, device=x.device)
        nf = _lengths(n_frames, B, Tn, x.device)
        if W:                                                    # data.py:269, 275-280
            x = ops.time_warp(x, nf, W, seed_dev)
        ops.spec_augment_(cfg, x, nf, seed_dev)
        out = x[0] if single else x
        return out if text is None else (out, text)

    return _wrapper


def delta_accelerate(audio, text=None, n_frames=None):
    """data.py:310-328: [T, v, 1] -> [T, v, 3] (x, delta, delta-delta); batch form [B, T, v, 1] + n_frames."""
    from . import ops
    x = _to_device(audio)
    single = x.dim() == 3
    if single:
        x = x[None]
    nf = None if n_frames is None else _to_device(n_frames, torch.int32)
    out = ops.delta_accelerate(x, nf)
    out = out[0] if single else out
    return out if text is None else (out, text)
