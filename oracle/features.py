"""Oracle (TEST INFRASTRUCTURE): audio front end of the training path, numpy float64.

Follows /root/reference/speech_recognition/data.py:
  make_spectrogram          data.py:122-142   (tf.signal.stft, tf.abs)
  make_log_mel_spectrogram  data.py:145-189   (tf.signal.stft / linear_to_mel_weight_matrix [TF-sem])
  make_mfcc                 data.py:192-241   (+ tf.signal.mfccs_from_log_mel_spectrograms [TF-sem])
  spec_augment              data.py:244-307   (time warp data.py:275-280 not restated: W is null in
                                               every shipped data config)
  delta_accelerate          data.py:310-328
and the zero padding of `padded_batch` in run/train.py:189-197.
"""
import numpy as np

from . import rng

STREAM_SPECAUG = 3  # RNG stream id of the SpecAugment draws (see oracle/rng.py)


def num_frames(n_samples: int, frame_length: int, frame_step: int) -> int:
    """[TF-sem] tf.signal.frame(pad_end=False): 1 + (N - frame_length) // frame_step.
    Same number as the reference test's (N - frame_length + frame_step) // frame_step
    (tests/test_data.py:60-145)."""
    if n_samples < frame_length:
        return 0
    return 1 + (n_samples - frame_length) // frame_step


def hann_periodic(n: int) -> np.ndarray:
    """[TF-sem] tf.signal.hann_window(periodic=True): 0.5 - 0.5 cos(2 pi k / n)."""
    k = np.arange(n, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)


def mel_weight_matrix(num_mel_bins, num_spectrogram_bins, sample_rate, lower_edge_hertz, upper_edge_hertz):
    """[TF-sem] tf.signal.linear_to_mel_weight_matrix (data.py:177-179): HTK mel scale
    1127 ln(1 + f/700), triangles on the mel axis, no area normalisation, DC row zero.
    Returned in float64; the TF op hands back float32 (callers round)."""
    def hz2mel(f):
        return 1127.0 * np.log1p(np.asarray(f, dtype=np.float64) / 700.0)

    nyquist = sample_rate / 2.0
    lin = np.linspace(0.0, nyquist, num_spectrogram_bins)[1:]
    bins_mel = hz2mel(lin)[:, None]
    edges = np.linspace(hz2mel(lower_edge_hertz), hz2mel(upper_edge_hertz), num_mel_bins + 2)
    lower, center, upper = edges[:-2][None, :], edges[1:-1][None, :], edges[2:][None, :]
    lower_slopes = (bins_mel - lower) / (center - lower)
    upper_slopes = (upper - bins_mel) / (upper - center)
    w = np.maximum(0.0, np.minimum(lower_slopes, upper_slopes))
    return np.concatenate([np.zeros((1, num_mel_bins)), w], axis=0)


def log_mel_spectrogram(audio, sample_rate, frame_length, frame_step, fft_length,
                        num_mel_bins=80, lower_edge_hertz=80.0, upper_edge_hertz=7600.0, epsilon=1e-12):
    """data.py:169-187: stft -> abs -> square -> matmul(mel) -> log(x + eps) -> [T, mel, 1]."""
    audio = np.asarray(audio, dtype=np.float64)
    T = num_frames(audio.shape[0], frame_length, frame_step)
    idx = np.arange(T)[:, None] * frame_step + np.arange(frame_length)[None, :]
    frames = audio[idx] * hann_periodic(frame_length)[None, :]
    spec = np.fft.rfft(frames, n=fft_length, axis=1)           # [T, fft/2+1]
    power = np.abs(spec) ** 2
    mel = mel_weight_matrix(num_mel_bins, fft_length // 2 + 1, sample_rate, lower_edge_hertz, upper_edge_hertz)
    mel = mel.astype(np.float32).astype(np.float64)            # TF returns a float32 matrix
    return np.log(power @ mel + epsilon)[:, :, None]


def stft_magnitude(audio, frame_length, frame_step, fft_length=None):
    """[TF-sem] tf.abs(tf.signal.stft(audio, frame_length, frame_step, fft_length)): periodic Hann window, frames
    without end padding, rfft of length fft_length (default: the smallest power of two >= frame_length; a frame
    longer than fft_length is cropped by the rfft)."""
    if fft_length is None:
        fft_length = 1 << max(int(frame_length) - 1, 0).bit_length()
    audio = np.asarray(audio, dtype=np.float64)
    T = num_frames(audio.shape[0], frame_length, frame_step)
    idx = np.arange(T)[:, None] * frame_step + np.arange(frame_length)[None, :]
    frames = audio[idx] * hann_periodic(frame_length)[None, :]
    return np.abs(np.fft.rfft(frames, n=fft_length, axis=1))


def spectrogram(audio, frame_length, frame_step, fft_length=None):
    """data.py:134-137: [T, fft_length // 2 + 1, 1]."""
    return stft_magnitude(audio, frame_length, frame_step, fft_length)[:, :, None]


def mfcc(audio, sample_rate, frame_length, frame_step, fft_length, num_mel_bins=80, num_mfcc=40, lower_edge_hertz=80.0,
         upper_edge_hertz=7600.0, epsilon=1e-12):
    """data.py:218-234: log-mel, then [TF-sem] tf.signal.mfccs_from_log_mel_spectrograms =
    dct(type 2, norm None: X_k = 2 sum_n x_n cos(pi k (2n + 1) / (2N))) * rsqrt(2N), first num_mfcc coefficients."""
    lm = log_mel_spectrogram(audio, sample_rate, frame_length, frame_step, fft_length, num_mel_bins, lower_edge_hertz,
                             upper_edge_hertz, epsilon)[:, :, 0]
    N = num_mel_bins
    n, k = np.arange(N)[:, None], np.arange(N)[None, :]
    dct2 = lm @ (2.0 * np.cos(np.pi * k * (2 * n + 1) / (2.0 * N)))
    return (dct2 / np.sqrt(2.0 * N))[:, :num_mfcc, None]


def spec_augment_params(seed, clip, num_time, v, F, m_F, T, p, m_T):
    """Draws of data.py:282-301 with the build's stateless RNG.  Returns
    ([(f0, f), ...], [(t0, t), ...]); zeroed ranges are [f0, f0+f) and [t0, t0+t)."""
    freq, time = [], []
    if F and m_F:
        for i in range(m_F):
            f = rng.uniform_int(seed, STREAM_SPECAUG, clip * 64 + 2 * i, F)
            f0 = rng.uniform_int(seed, STREAM_SPECAUG, clip * 64 + 2 * i + 1, v - f)
            freq.append((f0, f))
    if T and p and m_T:
        applied = 0
        max_maskable = int(np.float32(num_time) * np.float32(p))
        for j in range(m_T):
            t = rng.uniform_int(seed, STREAM_SPECAUG, clip * 64 + 32 + 2 * j, T)
            t = min(t, max_maskable - applied)
            t = max(t, 0)  # reference would raise for negative spans; never happens for p<=1
            applied += t
            t0 = rng.uniform_int(seed, STREAM_SPECAUG, clip * 64 + 32 + 2 * j + 1, num_time - t)
            time.append((t0, t))
    return freq, time


def spec_augment(x, freq, time):
    """data.py:282-301: multiply by 0/1 masks (mask value is 0.0, not the mean). x: [T, v, 1]."""
    x = np.array(x, dtype=np.float64, copy=True)
    for f0, f in freq:
        x[:, f0:f0 + f, :] *= 0.0
    for t0, t in time:
        x[t0:t0 + t, :, :] *= 0.0
    return x


def delta_accelerate(x):
    """data.py:319-324: delta[t] = x[t]-x[t-1] with x[-1]=0; deltas likewise on delta; concat."""
    z = np.zeros_like(x[:1])
    d = x - np.concatenate([z, x[:-1]], axis=0)
    dd = d - np.concatenate([z, d[:-1]], axis=0)
    return np.concatenate([x, d, dd], axis=2)


def batch_features(audio, n_samples, cfg, seed=0, spec_aug=None, use_delta=True, T_out=None):
    """Whole front end for a padded batch: audio [B, Nmax], n_samples [B] -> [B, T, mel, C].

    Per clip: log-mel (a1) -> SpecAugment (a2, run/train.py:99-110) -> delta (a3,
    run/train.py:113-116) -> zero padding to the batch maximum (a4, run/train.py:189-197).
    cfg: dict with sample_rate, frame_length, frame_step, fft_length, num_mel_bins,
    lower_edge_hertz, upper_edge_hertz.  spec_aug: dict(F, m_F, T, p, m_T) or None."""
    B = audio.shape[0]
    Ts = [num_frames(int(n), cfg["frame_length"], cfg["frame_step"]) for n in n_samples]
    T_out = T_out or max(Ts)
    C = 3 if use_delta else 1
    ftype = cfg.get("feature_type", "log-mel-spectrogram")            # data_config.py:77-101
    v = {"log-mel-spectrogram": cfg.get("num_mel_bins"), "spectrogram": cfg["fft_length"] // 2 + 1, "mfcc": cfg.get("num_mfcc")}[ftype]
    out = np.zeros((B, T_out, v, C), dtype=np.float64)
    for b in range(B):
        clip = audio[b, : int(n_samples[b])]
        if ftype == "spectrogram":
            x = spectrogram(clip, cfg["frame_length"], cfg["frame_step"], cfg["fft_length"])
        elif ftype == "mfcc":
            x = mfcc(clip, cfg["sample_rate"], cfg["frame_length"], cfg["frame_step"], cfg["fft_length"], cfg["num_mel_bins"],
                     cfg["num_mfcc"], cfg["lower_edge_hertz"], cfg["upper_edge_hertz"])
        else:
            x = log_mel_spectrogram(clip, cfg["sample_rate"], cfg["frame_length"],
                                    cfg["frame_step"], cfg["fft_length"], cfg["num_mel_bins"],
                                    cfg["lower_edge_hertz"], cfg["upper_edge_hertz"])
        if spec_aug:
            fr, tm = spec_augment_params(seed, b, Ts[b], v, spec_aug.get("F"), spec_aug.get("m_F"),
                                         spec_aug.get("T"), spec_aug.get("p"), spec_aug.get("m_T"))
            x = spec_augment(x, fr, tm)
        if use_delta:
            x = delta_accelerate(x)
        out[b, : Ts[b]] = x
    return out
