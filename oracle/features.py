"""Oracle (TEST INFRASTRUCTURE): audio front end of the training path, numpy float64.

Follows /root/reference/speech_recognition/data.py:
  make_spectrogram          data.py:122-142   (tf.signal.stft, tf.abs)
  make_log_mel_spectrogram  data.py:145-189   (tf.signal.stft / linear_to_mel_weight_matrix [TF-sem])
  make_mfcc                 data.py:192-241   (+ tf.signal.mfccs_from_log_mel_spectrograms [TF-sem])
  spec_augment              data.py:244-307   (time warp data.py:275-280 = tfa.image.sparse_image_warp [TF-sem],
                                               restated below from tensorflow-addons' published algorithm)
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


STREAM_TIMEWARP = 5  # RNG stream id of the two time-warp draws (index 2*clip, 2*clip + 1)


def time_warp_params(seed, clip, num_time, W):
    """data.py:276-277: src = uniform((), W, num_time - W), dst = src + uniform((), -W, W) with the build's RNG.
    None when the first range is empty (the reference raises there; the build leaves the clip unchanged)."""
    if num_time <= 2 * W or num_time < 2:
        return None
    src = W + rng.uniform_int(seed, STREAM_TIMEWARP, 2 * clip, num_time - 2 * W)
    dst = src - W + rng.uniform_int(seed, STREAM_TIMEWARP, 2 * clip + 1, 2 * W)
    return src, dst


def _phi2(r):
    """tfa interpolate_spline._phi, order 2, on SQUARED distances: 0.5 r log(max(r, 1e-10))."""
    return 0.5 * r * np.log(np.maximum(r, 1e-10))


def sparse_image_warp(image, src, dst, num_boundary_points=3):
    """[TF-sem] tfa.image.sparse_image_warp(image [H, W, C], src [n, 2], dst [n, 2] as (y, x), interpolation_order=2,
    regularization_weight=0): zero-flow control points on the border of a (num_boundary_points + 1)^2 grid are added,
    a polyharmonic spline through the flows (dst - src) AT THE DESTINATION points is solved
    ([[phi(D), P], [P^T, 0]] [w; v] = [f; 0], P = [y, x, 1]) and evaluated on every pixel, and
    tfa.image.dense_image_warp reads image[y - flow_y, x - flow_x] bilinearly (floor clamped to [0, size - 2],
    weight clamped to [0, 1]).  float64 throughout (TensorFlow: float32)."""
    image = np.asarray(image, np.float64)
    H, Wd, _ = image.shape
    src, dst = np.asarray(src, np.float64).reshape(-1, 2), np.asarray(dst, np.float64).reshape(-1, 2)
    flows = dst - src
    k = num_boundary_points - 1
    ys, xs = np.meshgrid(np.linspace(0, H - 1, k + 2), np.linspace(0, Wd - 1, k + 2), indexing="ij")
    border = (xs == 0) | (xs == Wd - 1) | (ys == 0) | (ys == H - 1)
    bpts = np.stack([ys[border], xs[border]], axis=-1).astype(np.float32).astype(np.float64)
    pts = np.concatenate([dst, bpts], axis=0)
    f = np.concatenate([flows, np.zeros_like(bpts)], axis=0)
    n = pts.shape[0]
    d2 = ((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1)
    P = np.concatenate([pts, np.ones((n, 1))], axis=1)
    lhs = np.block([[_phi2(d2), P], [P.T, np.zeros((3, 3))]])
    rhs = np.concatenate([f, np.zeros((3, 2))], axis=0)
    wv = np.linalg.solve(lhs, rhs)
    w, v = wv[:n], wv[n:]
    gy, gx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(Wd, dtype=np.float64), indexing="ij")
    q = np.stack([gy.ravel(), gx.ravel()], axis=1)
    qd2 = ((q[:, None, :] - pts[None, :, :]) ** 2).sum(-1)
    flow = _phi2(qd2) @ w + np.concatenate([q, np.ones((q.shape[0], 1))], axis=1) @ v
    query = q - flow
    fl, al = [], []
    for dim, size in ((0, H), (1, Wd)):
        fdim = np.minimum(np.maximum(0.0, np.floor(query[:, dim])), size - 2)
        fl.append(fdim.astype(np.int64))
        al.append(np.clip(query[:, dim] - fdim, 0.0, 1.0)[:, None])
    tl, tr = image[fl[0], fl[1]], image[fl[0], fl[1] + 1]
    bl, br = image[fl[0] + 1, fl[1]], image[fl[0] + 1, fl[1] + 1]
    top = al[1] * (tr - tl) + tl
    bot = al[1] * (br - bl) + bl
    return (al[0] * (bot - top) + top).reshape(image.shape), flow.reshape(H, Wd, 2)


def time_warp(x, src_time, dst_time):
    """data.py:278-280: one control point (src_time, v // 2) -> (dst_time, v // 2), num_boundary_points=3."""
    v = x.shape[1]
    out, _ = sparse_image_warp(x, [[src_time, v // 2]], [[dst_time, v // 2]], 3)
    return out


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
        if spec_aug and spec_aug.get("W"):                            # data.py:275-280, before the masks
            tw = time_warp_params(seed, b, Ts[b], spec_aug["W"])
            if tw is not None:
                x = time_warp(x, tw[0], tw[1])
        if spec_aug:
            fr, tm = spec_augment_params(seed, b, Ts[b], v, spec_aug.get("F"), spec_aug.get("m_F"),
                                         spec_aug.get("T"), spec_aug.get("p"), spec_aug.get("m_T"))
            x = spec_augment(x, fr, tm)
        if use_delta:
            x = delta_accelerate(x)
        out[b, : Ts[b]] = x
    return out
