"""Thin tensor-level wrappers over the C ABI (include/asr_mi355x.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every computation below
is a hand-written gfx950 kernel inside libasr_mi355x.so.  All wrappers launch on
``torch.cuda.current_stream()`` and never synchronise.
"""
import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import RNN_TYPES, check


def lib():
    return _lib.load()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _dev(t, dtype=torch.float32, name="tensor"):
    if t is None:
        return
    if not t.is_cuda:
        raise ValueError(f"{name} must live on the GPU (speech_recognition_amd has no CPU path)")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")


def rnn_type_id(rnn_type: str) -> int:
    if rnn_type not in RNN_TYPES:
        raise ValueError(f"rnn_type: {rnn_type} is invalid!")  # same message as las.py:17
    return RNN_TYPES[rnn_type]


# ----------------------------------------------------------------------------------------- front end
class LogmelPlan:
    """Constant tables of the fused front-end kernel for one DataConfig (built once, on host in
    float64 inside the library, uploaded once)."""

    FEATURE_TYPES = {"log-mel-spectrogram": 0, "spectrogram": 1, "mfcc": 2}

    def __init__(self, sample_rate, frame_length, frame_step, fft_length, num_mel_bins=80, lower_edge_hertz=80.0,
                 upper_edge_hertz=7600.0, epsilon=1e-12, use_delta=True, spec_augment=None, device="cuda",
                 feature_type="log-mel-spectrogram", num_mfcc=None):
        """feature_type (data_config.py:77-101): "log-mel-spectrogram" (data.py:145-189), "spectrogram"
        (data.py:122-142, the mel arguments are ignored) or "mfcc" (data.py:192-241, first num_mfcc coefficients)."""
        if feature_type not in self.FEATURE_TYPES:
            raise ValueError(f"feature_type {feature_type!r}: expected one of {sorted(self.FEATURE_TYPES)}")
        ft = self.FEATURE_TYPES[feature_type]
        num_mel_bins = int(num_mel_bins or 0)
        sa = spec_augment or {}
        enable = bool(spec_augment) and bool(sa.get("enable", True))
        # SpecAugment's time warp (data.py:275-280) resamples whole clips along time, so it cannot live in the fused
        # kernel's per-tile pass: the kernel then emits plain features and a StoredFeaturePlan does warp -> masks -> delta
        self._tail = None
        if enable and sa.get("W"):
            self._tail = StoredFeaturePlan(0, use_delta, spec_augment)       # v is filled in below
            self._final_channels = 3 if use_delta else 1
            enable, use_delta = False, False
        use_f = bool(sa.get("F")) and bool(sa.get("m_F"))
        use_t = bool(sa.get("T")) and bool(sa.get("p")) and bool(sa.get("m_T"))
        self.cfg = _lib.LogmelCfg(sample_rate, frame_length, frame_step, fft_length, num_mel_bins, lower_edge_hertz,
                                  upper_edge_hertz, epsilon, 1 if use_delta else 0, 1 if enable else 0,
                                  int(sa.get("F") or 0) if use_f else 0, int(sa.get("m_F") or 0) if use_f else 0,
                                  int(sa.get("T") or 0) if use_t else 0, int(sa.get("m_T") or 0) if use_t else 0,
                                  float(sa.get("p") or 0.0) if use_t else 0.0, ft, int(num_mfcc or 0))
        n1, n2, n3 = C.c_long(), C.c_long(), C.c_long()
        check(lib().asr_logmel_table_sizes(C.byref(self.cfg), C.byref(n1), C.byref(n2), C.byref(n3)))
        tw = np.empty(n1.value, np.float32)
        mw = np.empty(n2.value, np.float32)
        mr = np.empty(n3.value, np.int32)
        check(lib().asr_logmel_build_tables(C.byref(self.cfg), tw.ctypes.data_as(C.c_void_p), mw.ctypes.data_as(C.c_void_p),
                                            mr.ctypes.data_as(C.c_void_p)))
        bins = fft_length // 2 + 1
        mel_cols = 1 if ft == 1 else num_mel_bins                # spectrogram: the library keeps one dummy mel column
        self.melw_host = mw[:bins * mel_cols].reshape(bins, mel_cols)
        # features per frame (the reference's frequency_dim, data_config.py:65-74); `num_mel_bins` keeps naming it for
        # the callers that size their buffers with it
        self.num_features = (num_mel_bins, bins, int(num_mfcc or 0))[ft]
        self.feature_type = feature_type
        self.tw = torch.from_numpy(tw).to(device)
        self.melw = torch.from_numpy(mw).to(device)
        self.melrange = torch.from_numpy(mr).to(device)
        self.channels = 3 if use_delta else 1
        self.num_mel_bins = self.num_features
        self.frame_length, self.frame_step = frame_length, frame_step
        if self._tail is not None:
            self._tail.set_num_features(self.num_features)
            self.channels = self._final_channels
            self._raw = {}

    def num_frames(self, n_samples: int) -> int:
        return 0 if n_samples < self.frame_length else 1 + (n_samples - self.frame_length) // self.frame_step

    def __call__(self, audio: torch.Tensor, n_samples: torch.Tensor, T_out: int, seed: Optional[torch.Tensor] = None,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """audio [B, n_max] f32, n_samples [B] i32 -> [B, T_out, mel, C] f32 (zero beyond each clip)."""
        _dev(audio, name="audio")
        _dev(n_samples, torch.int32, "n_samples")
        _dev(seed, torch.int32, "seed")
        B, n_max = audio.shape
        if out is None:
            out = torch.empty(B, T_out, self.num_mel_bins, self.channels, device=audio.device, dtype=torch.float32)
        assert audio.is_contiguous() and out.is_contiguous()
        dst = out
        if self._tail is not None:
            key = (B, T_out)
            if key not in self._raw:
                self._raw[key] = torch.empty(B, T_out, self.num_mel_bins, 1, device=audio.device, dtype=torch.float32)
            dst = self._raw[key]
        check(lib().asr_logmel_features(C.byref(self.cfg), _p(audio), _p(n_samples), B, n_max, _p(self.tw), _p(self.melw),
                                        _p(self.melrange), _p(seed), _p(dst), T_out, _stream()))
        if self._tail is not None:
            # frames per clip, on the device (tf.signal.frame without end padding)
            n_frames = torch.where(n_samples >= self.frame_length,
                                   torch.div(n_samples - self.frame_length, self.frame_step, rounding_mode="floor") + 1,
                                   torch.zeros_like(n_samples)).to(torch.int32)
            self._tail(dst, n_frames, T_out, seed, out)
        return out


class StoredFeaturePlan:
    """Front end for batches that arrive as stored log-mel frames [B, T, v, 1] (run/train.py:70-74
    --use-tfrecord): SpecAugment (training) and delta/delta-delta on the device.  Same call interface as
    LogmelPlan, with `audio` = the stored features and `n_samples` = frames per clip."""

    def __init__(self, num_mel_bins, use_delta=True, spec_augment=None):
        sa = spec_augment or {}
        enable = bool(spec_augment) and bool(sa.get("enable", True))
        self._sa = sa if enable else {}
        self.W = int(sa.get("W") or 0) if enable else 0          # data.py:269: use_time_warping = bool(W)
        self.use_delta = use_delta
        self.channels = 3 if use_delta else 1
        self._scratch = {}
        self.set_num_features(num_mel_bins)

    def set_num_features(self, v):
        sa = self._sa
        self.cfg = spec_augment_cfg(v, sa.get("F"), sa.get("m_F"), sa.get("T"), sa.get("p"), sa.get("m_T")) if (sa and v) \
            else spec_augment_cfg(max(int(v), 1))
        self.num_mel_bins = v

    def num_frames(self, n_frames: int) -> int:
        return n_frames

    def __call__(self, feats, n_frames, T_out, seed=None, out=None):
        _dev(feats, name="feats")
        B, T = feats.shape[:2]
        assert T_out == T and feats.is_contiguous()
        if out is None:
            out = torch.empty(B, T, self.num_mel_bins, self.channels, device=feats.device, dtype=torch.float32)
        src = feats
        if self.cfg.sa_enable or self.W:
            if seed is None:
                raise ValueError("StoredFeaturePlan: SpecAugment needs a device seed")
            if self.use_delta:
                key = tuple(feats.shape)
                if key not in self._scratch:
                    self._scratch[key] = torch.empty_like(feats)
                src = self._scratch[key]
            else:
                src = out.view(feats.shape)
            if self.W:                                           # the warp gathers: feats -> src, then masks in place
                time_warp(feats.view(B, T, self.num_mel_bins, 1), n_frames, self.W, seed, src.view(B, T, self.num_mel_bins, 1))
            else:
                src.copy_(feats)
            if self.cfg.sa_enable:
                spec_augment_(self.cfg, src.view(B, T, self.num_mel_bins, 1), n_frames, seed)
        if self.use_delta:
            delta_accelerate(src, n_frames, out)
        elif src is feats:
            out.view(feats.shape).copy_(feats)
        return out


def spec_augment_cfg(v, F=None, m_F=None, T=None, p=None, m_T=None):
    """asr_logmel_cfg carrying only the SpecAugment fields (for asr_spec_augment on stored features)."""
    use_f, use_t = all([F, m_F]), all([T, p, m_T])
    return _lib.LogmelCfg(0, 0, 0, 0, int(v), 0.0, 0.0, 0.0, 0, 1 if (use_f or use_t) else 0, int(F) if use_f else 0,
                          int(m_F) if use_f else 0, int(T) if use_t else 0, int(m_T) if use_t else 0, float(p) if use_t else 0.0, 0, 0)


def spec_augment_(cfg, x, n_frames, seed):
    """In-place SpecAugment of x [B, T, v, C]; n_frames i32 [B] or None; seed: device i32/u32 [>=1]."""
    _dev(x, name="x")
    _dev(n_frames, torch.int32, "n_frames")
    _dev(seed, torch.int32, "seed")
    assert x.dim() == 4 and x.is_contiguous() and x.shape[2] == cfg.num_mel_bins
    check(lib().asr_spec_augment(C.byref(cfg), _p(x), _p(n_frames), x.shape[0], x.shape[1], x.shape[3], _p(seed), _stream()))
    return x


_tw_coef = {}


def time_warp(x, n_frames, W, seed, out=None):
    """SpecAugment time warping (data.py:275-280) of x [B, T, v, C] into `out` (a new tensor by default);
    n_frames i32 [B] or None; seed: device i32/u32 [>=1]."""
    _dev(x, name="x")
    _dev(n_frames, torch.int32, "n_frames")
    _dev(seed, torch.int32, "seed")
    assert x.dim() == 4 and x.is_contiguous()
    B, T, v, Cc = x.shape
    if out is None:
        out = torch.empty_like(x)
    assert out.is_contiguous() and out.data_ptr() != x.data_ptr()
    key = (B, x.device)
    if key not in _tw_coef:
        _tw_coef[key] = torch.zeros(B, 32, device=x.device, dtype=torch.float32)
    check(lib().asr_time_warp(_p(x), _p(n_frames), B, T, v, Cc, int(W), _p(seed), _p(_tw_coef[key]), _p(out), _stream()))
    return out


def delta_accelerate(x, n_frames=None, out=None):
    """x [B, T, v] (or [B, T, v, 1]) -> [B, T, v, 3] = (x, delta, delta-delta)."""
    _dev(x, name="x")
    _dev(n_frames, torch.int32, "n_frames")
    assert x.is_contiguous() and (x.dim() == 3 or (x.dim() == 4 and x.shape[3] == 1))
    B, T, v = x.shape[:3]
    if out is None:
        out = torch.empty(B, T, v, 3, device=x.device, dtype=torch.float32)
    check(lib().asr_delta_accelerate(_p(x), _p(n_frames), B, T, v, _p(out), _stream()))
    return out


# ----------------------------------------------------------------------------------------- GEMM
def _mat(t, name):
    _dev(t, name=name)
    if t.dim() not in (2, 3) or t.stride(-1) != 1:
        raise ValueError(f"{name}: need a 2-D/3-D tensor with unit inner stride, got shape {tuple(t.shape)} stride {t.stride()}")
    return t


_compute = {"gemm": 0}
# How an f32 product is evaluated when mixed precision is off (asr_gemm_desc.compute): "mfma" = 0, the f32 MFMA (exact products,
# 1/16 of the bf16 matrix rate on gfx950); "split9" = 2, every product as the nine bf16 pair products of exact three-way splits
# of both operands on the bf16 MFMA (2^-32 relative per product - tighter than an f32 FMA chain - f32 accumulation); "split6" = 3,
# the same without the three pairs of weight <= 2^-24.  ASR_GEMM_F32 selects it.
_F32_MODES = {"mfma": 0, "split9": 2, "split6": 3}
if os.environ.get("ASR_GEMM_F32", "split6") not in _F32_MODES:
    raise ValueError(f"ASR_GEMM_F32={os.environ['ASR_GEMM_F32']!r}: expected one of {sorted(_F32_MODES)}")
_f32_mode = {"compute": _F32_MODES[os.environ.get("ASR_GEMM_F32", "split6")]}


def set_f32_gemm_mode(name: str):
    """'mfma' | 'split9' | 'split6' (see above); returns the previous name."""
    old = next(k for k, v in _F32_MODES.items() if v == _f32_mode["compute"])
    _f32_mode["compute"] = _F32_MODES[name]
    prev = lib().asr_set_f32_product_mode(_F32_MODES[name])          # the convolutions follow (process-wide switch of the library)
    if prev < 0:                                                        # (>= 0: the previous mode; negative: an asr_status)
        check(prev)
    return old


def f32_gemm_mode() -> str:
    return next(k for k, v in _F32_MODES.items() if v == _f32_mode["compute"])


def set_mixed_precision(on: bool):
    """train.py:62-66 --mixed-precision on MI355X: every dense contraction that goes through `gemm` (input and
    output projections, attention keys, vocabulary layer, and their gradients) rounds its operands to bf16 on the
    way into the bf16 MFMA, and the wide (H >= 512) recurrent step kernels read bf16 images of their weights
    (ParamStore.refresh_bf16 / PackedCell.pack); storage, accumulation, epilogues, gate math, the other recurrent
    kernels, convolutions, softmax / CTC, batch norm and Adam stay f32 (f32 master weights).  A process-wide switch,
    like the Keras global policy."""
    _compute["gemm"] = 1 if on else 0


def mixed_precision() -> bool:
    return bool(_compute["gemm"])


# bf16 images of f32 weight buffers (the flat parameter buffer of a model): the wide recurrent step kernels read their
# weights from the image when the mixed-precision switch is on.  (flat f32 tensor, bf16 image of the same length)
_mirrors = []


def register_bf16_mirror(flat: torch.Tensor, image: torch.Tensor):
    _mirrors[:] = [(f, m) for f, m in _mirrors if f.data_ptr() != flat.data_ptr()]
    _mirrors.append((flat, image))


def bf16_twin(t: Optional[torch.Tensor]) -> Optional[int]:
    """Address of the bf16 image of the f32 view `t` (same element offset inside a registered mirror), or None."""
    if t is None or not _compute["gemm"]:
        return None
    ptr = t.data_ptr()
    for flat, image in _mirrors:
        base = flat.data_ptr()
        if base <= ptr < base + flat.numel() * 4:
            return image.data_ptr() + (ptr - base) // 2
    return None


def f32_to_bf16(src: torch.Tensor, dst: torch.Tensor):
    """dst (bfloat16, same numel) = round-to-nearest-even of src (float32), on the current stream."""
    _dev(src, name="src")
    assert dst.dtype == torch.bfloat16 and dst.numel() == src.numel() and src.is_contiguous() and dst.is_contiguous()
    check(lib().asr_f32_to_bf16(_p(src), C.c_void_p(dst.data_ptr()), src.numel(), _stream()))
    return dst


def gemm(a, b, c, *, trans_a=False, trans_b=False, alpha=1.0, accumulate=0, bias=None, relu=False, a_scale=None,
         a_rpg=0, c_scale=None, c_rpg=0, split_k=1, compute=None, a_scale_stride=0):
    """c (+)= alpha * op(a) @ op(b) (+ bias).  2-D operands, or 3-D with a leading batch axis; a 2-D `c`
    with 3-D a/b means split-K over the batch axis (atomic accumulation, requires accumulate).
    compute: None = the process-wide mode (set_mixed_precision), 0 = f32 operands, 1 = bf16 operands."""
    _mat(a, "a"), _mat(b, "b"), _mat(c, "c")
    batch = a.shape[0] if a.dim() == 3 else 1
    if b.dim() == 3 and a.dim() == 3:
        assert b.shape[0] == batch
    am, ak = (a.shape[-1], a.shape[-2]) if trans_a else (a.shape[-2], a.shape[-1])
    bk, bn = (b.shape[-1], b.shape[-2]) if trans_b else (b.shape[-2], b.shape[-1])
    if ak != bk or c.shape[-2] != am or c.shape[-1] != bn:
        raise ValueError(f"gemm shape mismatch: op(a) [{am},{ak}] op(b) [{bk},{bn}] c {tuple(c.shape)}")
    d = _lib.GemmDesc()
    d.trans_a, d.trans_b = int(trans_a), int(trans_b)
    d.M, d.N, d.K, d.batch = am, bn, ak, batch
    d.split_k = int(split_k)
    d.lda, d.ldb, d.ldc = a.stride(-2), b.stride(-2), c.stride(-2)
    d.stride_a = a.stride(0) if a.dim() == 3 else 0
    d.stride_b = b.stride(0) if b.dim() == 3 else 0
    d.stride_c = c.stride(0) if c.dim() == 3 else 0
    d.stride_a_scale = int(a_scale_stride)     # per-batch offset (elements) into a_scale for 3-D operands
    d.alpha = alpha
    d.accumulate = int(accumulate)
    if batch > 1 and c.dim() == 2:
        if not accumulate:
            raise ValueError("split-K gemm (2-D c, batched a/b) accumulates atomically: pass accumulate=1 and pre-zero c")
        d.accumulate = 2
    d.relu = int(relu)
    d.bias = bias.data_ptr() if bias is not None else None
    d.a_scale = a_scale.data_ptr() if a_scale is not None else None
    d.a_rpg = int(a_rpg)
    d.c_scale = c_scale.data_ptr() if c_scale is not None else None
    d.c_rpg = int(c_rpg)
    d.compute = (_compute["gemm"] or _f32_mode["compute"]) if compute is None else int(compute)
    if d.compute == 1 and _bf16_images["on"] and _gemm_bf16_images(d, a, b, c, trans_a, trans_b, a_scale, a_rpg, a_scale_stride):
        return c
    check(lib().asr_gemm_f32(C.byref(d), _p(a), _p(b), _p(c), _stream()))
    return c


# bf16 operand images for the large products under mixed precision (csrc/gemm16.hip): both operands are rewritten as k-contiguous
# bf16 matrices by a memory-bound pass, then multiplied by the bf16-operand kernel.  Pays when every dimension is large (the wide
# models); the small products keep the f32-operand kernel (whose fragments are rounded to bf16 on the way out of LDS: same numerics).
_bf16_images = {"on": os.environ.get("ASR_GEMM_BF16_IMAGES", "1") != "0", "min_dim": int(os.environ.get("ASR_GEMM_BF16_MIN", "512")),
                "scratch": {}}


def _image_scratch(role, rows, cols, layout=None):
    """bf16 [rows, cols8] scratch, one per (role, stream, rows, EXACT cols, layout): GEMMs of one stream run in order, so consecutive
    products of one shape may share it; products on different streams never do.  The key carries the exact contraction length, not
    the padded one, and `layout` (the (B, T) of a shifted image): the image passes never write the k padding [cols, cols8) nor the
    'no predecessor' column of a shifted image, which must read as 0 - two products that only agree on the padded size (K = 1004 and
    K = 1008; (B, T) and (2B, T/2)) would otherwise read each other's stale values there."""
    cols8 = (cols + 7) // 8 * 8
    key = (role, torch.cuda.current_stream().cuda_stream, rows, cols, layout)
    buf = _bf16_images["scratch"].get(key)
    if buf is None:
        # zero-filled once: what the image pass of THIS shape never writes stays 0 for the life of the buffer
        buf = _bf16_images["scratch"][key] = torch.zeros(rows, cols8, device="cuda", dtype=torch.bfloat16)
    return buf


def f32_to_bf16_image(src, dst, *, transpose=False, scale=None, rows_per_group=0, dst_rows_per_batch=0, dst_shift=0):
    """dst (bf16 [rows, >= cols] or, transposed, [cols, >= rows]) = bf16(src * scale[row / rows_per_group]).  src: f32 2-D with unit
    inner stride, or 3-D [batch, rows, cols] (flattened row-major over (batch, row)).  dst_rows_per_batch / dst_shift (transposed, 3-D
    source): batch b's rows land at columns b * dst_rows_per_batch + dst_shift + row; the other columns are left alone."""
    assert src.dtype == torch.float32 and dst.dtype == torch.bfloat16 and src.stride(-1) == 1 and dst.stride(-1) == 1
    if src.dim() == 3:
        rows, rpb, bstr = src.shape[0] * src.shape[1], src.shape[1], src.stride(0)
    else:
        rows, rpb, bstr = src.shape[0], 0, 0
    check(lib().asr_f32_to_bf16_image(_p(src), src.stride(-2), rows, src.shape[-1], rpb, bstr, _p(scale) if scale is not None else None,
                                      int(rows_per_group), int(transpose), C.c_void_p(dst.data_ptr()), dst.stride(0), int(dst_rows_per_batch),
                                      int(dst_shift), _stream()))
    return dst


def f32_to_bf16_image_tb(src3d, dst, *, scale=None, dst_shift=0):
    """Transposed bf16 image with TIME-MAJOR columns (asr_f32_to_bf16_image_tb): src3d f32 [nbatch, rows, cols] (any batch / row strides, unit
    inner stride) -> dst bf16 [cols, >= nbatch * rows + dst_shift], dst[c][t * nbatch + b + dst_shift] = bf16(src3d[b, t, c] * scale[b, c]) - the
    column order of the transposed ds image the wide BPTT sweep writes."""
    assert src3d.dtype == torch.float32 and dst.dtype == torch.bfloat16 and src3d.dim() == 3 and src3d.stride(2) == 1 and dst.stride(-1) == 1
    nb, rows, cols = src3d.shape
    check(lib().asr_f32_to_bf16_image_tb(_p(src3d), src3d.stride(1), nb, rows, cols, src3d.stride(0), _p(scale) if scale is not None else None,
                                         C.c_void_p(dst.data_ptr()), dst.stride(0), int(dst_shift), _stream()))
    return dst


def bf16_images_pay(M, N, K) -> bool:
    """Whether ops.gemm would route an [M,K] x [K,N] product through bf16 images (mixed precision, large in every dimension)."""
    md = _bf16_images["min_dim"]
    return bool(mixed_precision() and _bf16_images["on"] and M >= md // 2 and N >= md // 2 and K >= md and M * N >= md * md)


def image_scratch(role, rows, cols, layout=None):
    """A zero-initialised bf16 [rows, cols rounded up to 8] scratch per (role, stream, exact shape, layout) - see _image_scratch."""
    return _image_scratch(role, rows, cols, layout)


def gemm_bf16_nt(a16, b16, c, *, alpha=1.0, accumulate=0, bias=None, relu=False, c_scale=None, c_rpg=0, split_k=1, K=None):
    """c [M,N] (+)= alpha * a16 [M,K] @ b16 [N,K]^T (+ bias): bf16 operands in memory, f32 accumulation (asr_gemm_bf16_nt)."""
    assert a16.dtype == torch.bfloat16 and b16.dtype == torch.bfloat16 and c.dtype == torch.float32
    d = _lib.GemmDesc()
    d.trans_a, d.trans_b = 0, 1
    d.M, d.N, d.K, d.batch = a16.shape[0], b16.shape[0], int(K if K is not None else a16.shape[1]), 1
    d.split_k = int(split_k)
    d.lda, d.ldb, d.ldc = a16.stride(0), b16.stride(0), c.stride(0)
    d.alpha, d.accumulate, d.relu = alpha, int(accumulate), int(relu)
    d.bias = bias.data_ptr() if bias is not None else None
    d.c_scale = c_scale.data_ptr() if c_scale is not None else None
    d.c_rpg = int(c_rpg)
    d.compute = 1
    check(lib().asr_gemm_bf16_nt(C.byref(d), C.c_void_p(a16.data_ptr()), C.c_void_p(b16.data_ptr()), _p(c), _stream()))
    return c


def _gemm_bf16_images(d, a, b, c, trans_a, trans_b, a_scale, a_rpg, a_scale_stride):
    """Route one product through bf16 images if it qualifies; returns False (nothing done) otherwise."""
    md = _bf16_images["min_dim"]
    flat = a.dim() == 3 and b.dim() == 3 and c.dim() == 2 and trans_a and not trans_b          # split-K over the batch: one long K
    if a.dim() != b.dim() or (a.dim() == 3 and not flat) or c.dim() != 2:
        return False
    K = d.K * d.batch if flat else d.K
    if d.M < md // 2 or d.N < md // 2 or K < md or d.M * d.N < md * md:
        return False
    if a.stride(-1) != 1 or b.stride(-1) != 1 or c.stride(-1) != 1:
        return False
    K8 = (K + 7) // 8 * 8
    scale, rpg = a_scale, a_rpg
    if flat and a_scale is not None:
        if a_scale_stride != a.shape[-1] or a_rpg < a.shape[1]:
            return False
        rpg = a.shape[1]                                          # one scale row per batch
    # straight images read float4s
    for t, tr in ((a, trans_a), (b, not trans_b)):
        if not tr and (t.shape[-1] % 4 or t.stride(-2) % 4 or t.data_ptr() % 16):
            return False
    if not trans_a and scale is not None and scale.data_ptr() % 16:
        return False
    a16 = f32_to_bf16_image(a, _image_scratch("a", d.M, K), transpose=trans_a, scale=scale, rows_per_group=rpg)
    b16 = f32_to_bf16_image(b, _image_scratch("b", d.N, K), transpose=not trans_b)
    acc = 1 if d.accumulate else 0                                # (the batch split became one product: no atomics needed)
    if d.split_k > 1:
        acc = 2
    d2 = _lib.GemmDesc()
    d2.trans_a, d2.trans_b = 0, 1
    d2.M, d2.N, d2.K, d2.batch = d.M, d.N, K8, 1
    d2.split_k = d.split_k
    d2.lda, d2.ldb, d2.ldc = a16.stride(0), b16.stride(0), c.stride(0)
    d2.alpha, d2.accumulate, d2.relu = d.alpha, acc, d.relu
    d2.bias, d2.c_scale, d2.c_rpg, d2.compute = d.bias, d.c_scale, d.c_rpg, 1
    check(lib().asr_gemm_bf16_nt(C.byref(d2), C.c_void_p(a16.data_ptr()), C.c_void_p(b16.data_ptr()), _p(c), _stream()))
    return True


def bf16_to_f32(src: torch.Tensor, dst: torch.Tensor):
    """dst (float32) = src (bfloat16, same numel), on the current stream."""
    assert src.dtype == torch.bfloat16 and dst.dtype == torch.float32 and dst.numel() == src.numel() and src.is_contiguous() and dst.is_contiguous()
    check(lib().asr_bf16_to_f32(C.c_void_p(src.data_ptr()), _p(dst), src.numel(), _stream()))
    return dst


def debug_occupy(blocks: int, threads: int, microseconds: int):
    """Diagnostic: keep `blocks` workgroups busy for `microseconds` on the current stream (asr_debug_occupy)."""
    check(lib().asr_debug_occupy(int(blocks), int(threads), int(microseconds), _stream()))


# Whole-chip sweeps (the decoder sweeps and the wide layers' sweeps: one workgroup per compute unit, all of which wait for each
# other) need the device to themselves: a second PROCESS running the same grids on the same GPU can hold part of the chip and
# starve this one until the start handshake gives up ("absent workgroup", sweep_common.h) - nothing in HIP gang-schedules across
# processes.  One process per GPU (run.train, bench.py) is exclusive by construction; TrainStep detects ranks that share a
# device and clears this switch, and the models then keep those loops on the per-step kernels (the encoder sweeps stay: they are
# sized to 3/4 of the chip, asr_sweep_capacity).
_tenancy = {"exclusive": True}


def set_device_exclusive(flag: bool):
    _tenancy["exclusive"] = bool(flag)


def device_exclusive() -> bool:
    return _tenancy["exclusive"]


def debug_stream_memory(buf: torch.Tensor, blocks: int, microseconds: int):
    """Diagnostic: `blocks` workgroups stream `buf` (f32, first half copied onto the second) for `microseconds` on the current
    stream (asr_debug_stream_memory): a memory-side co-tenant."""
    assert buf.dtype == torch.float32 and buf.is_contiguous()
    check(lib().asr_debug_stream_memory(_p(buf), buf.numel() * 4, int(blocks), int(microseconds), _stream()))


# ----------------------------------------------------------------------------------------- recurrent layers
def rnn_geometry(rnn_type: str, H: int, Ks: Sequence[int]) -> _lib.RnnGeom:
    g = _lib.RnnGeom()
    arr = (C.c_int * len(Ks))(*Ks)
    check(lib().asr_rnn_geometry(rnn_type_id(rnn_type), H, len(Ks), arr, C.byref(g)))
    return g


class PackedCell:
    """MFMA-fragment-order image of the weights multiplying a cell's concatenated inputs.
    `weights`: list of (W [K_i, G*H] view with unit inner stride, is_recurrent)."""

    def __init__(self, rnn_type: str, H: int, Ks: Sequence[int], device="cuda"):
        self.rnn_type, self.H, self.Ks = rnn_type, H, list(Ks)
        self.geom = rnn_geometry(rnn_type, H, Ks)
        self.Wp = torch.empty(self.geom.wp_floats, device=device, dtype=torch.float32)
        self.Wp16 = None                                         # bf16 image, kept fresh by pack() under mixed precision

    def pack(self, weights):
        n = len(weights)
        assert n == len(self.Ks)
        Wp = (C.c_void_p * n)(*[w.data_ptr() for w, _ in weights])
        ld = (C.c_long * n)(*[w.stride(0) for w, _ in weights])
        K = (C.c_int * n)(*self.Ks)
        rec = (C.c_int * n)(*[int(r) for _, r in weights])
        for (w, _), k in zip(weights, self.Ks):
            _dev(w, name="weight")
            assert w.shape[0] == k and w.stride(1) == 1
        check(lib().asr_rnn_pack(rnn_type_id(self.rnn_type), self.H, n, Wp, ld, K, rec, _p(self.Wp), _stream()))
        if mixed_precision():
            if self.Wp16 is None:
                self.Wp16 = torch.empty(self.Wp.numel(), device=self.Wp.device, dtype=torch.bfloat16)
            f32_to_bf16(self.Wp, self.Wp16)
        return self

    def wp16_ptr(self):
        return self.Wp16.data_ptr() if (self.Wp16 is not None and mixed_precision()) else None

    def pack_desc(self, weights):
        """asr_rnn_pack_desc of this cell for pack_cells()."""
        n = len(weights)
        assert n == len(self.Ks)
        d = _lib.RnnPackDesc()
        d.rnn_type, d.H, d.nseg = rnn_type_id(self.rnn_type), self.H, n
        for i, ((w, rec), k) in enumerate(zip(weights, self.Ks)):
            _dev(w, name="weight")
            assert w.shape[0] == k and w.stride(1) == 1
            d.W[i], d.ldw[i], d.K[i], d.is_rec[i] = w.data_ptr(), w.stride(0), k, int(rec)
        d.Wp = self.Wp.data_ptr()
        return d


def pack_cells(cells_and_weights):
    """PackedCell.pack for many cells in ONE launch (asr_rnn_pack_many): [(cell, weights), ...]."""
    if not cells_and_weights:
        return
    descs = (_lib.RnnPackDesc * len(cells_and_weights))(*[c.pack_desc(w) for c, w in cells_and_weights])
    check(lib().asr_rnn_pack_many(len(cells_and_weights), descs, _stream()))
    if mixed_precision():
        for c, _ in cells_and_weights:
            if c.Wp16 is None:
                c.Wp16 = torch.empty(c.Wp.numel(), device=c.Wp.device, dtype=torch.bfloat16)
            f32_to_bf16(c.Wp, c.Wp16)


def _arr2(vals, ctype=C.c_void_p):
    vals = list(vals) + [None] * (2 - len(vals))
    if ctype is C.c_void_p:
        return (C.c_void_p * 2)(*[None if v is None else (v.data_ptr() if isinstance(v, torch.Tensor) else v) for v in vals])
    return (ctype * 2)(*[0 if v is None else v for v in vals])


def make_rnn_seq(rnn_type, B, T, H, dirs, mask, y, y_cols):
    """dirs: list (1 or 2) of dicts with keys pre, cell(PackedCell), bias_rec, h0, c0, hseq, cseq, saved, reverse."""
    s = _lib.RnnSeq()
    s.rnn_type, s.B, s.T, s.H, s.ndir = rnn_type_id(rnn_type), B, T, H, len(dirs)
    s.reverse = (C.c_int * 2)(*([int(d.get("reverse", False)) for d in dirs] + [0] * (2 - len(dirs))))
    s.pre = _arr2([d["pre"] for d in dirs])
    s.Wp = _arr2([d["cell"].Wp for d in dirs])
    s.Wp16 = _arr2([d["cell"].wp16_ptr() for d in dirs])
    s.U16 = _arr2([bf16_twin(d.get("U")) for d in dirs])
    s.U = _arr2([d.get("U") for d in dirs])
    s.ldu = _arr2([d["U"].stride(0) if d.get("U") is not None else 0 for d in dirs], C.c_long)
    s.bias_rec = _arr2([d.get("bias_rec") for d in dirs])
    s.h0 = _arr2([d.get("h0") for d in dirs])
    s.h0_ld = _arr2([d["h0"].stride(0) if d.get("h0") is not None else 0 for d in dirs], C.c_long)
    s.c0 = _arr2([d.get("c0") for d in dirs])
    s.c0_ld = _arr2([d["c0"].stride(0) if d.get("c0") is not None else 0 for d in dirs], C.c_long)
    s.rec_mult = _arr2([d.get("rec_mult") for d in dirs])
    s.mask = mask.data_ptr() if mask is not None else None
    s.hseq = _arr2([d["hseq"] for d in dirs])
    s.cseq = _arr2([d.get("cseq") for d in dirs])
    s.y = y.data_ptr()
    s.y_ld = y.stride(1)
    s.y_col = (C.c_int * 2)(*(list(y_cols) + [0] * (2 - len(y_cols))))
    s.saved = _arr2([d.get("saved") for d in dirs])
    s.coef = _arr2([d.get("coef") for d in dirs])
    return s


def rnn_coef_width(rnn_type) -> int:
    """Floats per (row, step, unit) of the backward-coefficient buffer the forward sweep writes for the BPTT sweep (asr_rnn_seq.coef)."""
    return 4 if rnn_type == "rnn" else 8


def rnn_seq_fwd(seq):
    check(lib().asr_rnn_seq_fwd(C.byref(seq), _stream()))


def rnn_persist_supported(rnn_type, B, T, H, ndir=2):
    """True when the one-launch forward sweep (rnn_sweep.hip) takes this layer."""
    return bool(lib().asr_rnn_sweep_supported(rnn_type_id(rnn_type), B, T, H, ndir))


def rnn_persist_ws(B, H, ndir=2, device="cuda"):
    return torch.zeros(int(lib().asr_rnn_sweep_ws_floats(B, H, ndir)), device=device, dtype=torch.float32)


def rnn_seq_fwd_persist(seq, ws, err_flag=None):
    """One launch for the whole sequence; ws from rnn_persist_ws (error word = ws[-32] bits); err_flag: optional device
    float that is set to 1.0 when a hand-off times out (sticky: the library never clears it)."""
    check(lib().asr_rnn_sweep_fwd(C.byref(seq), _p(ws), _p(err_flag), _stream()))


def rnn_sweep_wide_supported(rnn_type, B, T, H, ndir=2) -> bool:
    """True when the wide one-launch forward sweep (rnn_sweep_wide.hip: bf16 weights resident, mixed precision) takes this layer."""
    return mixed_precision() and bool(lib().asr_rnn_sweep_wide_supported(rnn_type_id(rnn_type), B, T, H, ndir))


def rnn_sweep_wide_ws(B, H, ndir=2, device="cuda"):
    return torch.zeros(int(lib().asr_rnn_sweep_wide_ws_floats(B, H, ndir)), device=device, dtype=torch.float32)


def rnn_sweep_wide_fwd(seq, ws, err_flag=None):
    check(lib().asr_rnn_sweep_wide_fwd(C.byref(seq), _p(ws), _p(err_flag), _stream()))


def rnn_sweep_wide_bwd_supported(rnn_type, B, T, H, ndir=2) -> bool:
    """True when the wide one-launch BPTT sweep (rnn_sweep_wide_bwd.hip: resident bf16 blocks of U, bf16 partial sums; mixed precision) takes
    this layer."""
    return mixed_precision() and bool(lib().asr_rnn_sweep_wide_bwd_supported(rnn_type_id(rnn_type), B, T, H, ndir))


def rnn_sweep_wide_bwd_ws(B, H, ndir=2, device="cuda"):
    return torch.zeros(int(lib().asr_rnn_sweep_wide_bwd_ws_floats(B, H, ndir)), device=device, dtype=torch.float32)


def _rnn_seq_grad(dy, dirs_grad):
    g = _lib.RnnSeqGrad()
    g.dy = dy.data_ptr()
    g.dy_ld = dy.stride(1)
    g.dh_last = _arr2([d.get("dh_last") for d in dirs_grad])
    g.dh_last_ld = _arr2([d["dh_last"].stride(0) if d.get("dh_last") is not None else 0 for d in dirs_grad], C.c_long)
    g.dc = _arr2([d.get("dc") for d in dirs_grad])
    g.dy_carry = _arr2([d.get("dy_carry") for d in dirs_grad])
    g.direct = _arr2([d.get("direct") for d in dirs_grad])
    g.dh0 = _arr2([d.get("dh0") for d in dirs_grad])
    g.dh0_ld = _arr2([d["dh0"].stride(0) if d.get("dh0") is not None else 0 for d in dirs_grad], C.c_long)
    g.ds = _arr2([d.get("ds") for d in dirs_grad])
    g.db = _arr2([d.get("db") for d in dirs_grad])
    g.db_rec = _arr2([d.get("db_rec") for d in dirs_grad])
    g.ds16 = _arr2([d.get("ds16") for d in dirs_grad])
    g.ds16T = _arr2([d.get("ds16T") for d in dirs_grad])
    lds = [d["ds16T"].stride(0) for d in dirs_grad if d.get("ds16T") is not None]
    assert len(set(lds)) <= 1
    g.ds16T_ld = lds[0] if lds else 0
    return g


def rnn_sweep_wide_bwd(seq, dy, dirs_grad, ws, err_flag=None):
    """The wide layer's BPTT in one launch; dirs_grad as for rnn_seq_bwd, with `ds` [B,T,4H] tensors of their own (out) and / or the bf16
    images the layer's products read: `ds16` bf16 [B*T, 4H], `ds16T` bf16 [4H, K8] (columns t * B + b, zero-initialised padding), and
    optionally `db` [4H] (+= the bias gradient)."""
    g = _rnn_seq_grad(dy, dirs_grad)
    check(lib().asr_rnn_sweep_wide_bwd(C.byref(seq), C.byref(g), _p(ws), _p(err_flag), _stream()))


def rnn_persist_error(ws) -> int:
    """Non-zero if a hand-off of the last one-launch sweep timed out (synchronises): (who gave up | step << 8)."""
    return int(ws[-32:].view(torch.int32)[0].item())


def sweep_diag_words(ws, decoder=False):
    """The 32 diagnosis words of a sweep workspace (csrc/sweep_common.h) as a float32 view: the last 32 floats of an encoder /
    wide sweep workspace, 288 floats from the end of a decoder sweep workspace (256 per-workgroup records follow them)."""
    n = ws.numel()
    return ws[n - 288:n - 256] if decoder else ws[n - 32:]


_ABORT_STAGE = {
    "rnn_sweep_fwd": {1: "gather of h(t-1)", 3: "XCD-id exchange", 15: "start handshake (the grid did not assemble)"},
    "rnn_sweep_bwd": {1: "gather of the partial dh blocks", 2: "owner waiting for the other gather waves", 3: "publish wave waiting for the gather waves",
                      4: "publish wave waiting for its contraction partners", 5: "XCD-id exchange", 15: "start handshake (the grid did not assemble)"},
    "rnn_sweep_wide": {1: "probe of h(t-1)", 2: "gather of h(t-1)", 15: "start handshake (the grid did not assemble)"},
    "decoder_sweep_fwd": {1: "gather of h1", 2: "gather of the chunk partials", 3: "gather of the context", 4: "gather of h0",
                          15: "start handshake (the grid did not assemble)"},
    "decoder_sweep_bwd": {1: "layer-1 gather", 2: "layer-0 gather", 3: "context-gradient gather", 15: "start handshake (the grid did not assemble)"},
}


def sweep_diagnosis(ws, kind, decoder=False, clear=False):
    """None when the last launch of the sweep that owns `ws` ended normally, else a dict describing its first time-out
    (synchronises): the error word decoded (stage, step), the record of the first workgroup that gave up and the verdict
    'absent workgroup' (arrivals < expected: a compute unit was held by another tenant for the whole spin limit) or 'lost
    hand-off' (every workgroup was resident)."""
    view = sweep_diag_words(ws, decoder).view(torch.int32)
    w = view.cpu().tolist()
    if w[0] == 0 and w[24] == 0:
        return None
    # the sticky record (words 16-24) survives later launches; the per-launch words describe the LAST launch only
    rec, expected, sticky = (w[16:24], w[23], True) if w[24] else (w[8:16], w[5], False)
    word = rec[0] if (sticky or w[15]) else w[0]
    code, step = word & 255, ((word >> 8) & 0xFF) if decoder else ((word >> 8) & 0xFFFFFF)
    out = dict(kernel=kind, error_word=word & 0xFFFFFFFF, stage=_ABORT_STAGE.get(kind, {}).get(code, f"LDS hand-over (code {code})"), step=step,
               launches_that_gave_up=w[24], expected=expected)
    if sticky or w[15]:
        out.update(block=(rec[1] & 0xFFF, (rec[1] >> 12) & 0xFFF, (rec[1] >> 24) & 0xFF), xcc_id=rec[2], arrived=rec[3], local_mode=rec[4], wave=rec[5],
                   verdict="absent workgroup" if 0 < expected and rec[3] < expected else "lost hand-off")
    if clear:
        view[16:25].zero_()
    return out


def sweep_gate(diag_words, max_us=300):
    """One wave on the current stream that waits (at most max_us) until the sweep owning `diag_words` has all its workgroups
    resident (asr_sweep_gate): put in front of work that is to run beside that sweep on another stream."""
    check(lib().asr_sweep_gate(_p(diag_words), int(max_us), _stream()))


def rnn_sweep_set_spin_limit(polls: int):
    """Polls before a hand-off of the one-launch sweeps gives up (tests force time-outs with 0)."""
    lib().asr_rnn_sweep_set_spin_limit(int(polls))


def rnn_seq_bwd(seq, dy, dirs_grad, persist_ws=None, err_flag=None):
    """dirs_grad: list of dicts with keys dh_last, dc, dy_carry, direct ([B,H] scratch), dh0, ds.
    persist_ws: scratch from rnn_persist_bwd_ws -> the one-launch sweep instead of one launch per step; the sweep writes the
    gate-sum gradients to the `ds` tensors ([B,T,NS*H], not the saved activations), the per-step kernels over `saved`."""
    g = _lib.RnnSeqGrad()
    g.dy = dy.data_ptr()
    g.dy_ld = dy.stride(1)
    g.dh_last = _arr2([d.get("dh_last") for d in dirs_grad])
    g.dh_last_ld = _arr2([d["dh_last"].stride(0) if d.get("dh_last") is not None else 0 for d in dirs_grad], C.c_long)
    g.dc = _arr2([d.get("dc") for d in dirs_grad])
    g.dy_carry = _arr2([d.get("dy_carry") for d in dirs_grad])
    g.direct = _arr2([d.get("direct") for d in dirs_grad])
    g.dh0 = _arr2([d.get("dh0") for d in dirs_grad])
    g.dh0_ld = _arr2([d["dh0"].stride(0) if d.get("dh0") is not None else 0 for d in dirs_grad], C.c_long)
    g.ds = _arr2([d.get("ds") for d in dirs_grad])
    g.db = _arr2([d.get("db") for d in dirs_grad])
    g.db_rec = _arr2([d.get("db_rec") for d in dirs_grad])
    if persist_ws is not None:
        check(lib().asr_rnn_sweep_bwd(C.byref(seq), C.byref(g), _p(persist_ws), _p(err_flag), _stream()))
    else:
        check(lib().asr_rnn_seq_bwd(C.byref(seq), C.byref(g), _stream()))


def rnn_persist_bwd_supported(rnn_type, B, T, H, ndir=2):
    return bool(lib().asr_rnn_sweep_bwd_supported(rnn_type_id(rnn_type), B, T, H, ndir))


def rnn_persist_bwd_ws(B, H, ndir=2, device="cuda"):
    return torch.zeros(int(lib().asr_rnn_sweep_bwd_ws_floats(B, H, ndir)), device=device, dtype=torch.float32)


def back_src(D, W, rnn_type, H, kind, drop=None):
    """asr_rnn_back_src for a consumer whose ds rows are D ([B, NS*H] view) and whose Keras kernel is W
    (rows = the units / input features receiving the gradient).  kind: 'rec' (recurrent kernel: the
    consumer read this cell's state) or 'input' (input kernel: it read this cell's output).
    drop: (rate, stream, ld, off) of the consumer's input dropout or None."""
    s = _lib.RnnBackSrc()
    s.D, s.ldd, s.W, s.ldw = D.data_ptr(), D.stride(0), W.data_ptr(), W.stride(0)
    s.W16 = bf16_twin(W)
    if rnn_type == "gru":
        if kind == "rec":
            s.nseg = 2
            s.d_col0[0], s.w_col0[0], s.len[0] = 0, 0, 2 * H
            s.d_col0[1], s.w_col0[1], s.len[1] = 3 * H, 2 * H, H
        else:
            s.nseg = 1
            s.d_col0[0], s.w_col0[0], s.len[0] = 0, 0, 3 * H
    else:
        s.nseg = 1
        s.d_col0[0], s.w_col0[0], s.len[0] = 0, 0, (4 if rnn_type == "lstm" else 1) * H
    if drop is not None and drop[0] > 0:
        s.drop_rate, s.drop_stream, s.drop_ld, s.drop_off = drop
    return s


# ----------------------------------------------------------------------------------------- decoder sweep
def decoder_sweep_supported(rnn_type, num_layers, B, U, T2, Hd, D) -> bool:
    if rnn_type != "lstm":
        return False
    return bool(lib().asr_decoder_sweep_supported(rnn_type_id(rnn_type), num_layers, B, U, T2, Hd, D))


def decoder_sweep_ws(Hd, D, device="cuda"):
    return torch.zeros(int(lib().asr_decoder_sweep_ws_floats(Hd, D)), device=device, dtype=torch.float32)


def decoder_sweep_error(ws) -> int:
    """Non-zero if a hand-off of the last decoder sweep timed out (synchronises): (stage code | step << 8)."""
    return int(ws[-288:-287].view(torch.int32)[0].item())


def decoder_sweep_fwd(desc: "_lib.DecoderSweep", ws, err_flag=None):
    """All decoder steps of a teacher-forced LAS forward pass in one launch (asr_decoder_sweep_fwd)."""
    check(lib().asr_decoder_sweep_fwd(C.byref(desc), _p(ws), _p(err_flag), _stream()))


def decoder_sweep_bwd_supported(rnn_type, num_layers, B, U, T2, Hd, D) -> bool:
    if rnn_type != "lstm":
        return False
    return bool(lib().asr_decoder_sweep_bwd_supported(rnn_type_id(rnn_type), num_layers, B, U, T2, Hd, D))


def decoder_sweep_bwd_ws(Hd, D, device="cuda"):
    return torch.zeros(int(lib().asr_decoder_sweep_bwd_ws_floats(Hd, D)), device=device, dtype=torch.float32)


def decoder_sweep_bwd(desc: "_lib.DecoderSweepGrad", ws, err_flag=None):
    """All decoder steps of the backward pass of a teacher-forced LAS step in one launch (asr_decoder_sweep_bwd)."""
    check(lib().asr_decoder_sweep_bwd(C.byref(desc), _p(ws), _p(err_flag), _stream()))


# ----------------------------------------------------------------------------------------- cells (decoder steps)
def rnn_cell_fwd(rnn_type, B, H, steps, seed=None):
    """steps: list (1 or 2 directions) of _lib.RnnStepFwd."""
    arr = (_lib.RnnStepFwd * len(steps))(*steps)
    check(lib().asr_rnn_cell_fwd(rnn_type_id(rnn_type), B, H, len(steps), arr, _p(seed), _stream()))


def rnn_cell_bwd(rnn_type, B, steps, seed=None):
    arr = (_lib.RnnStepBwd * len(steps))(*steps)
    check(lib().asr_rnn_cell_bwd(rnn_type_id(rnn_type), B, len(steps), arr, _p(seed), _stream()))


def token_mask(tok, pad, out):
    """out[i] = tok[i] != pad over contiguous int32 tokens; out uint8 with unit stride."""
    check(lib().asr_token_mask(_p(tok), tok.numel(), int(pad), _p(out), 1, _stream()))
    return out


# ----------------------------------------------------------------------------------------- conv
def conv_desc(x_shape, w_shape, strides):
    B, H, W, Cc = x_shape
    kh, kw, Ci, O = w_shape
    if Ci != Cc:
        raise ValueError(f"conv2d: input channels {Cc} != kernel channels {Ci}")
    sh, sw = (strides, strides) if isinstance(strides, int) else strides
    return _lib.ConvDesc(B, H, W, Cc, kh, kw, sh, sw, O)


def conv_out_dims(d):
    ho, wo = C.c_int(), C.c_int()
    check(lib().asr_conv2d_out_dims(C.byref(d), C.byref(ho), C.byref(wo)))
    return ho.value, wo.value


# Scratch of the row-staged convolutions (csrc/conv_halo.hip: the kernel re-ordered and split into bf16 planes, rewritten by every call), one buffer
# per (geometry, pass, stream): calls on one stream are ordered, calls on different streams must not share it.
_conv_halo_scratch = {}


def _conv_halo_ws(d, which):
    """(buffer, bytes) for asr_conv2d_*_halo, or None when the geometry / product mode takes the general kernels."""
    need = lib().asr_conv2d_halo_workspace(C.byref(d), which)
    if need <= 0 or _f32_mode["compute"] == 0:
        return None
    key = (d.B, d.H, d.W, d.C, d.kh, d.kw, d.sh, d.sw, d.O, which, torch.cuda.current_stream().cuda_stream)
    buf = _conv_halo_scratch.get(key)
    if buf is None or buf.numel() < need:
        buf = torch.empty(need, dtype=torch.uint8, device="cuda")
        _conv_halo_scratch[key] = buf
    return buf, need


def conv2d_fwd(x, w, bias, strides, y=None, seed=None, drop_stream=0, drop_rate=0.0):
    _dev(x, name="x"), _dev(w, name="w")
    d = conv_desc(x.shape, w.shape, strides)
    Ho, Wo = conv_out_dims(d)
    if y is None:
        y = torch.empty(d.B, Ho, Wo, d.O, device=x.device, dtype=torch.float32)
    assert x.is_contiguous() and w.is_contiguous() and y.is_contiguous()
    ws = _conv_halo_ws(d, 0) if not drop_rate > 0 and x.data_ptr() % 16 == 0 else None
    if ws is not None:
        check(lib().asr_conv2d_fwd_halo(C.byref(d), _p(x), _p(w), _p(bias), _p(y), C.c_void_p(ws[0].data_ptr()), ws[1], _stream()))
        return y
    check(lib().asr_conv2d_fwd(C.byref(d), _p(x), _p(w), _p(bias), _p(y), _p(seed) if drop_rate > 0 else None, drop_stream,
                               float(drop_rate), _stream()))
    return y


def conv2d_bwd_filter(x, dy, dw, strides):
    d = conv_desc(x.shape, dw.shape, strides)
    assert x.is_contiguous() and dy.is_contiguous() and dw.is_contiguous()
    check(lib().asr_conv2d_bwd_filter(C.byref(d), _p(x), _p(dy), _p(dw), _stream()))
    return dw


def conv2d_bwd_data(dy, w, dx, strides):
    d = conv_desc(dx.shape, w.shape, strides)
    assert dx.is_contiguous() and dy.is_contiguous() and w.is_contiguous()
    ws = _conv_halo_ws(d, 1) if dy.data_ptr() % 16 == 0 else None
    if ws is not None:
        check(lib().asr_conv2d_bwd_data_halo(C.byref(d), _p(dy), _p(w), _p(dx), C.c_void_p(ws[0].data_ptr()), ws[1], _stream()))
        return dx
    check(lib().asr_conv2d_bwd_data(C.byref(d), _p(dy), _p(w), _p(dx), _stream()))
    return dx


# ----------------------------------------------------------------------------------------- memory-bound layers
def fill(t, value=0.0):
    assert t.is_contiguous()
    _dev(t, name="t")
    check(lib().asr_fill_f32(_p(t), t.numel(), float(value), _stream()))
    return t


def frame_mask(x, group, Tout, out=None):
    """x [B,T,...] -> uint8 [B,Tout]; out[b,j] = any(x[b, j*group:(j+1)*group] != 0)."""
    _dev(x, name="x")
    B, T = x.shape[:2]
    FC = x.numel() // (B * T)
    if out is None:
        out = torch.empty(B, Tout, device=x.device, dtype=torch.uint8)
    assert x.is_contiguous()
    check(lib().asr_frame_mask(_p(x), B, T, FC, group, Tout, _p(out), _stream()))
    return out


def colsum(a, out):
    """out[c] += sum_r a[r, c] (a 2-D, unit inner stride)."""
    check(lib().asr_colsum(_p(a), a.shape[0], a.shape[1], a.stride(0), _p(out), _stream()))
    return out


def colsum_weighted(a, w, out):
    """out[c] += sum_r w[r] a[r, c] (a 2-D with unit inner stride, w [rows] contiguous)."""
    check(lib().asr_colsum_weighted(_p(a), a.shape[0], a.shape[1], a.stride(0), _p(w), _p(out), _stream()))
    return out


def rowdot(a, x, y):
    """y[r] = a[r, :] . x."""
    check(lib().asr_rowdot(_p(a), a.shape[0], a.shape[1], a.stride(0), _p(x), _p(y), _stream()))
    return y


def rank1_add(c, u, v):
    """c[r, :] += u[r] * v."""
    check(lib().asr_rank1_add(_p(c), c.shape[0], c.shape[1], c.stride(0), _p(u), _p(v), _stream()))
    return c


def bn_fwd(x, gamma, beta, y, mean, rstd, moving_mean, moving_var, ws, *, relu, training, eps=1e-3, momentum=0.99):
    M, Cc = x.shape
    check(lib().asr_bn_fwd(_p(x), M, Cc, x.stride(0), _p(gamma), _p(beta), eps, momentum, int(relu), int(training), _p(y),
                           y.stride(0), _p(mean), _p(rstd), _p(moving_mean), _p(moving_var), _p(ws), _stream()))
    return y


def bn_bwd(x, y, dy, mean, rstd, gamma, dx, dgamma, dbeta, ws, *, relu):
    M, Cc = x.shape
    check(lib().asr_bn_bwd(_p(x), _p(y), _p(dy), M, Cc, x.stride(0), y.stride(0) if y is not None else 0, dy.stride(0), _p(mean),
                           _p(rstd), _p(gamma), int(relu), _p(dx), dx.stride(0), _p(dgamma), _p(dbeta), _p(ws), _stream()))
    return dx


def rowdrop(stream0=0, stream_step=0, period=1, idx_ld=0, idx_off=0, rate=0.0):
    return _lib.RowDrop(stream0, stream_step, period, idx_ld, idx_off, rate)


def dropout_rows(x, y, seed, rd):
    R, K = x.shape
    check(lib().asr_dropout_rows(_p(x), x.stride(0), _p(y), y.stride(0), R, K, _p(seed), rd.stream0, rd.stream_step, rd.period,
                                 rd.idx_ld, rd.idx_off, rd.rate, _stream()))
    return y


def dropout_flat(x, seed, stream_id, rate):
    assert x.is_contiguous()
    check(lib().asr_dropout_flat(_p(x), x.numel(), _p(seed), stream_id, float(rate), _stream()))
    return x


def dropout_table(out, seed, stream_id, rate):
    assert out.is_contiguous()
    check(lib().asr_dropout_table(_p(out), out.numel(), _p(seed), stream_id, float(rate), _stream()))
    return out


def dropout_tables(tables, seed):
    """Several dropout_table()s in one launch: tables = [(out, stream_id, rate), ...]."""
    n = len(tables)
    if n == 0:
        return
    outs = (C.c_void_p * n)(*[t[0].data_ptr() for t in tables])
    ns = (C.c_long * n)(*[t[0].numel() for t in tables])
    ids = (C.c_uint32 * n)(*[int(t[1]) for t in tables])
    rates = (C.c_float * n)(*[float(t[2]) for t in tables])
    for t in tables:
        assert t[0].is_contiguous()
    check(lib().asr_dropout_tables(n, outs, ns, ids, rates, _p(seed), _stream()))


def embedding_fwd(E, tok, out, seed=None, drop1=None, drop2=None):
    R = tok.numel()
    V, Hd = E.shape
    check(lib().asr_embedding(0, _p(E), _p(tok), R, Hd, V, _p(out), out.stride(-2), _p(seed),
                              C.byref(drop1) if drop1 else None, C.byref(drop2) if drop2 else None, _stream()))
    return out


def embedding_bwd(dE, tok, dx, seed=None, drop1=None, drop2=None):
    R = tok.numel()
    V, Hd = dE.shape
    check(lib().asr_embedding(1, _p(dE), _p(tok), R, Hd, V, _p(dx), dx.stride(-2), _p(seed),
                              C.byref(drop1) if drop1 else None, C.byref(drop2) if drop2 else None, _stream()))
    return dE


def argmax_rows(x, out):
    check(lib().asr_argmax_rows(_p(x), x.stride(0), x.shape[0], x.shape[1], _p(out), _stream()))
    return out


# ----------------------------------------------------------------------------------------- attention / loss / optimizer
def attn_step_fwd(h, Kq, s0, mask, enc, e, p, ctx, images=None):
    """images: (Kq16, enc16) bfloat16 copies of Kq / enc (mixed precision) or None."""
    B, T, Hd = Kq.shape
    D = enc.shape[2]
    if images is not None:
        check(lib().asr_attn_step_fwd_bf16(_p(h), h.stride(0), C.c_void_p(images[0].data_ptr()), _p(s0), _p(mask),
                                           C.c_void_p(images[1].data_ptr()), B, T, Hd, D, _p(e), _p(p), _p(ctx), ctx.stride(0), _stream()))
        return
    check(lib().asr_attn_step_fwd(_p(h), h.stride(0), _p(Kq), _p(s0), _p(mask), _p(enc), B, T, Hd, D, _p(e), _p(p), _p(ctx),
                                  ctx.stride(0), _stream()))


def attn_step_bwd(dctx, p, Kq, enc, dp, ds, dh, accumulate, images=None):
    B, T, Hd = Kq.shape
    D = enc.shape[2]
    if images is not None:
        check(lib().asr_attn_step_bwd_bf16(_p(dctx), dctx.stride(0), _p(p), C.c_void_p(images[0].data_ptr()), C.c_void_p(images[1].data_ptr()),
                                           B, T, Hd, D, _p(dp), _p(ds), _p(dh), dh.stride(0), int(accumulate), _stream()))
        return
    check(lib().asr_attn_step_bwd(_p(dctx), dctx.stride(0), _p(p), _p(Kq), _p(enc), B, T, Hd, D, _p(dp), _p(ds), _p(dh),
                                  dh.stride(0), int(accumulate), _stream()))


def attn_fused_supported(T, Hd, D):
    return bool(lib().asr_attn_fused_supported(T, Hd, D))


def attn_fused_ws(B, Hd, D, device="cuda"):
    """(scratch f32, tickets i32 zeros) for attn_fused_fwd / attn_fused_bwd."""
    return (torch.empty(int(lib().asr_attn_fused_ws_floats(B, Hd, D)), device=device, dtype=torch.float32),
            torch.zeros(B, device=device, dtype=torch.int32))


def attn_fused_fwd(h, Kq, s0, mask, enc, fws, p, ctx):
    """attn_step_fwd in one launch; fws from attn_fused_ws."""
    B, T, Hd = Kq.shape
    D = enc.shape[2]
    check(lib().asr_attn_fused_fwd(_p(h), h.stride(0), _p(Kq), _p(s0), _p(mask), _p(enc), B, T, Hd, D, _p(fws[0]), _p(fws[1]), _p(p),
                                   _p(ctx), ctx.stride(0), _stream()))


def attn_fused_bwd(dctx, p, Kq, enc, fws, ds, dh, accumulate):
    """attn_step_bwd in one launch; fws from attn_fused_ws."""
    B, T, Hd = Kq.shape
    D = enc.shape[2]
    check(lib().asr_attn_fused_bwd(_p(dctx), dctx.stride(0), _p(p), _p(Kq), _p(enc), B, T, Hd, D, _p(fws[0]), _p(fws[1]), _p(ds),
                                   _p(dh), dh.stride(0), int(accumulate), _stream()))


def softmax_xent(logits, labels, stats, ignore_index=0, write_grad=True, grad_scale=1.0):
    """logits [R, V] overwritten with the gradient; stats (3 floats) must be pre-zeroed."""
    R, V = logits.shape
    _dev(labels, torch.int32, "labels")
    check(lib().asr_softmax_xent(_p(logits), logits.stride(0), _p(labels), R, V, ignore_index, _p(stats), int(write_grad),
                                 float(grad_scale), _stream()))


def lr_schedule(total_steps, max_learning_rate, min_learning_rate, warmup_rate=0.0, warmup_steps=0, offset_steps=0):
    s = _lib.LrSchedule()
    check(lib().asr_lr_schedule_init(C.byref(s), int(total_steps), float(max_learning_rate), float(min_learning_rate),
                                     float(warmup_rate), int(warmup_steps or 0), int(offset_steps or 0)))
    return s


def adam_step(params, grads, m, v, state, sched, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0, skip_flag=None):
    """skip_flag: optional device float; non-zero = the step's gradients are invalid, nothing is updated."""
    check(lib().asr_adam_step(_p(params), _p(grads), _p(m), _p(v), params.numel(), _p(state), C.byref(sched), beta1, beta2,
                              eps, grad_scale, _p(skip_flag), _stream()))


def advance_state(state, skip_flag=None):
    check(lib().asr_advance_state(_p(state), _p(skip_flag), _stream()))


def ctc_workspace_floats(B, T, L):
    return int(lib().asr_ctc_workspace_floats(B, T, L))


def ctc_loss(logits2d, labels, B, T, blank, pad, ws, per_sample, stats, write_grad=True, grad_scale=1.0):
    """logits2d [B*T, V] (overwritten with the gradient when write_grad), labels i32 [B, L]."""
    _dev(labels, torch.int32, "labels")
    V, L = logits2d.shape[1], labels.shape[1]
    assert labels.is_contiguous() and logits2d.shape[0] == B * T
    check(lib().asr_ctc_loss(_p(logits2d), logits2d.stride(0), _p(labels), B, T, V, L, int(blank), int(pad), _p(ws), _p(per_sample),
                             _p(stats), int(write_grad), float(grad_scale), _stream()))


def mask_rows(x2d, mask, out2d):
    """out[r,:] = x[r,:] * mask[r]; mask uint8 [R] contiguous."""
    check(lib().asr_mask_rows(_p(x2d), x2d.stride(0), _p(mask), x2d.shape[0], x2d.shape[1], _p(out2d), out2d.stride(0), _stream()))
    return out2d
