"""Helpers shared by the parity tests."""
import numpy as np
import torch


def rel_err(got, ref):
    """max |got-ref| / max(|ref|, tiny): scale-aware max error."""
    got = got.detach().double().cpu() if isinstance(got, torch.Tensor) else torch.as_tensor(np.asarray(got)).double()
    ref = ref.detach().double().cpu() if isinstance(ref, torch.Tensor) else torch.as_tensor(np.asarray(ref)).double()
    assert got.shape == ref.shape, f"shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    if ref.numel() == 0:
        return 0.0
    denom = max(float(ref.abs().max()), 1e-30)
    return float((got - ref).abs().max()) / denom


def assert_close(got, ref, tol, what=""):
    e = rel_err(got, ref)
    assert e <= tol, f"{what}: max-normalised error {e:.3e} > {tol:.1e}"
    return e


def gpu(x, dtype=torch.float32):
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(x)
    return x.detach().to(dtype).cuda().contiguous()
