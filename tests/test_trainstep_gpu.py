"""TrainStep housekeeping on the GPU: per-shape buffers and graphs are bounded (least recently used shapes are
released), and a shape that comes back after eviction trains on as before."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_shape_cache_is_bounded_and_eviction_is_transparent():
    from speech_recognition_amd.models import LAS
    from speech_recognition_amd.training import TrainStep
    from speech_recognition_amd.utils import LRScheduler

    def run(max_shapes):
        model = LAS("lstm", 41, 8, 8, 1, 1, 0.1, 0.99, seed=3).build(20, 3)
        tr = TrainStep(model, LRScheduler(100, 1e-3, 1e-5), frontend=None, use_graph=True)
        tr.max_shapes = max_shapes
        g = torch.Generator().manual_seed(0)
        shapes = [(2, 30, 5), (2, 34, 5), (3, 30, 6), (2, 38, 4)]
        batches = [(torch.randn(B, T, 20, 3, generator=g).cuda(), torch.full((B,), T, dtype=torch.int32).cuda(),
                    torch.randint(1, 41, (B, L), generator=g, dtype=torch.int32).cuda()) for B, T, L in shapes]
        losses = []
        for i in range(16):                              # every shape is seen four times: eager, capture, replay, replay
            ws = tr.step(*batches[i % 4], use_teacher_forcing=True)
            losses.append(tr.read_stats(ws)[0])
            assert len(tr._shapes) <= max_shapes and len(model._ws) <= max_shapes
        return losses

    bounded, unbounded = run(2), run(16)
    assert all(np.isfinite(bounded))
    np.testing.assert_allclose(bounded, unbounded, rtol=1e-4)    # same training trajectory with and without eviction


def test_mixed_precision_tracks_f32():
    """--mixed-precision (bf16 operands in the dense contractions, f32 accumulation and weights): logits and loss
    gradients of LAS-mini stay within bf16 rounding of the f32 run on the same batch - and are not identical
    (the switch does reach the kernels)."""
    import torch

    from speech_recognition_amd import ops
    from tests import test_las_gpu as TL
    cfg = TL.mk_cfg("lstm")
    audio, tokens, _ = TL.inputs(B=4, T=38)
    outs = {}
    for mode in (False, True):
        ops.set_mixed_precision(mode)
        try:
            m, _ = TL.build(cfg)
            outs[mode] = m.forward(audio.cuda(), tokens.cuda().to(torch.int32)[:, :-1].contiguous(), training=False,
                                   use_teacher_forcing=True).clone()
        finally:
            ops.set_mixed_precision(False)
    a, b = outs[False], outs[True]
    scale = float(a.abs().max())
    diff = float((a - b).abs().max())
    assert 0.0 < diff < 3e-2 * scale, (diff, scale)
