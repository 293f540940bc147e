"""Data-parallel TrainStep on the GPU: two ranks (two processes sharing the one MI355X, gloo moving the
CUDA buckets) run three captured-graph training steps on different batches; their parameters must stay
identical to each other and equal to a single-process emulation of the same thing (each replica's
gradient computed with loss scale 1/2, summed, one Adam step) - i.e. the side-stream bucket exchange,
its events and the hipGraph replays compose into exactly the replica-mean update of
tf.distribute.MirroredStrategy (utils.py:148-149).  RCCL itself is exercised by bench.py --gpus N."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

CFG = dict(rnn_type="lstm", vocab_size=61, encoder_hidden_dim=16, decoder_hidden_dim=16, num_encoder_layers=2, num_decoder_layers=2,
           dropout=0.15, teacher_forcing_rate=0.99, pad_id=0)
STEPS, B, T, L = 3, 3, 46, 6


def _batch(rank, step):
    g = torch.Generator().manual_seed(100 * rank + step)
    feats = torch.randn(B, T, 20, 3, generator=g)
    feats[1, 30:] = 0.0
    toks = torch.randint(1, 61, (B, L), generator=g, dtype=torch.int32)
    toks[2, 4:] = 0
    return feats, torch.full((B,), T, dtype=torch.int32), toks


def _model():
    from speech_recognition_amd.models import LAS
    return LAS(**CFG, seed=11).build(20, 3)


def _worker(rank, world, port, q):
    # Two processes time-share this one GPU, so the persistent recurrent kernels (which need all their
    # workgroups resident together, one process per GPU in production) are switched off here; the
    # per-step kernels compute the same values.
    os.environ["ASR_PERSISTENT_RNN"] = "0"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import speech_recognition_amd  # noqa: F401
        from speech_recognition_amd.training import TrainStep
        from speech_recognition_amd.utils import DeviceStrategy, LRScheduler
        torch.cuda.set_device(0)
        model = _model()
        trainer = TrainStep(model, LRScheduler(100, 1e-2, 1e-4), frontend=None, strategy=DeviceStrategy(torch.device("cuda", 0), world, rank),
                            use_graph=True)
        losses, grad0 = [], None
        for s in range(STEPS):
            f, n, t = _batch(rank, s)
            ws = trainer.step(f.cuda(), n.cuda(), t.cuda(), use_teacher_forcing=True)
            losses.append(trainer.read_stats(ws)[0])
            if s == 0:
                grad0 = model.store.grad.cpu().numpy().copy()     # the all-reduced gradient the update used
        q.put((rank, {k: v.numpy().copy() for k, v in model.state_dict().items()}, losses, grad0))   # by value (no fd passing)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_training_equals_manual_replica_mean():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 1500)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=240) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(60)
    (_, p0, l0, g0), (_, p1, l1, g1) = res
    p0, p1 = ({k: torch.from_numpy(v) for k, v in p.items()} for p in (p0, p1))
    assert np.array_equal(g0, g1)                              # both ranks hold the same reduced gradient
    for k in p0:      # every trainable variable is bit-identical across ranks; BN moving statistics are per replica (Q9)
        if not k.endswith(("moving_mean", "moving_variance")):
            assert torch.equal(p0[k], p1[k]), k

    # single-process emulation: two replicas with their own BatchNorm statistics and shared weights
    from speech_recognition_amd import ops
    from speech_recognition_amd.utils import LRScheduler
    reps = [_model(), _model()]
    sched = LRScheduler(100, 1e-2, 1e-4).device_schedule()
    ref_losses = [[], []]
    for s in range(STEPS):
        total = torch.zeros_like(reps[0].store.grad)
        for r, m in enumerate(reps):
            f, n, t = _batch(r, s)
            f, t = f.cuda(), t.cuda()
            ws, labels = m.train_workspace(B, T, L)
            m.set_targets(ws, t, labels)
            ops.fill(m.store.grad, 0.0)
            m.pack_weights()
            m.forward_ws(ws, f, True, True)
            m.loss_and_grad(ws, labels, 0.5)
            m.backward_ws(ws, f)
            torch.cuda.synchronize()
            ref_losses[r].append(float(ws.stats[0]))
            total += m.store.grad
        if s == 0:      # same parameters on both sides: the exchanged gradient is the sum of the two scaled replica gradients
            ref_g = total.cpu().numpy()
            assert np.abs(g0 - ref_g).max() <= 1e-5 * np.abs(ref_g).max(), np.abs(g0 - ref_g).max()
        before_last = reps[0].store.flat.clone()
        for m in reps:
            m.store.grad.copy_(total)
            ops.adam_step(m.store.flat, m.store.grad, m.store.adam_m, m.store.adam_v, m.state, sched, 0.9, 0.999, 1e-7)
            ops.advance_state(m.state)
            m.weights_changed()
        torch.cuda.synchronize()
    np.testing.assert_allclose(l0, ref_losses[0], rtol=1e-3)
    np.testing.assert_allclose(l1, ref_losses[1], rtol=1e-3)
    # Later steps: atomically accumulated gradients differ in their last bits between runs and Adam's first steps
    # amplify that for near-zero gradients (update = lr * g / (|g| + eps) flips with the sign of g), so the
    # parameters are compared statistically: all but a sliver of the elements agree to 1e-4.
    for r, got in enumerate((p0, p1)):
        ref = reps[r].state_dict()
        bad = sum(int(((got[k] - ref[k]).abs() > 1e-4).sum()) for k in ref)
        count = sum(v.numel() for v in ref.values())
        worst = sorted(((float((got[k] - ref[k]).abs().max()), k) for k in ref), reverse=True)[:4]
        k0 = "listener/conv1/kernel"
        off = reps[0].store.offsets[k0]
        prev = before_last[off:off + got[k0].numel()].view(got[k0].shape).cpu()
        assert bad < 0.01 * count, (r, bad, count, worst, "vs params before the last update:", float((got[k0] - prev).abs().max()))
