"""Data-parallel TrainStep on the GPU: two ranks (two processes sharing the one MI355X, gloo moving the CUDA buckets) run
captured-graph training steps on different batches with the one-launch ENCODER sweeps on (the whole-chip decoder sweeps are off: see
_worker - two processes cannot both own every compute unit of one GPU).

What is asserted, at EVERY step (exactly where exactness is defined, tightly where only the order of atomic f32 sums differs):
  * replicas built WITHOUT a seed start from rank 0's weights (TrainStep broadcasts them) and stay bit-identical;
  * both ranks hold the same all-reduced gradient, bit for bit;
  * that gradient equals the sum of the two replica gradients (each computed with loss scale 1/2) of a single-process
    emulation that starts the step from the ranks' own parameters: 1e-5 of its largest entry (atomic accumulation order);
  * the parameters after the step are, bit for bit, Adam applied to (parameters before, that gradient) - the side-stream
    bucket exchange (five buckets in reverse-backward order), its events and the hipGraph replays compose into exactly the
    replica-mean update of tf.distribute.MirroredStrategy (utils.py:148-149).
RCCL itself needs more than one GPU and is exercised by bench.py --gpus N."""
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


def _model(seed=None):
    from speech_recognition_amd.models import LAS
    return LAS(**CFG, seed=seed).build(20, 3)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    # Two ranks SHARE the one GPU here (a test arrangement: real data parallelism is one process per GPU).  The decoder sweeps are
    # whole-chip launches - 256 workgroups, one per compute unit, that wait for each other - and two of them from two processes can
    # each hold part of the chip and starve the other until the start handshake gives up ("absent workgroup", sweep_common.h; seen
    # once in ~10 runs of this test).  The per-step decoder kernels run here; the encoder sweeps (<= 3/4 of the chip, two fit side by
    # side) stay on.  The decoder sweeps under co-tenancy are covered in-process: test_trainstep_gpu.py (foreign kernel holding CUs).
    os.environ["ASR_DECODER_SWEEP"] = "0"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import speech_recognition_amd  # noqa: F401
        from speech_recognition_amd import layers
        from speech_recognition_amd.training import TrainStep
        from speech_recognition_amd.utils import DeviceStrategy, LRScheduler
        assert layers.PERSISTENT_RNN
        torch.cuda.set_device(0)
        model = _model()                                          # no seed: every rank draws its own initial weights ...
        own = model.store.flat.cpu().numpy().copy()
        trainer = TrainStep(model, LRScheduler(100, 1e-2, 1e-4), frontend=None, strategy=DeviceStrategy(torch.device("cuda", 0), world, rank),
                            use_graph=True)                       # ... and TrainStep replaces them with rank 0's
        assert len(model.store.bucket_ranges) == 2 + CFG["num_encoder_layers"]
        params, grads, losses, persistent = [model.store.flat.cpu().numpy().copy()], [], [], []
        for s in range(STEPS):
            f, n, t = _batch(rank, s)
            ws = trainer.step(f.cuda(), n.cuda(), t.cuda(), use_teacher_forcing=True)
            losses.append(trainer.read_stats(ws)[0])              # also raises if a sweep hand-off timed out
            grads.append(model.store.grad.cpu().numpy().copy())   # the all-reduced gradient the update used
            params.append(model.store.flat.cpu().numpy().copy())
            persistent.append(all("persist_ws" in lw["rnn"] and "persist_bwd_ws" in lw["rnn"] for lw in ws.layers))
        q.put((rank, own, params, grads, losses, persistent, model.state.cpu().numpy().copy()))   # by value (no fd passing)
    except BaseException:                                        # surface the failure at once instead of letting the parent wait
        import traceback
        q.put(("error", rank, traceback.format_exc()))
        raise
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
    res = []
    for _ in procs:
        r = q.get(timeout=240)
        if r[0] == "error":
            for p in procs:
                p.join(5)
                if p.is_alive():
                    p.kill()                                     # the exact processes this test started
            pytest.fail(f"rank {r[1]} failed:\n{r[2]}")
        res.append(r)
    res.sort(key=lambda x: x[0])
    for p in procs:
        p.join(60)
    (_, own0, p0, g0, l0, pers0, st0), (_, own1, p1, g1, l1, pers1, st1) = res
    assert not np.array_equal(own0, own1), "the seedless replicas were meant to start different"
    assert np.array_equal(p0[0], own0) and np.array_equal(p1[0], own0), "every replica starts from rank 0's weights"
    assert all(pers0) and all(pers1), "the encoder layers ran the one-launch sweeps"
    assert np.array_equal(st0, st1) and int(st0[0]) == STEPS and int(st0[2]) == 0

    # single-process emulation: two replicas with their own BatchNorm statistics, started at every step from the ranks' parameters
    from speech_recognition_amd import ops
    from speech_recognition_amd.utils import LRScheduler
    reps = [_model(), _model()]
    ref = _model()                                               # carries the Adam moments of the reference update
    sched = LRScheduler(100, 1e-2, 1e-4).device_schedule()
    ref.store.flat.copy_(torch.from_numpy(p0[0]))
    for s in range(STEPS):
        assert np.array_equal(g0[s], g1[s]), f"step {s}: the ranks hold different reduced gradients"
        assert np.array_equal(p0[s + 1], p1[s + 1]), f"step {s}: the replicas diverged"
        total = torch.zeros_like(reps[0].store.grad)
        step_losses = []
        for r, m in enumerate(reps):
            m.store.flat.copy_(torch.from_numpy(p0[s]))
            m.weights_changed()
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
            step_losses.append(float(ws.stats[0]))
            total += m.store.grad
            ops.advance_state(m.state)                           # the dropout seed moves on as in the trainer
        ref_g = total.cpu().numpy()
        scale = np.abs(ref_g).max()
        assert np.abs(g0[s] - ref_g).max() <= 1e-5 * scale, (s, np.abs(g0[s] - ref_g).max(), scale)
        assert abs(l0[s] - step_losses[0]) <= 1e-5 * abs(step_losses[0]) and abs(l1[s] - step_losses[1]) <= 1e-5 * abs(step_losses[1])
        # the update itself is deterministic: Adam on the ranks' own reduced gradient reproduces their parameters exactly
        ref.store.grad.copy_(torch.from_numpy(g0[s]))
        ops.adam_step(ref.store.flat, ref.store.grad, ref.store.adam_m, ref.store.adam_v, ref.state, sched, 0.9, 0.999, 1e-7)
        ops.advance_state(ref.state)
        torch.cuda.synchronize()
        assert np.array_equal(ref.store.flat.cpu().numpy(), p0[s + 1]), f"step {s}: parameters are not Adam(previous, reduced gradient)"
