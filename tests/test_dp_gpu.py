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
    # once in ~10 runs of this test in round 3, which then switched them off by environment variable).  Round 4: TrainStep itself
    # detects ranks that share a device (all-gather of host + device uuid) and keeps the whole-chip loops on the per-step kernels
    # (ops.set_device_exclusive) - asserted below; the encoder sweeps (<= 3/4 of the chip, two fit side by side) stay on.  The
    # decoder sweeps under co-tenancy are covered in-process: test_trainstep_gpu.py (foreign kernel holding CUs).
    os.environ.pop("ASR_DECODER_SWEEP", None)
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
        from speech_recognition_amd import ops
        from speech_recognition_amd.models import las as las_mod
        assert las_mod.DECODER_SWEEP and las_mod.DECODER_SWEEP_BWD, "no environment override: the product decides"
        assert trainer.shared_device and not ops.device_exclusive(), "two ranks on one GPU must be detected"
        assert len(model.store.bucket_ranges) == 2 + CFG["num_encoder_layers"]
        params, grads, losses, persistent = [model.store.flat.cpu().numpy().copy()], [], [], []
        for s in range(STEPS):
            f, n, t = _batch(rank, s)
            ws = trainer.step(f.cuda(), n.cuda(), t.cuda(), use_teacher_forcing=True)
            losses.append(trainer.read_stats(ws)[0])              # also raises if a sweep hand-off timed out
            grads.append(model.store.grad.cpu().numpy().copy())   # the all-reduced gradient the update used
            params.append(model.store.flat.cpu().numpy().copy())
            persistent.append(all("persist_ws" in lw["rnn"] and "persist_bwd_ws" in lw["rnn"] for lw in ws.layers))
            assert not getattr(ws, "_sweep_ok", False) and not getattr(ws, "_sweep_bwd_ok", False), "whole-chip sweeps on a shared device"
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


def _dp_path_worker(port, q):
    """A process of its own: a single-rank RCCL group ("nccl", world size 1) must not leak into the other tests' process."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        import speech_recognition_amd  # noqa: F401
        from speech_recognition_amd import ops
        from speech_recognition_amd.training import TrainStep
        from speech_recognition_amd.utils import DeviceStrategy, LRScheduler
        out = {}
        for mixed in (False, True):
            ops.set_mixed_precision(mixed)
            runs = {}
            for name, force in (("single", False), ("dp", True), ("single_again", False), ("dp_native", True)):
                # dp_native: the collective through the C ABI (asr_allreduce_bucket -> RCCL), the whole data-parallel step ONE graph
                os.environ["ASR_NATIVE_COLLECTIVE"] = "1" if name == "dp_native" else "0"
                model = _model(seed=21)
                trainer = TrainStep(model, LRScheduler(100, 1e-2, 1e-4), frontend=None, strategy=DeviceStrategy(torch.device("cuda", 0), 1, 0),
                                    use_graph=True, force_dp_path=force)
                assert trainer.segmented == force and trainer.exchange.active == force
                assert (trainer.exchange.native is not None) == (name == "dp_native")
                if force:
                    assert trainer.exchange.wire_dtype == (torch.bfloat16 if mixed else torch.float32)
                losses = []
                for s in range(4):                                 # eager, capture, replay, replay
                    f, n, t = _batch(0, s)
                    ws = trainer.step(f.cuda(), n.cuda(), t.cuda(), use_teacher_forcing=True)
                    losses.append(trainer.read_stats(ws)[0])
                c = next(iter(trainer._shapes.values()))
                runs[name] = dict(flat=model.store.flat.cpu().numpy().copy(), losses=losses, graphs=len(c["graphs"]),
                                  sweeps=bool(getattr(ws, "_sweep_ok", False) and getattr(ws, "_sweep_bwd_ok", False)))
            out[mixed] = runs
        ops.set_mixed_precision(False)
        os.environ["ASR_NATIVE_COLLECTIVE"] = "0"
        q.put(("ok", out))
    except BaseException:
        import traceback
        q.put(("error", traceback.format_exc()))
        raise
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_data_parallel_code_path_on_one_gpu_with_a_single_rank_rccl_group():
    """what a replica of an N-GPU job executes - one captured graph per gradient bucket's backward segment, the
    bucket all-reduces issued through torch.distributed "nccl" (= RCCL) on the communication stream between them, events both ways,
    bf16 wire format under mixed precision - on one GPU with a world-size-1 RCCL group (TrainStep(force_dp_path=True)).  After four
    steps (eager, capture, two replays) the parameters equal the single-graph step's: f32 wire to the run-to-run noise of the
    atomically accumulated weight gradients (1e-5 of the largest parameter; two runs of the SAME path differ by as much), bf16 wire to
    the rounding of the gradients (2^-8 relative per bucket entry, through Adam: 2e-2 of the largest parameter change)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_dp_path_worker, args=(29700 + (os.getpid() % 1500), q))
    p.start()
    r = q.get(timeout=240)
    p.join(60)
    if r[0] == "error":
        pytest.fail(r[1])
    for mixed, runs in r[1].items():
        single, again = runs["single"], runs["single_again"]
        assert single["graphs"] == 1 and runs["dp"]["graphs"] == 2 + (2 + CFG["num_encoder_layers"]), (single["graphs"], runs["dp"]["graphs"])
        assert runs["dp_native"]["graphs"] == 1, "native collectives: forward, backward with the collectives inside and the update, one graph"
        scale = np.abs(single["flat"]).max()
        noise = np.abs(single["flat"] - again["flat"]).max() / scale
        for name in ("dp", "dp_native"):
            dp = runs[name]
            assert single["sweeps"] and dp["sweeps"], "one rank owns its GPU: the decoder sweeps stay on in the data-parallel path"
            diff = np.abs(single["flat"] - dp["flat"]).max() / scale
            print(f"mixed={mixed}: {name} vs single graph {diff:.2e} (same path twice {noise:.2e})")
            if not mixed:
                assert diff <= max(1e-5, 10 * noise), (name, diff, noise)
                assert all(abs(a - b) <= 1e-5 * abs(a) for a, b in zip(single["losses"], dp["losses"]))
            else:
                # (lr 1e-2 x 4 steps of sign-like Adam updates, gradients rounded to bf16 on the wire.)  Under mixed precision two runs of the
                # SAME path occasionally differ by as much as the full 4 x lr on single entries - an f32 difference from the atomically
                # accumulated sums that straddles a bf16 rounding boundary moves a product by 2^-8, and Adam turns the sign of a near-zero
                # gradient into a whole update (seen: 2.9e-8, 4e-3, 3.6e-2) - so the bound follows the noise measured in this very process
                assert diff <= max(2e-2, 1.5 * noise), (name, diff, noise)
                assert all(abs(a - b) <= 2e-2 * abs(a) for a, b in zip(single["losses"], dp["losses"]))
