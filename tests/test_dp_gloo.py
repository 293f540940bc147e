"""CPU, world_size 2, gloo: the data-parallel gradient exchange.  Every rank builds the same flat
ParamStore layout, fills its gradient buffer with rank-dependent values scaled by 1/world (as the
training step does), exchanges bucket by bucket, and must end with the replica mean in every element -
bucket views must tile the buffer exactly, in the order the backward pass completes them."""
import os
from collections import OrderedDict

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import speech_recognition_amd  # noqa: F401
        from speech_recognition_amd.params import ParamStore
        from speech_recognition_amd.training import GradientExchange
        shapes = OrderedDict([("dec/w", (7, 5)), ("dec/b", (5,)), ("enc/w", (3, 3, 2, 4)), ("enc/b", (4,)), ("enc/u", (9, 11))])
        store = ParamStore(shapes, [["dec/w", "dec/b"], ["enc/w", "enc/b", "enc/u"]], device="cpu")
        # bucket ranges tile [0, numel) and every variable lies inside exactly one bucket
        assert store.bucket_ranges[0][0] == 0 and store.bucket_ranges[-1][1] == store.numel
        assert store.bucket_ranges[0][1] == store.bucket_ranges[1][0]
        for n in shapes:
            o = store.offsets[n]
            assert o % 4 == 0 and sum(a <= o < b for a, b in store.bucket_ranges) == 1
        for i, n in enumerate(shapes):
            store.g[n].fill_(float((rank + 1) * (i + 1)) / world)      # local gradient already scaled by 1/world
        ex = GradientExchange(world)
        for bucket in store.bucket_views():
            ex.reduce_async(bucket)
        ex.wait()
        ok = True
        for i, n in enumerate(shapes):
            expect = sum((r + 1) * (i + 1) for r in range(world)) / world   # replica mean
            ok &= bool(torch.allclose(store.g[n], torch.full(shapes[n], expect)))
        # padding between variables is never touched by a named view
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_bucketed_gradient_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in procs)
    for p in procs:
        p.join(30)
    assert res == [(0, True), (1, True)]


def test_single_replica_exchange_is_a_no_op():
    import speech_recognition_amd  # noqa: F401
    from speech_recognition_amd.training import GradientExchange
    t = torch.arange(5.0)
    ex = GradientExchange(1)
    ex.reduce_async(t)
    ex.wait()
    assert torch.equal(t, torch.arange(5.0))


# ---------------------------------------------------------------------------------------------- replica synchronisation
class _FakeModel:
    """The parts of a model that training.sync_replicas touches, on the CPU, initialised differently on every rank."""

    def __init__(self, rank):
        import random
        from collections import OrderedDict

        from speech_recognition_amd.params import ParamStore
        g = torch.Generator().manual_seed(1000 + rank)
        shapes = OrderedDict([("a/w", (5, 3)), ("a/b", (3,)), ("b/w", (4, 4))])
        self.store = ParamStore(shapes, [["a/w", "a/b"], ["b/w"]], device="cpu")
        for t in (self.store.flat, self.store.adam_m, self.store.adam_v):
            t.copy_(torch.randn(t.shape, generator=g))
        self.buffers = {"bn/moving_mean": torch.randn(4, generator=g), "bn/moving_variance": torch.rand(4, generator=g)}
        self.state = torch.tensor([rank * 7, 100 + rank, 0, 0], dtype=torch.int32)
        self._py_rng = random.Random(rank)
        self.changed = 0

    def weights_changed(self):
        self.changed += 1


def _sync_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import speech_recognition_amd  # noqa: F401
        from speech_recognition_amd.training import sync_replicas
        m = _FakeModel(rank)
        sync_replicas(m, world)
        q.put((rank, m.store.flat.clone().numpy(), m.store.adam_m.clone().numpy(), m.store.adam_v.clone().numpy(),
               {k: v.clone().numpy() for k, v in m.buffers.items()}, m.state.clone().numpy(), [m._py_rng.random() for _ in range(3)], m.changed))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_sync_replicas_gives_every_rank_rank0_state():
    """replicas built without a seed start from different weights; TrainStep broadcasts rank 0's parameters,
    Adam moments, BatchNorm statistics, device state words and the teacher-forcing RNG before the first step."""
    import numpy as np
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_sync_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=90) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(30)
    r0, r1 = res
    ref = _FakeModel(0)
    assert np.array_equal(r0[1], ref.store.flat.numpy()) and np.array_equal(r0[5], ref.state.numpy())    # rank 0 is the source
    for i in (1, 2, 3, 5):
        assert np.array_equal(r0[i], r1[i]), i
    for k in r0[4]:
        assert np.array_equal(r0[4][k], r1[4][k]), k
    assert r0[6] == r1[6]                 # the same teacher-forcing coins from here on
    assert r0[7] == r1[7] == 1            # packed weight images are refreshed


# ---------------------------------------------------------------------------------------------- the sticky error flag under data parallelism
def _flag_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import speech_recognition_amd  # noqa: F401
        from speech_recognition_amd.params import ParamStore
        from speech_recognition_amd.training import GradientExchange
        shapes = OrderedDict([("voc/w", (6, 9)), ("dec/w", (7, 5)), ("enc2/w", (4, 4)), ("enc1/w", (3, 8)), ("enc0/w", (5, 5))])
        store = ParamStore(shapes, [[n] for n in shapes], device="cpu")
        schedule = [[], [0, 1], [2], [3], [4]]                 # LAS.bucket_schedule(): the vocabulary bucket travels with the decoder's
        results = []
        for step, failing_rank in enumerate((None, 1, None, 0)):
            store.grad.zero_()                                  # (the step's fill: clears the flag too - it lives in the last bucket)
            for i, n in enumerate(shapes):
                store.g[n].fill_(float(rank + 1 + step) / world)
            if failing_rank == rank:
                store.err_flag.fill_(1.0)                       # what a timed-out sweep on THIS replica leaves behind
            ex = GradientExchange(world)
            buckets = store.bucket_views()
            for done in schedule:                               # bucket k reduced when its segment has been enqueued
                for b in done:
                    ex.reduce_async(buckets[b])
            ex.wait()
            skip = bool(store.err_flag[0] != 0)                 # adam_step / advance_state read exactly this cell (skip_flag)
            mean = sum(r + 1 + step for r in range(world)) / world
            grads_ok = all(bool(torch.allclose(store.g[n], torch.full(shapes[n], mean))) for n in shapes)
            results.append((skip, float(store.err_flag[0]), grads_ok))
        q.put((rank, results))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_one_replicas_sweep_timeout_makes_every_rank_skip_the_same_step():
    """the sweeps' time-out flag lives in the last gradient bucket, so the all-reduce that sums the gradients also
    tells every replica that SOME replica's step is invalid: all ranks skip that update (parameters stay bit-identical) and none of
    the others.  World size 2 on gloo, the bucket schedule LAS uses under data parallelism (vocabulary bucket held back one segment)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_flag_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=90) for _ in procs)
    for p in procs:
        p.join(30)
    assert res[0] == res[1], "both replicas must see the same flag and the same reduced gradients"
    assert [s for s, _, _ in res[0]] == [False, True, False, True]
    assert all(ok for _, _, ok in res[0])


def test_bucket_schedules_reduce_every_bucket_exactly_once():
    """The per-segment bucket schedule of both models (with and without the overlap scheduler) is a partition of the buckets, and no
    bucket is scheduled before the segment that completes it."""
    import speech_recognition_amd  # noqa: F401
    from speech_recognition_amd import layers
    from speech_recognition_amd.models import LAS, DeepSpeech2
    for overlap in (False, True):
        layers.Overlap.enabled = overlap
        try:
            las = LAS("lstm", 40, 16, 16, 3, 2, 0.1, 0.9, device="cpu")
            ds2 = DeepSpeech2(2, [4, 4], [[5, 3], [5, 3]], [[2, 2], [1, 2]], "gru", 4, 16, 0.1, 0.0, 20, 3, 0, device="cpu")
            for m, nb in ((las, 2 + 3), (ds2, 1 + 4)):
                sched = m.bucket_schedule()
                assert len(sched) == nb and sorted(b for seg in sched for b in seg) == list(range(nb)), (type(m).__name__, overlap, sched)
                assert all(b <= k for k, seg in enumerate(sched) for b in seg), "a bucket is complete only after its own segment"
        finally:
            layers.Overlap.enabled = False
