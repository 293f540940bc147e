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
