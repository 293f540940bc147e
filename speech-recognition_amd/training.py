"""The training step of run/train.py:158-217 (Keras ``compile`` + ``fit`` inner loop) on MI355X.

One step = [audio front end] -> forward(training=True) -> loss -> backward -> (data-parallel gradient
all-reduce) -> Adam with the LRScheduler -> advance the step counter / dropout seed.  Everything
between the host->device copy of the batch and the parameter update is kernels of libasr_mi355x.so
on one HIP stream; with ``use_graph`` the sequence is captured once per input shape into hipGraphs
and replayed, which removes the per-launch host cost of the ~10^3-10^4 small dependent kernels of
the recurrent sweeps (launch boundaries inside a graph cost ~1.5 us on MI355X).

Data parallelism (utils.py:148-149 MirroredStrategy in the reference): one process per GPU; the flat
gradient buffer is split into buckets in the order the backward pass completes them (decoder side
first), and bucket k is all-reduced over RCCL on a side stream while backward segment k+1 runs.
Each rank scales its loss gradient by 1/world_size, so the summed gradient is the replica mean
([TF-sem] Keras scales SUM_OVER_BATCH_SIZE losses by 1/num_replicas).
"""
from typing import Dict, Optional

import torch

import os

from . import ops
from . import rng as R

_ROCTX = os.environ.get("ASR_ROCTX", "1") != "0"


class NativeComm:
    """An RCCL communicator owned by libasr_mi355x.so (asr_comm_init): the collective entry point of the C ABI
    (asr_allreduce_bucket) issues straight into RCCL on a HIP stream, so the bucket all-reduces can be captured into the same
    hipGraph as the backward segments around them.  The 128-byte unique id travels through torch.distributed's process group
    (any backend); with one rank nothing travels."""

    def __init__(self, world: int, rank: int, group=None):
        import ctypes as C

        from . import _lib
        self.lib = _lib.load()
        if not self.lib.asr_comm_available():
            raise RuntimeError("librccl.so.1 cannot be resolved in this process")
        buf = C.create_string_buffer(128)
        if rank == 0:
            _lib.check(self.lib.asr_comm_unique_id(buf))
        ident = bytes(buf.raw)
        if world > 1:
            import torch.distributed as dist
            box = [ident]
            dist.broadcast_object_list(box, src=0, group=group)
            ident = box[0]
        self.handle = C.c_void_p()
        _lib.check(self.lib.asr_comm_init(ident, int(world), int(rank), C.byref(self.handle)))
        self.world, self.rank = world, rank

    def all_reduce(self, bucket: torch.Tensor, wire: Optional[torch.Tensor] = None):
        """In-place SUM of the f32 `bucket` over the ranks, on the current stream; wire: bf16 staging of the same length or None."""
        import ctypes as C

        from . import _lib
        assert bucket.dtype == torch.float32 and bucket.is_contiguous() and bucket.is_cuda
        _lib.check(self.lib.asr_allreduce_bucket(self.handle, C.c_void_p(bucket.data_ptr()), bucket.numel(),
                                                 C.c_void_p(wire.data_ptr()) if wire is not None else None,
                                                 C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def close(self):
        if self.handle:
            self.lib.asr_comm_destroy(self.handle)
            self.handle = None


class GradientExchange:
    """Data-parallel gradient exchange (the implicit all-reduce of tf.distribute.MirroredStrategy,
    utils.py:148-149): sums the buckets of a flat gradient buffer across the ranks of a process group.
    On GPUs the reduction of bucket k is issued on a side stream as soon as backward segment k+1 has
    been enqueued, so RCCL traffic over xGMI overlaps the remaining backward kernels; on CPU tensors
    (gloo; used by the unit tests) it runs synchronously.  Replicas are expected to have scaled their
    loss gradient by 1/world_size, so SUM yields the replica mean.
    wire_dtype=torch.bfloat16 (the las_large configuration): a bucket crosses the fabric as bf16 - converted on
    the communication stream before and after the collective, half the bytes on the 153 GB/s xGMI links - and the sum of
    the rounded replica gradients lands back in the f32 buffer (every rank ends with the same values)."""

    def __init__(self, world_size: int, group=None, compute_stream=None, wire_dtype=torch.float32, force=False, native: Optional["NativeComm"] = None):
        """force: run the exchange (side stream, events, wire conversion, the collective itself) even for a single rank - the
        data-parallel code path on one GPU (bench.py --dp-path, tests/test_dp_gpu.py); needs an initialised process group.
        native: a NativeComm - the collective is then the C ABI's asr_allreduce_bucket (RCCL called from the library, capturable)
        instead of torch.distributed's all_reduce."""
        self.world, self.group, self.stream = world_size, group, compute_stream
        self.native = native
        self.active = world_size > 1 or bool(force)
        self.comm_stream = torch.cuda.Stream() if (compute_stream is not None and self.active) else None
        self.wire_dtype = wire_dtype
        self._wire = {}                                   # bucket address -> bf16 staging buffer
        self._pending = []

    def _staging(self, bucket):
        key = (bucket.data_ptr(), bucket.numel())
        if key not in self._wire:
            self._wire[key] = torch.empty(bucket.numel(), dtype=torch.bfloat16, device=bucket.device)
        return self._wire[key]

    def _all_reduce(self, bucket, dist):
        if self.native is not None and bucket.is_cuda:
            self.native.all_reduce(bucket, self._staging(bucket) if self.wire_dtype == torch.bfloat16 else None)
            return
        if self.wire_dtype != torch.bfloat16:
            dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group)
            return
        wire = self._staging(bucket)
        if bucket.is_cuda:
            ops.f32_to_bf16(bucket, wire)
            dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group)
            ops.bf16_to_f32(wire, bucket)
        else:                                             # the gloo unit test only: no product path runs on CPU tensors
            wire.copy_(bucket)
            dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group)
            bucket.copy_(wire)

    def reduce_async(self, bucket: torch.Tensor):
        """Start summing `bucket` (a contiguous 1-D view) across ranks."""
        if not self.active or bucket is None or bucket.numel() == 0:
            return
        import torch.distributed as dist
        if self.comm_stream is None:                      # CPU / gloo path
            self._all_reduce(bucket, dist)
            return
        ready = torch.cuda.Event()
        ready.record(self.stream)
        with torch.cuda.stream(self.comm_stream):
            self.comm_stream.wait_event(ready)
            self._all_reduce(bucket, dist)
            done = torch.cuda.Event()
            done.record(self.comm_stream)
        self._pending.append(done)

    def wait(self):
        """Make the compute stream wait for every reduction started since the last wait()."""
        for ev in self._pending:
            self.stream.wait_event(ev)
        self._pending = []


def sync_replicas(model, world: int, group=None, src: int = 0):
    """Make every data-parallel replica start from rank `src`'s state (tf.distribute.MirroredStrategy creates mirrored,
    identical variables - utils.py:148-149; here every rank builds its own model, so without this they would differ whenever
    no --seed is given): parameters, Adam moments, BatchNorm moving statistics, the device state words (step counter, dropout
    seed) and the host RNG of the teacher-forcing coin.  Gradients alone are exchanged afterwards, so replicas that start
    equal stay equal.  Works on whatever device the tensors live on (RCCL for GPU tensors under "nccl", gloo in the tests)."""
    if world <= 1:
        return
    import random

    import torch.distributed as dist
    st = model.store
    for t in (st.flat, st.adam_m, st.adam_v):
        dist.broadcast(t, src=src, group=group)
    for name in sorted(getattr(model, "buffers", {})):
        dist.broadcast(model.buffers[name], src=src, group=group)
    dist.broadcast(model.state, src=src, group=group)
    seed = [random.randrange(2 ** 31)]
    dist.broadcast_object_list(seed, src=src, group=group)
    if hasattr(model, "_py_rng"):
        model._py_rng = random.Random(seed[0])
    if hasattr(model, "weights_changed"):
        model.weights_changed()


def sweep_diagnoses(ws, clear=True):
    """What the one-launch sweeps of a model workspace have on record about hand-off time-outs (ops.sweep_diagnosis: stage, step,
    first workgroup that gave up, 'absent workgroup' or 'lost hand-off'), one dict per sweep that gave up since the last call."""
    out = []
    for i, lw in enumerate(getattr(ws, "layers", [])):
        buf = lw.get("rnn") if isinstance(lw, dict) else None
        if not buf:
            continue
        for key, kind in (("persist_ws", "rnn_sweep_fwd"), ("persist_bwd_ws", "rnn_sweep_bwd"), ("wide_ws", "rnn_sweep_wide"),
                          ("wide_bwd_ws", "rnn_sweep_wide_bwd")):
            if key in buf:
                d = ops.sweep_diagnosis(buf[key], kind, clear=clear)
                if d:
                    out.append(dict(d, layer=i))
    for key, kind in (("dsweep_ws", "decoder_sweep_fwd"), ("dsweep_bwd_ws", "decoder_sweep_bwd")):
        w = getattr(ws, key, None)
        if w is not None:
            d = ops.sweep_diagnosis(w, kind, decoder=True, clear=clear)
            if d:
                out.append(d)
    return out


class SweepTimeout(RuntimeError):
    """A hand-off of a one-launch sweep timed out; `.reports` holds sweep_diagnoses() of the affected workspace."""

    def __init__(self, msg, reports):
        super().__init__(msg)
        self.reports = reports


class TrainStep:
    def __init__(self, model, lr_schedule, frontend: Optional[ops.LogmelPlan] = None, strategy=None, use_graph: bool = True,
                 beta1=0.9, beta2=0.999, eps=1e-7, eval_frontend=None, force_dp_path: bool = False):
        """model: LAS or DeepSpeech2 (built lazily on the first batch); lr_schedule: utils.LRScheduler;
        frontend: LogmelPlan when batches arrive as raw audio, StoredFeaturePlan when they are stored
        log-mel frames, None when they are finished feature tensors; eval_frontend: the same without
        SpecAugment, used by evaluate().  force_dp_path: take the data-parallel step (one captured graph per gradient bucket's
        backward segment, bucket all-reduces on the communication stream between them, bf16 wire under mixed precision) even
        with a single rank - what a replica of an N-GPU job executes, measurable and checkable on one GPU."""
        self.model, self.frontend, self.strategy = model, frontend, strategy
        self.eval_frontend = eval_frontend
        self.sched_host = lr_schedule
        self.sched = lr_schedule.device_schedule()
        self.use_graph = use_graph
        self.beta1, self.beta2, self.eps = beta1, beta2, eps
        self.world = strategy.world_size if strategy is not None else 1
        self.group = getattr(strategy, "group", None)
        self._shapes: Dict[tuple, dict] = {}
        self.max_shapes = 16                       # input shapes kept alive (workspace + captured graphs each), least recently used evicted
        self.stream = torch.cuda.Stream()          # graphs cannot capture the legacy default stream
        # --mixed-precision (BASELINE configs[4], las_large): gradients cross the fabric as bf16
        self.segmented = self.world > 1 or bool(force_dp_path)
        wire = torch.bfloat16 if (ops.mixed_precision() and self.segmented) else torch.float32
        # ASR_NATIVE_COLLECTIVE=1: the bucket all-reduces go through the C ABI (asr_allreduce_bucket: RCCL called by the library on the
        # communication stream) and, being capturable, the whole data-parallel step becomes ONE hipGraph (forward, every backward
        # segment, the collectives forked beside them) instead of one graph per segment with host-side collectives in between.
        # Default off until a multi-GPU node has run it: the torch.distributed route is the one the N > 1 CPU / GPU tests cover.
        native = None
        if self.segmented and os.environ.get("ASR_NATIVE_COLLECTIVE", "0") == "1":
            native = NativeComm(self.world, strategy.rank if strategy is not None else 0, self.group)
        self.exchange = GradientExchange(self.world, self.group, self.stream, wire_dtype=wire, force=force_dp_path, native=native)
        self.one_graph = native is not None
        model.bucket_sync = self.segmented           # per-bucket completion of side-stream work only when buckets are exchanged
        if self.world > 1:
            self._detect_shared_device()
        self.iterations = 0
        self._replicas_synced = False
        if getattr(model, "built", False):
            self.sync_replicas()

    def _detect_shared_device(self):
        """Ranks that share one GPU (a test arrangement; run.train and bench.py bind one process per GPU) cannot both run whole-chip
        sweeps: see ops.set_device_exclusive.  Every rank learns every rank's (host, device) and all of them take the same decision."""
        import socket

        import torch.distributed as dist
        dev = torch.device(getattr(self.model, "device", None) or "cuda")
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        try:
            ident = str(torch.cuda.get_device_properties(idx).uuid)
        except Exception:
            ident = f"cuda:{idx}"
        mine = (socket.gethostname(), ident)
        everyone = [None] * self.world
        dist.all_gather_object(everyone, mine, group=self.group)
        shared = len(set(everyone)) < len(everyone)
        self.shared_device = shared
        if shared:
            ops.set_device_exclusive(False)

    def sync_replicas(self):
        """Broadcast rank 0's model state to every replica (see training.sync_replicas); called once the model is built, and
        again by the caller after anything that changes one replica only (load_weights on rank 0, ...)."""
        if self.world > 1:
            self.synchronize()
            sync_replicas(self.model, self.world, self.group)
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        self._replicas_synced = True

    # ------------------------------------------------------------------------------------------ buffers
    def _ctx(self, audio, n_samples, tokens):
        key = (tuple(audio.shape), tuple(tokens.shape))
        if key in self._shapes:
            self._shapes[key] = self._shapes.pop(key)      # most recently used last
            return self._shapes[key]
        if len(self._shapes) >= self.max_shapes:           # real data: every padded batch shape has its own buffers + graphs
            self.synchronize()
            old = self._shapes.pop(next(iter(self._shapes)))
            self.model.release_workspace(old["ws"])
            old.clear()
        dev = self.model.device
        c = dict(audio=torch.empty(audio.shape, dtype=torch.float32, device=dev),
                 n_samples=torch.empty(audio.shape[0], dtype=torch.int32, device=dev) if self.frontend else None,
                 tokens=torch.empty(tokens.shape, dtype=torch.int32, device=dev), graphs={}, warm=set())
        B = audio.shape[0]
        if self.frontend is not None:
            T = self.frontend.num_frames(audio.shape[1])
            c["feats"] = torch.empty(B, T, self.frontend.num_mel_bins, self.frontend.channels, device=dev)
        else:
            c["feats"] = c["audio"]
        f = c["feats"]
        self.model._ensure_built(f.shape[2], f.shape[3])
        if not self._replicas_synced:                      # the model was built lazily on this first batch
            self.sync_replicas()
        c["ws"], c["labels"] = self.model.train_workspace(B, f.shape[1], tokens.shape[1])
        # The buffers above were created on whatever stream was current (the caller's: normally the default stream), some of them
        # by an asynchronous fill kernel (torch.zeros / torch.ones); the step runs on self.stream, a non-blocking stream that is not
        # ordered with the default one.  Found in round 4: with a foreign kernel holding most compute units the zero-fill of the token
        # rows landed AFTER the first step had written and read them in its forward pass and BEFORE its backward pass - the embedding
        # gradient of that step then went to row 0.  New shapes are rare: wait for the device once.
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self._shapes[key] = c
        return c

    # ------------------------------------------------------------------------------------------ pieces
    def _fwd_loss(self, c, teacher):
        m = self.model
        if self.frontend is not None:
            self.frontend(c["audio"], c["n_samples"], c["feats"].shape[1], seed=m.seed if self.frontend.cfg.sa_enable else None,
                          out=c["feats"])
        ops.fill(m.store.grad, 0.0)
        m.pack_weights()
        m.forward_ws(c["ws"], c["feats"], True, teacher)
        m.loss_and_grad(c["ws"], c["labels"], 1.0 / self.world)

    def _update(self):
        m = self.model
        # store.err_flag (inside the last gradient bucket, so already summed over the replicas): a recurrent sweep timed out somewhere
        # in this step - the update is skipped on every rank and the sticky error word state[2] is set instead
        ops.adam_step(m.store.flat, m.store.grad, m.store.adam_m, m.store.adam_v, m.state, self.sched, self.beta1, self.beta2, self.eps,
                      skip_flag=m.store.err_flag)
        ops.advance_state(m.state, m.store.err_flag)

    def _segments(self, c, teacher):
        """The step as a list of stream-ordered callables.  With torch.distributed collectives (not capturable around the bucket
        hand-over): segment 0 = front end + forward + loss, then one segment per gradient bucket (bucket k is complete after segment
        k+1), the update after the last all-reduce.  A single replica, and replicas on the library's own capturable collectives
        (ASR_NATIVE_COLLECTIVE=1), run the whole step as ONE callable = one captured graph."""
        segs = [lambda: self._fwd_loss(c, teacher)]
        bsegs = self.model.backward_segments(c["ws"], c["feats"])
        if self.segmented and self.one_graph:
            # native collectives: one callable = one captured graph; the bucket all-reduces are enqueued (forked onto the communication
            # stream) at the points of the backward pass where their buckets complete, and joined before the update
            def whole():
                m = self.model
                buckets = m.store.bucket_views()
                done = m.bucket_schedule() if hasattr(m, "bucket_schedule") else [[k] for k in range(len(buckets))]
                self._fwd_loss(c, teacher)
                for k, fn in enumerate(bsegs):
                    fn()
                    for b in done[k]:
                        self.exchange.reduce_async(buckets[b])
                self.exchange.wait()
                self._update()
            return [whole]
        elif self.segmented:
            segs += bsegs
        else:
            # one replica: forward, backward and the update are ONE callable = one captured graph per step (round 4; three before)
            def whole_step():
                self._fwd_loss(c, teacher)
                for fn in bsegs:
                    fn()
                self._update()
            return [whole_step]
        return segs

    def _run_segment(self, c, teacher, k, fn):
        # roctx range per segment (torch.cuda.nvtx is roctx on ROCm): rocprofv3 --marker-trace shows "asr:seg<k>" around the launches /
        # graph replays of each stage of the step (tracing)
        if _ROCTX:
            torch.cuda.nvtx.range_push(f"asr:seg{k}" if k != "update" else "asr:update")
            try:
                self._run_segment_inner(c, teacher, k, fn)
            finally:
                torch.cuda.nvtx.range_pop()
        else:
            self._run_segment_inner(c, teacher, k, fn)

    def _run_segment_inner(self, c, teacher, k, fn):
        if not self.use_graph:
            fn()
            return
        gkey = (teacher, k)
        if gkey not in c["warm"]:        # first call per segment runs eagerly (allocations, lazy init)
            fn()
            c["warm"].add(gkey)
            return
        if gkey not in c["graphs"]:
            g = torch.cuda.CUDAGraph()
            # thread_local: RCCL's watchdog thread polls events while this thread captures; only this
            # thread's own calls are part of (and can invalidate) the capture
            with torch.cuda.graph(g, stream=self.stream, capture_error_mode="thread_local"):
                fn()
            c["graphs"][gkey] = g
        c["graphs"][gkey].replay()

    # ------------------------------------------------------------------------------------------ step
    def step(self, audio, n_samples, tokens, use_teacher_forcing: Optional[bool] = None):
        """audio: f32 [B, N] raw samples (frontend given) or f32 [B,T,F,C] features; n_samples: i32 [B]
        (ignored for features); tokens: i32 [B, L] full token rows (BOS ... EOS, zero padded).
        Returns the workspace whose ``stats`` tensor holds [loss, #correct, #kept] on the device."""
        m = self.model
        c = self._ctx(audio, n_samples, tokens)
        if use_teacher_forcing is None:
            use_teacher_forcing = m.draw_teacher_forcing()
        teacher = bool(use_teacher_forcing)
        with torch.cuda.stream(self.stream):
            c["audio"].copy_(audio, non_blocking=True)
            if c["n_samples"] is not None:
                c["n_samples"].copy_(n_samples, non_blocking=True)
            c["tokens"].copy_(tokens, non_blocking=True)
            m.set_targets(c["ws"], c["tokens"], c["labels"])   # layout copies of the token rows (not captured)
            segs = self._segments(c, teacher)
            buckets = m.store.bucket_views() if (self.segmented and not self.one_graph) else []
            assert not buckets or len(buckets) == len(segs) - 1, "one gradient bucket per backward segment"
            # buckets complete when backward segment k ends (a model that runs a stage's weight gradients beside the next stage's sweep
            # completes them one segment late: bucket_schedule)
            done = m.bucket_schedule() if hasattr(m, "bucket_schedule") else [[k] for k in range(len(buckets))]
            assert not buckets or sorted(b for d in done for b in d) == list(range(len(buckets))), "every bucket is reduced exactly once"
            for k, fn in enumerate(segs):
                self._run_segment(c, teacher, k, fn)
                if k >= 1 and buckets:
                    for b in done[k - 1]:
                        self.exchange.reduce_async(buckets[b])
            self.exchange.wait()
            if self.segmented and not self.one_graph:
                self._run_segment(c, teacher, "update", self._update)
        m.weights_changed()
        self.iterations += 1
        return c["ws"]

    def evaluate(self, audio, n_samples, tokens, use_teacher_forcing: Optional[bool] = None):
        """Validation pass of model.fit (run/train.py:203): forward(training=False) + loss + metric on one
        batch, no SpecAugment, no update.  The teacher-forcing coin is drawn as in training (las.py:366 has
        no training guard).  Returns the host list [loss, #correct, #kept]."""
        m = self.model
        c = self._ctx(audio, n_samples, tokens)
        if use_teacher_forcing is None:
            use_teacher_forcing = m.draw_teacher_forcing()
        fe = self.eval_frontend if self.eval_frontend is not None else self.frontend
        with torch.cuda.stream(self.stream):
            c["audio"].copy_(audio, non_blocking=True)
            if c["n_samples"] is not None:
                c["n_samples"].copy_(n_samples, non_blocking=True)
            c["tokens"].copy_(tokens, non_blocking=True)
            m.set_targets(c["ws"], c["tokens"], c["labels"])
            if getattr(m.store, "err_flag", None) is not None:
                ops.fill(m.store.err_flag, 0.0)            # a time-out in this forward-only pass must be seen below, not erased by the next step
            if fe is not None:
                if fe.cfg.sa_enable:
                    raise ValueError("evaluate: the evaluation front end must not apply SpecAugment")
                fe(c["audio"], c["n_samples"], c["feats"].shape[1], out=c["feats"])
            m.pack_weights()
            m.forward_ws(c["ws"], c["feats"], False, bool(use_teacher_forcing))
            m.loss_and_grad(c["ws"], c["labels"], 1.0)
        return self.read_stats(c["ws"], check_flag=True)

    def synchronize(self):
        self.stream.synchronize()
        if self.exchange.comm_stream is not None:
            self.exchange.comm_stream.synchronize()

    def read_stats(self, ws, check_flag=False):
        """Host copy of [loss, #correct, #kept] (synchronises).  Also checks the model's sticky error word: if a hand-off of a
        one-launch recurrent sweep timed out in ANY step since the last check (on any rank), those steps were skipped (parameters,
        moments and the step counter untouched) and this raises - a time-out is never silent and never trains on garbage.
        check_flag: also look at the step's own error flag (a forward-only pass - evaluate() - never reaches the update that folds
        the flag into the sticky word)."""
        self.synchronize()
        st = self.model.state.cpu()
        flag = float(self.model.store.err_flag[0]) if (check_flag and getattr(self.model.store, "err_flag", None) is not None) else 0.0
        if int(st[2]) != 0 or flag != 0.0:
            self.model.state[2] = 0
            if flag != 0.0:
                ops.fill(self.model.store.err_flag, 0.0)
            reports = sweep_diagnoses(ws)
            detail = "".join(f"\n  {r}" for r in reports)
            what = "the affected training steps were skipped" if not check_flag else "the results of this pass are invalid"
            raise SweepTimeout("one-launch recurrent sweep: an inter-workgroup hand-off timed out; " + what + detail +
                               "\n  ('absent workgroup': a compute unit was held by another tenant for the whole spin limit; 'lost hand-off': all "
                               "workgroups were resident) - rerun with ASR_PERSISTENT_RNN=0 to use the per-step kernels", reports)
        return [float(v) for v in ws.stats[:3].cpu()]
