#!/usr/bin/env python3
"""Headline benchmark: audio-seconds/second of training (BASELINE.json metric) on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload las_small|deepspeech|las_large]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default workload = SURVEY.md 8d config 2 (the one BASELINE.json's metric is quoted on): las_small.yml +
libri_config.yml, synthetic 10 s / 16 kHz clips, batch 32 per GPU, 65-token rows (64 decoder steps),
SpecAugment (F=27, m_F=2, T=100, p=1.0, m_T=2) and delta features computed ON the GPU inside the step,
dropout 0.15 active, teacher forcing on, forward + backward + (RCCL gradient all-reduce) + Adam(lr 2e-4,
LRScheduler).  fp32 throughout.  --workload deepspeech / las_large run SURVEY.md 8d configs 4 / 5 the
same way (las_large in fp32: this build has no bf16 path, the line says so).
Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the roofline / cpu_baseline fields).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

SAMPLE_RATE = 16000
PEAK_F32_MFMA = 157.3e12   # MI355X_MICROARCH.md: dense f32-input MFMA peak
PEAK_BF16_MFMA = 2.5e15    # MI355X_MICROARCH.md: dense bf16 MFMA peak
PEAK_HBM = 8.0e12          # MI355X_MICROARCH.md: HBM3E ~8 TB/s

# SURVEY.md 8d / BASELINE.md 4: algorithmic training flops per step = 3 x forward, key projection counted once
WORKLOADS = {
    # traffic: HBM-side bytes per step from the PMC passes of profiles/r01_las_small_pmc_hbm_traffic.txt
    # (2 x FETCH_SIZE per the gfx950 correction for wide reads + WRITE_SIZE) - an upper estimate, measured offline
    "las_small": dict(model="las_small.yml", clip_seconds=10.0, batch=32, tokens=65, flops=363.3e9, traffic=2 * 7.34e9 + 4.12e9,
                      metric="audio-seconds/sec training (las_small, 10s clips, bs32)",
                      text="las_small.yml + libri_config.yml, synthetic 10 s 16 kHz clips, batch 32 per GPU, 64 decoder steps, "
                           "SpecAugment+delta on GPU, dropout 0.15, teacher forcing on, fwd+bwd+Adam(lr 2e-4)"),
    "deepspeech": dict(model="deepspeech.yml", clip_seconds=15.0, batch=16, tokens=96, flops=593.4e9,
                       metric="audio-seconds/sec training (deepspeech, 15s clips, bs16)",
                       text="deepspeech.yml + libri_config.yml, synthetic 15 s 16 kHz clips, batch 16 per GPU, 96 CTC labels, "
                            "SpecAugment+delta on GPU, dropout 0.1, mask mode 'intended', fwd+CTC+bwd+Adam(lr 2e-4)"),
    "las_large": dict(model="las_large.yml", clip_seconds=20.0, batch=64, tokens=128, flops=22.85e12,
                      metric="audio-seconds/sec training (las_large, 20s clips, bs64)",
                      text="las_large.yml + libri_config.yml, synthetic 20 s 16 kHz clips, batch 64 per GPU, 127 decoder steps, "
                           "SpecAugment+delta on GPU, teacher forcing on, fwd+bwd+Adam(lr 2e-4)", precision="bf16"),
}
PRECISION_TEXT = {"f32": "f32 throughout",
                  "bf16": "mixed precision: dense contractions (gemm) and the wide (H >= 512) recurrent step kernels with bf16 operands on "
                          "the bf16 MFMA, f32 accumulation; other recurrent cells, convolutions, softmax/CTC, BN, Adam and all storage f32"}


def load_yaml(name):
    import yaml
    with open(os.path.join(ROOT, "resources", "configs", name)) as f:
        return yaml.safe_load(f)


def synthetic_batch(rank, wl, B=None):
    """audio N(0, 0.1^2) clipped to [-1,1], seed 1234(+rank); tokens [2, U{17..15999}..., 3], seed 4321
    (DeepSpeech2 labels avoid the blank index 14 - they are >= 17 anyway)."""
    B = B or wl["batch"]
    g = np.random.default_rng(1234 + rank)
    audio = np.clip(g.standard_normal((B, int(wl["clip_seconds"] * SAMPLE_RATE)), dtype=np.float32) * 0.1, -1.0, 1.0)
    g2 = np.random.default_rng(4321 + rank)
    toks = g2.integers(17, 16000, size=(B, wl["tokens"]), dtype=np.int32)
    toks[:, 0], toks[:, -1] = 2, 3
    n = np.full((B,), audio.shape[1], np.int32)
    return audio, n, toks


def build_trainer(wl, strategy=None, use_graph=True):
    from speech_recognition_amd import ops
    from speech_recognition_amd.configs import get_model_config
    from speech_recognition_amd.training import TrainStep
    from speech_recognition_amd.utils import LRScheduler
    dc = load_yaml("libri_config.yml")
    sa = dict(dc["spec_augment"], enable=True)     # 8d: SpecAugment on with the shipped parameters
    plan = ops.LogmelPlan(dc["sample_rate"], dc["frame_length"], dc["frame_step"], dc["fft_length"], dc["num_mel_bins"],
                          dc["lower_edge_hertz"], dc["upper_edge_hertz"], use_delta=dc["use_delta_accelerate"], spec_augment=sa)
    model = get_model_config(os.path.join(ROOT, "resources", "configs", wl["model"])).create_model(seed=1234)
    sched = LRScheduler(total_steps=100000, max_learning_rate=2e-4, min_learning_rate=1e-5)
    return TrainStep(model, sched, frontend=plan, strategy=strategy, use_graph=use_graph), model


def host_threads():
    """Threads for the CPU baseline: the cores this process may run on, capped at the GPU box's share (16 per
    GPU) - os.cpu_count() reports the whole host and oversubscribing it stalls the intra-op thread pool."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def cpu_baseline(budget_s=25.0):
    """The oracle (oracle/: CPU restatement of the reference, torch-CPU fp32) timed on a bounded sample
    of the headline workload: ONE las_small training step (features -> forward -> loss -> backward -> Adam)
    on a reduced batch.  Reported baseline only (kind "port": TensorFlow, the reference's engine, is not
    installed and cannot be)."""
    from oracle import features as OF
    from oracle import las as OLAS
    from oracle import measure as OM
    threads = host_threads()
    torch.set_num_threads(threads)
    wl = WORKLOADS["las_small"]
    mc, dc = load_yaml("las_small.yml"), load_yaml("libri_config.yml")
    g = torch.Generator().manual_seed(0)
    shapes = OLAS.param_shapes(mc)
    params = {}
    for k, s in shapes.items():
        if k.endswith(("gamma", "moving_variance")):
            params[k] = torch.ones(s)
        elif k.endswith(("bias", "beta", "moving_mean")):
            params[k] = torch.zeros(s)
        else:
            params[k] = torch.randn(s, generator=g) * 0.05
    train = {k: v.requires_grad_(True) for k, v in params.items() if not k.endswith(("moving_mean", "moving_variance"))}
    m = {k: torch.zeros_like(v) for k, v in train.items()}
    vv = {k: torch.zeros_like(v) for k, v in train.items()}

    def one_step(audio, n, toks, spec_aug):
        for v in train.values():
            v.grad = None
        t0 = time.perf_counter()
        feats = torch.from_numpy(OF.batch_features(audio.astype(np.float64), n, dc, seed=1, spec_aug=spec_aug).astype(np.float32))
        tk = torch.from_numpy(toks)
        logits = OLAS.las_forward(params, mc, feats, tk[:, :-1], training=True, seed=1, use_teacher_forcing=True)
        loss = OM.sparse_categorical_crossentropy(tk[:, 1:], logits, 0)
        loss.backward()
        with torch.no_grad():
            OM.adam_step({k: v for k, v in train.items()}, {k: v.grad for k, v in train.items()}, m, vv, 0, 2e-4)
        return time.perf_counter() - t0

    B = 4
    audio, n, toks = synthetic_batch(0, wl, B)
    sa = {k: dc["spec_augment"][k] for k in ("F", "m_F", "T", "p", "m_T")}
    dt = one_step(audio, n, toks, sa)
    out = {"value": round(B * wl["clip_seconds"] / dt, 3), "unit": "audio-s/s", "cores": threads, "kind": "port",
           "sample": f"1 las_small training step (front end+fwd+bwd+Adam), batch {B} x 10 s clips, torch-CPU fp32 oracle, {dt:.1f} s"}
    # BASELINE.json configs[0] (the reference's own CPU-runnable case): las_small + libri_config on the two-clip
    # tests/data/wav_dataset.tsv, batch 2 - the same restatement on the reference's fixture (66150 samples read at
    # 16 kHz = 4.13 s per clip, SpecAugment off as shipped)
    try:
        from speech_recognition_amd.data import SentencePieceTokenizer, get_dataset
        fix = os.path.join(ROOT, "tests", "golden", "reference_fixtures")
        tok = SentencePieceTokenizer(os.path.join(fix, "sp_model_unigram_16K_libri.model"))
        ex = list(get_dataset(os.path.join(fix, "wav_dataset.tsv"), "wav", 16000, tok))
        L = max(len(t) for _, t in ex)
        a2 = np.stack([a for a, _ in ex])
        t2 = np.stack([np.pad(t, (0, L - len(t))) for _, t in ex]).astype(np.int32)
        dt2 = one_step(a2, np.full((len(ex),), a2.shape[1], np.int32), t2, None)
        secs = a2.shape[0] * a2.shape[1] / 16000.0
        out["reference_config"] = {"workload": "BASELINE configs[0]: las_small + libri_config on wav_dataset.tsv, batch 2, one training step",
                                   "value": round(secs / dt2, 3), "unit": "audio-s/s", "seconds": round(dt2, 2)}
    except Exception as e:
        out["reference_config"] = {"workload": "BASELINE configs[0]", "value": None, "error": str(e)}
    return out


def time_kernel(stream, fn, iters=20):
    """Average duration (s) of fn() launched back to back on `stream`, by HIP events on that stream."""
    with torch.cuda.stream(stream):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(iters):
            fn()
        e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def kernel_rooflines(trainer, model, audio_d, n_d):
    """Per-kernel rooflines of the las_small step: the three heaviest non-recurrent kernels and the persistent recurrent
    sweep, each timed alone on the trainer's stream: the encoder input-projection GEMM (MFMA bound), the vocabulary GEMM (MFMA bound) and the
    fused front end (HBM bound, algorithmic bytes = SURVEY.md 8d: 160 KB per audio-second)."""
    from speech_recognition_amd import ops
    out = []
    B, He, Hd, V = audio_d.shape[0], model.He, model.Hd, model.V
    c = next(iter(trainer._shapes.values()))
    ws = c["ws"]
    M = B * ws.T2
    a = torch.randn(M, 2 * He, device="cuda")
    w = torch.randn(2 * He, 8 * He, device="cuda") * 0.05     # both directions' [Din, 4H] kernels side by side
    y = torch.empty(M, 8 * He, device="cuda")
    t = time_kernel(trainer.stream, lambda: ops.gemm(a, w, y))
    fl = 2.0 * M * 2 * He * 8 * He
    out.append({"kernel": f"gemm_f32 encoder input projection [{M}x{2 * He}]x[{2 * He}x{8 * He}]", "bound": "mfma",
                "achieved": round(fl / t / 1e12, 2), "peak": PEAK_F32_MFMA / 1e12, "unit": "TFLOP/s", "frac": round(fl / t / PEAK_F32_MFMA, 4),
                "us": round(t * 1e6, 1)})
    U = ws.U
    yd = torch.randn(U * B, Hd, device="cuda")
    wv = torch.randn(Hd, V, device="cuda") * 0.05
    lg = torch.empty(U * B, V, device="cuda")
    t = time_kernel(trainer.stream, lambda: ops.gemm(yd, wv, lg))
    fl = 2.0 * U * B * Hd * V
    out.append({"kernel": f"gemm_f32 vocabulary projection [{U * B}x{Hd}]x[{Hd}x{V}]", "bound": "mfma",
                "achieved": round(fl / t / 1e12, 2), "peak": PEAK_F32_MFMA / 1e12, "unit": "TFLOP/s", "frac": round(fl / t / PEAK_F32_MFMA, 4),
                "us": round(t * 1e6, 1)})
    fe = trainer.frontend
    feats = c["feats"]
    t = time_kernel(trainer.stream, lambda: fe(audio_d, n_d, feats.shape[1], seed=model.seed, out=feats))
    byts = audio_d.numel() * 4 + feats.numel() * 4
    out.append({"kernel": "logmel_kernel (log-mel + SpecAugment + delta, fused)", "bound": "hbm", "achieved": round(byts / t / 1e9, 1),
                "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": round(byts / t / PEAK_HBM, 4), "us": round(t * 1e6, 1)})
    # the dominant kernel of the step: the persistent recurrent sweep (one launch per encoder layer).  It is bound by the
    # latency of its T' dependent hand-offs, not by the matrix pipe: report both views
    try:
        from speech_recognition_amd import layers as _layers
        buf = ws.layers[1]["rnn"]
        if _layers.PERSISTENT_RNN and "persist_ws" in buf:
            T2 = ws.T2
            t = time_kernel(trainer.stream, lambda: ops.rnn_seq_fwd_persist(buf["seq"], buf["persist_ws"]), iters=5)
            fl = 2.0 * B * T2 * He * 4 * He * 2
            out.append({"kernel": f"rnn_seq_fwd_persist_kernel (BiLSTM layer, H={He}, B={B}, {T2} dependent steps in one launch)",
                        "bound": "latency (reported against mfma)", "achieved": round(fl / t / 1e12, 2), "peak": PEAK_F32_MFMA / 1e12,
                        "unit": "TFLOP/s", "frac": round(fl / t / PEAK_F32_MFMA, 4), "us": round(t * 1e6, 1),
                        "us_per_dependent_step": round(t * 1e6 / T2, 2)})
    except Exception as e:   # never take the measured line down
        out.append({"kernel": "rnn_seq_fwd_persist_kernel", "error": str(e)})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="las_small")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", choices=["f32", "bf16"], default=None,
                    help="default: f32 for las_small / deepspeech (the headline dtype), bf16 mixed precision for las_large (BASELINE configs[4])")
    args = ap.parse_args()
    wl = WORKLOADS[args.workload]
    precision = args.precision or wl.get("precision", "f32")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    # The CPU baseline runs first, before this process touches the GPU, under an alarm: it is a reported
    # side number and must never hold up or take down the measured line.
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        import signal

        def _too_slow(signum, frame):
            raise TimeoutError("CPU baseline exceeded 150 s")
        signal.signal(signal.SIGALRM, _too_slow)
        signal.alarm(150)
        try:
            cpu = cpu_baseline()
        except BaseException as e:
            cpu = {"value": None, "unit": "audio-s/s", "cores": host_threads(), "kind": "port", "sample": f"failed: {e}"}
        finally:
            signal.alarm(0)
    torch.cuda.set_device(local)
    strategy = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        from speech_recognition_amd.utils import DeviceStrategy
        strategy = DeviceStrategy(torch.device("cuda", local), world, rank)

    audio, n, toks = synthetic_batch(rank, wl)
    audio_d, n_d, toks_d = torch.from_numpy(audio).cuda(), torch.from_numpy(n).cuda(), torch.from_numpy(toks).cuda()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def measure():
        trainer, model = build_trainer(wl, strategy, use_graph=not args.no_graph)
        for _ in range(max(args.warmup, 3)):      # >= 3: eager warm-up, graph capture, first replay
            ws = trainer.step(audio_d, n_d, toks_d, use_teacher_forcing=True)
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(trainer.stream)
        for _ in range(args.steps):
            ws = trainer.step(audio_d, n_d, toks_d, use_teacher_forcing=True)
        ev1.record(trainer.stream)
        barrier()
        dt = time.perf_counter() - t0
        failed = 0.0
        try:
            stats = trainer.read_stats(ws)
        except RuntimeError as e:                  # a persistent-kernel hand-off timed out on this rank
            print(f"[bench] rank {rank}: {e}", file=sys.stderr)
            stats, failed = [float("nan")] * 3, 1.0
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([dt, failed], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt, failed = float(t[0]), float(t[1])
        return trainer, model, dt, ev0.elapsed_time(ev1) / args.steps, stats, failed > 0

    from speech_recognition_amd import layers as _layers
    from speech_recognition_amd import ops as _ops
    _ops.set_mixed_precision(precision == "bf16")
    trainer, model, dt, dev_ms, (loss, correct, kept), failed = measure()
    if failed and _layers.PERSISTENT_RNN:
        # never report a step whose results are invalid: redo the whole measurement on the per-step recurrent
        # kernels (every rank takes this branch together - the flag was all-reduced)
        _layers.PERSISTENT_RNN = False
        del trainer, model
        torch.cuda.empty_cache()
        trainer, model, dt, dev_ms, (loss, correct, kept), failed = measure()
    if failed:
        raise SystemExit("bench: the training step reported an error on both recurrent paths")
    if rank != 0:
        return
    ms = dt / args.steps * 1e3
    value = world * wl["batch"] * wl["clip_seconds"] * args.steps / dt
    achieved = wl["flops"] / (dev_ms * 1e-3)
    peak = PEAK_BF16_MFMA if precision == "bf16" else PEAK_F32_MFMA
    out = {
        "metric": wl["metric"], "value": round(value, 1), "unit": "audio-s/s",
        "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 3), "ms_per_step": round(ms, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": precision, "data": "synthetic",
        "config": {"workload": wl["text"] + "; " + PRECISION_TEXT[precision], "global_batch": wl["batch"] * world, "clip_seconds": wl["clip_seconds"],
                   "parallelism": f"dp{world}", "hip_graph": not args.no_graph, "persistent_rnn": bool(_layers.PERSISTENT_RNN),
                   "final_loss": round(loss, 4)},
        "roofline": {"bound": "mfma", "achieved": round(achieved / 1e12, 3), "peak": peak / 1e12, "unit": "TFLOP/s",
                     "frac": round(achieved / peak, 4), "traffic": wl.get("traffic") if precision == "f32" else None,
                     "kernel": f"whole training step (algorithmic {wl['flops'] / 1e9:.1f} GFLOP/step, SURVEY.md 8d) over HIP-event step time"},
    }
    if args.workload == "las_small" and world == 1:
        try:
            out["roofline"]["kernels"] = kernel_rooflines(trainer, model, audio_d, n_d)
        except Exception as e:  # per-kernel extras must never take the measured line down
            out["roofline"]["kernels"] = f"failed: {e}"
    if cpu is not None:
        out["cpu_baseline"] = cpu
    print(json.dumps(out))


if __name__ == "__main__":
    main()
