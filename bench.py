#!/usr/bin/env python3
"""Headline benchmark: audio-seconds/second of LAS training (BASELINE.json metric) on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W]            (N=1: plain python)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = SURVEY.md 8d config 2: las_small.yml + libri_config.yml, synthetic 10 s / 16 kHz clips,
batch 32 per GPU, 65-token rows (64 decoder steps), SpecAugment (F=27, m_F=2, T=100, p=1.0, m_T=2) and
delta features computed ON the GPU inside the step, dropout 0.15 active, teacher forcing on,
forward + backward + (RCCL gradient all-reduce) + Adam(lr 2e-4, LRScheduler).  fp32 throughout.
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

CLIP_SECONDS, SAMPLE_RATE, BATCH, TOKENS = 10.0, 16000, 32, 65
# SURVEY.md 8d / BASELINE.md 4: algorithmic training flops per step (3 x forward, key projection counted once)
ALGO_FLOPS_PER_STEP = 363.3e9
PEAK_F32_MFMA = 157.3e12   # MI355X_MICROARCH.md: dense f32-input MFMA peak


def load_yaml(name):
    import yaml
    with open(os.path.join(ROOT, "resources", "configs", name)) as f:
        return yaml.safe_load(f)


def synthetic_batch(rank, B=BATCH):
    """N(0, 0.1^2) clipped to [-1,1], seed 1234(+rank); tokens [2, 63 x U{17..15999}, 3], seed 4321."""
    g = np.random.default_rng(1234 + rank)
    audio = np.clip(g.standard_normal((B, int(CLIP_SECONDS * SAMPLE_RATE)), dtype=np.float32) * 0.1, -1.0, 1.0)
    g2 = np.random.default_rng(4321 + rank)
    toks = g2.integers(17, 16000, size=(B, TOKENS), dtype=np.int32)
    toks[:, 0], toks[:, -1] = 2, 3
    n = np.full((B,), audio.shape[1], np.int32)
    return audio, n, toks


def build_trainer(strategy=None, use_graph=True):
    from speech_recognition_amd import ops
    from speech_recognition_amd.models import LAS
    from speech_recognition_amd.training import TrainStep
    from speech_recognition_amd.utils import LRScheduler
    mc, dc = load_yaml("las_small.yml"), load_yaml("libri_config.yml")
    sa = dict(dc["spec_augment"], enable=True)     # 8d: SpecAugment on with the shipped parameters
    plan = ops.LogmelPlan(dc["sample_rate"], dc["frame_length"], dc["frame_step"], dc["fft_length"], dc["num_mel_bins"],
                          dc["lower_edge_hertz"], dc["upper_edge_hertz"], use_delta=dc["use_delta_accelerate"], spec_augment=sa)
    model = LAS(mc["rnn_type"], mc["vocab_size"], mc["encoder_hidden_dim"], mc["decoder_hidden_dim"], mc["num_encoder_layers"],
                mc["num_decoder_layers"], mc["dropout"], mc["teacher_forcing_rate"], mc["pad_id"], seed=1234)
    sched = LRScheduler(total_steps=100000, max_learning_rate=2e-4, min_learning_rate=1e-5)
    return TrainStep(model, sched, frontend=plan, strategy=strategy, use_graph=use_graph), model


def cpu_baseline(budget_s=25.0):
    """The oracle (oracle/: CPU restatement of the reference, torch-CPU fp32) timed on a bounded sample
    of the same workload: ONE las_small training step (features -> forward -> loss -> backward -> Adam)
    on a reduced batch.  Reported baseline only (kind "port": TensorFlow, the reference's engine, is not
    installed and cannot be)."""
    from oracle import features as OF
    from oracle import las as OLAS
    from oracle import measure as OM
    torch.set_num_threads(os.cpu_count() or 1)
    mc, dc = load_yaml("las_small.yml"), load_yaml("libri_config.yml")
    B = 4
    audio, n, toks = synthetic_batch(0, B)
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
    t0 = time.perf_counter()
    sa = {k: dc["spec_augment"][k] for k in ("F", "m_F", "T", "p", "m_T")}
    feats = torch.from_numpy(OF.batch_features(audio.astype(np.float64), n, dc, seed=1, spec_aug=sa).astype(np.float32))
    tk = torch.from_numpy(toks)
    logits = OLAS.las_forward(params, mc, feats, tk[:, :-1], training=True, seed=1, use_teacher_forcing=True)
    loss = OM.sparse_categorical_crossentropy(tk[:, 1:], logits, 0)
    loss.backward()
    with torch.no_grad():
        OM.adam_step({k: v for k, v in train.items()}, {k: v.grad for k, v in train.items()}, m, vv, 0, 2e-4)
    dt = time.perf_counter() - t0
    return {"value": round(B * CLIP_SECONDS / dt, 3), "unit": "audio-s/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"1 las_small training step (front end+fwd+bwd+Adam), batch {B} x 10 s clips, torch-CPU fp32 oracle, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    torch.cuda.set_device(local)
    strategy = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        from speech_recognition_amd.utils import DeviceStrategy
        strategy = DeviceStrategy(torch.device("cuda", local), world, rank)

    trainer, model = build_trainer(strategy, use_graph=not args.no_graph)
    audio, n, toks = synthetic_batch(rank)
    audio_d, n_d, toks_d = torch.from_numpy(audio).cuda(), torch.from_numpy(n).cuda(), torch.from_numpy(toks).cuda()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

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
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    loss, correct, kept = trainer.read_stats(ws)
    if rank != 0:
        return
    ms = dt / args.steps * 1e3
    value = world * BATCH * CLIP_SECONDS * args.steps / dt
    dev_ms = ev0.elapsed_time(ev1) / args.steps
    achieved = ALGO_FLOPS_PER_STEP / (dev_ms * 1e-3)
    out = {
        "metric": "audio-seconds/sec training (las_small, 10s clips, bs32)", "value": round(value, 1), "unit": "audio-s/s",
        "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 3), "ms_per_step": round(ms, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "las_small.yml + libri_config.yml, synthetic 10 s 16 kHz clips, batch 32 per GPU, 64 decoder steps, "
                               "SpecAugment+delta on GPU, dropout 0.15, teacher forcing on, fwd+bwd+Adam(lr 2e-4)",
                   "global_batch": BATCH * world, "clip_seconds": CLIP_SECONDS, "parallelism": f"dp{world}",
                   "hip_graph": not args.no_graph, "final_loss": round(loss, 4)},
        "roofline": {"bound": "mfma", "achieved": round(achieved / 1e12, 3), "peak": PEAK_F32_MFMA / 1e12, "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_F32_MFMA, 4), "traffic": None,
                     "kernel": "whole training step (algorithmic 363.3 GFLOP/step, SURVEY.md 8d) over HIP-event step time"},
    }
    if not args.no_cpu_baseline and world == 1:
        try:
            out["cpu_baseline"] = cpu_baseline()
        except Exception as e:  # the baseline must never take the measured line down
            out["cpu_baseline"] = {"value": None, "unit": "audio-s/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
