#!/usr/bin/env python3
"""Headline benchmark: audio-seconds/second of training (BASELINE.json metric) on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload las_small|deepspeech|las_large]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default workload = SURVEY.md 8d config 2 (the one BASELINE.json's metric is quoted on): las_small.yml +
libri_config.yml, synthetic 10 s / 16 kHz clips, batch 32 per GPU, 65-token rows (64 decoder steps),
SpecAugment (F=27, m_F=2, T=100, p=1.0, m_T=2) and delta features computed ON the GPU inside the step,
dropout 0.15 active, teacher forcing on, forward + backward + (RCCL gradient all-reduce) + Adam(lr 2e-4,
LRScheduler).  fp32 throughout.  --workload deepspeech / las_large run SURVEY.md 8d configs 4 / 5 the
same way (las_large defaults to --precision bf16: mixed precision, BASELINE configs[4]; --precision f32 for comparison).
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
    # traffic: memory-side bytes per step from the PMC passes of profiles/r04_las_small_pmc_hbm_traffic.txt (MiB per step there:
    # FETCH_SIZE 3389.5, WRITE_SIZE 3862.6; 2 x FETCH_SIZE per the gfx950 correction for wide reads + WRITE_SIZE) - an upper estimate,
    # measured offline; dominant_traffic: the same for one launch of rnn_sweep_bwd_kernel (FETCH 410.86 / 3, WRITE 931.36 / 3 MiB: until the rows of its workgroup squares sat on one XCD each, round 4, FETCH was 3215 - eight L2s fetching the same coefficient packs); most of the writes are
    # the inter-workgroup exchange of the one-launch sweeps (write-through stores), which the counters
    # tally at the fabric although the Infinity Cache serves it
    # algorithmic_bytes: activations kept for backward written once + read once (2 x 0.9 GB), logits 131 MB x 5 touches,
    # Adam 28 B x 16 M parameters, features 51 MB, every weight matrix read twice (forward product, input gradient)
    "las_small": dict(model="las_small.yml", clip_seconds=10.0, batch=32, tokens=65, flops=363.3e9, traffic=(2 * 3389.5 + 3862.6) * 2 ** 20, algorithmic_bytes=3.1e9,
                      traffic_source="offline rocprofv3 PMC passes, profiles/r04_las_small_pmc_hbm_traffic.txt (not measured in this run)",
                      dominant_traffic=(2 * 410.86 + 931.36) / 3 * 2 ** 20,
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
F32_PRODUCTS_TEXT = {"mfma": "f32 MFMA (v_mfma_f32_32x32x2_f32)",
                     "split9": "every f32 product as the nine bf16 pair products of exact three-way operand splits on the bf16 MFMA (2^-32 per product), f32 accumulation",
                     "split6": "every f32 product as six bf16 pair products of exact three-way operand splits on the bf16 MFMA (the pairs of weight <= 2^-24 left out: "
                               "<= 3 * 2^-24 |a b| per product; measured error against float64 below the f32 MFMA's, tests/test_gemm_gpu.py), f32 accumulation"}
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


def build_trainer(wl, strategy=None, use_graph=True, force_dp_path=False):
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
    return TrainStep(model, sched, frontend=plan, strategy=strategy, use_graph=use_graph, force_dp_path=force_dp_path), model


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

    # the headline's own batch (32 x 10 s clips): one untimed warm-up step (thread pool, allocator, autograd graph caches), then two
    # timed steps - about 20 s of CPU work on the GPU box's 16 host threads (round 3 sampled batch 8).  A slow
    # host (warm-up beyond half the budget) falls back to batch 8 and says so in `sample`
    B, timed = wl["batch"], 2
    audio, n, toks = synthetic_batch(0, wl, B)
    sa = {k: dc["spec_augment"][k] for k in ("F", "m_F", "T", "p", "m_T")}
    warm = one_step(audio, n, toks, sa)
    if warm > budget_s / 2:
        B, timed = 8, 1
        audio, n, toks = synthetic_batch(0, wl, B)
    dts = [one_step(audio, n, toks, sa) for _ in range(timed)]
    dt = sum(dts) / len(dts)
    out = {"value": round(B * wl["clip_seconds"] / dt, 3), "unit": "audio-s/s", "cores": threads, "kind": "port",
           "sample": f"{timed} las_small training steps (front end+fwd+bwd+Adam) after 1 warm-up step ({warm:.1f} s at batch {wl['batch']}), batch {B} of the "
                     f"headline's 32 x 10 s clips, torch-CPU fp32 oracle, {dt:.2f} s/step (min {min(dts):.2f}, max {max(dts):.2f})"}
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


def _mfma_entry(name, flops, t, peak, **extra):
    return dict({"kernel": name, "bound": "mfma", "achieved": round(flops / t / 1e12, 2), "peak": peak / 1e12, "unit": "TFLOP/s",
                 "frac": round(flops / t / peak, 4), "us": round(t * 1e6, 1)}, **extra)


def _hbm_entry(name, byts, t, **extra):
    return dict({"kernel": name, "bound": "hbm", "achieved": round(byts / t / 1e9, 1), "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                 "frac": round(byts / t / PEAK_HBM, 4), "algorithmic_bytes": int(byts), "us": round(t * 1e6, 1)}, **extra)


def _sweep_entries(trainer, model, buf, layer, rt, B, T2, H, peak):
    """The one-launch recurrent sweeps of one bidirectional layer, timed alone on the trainer's stream.  They are bound by
    the latency of their T' dependent hand-offs, not by a pipe: the entry gives the microseconds per dependent
    step next to the (small) fraction of the f32 MFMA peak the recurrent products reach."""
    from speech_recognition_amd import ops
    out = []
    ng = 4 if rt == "lstm" else (3 if rt == "gru" else 1)
    err = getattr(model.store, "err_flag", None)
    if "persist_ws" in buf:
        t = time_kernel(trainer.stream, lambda: ops.rnn_seq_fwd_persist(buf["seq"], buf["persist_ws"], err), iters=5)
        fl = 2.0 * B * T2 * H * ng * H * 2
        out.append(_mfma_entry(f"rnn_sweep_fwd_kernel (Bi{rt.upper()} layer, H={H}, B={B}, {T2} dependent steps in one launch)", fl, t,
                               PEAK_F32_MFMA, bound="latency (reported against mfma)", us_per_dependent_step=round(t * 1e6 / T2, 2)))
    if "persist_bwd_ws" in buf:
        dirs = buf["dirs"]
        dy = torch.randn(B, T2, 2 * H, device="cuda") * 1e-3
        zeros = [torch.zeros(B, H, device="cuda") for _ in range(2)]
        dcs = [torch.zeros(B, H, device="cuda") for _ in range(2)]
        gds = [dict(dh_last=None, dc=dcs[d] if rt == "lstm" else None, dy_carry=dd["dy_carry"] if buf["mask"] is not None else None,
                    direct=dd["direct"], dh0=dd["dh0"], ds=dd["ds"]) for d, dd in enumerate(dirs)]

        def restore():                                            # (the sweep leaves the saved activations alone: ds goes to its own buffer)
            for dd, dc in zip(dirs, dcs):
                dc.copy_(zeros[0])
                if buf["mask"] is not None:
                    ops.fill(dd["dy_carry"], 0.0)

        def bwd():
            restore()
            ops.rnn_seq_bwd(buf["seq"], dy, gds, buf["persist_bwd_ws"], err)

        t = time_kernel(trainer.stream, bwd, iters=5) - time_kernel(trainer.stream, restore, iters=5)
        fl = 2.0 * B * T2 * H * ng * H * 2
        out.append(_mfma_entry(f"rnn_sweep_bwd_kernel (Bi{rt.upper()} layer {layer}, H={H}, B={B}, {T2} dependent steps in one launch)", fl, t,
                               PEAK_F32_MFMA, bound="latency (reported against mfma)", us_per_dependent_step=round(t * 1e6 / T2, 2)))
    # the wide layers' sweeps under mixed precision (las_large): bound by the 128 KB every workgroup gathers per dependent step
    # (256 workgroups x B x H bf16 through the fabric) - reported against the bf16 MFMA peak like the others, with the
    # exchange bandwidth they reach next to it
    if "wide_ws" in buf:
        t = time_kernel(trainer.stream, lambda: ops.rnn_sweep_wide_fwd(buf["seq"], buf["wide_ws"], err), iters=3)
        fl = 2.0 * B * T2 * H * ng * H * 2
        out.append(_mfma_entry(f"rnn_sweepw_fwd_kernel (BiLSTM layer, H={H}, B={B}, {T2} dependent steps in one launch, bf16 weights resident)", fl, t,
                               PEAK_BF16_MFMA, bound="latency / exchange (reported against mfma)", us_per_dependent_step=round(t * 1e6 / T2, 2),
                               exchange_GBps=round(2 * (H // 8) * 64 * H * 2 * T2 / t / 1e9, 1)))
    if "wide_bwd_ws" in buf:
        dirs = buf["dirs"]
        dy = torch.randn(B, T2, 2 * H, device="cuda") * 1e-3
        dcs = [torch.zeros(B, H, device="cuda") for _ in range(2)]
        gds = [dict(dh_last=None, dc=dcs[d], dy_carry=dd["dy_carry"] if buf["mask"] is not None else None, direct=dd["direct"], dh0=dd["dh0"],
                    ds=dd["ds"]) for d, dd in enumerate(dirs)]

        def wbwd():
            for dc in dcs:
                dc.zero_()
            ops.rnn_sweep_wide_bwd(buf["seq"], dy, gds, buf["wide_bwd_ws"], err)

        t = time_kernel(trainer.stream, wbwd, iters=3)
        fl = 2.0 * B * T2 * H * ng * H * 2
        out.append(_mfma_entry(f"rnn_sweepw_bwd_kernel (BiLSTM layer {layer}, H={H}, B={B}, {T2} dependent steps in one launch, 32x4 grid per "
                               "direction, bf16 partial sums)", fl, t, PEAK_BF16_MFMA, bound="latency / exchange (reported against mfma)",
                               us_per_dependent_step=round(t * 1e6 / T2, 2), exchange_GBps=round(256 * 64 * H * 2 * T2 / t / 1e9, 1)))
    return out


def kernel_rooflines(trainer, model, audio_d, n_d, precision):
    """Per-kernel rooflines of the step, each kernel timed alone on the trainer's stream by HIP events: the heaviest dense
    products (MFMA bound; bf16 peak under mixed precision), the fused front end (HBM bound, algorithmic bytes = SURVEY.md 8d:
    160 KB per audio-second = audio in + features out), the DeepSpeech2 convolution, and the one-launch sweeps (latency bound)."""
    from speech_recognition_amd import layers as _layers
    from speech_recognition_amd import ops
    out = []
    peak = PEAK_BF16_MFMA if precision == "bf16" else PEAK_F32_MFMA
    gk = "gemm_bf16" if precision == "bf16" else "gemm_f32"
    c = next(iter(trainer._shapes.values()))
    ws = c["ws"]
    B = audio_d.shape[0]
    is_las = hasattr(model, "Hd")

    def gemm_entry(label, M, K, N):
        a = torch.randn(M, K, device="cuda")
        w = torch.randn(K, N, device="cuda") * 0.05
        y = torch.empty(M, N, device="cuda")
        t = time_kernel(trainer.stream, lambda: ops.gemm(a, w, y))
        extra = {}
        if precision != "bf16" and ops.f32_gemm_mode() != "mfma":
            # the f32 products run on the bf16 matrix pipe as 9 / 6 pair products each: `frac` stays the algorithmic f32 flops against the
            # f32 MFMA peak (what the step needs, against what the f32 pipe would give); bf16_pipe_frac prices the executed pair products
            pairs = 9 if ops.f32_gemm_mode() == "split9" else 6
            extra = {"f32_products": ops.f32_gemm_mode(), "bf16_pipe_frac": round(pairs * 2.0 * M * K * N / t / PEAK_BF16_MFMA, 4)}
        out.append(_mfma_entry(f"{gk} {label} [{M}x{K}]x[{K}x{N}]", 2.0 * M * K * N, t, peak, **extra))

    if is_las:
        He, Hd, V, T2 = model.He, model.Hd, model.V, ws.T2
        ng = 4 if model.rt == "lstm" else (3 if model.rt == "gru" else 1)
        gemm_entry("encoder input projection", B * T2, 2 * He, 2 * ng * He)   # both directions' [Din, 4H] kernels side by side
        gemm_entry("vocabulary projection", ws.U * B, Hd, V)
        rt, H, lbuf, layer = model.rt, He, ws.layers[1]["rnn"], 1
    else:
        H, T2 = model.H, ws.T2
        rt = model.rt
        ng = 4 if rt == "lstm" else (3 if rt == "gru" else 1)
        gemm_entry("recurrent layer input projection", B * T2, 2 * H, 2 * ng * H)
        gemm_entry("character projection", B * T2, 2 * H, model.V)
        lbuf, layer = ws.layers[1]["rnn"], 1
        # the convolution stack runs on the f32 MFMA in both precisions (implicit GEMM, conv.hip): layer 1 is the heavy one
        i = 1
        x, y = ws.conv[i - 1], ws.conv[i]
        wk = model.store.p[f"convolution/conv_layers/{i}/kernel"]
        bk = model.store.p[f"convolution/conv_layers/{i}/bias"]
        fl = 2.0 * y.numel() * wk.shape[0] * wk.shape[1] * wk.shape[2]
        dwk, dxk = torch.zeros_like(wk), torch.empty_like(x)
        dyk = torch.randn_like(y)
        shape = f"[{'x'.join(map(str, x.shape))}] * [{'x'.join(map(str, wk.shape))}] stride {tuple(model.strides[i])}"
        for nm, fn in (("conv2d_fwd", lambda: ops.conv2d_fwd(x, wk, bk, model.strides[i], y)),
                       ("conv2d_bwd_filter", lambda: ops.conv2d_bwd_filter(x, dyk, dwk, model.strides[i])),
                       ("conv2d_bwd_data", lambda: ops.conv2d_bwd_data(dyk, wk, dxk, model.strides[i]))):
            t = time_kernel(trainer.stream, fn)
            out.append(_mfma_entry(f"{nm} layer {i} {shape}", fl, t, PEAK_F32_MFMA))
    fe = trainer.frontend
    feats = c["feats"]
    t = time_kernel(trainer.stream, lambda: fe(audio_d, n_d, feats.shape[1], seed=model.seed, out=feats))
    out.append(_hbm_entry("logmel_kernel (log-mel + SpecAugment + delta, fused)", audio_d.numel() * 4 + feats.numel() * 4, t))
    try:
        if _layers.PERSISTENT_RNN:
            out += _sweep_entries(trainer, model, lbuf, layer, rt, B, T2, H, peak)
        if is_las and getattr(ws, "dsweep_ws", None) is not None:
            t = time_kernel(trainer.stream, lambda: model._decoder_sweep(ws, True), iters=5)
            Hd, D, U = model.Hd, 2 * model.He, ws.U
            # per step: query (Hd x Hd), energies + context (2 x T2 x D... here Hd-wide keys), two LSTM cells
            fl = 2.0 * B * U * (Hd * Hd + T2 * Hd + T2 * D + (D + Hd + Hd) * 4 * Hd + 2 * Hd * 4 * Hd)
            out.append(_mfma_entry(f"decoder_sweep_fwd_kernel (attention + 2 LSTM cells, Hd={Hd}, B={B}, {U} steps x 4 dependent hand-offs "
                                   "in one launch)", fl, t, PEAK_F32_MFMA, bound="latency (reported against mfma)",
                                   us_per_dependent_step=round(t * 1e6 / U, 2)))
            if getattr(ws, "dsweep_bwd_ws", None) is not None:
                t = time_kernel(trainer.stream, lambda: model._decoder_sweep_bwd(ws), iters=5)
                out.append(_mfma_entry(f"decoder_sweep_bwd_kernel (2 LSTM cells + attention backwards, Hd={Hd}, B={B}, {U} steps x 3 dependent "
                                       "hand-offs in one launch)", 2.0 * fl, t, PEAK_F32_MFMA, bound="latency (reported against mfma)",
                                       us_per_dependent_step=round(t * 1e6 / U, 2)))
    except Exception as e:   # never take the measured line down
        out.append({"kernel": "one-launch sweeps", "error": str(e)})
    return out


def quick_workload(wname, steps, use_graph):
    """A short single-GPU measurement of another BASELINE workload (same step function, synthetic batch of ITS geometry): 3 warm-up
    steps, `steps` timed ones bracketed by synchronize()."""
    from speech_recognition_amd import ops
    wl = WORKLOADS[wname]
    precision = wl.get("precision", "f32")
    ops.set_mixed_precision(precision == "bf16")
    audio, n, toks = synthetic_batch(0, wl)
    audio_d, n_d, toks_d = torch.from_numpy(audio).cuda(), torch.from_numpy(n).cuda(), torch.from_numpy(toks).cuda()
    trainer, model = build_trainer(wl, None, use_graph=use_graph)
    for _ in range(3):
        ws = trainer.step(audio_d, n_d, toks_d, use_teacher_forcing=True)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(trainer.stream)
    for _ in range(steps):
        ws = trainer.step(audio_d, n_d, toks_d, use_teacher_forcing=True)
    ev1.record(trainer.stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stats = trainer.read_stats(ws)                 # raises if a sweep timed out: the caller records the error instead of a number
    dev_s = ev0.elapsed_time(ev1) * 1e-3 / steps
    peak = PEAK_BF16_MFMA if precision == "bf16" else PEAK_F32_MFMA
    res = {"metric": wl["metric"], "value": round(wl["batch"] * wl["clip_seconds"] * steps / dt, 1), "unit": "audio-s/s", "ms_per_step": round(dt / steps * 1e3, 3),
           "steps": steps, "dtype": precision, "workload": wl["text"] + "; " + PRECISION_TEXT[precision], "final_loss": round(stats[0], 4),
           "roofline": {"bound": "mfma", "achieved": round(wl["flops"] / dev_s / 1e12, 3), "peak": peak / 1e12, "unit": "TFLOP/s",
                        "frac": round(wl["flops"] / dev_s / peak, 4), "kernel": "whole training step"}}
    del trainer, model
    torch.cuda.empty_cache()
    return res


def dp_path_measure(wl, steps, use_graph, audio_d, n_d, toks_d, single_ms):
    """The data-parallel code path on ONE GPU: a world-size-1 RCCL group, TrainStep(force_dp_path=True) - one
    captured graph per gradient bucket's backward segment, the bucket all-reduces through torch.distributed "nccl" on the
    communication stream between them (bf16 wire under mixed precision).  Reports its ms/step next to the single-graph step's and
    how far the parameters of the two paths are apart after the same `check_steps` steps from the same seed (f32 atomics make two
    runs of the SAME path differ at the 1e-6 level: `same_path_rel_diff`)."""
    import torch.distributed as dist
    from speech_recognition_amd.utils import DeviceStrategy
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
        created = True
    try:
        strategy = DeviceStrategy(torch.device("cuda", torch.cuda.current_device()), 1, 0)

        def run(force, nsteps, timed):
            from speech_recognition_amd.training import TrainStep  # noqa: F401
            trainer, model = build_trainer(wl, strategy, use_graph=use_graph, force_dp_path=force)
            for _ in range(3):
                ws = trainer.step(audio_d, n_d, toks_d, use_teacher_forcing=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(nsteps):
                ws = trainer.step(audio_d, n_d, toks_d, use_teacher_forcing=True)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / max(nsteps, 1)
            trainer.read_stats(ws)
            flat = model.store.flat.clone()
            c = next(iter(trainer._shapes.values()))
            info = dict(graphs=len(c["graphs"]), buckets=len(model.store.bucket_ranges), wire=str(trainer.exchange.wire_dtype).replace("torch.", ""))
            del trainer, model
            torch.cuda.empty_cache()
            return dt, flat, info

        check_steps = 3
        os.environ["ASR_NATIVE_COLLECTIVE"] = "0"
        dt_dp, flat_dp_long, info = run(True, steps, True)
        _, flat_a, _ = run(False, check_steps, False)
        _, flat_b, _ = run(True, check_steps, False)
        _, flat_c, _ = run(False, check_steps, False)
        scale = float(flat_a.abs().max())
        out = {"ms_per_step": round(dt_dp * 1e3, 3), "single_graph_ms_per_step": round(single_ms, 3),
               "overhead_frac": round(dt_dp * 1e3 / single_ms - 1.0, 4), "steps": steps, "captured_graphs_per_step": info["graphs"],
               "gradient_buckets": info["buckets"], "wire_dtype": info["wire"], "collective": "torch.distributed nccl (RCCL), world size 1",
               "param_rel_diff_vs_single_graph": float((flat_a - flat_b).abs().max()) / scale,
               "same_path_rel_diff": float((flat_a - flat_c).abs().max()) / scale, "check_steps": 3 + check_steps}
        # the same step with the collectives issued by the C ABI (asr_allreduce_bucket) and captured, with the backward segments, in ONE graph
        try:
            os.environ["ASR_NATIVE_COLLECTIVE"] = "1"
            dt_nat, _, info_n = run(True, steps, True)
            _, flat_n, _ = run(True, check_steps, False)
            out["native_collective"] = {"ms_per_step": round(dt_nat * 1e3, 3), "overhead_frac": round(dt_nat * 1e3 / single_ms - 1.0, 4),
                                        "captured_graphs_per_step": info_n["graphs"], "collective": "asr_allreduce_bucket (RCCL called by libasr_mi355x.so), world size 1",
                                        "param_rel_diff_vs_single_graph": float((flat_a - flat_n).abs().max()) / scale}
        except Exception as e:
            out["native_collective"] = {"error": str(e)}
        finally:
            os.environ["ASR_NATIVE_COLLECTIVE"] = "0"
        return out
    finally:
        if created:
            dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="las_small")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-rooflines", action="store_true", help="skip the per-kernel timing loops (clean rocprof call counts)")
    ap.add_argument("--no-extra-workloads", action="store_true",
                    help="skip everything measured outside the timed region: the non-teacher-forced step, the other evaluations of an f32 product, "
                         "the deepspeech / las_large measurements the default (las_small, one GPU) run appends as extra_workloads")
    ap.add_argument("--no-dp-path", action="store_true",
                    help="skip the short measurement of the data-parallel code path (single-rank RCCL group) the default one-GPU run appends as dp_path")
    ap.add_argument("--precision", choices=["f32", "bf16"], default=None,
                    help="default: f32 for las_small / deepspeech (the headline dtype), bf16 mixed precision for las_large (BASELINE configs[4])")
    args = ap.parse_args()
    # stdout carries ONE JSON line and nothing else: RCCL prints a version banner to file descriptor 1 when its first communicator
    # comes up (and libraries may do likewise), so everything but the final print is sent to stderr
    json_out = os.fdopen(os.dup(1), "w")
    sys.stdout.flush()
    os.dup2(2, 1)
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

    sweep_errors = []      # what the sweeps recorded about every time-out of this run (goes into the JSON line: a retry is never silent)

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
        except RuntimeError as e:                  # a hand-off of a one-launch sweep timed out on this rank: keep the evidence
            print(f"[bench] rank {rank}: {e}", file=sys.stderr)
            sweep_errors.append({"rank": rank, "attempt": len(sweep_errors) + 1, "reports": getattr(e, "reports", [])})
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
    retried = 0
    if failed and _layers.PERSISTENT_RNN:
        # never report a step whose results are invalid.  The one time-out ever recorded (round 2) was the BPTT sweep's re-arm lag,
        # fixed in round 3; the safety net stays, and it is never silent: config.sweep_errors carries the diagnosis
        # record of every attempt that failed, config.remeasured how far down this ladder the run went.  Measure once more as
        # configured; a second failure redoes the whole measurement on the per-step recurrent kernels (every rank takes these
        # branches together - the flag was all-reduced)
        retried = 1
        del trainer, model
        torch.cuda.empty_cache()
        trainer, model, dt, dev_ms, (loss, correct, kept), failed = measure()
    if failed and _layers.PERSISTENT_RNN:
        retried = 2
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
    step_roof = {"bound": "mfma", "achieved": round(achieved / 1e12, 3), "peak": peak / 1e12, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                 "kernel": f"whole training step (algorithmic {wl['flops'] / 1e9:.1f} GFLOP/step, SURVEY.md 8d) over HIP-event step time"}
    if wl.get("algorithmic_bytes"):
        # the HBM view of the same step: algorithmic bytes over the step time - the step is nowhere near either roof,
        # it is bound by the dependent-step latency of its recurrent sweeps (the kernels list says where the time goes)
        step_roof["algorithmic_bytes"] = int(wl["algorithmic_bytes"])
        step_roof["hbm_frac"] = round(wl["algorithmic_bytes"] / (dev_ms * 1e-3) / PEAK_HBM, 4)
    kernels = None
    if world == 1 and not args.no_kernel_rooflines:
        try:
            kernels = kernel_rooflines(trainer, model, audio_d, n_d, precision)
        except Exception as e:  # per-kernel extras must never take the measured line down
            kernels = f"failed: {e}"
    # `roofline` = the DOMINANT kernel of the step (largest share of kernel time in profiles/: the encoder BPTT sweep, three launches
    # per las_small step), timed alone on the trainer's stream by HIP events just above; `roofline.step` = the whole step against
    # the same roof; `roofline.kernels` = every timed kernel.  traffic: memory-side bytes per LAUNCH of that kernel from the offline
    # rocprofv3 PMC passes (FETCH_SIZE x 2 per the gfx950 correction + WRITE_SIZE), not measured in this run: traffic_source says where
    roof = dict(step_roof)
    dom = next((k for k in kernels if isinstance(k, dict) and k.get("kernel", "").startswith("rnn_sweep_bwd_kernel")), None) if isinstance(kernels, list) else None
    if dom is not None:
        roof = {"bound": "mfma", "achieved": dom["achieved"], "peak": dom["peak"], "unit": dom["unit"], "frac": dom["frac"], "kernel": dom["kernel"],
                "us_per_launch": dom["us"], "us_per_dependent_step": dom.get("us_per_dependent_step"),
                "traffic": wl.get("dominant_traffic") if precision == "f32" else None, "traffic_source": wl.get("traffic_source"),
                "note": "latency-bound: one launch = T' dependent steps of ~3 us; the recurrent product's flops over the launch time against the f32 MFMA peak"}
    roof["scope"] = "dominant kernel, one launch" if dom is not None else "whole training step"     # (r02 lines carried the step here)
    roof["step"] = dict(step_roof, traffic=wl.get("traffic") if precision == "f32" else None, traffic_source=wl.get("traffic_source"))
    if kernels is not None:
        roof["kernels"] = kernels
    out = {
        "metric": wl["metric"], "value": round(value, 1), "unit": "audio-s/s",
        "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 3), "ms_per_step": round(ms, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": precision, "data": "synthetic",
        "config": {"workload": wl["text"] + "; " + PRECISION_TEXT[precision], "global_batch": wl["batch"] * world, "clip_seconds": wl["clip_seconds"],
                   "parallelism": f"dp{world}", "hip_graph": not args.no_graph, "persistent_rnn": bool(_layers.PERSISTENT_RNN), "remeasured": retried,
                   "sweep_errors": sweep_errors, "overlap": bool(_layers.Overlap.enabled), "final_loss": round(loss, 4),
                   "f32_products": _ops.f32_gemm_mode() + ": " + F32_PRODUCTS_TEXT[_ops.f32_gemm_mode()] + " (dense products and convolutions; ASR_GEMM_F32=mfma|split9|split6)"},
        "roofline": roof,
    }
    if cpu is not None:
        out["cpu_baseline"] = cpu
    # the reference draws ONE teacher-forcing coin per batch (las.py:366-372, rate 0.99): 1 % of the steps feed the arg-max
    # of the previous logits back and cannot batch the embedding / vocabulary layer or take the decoder sweeps.  `value` is the
    # teacher-forced step (the path 99 % of the steps take); the off-path step is timed here, outside the timed region, and
    # `value_blended` weighs the two by the coin
    if hasattr(model, "Hd") and world == 1 and not args.no_extra_workloads:
        try:
            k2 = max(3, min(args.steps, 6))
            for _ in range(3):
                ws = trainer.step(audio_d, n_d, toks_d, use_teacher_forcing=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(k2):
                ws = trainer.step(audio_d, n_d, toks_d, use_teacher_forcing=False)
            torch.cuda.synchronize()
            ms_off = (time.perf_counter() - t0) / k2 * 1e3
            trainer.read_stats(ws)
            rate = float(getattr(model, "teacher_forcing_rate", 0.99))
            blended = rate * ms + (1.0 - rate) * ms_off
            out["non_teacher_forced"] = {"ms_per_step": round(ms_off, 3), "steps": k2, "teacher_forcing_rate": rate,
                                         "note": "arg-max feedback: per-step decoder kernels, Dense(V) per step (las.py:372)"}
            out["value_blended"] = round(wl["batch"] * wl["clip_seconds"] / (blended * 1e-3), 1)
        except Exception as e:
            out["non_teacher_forced"] = {"error": str(e)}
    # the same step under the other evaluations of an f32 product (short runs, OUTSIDE the timed region): what the choice is worth
    if world == 1 and precision == "f32" and not args.no_extra_workloads:
        try:
            del trainer, model
        except NameError:
            pass
        torch.cuda.empty_cache()
        modes = {}
        cur = _ops.f32_gemm_mode()
        for mname in ("mfma", "split9", "split6"):                 # (the current mode too: all three losses after the SAME 13 steps)
            try:
                _ops.set_f32_gemm_mode(mname)
                tr2, m2 = build_trainer(wl, None, use_graph=not args.no_graph)
                for _ in range(3):
                    ws2 = tr2.step(audio_d, n_d, toks_d, use_teacher_forcing=True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    ws2 = tr2.step(audio_d, n_d, toks_d, use_teacher_forcing=True)
                torch.cuda.synchronize()
                dt2 = (time.perf_counter() - t0) / 10
                st2 = tr2.read_stats(ws2)
                modes[mname] = {"ms_per_step": round(dt2 * 1e3, 3), "loss_after_13_steps": round(st2[0], 5)}
                del tr2, m2
                torch.cuda.empty_cache()
            except Exception as e:
                modes[mname] = {"error": str(e)}
            finally:
                _ops.set_f32_gemm_mode(cur)
        out["f32_product_modes"] = modes
    # the data-parallel step of a replica (segmented graphs + bucket all-reduces through RCCL) on this one GPU, OUTSIDE the timed region
    if world == 1 and not args.no_dp_path:
        try:
            del trainer, model
        except NameError:
            pass
        torch.cuda.empty_cache()
        try:
            out["dp_path"] = dp_path_measure(wl, 10, not args.no_graph, audio_d, n_d, toks_d, ms)
        except Exception as e:
            out["dp_path"] = {"error": str(e)}
    # BASELINE configs[3] / [4] at their single-GPU geometry, measured in the same driver-visible run (short, OUTSIDE the headline's
    # timed region, after it): DeepSpeech2 f32 and las_large under mixed precision
    if args.workload == "las_small" and world == 1 and not args.no_extra_workloads:
        try:
            del trainer, model
        except NameError:
            pass
        torch.cuda.empty_cache()
        out["extra_workloads"] = []
        for wname, steps in (("deepspeech", 10), ("las_large", 4)):
            try:
                out["extra_workloads"].append(quick_workload(wname, steps, not args.no_graph))
            except Exception as e:
                out["extra_workloads"].append({"workload": wname, "error": str(e)})
        _ops.set_mixed_precision(precision == "bf16")
    json_out.write(json.dumps(out) + "\n")
    json_out.flush()


if __name__ == "__main__":
    main()
