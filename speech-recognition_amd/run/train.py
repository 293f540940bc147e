"""`python -m speech_recognition_amd.run.train` - the CLI of speech_recognition/run/train.py on MI355X.

Same flags, same config files, same outputs (train_configs.txt, data-config.yml, model-config.yml,
models/<checkpoint per epoch> + models/checkpoint, logs/train, logs/validation).  What differs is where
the work runs: the host pipeline only decodes files, tokenises and pads RAW AUDIO (or stored log-mel
frames with --use-tfrecord); log-mel, SpecAugment, delta, the model, the loss, back-propagation and
Adam all run on the GPU inside training.TrainStep.  With more than one process (torchrun) the global
batch is split across ranks and gradients are all-reduced over RCCL (MirroredStrategy, utils.py:148).

Extra flags (not in the reference): --pad-audio-multiple / --pad-token-multiple round the padded batch
shapes up so that the number of distinct shapes (each captures its own hipGraphs) stays small, and
--no-hip-graph runs the step eagerly.
"""
import argparse
import shutil
import sys

import numpy as np
import yaml

from ..configs import TrainConfig
from ..data import (Dataset, SentencePieceTokenizer, filter_example, get_dataset, get_tfrecord_dataset, slice_example)
from ..fit import Fit
from ..ops import StoredFeaturePlan
from ..training import TrainStep
from ..utils import LRScheduler, get_device_strategy, get_logger, path_join, set_random_seed

# fmt: off
parser = argparse.ArgumentParser(argument_default=argparse.SUPPRESS)
parser.add_argument("--from-file", type=str, help="load configs from file")

parser.add_argument("--data-config", type=str, help="data processing config file")
parser.add_argument("--model-config", type=str, help="model config file")
parser.add_argument("--sp-model-path", type=str, help="sentencepiece model path")
parser.add_argument("--train-dataset-paths", help="a tsv/tfrecord dataset file or multiple files ex) *.tsv")
parser.add_argument("--dev-dataset-paths", help="a tsv/tfrecord dataset file or multiple files ex) *.tsv")
parser.add_argument("--train-dataset-size", type=int, help="the number of training dataset examples")
parser.add_argument("--output-path", help="output directory to save log and model checkpoints")

parser.add_argument("--pretrained-model-path", type=str, help="pretrained model checkpoint")
parser.add_argument("--epochs", type=int)
parser.add_argument("--steps-per-epoch", type=int)
parser.add_argument("--learning-rate", type=float)
parser.add_argument("--min-learning-rate", type=float)
parser.add_argument("--warmup-rate", type=float)
parser.add_argument("--warmup-steps", type=int)
parser.add_argument("--batch-size", type=int)
parser.add_argument("--dev-batch-size", type=int)
parser.add_argument("--shuffle-buffer-size", type=int, help="shuffle buffer size")
parser.add_argument("--max-over-policy", type=str, choices=["filter", "slice"], help="policy for sequence whose length is over max")

parser.add_argument("--use-tfrecord", action="store_true", help="use tfrecord dataset")
parser.add_argument("--tensorboard-update-freq", type=int)
parser.add_argument("--mixed-precision", action="store_true", help="use mixed precision FP16")
parser.add_argument("--seed", type=int, help="Set random seed")
parser.add_argument("--skip-epochs", type=int, help="skip first N epochs and start N + 1 epoch")
parser.add_argument("--device", type=str, choices=["CPU", "GPU", "TPU"], help="device to use (TPU or GPU or CPU)")
# fmt: on

# flags of this build only (kept out of TrainConfig, which mirrors the reference's fields)
EXTRA_FLAGS = dict(pad_audio_multiple=1, pad_token_multiple=1, no_hip_graph=False)
parser.add_argument("--pad-audio-multiple", type=int, help="round the padded audio axis up to a multiple of this")
parser.add_argument("--pad-token-multiple", type=int, help="round the padded token axis up to a multiple of this")
parser.add_argument("--no-hip-graph", action="store_true", help="run the training step without hipGraph capture")


def _frames_to_samples(frames: int, frame_length: int, frame_step: int) -> int:
    """Largest sample count whose log-mel has `frames` frames (inverse of 1 + (N - L) // step)."""
    return frame_length + frames * frame_step - 1 if frames > 0 else frame_length - 1


def _raw_audio_policy(policy, data_config):
    """filter_example / slice_example (data.py:331-354) act on FEATURE frames in the reference; the raw-audio
    pipeline applies the same limits expressed in samples."""
    L, step = data_config.frame_length, data_config.frame_step
    max_frames, max_tokens = data_config.max_audio_length, data_config.max_token_length

    def frames(n):
        return 0 if n < L else 1 + (n - L) // step

    if policy == "filter":
        return lambda ds: ds.filter(lambda audio, text: frames(len(audio)) <= max_frames and np.size(text) <= max_tokens)
    keep = _frames_to_samples(max_frames, L, step)
    return lambda ds: ds.map(lambda audio, text: (audio[:keep], text[:max_tokens]))


def main(cfg: TrainConfig, **extra):
    opts = {**EXTRA_FLAGS, **extra}
    logger = get_logger("train")

    strategy = get_device_strategy(cfg.device)          # RuntimeError unless --device GPU on an MI355X
    rank, world = strategy.rank, strategy.world_size

    # train.py:88-90 seeds only when --seed is given.  Under data parallelism every rank must nevertheless draw the SAME
    # shuffles (each slices its share out of one global batch), so without --seed rank 0 draws a seed for all of them.
    seed = cfg.seed
    if seed is None and world > 1:
        import random

        import torch.distributed as dist
        box = [random.randrange(2 ** 31)]
        dist.broadcast_object_list(box, src=0)
        seed = box[0]
    if seed is not None:
        set_random_seed(seed)
        logger.info(f"[+] Set random seed to {seed}")

    # Copy config file
    if rank == 0:
        import os
        os.makedirs(cfg.output_path, exist_ok=True)
        with open(path_join(cfg.output_path, "train_configs.txt"), "w") as fout:
            for k, v in vars(cfg).items():
                if type(v) in [int, float, str]:
                    fout.write(f"{k:25}: {v}\n")
                    logger.info(f"{k:25}: {v}")
        shutil.copy(cfg.data_config_path, path_join(cfg.output_path, "data-config.yml"))
        shutil.copy(cfg.model_config_path, path_join(cfg.output_path, "model-config.yml"))

    if cfg.mixed_precision:
        # train.py:63-67 selects Keras' mixed_float16 policy (bf16 on TPU).  Here: bf16 operands on the bf16 MFMA for
        # every dense contraction, f32 accumulation, f32 master weights and f32 everywhere else; no loss scaling is
        # needed (bf16 keeps the f32 exponent range).
        from .. import ops
        ops.set_mixed_precision(True)
        logger.info("[+] --mixed-precision: dense contractions use bf16 operands with f32 accumulation (f32 weights and storage)")

    dc = cfg.data_config
    # Construct Dataset (host side: decode + tokenise only; features are computed on the GPU in the step)
    if cfg.use_tfrecord:
        logger.info(f"[+] Load TFRecord train dataset from {cfg.train_dataset_paths}")
        train_dataset = get_tfrecord_dataset(cfg.train_dataset_paths)
        logger.info(f"[+] Load TFRecord dev dataset from {cfg.train_dataset_paths}")
        dev_dataset = get_tfrecord_dataset(cfg.train_dataset_paths)    # sic: train.py:73-74 reads the train paths twice
        sa = vars(dc.spec_augment) if dc.spec_augment.enable else None
        frontend = StoredFeaturePlan(dc.frequency_dim, dc.use_delta_accelerate, sa)
        eval_frontend = StoredFeaturePlan(dc.frequency_dim, dc.use_delta_accelerate, None)
    else:
        logger.info(f"[+] Load Tokenizer from {cfg.sp_model_path}")
        tokenizer = SentencePieceTokenizer(cfg.sp_model_path, add_bos=True, add_eos=True)
        logger.info(f"[+] Load train dataset from {cfg.train_dataset_paths}")
        train_dataset = get_dataset(cfg.train_dataset_paths, dc.file_format, dc.sample_rate, tokenizer, cfg.shuffle_buffer_size > 1)
        logger.info(f"[+] Load dev dataset from {cfg.dev_dataset_paths}")
        dev_dataset = get_dataset(cfg.dev_dataset_paths, dc.file_format, dc.sample_rate, tokenizer)
        frontend = dc.logmel_plan(training=True, device=strategy.device)       # log-mel + SpecAugment + delta, fused
        eval_frontend = dc.logmel_plan(training=False, device=strategy.device)
    if dc.spec_augment.enable:
        logger.info("[+] Use SpecAugment (on device, inside the training step)")
    if dc.use_delta_accelerate:
        logger.info("[+] Use delta and deltas accelerate")

    # Apply max over policy
    if cfg.max_over_policy is not None:
        logger.info(f"[+] {cfg.max_over_policy.capitalize()} examples whose audio or token length is over than max value")
        if cfg.use_tfrecord:
            fn = (filter_example if cfg.max_over_policy == "filter" else slice_example)(dc.max_audio_length, dc.max_token_length)
        else:
            fn = _raw_audio_policy(cfg.max_over_policy, dc)
        train_dataset, dev_dataset = train_dataset.apply(fn), dev_dataset.apply(fn)
    elif cfg.device == "TPU":
        raise RuntimeError("You should set max-over-sequence-policy with TPU!")

    # Model Initialize
    logger.info("[+] Model Initialize")
    model = cfg.model_config.create_model(seed=seed)
    model.build(dc.frequency_dim, dc.feature_dim)
    if rank == 0:
        model.summary(print_fn=logger.info)

    # Load pretrained model
    if cfg.pretrained_model_path:
        logger.info("[+] Load weights of model")
        model.load_weights(cfg.pretrained_model_path)

    # Model Compile
    logger.info("[+] Model compile")
    schedule = LRScheduler(cfg.total_steps, cfg.learning_rate, cfg.min_learning_rate, cfg.warmup_rate, cfg.warmup_steps, cfg.offset_steps)
    trainer = TrainStep(model, schedule, frontend=frontend, eval_frontend=eval_frontend, strategy=strategy,
                        use_graph=not opts["no_hip_graph"])      # broadcasts rank 0's weights / moments / state to every replica

    # Shuffle & Make train example
    train_dataset = train_dataset.map(model.make_example)
    dev_dataset = dev_dataset.map(model.make_example)

    if cfg.steps_per_epoch:
        logger.info("[+] Repeat dataset")
        train_dataset = train_dataset.repeat()
        if cfg.skip_epochs:
            logger.info(f"[+] Skip Dataset by {cfg.skip_epochs}epoch x {cfg.steps_per_epoch} steps x {cfg.batch_size}")
            train_dataset = train_dataset.skip(cfg.steps_per_epoch * cfg.skip_epochs * cfg.batch_size)

    # Padded Batch.  Audio is padded in samples here (or stored frames with --use-tfrecord); the frame axis of
    # get_batching_shape is fixed only on TPU in the reference, so the GPU path always pads to the batch maximum.
    logger.info("[+] Pad Input data")
    train_dataset = (train_dataset.shuffle(cfg.shuffle_buffer_size, seed=seed)
                     .padded_batch(cfg.batch_size, with_lengths=True).prefetch(4))
    dev_dataset = dev_dataset.padded_batch(cfg.dev_batch_size, with_lengths=True)

    # Training
    logger.info("[+] Start training")
    fit = Fit(trainer, cfg.output_path, cfg.tensorboard_update_freq, logger, rank, world, opts["pad_audio_multiple"],
              opts["pad_token_multiple"], has_accuracy=bool(model.get_metrics()))
    fit(train_dataset, validation_data=dev_dataset, epochs=cfg.epochs, initial_epoch=cfg.skip_epochs,
        steps_per_epoch=cfg.steps_per_epoch)


if __name__ == "__main__":
    config = vars(parser.parse_args())
    if "from_file" in config:
        with open(config.pop("from_file")) as f:
            config = {**yaml.load(f, yaml.SafeLoader), **config}
    extra = {k: config.pop(k) for k in list(config) if k in EXTRA_FLAGS}
    sys.exit(main(TrainConfig(**config), **extra))
