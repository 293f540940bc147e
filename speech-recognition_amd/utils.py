"""Mirror of the parts of speech_recognition/utils.py the training path uses."""
import logging
import os
import random
import sys
from typing import Optional

import numpy as np
import torch


class LRScheduler:
    """utils.py:11-35: linear warm-up to max_learning_rate, then linear decay to min_learning_rate.
    The training step evaluates the same formula on the device (asr_adam_step); this host object is
    for logging and tests."""

    def __init__(self, total_steps: int, max_learning_rate: float, min_learning_rate: float, warmup_rate: float = 0.0,
                 warmup_steps: Optional[int] = 0, offset_steps: Optional[int] = 0):
        self.total_steps = total_steps
        self.warmup_rate = warmup_rate
        self.warmup_steps = int(total_steps * warmup_rate) + 1 if not warmup_steps else warmup_steps
        self.increasing_delta = max_learning_rate / self.warmup_steps if self.warmup_steps else 1e12
        self.decreasing_delta = (max_learning_rate - min_learning_rate) / (total_steps - self.warmup_steps)
        self.max_learning_rate = float(max_learning_rate)
        self.min_learning_rate = float(min_learning_rate)
        self.offset_steps = offset_steps or 0

    def __call__(self, step):
        step = float(step + self.offset_steps)
        lr = min(step * self.increasing_delta, self.max_learning_rate - (step - self.warmup_steps) * self.decreasing_delta)
        return max(lr, self.min_learning_rate)

    def device_schedule(self):
        from . import ops
        return ops.lr_schedule(self.total_steps, self.max_learning_rate, self.min_learning_rate, self.warmup_rate,
                               self.warmup_steps, self.offset_steps)


def levenshtein_distance(truth, hypothesis, normalize=True):
    """utils.py:80-101: edit distance between two sequences (strings or lists), optionally divided by
    len(truth) (the WER / CER of run/evaluate.py).  Two-row dynamic programme."""
    n = len(hypothesis)
    prev = list(range(n + 1))
    for i in range(1, len(truth) + 1):
        cur = [i] + [0] * n
        ti = truth[i - 1]
        for j in range(1, n + 1):
            cur[j] = min(prev[j - 1] + (ti != hypothesis[j - 1]), prev[j] + 1, cur[j - 1] + 1)
        prev = cur
    return prev[n] / len(truth) if normalize else prev[n]


def get_logger(name: str) -> logging.Logger:
    """utils.py:104-113."""
    logger = logging.getLogger(name)
    logger.propagate = False
    logger.setLevel(logging.DEBUG)
    if not logger.handlers:
        handler = logging.StreamHandler(sys.stdout)
        handler.setFormatter(logging.Formatter("[%(asctime)s] %(message)s"))
        logger.addHandler(handler)
    return logger


def path_join(*paths) -> str:
    return os.path.join(*paths)


def set_random_seed(seed: int):
    """utils.py:123-127 (random / numpy / framework RNG)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


class DeviceStrategy:
    """What utils.get_device_strategy (utils.py:130-156) returns here: the local GPU plus the
    data-parallel group (one process per GPU; RCCL over xGMI through torch.distributed)."""

    def __init__(self, device: torch.device, world_size: int, rank: int, group=None):
        self.device, self.world_size, self.rank, self.group = device, world_size, rank, group

    @property
    def num_replicas_in_sync(self):
        return self.world_size


def get_device_strategy(device: str) -> DeviceStrategy:
    """CPU / TPU are not supported by this build: the arithmetic exists only as gfx950 kernels.
    GPU: one process drives one MI355X; with WORLD_SIZE > 1 (torchrun) gradients are all-reduced
    over RCCL (the MirroredStrategy branch of the reference, utils.py:148-149)."""
    if device.upper() != "GPU":
        raise RuntimeError(f"device {device!r} is not available in speech_recognition_amd (MI355X only); use --device GPU")
    if not torch.cuda.is_available():
        raise RuntimeError("Cannot find GPU!")  # utils.py:144-145
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    return DeviceStrategy(torch.device("cuda", local), world, rank)
