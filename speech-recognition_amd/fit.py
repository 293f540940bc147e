"""`model.fit(...)` of run/train.py:201-217 for the MI355X build: the epoch loop around TrainStep.

What Keras does there and what this keeps: `epochs` x `steps_per_epoch` training steps (or one pass
over the dataset when steps_per_epoch is None), running loss / metric averages, a validation pass over
`validation_data` after every epoch, a weights-only checkpoint per epoch named by
`model.model_checkpoint_path` (ModelCheckpoint, train.py:208-212, plus TF's `checkpoint` state file) and
scalar logs every `update_freq` steps (TensorBoard callback, train.py:213-215; written here as JSON
lines under logs/train and logs/validation).

Device statistics are copied to the host in chunks, not per step, so the stream never drains inside
an epoch except at the log interval.
"""
import json
import os
import time
from typing import Optional

import numpy as np
import torch


def unpack_batch(batch, lengths):
    """(batch, lengths) from Dataset.padded_batch(..., with_lengths=True) after model.make_example ->
    (audio, n_audio, tokens[B, L]).  LAS examples are ((audio, tokens[:-1]), tokens[1:]) (las.py:396-406),
    DeepSpeech2 examples are (audio, tokens) (deepspeech2.py:192-202)."""
    if isinstance(batch[0], tuple):
        (audio, tok_in), tok_out = batch
        n_audio = lengths[0][0]
        tokens = np.concatenate([tok_in[:, :1], tok_out], axis=1)
    else:
        audio, tokens = batch
        n_audio = lengths[0]
    return audio, n_audio, tokens.astype(np.int32, copy=False)


def _pad_axis(x, axis, multiple):
    if multiple <= 1:
        return x
    n = x.shape[axis]
    want = -(-n // multiple) * multiple
    if want == n:
        return x
    pad = [(0, 0)] * x.ndim
    pad[axis] = (0, want - n)
    return np.pad(x, pad)


class ScalarLog:
    """Stand-in for the TensorBoard callback: one JSON object per line in <dir>/scalars.jsonl."""

    def __init__(self, directory: str):
        os.makedirs(directory, exist_ok=True)
        self._f = open(os.path.join(directory, "scalars.jsonl"), "a")

    def write(self, step: int, **scalars):
        self._f.write(json.dumps({"step": int(step), **{k: float(v) for k, v in scalars.items()}}) + "\n")
        self._f.flush()

    def close(self):
        self._f.close()


class Fit:
    def __init__(self, trainer, output_path: Optional[str] = None, update_freq: int = 1, logger=None, rank: int = 0, world: int = 1,
                 pad_audio_multiple: int = 1, pad_token_multiple: int = 1, has_accuracy: bool = True):
        self.trainer, self.model = trainer, trainer.model
        self.out, self.freq, self.logger = output_path, max(1, int(update_freq)), logger
        self.rank, self.world = rank, world
        self.pad_audio, self.pad_token = pad_audio_multiple, pad_token_multiple
        self.has_accuracy = has_accuracy
        self.history = []
        self._logs = {}
        if output_path and rank == 0:
            self._logs = {k: ScalarLog(os.path.join(output_path, "logs", k)) for k in ("train", "validation")}
            os.makedirs(os.path.join(output_path, "models"), exist_ok=True)

    def _info(self, msg):
        if self.logger is not None and self.rank == 0:
            self.logger.info(msg)

    def _device_batch(self, element):
        audio, n_audio, tokens = unpack_batch(*element)
        if self.world > 1:                                    # --batch-size is the GLOBAL batch (train.py:194)
            per = audio.shape[0] // self.world
            if per == 0:
                return None
            sl = slice(self.rank * per, (self.rank + 1) * per)
            audio, n_audio, tokens = audio[sl], n_audio[sl], tokens[sl]
        audio = _pad_axis(audio, 1, self.pad_audio)
        tokens = _pad_axis(tokens, 1, self.pad_token)
        return (torch.from_numpy(np.ascontiguousarray(audio)).pin_memory(), torch.from_numpy(np.ascontiguousarray(n_audio, np.int32)),
                torch.from_numpy(np.ascontiguousarray(tokens)).pin_memory())

    # ------------------------------------------------------------------------------------------ epoch pieces
    def _train_epoch(self, it, steps, epoch):
        tr = self.trainer
        chunk = torch.zeros(self.freq, 4, device=self.model.device)
        host_loss, host_correct, host_kept, seen = 0.0, 0.0, 0.0, 0
        pending, t0, ws = 0, time.time(), None

        def flush():
            nonlocal host_loss, host_correct, host_kept, seen, pending
            if pending == 0:
                return
            tr.read_stats(ws)                                 # synchronises + checks the persistent kernels' error words
            vals = chunk[:pending].cpu().numpy()
            host_loss += float(vals[:, 0].sum())
            host_correct += float(vals[:, 1].sum())
            host_kept += float(vals[:, 2].sum())
            seen += pending
            pending = 0
            if self._logs:
                scal = dict(loss=host_loss / seen, lr=tr.sched_host(tr.iterations))
                if self.has_accuracy:
                    scal["accuracy"] = host_correct / max(host_kept, 1.0)
                self._logs["train"].write(tr.iterations, **scal)

        done = 0
        while steps is None or done < steps:
            element = next(it, None)
            if element is None:
                break
            dev = self._device_batch(element)
            if dev is None:
                continue
            ws = tr.step(*dev)
            with torch.cuda.stream(tr.stream):
                chunk[pending].copy_(ws.stats, non_blocking=True)
            pending += 1
            done += 1
            if pending == self.freq:
                flush()
        flush()
        if seen == 0:
            raise RuntimeError("fit: the training dataset produced no batches")
        dt = time.time() - t0
        logs = dict(loss=host_loss / seen, steps=seen, seconds=dt)
        if self.has_accuracy:
            logs["accuracy"] = host_correct / max(host_kept, 1.0)
        return logs

    def _validate(self, dataset):
        tr = self.trainer
        loss, correct, kept, n = 0.0, 0.0, 0.0, 0
        for element in dataset:
            dev = self._device_batch(element)
            if dev is None:
                continue
            s = tr.evaluate(*dev)
            loss, correct, kept, n = loss + s[0], correct + s[1], kept + s[2], n + 1
        if n == 0:
            return {}
        logs = dict(val_loss=loss / n)
        if self.has_accuracy:
            logs["val_accuracy"] = correct / max(kept, 1.0)
        return logs

    def _checkpoint(self, epoch, logs):
        if not (self.out and self.rank == 0):
            return None
        fields = dict(epoch=epoch, **logs)
        fields.setdefault("val_loss", float("nan"))
        fields.setdefault("val_accuracy", float("nan"))
        name = self.model.model_checkpoint_path.format(**fields)
        path = os.path.join(self.out, "models", name)
        self.model.save_weights(path)
        with open(os.path.join(self.out, "models", "checkpoint"), "w") as f:    # TF's checkpoint state file
            f.write(f'model_checkpoint_path: "{name}"\nall_model_checkpoint_paths: "{name}"\n')
        self._info(f"Epoch {epoch}: saving model to {path}")
        return path

    # ------------------------------------------------------------------------------------------ fit
    def __call__(self, train_dataset, validation_data=None, epochs: int = 1, initial_epoch: int = 0, steps_per_epoch: Optional[int] = None):
        """Datasets yield (batch, lengths) pairs (Dataset.padded_batch(..., with_lengths=True))."""
        it = iter(train_dataset) if steps_per_epoch else None
        for epoch in range(initial_epoch, epochs):
            if not steps_per_epoch:
                it = iter(train_dataset)
            logs = self._train_epoch(it, steps_per_epoch, epoch)
            if validation_data is not None:
                logs.update(self._validate(validation_data))
            if self._logs and "val_loss" in logs:
                self._logs["validation"].write(self.trainer.iterations, **{k[4:]: v for k, v in logs.items() if k.startswith("val_")})
            self._info(f"Epoch {epoch + 1}/{epochs} - " + " - ".join(f"{k}: {v:.4f}" for k, v in logs.items()))
            logs["checkpoint"] = self._checkpoint(epoch + 1, {k: v for k, v in logs.items() if isinstance(v, float)})
            self.history.append(logs)
        for log in self._logs.values():
            log.close()
        return self.history
