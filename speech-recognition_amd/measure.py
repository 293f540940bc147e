"""Losses and metric of speech_recognition/measure.py on the HIP kernels.

The objects keep the reference's call conventions (``loss(y_true, y_pred)``, metric
``update_state`` / ``result``); the training loop itself uses the fused forward+gradient kernels
directly (models/*.loss_and_grad)."""
import torch

from . import ops


class SparseCategoricalCrossentropy:
    """measure.py:4-21: sparse softmax cross-entropy from logits, tokens equal to ignore_index dropped,
    Keras SUM_OVER_BATCH_SIZE = mean over the kept tokens."""

    def __init__(self, ignore_index: int = 0, from_logits=True, name="sparse_categorical_crossentropy"):
        if not from_logits:
            raise NotImplementedError("only from_logits=True (the value the reference models use)")
        self.ignore_index, self.name = ignore_index, name

    def __call__(self, y_true, y_pred):
        V = y_pred.shape[-1]
        logits = y_pred.reshape(-1, V).contiguous().clone()
        labels = y_true.reshape(-1).to(torch.int32).contiguous()
        stats = torch.zeros(4, device=logits.device)
        ops.softmax_xent(logits, labels, stats, self.ignore_index, write_grad=False)
        return stats[0]


class SparseCategoricalAccuracy:
    """measure.py:45-69: running (sum correct) / (count) over tokens != ignore_index."""

    def __init__(self, ignore_index: int = 0, name="accuracy"):
        self.ignore_index, self.name = ignore_index, name
        self.total_sum = 0.0
        self.total_count = 0.0

    def update_state(self, y_true, y_pred, sample_weight=None):
        if sample_weight is not None:
            raise NotImplementedError("sample_weight is not used by the reference training path")
        V = y_pred.shape[-1]
        logits = y_pred.reshape(-1, V).contiguous().clone()
        labels = y_true.reshape(-1).to(torch.int32).contiguous()
        stats = torch.zeros(4, device=logits.device)
        ops.softmax_xent(logits, labels, stats, self.ignore_index, write_grad=False)
        s = stats.cpu()
        self.update_from_stats(float(s[1]), float(s[2]))

    def update_from_stats(self, correct, count):
        self.total_sum += correct
        self.total_count += count

    def reset_states(self):
        self.total_sum = self.total_count = 0.0

    def result(self):
        return self.total_sum / self.total_count if self.total_count else float("nan")


class CTCLoss:
    """measure.py:24-42 (DeepSpeech2): per-sample CTC negative log-likelihood / label_length, logit
    length = full T' for every row; Keras mean over the batch."""

    def __init__(self, blank_index: int, pad_index: int = 0, name="ctc_loss"):
        self.blank_index, self.pad_index, self.name = blank_index, pad_index, name

    def __call__(self, y_true, y_pred):
        from .models.deepspeech2 import ctc_loss_only
        return ctc_loss_only(y_true, y_pred, self.blank_index, self.pad_index)
