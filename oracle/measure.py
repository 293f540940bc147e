"""Oracle (TEST INFRASTRUCTURE): losses / metric of /root/reference/speech_recognition/measure.py
and the optimizer + LR schedule of run/train.py:158-168, utils.py:11-35, in torch-CPU.
"""
import math

import torch


def sparse_categorical_crossentropy(y_true, logits, ignore_index=0):
    """measure.py:18-21 + Keras SUM_OVER_BATCH_SIZE [TF-sem]: mean of -log softmax(logits)[y]
    over tokens with y != ignore_index (0 when there is none: Keras `_safe_mean`)."""
    lp = torch.log_softmax(logits, dim=-1)
    nll = -lp.gather(-1, y_true.long().unsqueeze(-1)).squeeze(-1)
    keep = y_true != ignore_index
    n = int(keep.sum())
    if n == 0:
        return nll.sum() * 0.0
    return nll[keep].sum() / n


def sparse_categorical_accuracy(y_true, logits, ignore_index=0):
    """measure.py:55-69: (sum correct, count) over tokens with y != ignore_index."""
    pred = logits.reshape(-1, logits.shape[-1]).argmax(dim=-1)
    y = y_true.reshape(-1)
    keep = y != ignore_index
    return float((pred[keep] == y[keep]).sum()), float(keep.sum())


def _lse_rows(st):
    """log-sum-exp over dim 0 of st [k, S] whose entries may be -inf (a column of only -inf gives -inf, with zero gradient)."""
    m = st.max(dim=0).values
    dead = torch.isinf(m) & (m < 0)
    m_safe = torch.where(dead, torch.zeros_like(m), m)
    tot = torch.exp(st - m_safe).sum(dim=0)
    return torch.where(dead, m, m_safe + torch.log(torch.where(dead, torch.ones_like(tot), tot)))


def ctc_nll(logits, labels, label_len, blank):
    """[TF-sem] tf.nn.ctc_loss for ONE sample: logits [T, V] (log-softmax applied here), labels
    [L] ints, first `label_len` used.  Log-space alpha recursion over the blank-extended label row; differentiable.
    alpha_t(s) = lse(alpha_{t-1}(s), alpha_{t-1}(s-1), [alpha_{t-1}(s-2) if ext(s) is a label differing from ext(s-2)]) + lp_t(ext(s)),
    all S states of a frame at once (the label log-probabilities are gathered once: [T, S])."""
    T, V = logits.shape
    lp = torch.log_softmax(logits, dim=-1)
    lab = [int(x) for x in labels[:label_len]]
    S = 2 * len(lab) + 1
    ext = [blank if s % 2 == 0 else lab[s // 2] for s in range(S)]
    lpe = lp[:, torch.tensor(ext, dtype=torch.long)]                                  # [T, S]
    skip = torch.tensor([s >= 2 and ext[s] != blank and ext[s] != ext[s - 2] for s in range(S)])
    ninf = torch.full((S,), -float("inf"), dtype=logits.dtype)
    start = torch.arange(S) < 2
    alpha = torch.where(start, lpe[0], ninf)
    for t in range(1, T):
        a1 = torch.cat([ninf[:1], alpha[:-1]])
        a2 = torch.where(skip, torch.cat([ninf[:2], alpha[:-2]]), ninf) if S >= 2 else ninf
        alpha = _lse_rows(torch.stack([alpha, a1, a2])) + lpe[t]
    tail = alpha[S - 2:] if S >= 2 else alpha[S - 1:]
    return -_lse_rows(tail[:, None])[0]


def ctc_loss(y_true, logits, blank_index, pad_index=0):
    """CTCLoss.call (measure.py:32-42): label_len = count(y != pad); logit_len = T for every row
    (the encoder mask is not used); per-sample NLL / label_len; Keras mean over the batch."""
    B = y_true.shape[0]
    per = []
    for b in range(B):
        ll = int((y_true[b] != pad_index).sum())
        per.append(ctc_nll(logits[b].to(torch.float64) if logits.dtype == torch.float64 else logits[b].float(),
                           y_true[b], ll, blank_index) / ll)
    per = torch.stack(per)
    return per.mean(), per


class LRScheduler:
    """utils.py:11-35."""

    def __init__(self, total_steps, max_learning_rate, min_learning_rate, warmup_rate=0.0, warmup_steps=0,
                 offset_steps=0):
        self.warmup_steps = int(total_steps * warmup_rate) + 1 if not warmup_steps else warmup_steps
        self.increasing_delta = max_learning_rate / self.warmup_steps if self.warmup_steps else 1e12
        self.decreasing_delta = (max_learning_rate - min_learning_rate) / (total_steps - self.warmup_steps)
        self.max_learning_rate = max_learning_rate
        self.min_learning_rate = min_learning_rate
        self.offset_steps = offset_steps or 0

    def __call__(self, step):
        step = float(step + self.offset_steps)
        lr = min(step * self.increasing_delta,
                 self.max_learning_rate - (step - self.warmup_steps) * self.decreasing_delta)
        return max(lr, self.min_learning_rate)


def adam_step(params, grads, m, v, iterations, lr, beta1=0.9, beta2=0.999, eps=1e-7):
    """[TF-sem] Keras Adam (train.py:159-168): t = iterations+1;
    theta -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps).  In-place on dicts of tensors."""
    t = iterations + 1
    lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    for k in params:
        g = grads[k]
        m[k].mul_(beta1).add_(g, alpha=1.0 - beta1)
        v[k].mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
        params[k].sub_(lr_t * m[k] / (v[k].sqrt() + eps))
