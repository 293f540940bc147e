"""Shared pieces of run/evaluate.py and run/inference.py: model + searcher construction and the
host-batch -> GPU features -> greedy tokens loop."""
import numpy as np
import torch

from ..configs import DataConfig, get_model_config
from ..data import SentencePieceTokenizer
from ..models import LAS, DeepSpeech2
from ..ops import StoredFeaturePlan
from ..search import DeepSpeechSearcher, LAS_Searcher


def load_model_and_searcher(data_config: DataConfig, model_config_path: str, model_path: str, tokenizer: SentencePieceTokenizer, logger):
    """run/evaluate.py:66-91: create the model, restore `model_path`, pick the searcher."""
    model_config = get_model_config(model_config_path)
    model = model_config.create_model()
    model.build(data_config.frequency_dim, data_config.feature_dim)
    logger.info(f"[+] Load weights of model from {model_path}")
    model.load_weights(model_path)
    model.summary(print_fn=logger.info)
    bos_id, eos_id = tokenizer.tokenize("").tolist()           # evaluate.py:46: the BOS and EOS ids
    if isinstance(model, LAS):
        searcher = LAS_Searcher(model, data_config.max_token_length, bos_id, eos_id, model_config.pad_id)
    elif isinstance(model, DeepSpeech2):
        searcher = DeepSpeechSearcher(model, model_config.blank_index)
    else:
        raise ValueError(f"no searcher for {type(model).__name__}")
    return model, searcher


def feature_fn(data_config: DataConfig, stored_features: bool, device="cuda"):
    """(padded host batch, lengths) -> features [B, T, F, C] on the GPU: the fused log-mel + delta kernel for raw
    audio, the delta kernel alone for stored log-mel frames; never SpecAugment (evaluation)."""
    plan = StoredFeaturePlan(data_config.frequency_dim, data_config.use_delta_accelerate, None) if stored_features \
        else data_config.logmel_plan(training=False, device=device)

    def fn(batch: np.ndarray, lengths: np.ndarray) -> torch.Tensor:
        x = torch.from_numpy(np.ascontiguousarray(batch)).to(device)
        n = torch.from_numpy(np.ascontiguousarray(lengths, np.int32)).to(device)
        return plan(x, n, plan.num_frames(x.shape[1]))
    return fn


def strip_tokens(row, bos_id, eos_id, pad_id=0):
    """Token ids of one decoded row up to (not including) the first EOS, without BOS / padding."""
    out = []
    for t in (int(v) for v in row):
        if t == eos_id:
            break
        if t not in (bos_id, pad_id):
            out.append(t)
    return out
