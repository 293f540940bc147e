"""Mirror of speech_recognition/configs/model_config.py."""
from abc import ABCMeta, abstractmethod
from dataclasses import dataclass
from typing import List, Union

import yaml

from ._validate import validate_fields


class ModelConfig(metaclass=ABCMeta):
    @abstractmethod
    def create_model(self):
        pass

    @property
    @abstractmethod
    def model_name(self):
        pass


def get_model_config(model_config_path: str) -> Union["LASConfig", "DeepSpeechConfig"]:
    """Load model config file and return ModelConfig instance (model_config.py:20-36)."""
    with open(model_config_path) as f:
        model_config_dict = yaml.load(f, yaml.SafeLoader)
    model_name = model_config_dict["model_name"].lower()
    if model_name in ["ds2", "deepspeech2"]:
        return DeepSpeechConfig(**model_config_dict)
    if model_name in ["las"]:
        return LASConfig(**model_config_dict)
    raise ValueError(f"Model Name: {model_name} is invalid!")


@dataclass
class LASConfig(ModelConfig):
    """Config for LAS model initialize (model_config.py:39-76)."""

    rnn_type: str
    vocab_size: int
    encoder_hidden_dim: int
    decoder_hidden_dim: int
    num_encoder_layers: int
    num_decoder_layers: int
    dropout: float
    teacher_forcing_rate: float
    pad_id: int
    model_name: str = "LAS"

    def __post_init__(self):
        validate_fields(self)

    def create_model(self, **kwargs):
        from ..models import LAS
        return LAS(rnn_type=self.rnn_type, vocab_size=self.vocab_size, encoder_hidden_dim=self.encoder_hidden_dim,
                   decoder_hidden_dim=self.decoder_hidden_dim, num_encoder_layers=self.num_encoder_layers,
                   num_decoder_layers=self.num_decoder_layers, dropout=self.dropout,
                   teacher_forcing_rate=self.teacher_forcing_rate, pad_id=self.pad_id, **kwargs)


@dataclass
class DeepSpeechConfig(ModelConfig):
    """Config for DeepSpeech2 model initialize (model_config.py:79-125)."""

    num_conv_layers: int
    channels: List[int]
    kernel_sizes: List[List[int]]
    strides: List[List[int]]
    rnn_type: str
    num_reccurent_layers: int
    hidden_dim: int
    dropout: float
    recurrent_dropout: float
    vocab_size: int
    blank_index: int
    pad_index: int
    model_name: str = "DeepSpeech2"

    def __post_init__(self):
        validate_fields(self)

    def create_model(self, **kwargs):
        from ..models import DeepSpeech2
        return DeepSpeech2(num_conv_layers=self.num_conv_layers, channels=self.channels, kernel_sizes=self.kernel_sizes,
                           strides=self.strides, rnn_type=self.rnn_type, num_reccurent_layers=self.num_reccurent_layers,
                           hidden_dim=self.hidden_dim, dropout=self.dropout, recurrent_dropout=self.recurrent_dropout,
                           vocab_size=self.vocab_size, blank_index=self.blank_index, pad_index=self.pad_index, **kwargs)
