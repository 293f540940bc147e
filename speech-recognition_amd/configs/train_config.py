"""Mirror of speech_recognition/configs/train_config.py (fields = run.train CLI flags)."""
from dataclasses import dataclass, field
from math import ceil
from typing import Any, Optional

import yaml
from typing_extensions import Literal

from ._validate import REQUIRED, validate_fields
from .data_config import DataConfig
from .model_config import ModelConfig, get_model_config


@dataclass
class TrainConfig:
    # config paths (filled from data_config / model_config, train_config.py:66-74)
    data_config_path: str = ""
    model_config_path: str = ""
    # data processing config / model config: passed as PATHS, replaced by the loaded objects
    data_config: Any = REQUIRED
    model_config: Any = REQUIRED
    sp_model_path: Optional[str] = None
    train_dataset_paths: str = REQUIRED
    dev_dataset_paths: str = REQUIRED
    train_dataset_size: int = REQUIRED
    output_path: str = "output"
    pretrained_model_path: Optional[str] = None
    epochs: int = REQUIRED
    steps_per_epoch: Optional[int] = None
    learning_rate: float = REQUIRED
    min_learning_rate: float = 1.0e-5
    warmup_rate: float = 0.00
    warmup_steps: Optional[int] = None
    batch_size: int = REQUIRED
    dev_batch_size: int = REQUIRED
    shuffle_buffer_size: int = 10000
    max_over_policy: Optional[Literal["filter", "slice"]] = None
    use_tfrecord: bool = False
    tensorboard_update_freq: int = 1
    mixed_precision: bool = False
    seed: Optional[int] = None
    skip_epochs: int = 0
    device: Literal["CPU", "GPU", "TPU"] = "CPU"

    def __post_init__(self):
        validate_fields(self, skip=("data_config", "model_config"))
        if self.data_config is REQUIRED or self.model_config is REQUIRED:
            validate_fields(self)   # raises ValidationError naming the missing fields
        assert isinstance(self.data_config, str), "should pass 'data_config' parameter"
        assert isinstance(self.model_config, str), "should pass 'model_config' parameter"
        self.data_config_path = self.data_config
        self.model_config_path = self.model_config
        self.data_config = DataConfig.from_yaml(self.data_config)
        self.model_config = get_model_config(self.model_config)

    @classmethod
    def from_yaml(cls, file_path):
        with open(file_path) as f:
            return cls(**yaml.load(f, yaml.SafeLoader))

    @property
    def audio_pad_length(self):
        return None if self.device != "TPU" else self.data_config.max_audio_length

    @property
    def token_pad_length(self):
        return None if self.device != "TPU" else self.data_config.max_token_length

    @property
    def total_steps(self):
        return (self.steps_per_epoch or ceil(self.train_dataset_size / self.batch_size)) * self.epochs

    @property
    def offset_steps(self):
        return (self.steps_per_epoch or ceil(self.train_dataset_size / self.batch_size)) * self.skip_epochs
