from .data_config import DataConfig, SpecAugmentConfig
from .model_config import DeepSpeechConfig, LASConfig, get_model_config
from .train_config import TrainConfig

__all__ = ["DataConfig", "DeepSpeechConfig", "LASConfig", "SpecAugmentConfig", "TrainConfig", "get_model_config"]
