"""Config boundary types: the behaviours the reference pins in
tests/configs/test_{data,model,train}_config.py, restated against speech_recognition_amd.configs -
missing fields, wrong types, YAML round trips, derived properties (train_config.py:66-95,
data_config.py:23-106, model_config.py:20-125)."""
import os
from dataclasses import asdict

import pytest
import yaml
from pydantic import ValidationError

from speech_recognition_amd.configs import get_model_config
from speech_recognition_amd.configs.data_config import DataConfig, SpecAugmentConfig
from speech_recognition_amd.configs.model_config import DeepSpeechConfig, LASConfig
from speech_recognition_amd.configs.train_config import TrainConfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = os.path.join(ROOT, "resources", "configs")
LIBRI = os.path.join(CONFIGS, "libri_config.yml")
LAS_SMALL = os.path.join(CONFIGS, "las_small.yml")
LAS_LARGE = os.path.join(CONFIGS, "las_large.yml")
DS = os.path.join(CONFIGS, "deepspeech.yml")


def _yaml(path):
    with open(path) as f:
        return yaml.load(f, yaml.SafeLoader)


# ---------------------------------------------------------------- DataConfig (reference tests/configs/test_data_config.py)
def test_data_config_requires_its_fields():
    with pytest.raises(TypeError):
        DataConfig()


def test_data_config_rejects_unknown_file_format():
    with pytest.raises(ValidationError):
        DataConfig(**{**_yaml(LIBRI), "file_format": "hello"})


def test_data_config_round_trips_the_shipped_yaml():
    raw = _yaml(LIBRI)
    cfg = DataConfig(**raw)
    assert asdict(cfg) == raw
    assert cfg.feature_dim == 3                      # use_delta_accelerate: true
    assert cfg.frequency_dim == raw["num_mel_bins"] == 80
    assert isinstance(cfg.spec_augment, SpecAugmentConfig)
    assert DataConfig.from_yaml(LIBRI) == cfg


@pytest.mark.parametrize("field,value", [("audio_feature_type", "chroma"), ("sample_rate", "fast"), ("use_delta_accelerate", "maybe"),
                                          ("spec_augment", {"enable": "perhaps"})])
def test_data_config_rejects_wrong_types(field, value):
    with pytest.raises(ValidationError):
        DataConfig(**{**_yaml(LIBRI), field: value})


def test_data_config_feature_geometry_per_type():
    raw = _yaml(LIBRI)
    spec = DataConfig(**{**raw, "audio_feature_type": "spectrogram", "use_delta_accelerate": False})
    assert spec.frequency_dim == raw["fft_length"] // 2 + 1 and spec.feature_dim == 1
    mfcc = DataConfig(**{**raw, "audio_feature_type": "mfcc", "num_mfcc": 13})
    assert mfcc.frequency_dim == 13
    with pytest.raises(AssertionError):              # data_config.py:56-62: mel parameters are required for log-mel / mfcc
        DataConfig(**{**raw, "num_mel_bins": None})
    with pytest.raises(AssertionError):
        DataConfig(**{**raw, "audio_feature_type": "mfcc", "num_mfcc": None})


# ---------------------------------------------------------------- ModelConfig (reference tests/configs/test_model_config.py)
def test_model_configs_require_their_fields():
    with pytest.raises(TypeError):
        LASConfig()
    with pytest.raises(TypeError):
        DeepSpeechConfig()


def test_model_configs_reject_wrong_types():
    with pytest.raises(ValidationError):
        LASConfig(**{**_yaml(LAS_SMALL), "vocab_size": "good"})
    with pytest.raises(ValidationError):
        DeepSpeechConfig(**{**_yaml(DS), "channels": 55})          # must be a list


@pytest.mark.parametrize("path,cls", [(LAS_SMALL, LASConfig), (LAS_LARGE, LASConfig), (DS, DeepSpeechConfig)])
def test_get_model_config_equals_direct_construction(path, cls):
    assert cls(**_yaml(path)) == get_model_config(path)


def test_get_model_config_rejects_unknown_model_name(tmp_path):
    p = tmp_path / "bad.yml"
    p.write_text(yaml.safe_dump({**_yaml(LAS_SMALL), "model_name": "wav2letter"}))
    with pytest.raises(ValueError, match="is invalid"):
        get_model_config(str(p))


def test_model_config_values_of_the_headline_models():
    s, l, d = get_model_config(LAS_SMALL), get_model_config(LAS_LARGE), get_model_config(DS)
    assert (s.encoder_hidden_dim, s.decoder_hidden_dim, s.num_encoder_layers, s.num_decoder_layers, s.vocab_size) == (256, 256, 3, 2, 16000)
    assert l.encoder_hidden_dim == 1024 and l.vocab_size == 16000
    assert d.blank_index == 14 and d.num_reccurent_layers == 7 and d.hidden_dim == 128     # upstream spelling kept


# ---------------------------------------------------------------- TrainConfig (reference tests/configs/test_train_config.py)
def _train_kwargs(**over):
    kw = {"data_config": LIBRI, "model_config": LAS_SMALL, "train_dataset_paths": "hi", "dev_dataset_paths": "hello",
          "train_dataset_size": 10, "epochs": 1, "learning_rate": 1.0, "batch_size": 10, "dev_batch_size": 20}
    kw.update(over)
    return kw


def test_train_config_requires_its_fields():
    with pytest.raises(ValidationError):
        TrainConfig()


def test_train_config_missing_config_file():
    with pytest.raises(FileNotFoundError):
        TrainConfig(**_train_kwargs(data_config="nofile"))


def test_train_config_loads_nested_configs_and_keeps_the_scalars():
    kw = _train_kwargs()
    cfg = TrainConfig(**kw)
    assert cfg.audio_pad_length is None and cfg.token_pad_length is None          # only the TPU path pads to a fixed length
    assert cfg.data_config == DataConfig.from_yaml(LIBRI) and cfg.model_config == get_model_config(LAS_SMALL)
    assert cfg.data_config_path == LIBRI and cfg.model_config_path == LAS_SMALL
    scalars = {k: v for k, v in kw.items() if k not in ("data_config", "model_config")}
    assert all(item in asdict(cfg).items() for item in scalars.items())


@pytest.mark.parametrize("field,value", [("epochs", "many"), ("device", "NPU"), ("max_over_policy", "drop"), ("learning_rate", "big")])
def test_train_config_rejects_wrong_types(field, value):
    with pytest.raises(ValidationError):
        TrainConfig(**_train_kwargs(**{field: value}))


def test_train_config_step_arithmetic():
    """train_config.py:89-95: total_steps / offset_steps feed LRScheduler (utils.py:28-31)."""
    cfg = TrainConfig(**_train_kwargs(train_dataset_size=101, batch_size=10, epochs=3, skip_epochs=2))
    assert cfg.total_steps == 11 * 3 and cfg.offset_steps == 11 * 2
    cfg = TrainConfig(**_train_kwargs(steps_per_epoch=7, epochs=4, skip_epochs=1))
    assert cfg.total_steps == 28 and cfg.offset_steps == 7


def test_train_config_tpu_pads_to_the_data_config_maxima():
    cfg = TrainConfig(**_train_kwargs(device="TPU"))
    assert cfg.audio_pad_length == cfg.data_config.max_audio_length
    assert cfg.token_pad_length == cfg.data_config.max_token_length


def test_train_config_from_yaml(tmp_path):
    p = tmp_path / "train.yml"
    p.write_text(yaml.safe_dump(_train_kwargs(seed=3, mixed_precision=True)))
    cfg = TrainConfig.from_yaml(str(p))
    assert cfg.seed == 3 and cfg.mixed_precision is True and cfg.batch_size == 10
