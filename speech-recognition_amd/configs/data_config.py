"""Mirror of speech_recognition/configs/data_config.py (same fields, defaults, derived properties)."""
from dataclasses import dataclass
from typing import Optional

import yaml
from typing_extensions import Literal

from ._validate import validate_fields


@dataclass
class SpecAugmentConfig:
    """Parameters for SpecAugment (data_config.py:10-20)."""

    enable: bool
    W: Optional[int] = None
    F: Optional[int] = None
    m_F: Optional[int] = None
    T: Optional[int] = None
    p: Optional[float] = None
    m_T: Optional[int] = None

    def __post_init__(self):
        validate_fields(self)


@dataclass
class DataConfig:
    """Config for audio data processing or data dependant parameter (data_config.py:23-106)."""

    file_format: Literal["pcm", "wav", "flac", "mp3"]
    audio_feature_type: Literal["spectrogram", "log-mel-spectrogram", "mfcc"]
    sample_rate: int
    frame_length: int
    frame_step: int
    fft_length: int
    max_audio_length: int
    max_token_length: int
    use_delta_accelerate: bool
    spec_augment: SpecAugmentConfig
    num_mel_bins: Optional[int] = None
    num_mfcc: Optional[int] = None
    lower_edge_hertz: Optional[float] = None
    upper_edge_hertz: Optional[float] = None

    def __post_init__(self):
        if isinstance(self.spec_augment, dict):
            self.spec_augment = SpecAugmentConfig(**self.spec_augment)
        validate_fields(self, skip=("spec_augment",))
        if self.audio_feature_type in ["log-mel-spectrogram", "mfcc"]:
            assert all([self.num_mel_bins, self.lower_edge_hertz, self.upper_edge_hertz]), \
                '"num_mel_bins", "lower_edge_hertz", "upper_edge_hertz" is required'
        if self.audio_feature_type == "mfcc":
            assert self.num_mfcc, '"num_mfcc" is required'

    @property
    def feature_dim(self):
        return 3 if self.use_delta_accelerate else 1

    @property
    def frequency_dim(self):
        if self.audio_feature_type == "spectrogram":
            return self.fft_length // 2 + 1
        if self.audio_feature_type == "log-mel-spectrogram":
            return self.num_mel_bins
        if self.audio_feature_type == "mfcc":
            return self.num_mfcc

    @property
    def audio_feature_fn(self):
        """data_config.py:77-101: the per-clip feature function of this config (all three types run the fused kernel)."""
        from ..data import make_log_mel_spectrogram, make_mfcc, make_spectrogram
        if self.audio_feature_type == "spectrogram":
            return make_spectrogram(self.frame_length, self.frame_step, self.fft_length)
        if self.audio_feature_type == "log-mel-spectrogram":
            return make_log_mel_spectrogram(self.sample_rate, self.frame_length, self.frame_step, self.fft_length,
                                            self.num_mel_bins, self.lower_edge_hertz, self.upper_edge_hertz)
        if self.audio_feature_type == "mfcc":
            return make_mfcc(self.sample_rate, self.frame_length, self.frame_step, self.fft_length, self.num_mel_bins,
                             self.num_mfcc, self.lower_edge_hertz, self.upper_edge_hertz)

    def logmel_plan(self, training: bool, device="cuda"):
        """The fused GPU front end (features of audio_feature_type + SpecAugment when training + delta) for this config."""
        from .. import ops
        sa = None
        if training and self.spec_augment.enable:
            sa = dict(enable=True, W=self.spec_augment.W, F=self.spec_augment.F, m_F=self.spec_augment.m_F, T=self.spec_augment.T,
                      p=self.spec_augment.p, m_T=self.spec_augment.m_T)
        return ops.LogmelPlan(self.sample_rate, self.frame_length, self.frame_step, self.fft_length, self.num_mel_bins or 0,
                              self.lower_edge_hertz or 0.0, self.upper_edge_hertz or 0.0, use_delta=self.use_delta_accelerate,
                              spec_augment=sa, device=device, feature_type=self.audio_feature_type, num_mfcc=self.num_mfcc)

    @classmethod
    def from_yaml(cls, file_path) -> "DataConfig":
        with open(file_path) as f:
            return cls(**yaml.load(f, yaml.SafeLoader))
