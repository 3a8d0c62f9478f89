"""Model presets of the reference, mirrored for the Python harness.

Reference: Sources/Qwen3ASR/AudioEncoder.swift:28-88 (Qwen3AudioEncoderConfig .small/.large),
Sources/Qwen3ASR/Configuration.swift:47-108 (TextDecoderConfig), Qwen3ASR.swift:54-63 (token ids).
The authoritative copy used by the product is `qasr_default_config` in csrc/api.cpp; these
dataclasses only parameterise the synthetic-weight generator (qasr.synth).
"""
from dataclasses import dataclass


@dataclass(frozen=True)
class AudioEncoderConfig:
    d_model: int = 896
    heads: int = 14
    ffn_dim: int = 3584
    layers: int = 18
    n_mels: int = 128
    output_dim: int = 1024
    conv_channels: int = 480
    n_window: int = 50
    n_window_infer: int = 800
    ln_eps: float = 1e-5

    @property
    def conv_out_in(self):
        f = self.n_mels
        for _ in range(3):
            f = (f - 1) // 2 + 1
        return self.conv_channels * f


@dataclass(frozen=True)
class TextDecoderConfig:
    vocab: int = 151936
    hidden: int = 1024
    layers: int = 28
    heads: int = 16
    kv_heads: int = 8
    head_dim: int = 128
    inter: int = 3072
    rms_eps: float = 1e-6
    rope_theta: float = 1_000_000.0
    group_size: int = 64
    bits: int = 4


AUDIO_SMALL = AudioEncoderConfig()
AUDIO_LARGE = AudioEncoderConfig(d_model=1024, heads=16, ffn_dim=4096, layers=24, output_dim=2048)
# Qwen3AudioEncoderConfig.forcedAligner (AudioEncoder.swift:71-88): the large encoder projecting to the 1024-wide decoder
AUDIO_ALIGNER = AudioEncoderConfig(d_model=1024, heads=16, ffn_dim=4096, layers=24, output_dim=1024)
TEXT_SMALL = TextDecoderConfig()
TEXT_LARGE = TextDecoderConfig(hidden=2048, inter=6144)
AUDIO_TINY = AudioEncoderConfig(d_model=64, heads=2, ffn_dim=128, layers=2, output_dim=64, conv_channels=32,
                                n_window_infer=200)
TEXT_TINY = TextDecoderConfig(vocab=512, hidden=64, layers=2, heads=4, kv_heads=2, head_dim=32, inter=128)
