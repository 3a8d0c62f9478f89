"""Model presets, restated from the reference's compile-time Swift structs.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows:
  * Sources/Qwen3ASR/AudioEncoder.swift:28-88   (Qwen3AudioEncoderConfig .small/.large)
  * Sources/Qwen3ASR/Configuration.swift:47-108 (TextDecoderConfig .small/.large)
  * Sources/Qwen3ASR/Qwen3ASR.swift:54-63,181-193 (special token ids)
"""
from dataclasses import dataclass, replace


@dataclass(frozen=True)
class AudioEncoderConfig:
    d_model: int = 896
    heads: int = 14
    ffn_dim: int = 3584
    layers: int = 18
    n_mels: int = 128
    output_dim: int = 1024
    conv_channels: int = 480          # downsampleHiddenSize
    n_window: int = 50                # chunk = 2*n_window = 100 mel frames
    n_window_infer: int = 800
    ln_eps: float = 1e-5

    @property
    def chunk(self):
        return 2 * self.n_window

    @property
    def freq_after_conv(self):        # 128 -> 64 -> 32 -> 16
        f = self.n_mels
        for _ in range(3):
            f = (f - 1) // 2 + 1
        return f

    @property
    def conv_out_in(self):            # 7680
        return self.conv_channels * self.freq_after_conv


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
TEXT_SMALL_8BIT = replace(TEXT_SMALL, bits=8)
TEXT_LARGE = TextDecoderConfig(hidden=2048, inter=6144)
TEXT_LARGE_8BIT = replace(TEXT_LARGE, bits=8)

# A tiny geometry that keeps every structural rule (GQA 2:1, 3 convs, 2 windows ...) but
# runs in milliseconds on CPU.  Not a reference preset: used for goldens and fast parity.
AUDIO_TINY = AudioEncoderConfig(d_model=64, heads=2, ffn_dim=128, layers=2, output_dim=64, n_window_infer=200,
                                conv_channels=32)
TEXT_TINY = TextDecoderConfig(vocab=512, hidden=64, layers=2, heads=4, kv_heads=2,
                              head_dim=32, inter=128)


@dataclass(frozen=True)
class TokenIds:
    """Sources/Qwen3ASR/Qwen3ASR.swift:54-63 and :182-193 (defaults = reference values)."""
    im_start: int = 151644
    im_end: int = 151645          # also EOS (Qwen3ASRTokens.eosTokenId)
    audio_start: int = 151669
    audio_end: int = 151670
    audio_pad: int = 151676
    asr_text: int = 151704
    newline: int = 198
    system: int = 8948
    user: int = 872
    assistant: int = 77091

    @property
    def eos(self):
        return self.im_end


TOKENS = TokenIds()
# Tiny-vocab remap so the tiny geometry can run the same prompt template.
TOKENS_TINY = TokenIds(im_start=500, im_end=501, audio_start=502, audio_end=503, audio_pad=504,
                       asr_text=505, newline=198, system=300, user=301, assistant=302)


def detect_size(model_id: str) -> str:
    """Qwen3ASR.swift:581-586."""
    return "large" if ("1.7B" in model_id or "1.7b" in model_id) else "small"


def detect_bits(model_id: str) -> int:
    """Qwen3ASR.swift:590-601."""
    lower = model_id.lower()
    if "8bit" in lower or "8-bit" in lower:
        return 8
    if "4bit" in lower or "4-bit" in lower:
        return 4
    return 8 if detect_size(model_id) == "large" else 4
