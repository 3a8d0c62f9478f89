"""The one golden the reference holds on the floating-point path: the exact transcript of its own test clip.

  Tests/Qwen3ASRTests/E2EQwen3ASRGreedyDeterminismTests.swift:49-60
      model  aufklarer/Qwen3-ASR-0.6B-MLX-4bit, audio Tests/Qwen3ASRTests/Resources/test_audio.wav (24 kHz, here as
      tests/golden/test_audio.wav), expected text "Can you guarantee that the replacement part will be shipped tomorrow?"
  Tests/Qwen3ASRTests/Qwen3ASRIntegrationTests.swift:136-140,255-259,294-298
      the same clip contains {guarantee, replacement, shipped, tomorrow} for 0.6B-4bit, 0.6B-8bit and 1.7B-8bit.

No checkpoint exists offline, so the test is skipped unless QASR_CKPT_DIR points at a local copy
of the checkpoint directory (model-*.safetensors, vocab.json, merges.txt, tokenizer_config.json).  When it runs it settles
what nothing else can: the vDSP 2x FFT scaling (`qasr_config.fft_scale` 2.0 vs 1.0: both are tried, the snapshot must hold
for 2.0 and the result of 1.0 is printed), the mlx packing order of the 4-bit words, and the bf16-MFMA encoder deviation.
The harness resamples 24 kHz -> 16 kHz with scipy (the reference uses AVAudioConverter: closed source, out of scope), so a
sample-exact match of the input is not claimed -- only the transcript is."""
import json
import os
import numpy as np
import pytest
from conftest import GOLDEN
from qasr.model import Qwen3ASRModel, load_wav

pytestmark = pytest.mark.gpu
KAT = json.load(open(os.path.join(GOLDEN, "kat_reference_tests.json"), encoding="utf-8"))
CKPT = os.environ.get("QASR_CKPT_DIR")
MODEL_ID = os.environ.get("QASR_CKPT_ID", "aufklarer/Qwen3-ASR-0.6B-MLX-4bit")
SNAPSHOT = "Can you guarantee that the replacement part will be shipped tomorrow?"
KEYWORDS = ("guarantee", "replacement", "shipped", "tomorrow")


def test_snapshot_string_is_the_reference_one():
    """The expected text is data copied from the reference's test, not something this repo made up."""
    flat = json.dumps(KAT)
    assert SNAPSHOT in flat


@pytest.mark.skipif(not CKPT or not os.path.isdir(CKPT or ""),
                    reason="set QASR_CKPT_DIR to a local copy of aufklarer/Qwen3-ASR-0.6B-MLX-4bit (no network here): "
                           "the reference's transcript snapshot is the only golden that pins mel scaling / mlx packing")
def test_reference_transcript_snapshot():
    from scipy.signal import resample_poly
    pcm24, rate = load_wav(os.path.join(GOLDEN, "test_audio.wav"))
    assert rate == 24000
    pcm = resample_poly(pcm24.astype(np.float64), 2, 3).astype(np.float32)
    results = {}
    for fft_scale in (2.0, 1.0):
        m = Qwen3ASRModel.from_pretrained(CKPT, model_id=MODEL_ID, max_batch=1, max_audio_seconds=30, fft_scale=fft_scale)
        try:
            a = m.transcribe(pcm, sample_rate=16000)
            b = m.transcribe(pcm, sample_rate=16000)
            assert a == b, "greedy decoding must be deterministic across calls (E2EQwen3ASRGreedyDeterminismTests.swift:31-38)"
            results[fft_scale] = a
        finally:
            m.close()
    print("transcripts by fft_scale:", results)
    low = results[2.0].lower()
    assert all(k in low for k in KEYWORDS), results
    if "0.6B" in MODEL_ID and "4bit" in MODEL_ID:
        assert results[2.0] == SNAPSHOT, results
