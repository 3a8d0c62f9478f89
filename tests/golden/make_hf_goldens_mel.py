#!/usr/bin/env python3
"""Golden vectors for the log-mel front-end from an INDEPENDENT implementation: `transformers.audio_utils`
(`window_function`, `mel_filter_bank`, `spectrogram`; transformers 5.x in the build container) configured with the
REFERENCE's definition -- periodic Hann[400], frames zero-padded to a 512-point FFT, hop 160, reflect padding, power
spectrum, slaney-scale / slaney-norm filterbank on the 257-bin grid k * 16000 / 512, floor 1e-10, log10
(Sources/Qwen3ASR/AudioPreprocessing.swift:39-53,61-164,169-280).  This is NOT WhisperFeatureExtractor's 400-point
definition; it is the library's generic STFT/mel code driven by the reference's parameters, in float64.

Stored: the waveforms, the window, the filterbank and the RAW log10 mel spectrogram of every frame (before the reference's
max-relative clamp / affine / drop-last, which tests/test_oracle_mel.py applies in the reference's order).  The textbook FFT
scale is what a library produces: these goldens pin oracle/mel.py at fft_scale = 1.0; the Accelerate 2x convention
(fft_scale = 2.0) remains unobservable offline.

Run from the repo root in the build container:  python tests/golden/make_hf_goldens_mel.py
"""
import os
import sys
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "qwen3-asr-swift_amd"))
from transformers import audio_utils as au   # noqa: E402
from qasr import synth                       # noqa: E402


def fixture_slice(seconds=2.5, start=4.6):
    """a slice of the reference's speech fixture, decimated 24 kHz -> 16 kHz by linear interpolation (any 16 kHz signal
    will do: both sides consume the same samples)."""
    with wave.open(os.path.join(ROOT, "tests", "golden", "test_audio.wav"), "rb") as w:
        sr, n = w.getframerate(), w.getnframes()
        pcm = np.frombuffer(w.readframes(n), dtype=np.int16).astype(np.float32) / 32768.0
        if w.getnchannels() > 1:
            pcm = pcm.reshape(-1, w.getnchannels()).mean(1)
    t = np.arange(int(seconds * 16000)) / 16000.0 + start
    return np.interp(t * sr, np.arange(len(pcm)), pcm).astype(np.float32)


def main():
    win = au.window_function(400, "hann", periodic=True)
    fb = au.mel_filter_bank(257, 128, 0.0, 8000.0, 16000, norm="slaney", mel_scale="slaney")      # [257, 128]
    out = {"window": win.astype(np.float64), "filterbank": fb.astype(np.float64)}
    waves = {"synth": synth.synth_waveform(3, 2.5), "speech": fixture_slice(), "short": synth.synth_waveform(5, 0.031)}
    for name, pcm in waves.items():
        raw = au.spectrogram(pcm.astype(np.float64), win, frame_length=400, hop_length=160, fft_length=512, power=2.0,
                             center=True, pad_mode="reflect", mel_filters=fb, mel_floor=1e-10, log_mel="log10",
                             dtype=np.float64)
        out["wave/" + name] = pcm
        out["raw_log10/" + name] = raw.astype(np.float32)
        print(name, pcm.shape, raw.shape)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "hf_mel.npz"), **out)


if __name__ == "__main__":
    main()
