"""Mel oracle: shape contract of the reference tests (Qwen3ASRTests.swift:120-159), edge rules,
and agreement of the float32 restatement with a float64 re-derivation (<= 1e-4 abs)."""
import os
import numpy as np
import pytest
from conftest import GOLDEN
from oracle import mel
from oracle import mel as M
from qasr import synth


def test_shape_contract_1s():
    x = np.sin(2 * np.pi * 440 * np.arange(16000) / 16000).astype(np.float32)
    m = mel.log_mel(x)
    assert m.shape[0] == 128 and m.shape[1] > 90 and m.shape[1] == 100
    assert m.dtype == np.float32


@pytest.mark.parametrize("n,frames", [(80000, 500), (480000, 3000), (16000, 100), (400, 2), (1, 0), (159, 0), (160, 1)])
def test_frame_counts(n, frames):
    assert mel.num_mel_frames(n) == frames
    if 0 < frames <= 500:
        assert mel.log_mel(np.zeros(n, np.float32) + 0.01).shape == (128, frames)


def test_filterbank_properties():
    fb = mel.mel_filterbank()
    assert fb.shape == (128, 257) and fb.dtype == np.float32
    assert (fb >= 0).all() and (fb > 0).sum(0).max() <= 2           # <= 2 filters per bin
    nz = [(np.nonzero(r)[0].min(), np.nonzero(r)[0].max()) for r in fb]
    assert all(b[0] >= a[0] for a, b in zip(nz, nz[1:]))              # monotone centres
    assert fb[:, 0].sum() == 0 and fb[:, 256].sum() == 0              # DC / Nyquist get no weight


def test_f32_vs_f64():
    for k, sec in ((0, 1.0), (3, 5.0)):
        x = synth.synth_waveform(k, sec)
        a, b = mel.log_mel(x), mel.log_mel_f64(x)
        assert np.abs(a - b).max() < 1e-4


def test_fft_scale_is_a_constant_shift_above_the_floor():
    x = synth.synth_waveform(1, 1.0)
    a, b = mel.log_mel(x, fft_scale=2.0), mel.log_mel(x, fft_scale=1.0)
    assert np.abs((a - b) - np.log10(4.0) * 0.25).max() < 1e-5


def test_global_max_includes_dropped_frame():
    x = np.zeros(1600 + 160, np.float32) + 1e-4
    x[-100:] = 0.9                                  # energy only in the final (dropped) frame
    m, gmax = mel.log_mel(x, return_raw=True)
    assert m.shape[1] == mel.num_frames(x.shape[0]) - 1
    assert np.isclose(m.min(), (gmax - 8.0) * 0.25 + 1.0, atol=1e-6)   # clamp uses that max


def test_short_input_reflect_clamps():
    for n in (1, 2, 5, 199, 200, 201):
        p = mel.reflect_pad(np.arange(1, n + 1, dtype=np.float32))
        assert p.shape[0] == n + 400 and np.isfinite(p).all()
    with pytest.raises(ValueError):
        mel.log_mel(np.zeros(0, np.float32))


# ---- pinned by an independent implementation (tests/golden/hf_mel.npz, tests/golden/make_hf_goldens_mel.py) ------------
_G = np.load(os.path.join(GOLDEN, "hf_mel.npz"))


def test_window_and_filterbank_match_transformers_audio_utils():
    """R1: periodic Hann[400] and the slaney filterbank on the 512-point bin grid vs transformers.audio_utils
    (window_function / mel_filter_bank with the reference's parameters), float64 there, float32 here."""
    assert np.abs(M.hann_window() - _G["window"]).max() < 5e-7
    fb = M.mel_filterbank()
    assert fb.shape == (128, 257)
    assert np.abs(fb - _G["filterbank"].T).max() < 1.1e-6                     # 2.5e-5 of the largest weight (0.0424)


@pytest.mark.parametrize("name", ["synth", "speech", "short"])
def test_log_mel_matches_transformers_spectrogram(name):
    """R2 at the textbook FFT scale: reflect pad, 400-sample frames zero-padded to 512, power, mel, floor, log10 from the
    library's generic STFT (float64), then the reference's own tail in the reference's order -- max over ALL frames
    including the dropped one, clamp max - 8, * 0.25 + 1, drop the last frame (AudioPreprocessing.swift:283-296).
    What this does NOT pin: the Accelerate 2x FFT convention (fft_scale = 2.0), a constant +log10(4) before the clamp."""
    pcm, raw = _G["wave/" + name], _G["raw_log10/" + name].astype(np.float64)
    assert raw.shape == (128, M.num_frames(len(pcm)))
    want = (np.maximum(raw, raw.max() - 8.0) * 0.25 + 1.0)[:, :-1]
    got = M.log_mel(pcm, fft_scale=1.0)
    assert got.shape == want.shape
    err = np.abs(got - want).max()
    print(f"{name}: max |oracle - transformers| = {err:.2e}")
    assert err < 1e-4
    # and the 2x variant is that spectrum shifted by log10(4) before the clamp, wherever the 1e-10 floor is not active
    # (the speech slice starts in digital silence: floored bins do not shift)
    raw2 = np.where(raw > -10.0 + 1e-9, raw + np.log10(4.0), np.maximum(raw, -10.0))
    want2 = (np.maximum(raw2, raw2.max() - 8.0) * 0.25 + 1.0)[:, :-1]
    live = (raw > -9.0)[:, :-1]
    assert np.abs(M.log_mel(pcm, fft_scale=2.0) - want2)[live].max() < 1e-4
