"""Pins ONE constant of the Qwen3 log-mel -- `fft_scale = 2.0`, "vDSP_fft_zrip returns twice the DFT" -- with a numeric fixture the
reference itself holds (CPU only; see oracle/kaldi_fbank.py for the argument).

tests/golden/kaldi_fbank_input.wav / kaldi_fbank_reference.bin are the reference's own test resources
(Tests/SpeechWakeWordTests/Resources/fbank_input.wav, fbank_reference.bin: data, generated there by kaldi-native-fbank), compared at
the reference's own bars (SpeechWakeWordTests.swift:204-233: max |d| < 3e-3, mean |d| < 5e-5)."""
import os
import wave
import numpy as np
from conftest import GOLDEN
from oracle import kaldi_fbank as KF, mel as omel


def _fixture():
    with wave.open(os.path.join(GOLDEN, "kaldi_fbank_input.wav")) as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate()) == (1, 2, 16000)
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16).astype(np.float32) / np.float32(32768.0)
    raw = open(os.path.join(GOLDEN, "kaldi_fbank_reference.bin"), "rb").read()
    frames, bins = np.frombuffer(raw[:8], dtype=np.int32)
    ref = np.frombuffer(raw[8:], dtype=np.float32).reshape(frames, bins)
    return pcm, ref


def test_restatement_on_a_1x_fft_without_the_half_matches_kaldi_native_fbank():
    pcm, ref = _fixture()
    assert ref.shape == (100, 80) and KF.num_frames(pcm.shape[0]) == 100
    got = KF.compute(pcm)                                 # NumPy rfft = 1x the DFT, no 0.5
    d = np.abs(got - ref)
    print(f"kaldi fbank restatement vs the reference's fixture: max |d| {d.max():.2e}, mean |d| {d.mean():.2e}")
    assert d.max() < 3e-3 and d.mean() < 5e-5             # the reference's own bars


def test_the_reference_half_needs_a_2x_routine():
    """The reference multiplies every bin by 0.5 before squaring (KaldiFbank.swift:239-245).  On a routine that returns the
    DFT itself that puts every unfloored value ln 4 below the fixture -- hundreds of times the reference's own bar; on a routine that
    returns twice the DFT it reproduces the fixture.  So the reference's test can only pass on a 2x routine: that IS vDSP_fft_zrip."""
    pcm, ref = _fixture()
    on_1x = KF.compute(pcm, fft_gain=1.0, bin_scale=0.5)
    live = ref > np.log(KF.LOG_FLOOR) + 2.0               # values the energy floor does not touch
    assert live.mean() > 0.9
    shift = (ref - on_1x)[live]
    assert np.abs(shift - np.log(4.0)).max() < 3e-3 and np.abs(ref - on_1x).max() > 1.0
    on_2x = KF.compute(pcm, fft_gain=2.0, bin_scale=0.5)
    assert np.abs(on_2x - ref).max() < 3e-3


def test_qwen3_front_end_without_the_half_sees_four_times_the_power():
    """AudioPreprocessing.swift:241-249 squares the same routine's output with NO 0.5: the power is 4x the textbook one, which is what
    `fft_scale = 2.0` means in oracle/mel.py and csrc/mel.hip -- the per-clip maximum of the raw log10 mel sits log10(4) above the 1x
    value; after the max-relative clamp and the affine map the shift survives only where the 1e-10 floor bites (silence), so the two
    outputs agree on this speech-level clip and differ by at most log10(4) / 4 anywhere."""
    pcm, _ = _fixture()
    (m2, g2), (m1, g1) = omel.log_mel(pcm, fft_scale=2.0, return_raw=True), omel.log_mel(pcm, fft_scale=1.0, return_raw=True)
    assert abs(float(g2 - g1) - np.log10(4.0)) < 1e-5
    assert m2.shape == m1.shape and np.abs(m2 - m1).max() <= np.log10(4.0) / 4.0 + 1e-5
    quiet = np.concatenate([pcm[:4000], np.zeros(8000, dtype=np.float32)])
    q2, q1 = omel.log_mel(quiet, fft_scale=2.0), omel.log_mel(quiet, fft_scale=1.0)
    assert np.abs(q2 - q1).max() > 0.1                    # the floor does not move with the scale: digital silence tells the two apart
