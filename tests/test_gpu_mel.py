"""HIP mel front-end vs the CPU oracle (R1-R3 of SURVEY.md section 8a), through the C ABI.

Tolerance: 1e-4 abs in normalised log-mel units vs the float32 oracle (values are O(1); both
sides are float32 with different FFT factorizations; the f64 re-derivation bounds each at ~2e-6
on speech-like input, the 1e-4 bar leaves room for the log10 of near-floor bins)."""
import numpy as np
import pytest
from oracle import mel as omel
from qasr import synth
import gpu_util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = gpu_util.Engine("tiny", max_audio_seconds=30)
    yield e
    e.close()


@pytest.mark.parametrize("k,sec", [(0, 1.0), (1, 5.0), (2, 30.0), (3, 0.37)])
def test_mel_matches_oracle(eng, k, sec):
    x = synth.synth_waveform(k, sec)
    got = eng.mel(x)
    ref = omel.log_mel(x)
    assert got.shape == ref.shape
    err = np.abs(got - ref)
    assert err.max() < 1e-4, (err.max(), np.unravel_index(err.argmax(), err.shape))


@pytest.mark.parametrize("n", [1, 2, 159, 160, 161, 200, 201, 399, 400, 401, 1000, 16001])
def test_mel_ragged_short(eng, n):
    rng = np.random.default_rng(n)
    x = (0.1 * rng.standard_normal(n)).astype(np.float32)
    got = eng.mel(x)
    ref = omel.log_mel(x)
    assert got.shape == ref.shape
    if ref.size:
        assert np.abs(got - ref).max() < 1e-4


def test_mel_silence_and_dropped_frame_max(eng):
    x = np.zeros(16000, np.float32)
    assert np.abs(eng.mel(x) - omel.log_mel(x)).max() < 1e-6         # all bins at the 1e-10 floor
    y = np.zeros(1760, np.float32) + 1e-4
    y[-100:] = 0.9                                                    # max lives in the dropped frame
    assert np.abs(eng.mel(y) - omel.log_mel(y)).max() < 1e-4


@pytest.mark.parametrize("name", ["synth", "speech", "short"])
def test_device_mel_vs_independent_golden(name):
    """The DEVICE mel against tests/golden/hf_mel.npz directly (transformers.audio_utils' STFT + slaney filterbank driven by the
    reference's 512-point definition, float64; see tests/test_oracle_mel.py): engine at fft_scale = 1.0, the reference's tail
    (max over all frames incl. the dropped one, clamp, affine, drop-last) applied to the golden's raw log10 spectrogram."""
    import os
    from conftest import GOLDEN
    G = np.load(os.path.join(GOLDEN, "hf_mel.npz"))
    pcm, raw = G["wave/" + name], G["raw_log10/" + name].astype(np.float64)
    want = (np.maximum(raw, raw.max() - 8.0) * 0.25 + 1.0)[:, :-1]
    e = gpu_util.Engine("tiny", max_audio_seconds=4, fft_scale=1.0)
    try:
        got = e.mel(pcm)
    finally:
        e.close()
    assert got.shape == want.shape
    err = float(np.abs(got - want).max())
    print(f"{name}: max |device - transformers| = {err:.2e}")
    assert err < 1e-4
