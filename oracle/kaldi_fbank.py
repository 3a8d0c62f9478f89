"""Kaldi-compatible fbank: CPU restatement of `KaldiFbank.compute` (wake-word front end).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The wake-word model is NOT part of this build; this file exists for ONE
constant of the Qwen3-ASR hot path.

Why it is here.  The Qwen3 log-mel (`oracle/mel.py`, `csrc/mel.hip`) carries `fft_scale = 2.0`: Apple's `vDSP_fft_zrip` returns
twice the mathematical DFT, `AudioPreprocessing.swift:241-249` squares that output as it stands, so the reference's power spectrum
is 4x the textbook one.  Nothing in the reference's Qwen3 tests observes that factor (shape-only tests).  `KaldiFbank` makes the
SAME call (`vDSP_create_fftsetup(log2n, kFFTRadix2)` + `vDSP_fft_zrip(..., kFFTDirection_Forward)`,
Sources/SpeechWakeWord/KaldiFbank.swift:79,228-229), then multiplies every bin by `scale = 0.5` ("vDSP returns 2x DFT", :231-245) --
and the reference holds a NUMERIC fixture for the result: Tests/SpeechWakeWordTests/Resources/fbank_reference.bin, produced by
kaldi-native-fbank from fbank_input.wav, asserted at max |d| < 3e-3 and mean |d| < 5e-5
(Tests/SpeechWakeWordTests/SpeechWakeWordTests.swift:204-233).  Restated here on NumPy's 1x `rfft` WITHOUT the 0.5: if this matches the
fixture at the reference's own bars (tests/test_oracle_kaldi_fbank.py), then on a 1x routine the reference's 0.5 would put every
bin ln 4 = 1.386 below the fixture -- 460x its own bar -- so the routine the reference runs on IS a 2x routine, and the Qwen3 path,
which omits the 0.5, sees 4x the textbook power.

Follows Sources/SpeechWakeWord/KaldiFbank.swift:
  * :27-58    options: 16 kHz, 25 ms / 10 ms frames (400 / 160), 80 mel bins, 20 Hz .. Nyquist - 400 Hz, preemphasis 0.97,
              DC removal, snip_edges = false, power spectrum
  * :85-93    Povey window  pow(0.5 - 0.5 cos(2 pi i / (N - 1)), 0.85)  (f64 arithmetic, stored as Float)
  * :110-118  frame count  floor((n + shift / 2) / shift)
  * :155-188  window extraction: first sample f * shift - (length - shift) / 2, out-of-range indices mirrored (-i - 1, 2 n - i - 1)
  * :192-213  per-frame mean removal, preemphasis from the end (frame[0] -= 0.97 frame[0]), window
  * :217-251  512-point real FFT, power (see above)
  * :255-312  triangles in mel = 1127 ln(1 + hz / 700), linear in MEL between left / centre / right, no normalisation (f64 -> Float)
  * :139-147  log(max(mel energy, FLT_EPSILON))
Float32 wherever the reference computes in `Float`.
"""
import numpy as np

F32 = np.float32
SAMPLE_RATE, FRAME_LEN, FRAME_SHIFT, N_MELS, PADDED = 16000, 400, 160, 80, 512
N_BINS = PADDED // 2 + 1
LOW_FREQ, HIGH_FREQ, PREEMPH = 20.0, -400.0, 0.97
LOG_FLOOR = np.finfo(np.float32).eps          # Float.ulpOfOne (:71)


def povey_window():
    i = np.arange(FRAME_LEN, dtype=np.float64)
    raw = 0.5 - 0.5 * np.cos(2.0 * np.pi * i / float(FRAME_LEN - 1))
    return np.power(raw, 0.85).astype(F32)


def mel_filterbank():
    nyquist = SAMPLE_RATE / 2.0
    high = nyquist + HIGH_FREQ if HIGH_FREQ < 0 else HIGH_FREQ
    hz_to_mel = lambda hz: 1127.0 * np.log1p(hz / 700.0)
    mel_low, mel_high = hz_to_mel(LOW_FREQ), hz_to_mel(high)
    delta = (mel_high - mel_low) / float(N_MELS + 1)
    mel = hz_to_mel(np.arange(N_BINS, dtype=np.float64) * SAMPLE_RATE / PADDED)
    fb = np.zeros((N_MELS, N_BINS), dtype=np.float64)
    for m in range(N_MELS):
        left = mel_low + m * delta
        centre, right = left + delta, left + 2 * delta
        inside = (mel > left) & (mel < right)
        up = (mel - left) / (centre - left)
        down = (right - mel) / (right - centre)
        fb[m] = np.where(inside, np.where(mel <= centre, up, down), 0.0)
    return fb.astype(F32)


def num_frames(n):
    return 0 if n == 0 else int((n + FRAME_SHIFT / 2.0) / FRAME_SHIFT)


def _mirror(i, total):
    while i < 0 or i >= total:
        if i < 0:
            i = -i - 1
        if i >= total:
            i = 2 * total - i - 1
    return i


def compute(samples, fft_gain=1.0, bin_scale=1.0):
    """[frames, 80] log mel energies.  fft_gain = what the FFT routine returns relative to the mathematical DFT (NumPy: 1; vDSP's
    zrip: 2), bin_scale = the factor the caller applies to every bin before squaring (the reference: 0.5).  The product is what
    reaches the square; the defaults restate "the reference on its 2x routine" as 2 x 0.5 = 1."""
    x = np.asarray(samples, dtype=F32)
    n = x.shape[0]
    T = num_frames(n)
    win, fb = povey_window(), mel_filterbank()
    out = np.zeros((T, N_MELS), dtype=F32)
    g = F32(fft_gain) * F32(bin_scale)
    for f in range(T):
        start = f * FRAME_SHIFT - (FRAME_LEN - FRAME_SHIFT) // 2
        idx = np.arange(start, start + FRAME_LEN)
        bad = (idx < 0) | (idx >= n)
        if bad.any():
            idx = np.array([_mirror(int(i), n) if b else int(i) for i, b in zip(idx, bad)])
        frame = x[idx].astype(F32)
        frame = frame - F32(frame.sum(dtype=F32) / F32(FRAME_LEN))              # vDSP_meanv + vsadd
        pre = frame.copy()
        pre[1:] = frame[1:] - F32(PREEMPH) * frame[:-1]                         # from the end: every y[i] uses the original x[i - 1]
        pre[0] = frame[0] - F32(PREEMPH) * frame[0]
        pre = (pre * win).astype(F32)
        spec = np.fft.rfft(np.concatenate([pre, np.zeros(PADDED - FRAME_LEN, dtype=F32)]).astype(np.float64))
        re, im = (spec.real.astype(F32) * g).astype(F32), (spec.imag.astype(F32) * g).astype(F32)
        power = (re * re + im * im).astype(F32)
        mels = (fb @ power).astype(F32)
        out[f] = np.log(np.maximum(mels, F32(LOG_FLOOR)), dtype=F32)
    return out
