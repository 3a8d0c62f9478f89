"""Log-mel front-end: CPU restatement of `WhisperFeatureExtractor`.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows Sources/Qwen3ASR/AudioPreprocessing.swift:
  * :39-53    periodic Hann[400]
  * :61-164   slaney mel filterbank on the 512-point bin grid (k * 16000 / 512, k = 0..256)
  * :169-317  reflect pad 200, frames of 400 hop 160 zero-padded to 512, real FFT, power,
              mel = P . FB^T, clip 1e-10, log10, per-clip max over ALL frames, clamp max-8,
              *0.25 + 1, drop last frame, cap 120000 frames, return [128, T]

All arithmetic is float32, like the reference (Swift `Float`).

vDSP scaling.  `vDSP_fft_zrip` (forward) returns 2x the mathematical DFT (Apple's documented
packing convention; not visible in the Swift source).  The reference squares that output
directly (:241-249), so its power spectrum is 4x the textbook one and every log10 value is
+log10(4) higher before the max-relative clamp.  `fft_scale=2.0` (default) reproduces the
reference as it actually runs on Apple hardware; `fft_scale=1.0` gives the textbook value.
This cannot be observed offline => mel numerics are "parity unpinned" at this boundary.
"""
import numpy as np

SAMPLE_RATE = 16000
N_FFT = 400
HOP = 160
N_MELS = 128
PADDED_FFT = 512
N_BINS = PADDED_FFT // 2 + 1
MAX_FRAMES = 1200 * SAMPLE_RATE // HOP   # 120000 (:304)
F32 = np.float32


def hann_window():
    """:41-44 -- 0.5 * (1 - cos(2*pi*i/400)), Float32."""
    i = np.arange(N_FFT, dtype=F32)
    two_pi = F32(2.0) * F32(np.pi)
    return (F32(0.5) * (F32(1.0) - np.cos(two_pi * i / F32(N_FFT), dtype=F32))).astype(F32)


def _hz_to_mel(hz):
    """:72-78 (scalar Float32)."""
    hz = F32(hz)
    if hz < F32(1000.0):
        return F32(3.0) * hz / F32(200.0)
    logstep = F32(27.0) / np.log(F32(6.4), dtype=F32)
    return F32(15.0) + np.log(hz / F32(1000.0), dtype=F32) * logstep


def _mel_to_hz(mel):
    """:80-86 (scalar Float32)."""
    mel = F32(mel)
    if mel < F32(15.0):
        return F32(200.0) * mel / F32(3.0)
    logstep = np.log(F32(6.4), dtype=F32) / F32(27.0)
    return F32(1000.0) * np.exp((mel - F32(15.0)) * logstep, dtype=F32)


def mel_filterbank():
    """:61-164 -> FB [128, 257] float32 (slaney scale + slaney norm, 512-point bin grid)."""
    fft_freqs = (np.arange(N_BINS, dtype=F32) * F32(SAMPLE_RATE) / F32(PADDED_FFT)).astype(F32)
    mel_min = _hz_to_mel(0.0)
    mel_max = _hz_to_mel(F32(SAMPLE_RATE) / F32(2.0))
    n_pts = N_MELS + 2
    mel_pts = np.array([mel_min + F32(i) * (mel_max - mel_min) / F32(n_pts - 1) for i in range(n_pts)],
                       dtype=F32)
    filt = np.array([_mel_to_hz(m) for m in mel_pts], dtype=F32)
    diff = (filt[1:] - filt[:-1]).astype(F32)
    fb = np.zeros((N_MELS, N_BINS), dtype=F32)
    for m in range(N_MELS):
        down = (fft_freqs - filt[m]) / diff[m]            # rising edge  (:136)
        up = (filt[m + 2] - fft_freqs) / diff[m + 1]      # falling edge (:137)
        tri = np.maximum(F32(0.0), np.minimum(down, up)).astype(F32)
        enorm = F32(2.0) / (filt[m + 2] - filt[m])        # slaney norm (:149)
        fb[m] = tri * enorm
    return fb


def reflect_pad(audio):
    """:173-192 -- including the reference's index clamps for very short inputs."""
    n = audio.shape[0]
    pad = N_FFT // 2
    out = np.zeros(pad + n + pad, dtype=F32)
    for i in range(pad):
        src = min(pad - i, n - 1)
        out[i] = audio[max(0, src)]
    out[pad:pad + n] = audio
    for i in range(pad):
        src = n - 2 - i
        out[pad + n + i] = audio[max(0, src)]
    return out


def num_frames(n_samples):
    """:195 before the last-frame drop."""
    return (n_samples + 2 * (N_FFT // 2) - N_FFT) // HOP + 1


def num_mel_frames(n_samples):
    """Frames returned to the encoder: drop-last (:296) then cap (:304)."""
    return min(num_frames(n_samples) - 1, MAX_FRAMES)


def log_mel(audio, fft_scale=2.0, _fb=None, _win=None, return_raw=False):
    """`extractFeatures` (:169-317): float32 PCM at 16 kHz -> [128, T] float32."""
    audio = np.ascontiguousarray(audio, dtype=F32)
    if audio.ndim != 1 or audio.shape[0] == 0:
        raise ValueError("log_mel: need a non-empty mono float32 buffer")
    fb = mel_filterbank() if _fb is None else _fb
    win = hann_window() if _win is None else _win
    padded = reflect_pad(audio)
    nf = (padded.shape[0] - N_FFT) // HOP + 1
    idx = (np.arange(nf)[:, None] * HOP + np.arange(N_FFT)[None, :])
    frames = np.zeros((nf, PADDED_FFT), dtype=F32)
    frames[:, :N_FFT] = padded[idx] * win[None, :]                      # vDSP_vmul (:214)
    spec = np.fft.rfft(frames, axis=1)                                  # complex64
    assert spec.dtype == np.complex64
    s = F32(fft_scale)
    re = (spec.real * s).astype(F32)
    im = (spec.imag * s).astype(F32)
    power = (re * re + im * im).astype(F32)                             # :241-249
    mel = (power @ fb.T).astype(F32)                                    # vDSP_mmul (:267)
    mel = np.maximum(mel, F32(1e-10))                                   # :276
    logm = np.log10(mel, dtype=F32)                                     # :279
    gmax = logm.max()                                                   # :283 (all frames)
    logm = np.maximum(logm, gmax - F32(8.0))                            # :286-288
    out = (logm * F32(0.25) + F32(1.0)).astype(F32)                     # :291-293
    out = out[:nf - 1]                                                  # :296
    if out.shape[0] > MAX_FRAMES:                                       # :304-313
        out = out[:MAX_FRAMES]
    res = np.ascontiguousarray(out.T)                                   # [128, T]  (:315-316)
    if return_raw:
        return res, gmax
    return res


def log_mel_f64(audio, fft_scale=2.0):
    """Float64 re-derivation of the same pipeline (error yardstick for the f32 paths)."""
    audio = np.asarray(audio, dtype=np.float64)
    n = audio.shape[0]
    pad = N_FFT // 2
    padded = reflect_pad(audio.astype(F32)).astype(np.float64)
    i = np.arange(N_FFT)
    win = 0.5 * (1.0 - np.cos(2.0 * np.pi * i / N_FFT))
    nf = (padded.shape[0] - N_FFT) // HOP + 1
    idx = (np.arange(nf)[:, None] * HOP + i[None, :])
    frames = np.zeros((nf, PADDED_FFT))
    frames[:, :N_FFT] = padded[idx] * win[None, :]
    spec = np.fft.rfft(frames, axis=1) * fft_scale
    power = spec.real ** 2 + spec.imag ** 2
    fb = mel_filterbank().astype(np.float64)
    mel = np.maximum(power @ fb.T, 1e-10)
    logm = np.log10(mel)
    logm = np.maximum(logm, logm.max() - 8.0)
    out = (logm * 0.25 + 1.0)[:nf - 1][:MAX_FRAMES]
    return np.ascontiguousarray(out.T)
