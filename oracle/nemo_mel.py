"""NeMo-style log-mel front-ends of the Parakeet / Nemotron models: CPU restatement (BASELINE configs[4], restatable slice).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The encoder / prediction network / joint of these models are opaque CoreML bundles (`.mlmodelc`): nothing of their arithmetic is
in the reference's tree, so only the Swift signal processing in front of them and the greedy loops behind them (oracle/transducer.py)
can be restated.  Follows, function by function:

  * `Sources/ParakeetASR/MelPreprocessor.swift:52-202`            extract (Parakeet-TDT batch model)        -> extract(variant="tdt")
  * `Sources/ParakeetStreamingASR/StreamingMelPreprocessor.swift:62-186`   extract (EOU 120M)              -> extract(variant="eou")
  * `...ParakeetStreamingASR/StreamingMelPreprocessor.swift:193-273`,
    `Sources/NemotronStreamingASR/StreamingMelPreprocessor.swift:55-129`   extractRaw (the 160 ms Nemotron streamer and EOU)  -> extract_raw
  * `...ParakeetStreamingASR/StreamingMelPreprocessor.swift:280-393`       extractStreaming (running mean / std) -> StreamingMel
  * filterbank `MelPreprocessor.swift:207-266` (the three copies in the reference are the same code)       -> mel_filterbank

Common pipeline: pre-emphasis 0.97 -> centre padding 256 (reflect | zeros) -> frames of 512 @ hop 160 with a Hann[400] window at
offset 0 (left-aligned, periodic or symmetric) or 56 (torch.stft's centred placement) -> real FFT 512 -> power -> slaney mel 128 ->
ln(x + 2^-24) -> [per-feature normalisation over the valid frames] -> [128, nFrames], nFrames = n / 160 + 1, melLength = n / 160.

vDSP scaling: `vDSP_fft_zrip` returns 2x the DFT.  extract / extractStreaming square it as is (power = 4 |X|^2: a constant ln 4 that
the per-feature mean removes, up to the 2^-24 guard) -> `fft_scale` (default 2.0) as in oracle/mel.py.  extractRaw multiplies the
power by 0.25 and SAYS why (`NemotronStreamingASR/StreamingMelPreprocessor.swift:100`: "vDSP_fft_zrip scales 2x vs torch.stft --
divide power by 4"): its result is the textbook |X|^2 exactly (scaling by powers of two is exact in binary floating point), so
extract_raw has no free constant -- and that comment is the reference's own statement of the 2x convention oracle/mel.py assumes.
"""
import numpy as np

F32 = np.float32
SAMPLE_RATE, N_FFT, HOP, WIN, N_MELS, N_BINS, PAD = 16000, 512, 160, 400, 128, 257, 256
PREEMPH = F32(0.97)
LOG_GUARD = F32(5.960464477539063e-08)      # 2^-24 (MelPreprocessor.swift:18)


def hann_window(periodic):
    """MelPreprocessor.swift:27-31 (periodic: / N) | StreamingMelPreprocessor.swift:31-38 (symmetric: / (N - 1)); Float32."""
    i = np.arange(WIN, dtype=F32)
    den = F32(WIN if periodic else WIN - 1)
    return (F32(0.5) * (F32(1.0) - np.cos(F32(2.0) * F32(np.pi) * i / den, dtype=F32))).astype(F32)


def _hz_to_mel(hz):
    hz = F32(hz)
    if hz < F32(1000.0):
        return F32(3.0) * hz / F32(200.0)
    return F32(15.0) + np.log(hz / F32(1000.0), dtype=F32) * (F32(27.0) / np.log(F32(6.4), dtype=F32))


def _mel_to_hz(mel):
    mel = F32(mel)
    if mel < F32(15.0):
        return F32(200.0) * mel / F32(3.0)
    return F32(1000.0) * np.exp((mel - F32(15.0)) * (np.log(F32(6.4), dtype=F32) / F32(27.0)), dtype=F32)


def mel_filterbank():
    """MelPreprocessor.swift:207-266 -> [128, 257] float32: slaney scale, triangles from the two slopes, slaney norm."""
    fft_freqs = (np.arange(N_BINS, dtype=F32) * F32(SAMPLE_RATE) / F32(N_FFT)).astype(F32)
    mel_min, mel_max = _hz_to_mel(0.0), _hz_to_mel(F32(SAMPLE_RATE) / F32(2.0))
    npts = N_MELS + 2
    pts = np.array([mel_min + F32(i) * (mel_max - mel_min) / F32(npts - 1) for i in range(npts)], dtype=F32)
    filt = np.array([_mel_to_hz(m) for m in pts], dtype=F32)
    diff = (filt[1:] - filt[:-1]).astype(F32)
    fb = np.zeros((N_MELS, N_BINS), dtype=F32)
    for m in range(N_MELS):
        down = (fft_freqs - filt[m]) / diff[m]
        up = (filt[m + 2] - fft_freqs) / diff[m + 1]
        fb[m] = np.maximum(F32(0.0), np.minimum(down, up)).astype(F32) * (F32(2.0) / (filt[m + 2] - filt[m]))
    return fb


def preemphasis(audio):
    """x[n] - 0.97 x[n-1], first sample kept (MelPreprocessor.swift:54-65: vDSP_vsma = a * b + c)."""
    a = np.ascontiguousarray(audio, dtype=F32)
    out = a.copy()
    out[1:] = (a[:-1] * (-PREEMPH) + a[1:]).astype(F32)
    return out


def centre_pad(pre, reflect):
    """reflect: MelPreprocessor.swift:68-83 (index clamp `max(0, n - 2 - i)` on the right; the left side indexes pre[256 - i], which
    needs n > 256 -- the Swift code traps below that); zeros: StreamingMelPreprocessor.swift (Nemotron) :74-77."""
    n = pre.shape[0]
    out = np.zeros(PAD + n + PAD, dtype=F32)
    out[PAD:PAD + n] = pre
    if reflect:
        if n <= PAD:
            raise ValueError("reflect padding needs more than 256 samples (the reference indexes out of bounds)")
        for i in range(PAD):
            out[i] = pre[PAD - i]
            out[PAD + n + i] = pre[max(0, n - 2 - i)]
    return out


def num_frames(n):
    return (n + 2 * PAD - N_FFT) // HOP + 1          # = n // 160 + 1


def _log_mel_frames(audio, periodic, reflect, centred, power_scale):
    """-> ln(mel + guard) [nFrames, 128] float32 and melLength."""
    pre = preemphasis(audio)
    padded = centre_pad(pre, reflect)
    nf = (padded.shape[0] - N_FFT) // HOP + 1
    off = (N_FFT - WIN) // 2 if centred else 0
    idx = np.arange(nf)[:, None] * HOP + off + np.arange(WIN)[None, :]
    frames = np.zeros((nf, N_FFT), dtype=F32)
    frames[:, off:off + WIN] = padded[idx] * hann_window(periodic)[None, :]
    spec = np.fft.rfft(frames, axis=1)
    re, im = spec.real.astype(F32), spec.imag.astype(F32)
    power = ((re * re + im * im) * F32(power_scale)).astype(F32)
    mel = (power @ mel_filterbank().T).astype(F32)
    return np.log(mel + LOG_GUARD, dtype=F32), audio.shape[0] // HOP


def extract(audio, variant="tdt", fft_scale=2.0):
    """Per-utterance, per-feature normalised log-mel.  variant "tdt": ParakeetASR/MelPreprocessor.swift:52-202 (periodic Hann, float16
    output, std over melLength - 1); "eou": ParakeetStreamingASR/StreamingMelPreprocessor.swift:62-186 (symmetric Hann, float32
    output, `max(melLength - 1, 1)`).  -> ([128, nFrames], melLength); frames >= melLength are zero."""
    audio = np.ascontiguousarray(audio, dtype=F32)
    if audio.shape[0] == 0:
        if variant == "eou":
            return np.zeros((N_MELS, 1), dtype=F32), 0          # :63-66
        raise ValueError("extract: empty audio (the reference traps on audio[0])")
    logm, L = _log_mel_frames(audio, periodic=(variant == "tdt"), reflect=True, centred=False, power_scale=F32(fft_scale) * F32(fft_scale))
    x = logm.T.copy()                                           # [128, nFrames]
    nf = x.shape[1]
    out = np.zeros_like(x)
    for b in range(N_MELS):
        v = x[b, :L]
        mean = F32(v.sum(dtype=F32) / F32(L)) if L else F32(0)
        c = (v - mean).astype(F32)
        meansq = F32((c * c).sum(dtype=F32) / F32(L)) if L else F32(0)
        den = F32(L - 1) if variant == "tdt" else F32(max(L - 1, 1))
        with np.errstate(divide="ignore", invalid="ignore"):
            std = np.sqrt(F32(L) * meansq / den, dtype=F32)
            out[b, :L] = c * (F32(1.0) / (std + F32(1e-5)))
    if variant == "tdt":
        out = out.astype(np.float16)                            # :191-200
    return out, L


def extract_raw(audio):
    """extractRaw (Nemotron :55-129, EOU :193-273): symmetric Hann centred in the 512 frame, zero centre padding, power / 4, no
    normalisation, float32.  -> ([128, nFrames], melLength)."""
    audio = np.ascontiguousarray(audio, dtype=F32)
    if audio.shape[0] == 0:
        return np.zeros((N_MELS, 1), dtype=F32), 0
    logm, L = _log_mel_frames(audio, periodic=False, reflect=False, centred=True, power_scale=1.0)     # (2X)^2 * 0.25 = X^2 exactly
    return np.ascontiguousarray(logm.T), L


class StreamingMel:
    """extractStreaming (ParakeetStreamingASR/StreamingMelPreprocessor.swift:280-393): the DSP of extract("eou") with mean / std from
    sums accumulated over every chunk of the session (`runningSum`, `runningSumSq`, `runningCount`)."""

    def __init__(self, fft_scale=2.0):
        self.fft_scale = fft_scale
        self.reset()

    def reset(self):                                            # resetRunningStats :396-400
        self.sum = np.zeros(N_MELS, dtype=F32)
        self.sumsq = np.zeros(N_MELS, dtype=F32)
        self.count = 0

    def extract(self, audio):
        audio = np.ascontiguousarray(audio, dtype=F32)
        if audio.shape[0] == 0:
            return np.zeros((N_MELS, 1), dtype=F32), 0
        logm, L = _log_mel_frames(audio, periodic=False, reflect=True, centred=False, power_scale=F32(self.fft_scale) * F32(self.fft_scale))
        x = logm.T.copy()
        nf = x.shape[1]
        valid = min(L, nf)
        self.sum = (self.sum + x[:, :valid].sum(axis=1, dtype=F32)).astype(F32)
        self.sumsq = (self.sumsq + (x[:, :valid] * x[:, :valid]).sum(axis=1, dtype=F32)).astype(F32)
        self.count += valid
        n = F32(max(self.count, 1))
        out = np.zeros_like(x)
        for b in range(N_MELS):
            mean = F32(self.sum[b] / n)
            var = max(F32(self.sumsq[b] / n - mean * mean), F32(0))
            std = np.sqrt(F32(var) * n / max(n - F32(1), F32(1)), dtype=F32)
            out[b, :valid] = (x[b, :valid] - mean) * (F32(1.0) / (std + F32(1e-5)))
        return out, L


def fit_frames(mel, target):
    """StreamingSession.truncateMel / padMel (NemotronStreamingASR/StreamingSession.swift:245-274): [128, n] -> [128, target], zeros."""
    out = np.zeros((mel.shape[0], target), dtype=mel.dtype)
    k = min(target, mel.shape[1])
    out[:, :k] = mel[:, :k]
    return out
