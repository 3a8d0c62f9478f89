#!/usr/bin/env python3
"""Golden vectors for the Parakeet / Nemotron log-mel front-ends (BASELINE configs[4]) from INDEPENDENT implementations:
`torch.stft` -- the function the reference itself says its Swift code was matched against
(`Sources/ParakeetStreamingASR/StreamingMelPreprocessor.swift:34,210,228-229`: "Verified against NeMo: torch.hann_window(400,
periodic=False)", "torch.stft centers the window in the n_fft frame"; `Sources/NemotronStreamingASR/StreamingMelPreprocessor.swift:100`:
"vDSP_fft_zrip scales 2x vs torch.stft -- divide power by 4") -- and `transformers.audio_utils.mel_filter_bank`
(slaney scale + slaney norm on the 257-bin grid), all in float64:

    x -> pre-emphasis 0.97 -> torch.stft(n_fft 512, hop 160, win_length 400, window, center=True, pad_mode) -> |.|^2 -> mel 128
      -> ln(. + 2^-24) [-> per-feature (x - mean) / (std_unbiased + 1e-5) over the first n // 160 frames]

  raw/*   window = hann(400, periodic=False), pad_mode "constant": what extractRaw computes (no free constant, see oracle/nemo_mel.py)
  tdt/*   window = hann(400, periodic=True), pad_mode "reflect", normalised: MelPreprocessor.extract of Parakeet-TDT.
          torch.stft centres a win_length < n_fft window in the frame; the Swift code left-aligns it (`vDSP_vmul(... hannWindow ...
          &paddedFrame ...)` then zero-fills 400...511): a pure time shift of 56 samples, invisible in the power spectrum only up to the
          frame's content -- so the golden emulates the left-aligned placement by handing torch.stft a 512-long window = [hann400, 0 x 112].
  eou/*   the same with the symmetric window (StreamingMelPreprocessor.extract of the EOU model).
The x4 power scaling of vDSP in extract (ln 4, removed by the mean up to the guard) is not part of a library's output: these goldens
pin oracle/nemo_mel.py at fft_scale = 1.0 for tdt / eou and unconditionally for raw.

Run from the repo root in the build container:  python tests/golden/make_nemo_goldens.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "qwen3-asr-swift_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from transformers import audio_utils as au   # noqa: E402
from qasr import synth                       # noqa: E402
from make_hf_goldens_mel import fixture_slice   # noqa: E402

GUARD = 2.0 ** -24


def log_mel(pcm, window512, pad_mode, fb):
    x = torch.from_numpy(pcm.astype(np.float64))
    pre = torch.cat([x[:1], x[1:] - 0.97 * x[:-1]])
    spec = torch.stft(pre, n_fft=512, hop_length=160, win_length=512, window=window512, center=True, pad_mode=pad_mode,
                      return_complex=True)                                   # [257, frames]
    power = spec.real ** 2 + spec.imag ** 2
    mel = torch.from_numpy(fb.T.astype(np.float64)) @ power                  # [128, frames]
    return torch.log(mel + GUARD)


def normalise(logm, n_samples):
    L = n_samples // 160
    v = logm[:, :L]
    out = torch.zeros_like(logm)
    out[:, :L] = (v - v.mean(dim=1, keepdim=True)) / (v.std(dim=1, keepdim=True) + 1e-5)     # unbiased std, like NeMo
    return out


def main():
    fb = au.mel_filter_bank(257, 128, 0.0, 8000.0, 16000, norm="slaney", mel_scale="slaney")      # [257, 128]
    sym = torch.hann_window(400, periodic=False, dtype=torch.float64)
    per = torch.hann_window(400, periodic=True, dtype=torch.float64)
    z = torch.zeros(56, dtype=torch.float64)
    centred_sym = torch.cat([z, sym, z])                       # what torch.stft does itself with win_length = 400
    left_per = torch.cat([per, z, z])
    left_sym = torch.cat([sym, z, z])
    waves = {"chunk160ms": synth.synth_waveform(7, 2720 / 16000.0), "synth": synth.synth_waveform(3, 1.0), "speech": fixture_slice(1.5, 5.1),
             "sine1s": (0.5 * np.sin(2.0 * np.pi * 440.0 * np.arange(16000) / 16000.0)).astype(np.float32)}
    out = {"filterbank": fb.astype(np.float64)}
    for name, pcm in waves.items():
        pcm = pcm.astype(np.float32)
        out["wave/" + name] = pcm
        out["raw/" + name] = log_mel(pcm, centred_sym, "constant", fb).numpy().astype(np.float32)
        # cross-check the emulation of the centred placement against torch.stft's own handling of win_length < n_fft
        x = torch.from_numpy(pcm.astype(np.float64))
        pre = torch.cat([x[:1], x[1:] - 0.97 * x[:-1]])
        own = torch.stft(pre, n_fft=512, hop_length=160, win_length=400, window=sym, center=True, pad_mode="constant", return_complex=True)
        ref = torch.log(torch.from_numpy(fb.T.astype(np.float64)) @ (own.real ** 2 + own.imag ** 2) + GUARD)
        assert float((ref - log_mel(pcm, centred_sym, "constant", fb)).abs().max()) < 1e-9
        out["tdt/" + name] = normalise(log_mel(pcm, left_per, "reflect", fb), len(pcm)).numpy().astype(np.float32)
        out["eou/" + name] = normalise(log_mel(pcm, left_sym, "reflect", fb), len(pcm)).numpy().astype(np.float32)
        if name in ("chunk160ms", "speech"):        # the streaming variant's un-normalised frames (running statistics are the test's own arithmetic)
            out["eou_unnormalised/" + name] = log_mel(pcm, left_sym, "reflect", fb).numpy().astype(np.float32)
        print(name, pcm.shape, out["raw/" + name].shape)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "nemo_mel.npz"), **out)


if __name__ == "__main__":
    main()
