"""BASELINE configs[4] (Parakeet-TDT / Nemotron streaming / Parakeet-EOU): the batched device log-mel (csrc/nemo_mel.hip) through the C ABI
against oracle/nemo_mel.py, against the torch.stft goldens directly, at the streaming shape (64 concurrent streams x one 160 ms chunk,
hipGraph replay) and end to end through the restated streaming session with stand-in networks.  The encoder / prediction network /
joint of these models are opaque CoreML bundles in the reference: there is nothing of theirs to compare."""
import json
import os

import numpy as np
import pytest
from conftest import GOLDEN
from oracle import nemo_mel as NM, transducer as OT
from qasr import parakeet as QP, synth
from qasr.model import QasrError
from test_parakeet_cpu import FakeNet

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(GOLDEN, "nemo_mel.npz"))
KAT = json.load(open(os.path.join(GOLDEN, "kat_parakeet.json"), encoding="utf-8"))
# ln(mel + 2^-24): where the energy sits at the guard one f32 ulp of it moves the log by 6e-8 / 6e-8 = O(1) relative ... the values are
# compared where both sides computed the same sum in a different order: 1e-3 absolute on ln values in [-16.7, 5] (measured 2-4e-4)
TOL_RAW = 1.5e-3
TOL_NORM = 4e-3          # (x - mean) / (std + 1e-5) with f32 statistics over <= 3000 frames
TOL_F16 = 8e-3


@pytest.fixture(scope="module")
def dev():
    d = QP.NemoMelDevice(max_streams=64, max_samples=16000 * 31)
    yield d
    d.close()


def _wave(k, seconds):
    return synth.synth_waveform(k, seconds)


@pytest.mark.parametrize("n", [1, 159, 160, 257, 2720, 5280, 16000, 40001])
def test_raw_vs_oracle(dev, n):
    pcm = _wave(n % 7, max(n, 16) / 16000.0)[:n]
    got, lens = dev.extract_batch(QP.RAW, [pcm])
    want, L = NM.extract_raw(pcm)
    assert got.shape == (1,) + want.shape and lens[0] == L == n // 160
    assert np.abs(got[0] - want).max() < TOL_RAW


@pytest.mark.parametrize("variant,name", [(QP.TDT, "tdt"), (QP.EOU, "eou")])
@pytest.mark.parametrize("n", [257, 2720, 16000, 40001, 480000])
def test_normalised_vs_oracle(dev, variant, name, n):
    pcm = _wave(3 + n % 5, n / 16000.0 + 0.01)[:n]
    got, lens = dev.extract_batch(variant, [pcm])
    want, L = NM.extract(pcm, name)
    assert got.shape == (1,) + want.shape and lens[0] == L
    assert np.all(got[0][:, L:] == 0)
    if name == "tdt":
        assert np.array_equal(got[0], got[0].astype(np.float16).astype(np.float32), equal_nan=True)   # float16 values, like the reference's output array
    if name == "tdt" and L == 1:
        # MelPreprocessor.extract divides by melLength - 1 (MelPreprocessor.swift:168): one valid frame is 0 / 0 there, here and in the oracle
        assert np.isnan(got[0][:, 0]).all() and np.isnan(want[:, 0].astype(np.float32)).all()
        return
    d = np.abs(got[0] - want.astype(np.float32))
    assert d.max() < (TOL_F16 if name == "tdt" else TOL_NORM), d.max()


@pytest.mark.parametrize("name", ["chunk160ms", "synth", "speech", "sine1s"])
def test_device_vs_torch_stft_golden(name):
    """The device against the independent goldens directly (fft_scale = 1: a library's FFT scale; RAW has no free constant)."""
    d = QP.NemoMelDevice(max_streams=4, max_samples=16000 * 3, fft_scale=1.0)
    try:
        pcm = G["wave/" + name]
        raw, _ = d.extract_batch(QP.RAW, [pcm])
        assert np.abs(raw[0] - G["raw/" + name]).max() < TOL_RAW
        for variant, key, tol in ((QP.TDT, "tdt", TOL_F16), (QP.EOU, "eou", TOL_NORM)):
            got, _ = d.extract_batch(variant, [pcm])
            assert np.abs(got[0] - G[key + "/" + name]).max() < tol
    finally:
        d.close()
    d2 = QP.NemoMelDevice(max_streams=1, max_samples=16000 * 3, fft_scale=2.0)
    try:
        raw2, _ = d2.extract_batch(QP.RAW, [G["wave/" + name]])                      # extractRaw divides the vDSP factor out
        assert np.abs(raw2[0] - G["raw/" + name]).max() < TOL_RAW
    finally:
        d2.close()


def test_ragged_batch_equals_single_calls(dev):
    clips = [_wave(k, 0.3 + 0.37 * k) for k in range(9)] + [_wave(11, 0.02)[:300]]
    for variant in (QP.RAW, QP.TDT, QP.EOU):
        got, lens = dev.extract_batch(variant, clips)
        for b, c in enumerate(clips):
            one, l1 = dev.extract_batch(variant, [c])
            nf = len(c) // 160 + 1
            assert lens[b] == l1[0]
            assert np.array_equal(got[b][:, :nf], one[0], equal_nan=True), (variant, b)   # rows never interact: bit-identical (TDT at one valid frame: NaN, see above)
            assert np.all(got[b][:, nf:] == 0)


def test_64_streams_one_chunk_graph_replay(dev):
    """configs[4]'s shape: 64 concurrent streams, one 160 ms chunk (2720 samples -> 18 frames, fitted to the encoder's 17) per call.
    Uniform calls replay one captured graph; results equal the ragged (eager) path bit for bit and the oracle within tolerance."""
    rng = np.random.default_rng(0)
    for it in range(3):
        chunks = [(rng.standard_normal(2720) * 0.1).astype(np.float32) for _ in range(64)]
        got, lens = dev.extract_batch(QP.RAW, chunks, fit=17)
        ms, was_graph = dev.timing()
        assert was_graph and got.shape == (64, 128, 17) and (lens == 17).all()
        eager, _ = dev.extract_batch(QP.RAW, chunks[:63] + [chunks[0][:2000]], fit=17)    # ragged -> not the graph
        assert not dev.timing()[1]
        assert np.array_equal(got[:63], eager[:63])
        for b in (0, 31, 63):
            want = NM.fit_frames(NM.extract_raw(chunks[b])[0], 17)
            assert np.abs(got[b] - want).max() < TOL_RAW
    print(f"64 streams x 160 ms chunk: {ms * 1e3:.0f} us per call (H2D + 3 kernels + D2H, graph replay) = {64 * 0.16 / (ms * 1e-3):.0f} audio-s/s of front-end")
    # padding: a short final chunk is zero-filled to 17 frames (padMel)
    short, l2 = dev.extract_batch(QP.RAW, [chunks[0][:800]], fit=17)
    assert l2[0] == 5 and np.all(short[0][:, 6:] == 0) and np.abs(short[0][:, :6] - NM.extract_raw(chunks[0][:800])[0]).max() < TOL_RAW


def test_streaming_statistics_per_stream(dev):
    """extractStreaming: four interleaved streams, three chunks each; every stream's normalisation follows ITS running sums (oracle
    StreamingMel per stream); reset clears one stream only."""
    dev.reset_stats(-1)
    streams = [5, 0, 63, 17]
    oracles = {s: NM.StreamingMel() for s in streams}
    audio = {s: _wave(s, 1.6) for s in streams}
    for step in range(3):
        chunks = [audio[s][step * 8000: step * 8000 + 8000 + 37 * (s % 3)] for s in streams]
        got, lens = dev.extract_batch(QP.EOU_STREAMING, chunks, stream_ids=streams)
        for b, s in enumerate(streams):
            want, L = oracles[s].extract(chunks[b])
            nf = want.shape[1]
            assert lens[b] == L and np.abs(got[b][:, :nf] - want).max() < 2e-2, (step, s)   # f32 E[x^2] - mean^2 of values near -10
    dev.reset_stats(17)
    oracles[17].reset()
    got, _ = dev.extract_batch(QP.EOU_STREAMING, [audio[17][:8000], audio[5][:8000]], stream_ids=[17, 5])
    assert np.abs(got[0][:, :51] - oracles[17].extract(audio[17][:8000])[0]).max() < 2e-2
    assert np.abs(got[1][:, :51] - oracles[5].extract(audio[5][:8000])[0]).max() < 2e-2
    with pytest.raises(QasrError):
        dev.extract_batch(QP.EOU_STREAMING, [audio[5][:8000], audio[5][:8000]], stream_ids=[3, 3])    # one chunk per stream and call


def test_reference_error_and_edge_behaviour(dev):
    z, lens = dev.extract_batch(QP.RAW, [np.zeros(0, np.float32), _wave(0, 0.2)])     # `guard !audio.isEmpty` -> zeros, melLength 0
    assert lens[0] == 0 and not z[0].any() and lens[1] == 20
    z, lens = dev.extract_batch(QP.EOU, [np.zeros(0, np.float32)])
    assert z.shape == (1, 128, 1) and lens[0] == 0 and not z.any()
    with pytest.raises(QasrError):
        dev.extract_batch(QP.TDT, [np.zeros(0, np.float32)])                          # MelPreprocessor.extract reads audio[0]
    with pytest.raises(QasrError, match="256"):
        dev.extract_batch(QP.EOU, [np.zeros(256, np.float32)])                        # reflect padding indexes pre[256 - i]
    with pytest.raises(QasrError, match="qasr error 5"):
        dev.extract_batch(QP.RAW, [np.zeros(16000 * 31 + 1, np.float32)])
    with pytest.raises(QasrError, match="qasr error 5"):
        dev.extract_batch(QP.RAW, [np.zeros(400, np.float32)] * 65)
    sil, lens = dev.extract_batch(QP.EOU, [np.zeros(5120, np.float32)])               # testMelPreprocessorSilence: finite, zero variance
    assert np.isfinite(sil).all() and lens[0] == 32


@pytest.mark.parametrize("case", KAT["mel"], ids=lambda c: c["ref"].split()[-1])
def test_mel_reference_unit_tests_on_the_device(dev, case):
    sig = case["signal"]
    pcm = (np.sin(2.0 * np.pi * sig["hz"] * np.arange(sig["n"], dtype=np.float32) / 16000.0) * sig["amp"]).astype(np.float32) \
        if sig["kind"] == "sine" else np.zeros(sig["n"], np.float32)
    pre = QP.MelPreprocessor(dev) if case["variant"] == "tdt" else QP.StreamingMelPreprocessor(dev)
    mel, L = pre.extract(pcm)
    assert list(mel.shape[:2]) == case["shape01"]
    if case["variant"] == "tdt":
        assert mel.dtype == np.float16
    if "mel_length_gt" in case:
        assert L > case["mel_length_gt"]
    if "mel_length_eq" in case:
        assert L == case["mel_length_eq"]
    for b in range(case.get("zero_mean_bins", 0)):
        assert abs(float(mel[0, b, :L].astype(np.float32).mean())) < case["zero_mean_accuracy"]
    if "finite_first" in case:
        assert np.isfinite(mel.reshape(-1)[:case["finite_first"]]).all()


def test_streaming_session_end_to_end_vs_restated_session(dev):
    """NemotronStreamingASR.StreamingSession over the C ABI (chunk cutting, device mel fitted to 17 frames, RNNT greedy, vocabulary)
    against the restated session on the oracle mel, both driving the same stand-in networks: same chunks, mel within tolerance at the
    encoder's input, identical tokens, texts and partial sequence."""
    cfg = OT.NemotronStreamingConfig()
    table = {i: ("▁w%d" % i if i % 2 == 0 else "x%d" % i) for i in range(cfg.vocab_size)}
    audio = _wave(2, 1.3)
    seen = {"o": [], "q": []}
    na, nb = FakeNet(9, cfg.vocab_size, blank_bias=0.6), FakeNet(9, cfg.vocab_size, blank_bias=0.6)
    so = OT.NemotronSession(NM.extract_raw, lambda m: (seen["o"].append(m.copy()), 2)[1], na.decoder, na.joint, OT.StreamVocabulary(table))
    sq = QP.StreamingSession(QP.StreamingMelPreprocessor(dev), lambda m: (seen["q"].append(m.copy()), 2)[1], nb.decoder, nb.joint,
                             QP.NemotronVocabulary(table))
    po, pq = [], []
    for off in range(0, len(audio), 3000):
        po += so.push_audio(audio[off:off + 3000])
        pq += sq.push_audio(audio[off:off + 3000])
    fo, fq = so.finalize(), sq.finalize()
    assert len(so.chunks) == len(sq.chunks) and all(np.array_equal(a, b) for a, b in zip(so.chunks, sq.chunks))
    assert len(seen["o"]) == len(seen["q"]) > 5
    for a, b in zip(seen["o"], seen["q"]):
        assert a.shape == b.shape == (128, 17) and np.abs(a - b).max() < TOL_RAW
    assert so.tokens == sq.tokens and len(so.tokens) > 3
    assert [(p.text, p.is_final) for p in po] == [(p.text, p.is_final) for p in pq]
    assert fo[0].text == fq[0].text and fq[0].is_final and abs(fo[0].confidence - fq[0].confidence) < 1e-5
