"""BASELINE configs[4] (Parakeet-TDT / Nemotron streaming / Parakeet-EOU), restatable slice, CPU side:
  * oracle/nemo_mel.py against goldens from torch.stft + transformers' slaney filterbank (tests/golden/make_nemo_goldens.py);
  * oracle/transducer.py AND the C ABI's host entry points against the reference's own unit-test cases (tests/golden/kat_parakeet.json:
    config constants, vocabulary decode / decodeWords, log-softmax, confidence) and against each other on scripted and random networks;
  * the streaming session's chunk cutting (qasr_stream_chunker_*) against the restated StreamingSession bookkeeping.
No GPU call here; the device mel is tests/test_gpu_nemo_mel.py."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest
from conftest import GOLDEN
from oracle import mel as OM, nemo_mel as NM, transducer as OT
from qasr import _lib, parakeet as QP

KAT = json.load(open(os.path.join(GOLDEN, "kat_parakeet.json"), encoding="utf-8"))
G = np.load(os.path.join(GOLDEN, "nemo_mel.npz"))
WAVES = ("chunk160ms", "synth", "speech", "sine1s")


# ---- mel oracle vs independent goldens ---------------------------------------------------------------------------------------
def test_filterbank_is_the_qwen_filterbank_and_matches_the_library():
    fb = NM.mel_filterbank()
    assert np.array_equal(fb, OM.mel_filterbank())                 # the reference's copies are the same arithmetic
    assert np.abs(fb - G["filterbank"].T).max() < 1e-6


@pytest.mark.parametrize("name", WAVES)
def test_extract_raw_matches_torch_stft(name):
    """extractRaw has no free constant: (2 X)^2 / 4 is the library's |X|^2.  ln values reach -16.6 at the 2^-24 guard, where one f32
    ulp of the mel energy is 1e-7 / 6e-8: tolerance 5e-4 absolute (measured 1.1-1.6e-4)."""
    pcm = G["wave/" + name]
    mel, L = NM.extract_raw(pcm)
    assert mel.shape == G["raw/" + name].shape == (128, len(pcm) // 160 + 1) and L == len(pcm) // 160
    assert np.abs(mel - G["raw/" + name]).max() < 5e-4


@pytest.mark.parametrize("name", WAVES)
@pytest.mark.parametrize("variant", ["tdt", "eou"])
def test_extract_normalised_matches_torch_stft(name, variant):
    """Per-feature normalisation over the first n / 160 frames with the unbiased std, frames past melLength zero.  At the library's FFT
    scale (fft_scale = 1); the vDSP x4 only enters through the 2^-24 guard.  TDT output is float16 (tolerance = its rounding)."""
    pcm = G["wave/" + name]
    mel, L = NM.extract(pcm, variant, fft_scale=1.0)
    want = G[variant + "/" + name]
    assert mel.shape == want.shape and mel.dtype == (np.float16 if variant == "tdt" else np.float32)
    assert np.all(mel[:, L:] == 0)
    assert np.abs(mel.astype(np.float32) - want).max() < (6e-3 if variant == "tdt" else 4e-4)


def test_vdsp_scale_only_moves_bins_near_the_guard():
    """fft_scale 2 vs 1 is ln 4 on every un-normalised value except where the mel energy is comparable to 2^-24; after normalisation
    the difference vanishes on loud bins."""
    pcm = G["wave/synth"]                                               # tones + a noise floor: most bins carry energy in every frame
    a, L = NM.extract(pcm, "eou", fft_scale=1.0)
    b, _ = NM.extract(pcm, "eou", fft_scale=2.0)
    raw, _ = NM.extract_raw(pcm)
    loud = raw[:, :L].min(axis=1) > math.log(2.0 ** -24) + 9.0           # bins whose energy stays >> the guard in every frame
    assert loud.sum() >= 8
    assert np.abs(a[loud] - b[loud]).max() < 2e-3


def test_streaming_running_statistics():
    """extractStreaming: sums accumulate over chunks; after k chunks the normalisation uses the statistics of all k (against the
    golden's un-normalised frames and float64 arithmetic)."""
    pcm = G["wave/speech"]
    sm = NM.StreamingMel(fft_scale=1.0)
    chunks = [pcm[:8000], pcm[8000:16000], pcm[16000:24000]]
    s = np.zeros(128)
    s2 = np.zeros(128)
    cnt = 0
    for ch in chunks:
        got, L = sm.extract(ch)
        e, _ = NM.extract(ch, "eou", fft_scale=1.0)                       # only used for the frame count
        un = NM._log_mel_frames(ch, False, True, False, 1.0)[0].T.astype(np.float64)
        s += un[:, :L].sum(1)
        s2 += (un[:, :L] ** 2).sum(1)
        cnt += L
        mean = s / cnt
        std = np.sqrt(np.maximum(s2 / cnt - mean ** 2, 0) * cnt / max(cnt - 1, 1))
        want = (un[:, :L] - mean[:, None]) / (std[:, None] + 1e-5)
        assert got.shape == e.shape and np.all(got[:, L:] == 0)
        assert np.abs(got[:, :L] - want).max() < 5e-3                     # f32 sums of squares of values ~ -10: cancellation
    assert sm.count == cnt
    sm.reset()
    assert sm.count == 0 and not sm.sum.any()


@pytest.mark.parametrize("case", KAT["mel"], ids=lambda c: c["ref"].split()[-1])
def test_mel_reference_unit_tests_on_the_oracle(case):
    sig = case["signal"]
    if sig["kind"] == "sine":
        pcm = (np.sin(2.0 * np.pi * sig["hz"] * np.arange(sig["n"], dtype=np.float32) / 16000.0) * sig["amp"]).astype(np.float32)
    else:
        pcm = np.zeros(sig["n"], np.float32)
    mel, L = NM.extract(pcm, case["variant"])
    assert mel.shape[0] == 128
    if "mel_length_gt" in case:
        assert L > case["mel_length_gt"]
    if "mel_length_eq" in case:
        assert L == case["mel_length_eq"]
    for b in range(case.get("zero_mean_bins", 0)):
        assert abs(float(mel[b, :L].astype(np.float32).mean())) < case["zero_mean_accuracy"]
    if "finite_first" in case:
        assert np.isfinite(mel.reshape(-1)[:case["finite_first"]]).all()


# ---- config constants --------------------------------------------------------------------------------------------------------
def test_config_constants():
    for name, model, ocfg in (("parakeet_tdt", "parakeet-tdt", OT.ParakeetConfig()), ("parakeet_eou", "parakeet-eou", OT.ParakeetEOUConfig()),
                              ("nemotron_streaming", "nemotron-streaming", OT.NemotronStreamingConfig())):
        k = KAT["configs"][name]
        c = QP.transducer_config(model)
        assert (c.vocab_size, c.blank_id) == (k["vocab_size"], k["blank_id"]) == (ocfg.vocab_size, ocfg.blank_token_id)
        assert c.eou_id == k.get("eou_id", -1)
        assert c.n_durations == k.get("n_durations", 0) and list(c.durations)[:c.n_durations] == k.get("durations", [])
        assert c.max_symbols == 10
        for key, attr in (("encoder_hidden", "encoder_hidden"), ("encoder_layers", "encoder_layers"), ("decoder_hidden", "decoder_hidden"),
                          ("decoder_layers", "decoder_layers"), ("mel_frames", "mel_frames"), ("output_frames", "output_frames"),
                          ("pre_cache_size", "pre_cache_size"), ("chunk_ms", "chunk_ms"), ("hop_length", "hop_length"), ("win_length", "win_length"),
                          ("n_fft", "n_fft"), ("num_mel_bins", "num_mel_bins"), ("subsampling_factor", "subsampling_factor")):
            if key in k:
                assert getattr(ocfg, attr) == k[key], (name, key)
    assert QP.transducer_config("aufklarer/Parakeet-TDT-v3-CoreML-INT8").vocab_size == 8192
    assert QP.transducer_config("aufklarer/Parakeet-EOU-120M-CoreML-INT8").eou_id == 1024
    assert QP.transducer_config("aufklarer/Nemotron-Speech-Streaming-0.6B-CoreML-INT8").vocab_size == 1024
    lib = _lib.load(strict=True)
    assert lib.qasr_transducer_default_config(b"whisper", C.byref(_lib.QasrTransducerConfig())) != 0
    # the session's chunk geometry (StreamingSession.swift:113-115): 17 x 160 and 2 x 8 x 160
    n = OT.NemotronStreamingConfig()
    assert n.mel_frames * n.hop_length == 2720 and n.output_frames * n.subsampling_factor * n.hop_length == 2560


# ---- vocabulary ----------------------------------------------------------------------------------------------------------------
def _vocabs(case):
    table = {int(k): v for k, v in case["vocab"].items()}
    o = (OT.ParakeetVocabulary if case["style"] == 0 else OT.StreamVocabulary)(table)
    q = (QP.ParakeetVocabulary if case["style"] == 0 else QP.NemotronVocabulary)(table)
    return o, q


@pytest.mark.parametrize("case", KAT["vocab_decode"], ids=lambda c: c["ref"].split()[-1] + str(c["style"]))
def test_vocabulary_decode_kats(case):
    o, q = _vocabs(case)
    assert o.decode(case["ids"]) == case["text"]
    assert q.decode(case["ids"]) == case["text"]
    assert q.count == len(case["vocab"])


@pytest.mark.parametrize("case", KAT["decode_words"], ids=lambda c: c["ref"].split()[-1] + str(c["style"]))
def test_decode_words_kats(case):
    o, q = _vocabs(case)
    lps = case["log_probs"] if "log_probs" in case else [math.log(p) for p in case["log_probs_ln_of"]]
    want_conf = case["confidences"] if "confidences" in case else [math.exp(v) for v in case["confidences_exp_of"]]
    for impl in (o, q):
        words = impl.decode_words(case["ids"], lps)
        assert [w.word for w in words] == case["words"]
        for w, c in zip(words, want_conf):
            assert abs(w.confidence - c) <= max(case["accuracy"], 1e-6)
            assert 0.0 <= w.confidence <= 1.0


def test_vocabulary_fuzz_against_oracle(tmp_path):
    """random piece tables (with bare marks, interior marks, empty pieces, unknown ids) through both styles; vocab.json loader."""
    rng = np.random.default_rng(5)
    pieces = ["▁", "▁a", "b", "▁▁c", "d▁e", "", ",", "▁Hello", ".", "▁x▁", " y", "ß", "▁日本"]
    for trial in range(60):
        table = {int(i): pieces[int(rng.integers(len(pieces)))] for i in rng.choice(40, size=12, replace=False)}
        ids = rng.integers(0, 40, size=int(rng.integers(0, 14))).tolist()
        lps = (-rng.random(len(ids)) * 3).astype(np.float32).tolist()
        path = tmp_path / f"vocab{trial}.json"
        path.write_text(json.dumps({str(k): v for k, v in table.items()} | {"not-a-number": "zzz"}, ensure_ascii=(trial % 2 == 0)), encoding="utf-8")
        for O, Q in ((OT.ParakeetVocabulary, QP.ParakeetVocabulary), (OT.StreamVocabulary, QP.NemotronVocabulary)):
            o = O(table)
            for q in (Q(table), Q(path=path)):
                assert q.count == len(table)
                assert q.decode(ids) == o.decode(ids), (table, ids)
                a, b = o.decode_words(ids, lps), q.decode_words(ids, lps)
                assert [w.word for w in a] == [w.word for w in b], (table, ids)
                assert np.allclose([w.confidence for w in a], [w.confidence for w in b], atol=1e-6)
                if ids:
                    a, b = o.decode_words(ids, lps[:-1]), q.decode_words(ids, lps[:-1])
                    assert [(w.word, w.confidence) for w in a] == [(w.word, w.confidence) for w in b]
    lib = _lib.load(strict=True)
    h = C.c_void_p()
    assert lib.qasr_sp_vocab_load(str(tmp_path / "missing.json").encode(), 0, C.byref(h)) != 0
    (tmp_path / "bad.json").write_text("[1, 2]")
    assert lib.qasr_sp_vocab_load(str(tmp_path / "bad.json").encode(), 0, C.byref(h)) != 0


# ---- log-softmax / confidence ------------------------------------------------------------------------------------------------
def test_log_softmax_and_confidence_kats():
    lib = _lib.load(strict=True)
    k = KAT["log_softmax"]
    x = np.asarray(k["logits"], np.float32)
    for i, p in enumerate(k["probs"]):
        a = float(OT.log_softmax_at(x, i))
        b = float(lib.qasr_log_softmax_at(x.ctypes.data_as(C.POINTER(C.c_float)), len(x), i))
        assert abs(math.exp(a) - p) < k["accuracy"] and abs(a - b) < 1e-6 and a < 0
    rng = np.random.default_rng(0)
    for _ in range(50):
        x = (rng.standard_normal(1025) * 8).astype(np.float32)
        i = int(rng.integers(1025))
        assert abs(float(OT.log_softmax_at(x, i)) - float(lib.qasr_log_softmax_at(x.ctypes.data_as(C.POINTER(C.c_float)), 1025, i))) < 2e-5
    for lp in KAT["confidence_range"]["log_probs"]:
        v = np.asarray([lp], np.float32)
        c = float(lib.qasr_transducer_confidence(v.ctypes.data_as(C.POINTER(C.c_float)), 1))
        assert 0.0 <= c <= 1.0 and abs(c - float(OT.confidence([lp]))) < 1e-6
    assert lib.qasr_transducer_confidence(None, 0) == 0.0 and float(OT.confidence([])) == 0.0
    v = np.asarray([0.5, 0.7], np.float32)                               # positive "log-probs" clamp at 1 (min(1.0, exp))
    assert lib.qasr_transducer_confidence(v.ctypes.data_as(C.POINTER(C.c_float)), 2) == 1.0


# ---- greedy loops: scripted + random networks ----------------------------------------------------------------------------------
class FakeNet:
    """A deterministic stand-in for the CoreML prediction network + joint: the logits of (frame, fed-token history) come from a hash
    seeded generator, sharpened so that argmax is unambiguous; records what the loop fed."""

    def __init__(self, seed, vocab, n_dur=0, blank_bias=2.0, script=None):
        self.seed, self.vocab, self.n_dur, self.blank_bias, self.script = seed, vocab, n_dur, blank_bias, script
        self.fed = []
        self.joint_calls = []

    def decoder(self, token):
        self.fed.append(int(token))

    def joint(self, t):
        self.joint_calls.append((t, len(self.fed)))
        if self.script is not None:
            tok, dur = self.script(t, self.fed)
            tl = np.full(self.vocab + 1, -5.0, np.float32)
            tl[tok] = 4.0
            dl = np.full(max(self.n_dur, 1), -3.0, np.float32)
            dl[dur] = 2.0
        else:
            rng = np.random.default_rng([self.seed, t, len(self.fed), self.fed[-1] if self.fed else 0])
            tl = rng.standard_normal(self.vocab + 1).astype(np.float32)
            tl[self.vocab] += self.blank_bias * (1.0 + (len(self.fed) % 3))
            tl = tl.astype(np.float16).astype(np.float32)                 # the CoreML outputs are float16
            dl = rng.standard_normal(max(self.n_dur, 1)).astype(np.float32)
        return (tl, dl) if self.n_dur else tl


def test_tdt_greedy_scripted():
    """The loop's rules one by one (TDTGreedyDecoder.swift:96-132): priming with blank; blank -> +1 frame; token -> +max(duration, 1)
    and feed; duration bin 0 still advances one frame; ids below 274 are fed but not reported."""
    cfg = QP.transducer_config("parakeet-tdt")
    B = cfg.blank_id
    plan = {0: (300, 2), 2: (B, 0), 3: (100, 0), 4: (500, 4), 8: (B, 0), 9: (400, 1)}
    def script(t, fed):
        return plan[t]
    for run in ("oracle", "cabi"):
        net = FakeNet(0, cfg.vocab_size, 5, script=script)
        if run == "oracle":
            toks, lps, conf = OT.tdt_greedy(10, net.decoder, net.joint)
        else:
            toks, lps, conf = QP.TDTGreedyDecoder(cfg).decode(10, net.decoder, net.joint)
        assert toks == [300, 500, 400]
        assert net.fed == [B, 300, 100, 500, 400]
        assert [t for t, _ in net.joint_calls] == [0, 2, 3, 4, 8, 9]
        assert len(lps) == 3 and all(lp < 0 for lp in lps) and 0 < conf <= 1
    with pytest.raises(KeyError):
        OT.tdt_greedy(11, FakeNet(0, cfg.vocab_size, 5, script=script).decoder, FakeNet(0, cfg.vocab_size, 5, script=script).joint)
    # a failing callback aborts the C loop with a status, not a crash
    net = FakeNet(0, cfg.vocab_size, 5, script=script)
    with pytest.raises(QP.QasrError):
        QP.TDTGreedyDecoder(cfg).decode(11, net.decoder, net.joint)


@pytest.mark.parametrize("seed", range(6))
def test_tdt_greedy_random_network_equals_oracle(seed):
    cfg = QP.transducer_config("parakeet-tdt")
    T = 40 + 7 * seed
    a, b = FakeNet(seed, cfg.vocab_size, 5, blank_bias=1.5), FakeNet(seed, cfg.vocab_size, 5, blank_bias=1.5)
    to, lo, co = OT.tdt_greedy(T, a.decoder, a.joint)
    tq, lq, cq = QP.TDTGreedyDecoder(cfg).decode(T, b.decoder, b.joint)
    assert to == tq and a.fed == b.fed and a.joint_calls == b.joint_calls
    assert np.allclose(lo, lq, atol=2e-5) and abs(float(co) - cq) < 1e-5
    assert len(a.fed) > 3


def test_rnnt_greedy_scripted_rules():
    """RNNTGreedyDecoder: <= 10 symbols per frame, blank moves on, frameOffset shifts the frame index, the prediction network is NOT primed
    by decode(); with an EOU id the decode stops at it without reporting or feeding it."""
    cfg = QP.transducer_config("nemotron-streaming")
    B = cfg.blank_id
    def never_blank(t, fed):
        return (7 + len(fed) % 5, 0)
    for run in ("oracle", "cabi"):
        net = FakeNet(0, cfg.vocab_size, script=never_blank)
        if run == "oracle":
            toks, lps, eou = OT.rnnt_greedy(2, net.decoder, net.joint, cfg.vocab_size, B, frame_offset=3)
        else:
            toks, lps, eou = QP.RNNTGreedyDecoder(cfg).decode(2, net.decoder, net.joint, frame_offset=3)
        assert len(toks) == 20 and not eou and net.fed == toks                     # 10 per frame, no priming here
        assert [t for t, _ in net.joint_calls] == [3] * 10 + [4] * 10
    ecfg = QP.transducer_config("parakeet-eou")
    plan = {0: [5, 6, ecfg.blank_id], 1: [ecfg.eou_id], 2: [9]}
    def eou_script(t, fed):
        k = sum(1 for c in calls if c == t) - 1
        return (plan[t][k], 0)
    for run in ("oracle", "cabi"):
        calls = []
        net = FakeNet(0, ecfg.vocab_size, script=eou_script)
        j = net.joint
        def joint(t, j=j):
            calls.append(t)
            return j(t)
        if run == "oracle":
            toks, lps, eou = OT.rnnt_greedy(3, net.decoder, joint, ecfg.vocab_size, ecfg.blank_id, eou_id=ecfg.eou_id)
        else:
            toks, lps, eou = QP.RNNTGreedyDecoder(ecfg).decode(3, net.decoder, joint)
        assert toks == [5, 6] and eou and net.fed == [5, 6] and calls == [0, 0, 0, 1]   # frame 2 never reached


@pytest.mark.parametrize("model", ["nemotron-streaming", "parakeet-eou"])
@pytest.mark.parametrize("seed", range(4))
def test_rnnt_greedy_random_network_equals_oracle(model, seed):
    cfg = QP.transducer_config(model)
    a, b = FakeNet(seed, cfg.vocab_size, blank_bias=1.0), FakeNet(seed, cfg.vocab_size, blank_bias=1.0)
    eou = cfg.eou_id if cfg.eou_id >= 0 else None
    to, lo, eo = OT.rnnt_greedy(30, a.decoder, a.joint, cfg.vocab_size, cfg.blank_id, eou_id=eou, frame_offset=seed)
    tq, lq, eq = QP.RNNTGreedyDecoder(cfg).decode(30, b.decoder, b.joint, frame_offset=seed)
    assert to == tq and eo == eq and a.fed == b.fed and a.joint_calls == b.joint_calls
    assert np.allclose(lo, lq, atol=2e-5)


def test_greedy_argument_checks():
    lib = _lib.load(strict=True)
    cfg = QP.transducer_config("parakeet-tdt")
    net = FakeNet(0, cfg.vocab_size, 5)
    cb = QP._callbacks(net.decoder, net.joint, cfg.vocab_size + 1, 5)
    toks = np.zeros(2, np.int32)
    I = C.POINTER(C.c_int32)
    assert lib.qasr_tdt_greedy_decode(None, C.byref(cb), 4, toks.ctypes.data_as(I), None, 2, None) < 0
    assert lib.qasr_tdt_greedy_decode(C.byref(cfg), None, 4, toks.ctypes.data_as(I), None, 2, None) < 0
    assert lib.qasr_tdt_greedy_decode(C.byref(cfg), C.byref(cb), -1, toks.ctypes.data_as(I), None, 2, None) < 0
    assert lib.qasr_tdt_greedy_decode(C.byref(cfg), C.byref(cb), 0, toks.ctypes.data_as(I), None, 2, None) == 0       # empty encoder output
    bad = QP.transducer_config("parakeet-tdt")
    bad.n_durations = 0
    assert lib.qasr_tdt_greedy_decode(C.byref(bad), C.byref(cb), 4, toks.ctypes.data_as(I), None, 2, None) < 0
    # more tokens than the caller's buffer: a status, nothing written past the end
    def never_blank(t, fed):
        return (300, 1)
    net = FakeNet(0, cfg.vocab_size, 5, script=never_blank)
    cb = QP._callbacks(net.decoder, net.joint, cfg.vocab_size + 1, 5)
    assert lib.qasr_tdt_greedy_decode(C.byref(cfg), C.byref(cb), 9, toks.ctypes.data_as(I), None, 2, None) == -5


# ---- streaming session bookkeeping -------------------------------------------------------------------------------------------
def test_chunker_matches_the_restated_session():
    """qasr_stream_chunker_* cuts exactly the chunks NemotronSession (StreamingSession.pushAudio / finalize restated) hands to its mel:
    random push sizes, the 160-sample overlap, the zero-padded tail, nothing at all for an empty stream."""
    lib = _lib.load(strict=True)
    F = C.POINTER(C.c_float)
    rng = np.random.default_rng(3)
    for trial in range(12):
        total = int(rng.integers(0, 30000))
        audio = rng.standard_normal(total).astype(np.float32)
        sess = OT.NemotronSession(lambda ch: (np.zeros((128, 18), np.float32), 17), lambda m: 0, lambda t: None, lambda t: None, OT.StreamVocabulary({}))
        ck = C.c_void_p()
        assert lib.qasr_stream_chunker_create(sess.samples_per_chunk, sess.shift_samples, C.byref(ck)) == 0
        got = []
        chunk = np.empty(sess.samples_per_chunk, np.float32)
        off = 0
        while off < total:
            k = int(rng.integers(1, 6000))
            piece = audio[off:off + k]
            sess.push_audio(piece)
            assert lib.qasr_stream_chunker_push(ck, piece.ctypes.data_as(F), len(piece)) == 0
            while lib.qasr_stream_chunker_pop(ck, chunk.ctypes.data_as(F)):
                got.append(chunk.copy())
            assert lib.qasr_stream_chunker_buffered(ck) == sess.buf.shape[0]
            off += k
        sess.finalize()
        if lib.qasr_stream_chunker_flush(ck, chunk.ctypes.data_as(F)):
            got.append(chunk.copy())
        assert lib.qasr_stream_chunker_flush(ck, chunk.ctypes.data_as(F)) == 0
        lib.qasr_stream_chunker_destroy(ck)
        assert len(got) == len(sess.chunks)
        for a, b in zip(got, sess.chunks):
            assert np.array_equal(a, b)
        if total >= 2720:
            assert np.array_equal(got[1][:160], got[0][2560:]) if len(got) > 1 else True
    h = C.c_void_p()
    assert lib.qasr_stream_chunker_create(0, 1, C.byref(h)) != 0 and lib.qasr_stream_chunker_create(100, 101, C.byref(h)) != 0


def test_session_on_the_oracle_end_to_end():
    """The restated session with fake networks: partial transcripts grow, finalize reports the whole text once, an all-blank stream
    reports nothing (StreamingSession.swift:141,222-223)."""
    cfg = OT.NemotronStreamingConfig()
    vocab = OT.StreamVocabulary({i: ("▁w%d" % i if i % 2 == 0 else "x%d" % i) for i in range(cfg.vocab_size)})
    net = FakeNet(4, cfg.vocab_size, blank_bias=0.5)
    sess = OT.NemotronSession(NM.extract_raw, lambda mel: 2, net.decoder, net.joint, vocab)
    audio = (np.random.default_rng(1).standard_normal(16000) * 0.1).astype(np.float32)
    parts = sess.push_audio(audio[:9000]) + sess.push_audio(audio[9000:])
    fin = sess.finalize()
    assert net.fed[0] == cfg.blank_token_id and len(sess.chunks) == 7          # 16000 samples: 6 full chunks + the padded tail
    assert len(fin) == 1 and fin[0].is_final and fin[0].text == vocab.decode(sess.tokens)
    texts = [p.text for p in parts]
    assert all(b.startswith(a) or len(b) >= len(a) for a, b in zip(texts, texts[1:]))
    quiet = FakeNet(4, cfg.vocab_size, blank_bias=50.0)
    s2 = OT.NemotronSession(NM.extract_raw, lambda mel: 2, quiet.decoder, quiet.joint, vocab)
    assert s2.push_audio(audio) == [] and s2.finalize() == []
