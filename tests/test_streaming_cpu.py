"""qasr/streaming.py (mirror of StreamingASR.swift + StreamingVADProcessor.swift) with a scripted VAD and a fake ASR model:
the reference's own unit-test cases (Tests/Qwen3ASRTests/StreamingASRTests.swift:10-145) plus hand-derived event /
segment sequences for the control flow (hysteresis, duration filters, force-split, partial results, flush)."""
import numpy as np
import pytest
from qasr import streaming as S

CD = np.float32(512) / np.float32(16000)          # 0.032 s per VAD chunk


def test_lcp_kats():
    """StreamingASRTests.swift:10-48."""
    L = S.longest_common_prefix
    assert L([], []) == [] and L(["hello"], []) == [] and L([], ["hello"]) == []
    assert L(["can", "you", "guarantee"], ["can", "you", "guarantee"]) == ["can", "you", "guarantee"]
    assert L(["can", "you", "guarantee"], ["can", "you", "help"]) == ["can", "you"]
    assert L(["hello"], ["goodbye"]) == []
    assert L(["Can", "You"], ["can", "you", "help"]) == ["can", "you"]          # elements of b
    assert L(["the", "quick", "brown", "fox"], ["the", "quick"]) == ["the", "quick"]


def test_config_defaults():
    """StreamingASRTests.swift:51-75, Configuration.swift:84-91."""
    c = S.StreamingASRConfig()
    assert (c.max_segment_duration, c.max_tokens, c.emit_partial_results, c.partial_result_interval, c.language) == (10.0, 448, False, 1.0, None)
    v = c.vad_config
    assert (v.onset, v.offset, v.min_speech_duration, v.min_silence_duration) == (0.5, 0.35, 0.25, 0.1)
    c2 = S.StreamingASRConfig(max_segment_duration=15.0, language="en", max_tokens=256, emit_partial_results=True, partial_result_interval=0.5)
    assert (c2.max_segment_duration, c2.language, c2.max_tokens, c2.emit_partial_results, c2.partial_result_interval) == (15.0, "en", 256, True, 0.5)
    seg = S.TranscriptionSegment("hello world", 1.0, 2.5, True, 0)
    assert seg.text == "hello world" and seg.is_final and seg.segment_index == 0


class ScriptedVAD:
    """probability of chunk k = probs[k] (0 beyond the script)."""
    def __init__(self, probs):
        self.probs, self.k = list(probs), 0

    def __call__(self, chunk):
        assert chunk.shape == (512,)
        p = self.probs[self.k] if self.k < len(self.probs) else 0.0
        self.k += 1
        return p

    def reset(self):
        self.k = 0


def test_vad_state_machine_hand_derived():
    """5 silent chunks, 20 speech chunks, 10 silent chunks (StreamingVADProcessor.swift:163-225):
    onset at chunk 5 (t = 0.16); confirmed when next - 0.16 >= 0.25 -> chunk 12; offset at chunk 25 (t = 0.8);
    silence confirmed when next - 0.8 >= 0.1 -> chunk 28."""
    vad = ScriptedVAD([0.1] * 5 + [0.9] * 20 + [0.1] * 10)
    p = S.StreamingVADProcessor(vad)
    log = []
    for k in range(35):
        for ev in p.process(np.zeros(512, np.float32)):
            log.append((k, ev))
    assert [k for k, _ in log] == [12, 28]
    assert log[0][1] == ("speechStarted", np.float32(5) * CD)
    assert log[1][1] == ("speechEnded", S.SpeechSegment(np.float32(5) * CD, np.float32(25) * CD))
    assert p.flush() == [] and float(p.current_time) == pytest.approx(35 * 0.032)
    # a blip shorter than minSpeechDuration is dropped; speech resumed inside minSilenceDuration is one segment
    p = S.StreamingVADProcessor(ScriptedVAD([0.9, 0.9, 0.1] + [0.0] * 5 + [0.9] * 12 + [0.2, 0.2] + [0.9] * 5 + [0.0] * 6))
    ev = p.process(np.zeros(512 * 34, np.float32))
    assert [e[0] for e in ev] == ["speechStarted", "speechEnded"]
    assert ev[1][1] == S.SpeechSegment(np.float32(8) * CD, np.float32(27) * CD)
    # open speech at end of stream is closed by flush (partial last chunk zero-padded)
    p = S.StreamingVADProcessor(ScriptedVAD([0.9] * 40))
    ev = p.process(np.zeros(512 * 20 + 100, np.float32))
    assert [e[0] for e in ev] == ["speechStarted"]
    assert p.flush() == [("speechEnded", S.SpeechSegment(np.float32(0), np.float32(21) * CD))]
    # pending speech shorter than the minimum at flush: nothing
    p = S.StreamingVADProcessor(ScriptedVAD([0.0] * 3 + [0.9] * 3))
    assert p.process(np.zeros(512 * 6, np.float32)) == [] and p.flush() == []


class FakeASR:
    """Audio = sample-index ramp, so a span identifies itself: 'a-b' (or '' for spans starting at a multiple of 7777)."""
    class _Cfg:
        max_batch = 3
    cfg = _Cfg()

    def __init__(self):
        self.calls = []

    def _text(self, span):
        a = int(span[0])
        self.calls.append((a, a + len(span)))
        return f" s{a} e{a + len(span)} "

    def transcribe(self, audio, sample_rate=16000, language=None, max_tokens=448, context=None):
        self.calls_opts = (language, max_tokens, context)
        return self._text(audio).strip()

    def encode_text(self, text):
        return [len(text)]

    def transcribe_batch(self, clips, max_tokens=448, **opts):
        assert len(clips) <= self.cfg.max_batch
        self.batch_opts = opts
        return [[int(c[0]), int(c[0]) + len(c)] for c in clips]

    def detokenize(self, toks):
        return f" s{toks[0]} e{toks[1]} "


def _ramp(seconds):
    return np.arange(int(seconds * 16000), dtype=np.float32)


def test_stream_segments_hand_derived():
    asr = FakeASR()
    st = S.StreamingASR(asr, ScriptedVAD([0.1] * 5 + [0.9] * 20 + [0.1] * 10 + [0.9] * 30))
    segs = list(st.transcribe_stream(_ramp(2.08), config=S.StreamingASRConfig(language="en", max_tokens=99, context="ctx")))
    # segment 1: 0.16 .. 0.8 s -> samples 2560 .. 12800; segment 2 opens at chunk 35 (1.12 s) and is closed by the flush at
    # the end of the 65 chunks (2.08 s); the flush path does not advance the index (StreamingASR.swift:252-257)
    assert segs == [S.TranscriptionSegment("s2560 e12800", pytest.approx(0.16), pytest.approx(0.8), True, 0),
                    S.TranscriptionSegment("s17920 e33280", pytest.approx(1.12), pytest.approx(2.08), True, 1)]
    assert asr.calls_opts == ("en", 99, "ctx")


def test_force_split_and_partials():
    probs = [0.9] * 1000
    cfg = S.StreamingASRConfig(max_segment_duration=2.0)
    asr = FakeASR()
    segs = list(S.StreamingASR(asr, ScriptedVAD(probs)).transcribe_stream(_ramp(5.0), config=cfg))
    # speech from t = 0; force-split when now - start >= 2.0: after chunk 62 (now = 2.016), then from 2.016: 4.032; flush closes 4.032 .. 5.024
    assert [(s.text, s.segment_index, s.is_final) for s in segs] == [("s0 e32256", 0, True), ("s32256 e64512", 1, True), ("s64512 e80000", 2, True)]
    assert segs[1].start_time == pytest.approx(2.016) and segs[1].end_time == pytest.approx(4.032)
    assert segs[2].end_time == pytest.approx(157 * 0.032)              # VAD time runs to the padded last chunk; samples are clamped
    # with partial results every 1.0 s: partials carry the current index, words re-joined by single spaces
    cfg = S.StreamingASRConfig(max_segment_duration=2.0, emit_partial_results=True, partial_result_interval=1.0)
    segs = list(S.StreamingASR(FakeASR(), ScriptedVAD(probs)).transcribe_stream(_ramp(3.0), config=cfg))
    kinds = [(s.is_final, s.segment_index, s.text) for s in segs]
    assert kinds[0] == (False, 0, "s0 e16384")                         # first partial once now - 0 >= 1.0 (chunk 31 -> now 1.024)
    assert (True, 0, "s0 e32256") in kinds and kinds.index((True, 0, "s0 e32256")) > 0
    assert all(s.end_time > s.start_time for s in segs)
    finals = [s for s in segs if s.is_final]
    assert [s.segment_index for s in finals] == list(range(len(finals)))


def test_batched_form_equals_sequential():
    rng = np.random.default_rng(0)
    for trial in range(20):
        probs = []
        while len(probs) < 400:
            probs += [0.9] * int(rng.integers(1, 120)) + [0.05] * int(rng.integers(1, 40))
        cfg = S.StreamingASRConfig(max_segment_duration=float(rng.choice([1.0, 2.5, 10.0])), language="English", context="c")
        audio = _ramp(float(rng.uniform(3.0, 12.0)))
        seq = list(S.StreamingASR(FakeASR(), ScriptedVAD(probs)).transcribe_stream(audio, config=cfg))
        fake = FakeASR()
        bat = S.StreamingASR(fake, ScriptedVAD(probs)).transcribe_stream_batched(audio, config=cfg)
        assert bat == seq and len(seq) >= 1
        assert fake.batch_opts == {"language_ids": [len("language English")], "context_ids": [1]}
    with pytest.raises(ValueError):
        S.StreamingASR(FakeASR(), ScriptedVAD([])).transcribe_stream_batched(_ramp(1.0), config=S.StreamingASRConfig(emit_partial_results=True))
    with pytest.raises(ValueError):
        list(S.StreamingASR(FakeASR(), ScriptedVAD([])).transcribe_stream(_ramp(1.0), sample_rate=8000))
    assert S.StreamingASR(FakeASR(), ScriptedVAD([])).transcribe_stream_batched(_ramp(1.0)) == []


def test_range_guards():
    """StreamingASRTests.swift:107-145: empty / inverted spans are skipped, the end sample is clamped to the buffer."""
    asr = FakeASR()
    # speech "ends" beyond the buffer: 1 s of audio, VAD says speech until 1.5 s
    st = S.StreamingASR(asr, ScriptedVAD([0.9] * 47 + [0.0] * 10))
    segs = list(st.transcribe_stream(_ramp(1.0)))
    assert all(b <= 16000 and a < b for a, b in asr.calls) and len(segs) == 1 and segs[0].text.endswith("e16000")
