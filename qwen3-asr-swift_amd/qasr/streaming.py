"""Host-side mirror of the reference's `StreamingASR` (VAD-segmented transcription) around a `Qwen3ASRModel`.

Reference: Sources/Qwen3ASR/StreamingASR.swift:7-277 (TranscriptionSegment, StreamingASRConfig, StreamingASR,
longestCommonPrefix) and Sources/SpeechVAD/StreamingVADProcessor.swift:5-227 (VADEvent, four-state hysteresis machine),
Sources/SpeechVAD/Configuration.swift:84-91 (VADConfig.sileroDefault).

What is NOT here: the Silero VAD network itself (SpeechVAD/SileroVAD.swift -- a separate model, outside this repo's hot
path).  `StreamingVADProcessor` therefore takes any `process_chunk(512 float32 samples) -> speech probability` callable
(the Swift shim keeps passing `SileroVADModel.processChunk`); everything downstream of the probability -- hysteresis,
duration filtering, segment bookkeeping, force-splits, partial results, the calls into `transcribe` -- follows the
reference line by line.

New on this engine: `transcribe_stream_batched` first runs the VAD over the whole buffer (the reference's
`transcribeStream` also receives the complete audio array), then transcribes all final segments in ONE ragged
`transcribe_batch`, and yields exactly the segments `transcribe_stream` would (partial results need the sequential form).
"""
from collections import namedtuple
from dataclasses import dataclass, field
from typing import Optional
import numpy as np

CHUNK_SIZE = 512            # SileroVADModel.chunkSize (SileroVAD.swift:78)
SAMPLE_RATE = 16000

TranscriptionSegment = namedtuple("TranscriptionSegment", "text start_time end_time is_final segment_index")
SpeechSegment = namedtuple("SpeechSegment", "start_time end_time")
f32 = np.float32


@dataclass
class VADConfig:
    """VADConfig.sileroDefault (SpeechVAD/Configuration.swift:84-91)."""
    onset: float = 0.5
    offset: float = 0.35
    min_speech_duration: float = 0.25
    min_silence_duration: float = 0.1
    window_duration: float = 0.032
    step_ratio: float = 1.0


@dataclass
class StreamingASRConfig:
    """StreamingASRConfig defaults (StreamingASR.swift:26-53)."""
    max_segment_duration: float = 10.0
    vad_config: VADConfig = field(default_factory=VADConfig)
    language: Optional[str] = None
    max_tokens: int = 448
    emit_partial_results: bool = False
    partial_result_interval: float = 1.0
    context: Optional[str] = None


def longest_common_prefix(a, b):
    """StreamingASR.swift:266-276: case-insensitive, returns the elements of `b`."""
    out = []
    for x, y in zip(a, b):
        if x.lower() != y.lower():
            break
        out.append(y)
    return out


class StreamingVADProcessor:
    """StreamingVADProcessor.swift:39-227.  Events: ("speechStarted", time) / ("speechEnded", SpeechSegment)."""

    def __init__(self, process_chunk, config: VADConfig = None, reset_model=None):
        self.process_chunk, self.config, self.reset_model = process_chunk, config or VADConfig(), reset_model
        self.chunk_duration = f32(CHUNK_SIZE) / f32(SAMPLE_RATE)
        self.reset()

    def reset(self):
        self.buffer = np.zeros(0, dtype=np.float32)
        self.chunk_count = 0
        self.state = ("silence",)
        if self.reset_model:
            self.reset_model()

    @property
    def current_time(self):
        return f32(self.chunk_count) * self.chunk_duration

    def process(self, samples):
        self.buffer = np.concatenate([self.buffer, np.asarray(samples, dtype=np.float32)])
        events = []
        while self.buffer.shape[0] >= CHUNK_SIZE:
            chunk, self.buffer = self.buffer[:CHUNK_SIZE], self.buffer[CHUNK_SIZE:]
            prob = self.process_chunk(chunk)
            time = f32(self.chunk_count) * self.chunk_duration
            self.chunk_count += 1
            events += self._process_prob(f32(prob), time)
        return events

    def flush(self):
        events = []
        if self.buffer.shape[0]:
            last = np.concatenate([self.buffer, np.zeros(CHUNK_SIZE - self.buffer.shape[0], dtype=np.float32)])
            self.buffer = np.zeros(0, dtype=np.float32)
            prob = self.process_chunk(last)
            time = f32(self.chunk_count) * self.chunk_duration
            self.chunk_count += 1
            events += self._process_prob(f32(prob), time)
        end = f32(self.chunk_count) * self.chunk_duration
        kind = self.state[0]
        if kind == "pendingSpeech":
            start = self.state[1]
            if end - start >= f32(self.config.min_speech_duration):
                events += [("speechStarted", start), ("speechEnded", SpeechSegment(start, end))]
        elif kind == "speech":
            events.append(("speechEnded", SpeechSegment(self.state[1], end)))
        elif kind == "pendingSilence":
            events.append(("speechEnded", SpeechSegment(self.state[1], self.state[2])))
        self.state = ("silence",)
        return events

    def _process_prob(self, prob, time):
        c, events = self.config, []
        nxt = time + self.chunk_duration
        kind = self.state[0]
        if kind == "silence":
            if prob >= f32(c.onset):
                self.state = ("pendingSpeech", time)
        elif kind == "pendingSpeech":
            start = self.state[1]
            if prob < f32(c.offset):
                self.state = ("silence",)
            elif nxt - start >= f32(c.min_speech_duration):
                events.append(("speechStarted", start))
                self.state = ("speech", start)
        elif kind == "speech":
            if prob < f32(c.offset):
                self.state = ("pendingSilence", self.state[1], time)
        else:                                              # pendingSilence
            speech_start, silence_start = self.state[1], self.state[2]
            if prob >= f32(c.onset):
                self.state = ("speech", speech_start)
            elif nxt - silence_start >= f32(c.min_silence_duration):
                events.append(("speechEnded", SpeechSegment(speech_start, silence_start)))
                self.state = ("pendingSpeech", time) if prob >= f32(c.onset) else ("silence",)
        return events


class StreamingASR:
    """StreamingASR.swift:58-261.  `asr_model` needs `transcribe(audio, sample_rate=, language=, max_tokens=, context=)`
    (and `transcribe_batch` + `detokenize` for the batched form); `vad_process_chunk` as in StreamingVADProcessor."""

    def __init__(self, asr_model, vad_process_chunk, vad_reset=None):
        self.asr, self.vad_process_chunk, self.vad_reset = asr_model, vad_process_chunk, vad_reset

    def _transcribe(self, samples, a, b, cfg):
        return self.asr.transcribe(samples[a:b], sample_rate=SAMPLE_RATE, language=cfg.language, max_tokens=cfg.max_tokens,
                                   context=cfg.context)

    def _walk(self, samples, cfg, on_span):
        """The control flow of transcribeStream (StreamingASR.swift:84-258).  `on_span(kind, a, b, start_time, end_time)`
        is called for every span the reference transcribes, kind in {"final", "partial"}; returns nothing."""
        proc = StreamingVADProcessor(self.vad_process_chunk, cfg.vad_config, self.vad_reset)
        n = samples.shape[0]
        speech_start = None                   # sample index
        last_partial = f32(0)
        off = 0
        while off < n:
            end = min(off + CHUNK_SIZE, n)
            for ev in proc.process(samples[off:end]):
                if ev[0] == "speechStarted":
                    speech_start = int(ev[1] * f32(SAMPLE_RATE))
                    last_partial = ev[1]
                elif speech_start is not None:
                    seg = ev[1]
                    e = min(int(seg.end_time * f32(SAMPLE_RATE)), n)
                    if speech_start < e:
                        on_span("final", speech_start, e, seg.start_time, seg.end_time)
                    speech_start = None
            if speech_start is not None:
                now = proc.current_time
                sp0 = f32(speech_start) / f32(SAMPLE_RATE)
                dur = now - sp0
                if cfg.emit_partial_results and now - last_partial >= f32(cfg.partial_result_interval):
                    e = min(int(now * f32(SAMPLE_RATE)), n)
                    if not speech_start < e:
                        last_partial = now
                        off = end
                        continue
                    on_span("partial", speech_start, e, sp0, now)
                    last_partial = now
                if dur >= f32(cfg.max_segment_duration):       # force-split (both modes, :176-233)
                    e = min(int(now * f32(SAMPLE_RATE)), n)
                    if speech_start < e:
                        on_span("final", speech_start, e, sp0, now)
                    speech_start = int(now * f32(SAMPLE_RATE))
                    if cfg.emit_partial_results:
                        last_partial = now
            off = end
        for ev in proc.flush():
            if ev[0] == "speechEnded" and speech_start is not None:
                seg = ev[1]
                e = min(int(seg.end_time * f32(SAMPLE_RATE)), n)
                if speech_start < e:
                    on_span("flush", speech_start, e, seg.start_time, seg.end_time)

    def transcribe_stream(self, audio, sample_rate=SAMPLE_RATE, config: StreamingASRConfig = None):
        """Generator of TranscriptionSegment in the reference's order (sequential transcribe calls)."""
        if sample_rate != SAMPLE_RATE:
            raise ValueError("16 kHz input only (the reference resamples with AVAudioConverter: not reproducible)")
        cfg = config or StreamingASRConfig()
        samples = np.ascontiguousarray(audio, dtype=np.float32)
        out, state = [], {"index": 0}

        def on_span(kind, a, b, t0, t1):
            text = self._transcribe(samples, a, b, cfg).strip()
            if kind == "partial":
                words = [w for w in text.split(" ") if w]
                if words:
                    out.append(TranscriptionSegment(" ".join(words), float(t0), float(t1), False, state["index"]))
            elif text:
                out.append(TranscriptionSegment(text, float(t0), float(t1), True, state["index"]))
                if kind == "final":                       # the flush path does not advance the index (:252-257)
                    state["index"] += 1
        self._walk(samples, cfg, on_span)
        yield from out

    def transcribe_stream_batched(self, audio, sample_rate=SAMPLE_RATE, config: StreamingASRConfig = None):
        """Same segments as `transcribe_stream` without partial results, with all spans transcribed in one ragged batch."""
        if sample_rate != SAMPLE_RATE:
            raise ValueError("16 kHz input only")
        cfg = config or StreamingASRConfig()
        if cfg.emit_partial_results:
            raise ValueError("partial results need the sequential form (each one re-transcribes a growing span)")
        samples = np.ascontiguousarray(audio, dtype=np.float32)
        spans = []
        self._walk(samples, cfg, lambda kind, a, b, t0, t1: spans.append((kind, a, b, float(t0), float(t1))))
        if not spans:
            return []
        opts = {}
        if cfg.language:
            opts["language_ids"] = self.asr.encode_text("language " + cfg.language)
        if cfg.context:
            opts["context_ids"] = self.asr.encode_text(cfg.context)
        max_b = getattr(getattr(self.asr, "cfg", None), "max_batch", len(spans)) or len(spans)
        texts = []
        for i in range(0, len(spans), max_b):
            toks = self.asr.transcribe_batch([samples[a:b] for _, a, b, _, _ in spans[i:i + max_b]], max_tokens=cfg.max_tokens, **opts)
            texts += [self.asr.detokenize(t).strip() for t in toks]
        out, index = [], 0
        for (kind, _, _, t0, t1), text in zip(spans, texts):
            if text:
                out.append(TranscriptionSegment(text, t0, t1, True, index))
                if kind == "final":
                    index += 1
        return out
