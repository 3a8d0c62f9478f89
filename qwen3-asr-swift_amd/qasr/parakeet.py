"""Host-side mirror of the reference's Parakeet-TDT / Nemotron-streaming / Parakeet-EOU interfaces over the C ABI (include/qasr.h, last
section) -- BASELINE configs[4], the restatable slice.

Same names and argument meaning as the reference types so that the parity tests read like the reference's own tests:
  MelPreprocessor.extract                     Sources/ParakeetASR/MelPreprocessor.swift:52-202
  StreamingMelPreprocessor.extract / extractRaw / extractStreaming / resetRunningStats
                                              Sources/ParakeetStreamingASR/StreamingMelPreprocessor.swift, Sources/NemotronStreamingASR/StreamingMelPreprocessor.swift
  TDTGreedyDecoder.decode                     Sources/ParakeetASR/TDTGreedyDecoder.swift:45-143
  RNNTGreedyDecoder.decode                    Sources/NemotronStreamingASR/RNNTGreedyDecoder.swift:38-90, Sources/ParakeetStreamingASR/RNNTGreedyDecoder.swift:58-126
  ParakeetVocabulary / NemotronVocabulary     Sources/ParakeetASR/Vocabulary.swift, Sources/NemotronStreamingASR/Vocabulary.swift
  StreamingSession.pushAudio / finalize       Sources/NemotronStreamingASR/StreamingSession.swift:110-231
The encoder, prediction network and joint are opaque CoreML bundles in the reference; here they are Python callables of the caller
(decoder(token), joint(frame) -> logits [, duration logits], encoder(mel) -> valid frames).  The mel runs on the GPU (no CPU fallback).
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from .model import QasrError

_F = C.POINTER(C.c_float)
_I = C.POINTER(C.c_int32)
TDT, EOU, RAW, EOU_STREAMING = 0, 1, 2, 3
N_MELS, HOP = 128, 160


def _fptr(a):
    return a.ctypes.data_as(_F)


def _iptr(a):
    return a.ctypes.data_as(_I)


class NemoMelDevice:
    """One device context for the batched front-ends (qasr_nemo_mel_*)."""

    def __init__(self, device=0, max_streams=64, max_samples=16000 * 40, fft_scale=2.0):
        self.lib = _lib.load(strict=True)
        self.h = C.c_void_p()
        rc = self.lib.qasr_nemo_mel_create(int(device), int(max_streams), int(max_samples), float(fft_scale), C.byref(self.h))
        if rc != 0:
            raise QasrError(f"qasr error {rc}: {self.lib.qasr_nemo_mel_last_error(None).decode()}")

    def close(self):
        if self.h:
            self.lib.qasr_nemo_mel_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def extract_batch(self, variant, clips, stream_ids=None, fit=0):
        """-> (mel [B, 128, frames] float32, mel_len [B])."""
        clips = [np.ascontiguousarray(c, dtype=np.float32) for c in clips]
        B = len(clips)
        frames = fit if fit > 0 else max(1, max(self.lib.qasr_nemo_mel_num_frames(c.shape[0]) if c.shape[0] else 1 for c in clips))
        ptrs = (_F * B)(*[_fptr(c) for c in clips])
        ns = (C.c_size_t * B)(*[c.shape[0] for c in clips])
        out = np.empty((B, N_MELS, frames), dtype=np.float32)
        lens = np.zeros(B, dtype=np.int32)
        sid = None if stream_ids is None else np.ascontiguousarray(stream_ids, dtype=np.int32)
        rc = self.lib.qasr_nemo_mel_extract(self.h, int(variant), ptrs, ns, B, _iptr(sid) if sid is not None else None, _fptr(out), frames,
                                            _iptr(lens), int(fit))
        if rc != 0:
            raise QasrError(f"qasr error {rc}: {self.lib.qasr_nemo_mel_last_error(self.h).decode()}")
        return out, lens

    def reset_stats(self, stream=-1):
        rc = self.lib.qasr_nemo_mel_reset_stats(self.h, int(stream))
        if rc != 0:
            raise QasrError(f"qasr error {rc}: {self.lib.qasr_nemo_mel_last_error(self.h).decode()}")

    def timing(self):
        ms, g = C.c_float(), C.c_int()
        self.lib.qasr_nemo_mel_timing(self.h, C.byref(ms), C.byref(g))
        return ms.value, bool(g.value)


class MelPreprocessor:
    """ParakeetASR.MelPreprocessor: extract(audio) -> (mel [1, 128, T] float16 values, melLength)."""

    def __init__(self, dev=None, **kw):
        self.dev = dev or NemoMelDevice(**kw)

    def extract(self, audio):
        mel, lens = self.dev.extract_batch(TDT, [audio])
        return mel[:1].astype(np.float16), int(lens[0])


class StreamingMelPreprocessor:
    """ParakeetStreamingASR / NemotronStreamingASR StreamingMelPreprocessor (one stream = one running-statistics slot)."""

    def __init__(self, dev=None, stream=0, **kw):
        self.dev = dev or NemoMelDevice(**kw)
        self.stream = stream

    def extract(self, audio):
        mel, lens = self.dev.extract_batch(EOU, [audio])
        return mel[:1], int(lens[0])

    def extract_raw(self, audio):
        mel, lens = self.dev.extract_batch(RAW, [audio])
        return mel[:1], int(lens[0])

    def extract_streaming(self, audio):
        mel, lens = self.dev.extract_batch(EOU_STREAMING, [audio], stream_ids=[self.stream])
        return mel[:1], int(lens[0])

    def reset_running_stats(self):
        self.dev.reset_stats(self.stream)


def transducer_config(model):
    lib = _lib.load(strict=True)
    cfg = _lib.QasrTransducerConfig()
    if lib.qasr_transducer_default_config(model.encode(), C.byref(cfg)) != 0:
        raise QasrError(f"unknown transducer model {model!r}")
    return cfg


def _callbacks(decoder, joint, n_tok, n_dur):
    """Python callables -> the C callback struct.  joint(frame) returns token logits (and duration logits for TDT)."""
    def dec(_ctx, token):
        try:
            decoder(int(token))
            return 0
        except Exception:        # noqa: BLE001 -- an exception must not unwind through C
            return 1

    def jnt(_ctx, frame, tl, dl):
        try:
            r = joint(int(frame))
            if n_dur:
                t, d = r
                C.memmove(dl, np.ascontiguousarray(d, dtype=np.float32)[:n_dur].ctypes.data, 4 * n_dur)
            else:
                t = r
            C.memmove(tl, np.ascontiguousarray(t, dtype=np.float32)[:n_tok].ctypes.data, 4 * n_tok)
            return 0
        except Exception:        # noqa: BLE001
            return 1

    cb = _lib.QasrTransducerCallbacks(None, _lib.TD_DECODER_FN(dec), _lib.TD_JOINT_FN(jnt))
    return cb


class TDTGreedyDecoder:
    def __init__(self, config=None):
        self.lib = _lib.load(strict=True)
        self.cfg = config or transducer_config("parakeet-tdt")

    def decode(self, encoded_length, decoder, joint, cap=4096):
        """-> (tokens, token log-probs, confidence)."""
        cb = _callbacks(decoder, joint, self.cfg.vocab_size + 1, self.cfg.n_durations)
        toks = np.zeros(cap, dtype=np.int32)
        lps = np.zeros(cap, dtype=np.float32)
        conf = C.c_float()
        n = self.lib.qasr_tdt_greedy_decode(C.byref(self.cfg), C.byref(cb), int(encoded_length), _iptr(toks), _fptr(lps), cap, C.byref(conf))
        if n < 0:
            raise QasrError(f"qasr error {-n}")
        return toks[:n].tolist(), lps[:n].copy(), conf.value


class RNNTGreedyDecoder:
    def __init__(self, config=None, model="nemotron-streaming"):
        self.lib = _lib.load(strict=True)
        self.cfg = config or transducer_config(model)

    def decode(self, encoded_length, decoder, joint, frame_offset=0, cap=4096):
        """-> (tokens, token log-probs, eouDetected)."""
        cb = _callbacks(decoder, joint, self.cfg.vocab_size + 1, 0)
        toks = np.zeros(cap, dtype=np.int32)
        lps = np.zeros(cap, dtype=np.float32)
        eou = C.c_int32()
        n = self.lib.qasr_rnnt_greedy_decode(C.byref(self.cfg), C.byref(cb), int(encoded_length), int(frame_offset), _iptr(toks), _fptr(lps), cap,
                                             C.byref(eou))
        if n < 0:
            raise QasrError(f"qasr error {-n}")
        return toks[:n].tolist(), lps[:n].copy(), bool(eou.value)


@dataclass
class WordConfidence:
    word: str
    confidence: float


class _Vocabulary:
    STYLE = 0

    def __init__(self, id_to_token=None, path=None):
        self.lib = _lib.load(strict=True)
        self.h = C.c_void_p()
        if path is not None:
            rc = self.lib.qasr_sp_vocab_load(str(path).encode(), self.STYLE, C.byref(self.h))
        else:
            items = sorted((id_to_token or {}).items())
            ids = np.array([k for k, _ in items], dtype=np.int32)
            pcs = (C.c_char_p * len(items))(*[v.encode("utf-8") for _, v in items])
            rc = self.lib.qasr_sp_vocab_create(_iptr(ids), pcs, len(items), self.STYLE, C.byref(self.h))
        if rc != 0:
            raise QasrError(f"qasr error {rc}: vocabulary")

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.qasr_sp_vocab_destroy(self.h)
            self.h = None

    @property
    def count(self):
        return self.lib.qasr_sp_vocab_count(self.h)

    def decode(self, token_ids):
        t = np.ascontiguousarray(token_ids, dtype=np.int32)
        buf = C.create_string_buffer(64 * max(1, t.shape[0]) + 64)
        n = self.lib.qasr_sp_vocab_decode(self.h, _iptr(t), t.shape[0], buf, len(buf))
        if n < 0:
            raise QasrError("vocabulary decode failed")
        return buf.raw[:n].decode("utf-8")

    def decode_words(self, token_ids, log_probs):
        t = np.ascontiguousarray(token_ids, dtype=np.int32)
        lp = np.ascontiguousarray(log_probs, dtype=np.float32)
        buf = C.create_string_buffer(64 * max(1, t.shape[0]) + 64)
        conf = np.zeros(max(1, t.shape[0]), dtype=np.float32)
        n = self.lib.qasr_sp_vocab_decode_words(self.h, _iptr(t), t.shape[0], _fptr(lp), lp.shape[0], buf, len(buf), _fptr(conf), conf.shape[0])
        if n < 0:
            raise QasrError("vocabulary decodeWords failed")
        words = buf.value.decode("utf-8").split("\n") if n else []
        return [WordConfidence(w, float(c)) for w, c in zip(words, conf[:n])]


class ParakeetVocabulary(_Vocabulary):
    STYLE = 0


class NemotronVocabulary(_Vocabulary):
    STYLE = 1


ParakeetEOUVocabulary = NemotronVocabulary


@dataclass
class PartialTranscript:
    text: str
    is_final: bool
    confidence: float
    segment_index: int = 0


class StreamingSession:
    """NemotronStreamingASR.StreamingSession for ONE stream (the multi-stream driver below batches the mel of many)."""

    MEL_FRAMES, OUTPUT_FRAMES, SUBSAMPLING = 17, 2, 8            # NemotronStreamingConfig.default.streaming

    def __init__(self, mel, encoder, decoder, joint, vocabulary, config=None):
        self.lib = _lib.load(strict=True)
        self.mel, self.encoder, self.decoder, self.joint, self.vocab = mel, encoder, decoder, joint, vocabulary
        self.rnnt = RNNTGreedyDecoder(config)
        self.samples_per_chunk = self.MEL_FRAMES * HOP
        self.shift = self.OUTPUT_FRAMES * self.SUBSAMPLING * HOP
        self.ck = C.c_void_p()
        assert self.lib.qasr_stream_chunker_create(self.samples_per_chunk, self.shift, C.byref(self.ck)) == 0
        self.tokens, self.log_probs, self.chunks = [], [], []
        decoder(self.rnnt.cfg.blank_id)                          # StreamingSession.swift:93-99

    def __del__(self):
        if getattr(self, "ck", None):
            self.lib.qasr_stream_chunker_destroy(self.ck)
            self.ck = None

    def _confidence(self):
        lp = np.ascontiguousarray(self.log_probs, dtype=np.float32)
        return float(self.lib.qasr_transducer_confidence(_fptr(lp), lp.shape[0]))

    def push_audio(self, samples):
        s = np.ascontiguousarray(samples, dtype=np.float32)
        assert self.lib.qasr_stream_chunker_push(self.ck, _fptr(s), s.shape[0]) == 0
        out = []
        chunk = np.empty(self.samples_per_chunk, dtype=np.float32)
        while self.lib.qasr_stream_chunker_pop(self.ck, _fptr(chunk)):
            p = self._process(chunk.copy())
            if p is not None:
                out.append(p)
        return out

    def finalize(self):
        chunk = np.empty(self.samples_per_chunk, dtype=np.float32)
        if self.lib.qasr_stream_chunker_flush(self.ck, _fptr(chunk)):
            self._process(chunk.copy())
        if not self.tokens:
            return []
        return [PartialTranscript(self.vocab.decode(self.tokens), True, self._confidence())]

    def _process(self, chunk):
        self.chunks.append(chunk)
        mel, lens = self.mel.dev.extract_batch(RAW, [chunk], fit=self.MEL_FRAMES)
        if lens[0] <= 0:
            return None
        valid = self.encoder(mel[0])
        n = min(self.OUTPUT_FRAMES, valid)
        if n <= 0:
            return None
        toks, lps, _ = self.rnnt.decode(n, self.decoder, self.joint)
        self.tokens += toks
        self.log_probs += list(lps)
        text = self.vocab.decode(self.tokens)
        if text == "":
            return None
        return PartialTranscript(text, False, self._confidence())
