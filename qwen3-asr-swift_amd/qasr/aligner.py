"""Host-side mirror of the reference's `Qwen3ForcedAligner` over the C ABI (include/qasr.h, qasr_align*).

Same names and argument meaning as the Swift class (Sources/Qwen3ASR/ForcedAligner.swift):
  * `Qwen3ForcedAligner.from_pretrained(model_dir)`          <- fromPretrained (:385-440), local directory
  * `align(audio, text, sample_rate=16000, language="English")`      <- align (:226-331) -> [AlignedWord]
  * `align_long(audio, text, ...)`                           <- alignLong (:97-180)
The free functions expose the pure-CPU pieces the reference unit-tests (TextPreprocessor.splitIntoWordPairs,
TimestampCorrection.enforceMonotonicity / longestIncreasingSubsequencePositions, findTrailingPlateauStart).
There is no CPU fallback for the model forward.
"""
import ctypes as C
from collections import namedtuple
import numpy as np
from . import _lib
from .model import Qwen3ASRModel, QasrError, _fptr, _iptr

QASR_ERR_UNSUPPORTED = 7     # include/qasr.h

AlignedWord = namedtuple("AlignedWord", "text start_time end_time")     # AudioCommon/Protocols.swift AlignedWord


class UnsupportedLanguage(QasrError):
    """The reference splits this language with Apple's NLTokenizer (not reproducible): pass `words=`."""


def split_word_pairs(text, language="English"):
    """TextPreprocessor.splitIntoWordPairs -> [(surface, cleaned)]."""
    lib = _lib.load(strict=True)
    a, b = C.c_void_p(), C.c_void_p()
    n = lib.qasr_split_words(text.encode("utf-8"), language.encode("utf-8"), C.byref(a), C.byref(b))
    try:
        if n == -QASR_ERR_UNSUPPORTED:
            raise UnsupportedLanguage(language)
        if n < 0:
            raise QasrError(f"qasr_split_words failed ({n})")
        if n == 0:
            return []
        s = C.cast(a, C.c_char_p).value.decode("utf-8").split("\n")
        c = C.cast(b, C.c_char_p).value.decode("utf-8").split("\n")
        return list(zip(s, c))
    finally:
        lib.qasr_free(a)
        lib.qasr_free(b)


def split_words(text, language="English"):
    """TextPreprocessor.splitIntoWords (cleaned forms)."""
    return [c for _, c in split_word_pairs(text, language)]


def lis_positions(values):
    lib = _lib.load(strict=True)
    v = np.ascontiguousarray(values, dtype=np.int32)
    out = np.zeros(max(1, v.shape[0]), dtype=np.int32)
    n = lib.qasr_lis_positions(_iptr(v), v.shape[0], _iptr(out))
    if n < 0:
        raise QasrError("qasr_lis_positions failed")
    return out[:n].tolist()


def enforce_monotonicity(raw):
    lib = _lib.load(strict=True)
    v = np.ascontiguousarray(raw, dtype=np.int32)
    out = np.zeros(max(1, v.shape[0]), dtype=np.int32)
    if lib.qasr_enforce_monotonicity(_iptr(v), v.shape[0], _iptr(out)) != 0:
        raise QasrError("qasr_enforce_monotonicity failed")
    return out[:v.shape[0]].tolist()


def find_trailing_plateau_start(start_times, tolerance=0.1, min_size=5):
    lib = _lib.load(strict=True)
    v = np.ascontiguousarray(start_times, dtype=np.float32)
    return int(lib.qasr_find_trailing_plateau(_fptr(v), v.shape[0], float(tolerance), int(min_size)))


class Qwen3ForcedAligner(Qwen3ASRModel):
    """An engine created from an aligner preset: encoder `.forcedAligner`, decoder `.small`, Linear(1024, 5000)."""

    def __init__(self, preset="aligner-0.6B", model_dir=None, device=0, **capacity):
        super().__init__(preset=preset, model_dir=model_dir, device=device, **capacity)
        if self.cfg.classify_num <= 0:
            raise QasrError(f"preset {preset!r} has no timestamp head")

    @classmethod
    def from_pretrained(cls, model_dir, model_id="aufklarer/Qwen3-ForcedAligner-0.6B-4bit", device=0, **capacity):
        return cls(preset=model_id, model_dir=model_dir, device=device, **capacity)

    @classmethod
    def from_state_dict(cls, sd, preset="aligner-0.6B", device=0, **capacity):
        return super().from_state_dict(sd, preset=preset, device=device, **capacity)

    # ---- stage entry points ---------------------------------------------------------------------
    def prepare(self, text, language="English"):
        """TextPreprocessor.prepareForAlignment -> (slotted ids, timestamp positions, number of words)."""
        raw = text.encode("utf-8")
        ids = np.zeros(4 * len(raw) + 16, dtype=np.int32)
        ts = np.zeros(2 * len(raw) + 16, dtype=np.int32)
        n_ts, n_words = C.c_int32(), C.c_int32()
        n = self.lib.qasr_align_prepare(self.h, raw, language.encode("utf-8"), _iptr(ids), ids.shape[0], _iptr(ts),
                                        ts.shape[0], C.byref(n_ts), C.byref(n_words))
        if n < 0:
            raise QasrError(self.lib.qasr_last_error(self.h).decode())
        return ids[:n].tolist(), ts[:n_ts.value].tolist(), n_words.value

    def align_raw(self, audio, slotted_ids, ts_positions, want_logits=False):
        """-> raw class index per timestamp slot (+ logits [n_ts, classify_num] f32)."""
        pcm = np.ascontiguousarray(audio, dtype=np.float32)
        ids = np.ascontiguousarray(slotted_ids, dtype=np.int32)
        ts = np.ascontiguousarray(ts_positions, dtype=np.int32)
        raw = np.zeros(max(1, ts.shape[0]), dtype=np.int32)
        logits = np.zeros((ts.shape[0], self.cfg.classify_num), dtype=np.float32) if want_logits else None
        self._check(self.lib.qasr_align_raw(self.h, _fptr(pcm), pcm.shape[0], _iptr(ids), ids.shape[0], _iptr(ts),
                                            ts.shape[0], _iptr(raw), _fptr(logits) if want_logits else None))
        raw = raw[:ts.shape[0]].tolist()
        return (raw, logits) if want_logits else raw

    # ---- Qwen3ForcedAligner.align / alignLong -----------------------------------------------------
    def _result(self, out):
        self.last_raw_indices = [out.raw_indices[i] for i in range(out.n_indices)]
        self.last_passes = out.passes
        return [AlignedWord(out.words[i].text.decode("utf-8"), out.words[i].start_time, out.words[i].end_time)
                for i in range(out.n_words)]

    def align(self, audio, text=None, sample_rate=16000, language="English", words=None):
        """-> [AlignedWord].  `words` = [(surface, cleaned)] or [word] replaces the reference's word splitter (needed
        for the NLTokenizer languages)."""
        pcm = np.ascontiguousarray(audio, dtype=np.float32)
        out = _lib.QasrAlignment()
        if words is not None:
            pairs = [(w, w) if isinstance(w, str) else tuple(w) for w in words]
            s = (C.c_char_p * len(pairs))(*[p[0].encode("utf-8") for p in pairs])
            c = (C.c_char_p * len(pairs))(*[p[1].encode("utf-8") for p in pairs])
            self._check(self.lib.qasr_align_words(self.h, _fptr(pcm), pcm.shape[0], sample_rate, s, c, len(pairs), C.byref(out)))
        else:
            self._check(self.lib.qasr_align(self.h, _fptr(pcm), pcm.shape[0], sample_rate, text.encode("utf-8"),
                                            language.encode("utf-8"), C.byref(out)))
        return self._result(out)

    def align_batch(self, clips, texts, sample_rate=16000, language="English"):
        """New batched surface: B clips + B texts in one device pass -> [[AlignedWord]] (single-pass align per clip)."""
        arrs = [np.ascontiguousarray(c, dtype=np.float32) for c in clips]
        B = len(arrs)
        ptrs = (C.POINTER(C.c_float) * B)(*[_fptr(a) for a in arrs])
        ns = (C.c_size_t * B)(*[a.shape[0] for a in arrs])
        txt = (C.c_char_p * B)(*[t.encode("utf-8") for t in texts])
        out = (_lib.QasrAlignment * B)()
        self._check(self.lib.qasr_align_batch(self.h, ptrs, ns, B, sample_rate, txt, language.encode("utf-8"), out))
        res = []
        self.last_raw_batch = []
        for b in range(B):
            o = out[b]
            res.append([AlignedWord(o.words[i].text.decode("utf-8"), o.words[i].start_time, o.words[i].end_time) for i in range(o.n_words)])
            self.last_raw_batch.append([o.raw_indices[i] for i in range(o.n_indices)])
        return res

    def align_long(self, audio, text, sample_rate=16000, language="English"):
        pcm = np.ascontiguousarray(audio, dtype=np.float32)
        out = _lib.QasrAlignment()
        self._check(self.lib.qasr_align_long(self.h, _fptr(pcm), pcm.shape[0], sample_rate, text.encode("utf-8"),
                                             language.encode("utf-8"), C.byref(out)))
        return self._result(out)
