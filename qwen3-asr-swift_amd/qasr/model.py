"""Host-side mirror of the reference's `Qwen3ASRModel` over the C ABI (include/qasr.h).

Same names, argument meaning and error behaviour as the Swift class:
  * `Qwen3ASRModel.from_pretrained(model_dir)`  <- fromPretrained (Qwen3ASR.swift:608-668); takes a local
    directory in the reference's cache layout (no downloader: out of scope)
  * `transcribe(audio, sample_rate=16000, language=None, max_tokens=448, context=None)` <- Qwen3ASR.swift:131-164;
    never raises for run-time failures, returns "[qasr error: ...]" (the reference returns a bracketed
    diagnostic string, :151-154)
  * `is_loaded / unload() / memory_footprint`  <- Qwen3ASR+Memory.swift:3-18
  * `input_sample_rate`  <- Qwen3ASR+Protocols.swift:6
`transcribe_batch` and the `batch_*` calls are the new batched surface (the reference loops).
There is no CPU fallback: without libqasr.so or a GPU this module raises.
"""
import ctypes as C
import numpy as np
from . import _lib


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class QasrError(RuntimeError):
    pass


def load_wav(path):
    """AudioFileLoader.loadWAV (AudioFileLoader.swift:70-157) through the C ABI -> (float32 samples, rate)."""
    lib = _lib.load(strict=True)
    ptr, n, rate = C.POINTER(C.c_float)(), C.c_size_t(), C.c_int()
    rc = lib.qasr_load_wav(str(path).encode(), C.byref(ptr), C.byref(n), C.byref(rate))
    if rc != 0:
        raise QasrError(f"invalid WAV file ({rc}): {path}")
    try:
        return np.ctypeslib.as_array(ptr, shape=(n.value,)).copy() if n.value else np.zeros(0, np.float32), rate.value
    finally:
        lib.qasr_free(ptr)


class Qwen3ASRModel:
    input_sample_rate = 16000

    def __init__(self, preset="0.6B", model_dir=None, device=0, **capacity):
        self.lib = _lib.load(strict=True)
        self.cfg = _lib.QasrConfig()
        if self.lib.qasr_default_config(preset.encode(), C.byref(self.cfg)) != 0:
            raise QasrError("bad preset")
        self.cfg.device = device
        for k, v in capacity.items():
            setattr(self.cfg, k, v)
        self.h = C.c_void_p()
        rc = self.lib.qasr_create(model_dir.encode() if model_dir else None, C.byref(self.cfg), C.byref(self.h))
        if rc != 0:
            raise QasrError(f"qasr_create failed ({rc}): {self.lib.qasr_last_error(None).decode()}")
        self._keep = []

    # ---- construction ---------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, model_dir, model_id=None, device=0, **capacity):
        """`model_id` only drives size/bits detection, like the reference (Qwen3ASR.swift:617-618)."""
        return cls(preset=model_id or model_dir, model_dir=model_dir, device=device, **capacity)

    @classmethod
    def from_state_dict(cls, sd, preset="0.6B", device=0, **capacity):
        """Upload reference-named torch tensors (qasr.synth) through qasr_set_tensor: bf16 / f32 float tensors, int32
        tensors holding the uint32 words of MLX-quantised weights (qasr.synth.quantize_state_dict)."""
        import torch
        codes = {torch.bfloat16: 1, torch.float32: 0, torch.int32: 3}
        m = cls(preset=preset, device=device, **capacity)
        for name, t in sd.items():
            t = t.contiguous()
            if t.dtype not in codes:
                raise QasrError(f"{name}: expected bf16 / f32 / int32 (uint32 words), got {t.dtype}")
            shape = (C.c_int64 * t.dim())(*t.shape)
            m._check(m.lib.qasr_set_tensor(m.h, name.encode(), C.c_void_p(t.data_ptr()), codes[t.dtype], shape, t.dim()))
        m._check(m.lib.qasr_finalize(m.h))
        return m

    @classmethod
    def borrowed(cls, handle, cfg):
        """A view of an engine somebody else owns (qasr_dp_engine): every method works, close() leaves the engine alone."""
        m = cls.__new__(cls)
        m.lib = _lib.load(strict=True)
        m.cfg = cfg
        m.h = C.c_void_p(handle) if not isinstance(handle, C.c_void_p) else handle
        m._keep = []
        m._borrowed = True
        return m

    def _check(self, rc):
        if rc != 0:
            raise QasrError(f"qasr error {rc}: {self.lib.qasr_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "_borrowed", False):
            self.h = None
            return
        if self.h:
            self.lib.qasr_destroy(self.h)
            self.h = None

    # ---- memory management (ModelMemoryManageable) ---------------------------------------------
    @property
    def is_loaded(self):
        return bool(self.lib.qasr_is_loaded(self.h))

    def unload(self):
        self._check(self.lib.qasr_unload(self.h))

    @property
    def memory_footprint(self):
        return int(self.lib.qasr_memory_footprint(self.h)) if self.is_loaded else 0

    # ---- options ----------------------------------------------------------------------------------
    def set_vocab(self, id_to_token):
        ids = np.array(list(id_to_token.keys()), dtype=np.int32)
        toks = (C.c_char_p * len(ids))(*[t.encode("utf-8") for t in id_to_token.values()])
        self._check(self.lib.qasr_set_vocab(self.h, _iptr(ids), toks, len(ids)))

    def _options(self, max_tokens=0, ignore_eos=False, context_ids=None, language_ids=None,
                 repetition_penalty=1.0, no_repeat_ngram_size=0, temperature=0.0, seed=0):
        o = _lib.QasrOptions()
        o.max_tokens, o.ignore_eos = int(max_tokens), int(bool(ignore_eos))
        o.repetition_penalty, o.no_repeat_ngram_size = float(repetition_penalty), int(no_repeat_ngram_size)
        o.temperature, o.seed = float(temperature), int(seed)
        self._keep = []
        for name, ids in (("context", context_ids), ("language", language_ids)):
            if ids:
                arr = (C.c_int32 * len(ids))(*ids)
                self._keep.append(arr)
                setattr(o, name + "_ids", C.cast(arr, C.POINTER(C.c_int32)))
                setattr(o, "n_" + name, len(ids))
        return o

    # ---- transcribe -------------------------------------------------------------------------------
    def encode_text(self, text):
        """Qwen3Tokenizer.encode (Tokenizer.swift:195-289) with the engine's vocab + merges."""
        raw = text.encode("utf-8")
        ids = np.zeros(4 * len(raw) + 8, dtype=np.int32)
        n = self.lib.qasr_encode_text(self.h, raw, _iptr(ids), ids.shape[0])
        if n < 0:
            raise QasrError(self.lib.qasr_last_error(self.h).decode())
        return ids[:n].tolist()

    def set_merges(self, merges_txt):
        self._check(self.lib.qasr_set_merges(self.h, merges_txt.encode("utf-8")))

    def transcribe(self, audio, sample_rate=16000, language=None, max_tokens=448, context=None,
                   language_ids=None, context_ids=None, **decoding):
        """-> str.  `language` / `context` strings are BPE-encoded like the reference does
        (Qwen3ASR.swift:203-206: context; :228-232: "language " + lang); `decoding` = Qwen3DecodingOptions
        fields (repetition_penalty, no_repeat_ngram_size, temperature, seed)."""
        if context:
            context_ids = self.encode_text(context)
        if language is not None:
            language_ids = self.encode_text("language " + language)
        pcm = np.ascontiguousarray(audio, dtype=np.float32)
        res = _lib.QasrResult()
        o = self._options(max_tokens, False, context_ids, language_ids, **decoding)
        rc = self.lib.qasr_transcribe(self.h, _fptr(pcm), pcm.shape[0], int(sample_rate), C.byref(o), C.byref(res))
        if rc != 0:
            return f"[qasr error: {self.lib.qasr_last_error(self.h).decode()}]"
        return res.text.decode("utf-8")

    def transcribe_tokens(self, audio, **opt):
        return self.transcribe_batch([audio], **opt)[0]

    def transcribe_batch(self, clips, sample_rate=16000, **opt):
        clips = [np.ascontiguousarray(c, dtype=np.float32) for c in clips]
        B = len(clips)
        ptrs = (C.POINTER(C.c_float) * B)(*[_fptr(c) for c in clips])
        ns = (C.c_size_t * B)(*[c.shape[0] for c in clips])
        toks = np.full((B, self.cfg.max_new_tokens + 1), -1, dtype=np.int32)
        lens = np.zeros(B, dtype=np.int32)
        o = self._options(**opt)
        self._check(self.lib.qasr_transcribe_batch(self.h, ptrs, ns, B, sample_rate, C.byref(o), _iptr(toks), _iptr(lens)))
        return [toks[b, :lens[b]].tolist() for b in range(B)]

    def detokenize(self, tokens):
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        buf = C.create_string_buffer(16 * max(1, t.shape[0]) + 64)
        n = self.lib.qasr_detokenize(self.h, _iptr(t), t.shape[0], buf, len(buf))
        if n < 0:
            raise QasrError(self.lib.qasr_last_error(self.h).decode())
        return buf.raw[:n].decode("utf-8")

    # ---- split batch API ----------------------------------------------------------------------------
    def batch_begin(self, clips, **opt):
        self._clips = [np.ascontiguousarray(c, dtype=np.float32) for c in clips]
        B = len(self._clips)
        ptrs = (C.POINTER(C.c_float) * B)(*[_fptr(c) for c in self._clips])
        ns = (C.c_size_t * B)(*[c.shape[0] for c in self._clips])
        o = self._options(**opt)
        self._check(self.lib.qasr_batch_begin(self.h, ptrs, ns, B, C.byref(o)))
        self._B = B

    def batch_stage(self, clips):
        """Stage the NEXT batch under the current one (qasr_batch_stage); adopt it with batch_begin_staged()."""
        self._staged = [np.ascontiguousarray(c, dtype=np.float32) for c in clips]
        B = len(self._staged)
        ptrs = (C.POINTER(C.c_float) * B)(*[_fptr(c) for c in self._staged])
        ns = (C.c_size_t * B)(*[c.shape[0] for c in self._staged])
        self._check(self.lib.qasr_batch_stage(self.h, ptrs, ns, B))

    def batch_begin_staged(self, **opt):
        o = self._options(**opt)
        self._check(self.lib.qasr_batch_begin_staged(self.h, C.byref(o)))
        self._clips = self._staged
        self._B = len(self._clips)

    def batch_rewind(self):
        self._check(self.lib.qasr_batch_rewind(self.h))

    def batch_run(self):
        self._check(self.lib.qasr_batch_run(self.h))

    def batch_sync(self):
        self._check(self.lib.qasr_batch_sync(self.h))

    def batch_tokens(self):
        toks = np.full((self._B, self.cfg.max_new_tokens + 1), -1, dtype=np.int32)
        lens = np.zeros(self._B, dtype=np.int32)
        self._check(self.lib.qasr_batch_tokens(self.h, _iptr(toks), _iptr(lens)))
        return toks, lens

    def batch_timings(self):
        ms = (C.c_float * 5)()
        steps = C.c_int32()
        self._check(self.lib.qasr_batch_timings(self.h, ms, C.byref(steps)))
        return list(ms), steps.value

    def decode_structure(self):
        """(fused_qa, chain, dependent launches per layer) of the current batch's decode step (qasr_decode_structure)."""
        q, c, n = C.c_int(), C.c_int(), C.c_int()
        self._check(self.lib.qasr_decode_structure(self.h, C.byref(q), C.byref(c), C.byref(n)))
        return q.value, c.value, n.value

    def kernel_probe(self, which, reps=20):
        ms, by = C.c_float(), C.c_double()
        self._check(self.lib.qasr_kernel_probe(self.h, which, reps, C.byref(ms), C.byref(by)))
        return ms.value, by.value
