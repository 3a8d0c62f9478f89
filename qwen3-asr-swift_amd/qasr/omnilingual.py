"""Host-side mirror of the reference's `OmnilingualASRMLXModel` over the C ABI (include/qasr.h, qasr_ctc_*).

Same names, argument meaning and error behaviour as the Swift class (Sources/OmnilingualASR/MLX/OmnilingualMLXModel.swift):
  * `OmnilingualASRMLXModel.from_pretrained(model_dir, model_id=...)` <- fromPretrained (:45-119): local directory with
    model.safetensors + tokenizer.model; variant and bits detected from the model id like detectVariant / detectBits
  * `transcribe_audio(audio, sample_rate)` <- transcribeAudio (:141-193): raises for > 40 s input (the reference throws),
    "" for empty input; `transcribe(audio, sample_rate, language)` <- the SpeechRecognitionModel conformance
    (OmnilingualASRMLXModel+Protocols.swift:3-15): never raises, "" on failure, `language` ignored
  * `is_loaded / unload() / memory_footprint`, `sample_rate`
`transcribe_batch` is the new batched surface.  There is no CPU fallback: without libqasr.so or a GPU this module raises.
"""
import ctypes as C
import numpy as np
from . import _lib
from .model import QasrError

MAX_AUDIO_SECONDS = 40.0


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class OmnilingualASRMLXModel:
    sample_rate = 16000
    input_sample_rate = 16000

    def __init__(self, variant="300M", model_dir=None, device=0, **capacity):
        self.lib = _lib.load(strict=True)
        self.cfg = _lib.QasrCtcConfig()
        if self.lib.qasr_ctc_default_config(variant.encode(), C.byref(self.cfg)) != 0:
            raise QasrError(f"unknown Omnilingual variant {variant!r}")
        self.cfg.device = device
        for k, v in capacity.items():
            setattr(self.cfg, k, v)
        self.h = C.c_void_p()
        rc = self.lib.qasr_ctc_create(model_dir.encode() if model_dir else None, C.byref(self.cfg), C.byref(self.h))
        if rc != 0:
            raise QasrError(f"qasr_ctc_create failed ({rc}): {self.lib.qasr_ctc_last_error(None).decode()}")

    @classmethod
    def from_pretrained(cls, model_dir, model_id=None, device=0, **capacity):
        return cls(variant=model_id or model_dir, model_dir=model_dir, device=device, **capacity)

    @classmethod
    def from_state_dict(cls, sd, variant="300M", device=0, **capacity):
        """fairseq2-named tensors (qasr.synth.synth_omnilingual_state_dict): f32 / bf16 floats, int32 = uint32 words."""
        import torch
        codes = {torch.bfloat16: 1, torch.float32: 0, torch.int32: 3, torch.float16: 2}
        m = cls(variant=variant, device=device, **capacity)
        for name, t in sd.items():
            t = t.contiguous()
            if t.dtype not in codes:
                raise QasrError(f"{name}: unsupported dtype {t.dtype}")
            shape = (C.c_int64 * t.dim())(*t.shape)
            m._check(m.lib.qasr_ctc_set_tensor(m.h, name.encode(), C.c_void_p(t.data_ptr()), codes[t.dtype], shape, t.dim()))
        m._check(m.lib.qasr_ctc_finalize(m.h))
        return m

    @classmethod
    def from_synthetic(cls, variant="300M", device=0, seed=0, **capacity):
        """Seeded random weights of the named variant streamed tensor by tensor into the engine (benchmarks: no checkpoint offline)."""
        from . import synth
        m = cls(variant=variant, device=device, **capacity)

        def put(name, t):
            t = t.contiguous()
            shape = (C.c_int64 * t.dim())(*t.shape)
            m._check(m.lib.qasr_ctc_set_tensor(m.h, name.encode(), C.c_void_p(t.data_ptr()), 0, shape, t.dim()))
        synth.synth_omnilingual_state_dict(m.cfg, seed=seed, bits=0, sink=put)
        m._check(m.lib.qasr_ctc_finalize(m.h))
        return m

    def _check(self, rc):
        if rc != 0:
            raise QasrError(f"qasr error {rc}: {self.lib.qasr_ctc_last_error(self.h).decode()}")

    def close(self):
        if self.h:
            self.lib.qasr_ctc_destroy(self.h)
            self.h = None

    @property
    def is_loaded(self):
        return bool(self.lib.qasr_ctc_is_loaded(self.h))

    def unload(self):
        self._check(self.lib.qasr_ctc_unload(self.h))

    @property
    def memory_footprint(self):
        return int(self.lib.qasr_ctc_memory_footprint(self.h)) if self.is_loaded else 0

    def set_pieces(self, pieces):
        """pieces: list of (text, type) in id order (SentencePiece types: 1 normal, 2 unknown, 3 control, 5 unused, 6 byte)."""
        texts = (C.c_char_p * len(pieces))(*[p[0].encode("utf-8") for p in pieces])
        types = np.array([p[1] for p in pieces], dtype=np.int32)
        self._check(self.lib.qasr_ctc_set_pieces(self.h, texts, _iptr(types), len(pieces)))

    def num_frames(self, n_samples):
        return self.lib.qasr_ctc_num_frames(n_samples)

    def transcribe_batch(self, clips, sample_rate=16000):
        """-> collapsed token ids per clip."""
        clips = [np.ascontiguousarray(c, dtype=np.float32) for c in clips]
        B = len(clips)
        stride = max(1, max(self.num_frames(c.shape[0]) for c in clips))
        ptrs = (C.POINTER(C.c_float) * B)(*[_fptr(c) for c in clips])
        ns = (C.c_size_t * B)(*[c.shape[0] for c in clips])
        ids = np.full((B, stride), -1, dtype=np.int32)
        lens = np.zeros(B, dtype=np.int32)
        self._check(self.lib.qasr_ctc_transcribe_batch(self.h, ptrs, ns, B, sample_rate, _iptr(ids), stride, _iptr(lens)))
        return [ids[b, :lens[b]].tolist() for b in range(B)]

    def logits(self, audio):
        pcm = np.ascontiguousarray(audio, dtype=np.float32)
        out = np.empty((self.num_frames(pcm.shape[0]), self.cfg.vocab), dtype=np.float32)
        self._check(self.lib.qasr_ctc_logits(self.h, _fptr(pcm), pcm.shape[0], _fptr(out)))
        return out

    def detokenize(self, ids):
        t = np.ascontiguousarray(ids, dtype=np.int32)
        buf = C.create_string_buffer(64 * max(1, t.shape[0]) + 64)
        n = self.lib.qasr_ctc_detokenize(self.h, _iptr(t), t.shape[0], buf, len(buf))
        if n < 0:
            raise QasrError(self.lib.qasr_ctc_last_error(self.h).decode())
        return buf.raw[:n].decode("utf-8")

    def transcribe_audio(self, audio, sample_rate=16000, language=None):
        """Raises like the reference's throwing `transcribeAudio` (40 s cap, unloaded model)."""
        pcm = np.ascontiguousarray(audio, dtype=np.float32)
        text = C.c_char_p()
        self._check(self.lib.qasr_ctc_transcribe(self.h, _fptr(pcm), pcm.shape[0], int(sample_rate), C.byref(text)))
        return text.value.decode("utf-8")

    def transcribe(self, audio, sample_rate=16000, language=None):
        """SpeechRecognitionModel.transcribe: never raises, "" on failure (OmnilingualASRMLXModel+Protocols.swift:8-14)."""
        try:
            return self.transcribe_audio(audio, sample_rate, language)
        except QasrError:
            return ""

    def timings(self):
        ms = (C.c_float * 4)()
        self._check(self.lib.qasr_ctc_timings(self.h, ms))
        return list(ms)
