"""Helpers shared by the -m gpu parity tests: they call the product through the C ABI only."""
import ctypes as C
import numpy as np
from qasr import _lib


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _lib_dtype(name):
    return {"f32": 0, "bf16": 1, "f16": 2, "u32": 3}[name]


class Engine:
    def __init__(self, preset="tiny", **overrides):
        self.lib = _lib.load(strict=False)
        self.cfg = _lib.QasrConfig()
        assert self.lib.qasr_default_config(preset.encode(), C.byref(self.cfg)) == 0
        for k, v in overrides.items():
            setattr(self.cfg, k, v)
        self.h = C.c_void_p()
        rc = self.lib.qasr_create(None, C.byref(self.cfg), C.byref(self.h))
        if rc != 0:
            raise RuntimeError(f"qasr_create failed {rc}: {self.lib.qasr_last_error(None)}")

    def check(self, rc):
        if rc != 0:
            raise RuntimeError(f"qasr error {rc}: {self.lib.qasr_last_error(self.h).decode()}")

    def set_tuning(self, key, value):
        assert self.lib.qasr_set_tuning(key.encode(), int(value)) == 0, key

    def close(self):
        if self.h:
            self.lib.qasr_destroy(self.h)
            self.h = None

    def load_state_dict(self, sd):
        """Upload a reference-named torch state dict tensor by tensor, then finalize: bf16 / f32 float tensors, int32
        tensors = the uint32 words of MLX-quantised weights."""
        import torch
        codes = {torch.bfloat16: "bf16", torch.float32: "f32", torch.int32: "u32"}
        for name, t in sd.items():
            t = t.contiguous()
            assert t.dtype in codes, (name, t.dtype)
            shape = (C.c_int64 * t.dim())(*t.shape)
            self.check(self.lib.qasr_set_tensor(self.h, name.encode(), C.c_void_p(t.data_ptr()),
                                                _lib_dtype(codes[t.dtype]), shape, t.dim()))
        self.check(self.lib.qasr_finalize(self.h))

    def encode(self, mel):
        mel = np.ascontiguousarray(mel, dtype=np.float32)
        n_tok = self.lib.qasr_num_audio_tokens(self.h, mel.shape[1])
        out = np.empty((n_tok, self.cfg.enc_out_dim), dtype=np.float32)
        self.check(self.lib.qasr_encode(self.h, fptr(mel), mel.shape[1], fptr(out)))
        return out

    def options(self, max_tokens=0, ignore_eos=False, context_ids=None, language_ids=None,
                repetition_penalty=1.0, no_repeat_ngram_size=0, temperature=0.0, seed=0):
        o = _lib.QasrOptions()
        o.max_tokens, o.ignore_eos = max_tokens, int(ignore_eos)
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

    def prefill_logits(self, audio_embeds, **opt):
        emb = np.ascontiguousarray(audio_embeds, dtype=np.float32)
        logits = np.empty(self.cfg.vocab, dtype=np.float32)
        o = self.options(**opt)
        self.check(self.lib.qasr_prefill_logits(self.h, fptr(emb), emb.shape[0], C.byref(o), fptr(logits)))
        return logits

    def decode_forced(self, tokens):
        t = np.ascontiguousarray(tokens, dtype=np.int32)
        logits = np.empty((t.shape[0], self.cfg.vocab), dtype=np.float32)
        self.check(self.lib.qasr_decode_forced(self.h, iptr(t), t.shape[0], fptr(logits)))
        return logits

    def transcribe_batch(self, clips, **opt):
        clips = [np.ascontiguousarray(c, dtype=np.float32) for c in clips]
        B = len(clips)
        ptrs = (C.POINTER(C.c_float) * B)(*[fptr(c) for c in clips])
        ns = (C.c_size_t * B)(*[c.shape[0] for c in clips])
        stride = self.cfg.max_new_tokens + 1
        toks = np.full((B, stride), -7, dtype=np.int32)
        lens = np.zeros(B, dtype=np.int32)
        o = self.options(**opt)
        self.check(self.lib.qasr_transcribe_batch(self.h, ptrs, ns, B, 16000, C.byref(o), iptr(toks), iptr(lens)))
        return [toks[b, :lens[b]].tolist() for b in range(B)]

    def decode_structure(self):
        """(fused q|k|v + attention launch?, chain mode, dependent launches per decoder layer) of the prepared batch's decode step."""
        f, c, n = C.c_int(), C.c_int(), C.c_int()
        self.check(self.lib.qasr_decode_structure(self.h, C.byref(f), C.byref(c), C.byref(n)))
        return f.value, c.value, n.value

    def timings(self):
        ms = (C.c_float * 5)()
        steps = C.c_int32()
        self.check(self.lib.qasr_batch_timings(self.h, ms, C.byref(steps)))
        return list(ms), steps.value

    def mel(self, pcm):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        T = self.lib.qasr_num_mel_frames(pcm.shape[0])
        out = np.empty((128, T), dtype=np.float32)
        self.check(self.lib.qasr_mel(self.h, fptr(pcm), pcm.shape[0], fptr(out)))
        return out
