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

    def close(self):
        if self.h:
            self.lib.qasr_destroy(self.h)
            self.h = None

    def load_state_dict(self, sd):
        """Upload a reference-named torch state dict (bf16) tensor by tensor, then finalize."""
        import torch
        for name, t in sd.items():
            t = t.contiguous()
            assert t.dtype == torch.bfloat16, (name, t.dtype)
            shape = (C.c_int64 * t.dim())(*t.shape)
            self.check(self.lib.qasr_set_tensor(self.h, name.encode(), C.c_void_p(t.data_ptr()),
                                                _lib_dtype("bf16"), shape, t.dim()))
        self.check(self.lib.qasr_finalize(self.h))

    def encode(self, mel):
        mel = np.ascontiguousarray(mel, dtype=np.float32)
        n_tok = self.lib.qasr_num_audio_tokens(self.h, mel.shape[1])
        out = np.empty((n_tok, self.cfg.enc_out_dim), dtype=np.float32)
        self.check(self.lib.qasr_encode(self.h, fptr(mel), mel.shape[1], fptr(out)))
        return out

    def mel(self, pcm):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        T = self.lib.qasr_num_mel_frames(pcm.shape[0])
        out = np.empty((128, T), dtype=np.float32)
        self.check(self.lib.qasr_mel(self.h, fptr(pcm), pcm.shape[0], fptr(out)))
        return out
