"""Helpers shared by the -m gpu parity tests: they call the product through the C ABI only."""
import ctypes as C
import numpy as np
from qasr import _lib


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


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

    def mel(self, pcm):
        pcm = np.ascontiguousarray(pcm, dtype=np.float32)
        T = self.lib.qasr_num_mel_frames(pcm.shape[0])
        out = np.empty((128, T), dtype=np.float32)
        self.check(self.lib.qasr_mel(self.h, fptr(pcm), pcm.shape[0], fptr(out)))
        return out
