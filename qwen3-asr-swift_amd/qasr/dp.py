"""Utterance-batch data parallelism inside ONE process over the C ABI (qasr_dp_*, include/qasr.h): one engine + one host thread per listed
GPU, contiguous clip blocks, tokens gathered by each engine's device -> host copy into one [B, max_new_tokens + 1] block.  The reference
has no counterpart beyond its sequential file loop (Sources/AudioCLILib/TranscribeBatchCommand.swift:82-93).  The one-process-per-GPU form
of the same partition (torch.distributed, RCCL all_gather) is qasr/dist.py."""
import ctypes as C

import numpy as np

from . import _lib
from .model import Qwen3ASRModel, QasrError, _fptr, _iptr


class Qwen3ASRDataParallel:
    def __init__(self, devices, preset="0.6B", model_dir=None, **capacity):
        self.lib = _lib.load(strict=True)
        self.cfg = _lib.QasrConfig()
        if self.lib.qasr_default_config(preset.encode(), C.byref(self.cfg)) != 0:
            raise QasrError("bad preset")
        for k, v in capacity.items():
            setattr(self.cfg, k, v)
        dev = np.asarray(list(devices), dtype=np.int32)
        self.h = C.c_void_p()
        rc = self.lib.qasr_dp_create(model_dir.encode() if model_dir else None, C.byref(self.cfg), _iptr(dev), len(dev), C.byref(self.h))
        if rc != 0:
            raise QasrError(f"qasr_dp_create failed ({rc}): {self.lib.qasr_dp_last_error(None).decode()}")
        self.n_devices = self.lib.qasr_dp_n_devices(self.h)
        self._in_flight = {}                  # ticket -> (clips, pointer arrays, lengths, B, options, option arrays)
        self._keep = []

    @classmethod
    def from_state_dict(cls, sd, devices, preset="0.6B", **capacity):
        import torch
        codes = {torch.bfloat16: 1, torch.float32: 0, torch.int32: 3}
        m = cls(devices, preset=preset, **capacity)
        for name, t in sd.items():
            t = t.contiguous()
            shape = (C.c_int64 * t.dim())(*t.shape)
            m._check(m.lib.qasr_dp_set_tensor(m.h, name.encode(), C.c_void_p(t.data_ptr()), codes[t.dtype], shape, t.dim()))
        m._check(m.lib.qasr_dp_finalize(m.h))
        return m

    @classmethod
    def from_pretrained(cls, model_dir, devices, model_id=None, **capacity):
        return cls(devices, preset=model_id or model_dir, model_dir=model_dir, **capacity)

    def engine(self, i=0):
        """Engine i as a Qwen3ASRModel view (tokenizer, single-clip calls, stage entry points); owned by this object."""
        h = self.lib.qasr_dp_engine(self.h, int(i))
        if not h:
            raise QasrError(f"no engine {i}")
        return Qwen3ASRModel.borrowed(h, self.cfg)

    def _check(self, rc):
        if rc != 0:
            raise QasrError(f"qasr error {rc}: {self.lib.qasr_dp_last_error(self.h).decode()}")

    def close(self):
        if self.h:
            self.lib.qasr_dp_destroy(self.h)          # waits for batches still in flight
            self.h = None
            self._in_flight.clear()

    def transcribe_batch(self, clips, sample_rate=16000, **opt):
        """-> token id lists, clip order preserved.  Options as Qwen3ASRModel.transcribe_batch."""
        clips = [np.ascontiguousarray(c, dtype=np.float32) for c in clips]
        B = len(clips)
        stride = self.cfg.max_new_tokens + 1
        ptrs = (C.POINTER(C.c_float) * B)(*[_fptr(c) for c in clips])
        ns = (C.c_size_t * B)(*[c.shape[0] for c in clips])
        toks = np.full((B, stride), -1, dtype=np.int32)
        lens = np.zeros(B, dtype=np.int32)
        o = Qwen3ASRModel._options(self, **opt)
        self._check(self.lib.qasr_dp_transcribe_batch(self.h, ptrs, ns, B, int(sample_rate), C.byref(o), _iptr(toks), _iptr(lens)))
        return [toks[b, :lens[b]].tolist() for b in range(B)]

    def submit(self, clips, sample_rate=16000, **opt):
        """One whole batch to the next engine (round robin), without waiting: -> ticket for collect().  With a device listed twice the
        GPU keeps two passes in flight (qasr_dp_submit)."""
        clips = [np.ascontiguousarray(c, dtype=np.float32) for c in clips]
        B = len(clips)
        ptrs = (C.POINTER(C.c_float) * B)(*[_fptr(c) for c in clips])
        ns = (C.c_size_t * B)(*[c.shape[0] for c in clips])
        o = Qwen3ASRModel._options(self, **opt)
        t = C.c_int64(-1)
        self._check(self.lib.qasr_dp_submit(self.h, ptrs, ns, B, int(sample_rate), C.byref(o), C.byref(t)))
        self._in_flight[t.value] = (clips, ptrs, ns, B, o, self._keep)      # all of it is read by the engine's thread until collect()
        self._keep = []
        return t.value

    def collect(self, ticket, raw=False):
        """Waits for that batch: -> token id lists (raw: the [B, max_new_tokens + 1] block and the lens)."""
        B = self._in_flight[ticket][3] if ticket in self._in_flight else 0
        toks = np.full((B, self.cfg.max_new_tokens + 1), -1, dtype=np.int32)
        lens = np.zeros(B, dtype=np.int32)
        try:
            self._check(self.lib.qasr_dp_collect(self.h, int(ticket), _iptr(toks), _iptr(lens)))
        finally:
            self._in_flight.pop(ticket, None)
        return (toks, lens) if raw else [toks[b, :lens[b]].tolist() for b in range(B)]

    def timings(self):
        ms = (C.c_float * self.n_devices)()
        self._check(self.lib.qasr_dp_timings(self.h, ms, self.n_devices))
        return list(ms)
