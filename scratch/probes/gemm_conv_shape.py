import ctypes as C, os, sys
ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "qwen3-asr-swift_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gpu_util
eng = gpu_util.Engine("tiny", max_audio_seconds=2)
P16, PF = C.POINTER(C.c_uint16), C.POINTER(C.c_float)
for M, N, K in ((199680, 480, 4320), (199680, 512, 4352)):
    rng = np.random.default_rng(1)
    A = (rng.integers(0, 2**16, size=(M, K), dtype=np.uint16) & 0x3fff) | 0x3c00
    W = (rng.integers(0, 2**16, size=(N, K), dtype=np.uint16) & 0x3fff) | 0x3c00
    out = np.empty((M, N), np.float32)
    for form in (1, 2):
        ms = C.c_float()
        eng.check(eng.lib.qasr_gemm_probe(eng.h, A.ctypes.data_as(P16), W.ctypes.data_as(P16), None, M, N, K, form, 4, out.ctypes.data_as(PF), C.byref(ms)))
        print(f"{M}x{N}x{K} form {form}: {2.0*M*N*K/ms.value/1e9:6.0f} TFLOP/s ({ms.value*1e3:.0f} us)", flush=True)
eng.close()
