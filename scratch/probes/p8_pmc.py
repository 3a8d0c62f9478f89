import ctypes as C, os, sys
ROOT = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "qwen3-asr-swift_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gpu_util
eng = gpu_util.Engine("tiny", max_audio_seconds=2)
P16, PF = C.POINTER(C.c_uint16), C.POINTER(C.c_float)
M, N, K = 47968, 1024, 4096
rng = np.random.default_rng(1)
A = torch.from_numpy(rng.standard_normal((M, K), dtype=np.float32)).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
W = torch.from_numpy(rng.standard_normal((N, K), dtype=np.float32)).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
out = np.empty((M, N), np.float32)
ms = C.c_float()
eng.check(eng.lib.qasr_gemm_probe(eng.h, A.ctypes.data_as(P16), W.ctypes.data_as(P16), None, M, N, K, 2, 2, out.ctypes.data_as(PF), C.byref(ms)))
print("ms", ms.value)
eng.close()
