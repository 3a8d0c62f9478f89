import ctypes as C, os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "qwen3-asr-swift_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gpu_util
eng = gpu_util.Engine("tiny", max_audio_seconds=2)
P16, PF = C.POINTER(C.c_uint16), C.POINTER(C.c_float)
for M, N, K in ((47968, 3072, 1024), (47968, 1024, 4096), (47968, 8192, 2048)):
    rng = np.random.default_rng(1)
    A = torch.from_numpy(rng.standard_normal((M, K), dtype=np.float32)).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    W = torch.from_numpy(rng.standard_normal((N, K), dtype=np.float32)).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    out = np.empty((M, N), np.float32)
    ref = None
    line = []
    for form in (10, 11, 12, 13, 10):
        ms = C.c_float()
        eng.check(eng.lib.qasr_gemm_probe(eng.h, A.ctypes.data_as(P16), W.ctypes.data_as(P16), None, M, N, K, form, 5, out.ctypes.data_as(PF), C.byref(ms)))
        if ref is None: ref = out.copy()
        line.append(f"vmcnt({8 - 2 * (form - 10)}):{2.0 * M * N * K / ms.value / 1e9:5.0f}{'' if np.array_equal(out, ref) else '!'}")
    print(f"{M}x{N}x{K}: " + "  ".join(line), flush=True)
eng.close()
