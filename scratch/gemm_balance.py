"""Tile-count balance of the 128x128 GEMM form on N = 1024 shapes (o-proj K = 2048, down K = 3072 of the prompt pass): qasr_gemm_probe at the
production M (12992 -> 816 tiles = 3.19 per CU) against balanced M (8192 -> 512 = 2.0; 16384 -> 1024 = 4.0; 12288 -> 768 = 3.0)."""
import sys, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'qwen3-asr-swift_amd'); sys.path.insert(0, 'tests')
import numpy as np
import gpu_util
e = gpu_util.Engine("tiny", max_batch=1, max_audio_seconds=2, max_new_tokens=4)
rng = np.random.default_rng(0)
def bf16(a):
    return (a.astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
for K in (2048, 3072):
    W = bf16(rng.standard_normal((1024, K)) * 0.02)
    for M in (8192, 12288, 12992, 16384):
        A = bf16(rng.standard_normal((M, K)) * 0.5)
        out = np.empty((M, 1024), np.float32)
        for form in (1, 0, 2):
            ms = C.c_float()
            rc = e.lib.qasr_gemm_probe(e.h, A.ctypes.data_as(C.POINTER(C.c_uint16)), W.ctypes.data_as(C.POINTER(C.c_uint16)), None, M, 1024, K, form, 10,
                                       out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(ms))
            assert rc == 0
            tiles = -(-M // 128) * 8
            print(f"K={K} M={M:6d} form={form} tiles128={tiles:5d} ({tiles/256:.2f}/CU): {ms.value*1e3:7.1f} us  {2.0*M*1024*K/ms.value/1e9:6.0f} TFLOP/s", flush=True)
e.close()
