"""The MFMA GEMM (csrc/gemm.h, csrc/gemm_p8.h) by itself, through qasr_gemm_probe: C = A . W^T + bias with bf16 operands and
f32 accumulation.  No reference counterpart (MLX supplies the matmul there); the check is arithmetic:

  * every form (128x128 double-buffered | 128x128 single buffer | 256x256 persistent ping-pong) against a float64 product of
    the same bf16 values: |d| <= 2e-6 * sum_k |a_k w_k| + 1e-30 per element (f32 accumulation over K <= 8192 in MFMA order;
    the bound is ~20x the observed error and ~1000x below one bf16 ulp of the result);
  * the three forms against each other BIT FOR BIT: they sum every output in the same k order;
  * shapes: ragged M / N edges (1 row, 1 tile + 1, N = 4, N not a multiple of 16 / 128 / 256), K that is not a multiple of
    the 64-wide K-tile (8, 72, 1000), K-tile counts that are odd (the ring parity of gemm_p8), launches of fewer than 8
    workgroups and of several rounds of persistent tiles, and the benchmark's own shapes.
"""
import ctypes as C
import numpy as np
import pytest
import torch
import gpu_util

pytestmark = pytest.mark.gpu

SHAPES = [
    (1, 4, 8), (1, 16, 64), (17, 20, 72), (129, 260, 1000), (256, 256, 64), (257, 512, 128), (300, 1028, 192),
    (511, 100, 4320), (1000, 480, 4320), (549, 10288, 1024), (2049, 1024, 4096), (4097, 3072, 1024), (6000, 896, 3584),
    (12992, 1024, 2048),
]


@pytest.fixture(scope="module")
def eng():
    e = gpu_util.Engine("tiny", max_audio_seconds=2)
    yield e
    e.close()


def _bf16(x):
    t = torch.from_numpy(x).to(torch.bfloat16)
    return t.view(torch.int16).numpy().view(np.uint16), t.to(torch.float64).numpy()


def _run(eng, A16, W16, bias, M, N, K, form, reps=1):
    out = np.empty((M, N), np.float32)
    ms = C.c_float()
    rc = eng.lib.qasr_gemm_probe(eng.h, A16.ctypes.data_as(C.POINTER(C.c_uint16)), W16.ctypes.data_as(C.POINTER(C.c_uint16)),
                                 bias.ctypes.data_as(C.POINTER(C.c_float)) if bias is not None else None, M, N, K, form, reps,
                                 out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(ms))
    eng.check(rc)
    return out, ms.value


@pytest.mark.parametrize("M,N,K", SHAPES, ids=[f"{m}x{n}x{k}" for m, n, k in SHAPES])
def test_forms_vs_float64_and_each_other(eng, M, N, K):
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    A16, A = _bf16(rng.standard_normal((M, K), dtype=np.float32))
    W16, W = _bf16(rng.standard_normal((N, K), dtype=np.float32) / np.sqrt(K).astype(np.float32))
    bias = rng.standard_normal(N).astype(np.float32)
    want = A @ W.T + bias.astype(np.float64)
    mag = np.abs(A) @ np.abs(W).T + np.abs(bias)
    outs = [_run(eng, A16, W16, bias, M, N, K, form)[0] for form in (0, 1, 2, -1)]
    for form, got in zip((0, 1, 2, -1), outs):
        err = np.abs(got - want)
        worst = float((err / (mag + 1e-30)).max())
        assert np.isfinite(got).all() and worst <= 2e-6, (form, worst)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2]) and np.array_equal(outs[0], outs[3])
    nobias, _ = _run(eng, A16, W16, None, M, N, K, 2)
    assert np.abs(nobias - (want - bias)).max() <= 2e-6 * mag.max()


@pytest.mark.parametrize("M,N,K", [(4097, 3072, 1024), (2049, 1024, 4096), (12992, 6144, 1024)])
def test_p8_is_repeatable(eng, M, N, K):
    """Race screen for the ping-pong schedule of gemm_p8.h (LDS units restaged while the other wave group still computes,
    counted vmcnt across raw barriers): 25 launches of several tiles per workgroup must give the same bits every time and
    the bits of the single-buffer 128 x 128 form (a race would show as a wrong tile that comes and goes)."""
    rng = np.random.default_rng(K + N)
    A16, _ = _bf16(rng.standard_normal((M, K), dtype=np.float32))
    W16, _ = _bf16(rng.standard_normal((N, K), dtype=np.float32))
    ref, _ = _run(eng, A16, W16, None, M, N, K, 1)
    for i in range(25):
        got, _ = _run(eng, A16, W16, None, M, N, K, 2, reps=2)
        assert np.array_equal(got, ref), i


def test_refused_shapes(eng):
    z = np.zeros((8, 8), np.uint16)
    o = np.zeros((8, 8), np.float32)
    f = lambda m, n, k, form: eng.lib.qasr_gemm_probe(eng.h, z.ctypes.data_as(C.POINTER(C.c_uint16)), z.ctypes.data_as(C.POINTER(C.c_uint16)),
                                                      None, m, n, k, form, 1, o.ctypes.data_as(C.POINTER(C.c_float)), None)
    assert f(8, 8, 8, 2) == 0
    assert f(8, 8, 4, 2) != 0 and f(8, 6, 8, 2) != 0 and f(0, 8, 8, 2) != 0 and f(8, 8, 8, 3) != 0


def test_throughput_report(eng):
    """not an assertion on speed: prints TFLOP/s of the three forms on two benchmark shapes for the round's notes"""
    for M, N, K in ((12992, 6144, 1024), (12992, 4096, 1024), (12992, 1024, 2048), (12992, 1024, 3072), (12480, 2688, 896),
                    (12480, 896, 3584), (12480, 3584, 896), (47968, 3072, 1024), (47968, 1024, 4096)):
        rng = np.random.default_rng(1)
        A16, _ = _bf16(rng.standard_normal((M, K), dtype=np.float32))
        W16, _ = _bf16(rng.standard_normal((N, K), dtype=np.float32))
        line = []
        for form in (0, 1, 2):
            _, ms = _run(eng, A16, W16, None, M, N, K, form, reps=5)
            line.append(f"form {form}: {2.0 * M * N * K / ms / 1e9:7.0f} TFLOP/s")
        print(f"{M} x {N} x {K}: " + ", ".join(line))
