"""MLX affine quantisation (group 64, 4 / 8 bit): CPU restatement of what the reference calls into.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Call sites in the reference (the arithmetic itself lives in mlx-swift >= 0.30.0, `Package.swift:114`, third-party,
un-vendored, exact version unpinned -- `Package.resolved` is git-ignored):
  * Sources/MLXCommon/PreQuantizedEmbedding.swift:35-42  `dequantized(weight[x], scales:, biases:, groupSize:, bits:)`
  * Sources/MLXCommon/PreQuantizedEmbedding.swift:45-49  `quantizedMatmul(x, weight, scales:, biases:, transpose: true, ...)`
  * Sources/Qwen3ASR/QuantizedTextDecoder.swift:33-44,111-126  `QuantizedLinear` (MLXNN) -> the same `quantizedMatmul`
  * checkpoint triplets `X.weight` uint32 [out, in * bits / 32], `X.scales`, `X.biases` [out, in / 64]
    (Sources/MLXCommon/WeightLoading.swift:48-96)

Published algorithm restated here (mlx `ops.cpp: quantize / dequantize`, `backend/metal/kernels/quantized.h`):
  * storage: element i of a row lives in uint32 word i // (32 / bits), bits [bits * (i % (32 / bits)), +bits) -- LSB first;
  * w_hat[i] = scales[i // 64] * q[i] + biases[i // 64];
  * `dequantized` returns w_hat in the dtype of `scales` (bf16 in the shipped checkpoints: placeholders at
    PreQuantizedEmbedding.swift:27-29): ONE rounding of the f32 product-sum;
  * `quantizedMatmul` with few rows of x (decode, M = 1) runs the `qmv` kernel: per output y = sum over groups of
    (scale * sum_k q_k x_k + bias * sum_k x_k), everything in f32, one rounding of y to the activation dtype -- the
    dequantised weight is never rounded;
  * with many rows (the prompt pass, M = 406) it runs `qmm_t`, whose block loader writes w_hat into threadgroup memory
    IN THE ACTIVATION DTYPE (bf16) before the simdgroup matmul: the effective weight is bf16(w_hat).
    Which of the two runs is decided by a device-dependent row threshold (6 .. 32 rows); the reference only ever
    multiplies 1 row (decode) or the whole prompt, so `rows == 1` / `rows > 32` is the rule used here.
None of this is observable in the reference's tree and no reference test holds a quantised tensor; the single pin is the
end-to-end transcript snapshot (tests/test_gpu_e2e_snapshot.py), which needs the real checkpoint.  "parity unpinned" for
the packing order and the qmm rounding until then.
"""
import numpy as np
import torch

GROUP = 64
QMV_MAX_ROWS = 32          # above this many rows of x the reference's backend uses the qmm_t path (see module docstring)


def pack(q, bits):
    """q: integer array [..., n] with values in [0, 2^bits) -> uint32 [..., n * bits / 32], LSB first."""
    q = np.asarray(q).astype(np.uint32)
    per = 32 // bits
    assert q.shape[-1] % per == 0 and int(q.max(initial=0)) < (1 << bits)
    q = q.reshape(*q.shape[:-1], q.shape[-1] // per, per)
    shifts = (np.arange(per, dtype=np.uint32) * np.uint32(bits))
    return np.bitwise_or.reduce(q << shifts, axis=-1).astype(np.uint32)


def unpack(w, bits):
    """uint32 [..., m] -> uint32 [..., m * 32 / bits] of values in [0, 2^bits)."""
    w = np.asarray(w, dtype=np.uint32)
    per = 32 // bits
    shifts = (np.arange(per, dtype=np.uint32) * np.uint32(bits))
    q = (w[..., None] >> shifts) & np.uint32((1 << bits) - 1)
    return q.reshape(*w.shape[:-1], w.shape[-1] * per)


def quantize(w, bits, group=GROUP):
    """mlx `quantize` (affine, per group of `group` input elements of every row): -> (packed uint32, scales, biases),
    scales / biases float32 arrays [out, in / group] holding values representable in w's dtype (bf16).  Only used to build
    synthetic quantised checkpoints; parity never depends on how q, scale, bias were chosen."""
    w = torch.as_tensor(w)
    dt = w.dtype if w.dtype in (torch.bfloat16, torch.float16) else torch.float32
    x = w.to(torch.float32)
    out, n = x.shape
    assert n % group == 0
    g = x.reshape(out, n // group, group)
    n_bins = float((1 << bits) - 1)
    w_max, w_min = g.max(dim=-1).values, g.min(dim=-1).values
    mask = w_min.abs() > w_max.abs()
    scales = torch.clamp((w_max - w_min) / n_bins, min=1e-7)
    scales = torch.where(mask, scales, -scales)
    edge = torch.where(mask, w_min, w_max)
    q0 = torch.round(edge / scales)
    scales = torch.where(q0 != 0, edge / q0, scales)
    biases = torch.where(q0 == 0, torch.zeros_like(edge), edge)
    scales, biases = scales.to(dt).to(torch.float32), biases.to(dt).to(torch.float32)      # stored in the model dtype
    q = torch.clamp(torch.round((g - biases[..., None]) / scales[..., None]), 0, n_bins)
    return pack(q.reshape(out, n).numpy().astype(np.uint32), bits), scales.numpy(), biases.numpy()


def dequantize_f32(wq, scales, biases, bits, group=GROUP):
    """w_hat = scale * q + bias in float32 (no rounding of the result) -> torch [out, in]."""
    q = torch.from_numpy(unpack(wq, bits).astype(np.float32))
    s = torch.as_tensor(np.asarray(scales, dtype=np.float32)).repeat_interleave(group, dim=-1)
    b = torch.as_tensor(np.asarray(biases, dtype=np.float32)).repeat_interleave(group, dim=-1)
    return s * q + b


def dequantized(wq, scales, biases, bits, group=GROUP, dtype=torch.bfloat16):
    """`dequantized(...)`: w_hat rounded once to the dtype of the scales."""
    return dequantize_f32(wq, scales, biases, bits, group).to(dtype).to(torch.float32)


def quantized_matmul(x, wq, scales, biases, bits, group=GROUP, act_dtype=torch.bfloat16):
    """`quantizedMatmul(x, w, transpose: true)`: x [rows, in] float32 values of the activation dtype -> [rows, out] f32,
    NOT yet rounded to the activation dtype (the caller's rounding policy does that).  rows <= QMV_MAX_ROWS: exact f32
    dequantisation inside the dot product (qmv); more rows: weights rounded to the activation dtype first (qmm_t)."""
    x = torch.as_tensor(x, dtype=torch.float32)
    w = dequantize_f32(wq, scales, biases, bits, group)
    if x.shape[0] > QMV_MAX_ROWS:
        w = w.to(act_dtype).to(torch.float32)
    return x @ w.T


def qmv_factored(x, wq, scales, biases, bits, group=GROUP):
    """The qmv form spelled out (float64 accumulation): y = sum_g scale_g * (sum_k q_k x_k) + bias_g * (sum_k x_k).
    Used by the tests to show that it equals x @ dequantize_f32(...)^T -- the identity the HIP kernels rely on."""
    x = np.asarray(x, dtype=np.float64)
    q = unpack(wq, bits).astype(np.float64)
    out, n = q.shape
    G = n // group
    xs = x.reshape(x.shape[0], G, group)
    qg = q.reshape(out, G, group)
    dot = np.einsum("rgk,ogk->rog", xs, qg)
    xsum = xs.sum(axis=-1)
    return (dot * np.asarray(scales, dtype=np.float64)[None] + xsum[:, None, :] * np.asarray(biases, dtype=np.float64)[None]).sum(axis=-1)
