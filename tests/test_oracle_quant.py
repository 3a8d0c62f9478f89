"""MLX affine quantisation restatement (oracle/quant.py): packing order, dequantisation, the two quantizedMatmul paths,
and agreement with the product-side synthetic quantiser (qasr.synth).  CPU only.

What pins what: the reference holds no quantised tensor, so the LSB-first order and the group
layout below are the published mlx convention restated -- `tests/golden/kat_mlx_quant.json` freezes that convention as
explicit numbers so that the loader, the oracle and the HIP kernels cannot drift apart silently; the only external pin
is the real-checkpoint transcript snapshot (tests/test_gpu_e2e_snapshot.py)."""
import json
import os
import numpy as np
import pytest
import torch
from conftest import GOLDEN
from oracle import quant as Q
from qasr import synth

KAT = json.load(open(os.path.join(GOLDEN, "kat_mlx_quant.json")))


@pytest.mark.parametrize("case", KAT["pack"], ids=lambda c: c["name"])
def test_pack_order_kat(case):
    q = np.array(case["q"], dtype=np.uint32)
    words = np.array(case["words"], dtype=np.uint32)
    assert np.array_equal(Q.pack(q, case["bits"]), words)
    assert np.array_equal(Q.unpack(words, case["bits"]), q)


@pytest.mark.parametrize("case", KAT["dequant"], ids=lambda c: c["name"])
def test_dequant_kat(case):
    words = np.array(case["words"], dtype=np.uint32)[None, :]
    s = np.array([case["scales"]], dtype=np.float32)
    b = np.array([case["biases"]], dtype=np.float32)
    w = Q.dequantize_f32(words, s, b, case["bits"]).numpy()[0]
    assert np.array_equal(w[case["probe"]], np.array(case["expect_f32"], dtype=np.float32))
    wb = Q.dequantized(words, s, b, case["bits"]).numpy()[0]
    assert np.array_equal(wb[case["probe"]], np.array(case["expect_bf16"], dtype=np.float32))


@pytest.mark.parametrize("bits", [4, 8])
def test_roundtrip_and_quantizer_agreement(bits):
    g = torch.Generator().manual_seed(bits)
    w = (torch.randn(48, 256, generator=g) * 0.05).to(torch.bfloat16)
    wq, s, b = Q.quantize(w, bits)
    assert wq.dtype == np.uint32 and wq.shape == (48, 256 * bits // 32) and s.shape == (48, 4)
    w_hat = Q.dequantize_f32(wq, s, b, bits)
    step = np.abs(s).max()
    assert float((w_hat - w.float()).abs().max()) <= 1.5 * step               # about a quantisation step (scale and bias are themselves rounded to bf16)
    # the product-side synthetic quantiser packs the same bits
    pw, ps, pb = synth.quantize_linear(w, bits)
    assert np.array_equal(pw.numpy().view(np.uint32), wq)
    assert np.array_equal(ps.float().numpy(), s) and np.array_equal(pb.float().numpy(), b)


@pytest.mark.parametrize("bits", [4, 8])
def test_qmv_factored_identity(bits):
    """sum_g scale_g (sum q x) + bias_g (sum x)  ==  x . (scale q + bias): the form the HIP kernels compute."""
    g = torch.Generator().manual_seed(10 + bits)
    w = (torch.randn(32, 192, generator=g) * 0.1).to(torch.bfloat16)
    wq, s, b = Q.quantize(w, bits)
    x = (torch.randn(3, 192, generator=g)).to(torch.bfloat16).float()
    a = Q.qmv_factored(x.numpy(), wq, s, b, bits)
    ref = (x.double() @ Q.dequantize_f32(wq, s, b, bits).double().T).numpy()
    assert np.allclose(a, ref, rtol=0, atol=1e-9)


def test_matmul_paths_differ_only_by_weight_rounding():
    g = torch.Generator().manual_seed(3)
    w = (torch.randn(64, 128, generator=g) * 0.1).to(torch.bfloat16)
    wq, s, b = Q.quantize(w, 4)
    x1 = torch.randn(1, 128, generator=g).to(torch.bfloat16).float()
    xm = torch.randn(Q.QMV_MAX_ROWS + 1, 128, generator=g).to(torch.bfloat16).float()
    exact = Q.dequantize_f32(wq, s, b, 4)
    rounded = Q.dequantized(wq, s, b, 4)
    assert torch.equal(Q.quantized_matmul(x1, wq, s, b, 4), x1 @ exact.T)
    assert torch.equal(Q.quantized_matmul(xm, wq, s, b, 4), xm @ rounded.T)
    assert not torch.equal(exact, rounded)                                       # 4-bit q x bf16 scale + bf16 bias needs > 8 bits


def test_quantised_decoder_oracle_runs_and_tracks_the_float_one():
    """Tiny geometry, 8-bit: the quantised oracle decoder stays close to the float one it was quantised from (sanity of
    the wiring: embedding gather, per-linear paths, tied head)."""
    from oracle import config as C, decoder, precision as P
    sd = synth.synth_state_dict(C.AUDIO_TINY, C.TEXT_TINY, seed=3, init="stress")
    qsd = synth.quantize_state_dict(sd, 8)
    assert qsd["model.embed_tokens.weight"].dtype == torch.int32 and "model.layers.0.mlp.up_proj.scales" in qsd
    emb = P.bf16_round(torch.randn(40, C.TEXT_TINY.hidden, generator=torch.Generator().manual_seed(0)) * 0.5)
    with torch.no_grad():
        lf, _, _ = decoder.prefill(emb, decoder.Weights(sd), C.TEXT_TINY, P.REFERENCE, C.TOKENS_TINY)
        lq, st, _ = decoder.prefill(emb, decoder.Weights(qsd), C.TEXT_TINY, P.REFERENCE, C.TOKENS_TINY)
        l2 = decoder.decode_step(int(lq.argmax()), decoder.Weights(qsd), C.TEXT_TINY, st, P.REFERENCE)
    assert lq.shape == lf.shape and torch.isfinite(l2).all()
    assert float(torch.linalg.norm(lq - lf) / torch.linalg.norm(lf)) < 0.15
