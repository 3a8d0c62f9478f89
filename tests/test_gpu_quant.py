"""MLX 4 / 8-bit quantised text decoder on the device vs the CPU oracle, via the C ABI.

The oracle (oracle/quant.py, oracle/decoder.py: Weights.linear / embed_rows) restates what the reference's backend does
with a `QuantizedLinear` / `PreQuantizedEmbedding`: embedding rows and the prompt pass use bf16(scale * q + bias), a decode
step (one row of x per sequence) multiplies by scale * q + bias in f32 without rounding it.  The device keeps the packed
words in HBM and computes the decode-step form sum_g scale_g (sum q x) + bias_g (sum x) on the matrix cores
(csrc/dec_quant.hip); the prompt pass runs its bf16 GEMMs on bf16(scale * q + bias) copies.

Tolerances: those of the float decoder tests -- logits are bf16 values, max |d| within 6 bf16 ulps of the largest |logit|
(0.06 on the tiny geometry) and relative L2 < 3e-2 (1.5e-2 tiny), tokens teacher-forced within the same margin.
"""
import dataclasses
import numpy as np
import pytest
import torch
from oracle import config as C, decoder, pipeline, precision as P
from qasr import synth
import gpu_util

pytestmark = pytest.mark.gpu
A, T, TOK = C.AUDIO_TINY, C.TEXT_TINY, C.TOKENS_TINY


def _ulp_tol(ref, ulps=6.0):
    m = float(np.abs(ref).max())
    return ulps * 2.0 ** (np.floor(np.log2(max(m, 1e-3))) - 7)


def _check(got, ref, rel_bar, what):
    d, rel = np.abs(got - ref).max(), np.linalg.norm(got - ref) / np.linalg.norm(ref)
    print(f"{what}: max|d| {d:.4f} ({d / _ulp_tol(ref, 1.0):.1f} ulps) rel-L2 {rel:.2e}")
    assert d <= _ulp_tol(ref) and rel < rel_bar, (what, d, rel)
    assert ref[int(got.argmax())] >= ref.max() - _ulp_tol(ref)


@pytest.mark.parametrize("bits,sb_f32", [(4, False), (8, False), (4, True)], ids=["w4", "w8", "w4-f32scales"])
def test_tiny_geometry_generic_kernels(bits, sb_f32):
    """hidden 64 / inter 128: no tuned instantiation exists, every quantised product takes the generic kernel and the
    generic LM head (one argmax partial per row)."""
    sd = synth.synth_state_dict(A, T, seed=3, init="stress")
    qsd = synth.quantize_state_dict(sd, bits)
    if sb_f32:
        qsd = {k: (v.to(torch.float32) if k.endswith((".scales", ".biases")) else v) for k, v in qsd.items()}
    eng = gpu_util.Engine("tiny", max_audio_seconds=10, max_new_tokens=24, bits=bits)
    try:
        eng.load_state_dict(qsd)
        W = decoder.Weights(qsd)
        emb = P.bf16_round(torch.randn(33, T.hidden, generator=torch.Generator().manual_seed(1)) * 0.5)
        with torch.no_grad():
            toks, logits = decoder.greedy(emb, W, T, P.DEVICE, TOK, max_tokens=10, ignore_eos=True, return_logits=True)
        _check(eng.prefill_logits(emb.numpy()), logits[0].numpy(), 1.5e-2, "prompt pass")
        got = eng.decode_forced(toks[:-1])
        for i in range(len(toks) - 1):
            _check(got[i], logits[i + 1].numpy(), 1.5e-2, f"step {i}")
        # whole path, batch of ragged clips, teacher-forced; each clip alone gives the same tokens
        model = pipeline.OracleModel(qsd, A, T, TOK, P.DEVICE)
        clips = [synth.synth_waveform(0, 2.5), synth.synth_waveform(1, 1.0), synth.synth_waveform(2, 0.4)]
        out = eng.transcribe_batch(clips, max_tokens=8, ignore_eos=True)
        with torch.no_grad():
            for pcm, tk in zip(clips, out):
                e = model.encode(model.mel(pcm))
                lg, st, _ = decoder.prefill(e, model.W, T, P.DEVICE, TOK)
                for i, t in enumerate(tk):
                    assert lg[t] >= lg.max() - 0.06, (i, t)
                    if i + 1 < len(tk):
                        lg = decoder.decode_step(t, model.W, T, st, P.DEVICE)
        for pcm, tk in zip(clips, out):
            assert eng.transcribe_batch([pcm], max_tokens=8, ignore_eos=True)[0] == tk
    finally:
        eng.close()


@pytest.fixture(scope="module")
def full_sd():
    return synth.synth_state_dict(C.AUDIO_SMALL, dataclasses.replace(C.TEXT_SMALL, layers=3), seed=0, init="stress")


@pytest.mark.parametrize("bits,sb_f32", [(4, False), (8, False), (8, True)], ids=["w4", "w8", "w8-f32scales"])
def test_full_width_tuned_kernels(full_sd, bits, sb_f32):
    """Qwen3-ASR-0.6B widths (hidden 1024, inter 3072, 16/8 heads x 128, vocab 151 936) with 3 decoder layers so the CPU
    oracle stays affordable: the K = 1024 / 2048 / 3072 tuned quantised GEMVs, the persistent quantised LM head, the
    quantised embedding gather, at 1 and 32 batch rows."""
    t = dataclasses.replace(C.TEXT_SMALL, layers=3)
    qsd = synth.quantize_state_dict(full_sd, bits)
    if sb_f32:
        qsd = {k: (v.to(torch.float32) if k.endswith((".scales", ".biases")) else v) for k, v in qsd.items()}
    eng = gpu_util.Engine("0.6B", max_batch=32, max_audio_seconds=6, max_new_tokens=16, dec_layers=3, bits=bits)
    try:
        eng.load_state_dict(qsd)
        W = decoder.Weights(qsd)
        emb = P.bf16_round(torch.randn(65, t.hidden, generator=torch.Generator().manual_seed(2)) * 0.5)
        with torch.no_grad():
            toks, logits = decoder.greedy(emb, W, t, P.REFERENCE, C.TOKENS, max_tokens=7, ignore_eos=True, return_logits=True)
        _check(eng.prefill_logits(emb.numpy()), logits[0].numpy(), 3e-2, "prompt pass T=81")
        got = eng.decode_forced(toks[:-1])
        for i in range(len(toks) - 1):
            _check(got[i], logits[i + 1].numpy(), 3e-2, f"step {i}")
        # 32 rows (two batch tiles) == every clip alone, deterministic
        clips = [synth.synth_waveform(k, 1.0 + 0.11 * (k % 5)) for k in range(32)]
        a = eng.transcribe_batch(clips, max_tokens=5, ignore_eos=True)
        assert eng.transcribe_batch(clips, max_tokens=5, ignore_eos=True) == a
        for k in (0, 15, 16, 31):
            assert eng.transcribe_batch([clips[k]], max_tokens=5, ignore_eos=True)[0] == a[k]
        assert eng.transcribe_batch(clips[3:20], max_tokens=5, ignore_eos=True) == a[3:20]
    finally:
        eng.close()


@pytest.mark.parametrize("bits", [4, 8], ids=["w4", "w8"])
def test_1p7b_width_down_projection_in_two_column_phases(bits):
    """Qwen3-ASR-1.7B decoder widths (hidden 2048, inter 6144) with 2 layers: the K = 6144 down-projection does not fit one LDS
    image of its 16 activation rows and takes the two-phase form of decode_gemvq_kernel (KPH = 2); against the quantised oracle,
    against the generic kernel it replaced (`gemv_wide = 0`), at 1 and 18 batch rows (two row groups, the second nearly empty)."""
    t = dataclasses.replace(C.TEXT_LARGE, layers=2)
    sd = synth.synth_state_dict(dataclasses.replace(C.AUDIO_LARGE, layers=1), t, seed=1, init="stress")
    qsd = synth.quantize_state_dict(sd, bits)
    eng = gpu_util.Engine("1.7B", max_batch=18, max_audio_seconds=3, max_new_tokens=16, enc_layers=1, dec_layers=2, bits=bits)
    try:
        eng.load_state_dict(qsd)
        W = decoder.Weights(qsd)
        emb = P.bf16_round(torch.randn(40, t.hidden, generator=torch.Generator().manual_seed(4)) * 0.5)
        with torch.no_grad():
            toks, logits = decoder.greedy(emb, W, t, P.REFERENCE, C.TOKENS, max_tokens=6, ignore_eos=True, return_logits=True)
        _check(eng.prefill_logits(emb.numpy()), logits[0].numpy(), 3e-2, "prompt pass")
        got = eng.decode_forced(toks[:-1])
        for i in range(len(toks) - 1):
            _check(got[i], logits[i + 1].numpy(), 3e-2, f"step {i} (two-phase kernel)")
        eng.set_tuning("gemv_wide", 0)
        try:
            eng.prefill_logits(emb.numpy())
            gen = eng.decode_forced(toks[:-1])
        finally:
            eng.set_tuning("gemv_wide", 1)
        for i in range(len(toks) - 1):
            _check(gen[i], logits[i + 1].numpy(), 3e-2, f"step {i} (generic kernel)")
            assert np.abs(got[i] - gen[i]).max() <= _ulp_tol(gen[i])
        clips = [synth.synth_waveform(k, 0.6 + 0.07 * (k % 4)) for k in range(18)]
        a = eng.transcribe_batch(clips, max_tokens=4, ignore_eos=True)
        for k in (0, 15, 16, 17):
            assert eng.transcribe_batch([clips[k]], max_tokens=4, ignore_eos=True)[0] == a[k], k
    finally:
        eng.close()


@pytest.mark.parametrize("bits,sb_f32", [(4, False), (8, True)], ids=["w4", "w8-f32scales"])
def test_lm_head_weight_rings_agree_bit_for_bit(full_sd, bits, sb_f32):
    """The quantised LM head streams its weights either through a four-block register ring or through a wave-private LDS
    ring fed by direct-to-LDS loads (9-13 KiB in flight per wave); both consume the blocks in the same order with the same
    arithmetic: same logits bits at 1 row, same tokens at 32 rows (two batch tiles, shallower ring)."""
    qsd = synth.quantize_state_dict(full_sd, bits)
    if sb_f32:
        qsd = {k: (v.to(torch.float32) if k.endswith((".scales", ".biases")) else v) for k, v in qsd.items()}
    eng = gpu_util.Engine("0.6B", max_batch=32, max_audio_seconds=6, max_new_tokens=16, dec_layers=3, bits=bits)
    try:
        eng.load_state_dict(qsd)
        emb = P.bf16_round(torch.randn(33, 1024, generator=torch.Generator().manual_seed(4)) * 0.5).numpy()
        clips = [synth.synth_waveform(k, 1.0 + 0.13 * (k % 4)) for k in range(32)]
        res = []
        for ring in (0, 1):
            eng.set_tuning("lmh_q_ring", ring)
            first = eng.prefill_logits(emb)
            steps = eng.decode_forced([int(first.argmax()), 17, 151643, 5])
            res.append((first, steps, eng.transcribe_batch(clips, max_tokens=6, ignore_eos=True)))
        assert np.array_equal(res[0][0], res[1][0])
        assert all(np.array_equal(a, b) for a, b in zip(res[0][1], res[1][1]))
        assert res[0][2] == res[1][2]
    finally:
        eng.set_tuning("lmh_q_ring", 1)
        eng.close()


def test_quantised_differs_from_its_bf16_expansion(full_sd):
    """Why the packed kernels exist: a decode step on bf16(scale * q + bias) weights (what round 1 did at load) is NOT
    what the reference computes.  On the same 4-bit triplets the two give measurably different logits; the device
    follows the f32-dequantised form to within the usual tolerance while the bf16 expansion sits further away."""
    t = dataclasses.replace(C.TEXT_SMALL, layers=3)
    qsd = synth.quantize_state_dict(full_sd, 4)
    W = decoder.Weights(qsd)
    fsd = dict(full_sd)
    for k in list(qsd):
        if k.endswith(".scales"):
            stem = k[:-len(".scales")]
            fsd[stem + ".weight"] = W._dequant(stem, rounded=True).to(torch.bfloat16)
    emb = P.bf16_round(torch.randn(40, t.hidden, generator=torch.Generator().manual_seed(5)) * 0.5)
    with torch.no_grad():
        lq, sq, _ = decoder.prefill(emb, W, t, P.REFERENCE, C.TOKENS)
        tok = int(lq.argmax())
        step_q = decoder.decode_step(tok, W, t, sq, P.REFERENCE).numpy()
        lf, sf, _ = decoder.prefill(emb, decoder.Weights(fsd), t, P.REFERENCE, C.TOKENS)
        step_f = decoder.decode_step(tok, decoder.Weights(fsd), t, sf, P.REFERENCE).numpy()
    gap = np.linalg.norm(step_q - step_f) / np.linalg.norm(step_q)
    eng = gpu_util.Engine("0.6B", max_batch=1, max_audio_seconds=6, max_new_tokens=16, dec_layers=3, bits=4)
    try:
        eng.load_state_dict(qsd)
        eng.prefill_logits(emb.numpy())
        got = eng.decode_forced([tok])[0]
    finally:
        eng.close()
    d_q = np.linalg.norm(got - step_q) / np.linalg.norm(step_q)
    d_f = np.linalg.norm(got - step_f) / np.linalg.norm(step_f)
    print(f"oracle exact-vs-bf16-expansion gap {gap:.2e}; device vs exact {d_q:.2e}, device vs expansion {d_f:.2e}")
    assert d_q < 3e-2


def test_quantised_engine_device_bytes(full_sd):
    """What the device actually gives up for a 4-bit engine (hipMemGetInfo delta around create + load + finalize, capacity of one 6 s
    clip so that caches / workspaces are small): the reported footprint is a lower bound of it within the workspace size, and it holds
    no bf16 expansion of the decoder -- packed tensors + packed decode images + ONE layer of bf16 scratch (the prompt pass dequantises
    layer by layer), where bf16 copies of every decoder Linear would add 2 bytes per decoder parameter."""
    qsd = synth.quantize_state_dict(full_sd, 4)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    eng = gpu_util.Engine("0.6B", max_batch=1, max_audio_seconds=6, max_new_tokens=16, dec_layers=3, bits=4)
    try:
        eng.load_state_dict(qsd)
        torch.cuda.synchronize()
        free1, _ = torch.cuda.mem_get_info()
        used = free0 - free1
        fp = int(eng.lib.qasr_memory_footprint(eng.h))
        uploaded = sum(t.numel() * t.element_size() for t in qsd.values())
        dec_lin = [k for k in qsd if k.startswith("model.layers.") and k.endswith("_proj.weight")]
        dec_params = sum(qsd[k].numel() * 8 for k in dec_lin)                       # 8 four-bit values per uint32 word
        layer_bf16 = 2 * dec_params // 3
        print(f"device bytes {used / 1e6:.1f} MB, footprint {fp / 1e6:.1f} MB, uploaded {uploaded / 1e6:.1f} MB, one bf16 layer {layer_bf16 / 1e6:.1f} MB")
        assert uploaded < fp <= used + (2 << 20)                                   # allocator granularity
        assert used - fp < 400e6                                                   # caches + activation workspaces at this capacity
        # resident decoder weight bytes beyond the upload: decode images (~ the packed size again) + one layer of scratch, far below
        # a bf16 copy of all three layers
        dec_packed = sum(qsd[k].numel() * qsd[k].element_size() for k in qsd if k.startswith("model.layers.") and "_proj." in k)
        enc_up = sum(t.numel() * t.element_size() for k, t in qsd.items() if k.startswith("audio_tower."))
        assert fp - uploaded < enc_up + 1.25 * (dec_packed + qsd["model.embed_tokens.weight"].numel() * 4 * 1.2) + layer_bf16 + (8 << 20)
        assert fp - uploaded < enc_up + 2 * dec_params * 0.75 + qsd["model.embed_tokens.weight"].numel() * 4 * 1.5
    finally:
        eng.close()
