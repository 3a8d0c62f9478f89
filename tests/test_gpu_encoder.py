"""HIP audio encoder vs the CPU oracle (R4 of SURVEY.md section 8a), through the C ABI.

Two bars, both stated here:
  * vs oracle policy DEVICE (same bf16 rounding points as the MFMA path): relative L2 error < 1e-2.
    Measured: bit-identical for <= 2 tokens (T <= 16 frames), 4e-4 .. 5e-3 beyond.  Why not tighter:
    f32 sums are associated differently (MFMA k-order vs CPU BLAS), ~1e-7 relative; each such
    difference flips a later bf16 rounding with probability ~1e-4, a flip is 2^-8 relative, and
    flips cascade through the layers until the difference reaches the bf16 noise floor of the
    network -- the same ~4e-3 that separates DEVICE from REFERENCE.  For short inputs (where no
    flip happens) the match is exact, which is the structural check.
  * vs oracle policy REFERENCE (f32 encoder, what MLX does): relative L2 error < 2e-2 -- the measured
    cost of feeding MFMA bf16 operands (the one deliberate numerical deviation).
"""
import numpy as np
import pytest
import torch
from oracle import config as C, decoder, encoder, precision as P
from qasr import synth
import gpu_util

pytestmark = pytest.mark.gpu


def _check(got, sd, mel, cfg, exact=False):
    W = decoder.Weights(sd)
    with torch.no_grad():
        dev = P.bf16_round(encoder.encode(mel, W, cfg, P.DEVICE)).numpy()
        ref = encoder.encode(mel, W, cfg, P.REFERENCE).numpy()
    assert got.shape == dev.shape
    rel_dev = np.linalg.norm(got - dev) / np.linalg.norm(dev)
    rel_ref = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    bad = np.abs(got - dev) > (2.0 ** -7) * np.abs(dev) + 1e-3
    print(f"rel_dev={rel_dev:.2e} rel_ref={rel_ref:.2e} >2ulp={bad.mean():.3f}")
    assert rel_dev < 1e-2, rel_dev
    assert rel_ref < 2e-2, rel_ref
    if exact:
        assert rel_dev < 1e-3 and bad.mean() < 0.01, (rel_dev, bad.mean())
    return rel_dev, rel_ref


@pytest.fixture(scope="module")
def tiny():
    sd = synth.synth_state_dict(C.AUDIO_TINY, C.TEXT_TINY, seed=3, init="stress")
    e = gpu_util.Engine("tiny", max_audio_seconds=30)
    e.load_state_dict(sd)
    yield e, sd
    e.close()


@pytest.mark.parametrize("T", [1, 7, 13, 16, 50, 99, 100, 101, 250, 300, 530, 1030, 3000])
def test_encoder_tiny(tiny, T):
    eng, sd = tiny
    g = torch.Generator().manual_seed(T)
    mel = (torch.randn(128, T, generator=g) * 0.5).numpy()
    got = eng.encode(mel)
    assert got.shape[0] == encoder.get_output_length(T)
    _check(got, sd, mel, C.AUDIO_TINY, exact=T <= 16)


def test_encoder_full_size_5s(sd_small_stress):
    """Qwen3-ASR-0.6B geometry (d=896, 18 layers, 480 conv channels), 5 s clip = configs[0] shape."""
    from oracle import mel as omel
    a = C.AUDIO_SMALL
    full = sd_small_stress
    eng = gpu_util.Engine("0.6B", max_batch=1, max_audio_seconds=6)
    try:
        eng.load_state_dict(full)
        mel = omel.log_mel(synth.synth_waveform(0, 5.0))
        got = eng.encode(mel)
        assert got.shape == (65, 1024)
        _check(got, full, mel, a)
        # the implicit-GEMM convolutions decompose a K-tile's tap once per tile on the scalar unit (gemm.h AConv3x3s2W, default) or per
        # 16-byte chunk (AConv3x3s2): same addresses, same k order -> same bits
        eng.set_tuning("conv_ktile", 0)
        assert np.array_equal(eng.encode(mel), got)
    finally:
        eng.set_tuning("conv_ktile", 1)
        eng.close()


def _frac_over_2ulp(a, b):
    return float((np.abs(a - b) > (2.0 ** -7) * np.abs(b) + 1e-3).mean())


@pytest.mark.parametrize("depth", [1, 6, 18])
def test_encoder_flip_fraction_grows_like_a_reordered_cpu_sum(depth, sd_small_stress):
    """Distributional bound (replaces the dropped `bad.mean() < 0.05` check of round 1, which failed at 10-18 %).

    Claim under test: the device differs from the DEVICE-policy oracle only because f32 sums are associated differently,
    and each such difference flips a bf16 rounding now and then; flips then spread with depth.  If that is the mechanism,
    the CPU oracle evaluated in two summation orders (oracle.encoder.FLIP_K: every contraction over the reversed K axis)
    must show the same fraction of outputs off by more than 2 bf16 ulps as device-vs-oracle, at every depth.  So: full
    0.6B geometry, a 30 s clip, encoders truncated to 1 / 6 / 18 layers (qasr_config.enc_layers), and
        frac_gpu(depth) <= 2 * frac_cpu(depth) + 0.01
    Measured on the CPU (two orders): 0.163 / 0.192 / 0.219 of the outputs beyond 2 ulps at depth 1 / 6 / 18 with rel-L2
    4.2e-3 / 4.7e-3 / 4.9e-3 -- i.e. most of it is already there after the conv stem (K = 4320, 7680 contractions) and one
    layer, and depth adds little.  A kernel bug (wrong tap, wrong window, dropped bias) shows up in the rel-L2 half of the
    bound, which sits at ~1e-2."""
    import dataclasses
    from oracle import mel as omel
    a = dataclasses.replace(C.AUDIO_SMALL, layers=depth)
    sd = sd_small_stress
    eng = gpu_util.Engine("0.6B", max_batch=1, max_audio_seconds=30, enc_layers=depth)
    try:
        eng.load_state_dict(sd)
        mel = omel.log_mel(synth.synth_waveform(0, 30.0))
        got = eng.encode(mel)
        W = decoder.Weights(sd)
        with torch.no_grad():
            fwd = P.bf16_round(encoder.encode(mel, W, a, P.DEVICE)).numpy()
            encoder.FLIP_K = True
            try:
                rev = P.bf16_round(encoder.encode(mel, W, a, P.DEVICE)).numpy()
            finally:
                encoder.FLIP_K = False
        f_cpu, f_gpu = _frac_over_2ulp(rev, fwd), _frac_over_2ulp(got, fwd)
        rel_cpu = np.linalg.norm(rev - fwd) / np.linalg.norm(fwd)
        rel_gpu = np.linalg.norm(got - fwd) / np.linalg.norm(fwd)
        print(f"depth {depth}: >2ulp fraction cpu-reordered {f_cpu:.4f} device {f_gpu:.4f}; rel-L2 cpu {rel_cpu:.2e} device {rel_gpu:.2e}")
        assert f_gpu <= 2.0 * f_cpu + 0.01, (depth, f_cpu, f_gpu)
        assert rel_gpu <= 2.0 * rel_cpu + 1e-3, (depth, rel_cpu, rel_gpu)
    finally:
        eng.close()


def test_f32_vectors_are_not_rounded_at_load():
    """An f16 / f32 checkpoint's encoder biases and LayerNorm parameters stay f32 on the device (the reference runs its
    encoder in f32 on the widened tensors): with vectors that are NOT bf16-representable the device must follow the oracle
    that uses them as given, and sit measurably closer to it than to the oracle fed their bf16 roundings.  Matrices are
    bf16 here and there (MFMA operands; tests/test_oracle_loader_precision.py prices both roundings on the CPU)."""
    cfg = C.AUDIO_TINY
    sd = synth.synth_state_dict(cfg, C.TEXT_TINY, seed=9, init="stress", dtype=torch.float32)
    g = torch.Generator().manual_seed(1)
    exact, rounded = {}, {}
    for k, v in sd.items():
        if k.startswith("audio_tower.") and v.dim() == 1:
            v = (v + 0.37 * (v.abs() + 0.05) * torch.rand(v.shape, generator=g) * 2.0 ** -7).to(torch.float32)   # off the bf16 grid
            exact[k], rounded[k] = v, v.to(torch.bfloat16).to(torch.float32)
        else:
            exact[k] = rounded[k] = v.to(torch.bfloat16)
    e = gpu_util.Engine("tiny", max_audio_seconds=30)
    try:
        e.load_state_dict(exact)
        mel = (torch.randn(128, 300, generator=g) * 0.5).numpy()
        got = e.encode(mel)
        with torch.no_grad():
            want = encoder.encode(mel, decoder.Weights(exact), cfg, P.DEVICE).numpy()
            other = encoder.encode(mel, decoder.Weights(rounded), cfg, P.DEVICE).numpy()
        r_exact = np.linalg.norm(got - want) / np.linalg.norm(want)
        r_round = np.linalg.norm(got - other) / np.linalg.norm(other)
        r_gap = np.linalg.norm(want - other) / np.linalg.norm(want)
        print(f"device vs oracle(f32 vectors) {r_exact:.2e}; vs oracle(bf16-rounded vectors) {r_round:.2e}; oracle gap {r_gap:.2e}")
        assert r_exact < 1e-2 and r_exact < r_round
    finally:
        e.close()


def test_window_attention_kernels_agree(tiny):
    """The encoder's window attention at head_dim 64 runs on the wav2vec2 path's transposed-score kernel (a window = a clip of
    <= 104 tokens); the 16-row kernel of round 1 stays behind the knob.  Full 0.6B width (14 heads x 64), 2 layers, a clip with
    several windows and a ragged last one: both against the oracle, and close to each other."""
    import dataclasses
    cfg = dataclasses.replace(C.AUDIO_SMALL, layers=2)
    sd = synth.synth_state_dict(cfg, dataclasses.replace(C.TEXT_SMALL, layers=1), seed=21, init="stress")
    e = gpu_util.Engine("0.6B", max_audio_seconds=30, enc_layers=2, dec_layers=1)
    try:
        e.load_state_dict(sd)
        g = torch.Generator().manual_seed(5)
        mel = (torch.randn(128, 2350, generator=g) * 0.5).numpy()
        outs = []
        for knob in (0, 1):
            e.set_tuning("enc_attn", knob)
            got = e.encode(mel)
            _check(got, sd, mel, cfg)
            outs.append(got)
        rel = np.linalg.norm(outs[0] - outs[1]) / np.linalg.norm(outs[0])
        print(f"window attention kernels: rel-L2 {rel:.2e}")
        assert rel < 5e-3
    finally:
        e.set_tuning("enc_attn", 1)
        e.close()


@pytest.mark.parametrize("n", [100, 250, 300, 530])
def test_device_encoder_vs_transformers_golden(n):
    """The DEVICE encoder against the independent implementation's output directly (tests/golden/hf_tiny.npz: transformers'
    Qwen3-ASR audio tower in f32 on seeded f32 weights; the oracle agrees with it to 1e-4 in tests/test_oracle_hf.py).  Matrices
    go to the device as bf16 (what the loader does with an f32 checkpoint), vectors stay f32; the bar is the stated cost of the
    bf16 MFMA operands, relative L2 < 2e-2 (measured ~5e-3)."""
    import os
    from conftest import GOLDEN
    G = np.load(os.path.join(GOLDEN, "hf_tiny.npz"))
    sd = synth.synth_state_dict(C.AUDIO_TINY, C.TEXT_TINY, seed=1234, init="stress", dtype=torch.float32)
    dev_sd = {k: (v.to(torch.bfloat16) if (v.dim() >= 2 or not k.startswith("audio_tower.")) else v) for k, v in sd.items()}
    e = gpu_util.Engine("tiny", max_audio_seconds=30)
    try:
        e.load_state_dict(dev_sd)
        got = e.encode(G[f"mel_{n}"])
    finally:
        e.close()
    want = G[f"enc_out_{n}"]
    assert got.shape == want.shape
    rel = float(np.linalg.norm(got - want) / np.linalg.norm(want))
    print(f"T={n}: device vs transformers rel-L2 {rel:.2e}")
    assert rel < 2e-2
