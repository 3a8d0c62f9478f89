"""Full-size geometry (Qwen3-ASR-0.6B: 896/18/14 encoder, 1024/28/16/8x128 decoder, vocab 151936) on
seeded synthetic weights: exercises the shape-specialised kernels (tuned decode GEMVs, hd = 128
attention, K = 4320 implicit-GEMM convs) that the tiny geometry does not reach.

Bars: logits (bf16 values) within 6 bf16 ulps of the largest logit magnitude and relative L2 < 3e-2 (28 layers of bf16 rounding noise; measured 1.7e-2)
vs oracle DEVICE policy, teacher-forced token check with the same margin, plus size-independent properties at BASELINE batch sizes: batch invariance (a clip's tokens do
not depend on what else is in the batch or on its slot), determinism across runs.
"""
import numpy as np
import pytest
import torch
from oracle import config as C, decoder, pipeline, precision as P
from qasr import synth
import gpu_util

pytestmark = pytest.mark.gpu


def _tol(ref):
    """6 bf16 ulps at the largest logit magnitude (logits are bf16 values).  Where the number comes from: the maximum over
    the 151 936 logits is an extreme-value statistic of cascaded bf16 rounding flips -- on the CPU alone two rounding
    policies of the oracle (REFERENCE vs DEVICE vs F32, same input) already differ by 2.2-2.5 ulps at rel-L2 1.2-1.5e-2;
    the device sits at 3-4.5 ulps / rel-L2 1.6e-2 against the DEVICE policy, and which logit carries the maximum moves
    with any change of summation order or exp implementation (scratch/dbg_pa.py).  The robust bar is the rel-L2 one."""
    m = float(np.abs(ref).max())
    return 6.0 * 2.0 ** (np.floor(np.log2(max(m, 1e-3))) - 7)


@pytest.fixture(scope="module")
def full(sd_small_stress):
    sd = sd_small_stress
    e = gpu_util.Engine("0.6B", max_batch=32, max_audio_seconds=6, max_new_tokens=32)
    e.load_state_dict(sd)
    yield e, sd
    e.close()


def test_full_size_vs_oracle(full):
    eng, sd = full
    model = pipeline.OracleModel(sd, C.AUDIO_SMALL, C.TEXT_SMALL, C.TOKENS, P.DEVICE)
    pcm = synth.synth_waveform(0, 3.0)
    toks = eng.transcribe_batch([pcm], max_tokens=6, ignore_eos=True)[0]
    assert len(toks) == 6
    with torch.no_grad():
        mel = model.mel(pcm)
        got_mel = eng.mel(pcm)
        assert np.abs(got_mel - mel).max() < 1e-4
        emb = model.encode(mel)
        got_emb = eng.encode(mel)
        rel = np.linalg.norm(got_emb - P.bf16_round(emb).numpy()) / np.linalg.norm(emb.numpy())
        print("encoder rel", rel)
        assert rel < 1e-2
        # decoder: feed the GPU's own encoder output to both sides, compare logits teacher-forced
        emb_g = torch.from_numpy(got_emb)
        logits, state, ids = decoder.prefill(emb_g, model.W, model.text_cfg, P.DEVICE, model.tok)
        assert len(ids) == 16 + 39
        got = eng.prefill_logits(got_emb)
        ref = logits.numpy()
        d = np.abs(got - ref)
        rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        print("prefill logits max|d|", d.max(), "tol", _tol(ref), "rel", rel, "max|logit|", np.abs(ref).max())
        assert d.max() <= _tol(ref) and rel < 3e-2
        forced = eng.decode_forced(toks[:5])
        for i, t in enumerate(toks[:5]):
            assert logits[t] >= logits.max() - _tol(logits.numpy()), (i, t)
            logits = decoder.decode_step(t, model.W, model.text_cfg, state, P.DEVICE)
            ref = logits.numpy()
            d = np.abs(forced[i] - ref)
            rel = np.linalg.norm(forced[i] - ref) / np.linalg.norm(ref)
            print(f"step {i} logits max|d|", d.max(), "tol", _tol(ref), "rel", rel)
            assert d.max() <= _tol(ref) and rel < 3e-2


def test_qk_norm_rope_in_the_projection_epilogue_is_bit_identical(full):
    """Prompt pass: q/k RMSNorm + RoPE + cache write run in the q|k|v projection's epilogue on head tiles (gemm.h MODE 2, EpiQkHeads; knob
    pp_fuse_qk, default) or as a separate launch over the projection's output (qk_norm_rope_wide_kernel): the same arithmetic element for
    element -> same prompt-pass logits, same forced-step logits (the K cache rows), same tokens for a ragged batch (partial row tiles)."""
    eng, _ = full
    emb = P.bf16_round(torch.randn(77, 1024, generator=torch.Generator().manual_seed(11)) * 0.5).numpy()
    clips = [synth.synth_waveform(20 + k, 0.7 + 0.31 * (k % 9)) for k in range(19)]
    res = []
    for fuse in (1, 0):
        eng.set_tuning("pp_fuse_qk", fuse)
        first = eng.prefill_logits(emb)
        steps = eng.decode_forced([5, 77, 151643])
        res.append((first, steps, eng.transcribe_batch(clips, max_tokens=5, ignore_eos=True)))
    eng.set_tuning("pp_fuse_qk", 1)
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1]) and res[0][2] == res[1][2]


def test_batch_invariance_and_determinism(full):
    """B = 32 (two MFMA batch tiles, the benchmark's shape) vs B = 1: identical token streams."""
    eng, sd = full
    clips = [synth.synth_waveform(k, 1.0 + 0.13 * (k % 7)) for k in range(32)]
    a = eng.transcribe_batch(clips, max_tokens=5, ignore_eos=True)
    b = eng.transcribe_batch(clips, max_tokens=5, ignore_eos=True)
    assert a == b
    for k in (0, 15, 16, 31):
        assert eng.transcribe_batch([clips[k]], max_tokens=5, ignore_eos=True)[0] == a[k]
    sub = eng.transcribe_batch(clips[5:22], max_tokens=5, ignore_eos=True)      # 17 rows, other slots
    assert sub == a[5:22]


def test_stale_memory_does_not_leak_into_results(full):
    """Engines recycle HBM: a second engine created after a first one has dirtied memory (NaN patterns in
    what becomes the KV cache) must give the same tokens as a fresh one."""
    import torch as _t
    eng, sd = full
    clips = [synth.synth_waveform(3, 1.0)]
    ref = eng.transcribe_batch(clips, max_tokens=4, ignore_eos=True)
    if _t.cuda.is_available():
        junk = _t.full((1 << 28,), float("nan"), dtype=_t.bfloat16, device="cuda")   # 512 MB of NaNs, then freed
        del junk
        _t.cuda.empty_cache()
    e2 = gpu_util.Engine("0.6B", max_batch=2, max_audio_seconds=6, max_new_tokens=32)
    try:
        e2.load_state_dict(sd)
        assert e2.transcribe_batch(clips, max_tokens=4, ignore_eos=True) == ref
    finally:
        e2.close()


def test_30s_clips_batch8_properties():
    """configs[1] shape (8 x 30 s, full geometry): size-independent properties -- every clip yields exactly
    max_tokens ids in range, results are deterministic, independent of batch composition (B=8 vs B=1 vs a
    permuted batch) and identical through the row-group-split decode graph (QASR_DECODE_SPLIT is a launch-time
    partition of independent rows)."""
    sd = synth.synth_state_dict(C.AUDIO_SMALL, C.TEXT_SMALL, seed=0, init="hf")
    eng = gpu_util.Engine("0.6B", max_batch=8, max_audio_seconds=30, max_new_tokens=64)
    try:
        eng.load_state_dict(sd)
        clips = [synth.synth_waveform(k, 30.0) for k in range(8)]
        a = eng.transcribe_batch(clips, max_tokens=12, ignore_eos=True)
        assert all(len(t) == 12 and all(0 <= x < C.TEXT_SMALL.vocab for x in t) for t in a)
        assert eng.transcribe_batch(clips, max_tokens=12, ignore_eos=True) == a
        perm = [5, 2, 7, 0, 3, 6, 1, 4]
        b = eng.transcribe_batch([clips[i] for i in perm], max_tokens=12, ignore_eos=True)
        assert [b[perm.index(i)] for i in range(8)] == a
        assert eng.transcribe_batch([clips[3]], max_tokens=12, ignore_eos=True)[0] == a[3]
        # ragged: a 30 s clip next to very short ones
        mix = [clips[0], synth.synth_waveform(9, 0.3), clips[1][:100000], synth.synth_waveform(10, 12.34)]
        m = eng.transcribe_batch(mix, max_tokens=12, ignore_eos=True)
        assert m[0] == a[0]
        for i in (1, 2, 3):
            assert eng.transcribe_batch([mix[i]], max_tokens=12, ignore_eos=True)[0] == m[i]
        mel = eng.mel(clips[2])
        assert mel.shape == (128, 3000) and np.isfinite(mel).all()
        assert eng.encode(mel).shape == (390, 1024)
    finally:
        eng.close()


def test_1p7b_preset_geometry():
    """Qwen3-ASR-1.7B preset (encoder 1024/24/16 -> 2048, decoder hidden 2048 / inter 6144; AudioEncoder.swift:51-68,
    Configuration.swift:89-100) with 4 decoder/encoder layers to keep the CPU oracle affordable: exercises the
    K = 2048 tuned GEMVs, the K = 6144 down-projection (two-phase form of dec_gemv_wide.hip, and the generic kernel it replaced:
    the same teacher-forced check passes with `gemv_wide = 0`) and the 2048-wide persistent LM head."""
    import dataclasses
    a = dataclasses.replace(C.AUDIO_LARGE, layers=4)
    t = dataclasses.replace(C.TEXT_LARGE, layers=4)
    sd = synth.synth_state_dict(a, t, seed=2, init="stress")
    eng = gpu_util.Engine("1.7B", max_batch=4, max_audio_seconds=4, max_new_tokens=16, enc_layers=4, dec_layers=4)
    try:
        assert (eng.cfg.hidden, eng.cfg.inter, eng.cfg.enc_d_model, eng.cfg.enc_out_dim, eng.cfg.bits) == (2048, 6144, 1024, 2048, 8)
        eng.load_state_dict(sd)
        model = pipeline.OracleModel(sd, a, t, C.TOKENS, P.DEVICE)
        pcm = synth.synth_waveform(1, 2.0)
        toks = eng.transcribe_batch([pcm], max_tokens=5, ignore_eos=True)[0]
        with torch.no_grad():
            mel = model.mel(pcm)
            emb = model.encode(mel)
            got_emb = eng.encode(mel)
            assert got_emb.shape == (26, 2048)
            assert np.linalg.norm(got_emb - P.bf16_round(emb).numpy()) / np.linalg.norm(emb.numpy()) < 1e-2
            logits, state, _ = decoder.prefill(torch.from_numpy(got_emb), model.W, t, P.DEVICE, model.tok)
            got = eng.prefill_logits(got_emb)
            ref = logits.numpy()
            assert np.abs(got - ref).max() <= _tol(ref) and np.linalg.norm(got - ref) / np.linalg.norm(ref) < 3e-2
            for i, tk in enumerate(toks):
                assert logits[tk] >= logits.max() - _tol(logits.numpy()), (i, tk)
                if i + 1 < len(toks):
                    logits = decoder.decode_step(tk, model.W, t, state, P.DEVICE)
            # the kernel the K = 6144 matrix took before: the forced decode's logits of the two agree within the logit bound
            eng.prefill_logits(got_emb)
            lg_wide = eng.decode_forced(toks[:4])
            eng.set_tuning("gemv_wide", 0)
            try:
                eng.prefill_logits(got_emb)
                lg_gen = eng.decode_forced(toks[:4])
            finally:
                eng.set_tuning("gemv_wide", 1)
            assert np.abs(lg_wide - lg_gen).max() <= _tol(lg_gen) and np.linalg.norm(lg_wide - lg_gen) / np.linalg.norm(lg_gen) < 2e-2
        four = [synth.synth_waveform(k, 1.0 + 0.5 * k) for k in range(4)]
        b4 = eng.transcribe_batch(four, max_tokens=5, ignore_eos=True)
        assert b4[1] == eng.transcribe_batch([four[1]], max_tokens=5, ignore_eos=True)[0]
    finally:
        eng.close()
    # 18 rows: two 16-row groups on gridDim.y, the second with two live rows; rows never interact
    eng = gpu_util.Engine("1.7B", max_batch=18, max_audio_seconds=2, max_new_tokens=8, enc_layers=4, dec_layers=4)
    try:
        eng.load_state_dict(sd)
        clips = [synth.synth_waveform(k, 0.6 + 0.05 * k) for k in range(18)]
        b18 = eng.transcribe_batch(clips, max_tokens=4, ignore_eos=True)
        for k in (0, 15, 16, 17):
            assert b18[k] == eng.transcribe_batch([clips[k]], max_tokens=4, ignore_eos=True)[0], k
    finally:
        eng.close()


def test_real_speech_fixture(full):
    """The reference's own test clip (20 s, digital silence around 3 s of speech) at full geometry.  The harness
    resamples 24 kHz -> 16 kHz with scipy (the reference's AVAudioConverter is closed source and out of scope;
    both sides get the same 16 kHz samples).  Checks the parts that real audio stresses and synthetic noise does
    not: the 1e-10 floor / max-8 clamp on exact silence and a per-clip max set by a short loud segment."""
    import os
    from scipy.signal import resample_poly
    from conftest import GOLDEN
    from oracle import mel as omel
    from qasr.model import load_wav
    eng, sd = full
    pcm24, rate = load_wav(os.path.join(GOLDEN, "test_audio.wav"))
    assert rate == 24000
    pcm = resample_poly(pcm24.astype(np.float64), 2, 3).astype(np.float32)[: 16000 * 6 + 16000 * 3]   # 0-9 s: silence + speech
    pcm = pcm[16000 * 4:]                                                                            # 4-9 s (speech at 1.2-4.3 s)
    assert pcm.shape[0] == 80000                                                                      # configs[0]: 5 s clip
    ref_mel = omel.log_mel(pcm)
    got_mel = eng.mel(pcm)
    assert got_mel.shape == (128, 500)
    assert np.abs(got_mel - ref_mel).max() < 1e-4
    assert np.isclose(got_mel.min(), ref_mel.min(), atol=1e-6)            # the clamp floor (silent frames) matches
    model = pipeline.OracleModel(sd, C.AUDIO_SMALL, C.TEXT_SMALL, C.TOKENS, P.DEVICE)
    with torch.no_grad():
        emb = model.encode(ref_mel)
        got = eng.encode(ref_mel)
        assert got.shape == (65, 1024)                                    # 5 s -> 65 audio tokens, prompt 81
        assert np.linalg.norm(got - P.bf16_round(emb).numpy()) / np.linalg.norm(emb.numpy()) < 1e-2
    toks = eng.transcribe_batch([pcm], max_tokens=4, ignore_eos=True)[0]
    assert len(toks) == 4
