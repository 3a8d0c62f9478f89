"""HIP text decoder + greedy loop vs the CPU oracle (R5-R8 of SURVEY.md section 8a), via the C ABI.

Floating-point bar (stated): logits are bf16 values of magnitude ~1; GPU vs oracle(DEVICE policy,
same rounding points) must agree to max |d| < 0.06 (a few bf16 ulps; see test_gpu_encoder.py for
why bit equality is not reachable across different f32 association orders) and relative L2 < 1.5e-2.
Token bar: greedy tokens are checked teacher-forced -- the oracle is run along the GPU's own token
stream and every GPU token must be the oracle's argmax or within `MARGIN` of it (near-tie); the
first-token / EOS / length semantics are bit-exact integer checks.
"""
import dataclasses
import numpy as np
import pytest
import torch
from oracle import config as C, decoder, pipeline, precision as P
from qasr import synth
import gpu_util

pytestmark = pytest.mark.gpu
MARGIN = 0.06
A, T, TOK = C.AUDIO_TINY, C.TEXT_TINY, C.TOKENS_TINY


@pytest.fixture(scope="module")
def tiny():
    sd = synth.synth_state_dict(A, T, seed=3, init="stress")
    e = gpu_util.Engine("tiny", max_audio_seconds=30, max_new_tokens=48)
    e.load_state_dict(sd)
    yield e, sd, decoder.Weights(sd)
    e.close()


def _logit_check(got, ref):
    d = np.abs(got - ref)
    rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
    print(f"logits max|d|={d.max():.4f} rel={rel:.2e}")
    assert d.max() < MARGIN, d.max()
    assert rel < 1.5e-2, rel
    assert ref[int(got.argmax())] >= ref.max() - MARGIN


@pytest.mark.parametrize("n_audio", [0, 1, 33, 150, 390])
def test_prefill_logits(tiny, n_audio):
    eng, sd, W = tiny
    g = torch.Generator().manual_seed(n_audio)
    emb = P.bf16_round(torch.randn(n_audio, T.hidden, generator=g) * 0.5)
    got = eng.prefill_logits(emb.numpy())
    with torch.no_grad():
        ref, _, ids = decoder.prefill(emb, W, T, P.DEVICE, TOK)
    assert len(ids) == 16 + n_audio
    _logit_check(got, ref.numpy())


def test_prefill_with_context_and_language(tiny):
    eng, sd, W = tiny
    emb = P.bf16_round(torch.randn(20, T.hidden, generator=torch.Generator().manual_seed(1)) * 0.5)
    ctx, lang = [11, 12, 13], [40, 41]
    got = eng.prefill_logits(emb.numpy(), context_ids=ctx, language_ids=lang)
    with torch.no_grad():
        ref, _, ids = decoder.prefill(emb, W, T, P.DEVICE, TOK, context_ids=ctx, language_ids=lang)
    assert len(ids) == 16 + 20 + 5
    _logit_check(got, ref.numpy())


def test_forced_decode_steps(tiny):
    eng, sd, W = tiny
    emb = P.bf16_round(torch.randn(60, T.hidden, generator=torch.Generator().manual_seed(5)) * 0.5)
    with torch.no_grad():
        toks, logits = decoder.greedy(emb, W, T, P.DEVICE, TOK, max_tokens=20, ignore_eos=True, return_logits=True)
    got0 = eng.prefill_logits(emb.numpy())
    _logit_check(got0, logits[0].numpy())
    got = eng.decode_forced(toks[:-1])                  # feeding token i yields the logits of token i+1
    for i in range(len(toks) - 1):
        _logit_check(got[i], logits[i + 1].numpy())


def _teacher_forced_check(model, pcm, gpu_tokens, max_tokens, ignore_eos):
    """Run the oracle along the GPU token stream; every GPU token must be (near-)argmax."""
    with torch.no_grad():
        emb = model.encode(model.mel(pcm))
        logits, state, _ = decoder.prefill(emb, model.W, model.text_cfg, model.policy, model.tok)
        for i, t in enumerate(gpu_tokens):
            assert logits[t] >= logits.max() - MARGIN, (i, t, float(logits[t]), float(logits.max()))
            stop = (t == model.tok.eos and not ignore_eos) or i + 1 >= max_tokens
            if stop:
                assert i == len(gpu_tokens) - 1, "GPU kept generating past EOS / max_tokens"
                return
            logits = decoder.decode_step(t, model.W, model.text_cfg, state, model.policy)
    raise AssertionError("GPU stopped early without EOS / max_tokens")


def test_batch_pipeline_ragged(tiny):
    eng, sd, W = tiny
    model = pipeline.OracleModel(sd, A, T, TOK, P.DEVICE)
    clips = [synth.synth_waveform(0, 2.5), synth.synth_waveform(1, 1.0), synth.synth_waveform(2, 0.4),
             synth.synth_waveform(3, 7.3)]
    out = eng.transcribe_batch(clips, max_tokens=12, ignore_eos=True)
    assert [len(o) for o in out] == [12] * 4
    for pcm, toks in zip(clips, out):
        _teacher_forced_check(model, pcm, toks, 12, True)
    # batch invariance: each clip alone gives the same tokens, bit for bit
    for pcm, toks in zip(clips, out):
        assert eng.transcribe_batch([pcm], max_tokens=12, ignore_eos=True)[0] == toks


def test_eos_and_length_semantics(tiny):
    """EOS is appended then the row stops (Qwen3ASR.swift:378-379); rows stop independently."""
    eng, sd, W = tiny
    clips = [synth.synth_waveform(0, 2.5), synth.synth_waveform(1, 1.0)]
    free = eng.transcribe_batch(clips, max_tokens=10, ignore_eos=True)
    eos = free[0][3]
    e2 = gpu_util.Engine("tiny", max_audio_seconds=30, max_new_tokens=48, tok_im_end=eos)
    try:
        e2.load_state_dict(sd)
        cut = e2.transcribe_batch(clips, max_tokens=10)
        model = pipeline.OracleModel(sd, A, T, dataclasses.replace(TOK, im_end=eos), P.DEVICE)
        for pcm, toks in zip(clips, cut):
            assert 1 <= len(toks) <= 10
            assert eos not in toks[:-1]
            assert toks[-1] == eos or len(toks) == 10
            _teacher_forced_check(model, pcm, toks, 10, False)
        assert len(cut[0]) <= free[0].index(eos) + 1 or cut[0][:3] != free[0][:3]
        assert e2.transcribe_batch(clips, max_tokens=1) == [[cut[0][0]], [cut[1][0]]]
    finally:
        e2.close()


def test_no_graph_equals_graph(tiny):
    """The hipGraph-captured decode step replays exactly the eager step (tuning knob use_graph = 0 issues every
    step's launches directly on the stream)."""
    eng, sd, W = tiny
    clips = [synth.synth_waveform(4, 1.7), synth.synth_waveform(5, 0.8)]
    a = eng.transcribe_batch(clips, max_tokens=9, ignore_eos=True)
    eng.set_tuning("use_graph", 0)
    try:
        b = eng.transcribe_batch(clips, max_tokens=9, ignore_eos=True)
    finally:
        eng.set_tuning("use_graph", 1)
    c = eng.transcribe_batch(clips, max_tokens=9, ignore_eos=True)
    assert a == b == c
    # several steps per graph launch (default 8) against one step per launch, token budgets around the multiples of 8 and with
    # the EOS rule live (the poll interval is 8 steps: a multi-step graph must stop at the same token)
    for mt in (8, 9, 17, 26):
        for ignore in (True, False):
            got = {}
            for gs in (1, 4, 8):
                eng.set_tuning("graph_steps", gs)
                got[gs] = eng.transcribe_batch(clips, max_tokens=mt, ignore_eos=ignore)
            eng.set_tuning("graph_steps", 8)
            assert got[1] == got[4] == got[8], (mt, ignore)


def test_forced_decode_capacity(tiny):
    """qasr_decode_forced appends one K/V row per token: more tokens than the cache holds is QASR_ERR_CAPACITY, not an
    out-of-bounds append."""
    sd = tiny[1]
    e = gpu_util.Engine("tiny", max_audio_seconds=2, max_new_tokens=8, max_batch=1)
    try:
        e.load_state_dict(sd)
        emb = P.bf16_round(torch.randn(5, T.hidden, generator=torch.Generator().manual_seed(2)) * 0.5)
        e.prefill_logits(emb.numpy())
        cap = None
        with pytest.raises(RuntimeError, match="qasr error 5"):
            e.decode_forced([7] * 4096)
        # filling the cache exactly is fine; one more token is refused
        n_ok = 0
        while True:
            try:
                e.decode_forced([7] * 16)
                n_ok += 16
            except RuntimeError as ex:
                assert "qasr error 5" in str(ex)
                break
            assert n_ok < 4096
        assert n_ok >= 0
        with pytest.raises(RuntimeError, match="qasr error 1"):
            e.decode_forced([T.vocab])                                   # id outside the vocabulary
    finally:
        e.close()


def test_long_audio_multi_window(tiny):
    """75 s of audio (7500 frames, 75 chunks, 975 tokens, 38 attention windows, prompt 991) through the whole
    path on the tiny geometry: teacher-forced against the oracle, and the long clip does not disturb a short
    one batched with it."""
    sd = tiny[1]
    e = gpu_util.Engine("tiny", max_audio_seconds=80, max_new_tokens=16, max_batch=2)
    try:
        e.load_state_dict(sd)
        model = pipeline.OracleModel(sd, A, T, TOK, P.DEVICE)
        long_clip, short = synth.synth_waveform(6, 75.0), synth.synth_waveform(7, 1.3)
        out = e.transcribe_batch([long_clip, short], max_tokens=6, ignore_eos=True)
        _teacher_forced_check(model, long_clip, out[0], 6, True)
        assert e.transcribe_batch([short], max_tokens=6, ignore_eos=True)[0] == out[1]
        mel = e.mel(long_clip)
        assert mel.shape == (128, 7500)
        assert e.encode(mel).shape == (975, T.hidden)
    finally:
        e.close()


def test_device_decoder_vs_transformers_golden():
    """The DEVICE decoder against the independent implementation directly (tests/golden/hf_tiny.npz: transformers' Qwen3 text
    stack in f32 on seeded weights; the oracle at policy F32 agrees to 1e-4 and produces the same greedy ids in
    tests/test_oracle_hf.py).  The device runs the reference's bf16 decoder, so the bar is the bf16-vs-f32 distance the two CPU
    policies show (rel-L2 < 3e-2): prompt-pass logits and 15 teacher-forced steps along HF's greedy stream, and the device's
    own argmax must be HF's wherever HF's top-2 margin exceeds the logit error."""
    import os
    from conftest import GOLDEN
    G = np.load(os.path.join(GOLDEN, "hf_tiny.npz"))
    sd = synth.synth_state_dict(C.AUDIO_TINY, C.TEXT_TINY, seed=1234, init="stress", dtype=torch.float32)
    dev_sd = {k: (v.to(torch.bfloat16) if (v.dim() >= 2 or not k.startswith("audio_tower.")) else v) for k, v in sd.items()}
    e = gpu_util.Engine("tiny", max_audio_seconds=30, max_new_tokens=32)
    try:
        e.load_state_dict(dev_sd)
        first = e.prefill_logits(G["enc_out_250"])
        steps = e.decode_forced(G["dec_greedy_ids"][:15].astype(np.int32))
    finally:
        e.close()
    want = [G["dec_prefill_logits"]] + list(G["dec_step_logits"][:15])
    agree = 0
    for i, (got, ref) in enumerate(zip([first] + list(steps), want)):
        rel = float(np.linalg.norm(got - ref) / np.linalg.norm(ref))
        err = float(np.abs(got - ref).max())
        top2 = np.sort(ref)[-2:]
        assert rel < 3e-2, (i, rel)
        if top2[1] - top2[0] > 2 * err:
            assert int(got.argmax()) == int(ref.argmax()), i
            agree += 1
    print(f"device vs transformers: {agree} of 16 positions have a decisive margin, all agree")
    assert agree >= 4
