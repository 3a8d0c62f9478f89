"""Forced aligner on the GPU through the C ABI vs the oracle.

Tolerances: the classify logits are bf16 values produced after 2 (tiny) / 28 (full) decoder layers; the device differs
from the oracle by f32 association order and the flash-tile rounding of P (oracle policy DEVICE restates it), so logits
are compared within 6 bf16 ulps of the largest |logit| and rel-L2 < 3e-2 (the bounds used for the ASR logits in
test_gpu_full.py), and the raw class index must match wherever the oracle's top-2 margin exceeds that tolerance.
Host logic (slots, LIS fix-up, seconds, alignLong driver) must match exactly given the same raw indices."""
import numpy as np
import pytest
import torch
from oracle import aligner as OA, config as OC, pipeline, precision as P, tokenizer as otok
from qasr import synth, config as QC
from qasr.aligner import Qwen3ForcedAligner

pytestmark = pytest.mark.gpu

TS_TINY = 506


def _tol(ref):
    """6 bf16 ulps at the binade of the largest |logit| (a bf16 ulp there is 2^(floor(log2 max) - 7)); see test_gpu_full._tol.
    The 300 000 logits of the full-size case sit at 2 ulps typical / 2.7 ulps extreme (scratch/dbg_pa.py)."""
    return 6 * 2.0 ** (np.floor(np.log2(float(np.abs(ref).max()))) - 7)


def _check_logits(got, ref, raw):
    tol = _tol(ref)
    assert np.abs(got - ref).max() <= tol, (np.abs(got - ref).max(), tol)
    assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 3e-2
    srt = np.sort(ref, axis=1)
    decided = (srt[:, -1] - srt[:, -2]) > 2 * np.abs(got - ref).max()      # the measured error cannot flip these rows
    assert (np.asarray(raw)[decided] == ref.argmax(1)[decided]).all()
    assert (np.asarray(raw) == got.argmax(1)).all()          # device argmax == lowest index of the device logits' max


def _bpe_fixture():
    b2u = otok.byte_to_unicode()
    vocab = {}
    for b in range(256):
        vocab.setdefault(b2u[b], len(vocab))
    merges = "#version: 0.2\nl a\nla n\nĠ lan\ng u\ngu a\nĠlan gua\nE n\nĠ En\nĠEn g\nl i\nli s\n"
    for line in merges.split("\n"):
        if line and not line.startswith("#"):
            a, b = line.split(" ")
            vocab.setdefault(a + b, len(vocab))
    return vocab, merges


@pytest.fixture(scope="module")
def tiny():
    sd = synth.synth_state_dict(QC.AUDIO_TINY, QC.TEXT_TINY, seed=3, init="stress", classify_num=40)
    m = Qwen3ForcedAligner.from_state_dict(sd, preset="tiny-aligner", max_audio_seconds=300, max_prompt_extra=512)
    vocab, merges = _bpe_fixture()
    m.set_vocab({i: t for t, i in vocab.items()})
    m.set_merges(merges)
    oracle = pipeline.OracleModel(sd, OC.AUDIO_TINY, OC.TEXT_TINY, OC.TOKENS_TINY, P.DEVICE)
    yield m, oracle, vocab, otok.parse_merges(merges)
    m.close()


def _slotted(rng, n_words):
    ids, ts = [], []
    for w in range(n_words):
        ts.append(len(ids)); ids.append(TS_TINY)
        ids += rng.integers(10, 290, size=1 + w % 3).tolist()
        ts.append(len(ids)); ids.append(TS_TINY)
    return ids, ts


def test_tiny_forward_matches_oracle(tiny):
    m, oracle, _, _ = tiny
    rng = np.random.default_rng(0)
    for k, (sec, n_words) in enumerate(((3.0, 9), (1.0, 1), (7.3, 40))):
        pcm = synth.synth_waveform(k, sec)
        ids, ts = _slotted(rng, n_words)
        raw, logits = m.align_raw(pcm, ids, ts, want_logits=True)
        with torch.no_grad():
            emb = oracle.encode(oracle.mel(pcm))
            ref = OA.classify_logits(emb, ids, ts, oracle.W, oracle.text_cfg, oracle.policy, oracle.tok).numpy()
        assert logits.shape == ref.shape == (2 * n_words, 40)
        _check_logits(logits, ref, raw)


def test_tiny_align_end_to_end(tiny):
    """qasr_align = split -> slots (engine BPE) -> forward -> LIS fix-up -> seconds; host steps exact vs the oracle."""
    m, oracle, vocab, ranks = tiny
    text = "a language, English lists! (ok)  fin."
    pcm = synth.synth_waveform(4, 4.0)
    words = m.align(pcm, text)
    pairs = OA.split_word_pairs(text)
    ids, ts, surf = OA.prepare_for_alignment(pairs, lambda s: otok.encode(s, vocab, ranks), TS_TINY)
    assert m.prepare(text) == (ids, ts, len(surf))
    raw = m.last_raw_indices
    assert raw == m.align_raw(pcm, ids, ts)                   # same forward, deterministic
    exp = OA.words_from_indices(OA.enforce_monotonicity(raw), surf)
    assert [(w.text, w.start_time, w.end_time) for w in words] == [(t, pytest.approx(s, abs=0), pytest.approx(e, abs=0)) for t, s, e in exp]
    assert [w.text for w in words] == ["a", "language,", "English", "lists!", "(ok)", "fin."]
    assert all(w.end_time >= w.start_time for w in words)
    assert all(words[i].start_time >= words[i - 1].start_time for i in range(1, len(words)))
    assert m.last_passes == 1
    # caller-split words (the NLTokenizer languages) take the same path
    again = m.align(pcm, words=pairs)
    assert again == words
    with pytest.raises(Exception) as ei:
        m.align(pcm, "こんにちは", language="japanese")
    assert "NLTokenizer" in str(ei.value)


@pytest.mark.parametrize("seed,tseed,exp_passes", [(2, 1, 2), (4, 0, 2), (2, 2, 1), (3, 0, 1)])
def test_align_long_driver_matches_oracle(seed, tseed, exp_passes):
    """alignLong (ForcedAligner.swift:97-180) on 250 s of audio: the chunking driver, fed by the engine's own single-pass
    alignment, must reproduce the restated driver word for word.  Weight / text seeds were picked (scratch/
    align_long_explore.py) so that the cases cover: a trailing plateau -> second pass on the remaining audio and words
    (two cases), no plateau, and a plateau that starts at word 0 (nothing reliable: no words, where the reference traps)."""
    sd = synth.synth_state_dict(QC.AUDIO_TINY, QC.TEXT_TINY, seed=seed, init="stress", classify_num=40)
    m = Qwen3ForcedAligner.from_state_dict(sd, preset="tiny-aligner", max_audio_seconds=300, max_prompt_extra=512)
    try:
        vocab, merges = _bpe_fixture()
        m.set_vocab({i: t for t, i in vocab.items()})
        m.set_merges(merges)
        rng = np.random.default_rng(tseed)
        text = " ".join(rng.choice(["a", "language", "English", "lists", "lan", "gua"], size=60).tolist())
        pcm = synth.synth_waveform(tseed, 250.0)
        got = m.align_long(pcm, text)
        passes = m.last_passes
        exp, oracle_passes = OA.align_long(lambda a, t: [tuple(w) for w in m.align(a, t)], pcm, text)
        assert passes == oracle_passes == exp_passes
        assert [w.text for w in got] == [t for t, _, _ in exp]
        assert np.allclose([w.start_time for w in got], [s for _, s, _ in exp], rtol=0, atol=1e-4)
        assert np.allclose([w.end_time for w in got], [e for _, _, e in exp], rtol=0, atol=1e-4)
        # (how many words the second pass adds depends on the seeded random weights' argmax classes, which move with any
        #  last-bit change of a kernel; the driver equivalence above and the pass count are what this test pins)
        assert len(got) == len(exp) and (exp_passes == 1 or len(got) >= 10)
        short = m.align_long(pcm[:16000 * 20], text)              # below the 240 s bypass: exactly one pass
        assert m.last_passes == 1 and short == m.align(pcm[:16000 * 20], text)
    finally:
        m.close()


def test_align_batch_equals_single_clip_calls(tiny):
    """Ragged batch (different lengths, different texts, one text without any word) in one device pass == B single calls,
    bit for bit (clips are independent: per-clip mel max, windows, causal prompt)."""
    m, _, _, _ = tiny
    clips = [synth.synth_waveform(k, sec) for k, sec in enumerate((2.0, 6.5, 0.7, 4.0))]
    texts = ["a language", "English lists, a lan gua! fin", "...", "lists"]
    single = [m.align(c, t) for c, t in zip(clips, texts)]
    raws = []
    for c, t in zip(clips, texts):
        m.align(c, t)
        raws.append(m.last_raw_indices if t != "..." else [])
    got = m.align_batch(clips, texts)
    assert got == single and got[2] == []
    assert m.last_raw_batch == raws
    assert m.align_batch(clips[::-1], texts[::-1]) == single[::-1]          # batch order / composition invariance
    with pytest.raises(Exception):
        m.align_batch(clips * 3, texts * 3)                                # 12 clips > max_batch 8


def test_errors(tiny):
    m, _, _, _ = tiny
    pcm = synth.synth_waveform(0, 1.0)
    assert m.align(pcm, "") == [] and m.align(pcm, "?!") == []
    with pytest.raises(Exception):
        m.align(pcm, "a language", sample_rate=24000)
    with pytest.raises(Exception):
        m.align(np.zeros(0, np.float32), "a language")
    with pytest.raises(Exception):
        m.align_raw(pcm, [TS_TINY, 11, TS_TINY], [0, 5])        # slot position outside the text
    with pytest.raises(Exception) as ei:
        m.transcribe_batch([pcm], max_tokens=4)
    assert "forced aligner" in str(ei.value)
    too_long = "a " * 400                                      # 1200 slotted ids > max_prompt_extra (512)
    with pytest.raises(Exception):
        m.align(pcm, too_long)


def test_full_size_aligner_geometry():
    """Qwen3-ForcedAligner-0.6B geometry (24-layer 1024-wide encoder, 28-layer decoder, 5000 classes), synthetic weights,
    12 s clip with 30 words, vs the oracle."""
    sd = synth.synth_state_dict(QC.AUDIO_ALIGNER, QC.TEXT_SMALL, seed=0, init="hf", classify_num=5000)
    m = Qwen3ForcedAligner.from_state_dict(sd, preset="aligner-0.6B", max_audio_seconds=30)
    try:
        rng = np.random.default_rng(1)
        ids, ts = [], []
        for w in range(30):
            ts.append(len(ids)); ids.append(151705)
            ids += rng.integers(1000, 100000, size=1 + w % 3).tolist()
            ts.append(len(ids)); ids.append(151705)
        pcm = synth.synth_waveform(1, 12.0)
        raw, logits = m.align_raw(pcm, ids, ts, want_logits=True)
        oracle = pipeline.OracleModel(sd, OC.AUDIO_ALIGNER, OC.TEXT_SMALL, OC.TOKENS, P.DEVICE)
        with torch.no_grad():
            emb = oracle.encode(oracle.mel(pcm))
            ref = OA.classify_logits(emb, ids, ts, oracle.W, oracle.text_cfg, oracle.policy, oracle.tok).numpy()
        tol = _tol(ref)
        assert np.abs(logits - ref).max() <= tol
        assert np.linalg.norm(logits - ref) / np.linalg.norm(ref) < 3e-2
        assert (np.asarray(raw) == logits.argmax(1)).all()
        srt = np.sort(ref, axis=1)
        decided = (srt[:, -1] - srt[:, -2]) > 2 * np.abs(logits - ref).max()
        assert (np.asarray(raw)[decided] == ref.argmax(1)[decided]).all()
    finally:
        m.close()


def test_aligner_checkpoint_directory(tmp_path):
    """loadForcedAlignerWeights (WeightLoading.swift:135-232): `thinker.` key prefix, PyTorch-layout conv weights,
    un-quantised `lm_head.{weight,bias}`, vocab + merges files -> the same alignment as the in-memory upload."""
    import json
    from safetensors.torch import save_file
    sd = synth.synth_state_dict(QC.AUDIO_TINY, QC.TEXT_TINY, seed=9, init="stress", classify_num=40)
    disk = {}
    for k, v in sd.items():
        if ".conv2d" in k and k.endswith(".weight"):
            v = v.permute(0, 3, 1, 2).contiguous()             # [out, kH, kW, in] -> PyTorch [out, in, kH, kW]
        disk["thinker." + k] = v.contiguous()
    save_file(disk, str(tmp_path / "model.safetensors"))
    vocab, merges = _bpe_fixture()
    (tmp_path / "vocab.json").write_text(json.dumps(vocab, ensure_ascii=False), encoding="utf-8")
    (tmp_path / "merges.txt").write_text(merges, encoding="utf-8")
    m = Qwen3ForcedAligner(preset="tiny-aligner", model_dir=str(tmp_path), max_audio_seconds=10)
    ref = Qwen3ForcedAligner.from_state_dict(sd, preset="tiny-aligner", max_audio_seconds=10)
    try:
        ref.set_vocab({i: t for t, i in vocab.items()})
        ref.set_merges(merges)
        pcm = synth.synth_waveform(3, 5.0)
        text = "English language lists a lan"
        assert m.prepare(text) == ref.prepare(text)
        assert m.align(pcm, text) == ref.align(pcm, text)
        assert m.last_raw_indices == ref.last_raw_indices and len(m.last_raw_indices) == 10
    finally:
        m.close()
        ref.close()
