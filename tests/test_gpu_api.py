"""C-ABI surface on the GPU: detokeniser KATs (reference's own unit-test vectors), transcribe() text
contract, speech-core vtable adapter, safetensors / MLX-quantised checkpoint loading, error codes,
unload / memory footprint (R9, R10 and the boundary rows of SURVEY.md section 8b)."""
import ctypes as C
import json
import os
import numpy as np
import pytest
import torch
from conftest import GOLDEN
from oracle import config as OC, tokenizer as otok
from qasr import _lib, synth, config as QC
from qasr.model import Qwen3ASRModel, QasrError

pytestmark = pytest.mark.gpu
KAT = json.load(open(os.path.join(GOLDEN, "kat_reference_tests.json"), encoding="utf-8"))


@pytest.fixture(scope="module")
def sd():
    return synth.synth_state_dict(QC.AUDIO_TINY, QC.TEXT_TINY, seed=3, init="stress")


@pytest.fixture(scope="module")
def model(sd):
    m = Qwen3ASRModel.from_state_dict(sd, preset="tiny", max_audio_seconds=10, max_new_tokens=32)
    yield m
    m.close()


def _token_map(case):
    b2u = otok.byte_to_unicode()
    m = {int(k): "".join(b2u[b] for b in bs) for k, bs in case.get("map_bytes", {}).items()}
    m.update({int(k): s for k, s in case.get("map_literal", {}).items()})
    return m


@pytest.mark.parametrize("case", KAT["tokenizer_decode"], ids=lambda c: c["name"])
def test_detokenize_kats(sd, case):
    m = Qwen3ASRModel.from_state_dict(sd, preset="tiny", max_audio_seconds=2, max_new_tokens=8)
    try:
        m.set_vocab(_token_map(case))
        got = m.detokenize(case["tokens"])
        want = otok.strip_asr_prefix(otok.decode(case["tokens"], _token_map(case)))
        if "expect" in case and "<asr_text>" not in case["expect"]:
            assert got == case["expect"]
        elif "expect_contains" in case:
            assert case["expect_contains"] in got
        assert got == want
    finally:
        m.close()


def test_transcribe_text_contract(model, sd):
    vocab = {i: chr(ord("a") + i % 26) for i in range(QC.TEXT_TINY.vocab)}
    vocab[7] = "<asr_text>"
    vocab[501] = "<|im_end|>"
    model.set_vocab(vocab)
    pcm = synth.synth_waveform(0, 1.5)
    toks = model.transcribe_tokens(pcm, max_tokens=8, ignore_eos=True)
    text = model.transcribe(pcm, max_tokens=8)
    natural = model.transcribe_tokens(pcm, max_tokens=8)
    assert natural == toks[:len(natural)]
    assert text == otok.strip_asr_prefix(otok.decode(natural, vocab))
    assert model.transcribe(pcm, sample_rate=24000).startswith("[qasr error:")      # non-throwing, like the reference
    assert model.transcribe(np.zeros(0, np.float32)).startswith("[qasr error:")
    assert model.input_sample_rate == 16000


def test_stt_vtable(model):
    vt = _lib.ScSttVtable()
    assert model.lib.qasr_stt_vtable(model.h, C.byref(vt)) == 0
    assert vt.input_sample_rate(vt.context) == 16000
    assert not vt.begin_stream and not vt.push_chunk and not vt.cancel_stream
    pcm = synth.synth_waveform(2, 1.0)
    res = vt.transcribe(vt.context, pcm.ctypes.data_as(C.POINTER(C.c_float)), pcm.shape[0], 16000)
    assert res.text.decode() == model.transcribe(pcm, max_tokens=32)     # vtable uses the engine's cap
    assert res.language == b"" and res.confidence == 0.0 and res.start_time == 0.0 and res.end_time == 0.0


def test_errors_and_capacity(model):
    lib, h = model.lib, model.h
    big = np.zeros(16000 * 11, np.float32)
    toks = np.zeros((1, 33), np.int32)
    lens = np.zeros(1, np.int32)
    ptr = (C.POINTER(C.c_float) * 1)(big.ctypes.data_as(C.POINTER(C.c_float)))
    n = (C.c_size_t * 1)(big.shape[0])
    rc = lib.qasr_transcribe_batch(h, ptr, n, 1, 16000, None, toks.ctypes.data_as(C.POINTER(C.c_int32)),
                                   lens.ctypes.data_as(C.POINTER(C.c_int32)))
    assert rc == 5 and b"max_audio_seconds" in lib.qasr_last_error(h)            # QASR_ERR_CAPACITY
    o = _lib.QasrOptions()
    o.max_tokens = 1000
    small = synth.synth_waveform(0, 0.5)
    ptr = (C.POINTER(C.c_float) * 1)(small.ctypes.data_as(C.POINTER(C.c_float)))
    n = (C.c_size_t * 1)(small.shape[0])
    assert lib.qasr_transcribe_batch(h, ptr, n, 1, 16000, C.byref(o), toks.ctypes.data_as(C.POINTER(C.c_int32)),
                                     lens.ctypes.data_as(C.POINTER(C.c_int32))) == 5
    n0 = (C.c_size_t * 1)(0)
    assert lib.qasr_transcribe_batch(h, ptr, n0, 1, 16000, None, toks.ctypes.data_as(C.POINTER(C.c_int32)),
                                     lens.ctypes.data_as(C.POINTER(C.c_int32))) == 6   # QASR_ERR_EMPTY_AUDIO


def test_decode_capacity_is_checked_at_create():
    """The decode step holds a launch's batch rows in at most four 16-row tiles: a capacity beyond 64 rows is refused by qasr_create with a
    message (round 3: it was accepted and failed at the first qasr_batch_run); 64 itself is served."""
    import gpu_util
    with pytest.raises(RuntimeError, match="64 batch rows"):
        gpu_util.Engine("tiny", max_batch=65)
    with pytest.raises(RuntimeError, match="positive"):
        gpu_util.Engine("tiny", max_batch=0)
    e = gpu_util.Engine("tiny", max_batch=64, max_audio_seconds=2, max_new_tokens=8)
    try:
        e.load_state_dict(synth.synth_state_dict(QC.AUDIO_TINY, QC.TEXT_TINY, seed=3, init="stress"))
        clips = [synth.synth_waveform(k, 0.5 + 0.02 * k) for k in range(64)]
        out = e.transcribe_batch(clips, max_tokens=5, ignore_eos=True)
        assert [len(t) for t in out] == [5] * 64
        for k in (0, 17, 63):
            assert e.transcribe_batch([clips[k]], max_tokens=5, ignore_eos=True)[0] == out[k]
    finally:
        e.close()


def test_out_of_range_prompt_ids_are_refused(model):
    """context / language ids index the embedding table on the device: anything outside [0, vocab) or a negative count
    is QASR_ERR_INVALID on the host, never an out-of-bounds gather."""
    pcm = synth.synth_waveform(1, 0.8)
    V = QC.TEXT_TINY.vocab
    for bad in ({"context_ids": [-1]}, {"context_ids": [V]}, {"language_ids": [3, V + 7]}, {"language_ids": [-5]}):
        with pytest.raises(Exception, match="qasr error 1"):
            model.transcribe_batch([pcm], max_tokens=4, **bad)
    o = model._options(max_tokens=4, context_ids=[1, 2])
    o.n_context = -3
    clips = [pcm]
    ptrs = (C.POINTER(C.c_float) * 1)(clips[0].ctypes.data_as(C.POINTER(C.c_float)))
    ns = (C.c_size_t * 1)(clips[0].shape[0])
    toks = np.zeros((1, model.cfg.max_new_tokens + 1), np.int32)
    lens = np.zeros(1, np.int32)
    assert model.lib.qasr_transcribe_batch(model.h, ptrs, ns, 1, 16000, C.byref(o), toks.ctypes.data_as(C.POINTER(C.c_int32)),
                                           lens.ctypes.data_as(C.POINTER(C.c_int32))) == 1
    assert model.transcribe_batch([pcm], max_tokens=4, context_ids=[0, V - 1]) is not None      # the edges are fine


def test_refinalize_lifecycle(sd):
    """finalize -> transcribe -> replace a tensor -> (run refused) -> finalize -> transcribe: the second build must not
    replay a decode graph that points at the first build's freed KV caches / packed weights."""
    m = Qwen3ASRModel.from_state_dict(sd, preset="tiny", max_audio_seconds=4, max_new_tokens=16)
    try:
        clips = [synth.synth_waveform(0, 1.1), synth.synth_waveform(1, 0.7)]
        a = m.transcribe_batch(clips, max_tokens=8, ignore_eos=True)
        m.batch_begin(clips, max_tokens=8, ignore_eos=True)
        name = "model.layers.0.mlp.down_proj.weight"
        t = sd[name].contiguous()
        shape = (C.c_int64 * t.dim())(*t.shape)
        m._check(m.lib.qasr_set_tensor(m.h, name.encode(), C.c_void_p(t.data_ptr()), 1, shape, t.dim()))
        assert not m.is_loaded
        assert m.lib.qasr_batch_run(m.h) == 3 and m.lib.qasr_batch_rewind(m.h) == 3          # QASR_ERR_NOT_LOADED
        m._check(m.lib.qasr_finalize(m.h))
        assert m.lib.qasr_batch_run(m.h) != 0                                                 # the old batch is gone too
        assert m.transcribe_batch(clips, max_tokens=8, ignore_eos=True) == a                  # same weights, same tokens
        # a different tensor gives different tokens through the re-captured graph
        t2 = (t.float() * -1.0).to(torch.bfloat16).contiguous()
        m._check(m.lib.qasr_set_tensor(m.h, name.encode(), C.c_void_p(t2.data_ptr()), 1, shape, t2.dim()))
        m._check(m.lib.qasr_finalize(m.h))
        b = m.transcribe_batch(clips, max_tokens=8, ignore_eos=True)
        assert b != a
        m.unload()
        assert m.lib.qasr_batch_run(m.h) == 3
    finally:
        m.close()


def test_non_finite_logits_are_an_error_status(sd):
    """NaN weights make every logit NaN: the greedy bookkeeping flags it and qasr_batch_tokens returns QASR_ERR_HIP
    instead of quietly emitting token 0."""
    bad = dict(sd)
    bad["model.norm.weight"] = torch.full_like(sd["model.norm.weight"], float("nan"))
    m = Qwen3ASRModel.from_state_dict(bad, preset="tiny", max_audio_seconds=2, max_new_tokens=8)
    try:
        with pytest.raises(Exception, match="qasr error 2"):
            m.transcribe_batch([synth.synth_waveform(0, 0.6)], max_tokens=4, ignore_eos=True)
        assert "non-finite" in m.lib.qasr_last_error(m.h).decode()
        assert m.transcribe(synth.synth_waveform(0, 0.6), max_tokens=4).startswith("[qasr error:")
    finally:
        m.close()


def test_unload_and_footprint(sd):
    m = Qwen3ASRModel.from_state_dict(sd, preset="tiny", max_audio_seconds=2, max_new_tokens=8)
    try:
        assert m.is_loaded
        # the uploaded tensors (bf16) + what finalize derives from them (fused q|k|v / gate|up, fragment-major decode images): between one
        # and three times the checkpoint, and what the device actually gave up for weights is at least that much
        up = sum(t.numel() * 2 for t in sd.values())
        assert up < m.memory_footprint <= 3 * up
        m.unload()
        assert not m.is_loaded and m.memory_footprint == 0
        assert m.transcribe(synth.synth_waveform(0, 0.5), max_tokens=8).startswith("[qasr error:")
    finally:
        m.close()


@pytest.mark.parametrize("bits,sb_dtype", [(4, torch.bfloat16), (8, torch.bfloat16), (4, torch.float16)])
def test_safetensors_and_mlx_quantised_checkpoint(tmp_path, sd, bits, sb_dtype):
    """A sharded safetensors directory in the reference's layout (WeightLoading.swift:235-323): MLX-quantised triplets for
    every decoder Linear and the tied embedding, mixed on-disk float dtypes for the audio tower.  The loader must hand the
    engine exactly the tensors `qasr_set_tensor` would: same tokens as an engine built from the same triplets in memory
    (packed words untouched, bf16 scales kept, f16 scales widened to f32, f16 / f32 encoder vectors widened to f32 --
    nothing rounded except f16 / f32 MATRICES of the audio tower, which become bf16 MFMA operands).  Parity of the quantised
    kernels themselves against the oracle: tests/test_gpu_quant.py."""
    from safetensors.torch import save_file
    qsd = synth.quantize_state_dict(sd, bits)
    tensors, expect = {}, {}
    for name, t in qsd.items():
        if name.endswith(".scales") or name.endswith(".biases"):
            tensors[name] = t.to(sb_dtype)
            expect[name] = t if sb_dtype == torch.bfloat16 else t.to(sb_dtype).to(torch.float32)
        elif name.startswith("audio_tower.") and name.endswith(".bias"):
            tensors[name] = t.to(torch.float16)            # mixed on-disk float dtypes
            expect[name] = t.to(torch.float16).to(torch.float32)   # encoder vectors: widened to f32, no bit lost
        elif name.startswith("audio_tower.") and name.endswith("layer_norm.weight"):
            tensors[name] = t.to(torch.float32) * 1.0009765625      # off the bf16 grid: must arrive as it is on disk
            expect[name] = tensors[name]
        elif name.endswith("layer_norm.weight"):
            tensors[name] = t.to(torch.float32)
            expect[name] = t
        else:
            tensors[name] = t
            expect[name] = t
    keys = sorted(tensors)
    half = len(keys) // 2
    # safetensors has no uint32 in older torch: store as int32 and patch the dtype string in the header
    for i, part in enumerate((keys[:half], keys[half:])):
        path = tmp_path / f"model-0000{i + 1}-of-00002.safetensors"
        save_file({k: tensors[k].contiguous() for k in part}, str(path))
        raw = path.read_bytes()
        hlen = int.from_bytes(raw[:8], "little")
        header = json.loads(raw[8:8 + hlen])
        for k, v in header.items():
            if k != "__metadata__" and v["dtype"] == "I32":
                v["dtype"] = "U32"
        hb = json.dumps(header, separators=(",", ":")).encode()
        hb += b" " * (hlen - len(hb))
        assert len(hb) == hlen
        path.write_bytes(raw[:8] + hb + raw[8 + hlen:])
    (tmp_path / "vocab.json").write_text(json.dumps({"a": 0, "b": 1, "<asr_text>": 7}))
    (tmp_path / "tokenizer_config.json").write_text(json.dumps({"added_tokens_decoder": {"501": {"content": "<|im_end|>"}}}))
    m = Qwen3ASRModel(preset="tiny", model_dir=str(tmp_path), max_audio_seconds=4, max_new_tokens=16, bits=bits)
    ref = Qwen3ASRModel.from_state_dict(expect, preset="tiny", max_audio_seconds=4, max_new_tokens=16, bits=bits)
    try:
        assert m.is_loaded
        assert m.memory_footprint == ref.memory_footprint          # same bytes whether loaded from disk or set tensor by tensor
        # packed words + scales + biases + packed decode images + one layer of bf16 scratch: no resident bf16 expansion of the decoder
        # (a bf16 copy of every decoder Linear alone would be 2 bytes per parameter)
        dec_params = sum(int(np.prod(t.shape)) * (32 // bits) for k, t in expect.items() if k.startswith("model.layers.") and k.endswith(".weight") and t.dtype in (torch.int32, torch.uint32))
        packed_bytes = sum(t.numel() * t.element_size() for t in expect.values())
        layer_bf16 = 2 * dec_params // m.cfg.dec_layers
        assert m.memory_footprint <= 2 * packed_bytes + layer_bf16 + 4096, (m.memory_footprint, packed_bytes, layer_bf16)
        # (at this tiny geometry the audio tower dominates; the full-width statement is tests/test_gpu_quant.py::test_quantised_engine_device_bytes)
        pcm = synth.synth_waveform(1, 2.0)
        assert m.transcribe_tokens(pcm, max_tokens=10, ignore_eos=True) == ref.transcribe_tokens(pcm, max_tokens=10, ignore_eos=True)
        assert m.detokenize([0, 1, 501, 7, 1, 0]) == "ba"
    finally:
        m.close()
        ref.close()


def test_malformed_quantised_checkpoint_is_refused(tmp_path, sd):
    """Untrusted metadata: scales of the wrong shape / a non-float dtype, and shape entries that overflow, are load errors
    (QASR_ERR_IO), not out-of-bounds reads."""
    from safetensors.torch import save_file
    qsd = synth.quantize_state_dict(sd, 4)
    stem = "model.layers.0.mlp.up_proj"

    def write(mut):
        t = {k: v.contiguous() for k, v in qsd.items()}
        mut(t)
        path = tmp_path / "model.safetensors"
        save_file(t, str(path))
        raw = path.read_bytes()
        hlen = int.from_bytes(raw[:8], "little")
        header = json.loads(raw[8:8 + hlen])
        for k, v in header.items():
            if k != "__metadata__" and v["dtype"] == "I32" and k.endswith(".weight"):
                v["dtype"] = "U32"
        hb = json.dumps(header, separators=(",", ":")).encode()
        hb += b" " * (hlen - len(hb))
        path.write_bytes(raw[:8] + hb + raw[8 + hlen:])

    def short_scales(t):
        t[stem + ".scales"] = t[stem + ".scales"][:, :1].repeat(1, 1)[:-1]
    def int_biases(t):
        t[stem + ".biases"] = torch.zeros_like(t[stem + ".biases"], dtype=torch.int16)
    for mut in (short_scales, int_biases):
        write(mut)
        with pytest.raises(Exception, match="qasr_create failed"):
            Qwen3ASRModel(preset="tiny", model_dir=str(tmp_path), max_audio_seconds=2, max_new_tokens=8, bits=4)
    # a header whose JSON nests deeper than any checkpoint does
    p = tmp_path / "model.safetensors"
    hb = (b"[" * 5000) + (b"]" * 5000)
    p.write_bytes(len(hb).to_bytes(8, "little") + hb)
    with pytest.raises(Exception, match="qasr_create failed"):
        Qwen3ASRModel(preset="tiny", model_dir=str(tmp_path), max_audio_seconds=2, max_new_tokens=8, bits=4)


def test_missing_checkpoint_dir():
    with pytest.raises(Exception) as ei:
        Qwen3ASRModel(preset="tiny", model_dir="/nonexistent/dir")
    assert "qasr_create failed" in str(ei.value)


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


def test_bpe_encode_matches_oracle(model):
    vocab, merges = _bpe_fixture()
    assert max(vocab.values()) < QC.TEXT_TINY.vocab
    model.set_vocab({i: t for t, i in vocab.items()})
    model.set_merges(merges)
    ranks = otok.parse_merges(merges)
    for text in ["language English", "a language", " lan gua", "x\ty  z\nlanguage", "", "café 來", "language"]:
        assert model.encode_text(text) == otok.encode(text, vocab, ranks), text
    # strings and ids are interchangeable on the transcribe surface (Qwen3ASR.swift:203-206,228-232)
    pcm = synth.synth_waveform(5, 1.2)
    by_str = model.transcribe(pcm, language="English", context="a language", max_tokens=6)
    by_ids = model.transcribe(pcm, language_ids=model.encode_text("language English"),
                              context_ids=model.encode_text("a language"), max_tokens=6)
    assert by_str == by_ids and not by_str.startswith("[qasr error")
    plain = model.transcribe(pcm, max_tokens=6)
    assert isinstance(plain, str) and not plain.startswith("[qasr error")


def test_slow_path_decoding_options(model, sd):
    """Qwen3DecodingOptions: repetition penalty / no-repeat-ngram / temperature run pickNextToken on host logits
    (generateSlow, Qwen3ASR.swift:396-433).  Teacher-forced against the oracle sampler."""
    import torch
    from oracle import decoder, pipeline, precision as P
    oracle = pipeline.OracleModel(sd, OC.AUDIO_TINY, OC.TEXT_TINY, OC.TOKENS_TINY, P.DEVICE)
    pcm = synth.synth_waveform(0, 2.0)
    greedy = model.transcribe_batch([pcm], max_tokens=12, ignore_eos=True)[0]
    for opts in (dict(repetition_penalty=1.5), dict(no_repeat_ngram_size=2), dict(repetition_penalty=1.2, no_repeat_ngram_size=3)):
        got = model.transcribe_batch([pcm], max_tokens=12, ignore_eos=True, **opts)[0]
        assert len(got) == 12
        with torch.no_grad():
            emb = oracle.encode(oracle.mel(pcm))
            logits, state, _ = decoder.prefill(emb, oracle.W, oracle.text_cfg, oracle.policy, oracle.tok)
            hist = []
            for i, t in enumerate(got):
                lg = logits.numpy().copy()
                want = decoder.pick_next_token(lg, hist, opts.get("repetition_penalty", 1.0), opts.get("no_repeat_ngram_size", 0))
                if t != want:      # near-tie: the GPU's choice must be within the logit margin after the same edits
                    rp = opts.get("repetition_penalty", 1.0)
                    for g in set(hist):
                        lg[g] = lg[g] / rp if lg[g] > 0 else lg[g] * rp
                    assert lg[t] >= lg[want] - 0.06, (opts, i, t, want)
                hist.append(t)
                if i + 1 < len(got):
                    logits = decoder.decode_step(t, oracle.W, oracle.text_cfg, state, oracle.policy)
        if "no_repeat_ngram_size" in opts and opts["no_repeat_ngram_size"] == 2:
            bigrams = list(zip(got, got[1:]))
            assert len(set(bigrams)) == len(bigrams)                  # no repeated 2-gram
    # default options keep the fast path: identical to greedy
    assert model.transcribe_batch([pcm], max_tokens=12, ignore_eos=True, repetition_penalty=1.0)[0] == greedy
    a = model.transcribe_batch([pcm], max_tokens=10, ignore_eos=True, temperature=1.0, seed=3)[0]
    b = model.transcribe_batch([pcm], max_tokens=10, ignore_eos=True, temperature=1.0, seed=3)[0]
    c = model.transcribe_batch([pcm], max_tokens=10, ignore_eos=True, temperature=1.0, seed=4)[0]
    assert a == b and len(a) == 10 and all(0 <= x < QC.TEXT_TINY.vocab for x in a)
    assert a != c or a != greedy


def test_device_sampler_equals_host_sampler(model):
    """The non-default decoding options run pickNextToken on the device by default (sampler_pick_kernel: no logits round
    trip per step); `device_sampler = 0` keeps the reference's structure (logits to the host, csrc/sampler.cpp).  Both see
    the same logits, so with temperature 0 the token streams are identical -- repetition penalty over the distinct ids,
    n-gram bans (n = 1 bans every generated id), their combination, natural EOS and a ragged batch.  With temperature > 0
    the uniform stream is the same counter-based splitmix64 on both sides and only logf may differ in the last ulp."""
    lib = model.lib
    clips = [synth.synth_waveform(0, 2.0), synth.synth_waveform(1, 0.7), synth.synth_waveform(2, 3.1)]

    def run(dev, **kw):
        assert lib.qasr_set_tuning(b"device_sampler", dev) == 0
        try:
            return model.transcribe_batch(clips, **kw)
        finally:
            assert lib.qasr_set_tuning(b"device_sampler", 1) == 0

    for opts in (dict(repetition_penalty=1.5), dict(no_repeat_ngram_size=1), dict(no_repeat_ngram_size=2),
                 dict(repetition_penalty=1.3, no_repeat_ngram_size=3), dict(repetition_penalty=0.5)):
        for ignore in (True, False):
            for mt in (1, 2, 19):
                a = run(1, max_tokens=mt, ignore_eos=ignore, **opts)
                b = run(0, max_tokens=mt, ignore_eos=ignore, **opts)
                assert a == b, (opts, ignore, mt)
                if ignore:
                    assert all(len(t) == mt for t in a)
    uni = run(1, max_tokens=19, ignore_eos=True, no_repeat_ngram_size=1)
    assert all(len(set(t)) == len(t) for t in uni)                     # n = 1: no id twice
    # temperature: same seed -> same stream on both sides; a differing pick is allowed only where logf's last ulp can decide
    same = total = 0
    for seed in (3, 4, 5):
        a = run(1, max_tokens=12, ignore_eos=True, temperature=0.8, seed=seed, repetition_penalty=1.1)
        b = run(0, max_tokens=12, ignore_eos=True, temperature=0.8, seed=seed, repetition_penalty=1.1)
        assert a == run(1, max_tokens=12, ignore_eos=True, temperature=0.8, seed=seed, repetition_penalty=1.1)
        for x, y in zip(a, b):
            n = next((i for i, (p, q) in enumerate(zip(x, y)) if p != q), len(x))   # streams may part after a flipped pick
            same += n
            total += len(x)
    assert same >= 0.9 * total, (same, total)


def test_streaming_asr_batched_equals_sequential(model):
    """StreamingASR mirror (qasr/streaming.py) on the engine: bursts of signal separated by silence, an energy VAD in
    place of Silero (a separate model, out of scope); all segments in one ragged batch == one transcribe per segment."""
    from qasr import streaming as S
    vocab, merges = _bpe_fixture()
    model.set_vocab({i: t for t, i in vocab.items()})
    model.set_merges(merges)
    parts = []
    for k, (sec, gap) in enumerate(((1.3, 0.5), (2.2, 0.4), (0.2, 0.6), (3.1, 0.3))):      # the 0.2 s blip is below minSpeechDuration
        parts += [synth.synth_waveform(k, sec), np.zeros(int(gap * 16000), np.float32)]
    audio = np.concatenate(parts)

    def energy_vad(chunk):
        return 0.9 if float(np.sqrt(np.mean(chunk * chunk))) > 0.05 else 0.02
    cfg = S.StreamingASRConfig(max_tokens=6, max_segment_duration=2.5, language="English")
    st = S.StreamingASR(model, energy_vad)
    seq = list(st.transcribe_stream(audio, config=cfg))
    bat = st.transcribe_stream_batched(audio, config=cfg)
    assert bat == seq
    assert len(seq) >= 4 and all(s.is_final for s in seq)                # 3 bursts, the 3.1 s one force-split at 2.5 s
    assert [s.segment_index for s in seq] == list(range(len(seq)))
    # (after a force-split the closing segment still reports the VAD segment's own start time, as in the reference)
    assert all(seq[i].end_time > seq[i - 1].end_time for i in range(1, len(seq)))


def test_staged_batches_equal_plain_batches(sd):
    """qasr_batch_stage / qasr_batch_begin_staged (the next batch's staging + H2D under the current batch's kernels) give the tokens of
    qasr_batch_begin for every batch of a sequence with changing sizes and clip lengths; staging before the current batch has started,
    or adopting without a staged batch, is refused; a plain qasr_batch_begin discards a staged batch."""
    m = Qwen3ASRModel.from_state_dict(sd, preset="tiny", max_batch=6, max_audio_seconds=4, max_new_tokens=12)
    try:
        batches = [[synth.synth_waveform(10 * i + k, 0.4 + 0.3 * ((i + k) % 5)) for k in range(1 + (2 * i) % 6)] for i in range(5)]
        want = [m.transcribe_batch(b, max_tokens=7, ignore_eos=True) for b in batches]
        got = []
        m.batch_begin(batches[0], max_tokens=7, ignore_eos=True)
        with pytest.raises(QasrError, match="qasr error 1"):
            m.batch_stage(batches[1])                       # the current batch's log-mel has not been queued yet
        for i in range(len(batches)):
            m.batch_run()
            if i + 1 < len(batches):
                m.batch_stage(batches[i + 1])
            toks, lens = m.batch_tokens()
            got.append([toks[b, :lens[b]].tolist() for b in range(len(batches[i]))])
            if i + 1 < len(batches):
                m.batch_begin_staged(max_tokens=7, ignore_eos=True)
        assert got == want
        with pytest.raises(QasrError, match="qasr error 1"):
            m.batch_begin_staged(max_tokens=7)              # nothing staged
        m.batch_run()
        m.batch_stage(batches[2])
        m.batch_tokens()
        m.batch_rewind()
        with pytest.raises(QasrError, match="qasr error 1"):
            m.batch_run()                                   # the staged batch's samples replaced this batch's in the device buffer
        assert m.transcribe_batch(batches[1], max_tokens=7, ignore_eos=True) == want[1]      # plain begin: the staged batch is dropped
        with pytest.raises(QasrError, match="qasr error 1"):
            m.batch_begin_staged(max_tokens=7)
    finally:
        m.close()
