"""CPU-side checks of the C ABI: libqasr.so loads, exports every symbol include/qasr.h declares, the
ctypes structs match the header, and the GPU-free entry points (presets, frame counts) agree with the
reference's known answers.  No compute call is made here (there is no GPU and no CPU fallback)."""
import ctypes as C
import json
import os
import re
import pytest
from conftest import GOLDEN, ROOT
from qasr import _lib

HEADER = os.path.join(ROOT, "include", "qasr.h")
KAT = json.load(open(os.path.join(GOLDEN, "kat_reference_tests.json"), encoding="utf-8"))


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qasr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load(strict=True)
    declared = _declared_functions()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in qasr.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} lacks a ctypes signature"
    assert sorted(_lib.SIGNATURES) == declared


def test_config_struct_matches_header():
    text = open(HEADER).read()
    body = text[text.index("typedef struct qasr_config {"):text.index("} qasr_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for m in re.finditer(r"\b(int32_t|float)\s+([^;]+);", body):
        fields += [f.strip() for f in m.group(2).split(",")]
    assert [f for f, _ in _lib.QasrConfig._fields_] == fields


def _cfg(preset):
    lib = _lib.load()
    c = _lib.QasrConfig()
    assert lib.qasr_default_config(preset.encode(), C.byref(c)) == 0
    return c


def test_presets_match_reference_constants():
    k = KAT["config"]
    s = _cfg("0.6B")
    a, t = k["audio_small"], k["text_small"]
    assert (s.enc_d_model, s.enc_layers, s.enc_heads, s.enc_ffn, s.n_mels, s.enc_out_dim, s.conv_channels) == \
        (a["d_model"], a["layers"], a["heads"], a["ffn_dim"], a["n_mels"], a["output_dim"], a["conv_channels"])
    assert (s.hidden, s.dec_layers, s.heads, s.kv_heads, s.head_dim, s.inter, s.vocab, s.bits, s.group_size) == \
        (t["hidden"], t["layers"], t["heads"], t["kv_heads"], t["head_dim"], t["inter"], t["vocab"], t["bits"], t["group_size"])
    big = _cfg("1.7B")
    al, tl = k["audio_large"], k["text_large"]
    assert (big.enc_d_model, big.enc_layers, big.enc_heads, big.enc_ffn, big.enc_out_dim) == \
        (al["d_model"], al["layers"], al["heads"], al["ffn_dim"], al["output_dim"])
    assert (big.hidden, big.inter, big.dec_layers, big.heads, big.kv_heads, big.head_dim) == \
        (tl["hidden"], tl["inter"], tl["layers"], tl["heads"], tl["kv_heads"], tl["head_dim"])
    for model_id, size, bits in k["detect"]:
        c = _cfg(model_id)
        assert c.bits == bits
        assert (c.hidden == 2048) == (size == "large")
    p = KAT["prompt"]["full_no_context_A3"]
    assert [s.tok_im_start, s.tok_system, s.tok_newline, s.tok_im_end] == p[:4]
    assert (s.tok_user, s.tok_audio_start, s.tok_audio_pad, s.tok_audio_end, s.tok_assistant, s.tok_asr_text) == \
        (p[6], p[8], p[9], p[12], p[16], p[18])
    assert s.max_new_tokens == 448 and s.fft_scale == 2.0


@pytest.mark.parametrize("n,frames", [(80000, 500), (480000, 3000), (16000, 100), (400, 2), (1, 0), (160, 1)])
def test_frame_counts(n, frames):
    assert _lib.load().qasr_num_mel_frames(n) == frames


def test_error_paths_without_gpu():
    lib = _lib.load()
    assert lib.qasr_default_config(b"0.6B", None) != 0
    assert lib.qasr_input_sample_rate(None) == 16000
    assert lib.qasr_is_loaded(None) == 0
    assert lib.qasr_memory_footprint(None) == 0
    assert lib.qasr_transcribe(None, None, 0, 16000, None, None) != 0


# ---- pickNextToken through the C ABI (pure CPU function; reference KATs: Qwen3DecodingOptionsTests.swift:51-235)
import numpy as np  # noqa: E402


def _pick(logits, generated, repetition_penalty=1.0, no_repeat_ngram=0, temperature=0.0, rng=None):
    lib = _lib.load()
    lg = np.ascontiguousarray(logits, dtype=np.float32)
    gen = np.ascontiguousarray(generated, dtype=np.int32)
    st = C.c_uint64(rng if rng is not None else 1)
    tok = lib.qasr_pick_next_token(lg.ctypes.data_as(C.POINTER(C.c_float)), lg.shape[0],
                                   gen.ctypes.data_as(C.POINTER(C.c_int32)), gen.shape[0],
                                   repetition_penalty, no_repeat_ngram, temperature, C.byref(st))
    return tok, st.value


@pytest.mark.parametrize("case", KAT["sampler"], ids=lambda c: c["name"])
def test_sampler_kats(case):
    logits = np.full(case["n"], case["fill"], dtype=np.float32)
    for k, v in case["set"].items():
        logits[int(k)] = v
    for _ in range(3):
        assert _pick(logits, case["generated"], **case["opts"])[0] == case["expect"]


def test_sampler_matches_oracle_on_random_cases():
    from oracle import decoder
    rng = np.random.default_rng(0)
    for _ in range(200):
        n = int(rng.integers(4, 40))
        logits = rng.normal(size=n).astype(np.float32) * 3
        gen = rng.integers(0, n, size=int(rng.integers(0, 12))).tolist()
        rp = float(rng.choice([1.0, 1.1, 2.0]))
        ng = int(rng.choice([0, 2, 3]))
        assert _pick(logits, gen, rp, ng)[0] == decoder.pick_next_token(logits, gen, rp, ng)


def test_sampler_temperature_statistics():
    st = 7
    seen = set()
    for _ in range(50):
        t, st = _pick(np.zeros(16, np.float32), [], temperature=1.0, rng=st)
        seen.add(t)
    assert len(seen) >= 3
    peak = np.zeros(16, np.float32)
    peak[9] = 10.0
    hits, st = 0, 11
    for _ in range(50):
        t, st = _pick(peak, [], temperature=0.1, rng=st)
        hits += t == 9
    assert hits > 25
    assert _pick(peak, [], temperature=0.7, rng=5) == _pick(peak, [], temperature=0.7, rng=5)   # seeded => reproducible
