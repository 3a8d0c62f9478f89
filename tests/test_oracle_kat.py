"""Oracle vs the reference's own known-answer tests (integer / byte work: bit-exact)."""
import json
import os
import numpy as np
import pytest
from conftest import GOLDEN
from oracle import config as C, decoder, encoder, tokenizer

KAT = json.load(open(os.path.join(GOLDEN, "kat_reference_tests.json"), encoding="utf-8"))


def test_config_constants():
    k = KAT["config"]
    a, al, t, tl = C.AUDIO_SMALL, C.AUDIO_LARGE, C.TEXT_SMALL, C.TEXT_LARGE
    for key, v in k["audio_small"].items():
        assert getattr(a, key) == v, key
    for key, v in k["audio_large"].items():
        assert getattr(al, key) == v, key
    for key, v in k["text_small"].items():
        assert getattr(t, key) == v, key
    for key, v in k["text_large"].items():
        assert getattr(tl, key) == v, key
    assert C.TEXT_SMALL_8BIT.bits == 8 and C.TEXT_LARGE_8BIT.bits == 8
    for model_id, size, bits in k["detect"]:
        assert C.detect_size(model_id) == size
        assert C.detect_bits(model_id) == bits


def _token_map(case):
    b2u = tokenizer.byte_to_unicode()
    m = {}
    for k, bs in case.get("map_bytes", {}).items():
        m[int(k)] = "".join(b2u[b] for b in bs)
    for k, s in case.get("map_literal", {}).items():
        m[int(k)] = s
    return m


@pytest.mark.parametrize("case", KAT["tokenizer_decode"], ids=lambda c: c["name"])
def test_tokenizer_decode(case):
    got = tokenizer.decode(case["tokens"], _token_map(case))
    if "expect" in case:
        assert got == case["expect"]
    else:
        assert case["expect_contains"] in got


def test_strip_asr_prefix():
    assert tokenizer.strip_asr_prefix("language English<asr_text> hello there ") == "hello there"
    assert tokenizer.strip_asr_prefix("no marker") == "no marker"


def test_prompt_layout():
    p = KAT["prompt"]
    ids, a0 = decoder.build_prompt(3)
    assert ids == p["full_no_context_A3"] and a0 == 9
    assert ids[:5] == p["system_prefix_no_context"]
    ids_c, _ = decoder.build_prompt(3, context_ids=p["context_tokens"])
    assert ids_c[:8] == p["system_prefix_with_context"]
    ids_e, _ = decoder.build_prompt(3, context_ids=[])
    assert ids_e == ids                                     # empty context == nil
    assert len(decoder.build_prompt(p["audio_tokens_5s"])[0]) == p["len_5s"]
    assert len(decoder.build_prompt(p["audio_tokens_30s"])[0]) == p["len_30s"]
    lang = decoder.build_prompt(2, language_ids=[11, 12])[0]
    assert lang[-3:] == [11, 12, C.TOKENS.asr_text]


@pytest.mark.parametrize("case", KAT["sampler"], ids=lambda c: c["name"])
def test_sampler(case):
    logits = np.full(case["n"], case["fill"], dtype=np.float32)
    for k, v in case["set"].items():
        logits[int(k)] = v
    for _ in range(3):
        got = decoder.pick_next_token(logits, case["generated"], **case["opts"])
        assert got == case["expect"]


def test_sampler_temperature_statistics():
    rng = np.random.default_rng(0)
    seen = {decoder.pick_next_token(np.zeros(16, np.float32), [], temperature=1.0, rng=rng) for _ in range(50)}
    assert len(seen) >= 3
    peak = np.zeros(16, np.float32)
    peak[9] = 10.0
    hits = sum(decoder.pick_next_token(peak, [], temperature=0.1, rng=rng) == 9 for _ in range(50))
    assert hits > 25


def test_output_length_table():
    for T, exp in KAT["output_length"]["cases"]:
        assert encoder.get_output_length(T) == exp
        assert sum(encoder.tokens_for_chunk(c) for c in encoder.chunk_lengths(T, 100)) == exp


def test_window_lengths():
    a = C.AUDIO_SMALL
    assert encoder.window_lengths(3000, a) == [104, 104, 104, 78]
    assert encoder.window_lengths(2000, a) == [104, 104, 52]       # cu_seqlens [0,104,208,260]
    assert encoder.window_lengths(500, a) == [65]
    assert encoder.window_lengths(50, a) == [7]                    # single short chunk: no pad


def test_bpe_encode_restatement():
    """Qwen3Tokenizer.encode (Tokenizer.swift:195-289): whitespace starts the next word, byte-level mapping,
    lowest-rank pair merged everywhere, ids of unknown pieces dropped; no merges -> per-character lookup."""
    b2u = tokenizer.byte_to_unicode()
    G, TAB = b2u[0x20], b2u[0x09]                      # 'Ġ' and 'ĉ'
    vocab = {}
    for b in range(256):
        vocab.setdefault(b2u[b], len(vocab))
    lines = ["#version: 0.2", "l a", "la n", G + " lan", "g u", "gu a", G + "lan gua"]
    merges = "\n".join(lines) + "\n"
    for line in lines[1:]:
        a, b = line.split(" ")
        vocab.setdefault(a + b, len(vocab))
    ranks = tokenizer.parse_merges(merges)
    assert ranks == {"l a": 1, "la n": 2, G + " lan": 3, "g u": 4, "gu a": 5, G + "lan gua": 6}
    inv = {v: k for k, v in vocab.items()}
    assert [inv[i] for i in tokenizer.encode("a language", vocab, ranks)] == ["a", G + "langua", "g", "e"]
    assert "".join(inv[i] for i in tokenizer.encode("x\ty  z", vocab, ranks)) == "x" + TAB + "y" + G + G + "z"
    assert tokenizer.encode("ab", {"a": 1, "b": 2}, {}) == [1, 2]
