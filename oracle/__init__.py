"""CPU oracle for the Qwen3-ASR `transcribe()` hot path.

TEST INFRASTRUCTURE ONLY.  This package is a from-scratch CPU restatement of the
reference algorithm (ivan-digital/qwen3-asr-swift, `Sources/Qwen3ASR/*`), used as the
parity checker.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it.  The product path (`qwen3-asr-swift_amd/`) never imports,
links or executes anything under `oracle/`.

Pinning status (see DESIGN.md "Oracle"):
  * integer / byte work (prompt ids, tokenizer decode, sampler, output-length table):
    pinned by the reference's own unit-test KATs, copied as data into tests/golden/kat_*.json.
  * floating-point work (mel, encoder, decoder): the reference holds NO tensor-level golden
     and cannot execute on Linux.  The encoder/decoder restatement is
    pinned against the independently written `transformers.models.qwen3_asr` implementation
    on seeded random weights (tests/golden/make_hf_goldens.py); the mel restatement is
    pinned only by shape tests + a float64 re-derivation => "parity unpinned" for mel
    numerics at the Accelerate boundary (notably the vDSP 2x FFT scaling, see mel.py).
"""
