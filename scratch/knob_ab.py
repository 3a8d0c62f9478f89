#!/usr/bin/env python3
"""In-process A/B of tuning knobs on the bench workload (one engine, one resident batch, variants interleaved round by
round; median and min of the stage times over the rounds -- cdna_hip_programming.md rule 24).

usage: python scratch/knob_ab.py [--batch 32] [--seconds 30] [--tokens 128] [--rounds 5] [--bits 16] [--preset 0.6B] name=k1:v1,k2:v2 ...
   e.g. python scratch/knob_ab.py base= nt=gemv_nt:1 dant=da_nt:1 both=gemv_nt:1,da_nt:1 pf=kv_prefetch:1
"""
import argparse
import os
import statistics as st
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "qwen3-asr-swift_amd"))
import numpy as np                      # noqa: E402
from qasr import synth, config as QC    # noqa: E402
from qasr.model import Qwen3ASRModel    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--tokens", type=int, default=128)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--bits", type=int, default=16, choices=[16, 8, 4], help="16 = bf16 decoder, 4 / 8 = MLX-quantised, packed")
    ap.add_argument("--preset", default="0.6B", choices=["0.6B", "1.7B"])
    ap.add_argument("--allow-token-drift", action="store_true", help="variants that are not bit-identical by construction (another summation order)")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    variants = []
    for v in a.variants:
        name, _, spec = v.partition("=")
        knobs = {}
        for kv in filter(None, spec.split(",")):
            k, _, val = kv.partition(":")
            knobs[k] = int(val)
        variants.append((name, knobs))
    large = a.preset == "1.7B"
    sd = synth.synth_state_dict(QC.AUDIO_LARGE if large else QC.AUDIO_SMALL, QC.TEXT_LARGE if large else QC.TEXT_SMALL, seed=0, init="hf")
    if a.bits != 16:
        sd = synth.quantize_state_dict(sd, a.bits)
    m = Qwen3ASRModel.from_state_dict(sd, preset=a.preset, bits=a.bits, max_batch=a.batch, max_audio_seconds=int(np.ceil(a.seconds)),
                                      max_new_tokens=448)
    clips = [synth.synth_waveform(k, a.seconds) for k in range(a.batch)]
    m.batch_begin(clips, max_tokens=a.tokens, ignore_eos=True)
    all_knobs = sorted({k for _, kn in variants for k in kn})
    defaults = {}
    import ctypes as C
    for k in all_knobs:
        v = C.c_int()
        assert m.lib.qasr_get_tuning(k.encode(), C.byref(v)) == 0, k
        defaults[k] = v.value
    res = {name: [] for name, _ in variants}
    ref_tokens = None
    for r in range(a.rounds + 1):                       # round 0 = warm-up (graph capture per variant)
        for name, knobs in variants:
            for k in all_knobs:
                assert m.lib.qasr_set_tuning(k.encode(), knobs.get(k, defaults[k])) == 0
            m.batch_rewind()
            m.batch_run()
            toks, lens = m.batch_tokens()
            if ref_tokens is None:
                ref_tokens = toks.copy()
            assert a.allow_token_drift or np.array_equal(toks, ref_tokens), f"variant {name} changed the tokens"
            ms, steps = m.batch_timings()
            if r > 0:
                res[name].append(ms)
    print(f"B={a.batch} x {a.seconds:.0f} s, {a.tokens} tokens, {a.rounds} rounds; ms median (min): mel | encoder | prompt | decode | total")
    for name, knobs in variants:
        cols = list(zip(*res[name]))
        print(f"  {name:10s} {knobs}: " + " | ".join(f"{st.median(c):8.3f} ({min(c):8.3f})" for c in cols))
    m.close()


if __name__ == "__main__":
    main()
