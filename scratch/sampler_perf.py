#!/usr/bin/env python3
"""Non-default decoding options at the bench shape: pickNextToken on the device (default) against logits-to-host + csrc/sampler.cpp
(`device_sampler = 0`, the reference's own structure).  usage: python scratch/sampler_perf.py [--batch 32] [--seconds 30] [--tokens 64]"""
import argparse
import os
import sys
import time

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
    ap.add_argument("--tokens", type=int, default=64)
    a = ap.parse_args()
    sd = synth.synth_state_dict(QC.AUDIO_SMALL, QC.TEXT_SMALL, seed=0, init="hf")
    m = Qwen3ASRModel.from_state_dict(sd, preset="0.6B", max_batch=a.batch, max_audio_seconds=int(np.ceil(a.seconds)), max_new_tokens=448)
    clips = [synth.synth_waveform(k, a.seconds) for k in range(a.batch)]
    legs = [("greedy (fast path)", {}),
            ("repetition_penalty 1.3 + no_repeat_ngram 3", dict(repetition_penalty=1.3, no_repeat_ngram_size=3)),
            ("temperature 0.8", dict(temperature=0.8, seed=7))]
    print(f"B={a.batch} x {a.seconds:.0f} s, {a.tokens} tokens, EOS ignored; decode ms (device events) and wall ms of the whole pass")
    for name, opts in legs:
        for dev in ((1,) if not opts else (1, 0)):
            assert m.lib.qasr_set_tuning(b"device_sampler", dev) == 0
            res = []
            for r in range(3):
                m.batch_begin(clips, max_tokens=a.tokens, ignore_eos=True, **opts)
                t0 = time.perf_counter()
                m.batch_run()
                toks, lens = m.batch_tokens()
                wall = (time.perf_counter() - t0) * 1e3
                ms, steps = m.batch_timings()
                res.append((ms[3], wall))
            dec, wall = min(res)
            where = "" if not opts else (" | sampler on the device" if dev else " | sampler on the host")
            print(f"  {name}{where}: decode {dec:8.1f} ms, pass {wall:8.1f} ms")
    m.lib.qasr_set_tuning(b"device_sampler", 1)
    m.close()


if __name__ == "__main__":
    main()
