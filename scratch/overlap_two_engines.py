"""Does one MI355X run two bench passes faster side by side than one after the other?  Two engines on device 0 (own streams, own
buffers), one host thread each, every thread runs K passes of b clips x 30 s; compared with one engine running 2K passes.  The decode
stage is launch-latency-bound, the encoder / prompt pass MFMA-bound: whatever the hardware overlaps across the two
streams is throughput the single-stream pass leaves on the table.
  python scratch/overlap_two_engines.py [--batch 32] [--steps 6] [--engines 2]"""
import argparse
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "qwen3-asr-swift_amd"))
from qasr import config as QC, synth  # noqa: E402
from qasr.model import Qwen3ASRModel  # noqa: E402


PIPELINED = False


def passes(model, clips, n_dec, k):
    staged = False
    for _ in range(k):
        if staged:
            model.batch_begin_staged(max_tokens=n_dec, ignore_eos=True)
        else:
            model.batch_begin(clips, max_tokens=n_dec, ignore_eos=True)
        model.batch_run()
        if PIPELINED:
            model.batch_stage(clips)
            staged = True
        toks, lens = model.batch_tokens()
    assert (lens == n_dec).all()
    return toks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--engines", type=int, default=2)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--decode-tokens", type=int, default=128)
    ap.add_argument("--pipelined", action="store_true", help="each engine stages its next pass's PCM under its current pass (qasr_batch_stage)")
    a = ap.parse_args()
    global PIPELINED
    PIPELINED = a.pipelined
    torch.cuda.set_device(0)
    sd = synth.synth_state_dict(QC.AUDIO_SMALL, QC.TEXT_SMALL, seed=0, init="hf")
    cap = dict(device=0, max_batch=a.batch, max_audio_seconds=int(np.ceil(a.seconds)), max_new_tokens=448)
    engines = [Qwen3ASRModel.from_state_dict(sd, preset="0.6B", bits=16, **cap) for _ in range(a.engines)]
    for e in engines:
        e.lib.qasr_set_shared_device(e.h, 1)  # engines that run concurrently on one GPU keep to ordinary launches (include/qasr.h)
    clips = [synth.synth_waveform(k, a.seconds) for k in range(a.batch)]
    ref = passes(engines[0], clips, a.decode_tokens, 2)
    for e in engines[1:]:
        t = passes(e, clips, a.decode_tokens, 2)
        assert np.array_equal(t, ref)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    passes(engines[0], clips, a.decode_tokens, a.steps * a.engines)
    torch.cuda.synchronize()
    serial = time.perf_counter() - t0
    out = [None] * a.engines

    def work(i):
        out[i] = passes(engines[i], clips, a.decode_tokens, a.steps)

    th = [threading.Thread(target=work, args=(i,)) for i in range(a.engines)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    side = time.perf_counter() - t0
    for o in out:
        assert np.array_equal(o, ref)
    n = a.steps * a.engines
    audio = n * a.batch * a.seconds
    print(f"b={a.batch} x {n} passes: one engine {serial / n * 1e3:.2f} ms/pass = {audio / serial:.0f} audio-s/s; "
          f"{a.engines} engines side by side {side / n * 1e3:.2f} ms/pass = {audio / side:.0f} audio-s/s ({serial / side:.3f} x)", flush=True)


if __name__ == "__main__":
    main()
