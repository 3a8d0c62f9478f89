# forced aligner latency at the full Qwen3-ForcedAligner-0.6B geometry (synthetic weights): the reference's
# testForcedAlignerLatency scenario (ForcedAlignerTests.swift:378-436) -- one utterance, repeated
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'qwen3-asr-swift_amd')
import numpy as np
from qasr import synth, config as QC
from qasr.aligner import Qwen3ForcedAligner
sd = synth.synth_state_dict(QC.AUDIO_ALIGNER, QC.TEXT_SMALL, seed=0, init="hf", classify_num=5000)
m = Qwen3ForcedAligner.from_state_dict(sd, preset="aligner-0.6B", max_audio_seconds=300)
rng = np.random.default_rng(1)
for sec, n_words in ((20.0, 11), (30.0, 80), (120.0, 300), (290.0, 700)):
    ids, ts = [], []
    for w in range(n_words):
        ts.append(len(ids)); ids.append(151705)
        ids += rng.integers(1000, 100000, size=1 + w % 3).tolist()
        ts.append(len(ids)); ids.append(151705)
    pcm = synth.synth_waveform(1, sec)
    m.align_raw(pcm, ids, ts)
    t0 = time.perf_counter()
    for _ in range(5):
        m.align_raw(pcm, ids, ts)
    dt = (time.perf_counter() - t0) / 5
    ms = (C_float5 := None)
    import ctypes as C
    st = (C.c_float * 5)(); ns = C.c_int32()
    m.lib.qasr_batch_timings(m.h, st, C.byref(ns))
    print(f"{sec:6.0f} s audio, {n_words:4d} words ({len(ids)} text ids): {dt*1e3:7.2f} ms per align (host wall incl. H2D/D2H)  "
          f"device: mel {st[0]:.2f} encoder {st[1]:.2f} decoder pass + head {st[2]:.2f} ms  -> {sec/dt:7.0f} audio-s/s", flush=True)
m.close()

# batched: 8 clips x 30 s, 80 words each, one device pass
m = Qwen3ForcedAligner.from_state_dict(sd, preset="aligner-0.6B", max_audio_seconds=30, max_batch=8, max_prompt_extra=512)
import json
from oracle import tokenizer as otok
b2u = otok.byte_to_unicode()
m.set_vocab({b: b2u[b] for b in range(256)})
m.set_merges("#version: 0.2\n")
words = ["alpha", "beta", "gamma", "delta"]
texts = [" ".join(words[(i + k) % 4][:3] for i in range(80)) for k in range(8)]
clips = [synth.synth_waveform(k, 30.0) for k in range(8)]
m.align_batch(clips, texts)
t0 = time.perf_counter()
for _ in range(5):
    res = m.align_batch(clips, texts)
dt = (time.perf_counter() - t0) / 5
print(f"batch of 8 x 30 s, 80 words each: {dt*1e3:.2f} ms per batch -> {8*30/dt:.0f} audio-s/s, words {len(res[0])}", flush=True)
t0 = time.perf_counter()
for _ in range(5):
    for c, t in zip(clips, texts):
        m.align(c, t)
dt1 = (time.perf_counter() - t0) / 5
print(f"same 8 clips one at a time: {dt1*1e3:.2f} ms -> {8*30/dt1:.0f} audio-s/s", flush=True)
m.close()
