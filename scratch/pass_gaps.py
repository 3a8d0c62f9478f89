"""Host time of each call of the pipelined serving loop (bench.py's headline leg): where do the ~1.4 ms per pass between the stage sum and the wall time go?"""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'qwen3-asr-swift_amd')
import numpy as np, torch
from qasr import synth, config as QC
from qasr.model import Qwen3ASRModel
B = 32
sd = synth.synth_state_dict(QC.AUDIO_SMALL, QC.TEXT_SMALL, seed=0, init="hf")
m = Qwen3ASRModel.from_state_dict(sd, preset="0.6B", max_batch=B, max_audio_seconds=30, max_new_tokens=448)
clips = [synth.synth_waveform(k, 30.0) for k in range(B)]
m.batch_begin(clips, max_tokens=128, ignore_eos=True); m.batch_run(); m.batch_stage(clips); m.batch_tokens()
for it in range(5):
    t = [time.perf_counter()]
    m.batch_begin_staged(max_tokens=128, ignore_eos=True); t.append(time.perf_counter())
    m.batch_run(); t.append(time.perf_counter())
    m.batch_stage(clips); t.append(time.perf_counter())
    toks, lens = m.batch_tokens(); t.append(time.perf_counter())
    ms, st = m.batch_timings()
    d = [round((b - a) * 1e3, 3) for a, b in zip(t, t[1:])]
    print(f"begin_staged {d[0]} ms | run (enqueue) {d[1]} | stage {d[2]} | tokens (wait) {d[3]} | wall {round((t[-1]-t[0])*1e3,3)} | device stages sum {round(sum(ms[:4]),3)} total {round(ms[4],3)}", flush=True)
m.close()
