import sys, time, threading
sys.path.insert(0, '.'); sys.path.insert(0, 'qwen3-asr-swift_amd')
import numpy as np, torch
from qasr import synth, config as QC
from qasr.model import Qwen3ASRModel
NE = int(sys.argv[1]) if len(sys.argv) > 1 else 2
BT = 32
sd = synth.synth_state_dict(QC.AUDIO_SMALL, QC.TEXT_SMALL, seed=0, init="hf")
B = BT // NE
engines = [Qwen3ASRModel.from_state_dict(sd, preset="0.6B", max_batch=B, max_audio_seconds=30, max_new_tokens=448) for _ in range(NE)]
for e in engines:
    e.lib.qasr_set_shared_device(e.h, 1)      # engines that run concurrently on one GPU keep to ordinary launches (include/qasr.h)
clips = [synth.synth_waveform(k, 30.0) for k in range(BT)]
for i, e in enumerate(engines):
    e.batch_begin(clips[i * B:(i + 1) * B], max_tokens=128, ignore_eos=True)
    e.batch_sync()
def run(e, out, i):
    e.batch_rewind(); e.batch_run(); out[i] = e.batch_tokens()
for it in range(4):
    out = [None] * NE
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(e, out, i)) for i, e in enumerate(engines)]
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print(f"engines={NE} x B={B}: wall {dt*1e3:.1f} ms -> {BT*30/dt:.0f} audio-s/s; stage ms {[ [round(x,1) for x in e.batch_timings()[0]] for e in engines]}")
for e in engines: e.close()
