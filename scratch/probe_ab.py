import sys, os, time, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'qwen3-asr-swift_amd'); sys.path.insert(0, 'tests')
import numpy as np, torch
from qasr import synth
import gpu_util
from oracle import config as OC
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
sd = synth.synth_state_dict(OC.AUDIO_SMALL, OC.TEXT_SMALL, seed=0, init="hf")
e = gpu_util.Engine("0.6B", max_batch=B, max_audio_seconds=30, max_new_tokens=448)
e.load_state_dict(sd)
clips = [synth.synth_waveform(k, 30.0) for k in range(B)]
out = e.transcribe_batch(clips, max_tokens=64, ignore_eos=True)
ms, steps = e.timings(); print("stages", [round(x, 2) for x in ms], steps)
for rnd in range(3):
    for which, name in ((0, "layer gemvs"), (1, "decode attn"), (2, "lm head")):
        m = C.c_float(); by = C.c_double()
        e.check(e.lib.qasr_kernel_probe(e.h, which, 50, C.byref(m), C.byref(by)))
        print(f"round {rnd} probe {name}: {m.value*1e3:.1f} us  {by.value/1e6:.1f} MB -> {by.value/m.value/1e9:.2f} TB/s")
e.close()
