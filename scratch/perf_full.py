import sys, time, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'qwen3-asr-swift_amd'); sys.path.insert(0, 'tests')
import numpy as np, torch
from qasr import synth
import gpu_util
from oracle import config as OC   # geometry only (this is a scratch perf script, not the product)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NDEC = int(sys.argv[2]) if len(sys.argv) > 2 else 128
SEC = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
t0 = time.time()
sd = synth.synth_state_dict(OC.AUDIO_SMALL, OC.TEXT_SMALL, seed=0, init="hf")
print("weights", time.time() - t0)
e = gpu_util.Engine("0.6B", max_batch=B, max_audio_seconds=int(SEC + 0.999), max_new_tokens=448)
t0 = time.time(); e.load_state_dict(sd); print("upload+finalize", time.time() - t0)
clips = [synth.synth_waveform(k, SEC) for k in range(B)]
for it in range(3):
    t0 = time.time()
    out = e.transcribe_batch(clips, max_tokens=NDEC, ignore_eos=True)
    dt = time.time() - t0
    ms, steps = e.timings()
    print(f"iter {it}: wall {dt*1e3:.1f} ms  stages mel/enc/prefill/decode/total = {[round(x,2) for x in ms]} steps={steps}  audio-s/s={B*SEC/dt:.0f}")
print("tokens[0][:8]", out[0][:8], "lens", set(len(o) for o in out))
lib = e.lib
for which, name in ((0, "layer gemvs"), (1, "decode attn"), (2, "lm head")):
    ms = C.c_float(); by = C.c_double()
    e.check(lib.qasr_kernel_probe(e.h, which, 20, C.byref(ms), C.byref(by)))
    print(f"probe {name}: {ms.value*1e3:.1f} us  {by.value/1e6:.1f} MB  -> {by.value/ms.value/1e9:.2f} TB/s")
e.close()
