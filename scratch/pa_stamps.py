"""Prompt attention in isolation on the bench batch (qasr_kernel_probe 5): average launch time, causal TFLOP/s and -- with a `make DIAG=1`
library -- the per-wave phase sums of one stamped launch (stderr).  python scratch/pa_stamps.py [batch] [seconds]"""
import ctypes as C
import sys

sys.path.insert(0, '.'); sys.path.insert(0, 'qwen3-asr-swift_amd'); sys.path.insert(0, 'tests')
from qasr import synth
import gpu_util
from oracle import config as OC   # geometry only (scratch perf script)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
SEC = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
sd = synth.synth_state_dict(OC.AUDIO_SMALL, OC.TEXT_SMALL, seed=0, init="hf")
e = gpu_util.Engine("0.6B", max_batch=B, max_audio_seconds=int(SEC + 0.999), max_new_tokens=448)
e.load_state_dict(sd)
clips = [synth.synth_waveform(k, SEC) for k in range(B)]
e.transcribe_batch(clips, max_tokens=4, ignore_eos=True)
for knob, v in (("pa_vfrag", 1), ("pa_vfrag", 0), ("pa_vfrag", 1)):
    e.check(e.lib.qasr_set_tuning(knob.encode(), v))
    e.transcribe_batch(clips, max_tokens=4, ignore_eos=True)      # the V image the form reads is written by the prompt pass
    m = C.c_float(); fl = C.c_double()
    e.check(e.lib.qasr_kernel_probe(e.h, 5, 20, C.byref(m), C.byref(fl)))
    print(f"{knob}={v}: prompt attention {m.value * 1e3:.1f} us per launch, {fl.value / 1e9:.2f} GFLOP causal -> {fl.value / m.value / 1e9:.0f} TFLOP/s", flush=True)
if e.lib.qasr_set_tuning(b"pa_stamps", 1) == 0:
    m = C.c_float(); fl = C.c_double()
    e.check(e.lib.qasr_kernel_probe(e.h, 5, 5, C.byref(m), C.byref(fl)))
else:
    print("pa_stamps refused: not a DIAG build")
e.close()
