import sys, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'qwen3-asr-swift_amd'); sys.path.insert(0, 'tests')
from qasr import synth
import gpu_util
from oracle import config as OC
B = 32
sd = synth.synth_state_dict(OC.AUDIO_SMALL, OC.TEXT_SMALL, seed=0, init="hf")
e = gpu_util.Engine("0.6B", max_batch=B, max_audio_seconds=30, max_new_tokens=448)
e.load_state_dict(sd)
clips = [synth.synth_waveform(k, 30.0) for k in range(B)]
e.transcribe_batch(clips, max_tokens=4, ignore_eos=True)
ms, steps = e.timings(); print("stages", [round(x, 2) for x in ms])
for rnd in range(2):
    for which, name in ((3, "qkv gemm 12992x4096x1024"), (4, "gate/up swiglu 12992x6144x1024")):
        m = C.c_float(); by = C.c_double()
        e.check(e.lib.qasr_kernel_probe(e.h, which, 20, C.byref(m), C.byref(by)))
        print(f"round {rnd} {name}: {m.value*1e3:.1f} us -> {by.value/m.value/1e9:.0f} TFLOP/s")
e.close()
