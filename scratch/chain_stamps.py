"""In-kernel phase stamps of the persistent layer launch (csrc/dec_chain.hip) inside a real decode step: qasr_kernel_probe 6 prints, for the
middle layer, when each phase of each workgroup was staged / summed / signalled and when each hand-off wait ended (stderr).
python scratch/chain_stamps.py [batch] [chain mode]"""
import ctypes as C
import sys

sys.path.insert(0, '.'); sys.path.insert(0, 'qwen3-asr-swift_amd'); sys.path.insert(0, 'tests')
from qasr import synth
import gpu_util
from oracle import config as OC   # geometry only (scratch perf script)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
MODE = int(sys.argv[2]) if len(sys.argv) > 2 else 3          # chain mode; 0 = stamps of the q|k|v + attention launch (qa) instead
KNOBS = dict(kv.split(":") for kv in sys.argv[3:])
sd = synth.synth_state_dict(OC.AUDIO_SMALL, OC.TEXT_SMALL, seed=0, init="hf")
e = gpu_util.Engine("0.6B", max_batch=B, max_audio_seconds=30, max_new_tokens=448)
e.load_state_dict(sd)
clips = [synth.synth_waveform(k, 30.0) for k in range(B)]
for k, v in KNOBS.items():
    e.set_tuning(k, int(v))
if MODE:
    e.set_tuning("chain", MODE)
else:
    e.set_tuning("qa", 1)
e.transcribe_batch(clips, max_tokens=4, ignore_eos=True)
m = C.c_float(); fl = C.c_double()
print(f"chain {MODE}, {B} rows, {KNOBS}", file=sys.stderr, flush=True)
e.check(e.lib.qasr_kernel_probe(e.h, 6 if MODE else 7, 1, C.byref(m), C.byref(fl)))
e.set_tuning("chain", 0)
e.set_tuning("qa", 0)
e.close()
