"""Soak of the in-launch hand-offs: many passes at changing batch sizes on one engine; every pass's tokens must equal the first pass of that size, no error."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'qwen3-asr-swift_amd')
import numpy as np
from qasr import synth, config as QC
from qasr.model import Qwen3ASRModel
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
BITS = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sd = synth.synth_state_dict(QC.AUDIO_SMALL, QC.TEXT_SMALL, seed=0, init="hf")
if BITS != 16:
    sd = synth.quantize_state_dict(sd, BITS)
m = Qwen3ASRModel.from_state_dict(sd, preset="0.6B", max_batch=32, max_audio_seconds=30, max_new_tokens=448, **({"bits": BITS} if BITS != 16 else {}))
clips = [synth.synth_waveform(k, 30.0 - 0.5 * (k % 7)) for k in range(32)]
sizes = [32, 1, 8, 32, 3, 16, 32, 5, 17, 32]
ref = {}
t0 = time.time()
for it in range(N):
    b = sizes[it % len(sizes)]
    m.batch_begin(clips[:b], max_tokens=96, ignore_eos=True); m.batch_run()
    toks, lens = m.batch_tokens()
    key = b
    h = hash(toks[:b].tobytes())
    if key not in ref: ref[key] = h
    assert ref[key] == h, (it, b)
print(f"soak ok ({BITS} bit): {N} passes, sizes {sorted(ref)} in {time.time() - t0:.1f} s")
m.close()
