"""sha1 of the encoder's output (audio tokens) and of the first tokens of a small batch: to compare two builds bit for bit."""
import sys, hashlib
sys.path.insert(0, '.'); sys.path.insert(0, 'qwen3-asr-swift_amd'); sys.path.insert(0, 'tests')
import numpy as np
from qasr import synth
import gpu_util
from oracle import config as OC
sd = synth.synth_state_dict(OC.AUDIO_SMALL, OC.TEXT_SMALL, seed=0, init="hf")
e = gpu_util.Engine("0.6B", max_batch=4, max_audio_seconds=30, max_new_tokens=16)
e.load_state_dict(sd)
clips = [synth.synth_waveform(k, 30.0 - 3.7 * k) for k in range(4)]
mel = e.mel(clips[0])
enc = e.encode(mel)
print("encoder", enc.shape, hashlib.sha1(np.ascontiguousarray(enc).tobytes()).hexdigest())
toks = e.transcribe_batch(clips, max_tokens=12, ignore_eos=True)
print("tokens", hashlib.sha1(repr(toks).encode()).hexdigest())
e.close()
