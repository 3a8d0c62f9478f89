"""Passes in flight on one GPU (qasr_dp_submit / qasr_dp_collect): bench.py's side leg at several lane counts, same workload.
  python scratch/lanes_sweep.py [--lanes 1,2,3,4] [--steps 12]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from qasr import config as QC, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lanes", default="1,2,3,4")
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--seconds", type=float, default=30.0)
ap.add_argument("--bits", type=int, default=16, choices=[16, 8, 4])
a = ap.parse_args()
torch.cuda.set_device(0)
sd = synth.synth_state_dict(QC.AUDIO_SMALL, QC.TEXT_SMALL, seed=0, init="hf")
if a.bits != 16:
    sd = synth.quantize_state_dict(sd, a.bits)
clips = [synth.synth_waveform(k, a.seconds) for k in range(a.batch)]
cap = dict(device=0, max_batch=a.batch, max_audio_seconds=int(np.ceil(a.seconds)), max_new_tokens=448, bits=a.bits)
for n in [int(x) for x in a.lanes.split(",")]:
    r = bench.lanes_leg(sd, clips, 128, a.steps, a.seconds, cap, n)
    print(f"b={a.batch} bits={a.bits} lanes={n}: {r['ms_per_step']:.2f} ms per pass = {r['value']:.0f} audio-s/s ({a.steps} passes, fill and drain included)", flush=True)
