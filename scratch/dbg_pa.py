import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'qwen3-asr-swift_amd'); sys.path.insert(0, 'tests')
import numpy as np, torch
from oracle import aligner as OA, config as OC, pipeline, precision as P
from qasr import synth, config as QC
from qasr.aligner import Qwen3ForcedAligner
sd = synth.synth_state_dict(QC.AUDIO_ALIGNER, QC.TEXT_SMALL, seed=0, init="hf", classify_num=5000)
m = Qwen3ForcedAligner.from_state_dict(sd, preset="aligner-0.6B", max_audio_seconds=30)
rng = np.random.default_rng(1)
ids, ts = [], []
for w in range(30):
    ts.append(len(ids)); ids.append(151705)
    ids += rng.integers(1000, 100000, size=1 + w % 3).tolist()
    ts.append(len(ids)); ids.append(151705)
pcm = synth.synth_waveform(1, 12.0)
raw, logits = m.align_raw(pcm, ids, ts, want_logits=True)
oracle = pipeline.OracleModel(sd, OC.AUDIO_ALIGNER, OC.TEXT_SMALL, OC.TOKENS, P.DEVICE)
with torch.no_grad():
    emb = oracle.encode(oracle.mel(pcm))
    ref = OA.classify_logits(emb, ids, ts, oracle.W, oracle.text_cfg, oracle.policy, oracle.tok).numpy()
tol = 4 * 2.0 ** -8 * float(np.abs(ref).max())
d = np.abs(logits - ref)
print("FORM", os.environ.get("QASR_PA_FORM"), "max", d.max(), "tol", tol, "relL2", np.linalg.norm(logits - ref) / np.linalg.norm(ref))
print("per-row max err:", np.round(d.max(1), 4).tolist())
print("prompt length", 15 + emb.shape[0] + len(ids), "slot positions", [15 + emb.shape[0] + t for t in ts][:10])
