import sys, dataclasses
sys.path.insert(0, '.'); sys.path.insert(0, 'qwen3-asr-swift_amd'); sys.path.insert(0, 'tests')
import numpy as np, torch
from oracle import config as C, decoder, encoder, precision as P
from qasr import synth
import gpu_util
for layers in (0, 1, 2):
    cfg = dataclasses.replace(C.AUDIO_TINY, layers=layers)
    sd = synth.synth_state_dict(cfg, C.TEXT_TINY, seed=3, init="stress")
    e = gpu_util.Engine("tiny", max_audio_seconds=30, enc_layers=layers)
    e.load_state_dict(sd)
    for T in (7, 13, 16, 50, 100):
        g = torch.Generator().manual_seed(T)
        mel = (torch.randn(128, T, generator=g) * 0.5).numpy()
        got = e.encode(mel)
        W = decoder.Weights(sd)
        with torch.no_grad():
            dev = P.bf16_round(encoder.encode(mel, W, cfg, P.DEVICE)).numpy()
        d = np.abs(got - dev)
        print(layers, T, got.shape, "rel", np.linalg.norm(got-dev)/np.linalg.norm(dev), "max", d.max(), "rows bad", (d.max(1) > 1e-2).nonzero()[0][:10])
    e.close()
