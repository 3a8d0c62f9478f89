"""How long does the fused q|k|v + attention launch take when its K / V rows and weights come from the Infinity Cache instead of HBM?
Probe 1 walks the decoder's layers: with 2 layers (2 x 70 MB) everything stays in the 256 MiB cache, with 28 it comes from HBM as in the step."""
import sys, dataclasses
sys.path.insert(0, '.'); sys.path.insert(0, 'qwen3-asr-swift_amd'); sys.path.insert(0, 'tests')
from qasr import synth, config as QC
from qasr.model import Qwen3ASRModel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
for layers in (2, 3, 28):
    t = dataclasses.replace(QC.TEXT_SMALL, layers=layers)
    sd = synth.synth_state_dict(dataclasses.replace(QC.AUDIO_SMALL, layers=2), t, seed=0, init="hf")
    m = Qwen3ASRModel.from_state_dict(sd, preset="0.6B", max_batch=B, max_audio_seconds=30, max_new_tokens=448, enc_layers=2, dec_layers=layers)
    clips = [synth.synth_waveform(k, 30.0) for k in range(B)]
    m.batch_begin(clips, max_tokens=64, ignore_eos=True); m.batch_sync(); m.batch_run(); m.batch_tokens()
    for which, name in ((1, "qa"), (0, "o+gu+down")):
        ms, by = m.kernel_probe(which, 40)
        print(f"B={B} dec_layers={layers:2d} {name:10s}: {ms*1e3:7.2f} us per launch(group), {by/1e6:6.1f} MB -> {by/ms/1e6:6.0f} GB/s", flush=True)
    m.close()
