import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'qwen3-asr-swift_amd'); sys.path.insert(0, 'tests')
import numpy as np
from qasr import synth, config as QC
from qasr.aligner import Qwen3ForcedAligner, find_trailing_plateau_start
from test_gpu_aligner import _bpe_fixture
vocab, merges = _bpe_fixture()
for seed in range(8):
    sd = synth.synth_state_dict(QC.AUDIO_TINY, QC.TEXT_TINY, seed=seed, init="stress", classify_num=40)
    m = Qwen3ForcedAligner.from_state_dict(sd, preset="tiny-aligner", max_audio_seconds=300, max_prompt_extra=512)
    m.set_vocab({i: t for t, i in vocab.items()}); m.set_merges(merges)
    for tseed in range(3):
        rng = np.random.default_rng(tseed)
        text = " ".join(rng.choice(["a", "language", "English", "lists", "lan", "gua"], size=60).tolist())
        pcm = synth.synth_waveform(tseed, 250.0)
        one = m.align(pcm, text)
        p = find_trailing_plateau_start([w.start_time for w in one], 0.1, 5)
        got = m.align_long(pcm, text)
        print(f"seed {seed} text {tseed}: single-pass words {len(one)} plateau_start {p} raw[:12] {m.last_raw_indices[:12]} -> long passes {m.last_passes} words {len(got)}", flush=True)
    m.close()
