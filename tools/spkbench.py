"""Speaker-embedding network timing on the GPU box: python tools/spkbench.py [seconds=10]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zonos_amd import synth  # noqa: E402
from zonos_amd.speaker_cloning import SpeakerEmbeddingLDA  # noqa: E402

S = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
sd, lda = synth.speaker_state_dict(2468)
m = SpeakerEmbeddingLDA(sd, lda, device="cuda:0")
T = int(S * 100) + 1
feats = synth.speaker_features(1, "bench", 1, 80, T).to("cuda:0")
m.model.embed_features(feats, with_lda=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    m.model.embed_features(feats, with_lda=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
gmac = 2.9e9 * (20 + 40 + 128 + 6) * (T / 1000.0) / 1e9
print(f"speaker embedding of {S:.0f} s ({T} frames): {dt * 1e3:.1f} ms (~{2 * gmac / dt / 1e3:.1f} TFLOP/s fp32)")
