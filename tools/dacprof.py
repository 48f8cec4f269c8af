"""One DAC decode of S seconds under a profiler (rocprofv3 --kernel-trace): 2 warm-up decodes, then REP timed ones.
    python3 tools/dacprof.py [seconds=10] [batch=1] [rep=3]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zonos_amd import synth  # noqa: E402
from zonos_amd.autoencoder import DACAutoencoder  # noqa: E402

S = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
REP = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dac = DACAutoencoder(synth.dac_state_dict(4321), device="cuda:0")
frames = int(round(S * 44100 / 512))
codes = torch.from_numpy(synth.randint(5, "codes.prof", (B, 9, frames), 1024)).to("cuda:0")
for _ in range(2 + REP):
    dac.decode(codes)
torch.cuda.synchronize()
