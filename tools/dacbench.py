"""DAC codec timings on the GPU box: encode and decode of S seconds of 44.1 kHz audio (synthetic weights).
    python tools/dacbench.py [seconds=10] [batch=1]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zonos_amd import synth  # noqa: E402
from zonos_amd.autoencoder import DACAutoencoder  # noqa: E402

S = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dac = DACAutoencoder(synth.dac_state_dict(4321), device="cuda:0")
frames = int(round(S * 44100 / 512))
wav = synth.test_waveform(1, "bench", frames * 512, batch=B).to("cuda:0")
for name, fn, arg in (("encode", dac.encode, wav), ("decode", dac.decode, None)):
    if arg is None:
        arg = codes
    out = fn(arg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        out = fn(arg)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    if name == "encode":
        codes = out
        gflop = 2 * 0.300e3 * S / 10.0 * B      # ~300 GMAC per 10 s
    else:
        gflop = 1.608 * frames * B
    print(f"{name}: {B} x {S:.1f} s ({frames} frames): {dt * 1e3:.2f} ms = {B * S / dt:.0f}x real-time, ~{gflop / dt / 1e3:.1f} TFLOP/s fp32", flush=True)
