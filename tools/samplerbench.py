"""ms per decode step at batch 1 for greedy and sampled (temperature 1.0, min_p 0.1) decoding, with the one-workgroup step tail
(sample1_kernel) and with the ticketed sampler (zn_debug_tune(16, 2)).
    python tools/samplerbench.py [new tokens=400]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zonos_amd import synth  # noqa: E402
from zonos_amd.testing import build_model  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
model, _ = build_model(synth.FULL_CFG, 1234, "cuda:0")
eng = model.engine(1)
eng.call("zn_debug_eos_bias", float("-inf"))
cond = synth.conditioning(1234, "cond", 2, 24, 2048).to("cuda:0")
for name, sp in (("greedy", {"temperature": 0.0}), ("sampled", {"temperature": 1.0, "min_p": 0.1})):
    for t16 in (1, 2):
        eng.call("zn_debug_tune", 16, t16)
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.generate(cond, max_new_tokens=n, sampling_params=sp, seed=4242)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print(f"{name:8s} {'one workgroup' if t16 == 1 else 'ticketed     '}: {best / (n + 7) * 1e3:.4f} ms per step (incl. prefill)", flush=True)
eng.call("zn_debug_tune", 16, 1)
