"""Short-prompt prefill time of BASELINE config 2 (L_c = 24 conditioning positions + 1, 2 CFG rows): wall time of generate() at
two lengths, the intercept of the line through them = conditioning hand-over + prefill + first frame; and the prefill kernels alone
through zn_prefill between two device synchronisations.

    python3 tools/prefillbench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zonos_amd import synth  # noqa: E402
from zonos_amd.testing import build_model  # noqa: E402

model, _ = build_model(synth.FULL_CFG, 1234, "cuda:0")
eng = model.engine(1)
eng.call("zn_debug_eos_bias", float("-inf"))
for kv in os.environ.get("ZN_TUNE", "").split(","):
    if kv:
        eng.call("zn_debug_tune", int(kv.split(":")[0]), int(kv.split(":")[1]))
cond = synth.conditioning(1234, "cond", 2, 24, 2048).to("cuda:0")


def run(n):
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.generate(cond, max_new_tokens=n, sampling_params={"temperature": 0.0})
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


t1, t2 = run(9), run(137)
slope = (t2 - t1) / 128
print(f"generate(9 new tokens) {t1 * 1e3:.3f} ms, generate(137) {t2 * 1e3:.3f} ms: {slope * 1e3:.4f} ms per decode step, "
      f"intercept (prefill of 25 positions x 2 rows + first frame + host set-up) {(t1 - (9 + 7) * slope) * 1e3:.3f} ms")
