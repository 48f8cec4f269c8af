"""A/B of the decode step at the Zonos-v0.1-transformer dimensions, batch 1: whole-step kernel (zn_step_kernel.h) vs one chain launch
per block vs per-op launches; codes compared.
    python tools/stackbench.py [new_tokens]"""
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
outs = {}
for name, t8, t15 in (("chain", 1, 2), ("stack", 1, 1), ("chain", 1, 2), ("stack", 1, 1), ("launches", 2, 1)):
    eng.call("zn_debug_tune", 8, t8)
    eng.call("zn_debug_tune", 15, t15)
    model.generate(cond, max_new_tokens=32, sampling_params={"temperature": 0.0})
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model.generate(cond, max_new_tokens=n, sampling_params={"temperature": 0.0})
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    outs.setdefault(name, out)
    print(f"{name:9s} (path {eng.lib.zn_decode_path_detail(eng.h)}): {dt * 1e3 / (n + 8):.4f} ms per decode step ({n + 8} steps + prefill in {dt:.3f} s) = {n / 86.1328 / dt:.2f}x real-time AR only", flush=True)
    if name == "stack":
        same = torch.equal(outs["chain"], out)
        print(f"  codes identical to the chain path: {same}", flush=True)
        if not same:
            d = (outs["chain"] != out).any(dim=1)[0].nonzero()
            print(f"  first differing frame: {int(d[0]) if len(d) else -1} of {out.shape[-1]}", flush=True)
print("launches == chain:", torch.equal(outs["launches"], outs["chain"]))
