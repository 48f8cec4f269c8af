"""A/B of the decode step at the Zonos-v0.1-transformer dimensions, batch 1: persistent chain kernel vs per-op launches.
    python tools/chainbench.py [new_tokens]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zonos_amd import synth  # noqa: E402
from zonos_amd.testing import build_model  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 861
model, _ = build_model(synth.FULL_CFG, 1234, "cuda:0")
eng = model.engine(1)
eng.call("zn_debug_eos_bias", float("-inf"))
cond = synth.conditioning(1234, "cond", 2, 24, 2048).to("cuda:0")
outs = {}
for name, v in (("launches", 2), ("chain", 1), ("launches", 2), ("chain", 1)):
    eng.call("zn_debug_tune", 8, v)
    model.generate(cond, max_new_tokens=32, sampling_params={"temperature": 0.0})
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model.generate(cond, max_new_tokens=n, sampling_params={"temperature": 0.0})
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    outs.setdefault(name, out)
    print(f"{name:9s}: {dt * 1e3 / (n + 8):.4f} ms per decode step ({n + 8} steps + prefill in {dt:.3f} s) = {n / 86.1328 / dt:.2f}x real-time AR only, "
          f"path {eng.lib.zn_decode_path(eng.h)}", flush=True)
print("codes identical:", torch.equal(outs["launches"], outs["chain"]))

# timeline of workgroup 0's communication wave, one mid-stack layer of the last decode step
stamps = torch.zeros(52, 32, dtype=torch.int64, device="cuda:0")
eng.call("zn_debug_tune", 8, 1)
eng.call("zn_debug_chain_stamps", stamps.data_ptr())
model.generate(cond, max_new_tokens=64, sampling_params={"temperature": 0.0})
torch.cuda.synchronize()
eng.call("zn_debug_chain_stamps", None)
st = stamps.cpu().numpy()
names = ["input ready"]
for op in ("out1", "out2", "fc1", "fc2"):
    names += [f"{op} results", f"{op} arrived", f"{op} all arrived", f"{op} next input ready"]
names += ["in_proj results", "end"]
for li in (12, 13):
    t0 = st[li][0]
    print(f"layer {li} timeline (us after 'input ready'): " + " | ".join(f"{n} {(st[li][i] - t0) / 100.0:.2f}" for i, n in enumerate(names)))
print(f"gap between chain launches 12 -> 13 (end -> input ready, attention in between): {(st[13][0] - st[12][len(names) - 1]) / 100.0:.2f} us")
