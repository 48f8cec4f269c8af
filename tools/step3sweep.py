"""ms per decode step and the in-kernel timeline of block 13 of the three-role whole-step kernel (zn_step3_kernel.h): projection workgroup 0
(wave 0), bulk workgroup 0 (communication wave of row 0 and compute wave 0), attention workgroup 0.
    python tools/step3sweep.py [new_tokens]"""
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
eng.call("zn_debug_tune", 15, 4)
cond = synth.conditioning(1234, "cond", 2, 24, 2048).to("cuda:0")
stamps = torch.zeros(52, 32, dtype=torch.int64, device="cuda:0")
eng.call("zn_debug_chain_stamps", stamps.data_ptr())
best = 1e9
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model.generate(cond, max_new_tokens=n, sampling_params={"temperature": 0.0})
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
assert eng.lib.zn_decode_path_detail(eng.h) == 3, "the three-role kernel did not serve this run"
st = stamps.cpu().numpy().reshape(-1)
t0 = st[0]


def row(name, base, names):
    print(f"{name}: " + " ".join(f"{nm}={(st[base + i] - t0) / 100.0:.2f}" for i, nm in enumerate(names) if st[base + i] > 0))


print(f"three-role step kernel: {best * 1e3 / (n + 8):.4f} ms/step, checksum {int(out.sum())}; block 13, us after the projection workgroup's block start")
row("projection wg 0 wave 0", 0, ["block start", "poll a", "a in LDS", "op0 done", "y1 in LDS", "op1 done", "poll x2", "x2 swept", "LN(x2) in LDS", "op4 done"])
row("attention wg 0", 32, ["entry", "inputs in LDS", "scores", "pass 2", "reduced", "published"])
row("bulk wg 0 comm wave", 16, ["block start", "poll x1", "x1 swept", "fc1 in ready", "fc1 res", "m published", "fc2 res", "x2 published"])
row("bulk wg 0 compute wave 0", 40, ["prefetch starts", "parked", "fc1 in", "fc1 done", "fc2 in", "fc2 done", "heads in", "heads done"])
