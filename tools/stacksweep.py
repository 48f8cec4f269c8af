"""ms per decode step and the in-kernel timeline of block 13 of the whole-step kernel (workgroup 0's first communication wave).
    python tools/stacksweep.py [new_tokens] [audio prefix frames: the context the timeline is taken at]"""
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
P = int(sys.argv[2]) if len(sys.argv) > 2 else 0
prefix = torch.from_numpy(synth.randint(1234, "prefix", (1, 9, P), 1024)).to("cuda:0") if P > 0 else None
stamps = torch.zeros(52, 32, dtype=torch.int64, device="cuda:0")   # [0] streaming communication wave, [1] attention workgroup (first 8) + a compute wave (from 8)
eng.call("zn_debug_chain_stamps", stamps.data_ptr())
best = 1e9
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model.generate(cond, audio_prefix_codes=prefix, max_new_tokens=n, sampling_params={"temperature": 0.0})
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
at = stamps.cpu().numpy()[1]
st = stamps.cpu().numpy().reshape(-1)
names = ["block start", "poll starts", "a swept = op0 in",
         "o1 res", "o1 pub", "o1 swept", "o1 next", "o2 res", "o2 pub", "o2 swept", "o2 next", "f1 res", "f1 pub", "f1 -", "f1 next",
         "f2 res", "f2 pub", "f2 swept", "f2 next", "ip res", "ip pub"]
an = ["entry", "inputs in LDS", "scores", "maxima known", "P.V summed", "partials out", "published"]
print(f"contexts {24 + P + 1} .. {24 + P + n + 8}; attention workgroup 0, block 13 (us after its entry): " + " ".join(f"{nm}={(at[i] - at[0]) / 100.0:.2f}" for i, nm in enumerate(an) if at[i] > 0))
print(f"  (attention entry - streaming block start = {(at[0] - st[0]) / 100.0:.2f} us)")
cn = ["prefetch starts", "parked", "op0 in", "op0 done", "op1 in", "op1 done", "fc1 in", "fc1 done", "fc2 in", "fc2 done", "ip in", "ip done"]
print("compute wave 0 of streaming workgroup 0: " + " ".join(f"{nm}={(st[40 + i] - st[0]) / 100.0:.2f}" for i, nm in enumerate(cn)))
print(f"step kernel: {best * 1e3 / (n + 8):.4f} ms/step, checksum {int(out.sum())}; streaming workgroup 0, block 13: " + " ".join(f"{nm}={(st[i] - st[0]) / 100.0:.2f}" for i, nm in enumerate(names)))
