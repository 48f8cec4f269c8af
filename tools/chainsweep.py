"""Decode-step time and chain timeline for one library build (ZONOS_HIP_LIB selects it): python tools/chainsweep.py [tokens]"""
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
if os.environ.get("ZN_TUNE5"):
    eng.call("zn_debug_tune", 5, int(os.environ["ZN_TUNE5"]))
for kv in os.environ.get("ZN_TUNE", "").split(","):          # e.g. ZN_TUNE=12:2,13:3
    if kv:
        eng.call("zn_debug_tune", int(kv.split(":")[0]), int(kv.split(":")[1]))
cond = synth.conditioning(1234, "cond", 2, 24, 2048).to("cuda:0")
model.generate(cond, max_new_tokens=32, sampling_params={"temperature": 0.0})
best = 1e9
for _ in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model.generate(cond, max_new_tokens=n, sampling_params={"temperature": 0.0})
    torch.cuda.synchronize()
    best = min(best, time.perf_counter() - t0)
stamps = torch.zeros(52, 32, dtype=torch.int64, device="cuda:0")
eng.call("zn_debug_chain_stamps", stamps.data_ptr())
model.generate(cond, max_new_tokens=64, sampling_params={"temperature": 0.0})
torch.cuda.synchronize()
eng.call("zn_debug_chain_stamps", None)
st = stamps.cpu().numpy()
names = ["in"] + [f"{op}{k}" for op in ("o1", "o2", "f1", "f2") for k in (" res", " pub", " swept", " next")] + ["ip res", "end"]
li = 12
tl = " ".join(f"{nm}={(st[li][i] - st[li][0]) / 100.0:.2f}" for i, nm in enumerate(names))
print(f"{os.environ.get('ZONOS_HIP_LIB', 'default').split('/')[-1]} tune5={os.environ.get('ZN_TUNE5')} tune={os.environ.get('ZN_TUNE')}: {best * 1e3 / (n + 8):.4f} ms/step, checksum {int(out.sum())}; chain {((st[li][len(names) - 1] - st[li][0]) / 100.0):.2f} us, "
      f"attention(13) start +{(st[26 + 13][0] - st[12][len(names) - 1]) / 100.0:.2f} after chain(12) end: " + " ".join(f"{nm}={(st[26 + 13][i] - st[26 + 13][0]) / 100.0:.2f}" for i, nm in enumerate(["start", "len", "sc own", "sc all", "pv", "dpp+lds", "lsum", "barrier", "summed", "store issued"])) + f"; chain(13) in +{(st[13][0] - st[26 + 13][9]) / 100.0:.2f} after | "
      f"gap {(st[13][0] - st[12][len(names) - 1]) / 100.0:.2f} us | sweep passes y1/x1/x2 {st[li][24]}/{st[li][25]}/{st[li][27]} | {tl}", flush=True)
