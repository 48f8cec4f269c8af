"""Per-step logits of the whole-step kernel vs the chain path (teacher-free greedy; stops at the first differing step)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zonos_amd import synth  # noqa: E402
from zonos_amd.testing import build_model  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
model, _ = build_model(synth.FULL_CFG, 1234, "cuda:0")
eng = model.engine(1)
eng.call("zn_debug_eos_bias", float("-inf"))
cond = synth.conditioning(1234, "cond", 2, int(os.environ.get("ZN_LC", "24")), 2048).to("cuda:0")
res = {}
for name, t15 in (("chain", 2), ("stack", 1), ("stack2", 1), ("threerole", 4)):
    eng.call("zn_debug_tune", 15, t15)
    tr = {"logits": []}
    out = model.generate(cond, max_new_tokens=n, sampling_params={"temperature": 0.0}, _trace=tr)
    torch.cuda.synchronize()
    res[name] = (out.cpu(), [l.float().cpu() for l in tr["logits"]])
a, b = res["chain"][1], res["stack"][1]
for k in range(min(len(a), len(b))):
    fin = torch.isfinite(a[k])
    d = torch.where(fin, (a[k] - b[k]).abs(), torch.zeros_like(a[k]))
    if d.max().item() != 0:
        print(f"first differing call {k}: logits bit-equal {(d == 0).float().mean().item():.5f}, max|d| {d.max().item():.4g}")
        break
print("codes equal:", torch.equal(res["chain"][0], res["stack"][0]))
print("stack runs equal:", all(torch.equal(x, y) for x, y in zip(res["stack"][1], res["stack2"][1])))
print("three-role kernel equal to the chain path:", torch.equal(res["chain"][0], res["threerole"][0]) and all(torch.equal(x, y) for x, y in zip(res["chain"][1], res["threerole"][1])))
