"""Where do the chain kernel and the launches path first differ at full dims?  (debug aid)
    python tools/chaindiff.py [new_tokens] [l_c]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zonos_amd import synth  # noqa: E402
from zonos_amd.testing import build_model  # noqa: E402

CFG = synth.CHAIN_CFG if os.environ.get("ZN_DIFF_CFG") == "chain512" else synth.FULL_CFG
SEED = 55 if CFG is synth.CHAIN_CFG else 1234
model, _ = build_model(CFG, SEED, "cuda:0")
eng = model.engine(1)
eng.call("zn_debug_eos_bias", float("-inf"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lc = int(sys.argv[2]) if len(sys.argv) > 2 else 24
cond = synth.conditioning(SEED, "cond", 2, lc, CFG["d_model"]).to("cuda:0")
NL = CFG["n_layer"]
trace = torch.zeros(NL, 8, 2, CFG["d_model"], dtype=torch.bfloat16, device="cuda:0")
eng.call("zn_debug_trace", trace.data_ptr())


STACK = os.environ.get("ZN_DIFF_STACK") == "1"        # compare the whole-step kernel with the chain path instead


def run(chain):
    if STACK:
        eng.call("zn_debug_tune", 15, 3 if chain else 1)
    else:
        eng.call("zn_debug_tune", 8, 1 if chain else 2)
    eng.call("zn_debug_tune", 6, 1)
    tr = {"logits": [], "traces": []}
    tr["after_step"] = lambda step_idx, delayed, col: tr["traces"].append(trace.clone()) if step_idx >= 0 else None
    out = model.generate(cond, max_new_tokens=n, sampling_params={"temperature": 0.0}, _trace=tr)
    return out.cpu(), torch.stack(tr["logits"]).cpu(), tr["traces"]


a, la, ta = run(False)
b, lb, tb = run(True)
neq = (la.view(torch.int32) != lb.view(torch.int32)).flatten(1).sum(1)
first = int((neq > 0).nonzero()[0]) if bool((neq > 0).any()) else -1
print(f"codes equal {torch.equal(a, b)}; first differing call {first} (call k = decode step k-1)")
for k in range(len(ta)):
    d = (ta[k].view(torch.int16) != tb[k].view(torch.int16))
    if bool(d.any()):
        print(f"decode step {k} (call {k + 1}): first differing trace entries:")
        for li in range(NL):
            for which, nm in (((2, "q"), (1, "attention out"), (0, "x after block")) if STACK else ((2, "q"), (1, "attention out"), (7, "x after attention half"), (3, "m"), (0, "x after block"))):
                dd = d[li, 3:7].reshape(2, -1) if which == 3 else d[li, which]
                if bool(dd.any()):
                    idx = dd.nonzero()
                    r, f = int(idx[0][0]), int(idx[0][1])
                    print(f"  layer {li} {nm}: {int(dd.sum())} elements differ; first at row {r} feature {f}: launches {float((ta[k][li, 3:7].reshape(2, -1) if which == 3 else ta[k][li, which])[r, f]):.6g} chain {float((tb[k][li, 3:7].reshape(2, -1) if which == 3 else tb[k][li, which])[r, f]):.6g}")
                    if int(dd.sum()) < 12:
                        print("     all:", [(int(i[0]), int(i[1])) for i in idx])
            if not STACK and bool(d[li].any()) and li >= 2 + min(l for l in range(NL) if bool(d[l].any())):
                break
        if not STACK or k >= 2:
            break
