"""Diagnostic: multi-step graphs vs single-step launches, with the one-workgroup sampler and with the ticketed one."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zonos_amd import synth  # noqa: E402
from zonos_amd.testing import build_model  # noqa: E402

model, _ = build_model(synth.FULL_CFG, 1234, "cuda:0")
eng = model.engine(1)
cond = synth.conditioning(1234, "cond", 2, 24, 2048).to("cuda:0")
eng.call("zn_debug_eos_bias", float("-inf"))
G = {"temperature": 0.0}
res = {}
for t16 in (1, 2):
    for t6 in (2, 1):
        eng.call("zn_debug_tune", 16, t16)
        eng.call("zn_debug_tune", 6, t6)
        for rep in range(2):
            o = model.generate(cond, max_new_tokens=861, sampling_params=G).cpu()
            res[(t16, t6, rep)] = o
            print(f"sampler {'one-wg' if t16 == 1 else 'ticketed'} graph-steps {'8' if t6 == 2 else '1'} rep {rep}: checksum {int(o.sum())}", flush=True)
ref = res[(2, 2, 0)]
for k, o in res.items():
    if not torch.equal(o, ref):
        d = (o != ref).any(1)[0].nonzero()
        print(k, "differs from ticketed/8-step first at frame", int(d[0]), "codebooks", (o[0, :, int(d[0])] != ref[0, :, int(d[0])]).nonzero().flatten().tolist())
    else:
        print(k, "equal")
