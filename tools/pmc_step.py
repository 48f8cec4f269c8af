"""Whole decode steps of BASELINE config 2 for the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs):
    python3 tools/pmc_step.py [decode_steps]       # single-step launches: every kernel of a step is one dispatch row"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zonos_amd import synth  # noqa: E402
from zonos_amd.testing import build_model  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 72
model, _ = build_model(synth.FULL_CFG, 1234, "cuda:0")
eng = model.engine(1)
eng.call("zn_debug_eos_bias", float("-inf"))
eng.call("zn_debug_tune", 6, 1)        # no multi-step graphs: rocprofv3 sees each kernel as its own dispatch either way
cond = synth.conditioning(1234, "cond", 2, 24, 2048).to("cuda:0")
out = model.generate(cond, max_new_tokens=steps - 8, sampling_params={"temperature": 0.0})
torch.cuda.synchronize()
print(f"generated {tuple(out.shape)}: {steps} decode steps")
