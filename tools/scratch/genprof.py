import cProfile, pstats, sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from zonos_amd import synth
from zonos_amd.testing import build_model
model, _ = build_model(synth.FULL_CFG, 1234, "cuda:0")
eng = model.engine(1); eng.call("zn_debug_eos_bias", float("-inf"))
cond = synth.conditioning(1234, "cond", 2, 24, 2048).to("cuda:0")
g = lambda: model.generate(cond, max_new_tokens=861, sampling_params={"temperature": 0.0})
g(); torch.cuda.synchronize()
t=time.perf_counter(); g(); torch.cuda.synchronize(); print("generate wall", time.perf_counter()-t)
pr = cProfile.Profile(); pr.enable(); g(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
