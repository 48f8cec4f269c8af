"""Weight-streaming kernels of the decode step, cold (cycling through the 26 layers' weights: every launch streams from
HBM) vs hot (the same layer every launch: weights come from L2 / the Infinity Cache) - the bound on what any scheme that
prefetches weights beside the dependent chain could give.  python tools/hotcold.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zonos_amd import _lib, synth
from zonos_amd.testing import build_model
model, _ = build_model(synth.FULL_CFG, 1234, "cuda:0")
eng = model.engine(1); st = _lib.stream_ptr()
t, by = C.c_float(0), C.c_double(0)
for which, name in ((0, "LN+fc1+SiLU"), (1, "fc2+resid"), (2, "out_proj+resid"), (3, "LN+heads")):
    for hot in (0, 0x100):
        eng.call("zn_bench_kernel", which, 2 | hot, 260, C.byref(t), C.byref(by), st)
        print(f"  {name:16s} {'hot ' if hot else 'cold'} {t.value * 1e3:7.2f} us  {by.value / t.value / 1e6:7.1f} GB/s", flush=True)
