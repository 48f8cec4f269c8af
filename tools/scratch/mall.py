import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
import torch
from zonos_amd import _lib, synth
from zonos_amd.testing import build_model
model, _ = build_model(synth.FULL_CFG, 1234, "cuda:0")
eng = model.engine(1); st = _lib.stream_ptr()
t, by = C.c_float(0), C.c_double(0)
for which, name in ((0, "fc1"), (1, "fc2"), (2, "out_proj")):
    for same in (0, 0x100):
        eng.call("zn_bench_kernel", which, 2 | same, 260, C.byref(t), C.byref(by), st)
        print(f"{name:9s} {'same layer (cache-resident)' if same else 'cycling layers (HBM)      '}: {t.value*1e3:7.2f} us {by.value/t.value/1e6:8.1f} GB/s", flush=True)
