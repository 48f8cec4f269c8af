"""Weight-streaming projection kernels at 2 and 16 activation rows (GPU box): us per launch and GB/s from zn_bench_kernel
(HIP events around back-to-back launches cycling over the 26 layers' weights)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zonos_amd import _lib, synth  # noqa: E402
from zonos_amd.testing import build_model  # noqa: E402

model, _ = build_model(synth.FULL_CFG, 1234, "cuda:0")
eng = model.engine(8)
st = _lib.stream_ptr()
t, by = C.c_float(0), C.c_double(0)
for rows in (2, 16):
    for which, name in ((0, "LN+fc1+SiLU [16384x2048]"), (1, "fc2+resid [2048x8192]"), (2, "out_proj+resid [2048x2048]"), (3, "LN+heads [9225x2048]")):
        eng.call("zn_bench_kernel", which, rows, 260, C.byref(t), C.byref(by), st)
        print(f"rows {rows:2d} {name:28s} {t.value * 1e3:7.2f} us  {by.value / t.value / 1e6:7.1f} GB/s", flush=True)
