"""Run only one decode kernel (argument = zn_bench_kernel's `which`; 6 = the whole-step kernel, the dominant kernel at batch 1; 5 = the persistent chain launch of the per-block path; 0 = the LayerNorm+fc1+SiLU GEMV of the launches path) 260 times, cycling over the 26 layers' weights.
Meant to be wrapped by rocprofv3 (--kernel-trace --stats, or --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zonos_amd import _lib, synth  # noqa: E402
from zonos_amd.testing import build_model  # noqa: E402

model, _ = build_model(synth.FULL_CFG, 1234, "cuda:0")
eng = model.engine(1)
t, by = C.c_float(0), C.c_double(0)
which = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if which == 6:      # the whole-step kernel: one launch = a decode step's 26 blocks + heads at bench.STEP_KERNEL_CTX keys of context
    from bench import STEP_KERNEL_CTX
    eng.call("zn_bench_kernel", 6, 2 | (STEP_KERNEL_CTX << 16), 60, C.byref(t), C.byref(by), _lib.stream_ptr())
else:
    eng.call("zn_bench_kernel", which, 2, 260, C.byref(t), C.byref(by), _lib.stream_ptr())
print(f"kernel {which}: {t.value * 1e3:.2f} us/launch by HIP events, algorithmic {by.value / 1e6:.2f} MB/launch, {by.value / t.value / 1e6:.1f} GB/s")
