"""Decode-step time at batch 1 over the context length, per decode path (GPU box): hipGraph-replayed steps on a cache filled with
random rows, `steps` steps from each starting context.

    python tools/ctxsweep.py [steps=160] [contexts=100,300,450,...] [modes=stack,chain,launches]

modes: stack = the default (whole-step kernel); stack-nopre = the same with block 0's in_proj as a launch before it (zn_debug_tune(18, 2));
chain = one chain launch per block (zn_debug_tune(15, 2)); launches = per-op launches.
Prints ms per decode step and the algorithmic HBM rate (SURVEY.md section 8d bytes at the middle context of the run)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zonos_amd import _lib, synth  # noqa: E402
from zonos_amd.codebook_pattern import apply_delay_pattern  # noqa: E402
from zonos_amd.model import _sampling_struct  # noqa: E402
from zonos_amd.testing import build_model  # noqa: E402

MODES = {"stack": {15: 1, 8: 1, 18: 1}, "stack-nopre": {15: 1, 8: 1, 18: 2}, "chain": {15: 2, 8: 1, 18: 1}, "launches": {15: 1, 8: 2, 18: 1}}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    ctxs = [int(c) for c in (sys.argv[2] if len(sys.argv) > 2 else "100,300,450,600,900,1100,1600,2100,2600,3200,3800,4400,5000,5600").split(",")]
    modes = (sys.argv[3] if len(sys.argv) > 3 else "stack,chain").split(",")
    dev = "cuda:0"
    model, _ = build_model(synth.FULL_CFG, 1234, dev)
    eng = model.engine(1)
    eng.call("zn_debug_eos_bias", float("-inf"))
    extra = [tuple(int(v) for v in kv.split("=")) for kv in filter(None, os.environ.get("ZN_TUNE", "").split(","))]      # e.g. ZN_TUNE=17=4
    st = _lib.stream_ptr()
    sp = _sampling_struct({"temperature": 0.0}, 1)
    W = 3_200_290_816
    print(f"{'context':>12s} " + " ".join(f"{m:>22s}" for m in modes), flush=True)
    for L0 in ctxs:
        max_new = L0 + n + 64
        ip = model.setup_cache(2, L0 + n + 80)
        for i in ip.key_value_memory_dict:
            ip.key_value_memory_dict[i][0].normal_()
        cells = []
        for m in modes:
            for k, v in list(MODES[m].items()) + extra:
                eng.call("zn_debug_tune", k, v)
            codes = torch.randint(0, 1024, (1, 9, max_new), dtype=torch.int32, device=dev)
            codes[..., L0:] = -1
            delayed = apply_delay_pattern(codes, 1025).contiguous()
            kv = (C.c_void_p * 26)(*[ip.key_value_memory_dict[i][0].data_ptr() for i in range(26)])
            ip.lengths_per_sample.fill_(L0)
            eng.call("zn_gen_begin", 1, kv, ip.max_seqlen, ip.lengths_per_sample.data_ptr(), delayed.data_ptr(), delayed.shape[2], L0 + 9, max_new, 2.0, C.byref(sp), st)
            eng.call("zn_decode_steps", 24, st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng.call("zn_decode_steps", n, st)
            e1.record()
            torch.cuda.synchronize()
            done = C.c_int32(0)
            eng.call("zn_all_stopped", C.byref(done), st)        # surfaces a hand-off timeout
            ms = e0.elapsed_time(e1) / n
            path = eng.lib.zn_decode_path_detail(eng.h)
            eng.call("zn_gen_end")
            Lmid = L0 + 24 + n / 2
            rate = (W + 2 * Lmid * 53248 + 2 * 53248) / (ms * 1e-3) / 1e12
            cells.append(f"{ms:7.4f} ms p{path} {rate:4.2f}TB/s")
        print(f"{L0:5d}-{L0 + 24 + n:<6d} " + " ".join(f"{c:>22s}" for c in cells), flush=True)
    for k, v in MODES["stack"].items():
        eng.call("zn_debug_tune", k, v)
    print("counters:", eng.counters())


if __name__ == "__main__":
    main()
