"""Decode-step micro-benchmark (GPU box): time hipGraph-replayed decode steps at a fixed context length.

    python tools/kbench.py [L0=450] [steps=300]
Prints ms/step by torch events on the launch stream plus the zn_bench_kernel numbers.  Run under
`rocprofv3 --kernel-trace --stats` for the per-kernel split.
"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zonos_amd import _lib, synth  # noqa: E402
from zonos_amd.codebook_pattern import apply_delay_pattern  # noqa: E402
from zonos_amd.model import _sampling_struct  # noqa: E402
from zonos_amd.testing import build_model  # noqa: E402


def main():
    L0 = int(sys.argv[1]) if len(sys.argv) > 1 else 450
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    dev = "cuda:0"
    t0 = time.time()
    model, _ = build_model(synth.FULL_CFG, 1234, dev)
    print(f"model built in {time.time() - t0:.1f} s", flush=True)
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    nq, max_new = 9, L0 + n + 64
    eng = model.engine(B)
    eng.call("zn_debug_eos_bias", float("-inf"))
    for kv_ in filter(None, os.environ.get("ZN_TUNE", "").split(",")):      # e.g. ZN_TUNE=5=1,7=1
        k_, v_ = kv_.split("=")
        eng.call("zn_debug_tune", int(k_), int(v_))
    ip = model.setup_cache(2 * B, L0 + n + 80)
    for i in ip.key_value_memory_dict:
        ip.key_value_memory_dict[i][0].normal_()
    codes = torch.randint(0, 1024, (B, nq, max_new), dtype=torch.int32, device=dev)
    codes[..., L0:] = -1
    delayed = apply_delay_pattern(codes, 1025).contiguous()
    sp = _sampling_struct({"temperature": 0.0}, 1)
    kv = (C.c_void_p * 26)(*[ip.key_value_memory_dict[i][0].data_ptr() for i in range(26)])
    st = _lib.stream_ptr()
    ip.lengths_per_sample.fill_(L0)
    eng.call("zn_gen_begin", B, kv, ip.max_seqlen, ip.lengths_per_sample.data_ptr(), delayed.data_ptr(), delayed.shape[2], L0 + 9, max_new, 2.0, C.byref(sp), st)
    eng.call("zn_decode_steps", 20, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.call("zn_decode_steps", n, st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"B={B} context {L0}..{L0 + n}: {ms:.4f} ms/decode step  ({B * 1e3 / ms / 86.1328:.2f}x real-time aggregate, graph={eng.lib.zn_graph_active(eng.h)})", flush=True)
    if B > 1:
        return
    t, by = C.c_float(0), C.c_double(0)
    for which, name in ((0, "LN+fc1+SiLU"), (1, "fc2+resid"), (2, "out_proj+resid"), (3, "LN+heads")):
        eng.call("zn_bench_kernel", which, 2, 260, C.byref(t), C.byref(by), st)
        print(f"  {name:16s} {t.value * 1e3:7.2f} us  {by.value / t.value / 1e6:7.1f} GB/s", flush=True)


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] == "sweep"):
    main()


def sweep():
    """python tools/kbench.py sweep : GEMV grid-size sweep through zn_debug_tune + zn_bench_kernel."""
    dev = "cuda:0"
    model, _ = build_model(synth.FULL_CFG, 1234, dev)
    eng = model.engine(1)
    st = _lib.stream_ptr()
    t, by = C.c_float(0), C.c_double(0)
    only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
    for key, which, name in ((0, 4, "in_proj"), (2, 0, "fc1"), (3, 1, "fc2"), (1, 2, "out_proj"), (4, 3, "heads")):
        if only and name not in only:
            continue
        for tb in (96, 128, 192, 256, 384, 512, 768, 1024, 1536, 2048):
            eng.call("zn_debug_tune", key, tb)
            eng.call("zn_bench_kernel", which, 2, 260, C.byref(t), C.byref(by), st)
            print(f"{name:9s} target_blocks {tb:5d}: {t.value * 1e3:7.2f} us  {by.value / t.value / 1e6:7.1f} GB/s", flush=True)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "sweep":
    sweep()
