"""Decode step time with the P.V pass split from different KV capacities (GPU box): python tools/splitbench.py"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zonos_amd import _lib, synth  # noqa: E402
from zonos_amd.codebook_pattern import apply_delay_pattern  # noqa: E402
from zonos_amd.model import _sampling_struct  # noqa: E402
from zonos_amd.testing import build_model  # noqa: E402

dev = "cuda:0"
model, _ = build_model(synth.FULL_CFG, 1234, dev)
eng = model.engine(1)
eng.call("zn_debug_eos_bias", float("-inf"))
st = _lib.stream_ptr()
n = 100
for L0 in (600, 800, 1400, 1900):
    for split_from in (2048, 256):
        eng.call("zn_debug_tune", 9, split_from)
        max_new = L0 + n + 64
        ip = model.setup_cache(2, L0 + n + 40)
        for i in ip.key_value_memory_dict:
            ip.key_value_memory_dict[i][0].normal_()
        codes = torch.randint(0, 1024, (1, 9, max_new), dtype=torch.int32, device=dev)
        codes[..., L0:] = -1
        delayed = apply_delay_pattern(codes, 1025).contiguous()
        sp = _sampling_struct({"temperature": 0.0}, 1)
        kv = (C.c_void_p * 26)(*[ip.key_value_memory_dict[i][0].data_ptr() for i in range(26)])
        ip.lengths_per_sample.fill_(L0)
        eng.call("zn_gen_begin", 1, kv, ip.max_seqlen, ip.lengths_per_sample.data_ptr(), delayed.data_ptr(), delayed.shape[2], L0 + 9, max_new, 2.0, C.byref(sp), st)
        eng.call("zn_decode_steps", 10, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.call("zn_decode_steps", n, st)
        e1.record()
        torch.cuda.synchronize()
        print(f"L {L0 + 10:5d}..{L0 + 10 + n:5d} capacity {ip.max_seqlen:5d} split-from {split_from:5d}: {e0.elapsed_time(e1) / n:.4f} ms/step", flush=True)
