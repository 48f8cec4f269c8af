"""Decode-step time vs context length for the two attention paths (GPU box).

    python tools/attnbench.py [steps=100] [L0,L0,...]
tune[5] = fused-attention KV limit (1 disables the fused launch).
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zonos_amd import _lib, synth  # noqa: E402
from zonos_amd.codebook_pattern import apply_delay_pattern  # noqa: E402
from zonos_amd.model import _sampling_struct  # noqa: E402
from zonos_amd.testing import build_model  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    Ls = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [32, 128, 256, 448, 640, 896, 1400, 1900]
    dev = "cuda:0"
    model, _ = build_model(synth.FULL_CFG, 1234, dev)
    eng = model.engine(1)
    eng.call("zn_debug_eos_bias", float("-inf"))
    st = _lib.stream_ptr()
    for L0 in Ls:
        for fused, multi in ((2048, 2), (1, 2)):
            eng.call("zn_debug_tune", 5, fused); eng.call("zn_debug_tune", 6, multi)
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
            ms = e0.elapsed_time(e1) / n
            print(f"L {L0 + 10:5d}..{L0 + 10 + n:5d} max_len {ip.max_seqlen:5d} {'fused attention   ' if fused > 1 else 'two-pass attention'}: {ms:.4f} ms/step", flush=True)


main()
