"""BASELINE config 4 (GPU box): Zonos-v0.1-hybrid dimensions (recalled: 46 layers, Mamba2 except attention at
9/19/29/39), batch B utterances on one MI355X, synthetic weights and conditioning.

    python tools/hybridbench.py [B=8] [steps=200] [L0=64]
Prints ms per decode step (hipGraph replay, torch events on the launch stream), aggregate real-time factor and the
algorithmic HBM bytes per step (weights once + Mamba2 state read+write + KV read).  Run it under
`rocprofv3 --kernel-trace --stats` for the per-kernel split (mamba_ssm_kernel = the SSD state update).
Parity of this path is unpinned against the reference (tests/test_gpu_hybrid.py header)."""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zonos_amd import _lib, synth  # noqa: E402
from zonos_amd.backbone._hip import mamba2_dims  # noqa: E402
from zonos_amd.codebook_pattern import apply_delay_pattern  # noqa: E402
from zonos_amd.model import _sampling_struct  # noqa: E402
from zonos_amd.testing import build_model  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    L0 = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    dev = "cuda:0"
    cfg = synth.HYBRID_FULL_CFG
    t0 = time.time()
    model, _ = build_model(cfg, 1234, dev)
    print(f"hybrid model built in {time.time() - t0:.1f} s ({sum(p.numel() for p in model.parameters()) / 1e9:.3f} B parameters)", flush=True)
    nq, max_new = 9, L0 + n + 64
    eng = model.engine(B)
    eng.call("zn_debug_eos_bias", float("-inf"))
    for kv_ in filter(None, os.environ.get("ZN_TUNE", "").split(",")):      # e.g. ZN_TUNE=8=1
        k_, v_ = kv_.split("=")
        eng.call("zn_debug_tune", int(k_), int(v_))
    ip = model.setup_cache(2 * B, L0 + n + 80)
    nl = cfg["n_layer"]
    for i in cfg["attn_layer_idx"]:
        ip.key_value_memory_dict[i][0].normal_()
    codes = torch.randint(0, 1024, (B, nq, max_new), dtype=torch.int32, device=dev)
    codes[..., L0:] = -1
    delayed = apply_delay_pattern(codes, 1025).contiguous()
    sp = _sampling_struct({"temperature": 0.0}, 1)
    kv = (C.c_void_p * nl)(*[ip.key_value_memory_dict[i][0].data_ptr() for i in range(nl)])
    st = _lib.stream_ptr()
    ip.lengths_per_sample.fill_(L0)
    eng.call("zn_gen_begin", B, kv, ip.max_seqlen, ip.lengths_per_sample.data_ptr(), delayed.data_ptr(), delayed.shape[2], L0 + 9, max_new, 2.0, C.byref(sp), st)
    eng.call("zn_decode_steps", 16, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng.call("zn_decode_steps", n, st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    m = mamba2_dims(model.config.backbone)
    n_m = nl - len(cfg["attn_layer_idx"])
    d = cfg["d_model"]
    w_m = (m["d_in_proj"] * d + d * m["d_inner"]) * 2
    w_a = ((16 + 8) * 128 * d + d * d + 2 * cfg["d_ff"] * d + d * cfg["d_ff"]) * 2
    w_heads = 9 * 1025 * d * 2
    state = 2 * B * m["d_inner"] * m["d_state"] * 2 * 2           # rows x (read + write) bf16
    kvb = 2 * B * (L0 + n // 2) * 2 * 4 * 128 * 2
    total = n_m * (w_m + state) + len(cfg["attn_layer_idx"]) * (w_a + kvb) + w_heads
    print(f"B={B} context {L0}..{L0 + n}: {ms:.4f} ms/decode step = {B * 1e3 / ms / 86.1328:.2f}x real-time aggregate (graph={eng.lib.zn_graph_active(eng.h)})", flush=True)
    print(f"algorithmic bytes/step {total / 1e9:.3f} GB (weights {(n_m * w_m + len(cfg['attn_layer_idx']) * w_a + w_heads) / 1e9:.3f}, "
          f"Mamba2 state {n_m * state / 1e9:.3f}) -> {total / ms / 1e6:.0f} GB/s = {total / ms / 1e6 / 8000:.3f} of 8 TB/s", flush=True)


main()
