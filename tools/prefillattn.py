"""Prefill attention alone (zn_op_attn_prefill) at Zonos-v0.1 dims (16 q heads / 4 kv heads x 128, 2 CFG rows):
matrix-core kernel vs VALU kernel, per-launch time by HIP events and the resulting flop rate.  argv: positions ..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zonos_amd import _lib, synth  # noqa: E402
from zonos_amd.testing import build_model  # noqa: E402

sizes = [int(x) for x in sys.argv[1:]] or [257, 807, 2609, 5200]
dev = "cuda:0"
model, _ = build_model(dict(synth.FULL_CFG, n_layer=1), 1234, dev)
eng = model.engine(1)
st = _lib.stream_ptr()
for S in sizes:
    cap = (S + 7) // 8 * 8
    q = torch.randn(2, S, 2048, device=dev).to(torch.bfloat16)
    kv = torch.randn(2, cap, 2, 4, 128, device=dev).to(torch.bfloat16)
    outs = []
    for kernel, name in ((1, "mfma"), (2, "valu")):
        eng.call("zn_debug_tune", 10, kernel)
        out = torch.empty(2, S, 2048, dtype=torch.bfloat16, device=dev)
        for _ in range(2):
            eng.call("zn_op_attn_prefill", q.data_ptr(), kv.data_ptr(), cap, out.data_ptr(), S, 2, st)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 5
        a.record()
        for _ in range(n):
            eng.call("zn_op_attn_prefill", q.data_ptr(), kv.data_ptr(), cap, out.data_ptr(), S, 2, st)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / n
        flops = 2 * 16 * 128 * 2 * 2 * (S * (S + 1) / 2)          # rows x heads x hd x (QK + PV) x 2 x causal pairs
        print(f"S={S} {name}: {ms:.3f} ms/launch, {flops / ms / 1e9:.1f} TFLOP/s useful", flush=True)
        outs.append(out)
    eq = float((outs[0].view(torch.int16) == outs[1].view(torch.int16)).float().mean())
    print(f"S={S}: mfma vs valu bit-equal {eq:.5f}, max|d| {(outs[0].float() - outs[1].float()).abs().max().item():.3g}", flush=True)
eng.call("zn_debug_tune", 10, 1)
