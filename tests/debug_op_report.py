"""Ad-hoc op-level HIP-vs-oracle mismatch report (checker tooling, lives with the tests because only they may use the
oracle; run on the GPU box: python tests/debug_op_report.py).  Not collected by pytest (no test_ prefix)."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import zonos_oracle as zo  # noqa: E402
from zonos_amd import _lib, synth  # noqa: E402
from zonos_amd.testing import build_model  # noqa: E402


def bits(t):
    return t.contiguous().view(torch.int16)


def main():
    which = sys.argv[1:] or ["linear", "layer", "sampler"]
    dev = "cuda:0"
    if "sampler" in which:
        model, w = build_model(synth.TINY_CFG, 77, dev)
        eng = model.engine(1)
        st = _lib.stream_ptr()
        lg = torch.from_numpy(synth.normal(99, "logits", (2, 9, 1025), 3.0))
        gen = torch.from_numpy(synth.randint(99, "gen", (2, 9, 7), 1025))
        for name, kw, use_gen in (("plain softmax", dict(temperature=1.0), False), ("penalty+softmax", dict(temperature=1.0), True),
                                  ("min_p", dict(temperature=1.0, min_p=0.1), False)):
            p = dict(temperature=1.0, top_p=0.0, top_k=0, min_p=0.0, linear=0.0, conf=0.0, quad=0.0)
            p.update(kw)
            sp = _lib.zn_sampling(repetition_penalty=3.0, repetition_penalty_window=2, seed=5, **p)
            probs = torch.zeros(2, 9, 1025, dtype=torch.float32, device=dev)
            toks = torch.empty(2, 9, dtype=torch.int32, device=dev)
            lgd, gd = lg.to(dev), gen.to(torch.int32).to(dev)
            eng.call("zn_op_sample", lgd.data_ptr(), gd.data_ptr() if use_gen else None, 7 if use_gen else 0, C.byref(sp), 0, toks.data_ptr(), probs.data_ptr(), 2, st)
            torch.cuda.synchronize()
            src = zo.repetition_penalty(lg, gen, 3.0, 2) if use_gen else lg
            ref = zo.filtered_probs(src, p["temperature"], p["top_p"], p["top_k"], p["min_p"], p["linear"], p["conf"], p["quad"])
            d = (probs.cpu() - ref).abs()
            idx = np.unravel_index(int(d.argmax()), d.shape)
            print(f"[sampler {name}] max|dp| {d.max().item():.3g} at {idx}: got {probs.cpu()[idx].item():.6g} ref {ref[idx].item():.6g}; sum got {probs[0, 0].sum().item():.6f}")
    if "linear" in which or "layer" in which:
        model, w = build_model(synth.FULL_CFG, 1234, dev)
        eng = model.engine(1)
        st = _lib.stream_ptr()
    if "linear" in which:
        p = "backbone.layers.0."
        x = synth.conditioning(1234, "dbg.x", 2, 1, 2048)[:, 0].contiguous()
        for name, wk, ln in (("LN+in_proj", p + "mixer.in_proj.weight", True), ("out_proj", p + "mixer.out_proj.weight", False),
                             ("LN+fc1", p + "mlp.fc1.weight", True), ("LN+heads", "fused_heads.weight", True)):
            W = w[wk]
            out = torch.empty(2, W.shape[0], dtype=torch.bfloat16, device=dev)
            lw, lb = w[p + "norm.weight"], w[p + "norm.bias"]
            eng.call("zn_op_linear", x.to(dev).data_ptr(), lw.to(dev).data_ptr() if ln else None, lb.to(dev).data_ptr() if ln else None,
                     W.to(dev).data_ptr(), out.data_ptr(), 2, W.shape[0], W.shape[1], st)
            torch.cuda.synchronize()
            ref = F.linear(F.layer_norm(x, (2048,), lw, lb, 1e-5) if ln else x, W)
            mm = (bits(out.cpu()) != bits(ref)).float().mean().item()
            print(f"[linear {name}] N={W.shape[0]} K={W.shape[1]} bit-mismatch {mm:.6f} max|d| {(out.cpu().float() - ref.float()).abs().max().item():.4g}")
        x8 = synth.conditioning(1234, "dbg.x8", 2, 1, 8192)[:, 0].contiguous()
        W = w[p + "mlp.fc2.weight"]
        out = torch.empty(2, 2048, dtype=torch.bfloat16, device=dev)
        eng.call("zn_op_linear", x8.to(dev).data_ptr(), None, None, W.to(dev).data_ptr(), out.data_ptr(), 2, 2048, 8192, st)
        torch.cuda.synchronize()
        ref = F.linear(x8, W)
        print(f"[linear fc2] bit-mismatch {(bits(out.cpu()) != bits(ref)).float().mean().item():.6f}")
    if "layer" in which:
        for L in (1, 17, 300, 900):
            x = synth.conditioning(1234, f"ops.x.{L}", 2, 1, 2048)
            kv = torch.from_numpy(synth.normal(1234, f"ops.kv.{L}", (2, 904, 2, 4, 128))).to(torch.bfloat16)
            xd, kvd = x[:, 0].contiguous().to(dev), kv.to(dev)
            lengths = torch.full((2,), L - 1, dtype=torch.int32, device=dev)
            eng.call("zn_op_layer_decode", 0, xd.data_ptr(), kvd.data_ptr(), 904, lengths.data_ptr(), None, 2, st)
            torch.cuda.synchronize()
            cache = zo.Cache([kv.clone()], 904, L - 1, torch.full((2,), L - 1, dtype=torch.int32), zo.rope_table(16384, 128))
            cs = cache.rope[cache.lengths.long().unsqueeze(-1)]
            y = zo.layer_forward(w, 0, x, cache, cs, synth.FULL_CFG)
            mk = (bits(kvd[:, L - 1].cpu()) != bits(cache.kv[0][:, L - 1])).float().mean().item()
            my = (bits(xd.cpu()) != bits(y[:, 0])).float().mean().item()
            print(f"[layer0 L={L}] new K/V bit-mismatch {mk:.6f}; block output bit-mismatch {my:.6f}; max|d| {(xd.cpu().float() - y[:, 0].float()).abs().max().item():.4g}")


if __name__ == "__main__":
    main()
