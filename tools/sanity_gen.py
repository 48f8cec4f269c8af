"""Broader end-to-end sanity runs on the GPU box (no parity claim: finite outputs, code ranges, lengths, no error flags):
full-size transformer at B = 8 with the production sampler and natural EOS termination, full-size hybrid at B = 2."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zonos_amd import synth  # noqa: E402
from zonos_amd.testing import build_model  # noqa: E402

dev = "cuda:0"
for name, cfg, B in (("transformer", synth.FULL_CFG, 8), ("hybrid", synth.HYBRID_FULL_CFG, 2)):
    model, _ = build_model(cfg, 1234, dev, peaky=True)
    cond = torch.cat([synth.conditioning(10 + i, "cond", 2, 24, 2048)[0:1] for i in range(B)] +
                     [synth.conditioning(10 + i, "cond", 2, 24, 2048)[1:2] for i in range(B)], 0).to(dev)
    for sp in ({"temperature": 0.0}, dict(top_p=0.0, top_k=0, min_p=0.1, linear=0.5, conf=0.4, quad=0.0)):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        codes = model.generate(cond, max_new_tokens=172, batch_size=B, sampling_params=sp, seed=7)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert codes.dtype == torch.int64 and codes.shape[:2] == (B, 9) and int(codes.min()) >= 0 and int(codes.max()) <= 1023
        print(f"{name} B={B} sampler={'greedy' if 'temperature' in sp else 'unified+min_p'}: codes {tuple(codes.shape)} in {dt:.3f} s", flush=True)
    del model
    torch.cuda.empty_cache()
print("ok")
