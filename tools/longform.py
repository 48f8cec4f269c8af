"""BASELINE config 5 on one GPU: 30 s synthetic audio prefix (2584 frames) + 30 s generation (2584 new tokens), B = 1,
L_c = 24, greedy, EOS suppressed.  Reports prefill time, decode ms/step at long context, KV bytes and end-to-end RTF.
Optional argv[1] = prefix frames (0: no prefix - `python tools/longform.py 0 2580` is the reference's DEFAULT call, max_new_tokens = 86 * 30,
zonos/model.py:359), argv[2] = new tokens."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zonos_amd import synth  # noqa: E402
from zonos_amd.autoencoder import DACAutoencoder  # noqa: E402
from zonos_amd.testing import build_model  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2584
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2584
dev = "cuda:0"
dac = DACAutoencoder(synth.dac_state_dict(4321), device=dev)
model, _ = build_model(synth.FULL_CFG, 1234, dev, dac=dac)
eng = model.engine(1)
eng.call("zn_debug_eos_bias", float("-inf"))
cond = synth.conditioning(1234, "cond", 2, 24, 2048).to(dev)
prefix = torch.from_numpy(synth.randint(1234, "longprefix", (1, 9, P), 1024)).to(dev) if P > 0 else None
marks = {}


def cb(frame, step, max_steps):
    if step == 1:
        torch.cuda.synchronize()
        marks["first"] = time.perf_counter()
    return True


for it in range(3):        # 0 = warm-up; 1 = with a per-step callback (prefill mark; the steps go out one by one); 2 = the timed run (8-step graphs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    codes = model.generate(cond, audio_prefix_codes=prefix, max_new_tokens=N, sampling_params={"temperature": 0.0}, callback=cb if it == 1 else None)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    wav = dac.decode(codes)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if it == 1:
        prefill = marks["first"] - t0
    print(f"run {it}: generate {t1 - t0:.3f} s, DAC decode of {codes.shape[-1]} frames {t2 - t1:.3f} s", flush=True)
steps = N + 7
L = 24 + P + N + 9
kv = 2 * ((L + 7) // 8 * 8) * 53248
print(f"{'config 5' if P > 0 else 'default 30 s call'}: prefix {P} frames + {N} new tokens; context up to {L}; KV cache {kv / 1e9:.3f} GB (2 rows x {L} x 53248 B)")
print(f"  decode path of the last step: {eng.lib.zn_decode_path_detail(eng.h)} (2 = whole-step kernel); counters {eng.counters()}")
print(f"  prefill of {24 + P + 1} positions + first step: {prefill:.3f} s (with per-step callback sync)")
print(f"  new audio {N / 86.1328:.2f} s in {t2 - t0:.3f} s (generate + DAC of prefix+new) -> {N / 86.1328 / (t2 - t0):.2f}x real-time; codes shape {tuple(codes.shape)}")
print(f"  generate alone {t1 - t0:.3f} s = {(t1 - t0) * 1e3 / (N + 8):.4f} ms per decode step incl. prefill -> {N / 86.1328 / (t1 - t0):.2f}x real-time AR only")
