"""Soak of the default batch-1 decode path (whole-step kernel): N full-length generations (10 s each: prefill + 868 decode steps = 26 x 6
in-kernel hand-offs per step, 26 x 8 beyond 512 keys of context), every one compared with the first (the kernels are deterministic) - codes
must be identical, no hand-off wait may give up, the handle must still be on the whole-step kernel at the end.  Every 10th generation is
followed by a long one (the reference's default 30 s call, or 10 s behind a 2584-frame audio prefix = contexts 2.6 - 3.5 k: the key-block
attention role with 6 - 7 blocks and the change of instantiation at 3072 keys), each compared with its own first run.
    python tools/soak.py [generations] [sampled]      # "sampled": temperature 1.0 + min_p 0.1 with a fixed seed instead of greedy decoding"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zonos_amd import synth  # noqa: E402
from zonos_amd.testing import build_model  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
SP = {"temperature": 1.0, "min_p": 0.1} if len(sys.argv) > 2 and sys.argv[2] == "sampled" else {"temperature": 0.0}
model, _ = build_model(synth.FULL_CFG, 1234, "cuda:0")
eng = model.engine(1)
eng.call("zn_debug_eos_bias", float("-inf"))
cond = synth.conditioning(1234, "cond", 2, 24, 2048).to("cuda:0")
ref = None
t0 = time.perf_counter()
worst = 0.0
long_ref = {}
prefix = torch.from_numpy(synth.randint(1234, "longprefix", (1, 9, 2584), 1024)).to("cuda:0")


def long_one(kind):
    o = (model.generate(cond, max_new_tokens=2580, sampling_params=SP, seed=4242) if kind == "30s" else
         model.generate(cond, audio_prefix_codes=prefix, max_new_tokens=861, sampling_params=SP, seed=4242))
    torch.cuda.synchronize()
    p = eng.lib.zn_decode_path_detail(eng.h)
    same_ = kind not in long_ref or torch.equal(o, long_ref[kind])
    long_ref.setdefault(kind, o.clone())
    print(f"  long generation ({kind}): path {p}, identical to its first run: {same_}", flush=True)
    assert same_ and p == 2, "soak failed (long generation)"


pauses = []
for i in range(n):
    torch.cuda.synchronize()
    ta = time.perf_counter()
    out = model.generate(cond, max_new_tokens=861, sampling_params=SP, seed=4242)
    torch.cuda.synchronize()
    dt = time.perf_counter() - ta
    worst = max(worst, dt)
    c = eng.counters()
    if c["longest_wait_us"] > 0:                # a wait beyond 0.1 ms inside this generation: a pause of the device, observed from inside the kernel
        pauses.append((round(ta - t0, 2), i, c["longest_wait_us"], c["waits_over_200us"], round(dt * 1e3, 1)))
        print(f"  generation {i} at t = {ta - t0:.2f} s: longest in-kernel hand-off wait {c['longest_wait_us']} us ({c['waits_over_200us']} waits beyond 0.2 ms), "
              f"the generation took {dt * 1e3:.1f} ms", flush=True)
        eng.call("zn_debug_tune", 14, 13)
    path = eng.lib.zn_decode_path_detail(eng.h)
    if ref is None:
        ref = out.clone()
    same = torch.equal(out, ref)
    if not same or path != 2 or i % 10 == 0:
        print(f"generation {i}: {dt * 1e3:.1f} ms, path {path}, identical to the first: {same}", flush=True)
    assert same and path == 2, "soak failed"
    if i % 10 == 9:
        long_one("30s" if (i // 10) % 2 == 0 else "prefix")
hand_offs = n * 869 * 26 * 6
print(f"counters: {eng.counters()}")
print(f"pauses observed (t [s], generation, longest wait [us], long waits, generation [ms]): {pauses}")
if len(pauses) > 1:
    print("  intervals between them [s]:", [round(b[0] - a[0], 2) for a, b in zip(pauses[:-1], pauses[1:])])
assert eng.counters()["handoff_timeouts"] == 0
print(f"{n} generations ({hand_offs / 1e6:.1f} M in-kernel hand-offs, the long ones not counted) in {time.perf_counter() - t0:.1f} s, slowest {worst * 1e3:.1f} ms: all identical, no wait gave up")
