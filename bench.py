"""bench.py — headline benchmark of the Zonos hot path on MI355X (contract: task brief ④).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one synthetic utterance per GPU (BASELINE.json configs[1]):
Zonos-v0.1-transformer dims, bf16, batch 1, L_c = 24 synthetic conditioning positions, 861 new tokens (10 s of
audio) with EOS suppressed so that the prefill and all 868 decode steps run, then DAC decode of the [1, 9, 861]
codes.  Weights are seeded synthetic (no checkpoint exists offline).  Utterances shard over ranks with no data-path
collective (weak scaling); the optional gather of the output codes (RCCL all_gather) is inside the timed region.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRAME_RATE = 44100 / 512          # 86.1328 DAC frames per second of audio
HBM_PEAK = 8.0e12                 # B/s, MI355X spec (MI355X_MICROARCH.md)


def algorithmic_bytes_per_step(cfg, B, L):
    """weights read once (out_proj counted once) + KV read + KV write."""
    d, nl, F = cfg["d_model"], cfg["n_layer"], cfg["d_ff"]
    hd = d // cfg["num_heads"]
    nq, nkv = cfg["num_heads"] * hd, cfg["num_heads_kv"] * hd
    W = nl * ((nq + 2 * nkv) * d + d * nq + 2 * F * d + d * F) * 2 + 9 * 1025 * d * 2
    kv_pos = nl * 2 * nkv * 2          # bytes per row-position over all layers (53 248)
    return W + 2 * B * L * kv_pos + 2 * B * kv_pos


def main():
    # Libraries (RCCL prints a version banner) write to fd 1; the contract is ONE JSON line on stdout, so keep a private
    # copy of stdout for that line and point fd 1 at stderr for everything else.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=10.0, help="target audio length per utterance")
    ap.add_argument("--batch-per-gpu", type=int, default=1, help="utterances per GPU (default 1 = BASELINE config 2; 8 = config 3's share)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dac", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("ZN_BENCH_FORCE_DIST") == "1":      # the env var rehearses the RCCL path on one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from zonos_amd import _lib, synth
    from zonos_amd.autoencoder import DACAutoencoder
    from zonos_amd.testing import build_model

    cfg, seed = synth.FULL_CFG, 1234
    max_new = int(round(args.seconds * FRAME_RATE))             # 861 for 10 s
    l_c, B = 24, args.batch_per_gpu
    t0 = time.time()
    dac = None
    if not args.no_dac:
        dac = DACAutoencoder(synth.dac_state_dict(4321), device=dev)
    model, w_cpu = build_model(cfg, seed, dev, dac=dac)
    setup_s = time.time() - t0
    log(f"rank {rank}: model built in {setup_s:.1f} s")
    eng = model.engine(B)
    eng.call("zn_debug_eos_bias", float("-inf"))                 # suppress EOS: all max_new+7 steps run
    cond = torch.cat([synth.conditioning(seed + rank * 64 + i, "cond", 2, l_c, cfg["d_model"])[j:j + 1] for j in (0, 1) for i in range(B)], 0).to(dev)
    steps_per_utt = max_new + 7

    use_dac = dac is not None
    if use_dac:
        try:
            dac.decode(torch.zeros(1, 9, 4, dtype=torch.int64, device=dev))
        except _lib.ZonosHipError as e:
            if rank == 0:
                print(f"[bench] DAC decode unavailable ({e}); timing the AR path only", file=sys.stderr)
            use_dac = False

    def one_step():
        codes = model.generate(cond, max_new_tokens=max_new, cfg_scale=2.0, batch_size=B, sampling_params={"temperature": 0.0})
        wav = dac.decode(codes) if use_dac else None
        if dist is not None:
            out = [torch.empty_like(codes) for _ in range(world)]
            dist.all_gather(out, codes)                          # C1: optional gather of the output codes
        return codes, wav

    def fence():
        if dist is not None:
            dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one_step()
        torch.cuda.synchronize()
        log(f"rank {rank}: warmup {i} done")
    fence()
    t_ar = 0.0
    t0 = time.perf_counter()
    frames = 0
    for _ in range(args.steps):
        codes, wav = one_step()
        frames += codes.shape[-1] * codes.shape[0]
    fence()
    elapsed = time.perf_counter() - t0
    log(f"rank {rank}: {args.steps} timed steps in {elapsed:.3f} s")
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        fr = torch.tensor([frames], dtype=torch.float64, device=dev)
        dist.all_reduce(fr, op=dist.ReduceOp.SUM)
        frames = int(fr.item())
    assert codes.shape[-1] == max_new, codes.shape

    # AR-only time of one utterance (reported beside the whole-job value)
    torch.cuda.synchronize()
    ta = time.perf_counter()
    model.generate(cond, max_new_tokens=max_new, cfg_scale=2.0, batch_size=B, sampling_params={"temperature": 0.0})
    torch.cuda.synchronize()
    t_ar = time.perf_counter() - ta

    audio_s = frames / FRAME_RATE
    value = audio_s / elapsed
    result = {
        "metric": "audio-sec/sec (RTF), Zonos-v0.1-transformer AR decode loop + DAC decode", "value": round(value, 3), "unit": "audio-sec/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"Zonos-v0.1-transformer bf16, batch={B} per GPU, {args.seconds:g} s target audio ({max_new} frames, "
                               f"{steps_per_utt} decode steps + prefill of {l_c + 1} positions, EOS suppressed), greedy, cfg_scale 2",
                   "weights": "seeded synthetic (zonos_amd/synth.py)", "dac_decode_in_timed_region": bool(use_dac),
                   "codes_all_gather_in_timed_region": dist is not None, "parallelism": f"dp{world} (utterance sharding, no data-path collective)"},
        "dac_tokens_per_sec": round(world * B * args.steps * steps_per_utt * 9 / elapsed, 1),
        "frames_per_sec": round(frames / elapsed, 1),
        "ar_only": {"s_per_utterance": round(t_ar, 4), "ms_per_decode_step": round(1e3 * t_ar / (steps_per_utt + 1), 4),
                    "audio_sec_per_sec": round(B * max_new / FRAME_RATE / t_ar, 3)},
        "setup_s": round(setup_s, 1), "hipgraph_step": bool(eng.lib.zn_graph_active(eng.h)),
    }
    if rank == 0:
        # ---- roofline of the dominant kernel (LayerNorm + fc1 GEMV + SiLU gate: 67 MB of the 123 MB per layer)
        ms, by = C.c_float(0), C.c_double(0)
        eng.call("zn_bench_kernel", 0, 2 * B, 260, C.byref(ms), C.byref(by), _lib.stream_ptr())
        ach = by.value / (ms.value * 1e-3)
        result["roofline"] = {"bound": "hbm", "kernel": "gemv_kernel<R=2,NCH=4,KSPLIT=1,PRO_LN,EPI_SILU> (LayerNorm+fc1+SiLU-gate)",
                              "achieved": round(ach / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(ach / HBM_PEAK, 4),
                              "traffic": pmc_traffic(), "bytes_per_launch": by.value, "us_per_launch": round(ms.value * 1e3, 3)}
        Lavg = l_c + 1 + steps_per_utt / 2
        step_bytes = algorithmic_bytes_per_step(cfg, B, Lavg)
        step_s = t_ar / (steps_per_utt + 1)
        result["step_roofline"] = {"bound": "hbm", "algorithmic_bytes_per_decode_step": int(step_bytes), "achieved": round(step_bytes / step_s / 1e9, 1),
                                   "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(step_bytes / step_s / HBM_PEAK, 4)}
        others = {}
        for which, name in ((1, "fc2+residual"), (2, "out_proj+residual"), (3, "LayerNorm+heads")):
            eng.call("zn_bench_kernel", which, 2 * B, 260, C.byref(ms), C.byref(by), _lib.stream_ptr())
            others[name] = {"us_per_launch": round(ms.value * 1e3, 3), "GB/s": round(by.value / (ms.value * 1e-3) / 1e9, 1)}
        result["other_kernels"] = others
        if world == 1 and B == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(w_cpu, cfg, seed, l_c)
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    if dist is not None:
        dist.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


def cpu_baseline(w_cpu, cfg, seed, l_c, max_steps: int = 24, budget_s: float = 25.0):
    """The CPU oracle (restatement of the reference's torch CPU path, pinned bit-exact against the reference in the
    build container) timed on this box's host cores on a bounded sample of the same workload: the prefill plus up to
    `max_steps` decode steps of the same utterance, stopped after `budget_s` seconds."""
    from oracle import zonos_oracle as zo
    from zonos_amd import synth
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, int(os.environ.get("ZONOS_CPU_THREADS", "16")))   # a 1-GPU box's CPU share is 16 cores
    torch.set_num_threads(cores)
    cond = synth.conditioning(seed, "cond", 2, l_c, cfg["d_model"])
    stamps = []
    t_start = time.perf_counter()

    def cb(frame, step, max_s):
        stamps.append(time.perf_counter())
        return len(stamps) <= max_steps and (stamps[-1] - t_start) < budget_s
    zo.generate(w_cpu, cfg, cond, max_new_tokens=max_steps + 8, sampling_params={"temperature": 0.0}, callback=cb)
    deltas = sorted(b - a for a, b in zip(stamps[:-1], stamps[1:])) or [stamps[0] - t_start]
    s_per_step = deltas[len(deltas) // 2]
    log(f"cpu baseline: {len(stamps)} decode steps, median {s_per_step * 1e3:.1f} ms/step on {cores} threads")
    return {"value": round(1.0 / s_per_step / FRAME_RATE, 4), "unit": "audio-sec/sec", "cores": cores, "kind": "port",
            "sample": f"oracle/zonos_oracle.py generate() on the same utterance: prefill + {len(stamps)} decode steps, median step "
                      f"{s_per_step * 1e3:.1f} ms (torch CPU bf16, {cores} threads, wall budget {budget_s:g} s); DAC decode not included",
            "s_per_decode_step": round(s_per_step, 5)}


def pmc_traffic():
    """HBM bytes per launch of the dominant kernel from the PMC passes (FETCH_SIZE x2 + WRITE_SIZE; separate rocprofv3
    --pmc runs of tools/pmc_kernel.py, summary committed as profiles/r01_d_pmc_fc1.json); None if not collected."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "r01_d_pmc_fc1.json")))["traffic_bytes_per_launch"]
    except Exception:
        return None


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


if __name__ == "__main__":
    main()
