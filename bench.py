"""bench.py — headline benchmark of the Zonos hot path on MI355X (contract: task brief ④).

    python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher environment: this process spawns the N ranks itself (`python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`) BEFORE anything touches the GPU, relays rank 0's JSON line
and exits with the children's status.  Under an external `torch.distributed.run` (RANK/WORLD_SIZE set) it is a rank.

One "step" = one pass of the hot path over `--batch-per-gpu` synthetic utterances per GPU (default 1 = BASELINE.json
configs[1]): Zonos-v0.1-transformer dims, bf16, L_c = 24 synthetic conditioning positions, 861 new tokens (10 s of
audio) with EOS suppressed so that the prefill and all 868 decode steps run, then DAC decode of the [B, 9, 861] codes.
Weights are seeded synthetic (no checkpoint exists offline).  Utterances shard over ranks through
zonos_amd/parallel.py (utterance i -> rank i mod N, no data-path collective: weak scaling); the optional gather of
the output codes (RCCL all_gather) is inside the timed region.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRAME_RATE = 44100 / 512          # 86.1328 DAC frames per second of audio
HBM_PEAK = 8.0e12                 # B/s, MI355X spec (MI355X_MICROARCH.md)
PMC_FILES = {5: os.path.join(ROOT, "profiles", "pmc_chain.json"), 6: os.path.join(ROOT, "profiles", "pmc_step_kernel.json")}
KERNEL_SOURCES = ("zonos_amd/csrc/zn_step_kernel.h", "zonos_amd/csrc/zn_step_sched.h", "zonos_amd/csrc/zn_chain_kernel.h", "zonos_amd/csrc/zn_decode_kernels.h", "zonos_amd/csrc/zn_common.h")
STEP_KERNEL_CTX = 450             # keys in the cache for the whole-step kernel's roofline launches: the mean context of a 10 s utterance


def algorithmic_bytes_per_step(cfg, B, L):
    """weights read once (out_proj counted once) + KV read + KV write."""
    d, nl, F = cfg["d_model"], cfg["n_layer"], cfg["d_ff"]
    hd = d // cfg["num_heads"]
    nq, nkv = cfg["num_heads"] * hd, cfg["num_heads_kv"] * hd
    W = nl * ((nq + 2 * nkv) * d + d * nq + 2 * F * d + d * F) * 2 + 9 * 1025 * d * 2
    kv_pos = nl * 2 * nkv * 2          # bytes per row-position over all layers (53 248)
    return W + 2 * B * L * kv_pos + 2 * B * kv_pos


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--seconds", type=float, default=10.0, help="target audio length per utterance")
    ap.add_argument("--batch-per-gpu", type=int, default=1,
                    help="utterances per GPU (default 1 = BASELINE config 2).  BASELINE config 3 (64 utterances over 8 GPUs) is "
                         "`--gpus 8 --batch-per-gpu 8`; its one-GPU share is `--batch-per-gpu 8`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dac", action="store_true")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher/sharding rehearsal on CPU over gloo with a stand-in generate(); the line carries dry_run=true and no measurement")
    return ap.parse_args(argv)


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


# ------------------------------------------------------------------------------------------------ launcher
def spawn_ranks(n: int, argv: list[str]) -> int:
    """Start the N ranks as children (no GPU call has happened in this process), pass rank 0's JSON line through."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    log("spawning: " + " ".join(cmd))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # the ranks build 1.6 B synthetic parameters on the host before the timed region: N ranks x all cores would oversubscribe it
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(8, (os.cpu_count() or 8) // max(1, n)))))
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in p.stdout:
        ln = ln.strip()
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    rc = p.wait()
    if rc != 0:
        log(f"ranks exited with status {rc}")
        return rc
    if line is None:
        log("no JSON line from rank 0")
        return 1
    print(line, flush=True)
    return 0


# ------------------------------------------------------------------------------------------------ rank
def run_rank(args) -> int:
    # Libraries (RCCL prints a version banner) write to fd 1; the contract is ONE JSON line on stdout, so keep a private
    # copy of stdout for that line and point fd 1 at stderr for everything else.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    from zonos_amd import parallel

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    dry = args.dry_run
    if dry:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or os.environ.get("ZN_BENCH_FORCE_DIST") == "1":      # the env var rehearses the collective path on one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        if dry:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    def fence():
        if dist is not None:
            if dry:
                dist.barrier()
            else:
                dist.barrier(device_ids=[local_rank])
        if not dry:
            torch.cuda.synchronize()

    # who is here: one line per rank on stderr, and ranks_seen in the JSON line, so that a scaling record proves N ranks on N devices
    if world > 1 and "OMP_NUM_THREADS" not in os.environ:
        torch.set_num_threads(max(1, min(8, (os.cpu_count() or 8) // world)))
    me = {"rank": rank, "local_rank": local_rank, "device": "cpu (dry run)" if dry else torch.cuda.get_device_name(local_rank),
          "host_threads": torch.get_num_threads()}
    log(f"rank {rank}/{world}: device {me['device']}, {me['host_threads']} host threads")
    ranks_seen = [me]
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, me)
        ranks_seen = gathered
        if rank == 0 and not dry:
            try:
                v = torch.cuda.nccl.version()
                log(f"world_size {world}, backend nccl = RCCL {'.'.join(str(x) for x in v)}")
            except Exception as e:                                # version query is informational only
                log(f"world_size {world}, backend nccl (RCCL version unavailable: {e})")

    max_new = int(round(args.seconds * FRAME_RATE))             # 861 for 10 s
    l_c, B = 24, args.batch_per_gpu
    n_utt = world * B
    steps_per_utt = max_new + 7
    t0 = time.time()
    if dry:
        cfg, seed, model, dac, eng, w_cpu = {"d_model": 8}, 1234, None, None, None, None
        if os.environ.get("ZN_BENCH_DRYRUN_FAIL_RANK") == str(rank):   # test hook: a dying rank must fail the whole job
            raise SystemExit(3)

        def gen(cond, b=1):                                     # stand-in: ragged-free deterministic codes of the right shape
            g = torch.Generator().manual_seed(int(cond.abs().sum().item() * 1000) % 9973)
            return torch.randint(0, 1024, (b, 9, max_new), generator=g, dtype=torch.int64)
        conds = [torch.full((2, l_c, 8), float(i + 1)) for i in range(n_utt)]
        use_dac = False
    else:
        from zonos_amd import _lib, synth
        from zonos_amd.autoencoder import DACAutoencoder
        from zonos_amd.testing import build_model
        cfg, seed = synth.FULL_CFG, 1234
        dac = None if args.no_dac else DACAutoencoder(synth.dac_state_dict(4321), device=dev)
        model, w_cpu = build_model(cfg, seed, dev, dac=dac)
        eng = model.engine(B)
        eng.call("zn_debug_eos_bias", float("-inf"))             # suppress EOS: all max_new+7 steps run
        conds = [synth.conditioning(seed + i, "cond", 2, l_c, cfg["d_model"]).to(dev) for i in range(n_utt)]

        def gen(cond, b=1):
            return model.generate(cond, max_new_tokens=max_new, cfg_scale=2.0, batch_size=b, sampling_params={"temperature": 0.0})
        use_dac = dac is not None
        if use_dac:
            try:
                dac.decode(torch.zeros(1, 9, 4, dtype=torch.int64, device=dev))
            except _lib.ZonosHipError as e:
                if rank == 0:
                    log(f"DAC decode unavailable ({e}); timing the AR path only")
                use_dac = False
    setup_s = time.time() - t0
    log(f"rank {rank}/{world}: ready in {setup_s:.1f} s ({'dry run, cpu/gloo' if dry else str(dev)})")

    def one_step():
        mine = parallel.generate_sharded(gen, conds, gather=False, batch_size=B)      # this rank's utterances (i mod world == rank)
        wav = dac.decode(torch.stack(mine)) if use_dac else None
        if dist is not None:
            parallel.gather_codes(mine, n_utt)                   # C1: optional gather of the output codes (all_gather)
        return mine, wav

    for i in range(args.warmup):
        one_step()
        if not dry:
            torch.cuda.synchronize()
        log(f"rank {rank}: warmup {i} done")
    fence()

    def handoff_state():
        # hand-off timeouts are never silent (include/zonos_hip.h zn_get_counters): reported timeouts, generations that ran the launches path
        # because a timeout had demoted the handle, generations Zonos.generate repeated after one
        if dry:
            return (0, 0, 0)
        hc = model.handoff_counters()
        engines = [hc[k] for k in ("engine", "spare") if k in hc]
        return (sum(e["handoff_timeouts"] for e in engines), sum(e["fallback_generations"] for e in engines), hc["repeated_generations"])
    h0 = handoff_state()
    t0 = time.perf_counter()
    frames = 0
    for _ in range(args.steps):
        mine, wav = one_step()
        frames += sum(int(c.shape[-1]) for c in mine)
    fence()
    elapsed = time.perf_counter() - t0
    h1 = handoff_state()
    timed_events = [b - a for a, b in zip(h0, h1)]
    log(f"rank {rank}: {args.steps} timed steps in {elapsed:.3f} s")
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        fr = torch.tensor([frames], dtype=torch.float64, device=dev)
        dist.all_reduce(fr, op=dist.ReduceOp.SUM)
        frames = int(fr.item())
        ev = torch.tensor(timed_events + list(h1), dtype=torch.float64, device=dev)
        dist.all_reduce(ev, op=dist.ReduceOp.SUM)
        timed_events, h1 = [int(v) for v in ev[:3].tolist()], tuple(int(v) for v in ev[3:].tolist())
    assert all(c.shape[-1] == max_new for c in mine), [c.shape for c in mine]

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
        "setup_s": round(setup_s, 1),
        "ranks_seen": ranks_seen,
        # all ranks: bounded in-kernel hand-off waits that gave up (each voids a generation), generations that ran the launches path because of
        # one, generations repeated after one - since the handles were created, and inside the timed region (any of those fails the run)
        "handoff_timeouts": h1[0], "fallback_generations": h1[1], "repeated_generations": h1[2],
        "handoff_events_in_timed_region": {"timeouts": timed_events[0], "fallback_generations": timed_events[1], "repeated_generations": timed_events[2]},
    }
    if dry:
        result["dry_run"] = True
        result["value"] = None
    else:
        # AR-only time of one generate() call (reported beside the whole-job value)
        cond = torch.cat([c[0:1] for c in conds[rank::world]] + [c[1:2] for c in conds[rank::world]], 0)
        torch.cuda.synchronize()
        ta = time.perf_counter()
        gen(cond, B)
        torch.cuda.synchronize()
        t_ar = time.perf_counter() - ta
        result["ar_only"] = {"s_per_utterance": round(t_ar, 4), "ms_per_decode_step": round(1e3 * t_ar / (steps_per_utt + 1), 4),
                             "audio_sec_per_sec": round(B * max_new / FRAME_RATE / t_ar, 3)}
        result["hipgraph_step"] = bool(eng.lib.zn_graph_active(eng.h))
        result["chain_kernel_path"] = bool(eng.lib.zn_decode_path(eng.h))     # the generation just run: persistent kernels or launches
        result["decode_path"] = int(eng.lib.zn_decode_path_detail(eng.h))      # 0 launches, 1 chain launch per block, 2 whole-step kernel
    if rank == 0 and not dry:
        from zonos_amd import _lib
        # ---- roofline of the dominant kernel, HIP events on the launch stream; the launches cycle over the 26 layers' weights so
        # that each streams from HBM.  Batch 1: the persistent chain launch (out_proj x2, LayerNorm+fc1+SiLU, fc2, next in_proj:
        # ~70 % of the decode step's GPU time); larger batches: the LayerNorm + fc1 + SiLU-gate GEMM of the launches path.
        ms, by = C.c_float(0), C.c_double(0)
        path = result.get("decode_path", 0)
        which = (6 if path == 2 else 5 if path == 1 else 0) if B == 1 else 0
        if which == 6:       # one launch = every block of a decode step + the heads at a context of STEP_KERNEL_CTX keys
            eng.call("zn_bench_kernel", 6, 2 | (STEP_KERNEL_CTX << 16), 80, C.byref(ms), C.byref(by), _lib.stream_ptr())
        else:
            eng.call("zn_bench_kernel", which, 2 * B, 260, C.byref(ms), C.byref(by), _lib.stream_ptr())
        ach = by.value / (ms.value * 1e-3)
        traffic, traffic_note = pmc_traffic(which) if which in PMC_FILES else (None, "PMC summaries cover the batch-1 persistent kernels only")
        kname = {6: "step_kernel<NCH=4,T_OUT=2,T_FC1=10,T_FC2=5,T_IN=6,NBV=6> (whole decode step in one persistent launch: per block attention, out_proj x2, LayerNorm+fc1+SiLU-gate, fc2, "
                    f"next block's LayerNorm+in_proj+RoPE+KV append; norm_f + heads; context {STEP_KERNEL_CTX} keys)",
                 5: "chain_kernel<NCH=4,T_OUT=1,T_FC1=8,T_FC2=4,T_IN=2> (out_proj x2 + LayerNorm+fc1+SiLU-gate + fc2 + next block's LayerNorm+in_proj+RoPE+KV append, one persistent launch)",
                 0: "gemm16s_kernel<EPI_SILU, ., LNP> (5..16 rows: LayerNorm from the statistics out_proj's epilogue left + fc1 + SiLU-gate in one launch) / gemv LayerNorm+fc1+SiLU-gate (2..4 rows)"}[which]
        result["roofline"] = {"bound": "hbm", "kernel": kname,
                              "achieved": round(ach / 1e9, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(ach / HBM_PEAK, 4),
                              "traffic": traffic, "traffic_note": traffic_note, "bytes_per_launch": by.value, "us_per_launch": round(ms.value * 1e3, 3)}
        Lavg = l_c + 1 + steps_per_utt / 2
        step_bytes = algorithmic_bytes_per_step(cfg, B, Lavg)
        step_s = t_ar / (steps_per_utt + 1)
        result["step_roofline"] = {"bound": "hbm", "algorithmic_bytes_per_decode_step": int(step_bytes), "achieved": round(step_bytes / step_s / 1e9, 1),
                                   "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": round(step_bytes / step_s / HBM_PEAK, 4)}
        others = {}
        for which, name in ((0, "LayerNorm+fc1+SiLU-gate (launches path)"), (1, "fc2+residual (launches path)"), (2, "out_proj+residual (launches path)"), (3, "LayerNorm+heads")):
            eng.call("zn_bench_kernel", which, 2 * B, 260, C.byref(ms), C.byref(by), _lib.stream_ptr())
            others[name] = {"us_per_launch": round(ms.value * 1e3, 3), "GB/s": round(by.value / (ms.value * 1e-3) / 1e9, 1)}
        result["other_kernels"] = others
        if world == 1 and B == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(w_cpu, cfg, seed, l_c, max_new)
    if rank == 0:
        os.write(json_fd, (json.dumps(result) + "\n").encode())
    if dist is not None:
        fence()
        dist.destroy_process_group()
    if any(timed_events):
        log(f"FAILED: the timed region contained hand-off events {result['handoff_events_in_timed_region']}: its steps did not all run the measured path")
        return 4
    return 0


def cpu_baseline(w_cpu, cfg, seed, l_c, max_new, ar_steps: int = 128, budget_s: float = 40.0):
    """The CPU oracle (restatement of the reference's torch CPU path, pinned bit-exact against the reference in the
    build container) timed on this box's host cores on a bounded sample of the same workload (SURVEY.md §8d,
    BASELINE.md §3): the prefill plus `ar_steps` decode steps of the same utterance (stopped early past `budget_s`),
    then the DAC decode of a [1, 9, max_new] code tensor.  `value` extrapolates the per-step median to the whole
    utterance: max_new / 86.13 s of audio over (prefill + (max_new + 7) steps + DAC decode)."""
    import torch
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
        return len(stamps) <= ar_steps and (stamps[-1] - t_start) < budget_s
    zo.generate(w_cpu, cfg, cond, max_new_tokens=ar_steps + 8, sampling_params={"temperature": 0.0}, callback=cb)
    deltas = sorted(b - a for a, b in zip(stamps[:-1], stamps[1:])) or [stamps[0] - t_start]
    s_per_step = deltas[len(deltas) // 2]
    prefill_s = stamps[0] - t_start - s_per_step if len(stamps) > 1 else 0.0
    log(f"cpu baseline: {len(stamps)} decode steps, median {s_per_step * 1e3:.1f} ms/step on {cores} threads")
    dac_s, dac_note = None, "DAC decode leg skipped (AR leg used the wall budget)"
    if time.perf_counter() - t_start < budget_s:
        dw = synth.dac_state_dict(4321)
        codes = torch.randint(0, 1024, (1, 9, max_new), generator=torch.Generator().manual_seed(7), dtype=torch.int64)
        td = time.perf_counter()
        with torch.no_grad():
            zo.dac_decode(dw, codes)
        dac_s = time.perf_counter() - td
        dac_note = f"DAC decode of [1, 9, {max_new}] in {dac_s:.2f} s (fp32)"
        log("cpu baseline: " + dac_note)
    total = max(prefill_s, 0.0) + (max_new + 7) * s_per_step + (dac_s or 0.0)
    return {"value": round(max_new / FRAME_RATE / total, 4), "unit": "audio-sec/sec", "cores": cores, "kind": "port",
            "sample": f"oracle/zonos_oracle.py generate() on the same utterance: prefill + {len(stamps)} decode steps, median step "
                      f"{s_per_step * 1e3:.1f} ms (torch CPU bf16, {cores} threads, wall budget {budget_s:g} s), extrapolated to {max_new + 7} steps; {dac_note}",
            "s_per_decode_step": round(s_per_step, 5), "dac_decode_s": None if dac_s is None else round(dac_s, 3),
            "ar_only_audio_sec_per_sec": round(1.0 / s_per_step / FRAME_RATE, 4)}


def kernel_source_hash() -> str:
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(which=5):
    """HBM bytes per launch of the dominant kernel from the PMC passes (FETCH_SIZE x2 + WRITE_SIZE; separate rocprofv3
    --pmc runs of tools/pmc_kernel.py <which>, summarised by tools/pmc_summary.py into profiles/pmc_*.json).  The summary
    records the kernel's name and a hash of the kernel sources it was measured on: a summary of another kernel state is
    refused (traffic = null), never reported."""
    f = PMC_FILES[which]
    try:
        rec = json.load(open(f))
    except Exception:
        return None, f"no PMC summary ({os.path.relpath(f, ROOT)})"
    if rec.get("kernel_source_sha256_16") != kernel_source_hash():
        return None, f"PMC summary is stale: measured on kernel sources {rec.get('kernel_source_sha256_16')}, current {kernel_source_hash()}"
    return rec["traffic_bytes_per_launch"], f"{rec.get('kernel_name')}; rocprofv3 --pmc passes of {rec.get('date', '?')}"


def main() -> int:
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args.gpus, sys.argv[1:])
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
