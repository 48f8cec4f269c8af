"""N > 1 path on CPU: 2 ranks over gloo run the utterance sharding + ragged code gather of zonos_amd/parallel.py and must
reproduce the single-process result in utterance order (per-utterance equality with 1-GPU output)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from zonos_amd import parallel


def _fake_generate(cond: torch.Tensor) -> torch.Tensor:
    """Deterministic stand-in for Zonos.generate: ragged length and content derived from the conditioning."""
    seed = int(cond.abs().sum().item() * 1000) % 9973
    g = torch.Generator().manual_seed(seed)
    T = 5 + seed % 37
    return torch.randint(0, 1024, (1, 9, T), generator=g, dtype=torch.int64)


def _fake_generate_batch(cond: torch.Tensor, b: int) -> torch.Tensor:
    """Batched stand-in (rows [cond_0..cond_{b-1}, uncond_0..]): utterance j's codes are those of its single run, cut to a
    common length the way a fixed-length (EOS-suppressed) batch would be."""
    singles = [_fake_generate(torch.stack([cond[j], cond[b + j]])) for j in range(b)]
    return torch.cat([s[..., :5] for s in singles], dim=0)


def _worker(rank, world, port, n_utt, q, batch=1):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        conds = [torch.full((2, 3, 4), float(i + 1)) for i in range(n_utt)]
        if batch > 1:
            out = parallel.generate_sharded(_fake_generate_batch, conds, gather=True, batch_size=batch)
        else:
            out = parallel.generate_sharded(_fake_generate, conds, gather=True)
        q.put((rank, [o.tolist() for o in out]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_utt", [1, 5, 8])
def test_two_rank_sharded_generate_matches_single_process(n_utt):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_utt, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    conds = [torch.full((2, 3, 4), float(i + 1)) for i in range(n_utt)]
    single = [_fake_generate(c)[0].tolist() for c in conds]
    assert results[0] == single and results[1] == single


@pytest.mark.parametrize("n_utt,batch", [(6, 2), (7, 4), (16, 8)])
def test_two_rank_batched_sharding_matches_single_process(n_utt, batch):
    """Groups of `batch` utterances per generate() call (BASELINE config 3's per-GPU batches), ragged last group."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_utt, q, batch)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    conds = [torch.full((2, 3, 4), float(i + 1)) for i in range(n_utt)]
    single = [_fake_generate(c)[0][..., :5].tolist() for c in conds]
    assert results[0] == single and results[1] == single


def _run_bench(*argv, env=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None), e.pop("RANK", None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], capture_output=True, text=True, timeout=300, env=e)
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    return p.returncode, lines, (json.loads(lines[-1]) if lines and lines[-1].startswith("{") else None)


@pytest.mark.parametrize("batch", [1, 2])
def test_bench_spawns_its_own_ranks(batch):
    """`python bench.py --gpus 2` with no launcher environment starts its two ranks itself (gloo dry run: no GPU), goes
    through zonos_amd/parallel.py and prints exactly one JSON line carrying the contract's keys."""
    rc, lines, js = _run_bench("--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1", "--seconds", "0.5", "--batch-per-gpu", str(batch))
    assert rc == 0 and len(lines) == 1 and js is not None, (rc, lines)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in js, k
    assert js["n_gpus"] == 2 and js["dry_run"] is True and js["scaling"] == "weak"
    assert js["config"]["codes_all_gather_in_timed_region"] is True
    assert js["frames_per_sec"] > 0 and f"batch={batch} per GPU" in js["config"]["workload"]


def test_bench_propagates_a_failing_rank():
    rc, lines, js = _run_bench("--gpus", "2", "--dry-run", "--steps", "1", "--warmup", "0", "--seconds", "0.2", env={"ZN_BENCH_DRYRUN_FAIL_RANK": "1"})
    assert rc != 0 and js is None, (rc, lines)


def test_shard_indices_partition():
    for n in (0, 1, 7, 64):
        for w in (1, 2, 8):
            seen = sorted(i for r in range(w) for i in parallel.shard_indices(n, r, w))
            assert seen == list(range(n))
    assert parallel.gather_codes([torch.zeros(9, 3, dtype=torch.int64)], 1)[0].shape == (9, 3)
