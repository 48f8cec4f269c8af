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


def _worker(rank, world, port, n_utt, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        conds = [torch.full((2, 3, 4), float(i + 1)) for i in range(n_utt)]
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


def test_shard_indices_partition():
    for n in (0, 1, 7, 64):
        for w in (1, 2, 8):
            seen = sorted(i for r in range(w) for i in parallel.shard_indices(n, r, w))
            assert seen == list(range(n))
    assert parallel.gather_codes([torch.zeros(9, 3, dtype=torch.int64)], 1)[0].shape == (9, 3)
