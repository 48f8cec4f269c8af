"""Data-parallel utterance sharding: one process per GPU, weights replicated, utterance i -> rank
i mod world, no collective on the data path.  The only collective is the optional gather of the finished code
tensors (ragged in time: padded to the longest, lengths travel alongside) — RCCL all_gather on GPUs, gloo on CPU."""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_indices(n_utterances: int, rank: int, world: int) -> list[int]:
    """Round-robin assignment: utterance i runs on rank i % world."""
    return list(range(rank, n_utterances, world))


def gather_codes(local_codes: list[torch.Tensor], n_utterances: int, pad_value: int = 0, group=None) -> list[torch.Tensor] | None:
    """local_codes[j] = int64 [n_q, T_j] of the j-th utterance of this rank (shard_indices order).  Returns, on every
    rank, the list of all n_utterances code tensors in utterance order.  Works for world == 1 without a process group."""
    if not dist.is_available() or not dist.is_initialized():
        assert len(local_codes) == n_utterances
        return list(local_codes)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = local_codes[0].device if local_codes else torch.device("cpu")
    n_q = local_codes[0].shape[0] if local_codes else 0
    meta = torch.tensor([len(local_codes), n_q, max([c.shape[1] for c in local_codes], default=0)], dtype=torch.int64, device=dev)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    per_rank = max(int(m[0]) for m in metas)
    n_q = max(int(m[1]) for m in metas)
    t_max = max(int(m[2]) for m in metas)
    buf = torch.full((per_rank, n_q, t_max), pad_value, dtype=torch.int64, device=dev)
    lens = torch.zeros(per_rank, dtype=torch.int64, device=dev)
    for j, c in enumerate(local_codes):
        buf[j, :, : c.shape[1]] = c
        lens[j] = c.shape[1]
    bufs = [torch.empty_like(buf) for _ in range(world)]
    lenss = [torch.empty_like(lens) for _ in range(world)]
    dist.all_gather(bufs, buf, group=group)
    dist.all_gather(lenss, lens, group=group)
    out: list[torch.Tensor | None] = [None] * n_utterances
    for r in range(world):
        for j, i in enumerate(shard_indices(n_utterances, r, world)):
            out[i] = bufs[r][j, :, : int(lenss[r][j])].clone()
    assert all(o is not None for o in out)
    return out  # type: ignore[return-value]


def generate_sharded(generate_fn, conditionings: list[torch.Tensor], gather: bool = True, group=None):
    """Run `generate_fn(cond) -> int64 [1, n_q, T]` on this rank's share of `conditionings`; optionally gather."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    mine = [generate_fn(conditionings[i])[0] for i in shard_indices(len(conditionings), rank, world)]
    return gather_codes(mine, len(conditionings), group=group) if gather else mine
