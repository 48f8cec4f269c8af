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


def generate_sharded(generate_fn, conditionings: list[torch.Tensor], gather: bool = True, group=None, batch_size: int = 1):
    """Run `generate_fn` on this rank's share of `conditionings` (each [2, L_c, d] = [cond ‖ uncond] of one utterance);
    optionally gather.  batch_size == 1: `generate_fn(cond) -> int64 [1, n_q, T]` per utterance.  batch_size > 1: the
    share runs in groups of up to `batch_size` utterances, `generate_fn(cond [2b, L_c, d], b) -> int64 [b, n_q, T]`
    with rows [cond_0..cond_{b-1}, uncond_0..uncond_{b-1}] (Zonos.generate's batch layout)."""
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    share = shard_indices(len(conditionings), rank, world)
    if batch_size <= 1:
        mine = [generate_fn(conditionings[i])[0] for i in share]
    else:
        mine = []
        for j in range(0, len(share), batch_size):
            grp = [conditionings[i] for i in share[j:j + batch_size]]
            cond = torch.cat([c[0:1] for c in grp] + [c[1:2] for c in grp], dim=0)
            codes = generate_fn(cond, len(grp))
            mine.extend(codes[b] for b in range(len(grp)))
    return gather_codes(mine, len(conditionings), group=group) if gather else mine
