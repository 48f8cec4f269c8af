"""Codebook delay pattern (zonos/codebook_pattern.py:5-61).  Index bookkeeping done once per utterance on
whatever device the codes live on; no kernel needed."""
from __future__ import annotations

import torch


def apply_delay_pattern(codes: torch.Tensor, mask_token: int) -> torch.Tensor:
    """[B, n_q, T] -> [B, n_q, T + n_q]: codebook k shifted right by k + 1, gaps filled with mask_token."""
    b, n_q, t = codes.shape
    out = codes.new_full((b, n_q, t + n_q), mask_token)
    for k in range(n_q):
        out[:, k, k + 1:k + 1 + t] = codes[:, k]
    return out


def revert_delay_pattern(codes: torch.Tensor) -> torch.Tensor:
    """[B, n_q, T + n_q] -> [B, n_q, T]."""
    _, n_q, t = codes.shape
    return torch.stack([codes[:, k, k + 1:t - n_q + k + 1] for k in range(n_q)], dim=1)
