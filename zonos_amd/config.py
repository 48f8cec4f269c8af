"""Data contract of the hot path (mirrors zonos/config.py:9-149 field for field so config.json files and
callers written against the reference keep working)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Literal

import torch


@dataclass
class InferenceParams:
    """zonos/config.py:9-52.  key_value_memory_dict maps layer -> (kv [R, maxL, 2, Hkv, hd] bf16, None)."""
    max_seqlen: int
    max_batch_size: int
    seqlen_offset: int = 0
    batch_size_offset: int = 0
    key_value_memory_dict: dict = field(default_factory=dict)
    lengths_per_sample: torch.Tensor | None = None

    def reset(self, max_seqlen, max_batch_size):
        self.max_seqlen, self.max_batch_size, self.seqlen_offset = max_seqlen, max_batch_size, 0
        if self.lengths_per_sample is not None:
            self.lengths_per_sample.zero_()


@dataclass
class BackboneConfig:
    """zonos/config.py:55-84"""
    d_model: int = 1024
    d_intermediate: int = 0
    attn_mlp_d_intermediate: int = 0
    n_layer: int = 16
    ssm_cfg: dict = field(default_factory=dict)
    attn_layer_idx: list = field(default_factory=list)
    attn_cfg: dict = field(default_factory=dict)
    rms_norm: bool = False
    residual_in_fp32: bool = False
    norm_epsilon: float = 1e-5


@dataclass
class PrefixConditionerConfig:
    """zonos/config.py:87-102"""
    conditioners: list[dict]
    projection: Literal["none", "linear", "mlp"]


@dataclass
class ZonosConfig:
    """zonos/config.py:105-149"""
    backbone: BackboneConfig
    prefix_conditioner: PrefixConditionerConfig
    eos_token_id: int = 1024
    masked_token_id: int = 1025
    pad_vocab_to_multiple_of: int = 8
    codebook_dimension: int = 9

    @classmethod
    def from_dict(cls, d: dict) -> "ZonosConfig":
        d = dict(d)
        return cls(BackboneConfig(**d.pop("backbone")), PrefixConditionerConfig(**d.pop("prefix_conditioner")), **d)
