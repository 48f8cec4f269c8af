"""Helpers shared by tests, bench.py and smoke(): build a `Zonos` around synthetic weights (zonos_amd/synth.py)."""
from __future__ import annotations

import torch

from . import synth
from .autoencoder import DACAutoencoder
from .config import BackboneConfig, PrefixConditionerConfig, ZonosConfig
from .model import Zonos


def _attn_cfg(cfg: dict) -> dict:
    """config.json's backbone.attn_cfg for a synthetic configuration.  Hybrid configurations without an "attn_cfg" entry keep
    the attention form of the first hybrid tests (interleaved rotary over the whole head, no biases), spelled out here
    because mamba_ssm's MHA defaults differ (zonos_amd/backbone/_hip.py attention_options)."""
    base = dict(num_heads=cfg["num_heads"], num_heads_kv=cfg["num_heads_kv"])
    if not cfg.get("ssm_cfg"):
        return base
    if "attn_cfg" in cfg:
        return {**base, **cfg["attn_cfg"]}
    return {**base, "causal": True, "rotary_emb_dim": cfg["d_model"] // cfg["num_heads"], "rotary_emb_interleaved": True,
            "qkv_proj_bias": False, "out_proj_bias": False}


def zonos_config(cfg: dict, conditioners: list | None = None, projection: str = "none") -> ZonosConfig:
    return ZonosConfig(BackboneConfig(d_model=cfg["d_model"], n_layer=cfg["n_layer"], attn_mlp_d_intermediate=cfg["d_ff"],
                                      attn_layer_idx=list(cfg.get("attn_layer_idx", range(cfg["n_layer"]))), ssm_cfg=dict(cfg.get("ssm_cfg") or {}),
                                      attn_cfg=_attn_cfg(cfg), rms_norm=bool(cfg.get("rms_norm", False)),
                                      residual_in_fp32=bool(cfg.get("residual_in_fp32", False))),
                       PrefixConditionerConfig(list(conditioners or []), projection))


def build_model(cfg: dict, seed: int, device="cuda", dac: DACAutoencoder | None = None, peaky: bool = False, conditioners: list | None = None,
                projection: str = "none"):
    """Returns (model on `device`, CPU state dict).  The state dict uses the reference's key contract."""
    sd = synth.zonos_state_dict(cfg, seed, peaky=peaky)
    sd.update({"prefix_conditioner." + k: v for k, v in synth.conditioner_state_dict(conditioners or [], cfg["d_model"], seed, projection).items()})
    with torch.device("meta"):
        model = Zonos(zonos_config(cfg, conditioners, projection), autoencoder=dac or DACAutoencoder())
    model.load_state_dict({k: v for k, v in sd.items()}, assign=True, strict=True)
    model = model.to(device)
    return model.eval(), sd
