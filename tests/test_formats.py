"""Checkpoint / wire formats either side of the hot path: `Zonos.from_local` must accept the
reference's safetensors key contract — per-codebook `heads.{i}.weight` [1025, d] (fused on load, zonos/model.py:208-223),
`embeddings.{i}.weight` with 1026 rows zero-padded to 1032 (model.py:164-170), `backbone.layers.*` names — plus
config.json via ZonosConfig.from_dict (zonos/config.py:128-149).  CPU: loader logic; GPU: the loaded model generates the
same codes as the directly-built one and `decode_to_int16` yields PCM16."""
import json

import numpy as np
import pytest
import torch
from safetensors.torch import save_file

from zonos_amd import synth
from zonos_amd.testing import _attn_cfg
from zonos_amd.config import ZonosConfig
from zonos_amd.model import Zonos


def _write_checkpoint(tmp_path, cfg, seed, drop=(), extra=None):
    sd = synth.zonos_state_dict(cfg, seed)
    ck = {}
    for k, v in sd.items():
        if k == "fused_heads.weight":
            for i in range(9):
                ck[f"heads.{i}.weight"] = v[i * 1025:(i + 1) * 1025].clone()
        elif k.startswith("embeddings."):
            ck[k] = v[:1026].clone()                  # checkpoints carry 1026 rows (1024 + EOS + MASK)
        else:
            ck[k] = v.clone()
    ck["prefix_conditioner.norm.weight"] = torch.ones(cfg["d_model"], dtype=torch.bfloat16)
    ck["prefix_conditioner.norm.bias"] = torch.zeros(cfg["d_model"], dtype=torch.bfloat16)
    for k in drop:
        del ck[k]
    ck.update(extra or {})
    save_file(ck, str(tmp_path / "model.safetensors"))
    conf = {"backbone": {"d_model": cfg["d_model"], "n_layer": cfg["n_layer"], "attn_mlp_d_intermediate": cfg["d_ff"], "d_intermediate": 0,
                         "ssm_cfg": dict(cfg.get("ssm_cfg") or {}), "attn_layer_idx": list(cfg.get("attn_layer_idx", range(cfg["n_layer"]))),
                         "attn_cfg": ({"num_heads": cfg["num_heads"], "num_heads_kv": cfg["num_heads_kv"], "causal": True, "rotary_emb_dim": 32}
                                      if not cfg.get("ssm_cfg") else _attn_cfg(cfg)),
                         "rms_norm": bool(cfg.get("rms_norm", False)), "residual_in_fp32": bool(cfg.get("residual_in_fp32", False)), "norm_epsilon": 1e-5},
            "prefix_conditioner": {"conditioners": [], "projection": "none"},
            "eos_token_id": 1024, "masked_token_id": 1025, "pad_vocab_to_multiple_of": 8}
    (tmp_path / "config.json").write_text(json.dumps(conf))
    return sd


def test_from_local_key_contract_cpu(tmp_path):
    cfg = synth.TINY_CFG
    sd = _write_checkpoint(tmp_path, cfg, 77)
    zc = ZonosConfig.from_dict(json.load(open(tmp_path / "config.json")))
    assert zc.backbone.attn_cfg["num_heads_kv"] == cfg["num_heads_kv"] and zc.codebook_dimension == 9
    model = Zonos.from_local(str(tmp_path / "config.json"), str(tmp_path / "model.safetensors"), device="cpu")
    got = model.state_dict()
    assert got["fused_heads.weight"].shape == (9 * 1025, cfg["d_model"]) and got["fused_heads.weight"].dtype == torch.bfloat16
    assert torch.equal(got["fused_heads.weight"], sd["fused_heads.weight"])
    for i in range(9):
        e = got[f"embeddings.{i}.weight"]
        assert e.shape == (1032, cfg["d_model"])
        assert torch.equal(e[:1026], sd[f"embeddings.{i}.weight"][:1026]) and bool((e[1026:] == 0).all())
    for k, v in sd.items():
        if k.startswith("backbone."):
            assert torch.equal(got[k], v), k
    other = json.load(open(tmp_path / "config.json"))
    other["backbone"]["ssm_cfg"] = {"layer": "Mamba1"}
    (tmp_path / "mamba1.json").write_text(json.dumps(other))
    with pytest.raises(Exception, match="Mamba2"):
        Zonos.from_local(str(tmp_path / "mamba1.json"), str(tmp_path / "model.safetensors"), device="cpu")


def test_from_local_hybrid_attn_cfg_forms_cpu(tmp_path):
    """attn_cfg goes to mamba_ssm's MHA in the reference (_mamba_ssm.py:45-58): its defaults (biases on, no rotary) apply to
    the keys config.json leaves out.  A checkpoint written for those defaults loads (bias tensors and all); the same
    checkpoint against a config that switches the biases off is refused; unsupported forms are refused at construction."""
    from zonos_amd._lib import ZonosHipError
    cfg = dict(synth.HYBRID_TINY_CFG, attn_cfg={"causal": True}, rms_norm=True, residual_in_fp32=True)    # library defaults otherwise
    sd = _write_checkpoint(tmp_path, cfg, 79)
    assert "backbone.layers.2.mixer.in_proj.bias" in sd and "backbone.layers.2.norm.bias" not in sd and "backbone.layers.0.norm.bias" not in sd
    model = Zonos.from_local(str(tmp_path / "config.json"), str(tmp_path / "model.safetensors"), device="cpu")
    got = model.state_dict()
    assert torch.equal(got["backbone.layers.2.mixer.in_proj.bias"], sd["backbone.layers.2.mixer.in_proj.bias"])
    assert torch.equal(got["backbone.layers.2.mixer.out_proj.bias"], sd["backbone.layers.2.mixer.out_proj.bias"])
    conf = json.load(open(tmp_path / "config.json"))
    conf["backbone"]["attn_cfg"]["qkv_proj_bias"] = False
    (tmp_path / "nobias.json").write_text(json.dumps(conf))
    with pytest.raises(ZonosHipError, match="unexpected tensors.*in_proj.bias"):
        Zonos.from_local(str(tmp_path / "nobias.json"), str(tmp_path / "model.safetensors"), device="cpu")
    for bad in ({"rotary_emb_dim": 16}, {"causal": False}, {"softmax_scale": 0.5}, {"d_conv": 4}, {"window_size": 3}):
        conf = json.load(open(tmp_path / "config.json"))
        conf["backbone"]["attn_cfg"].update(bad)
        (tmp_path / "bad.json").write_text(json.dumps(conf))
        with pytest.raises(ZonosHipError):
            Zonos.from_local(str(tmp_path / "bad.json"), str(tmp_path / "model.safetensors"), device="cpu")


def test_from_local_refuses_mismatched_checkpoints(tmp_path):
    """A tensor the model does not know (the reference's strict load fails on it, model.py:174 — e.g. the bias tensors of
    an attention configuration this backbone does not implement) or a tensor the file lacks is an error, never a silent
    partial load."""
    from zonos_amd._lib import ZonosHipError
    cfg = synth.TINY_CFG
    _write_checkpoint(tmp_path, cfg, 77, extra={"backbone.layers.0.mixer.in_proj.bias": torch.zeros(8, dtype=torch.bfloat16)})
    with pytest.raises(ZonosHipError, match="unexpected tensors.*in_proj.bias"):
        Zonos.from_local(str(tmp_path / "config.json"), str(tmp_path / "model.safetensors"), device="cpu")
    _write_checkpoint(tmp_path, cfg, 77, drop=("backbone.layers.1.mlp.fc2.weight",))
    with pytest.raises(ZonosHipError, match="missing tensors.*fc2.weight"):
        Zonos.from_local(str(tmp_path / "config.json"), str(tmp_path / "model.safetensors"), device="cpu")


def test_from_local_hybrid_key_contract_cpu(tmp_path):
    """A hybrid checkpoint (mamba_ssm parameter names under layers.{i}.mixer, _mamba_ssm.py:43-60) loads key for key."""
    cfg = synth.HYBRID_TINY_CFG
    sd = _write_checkpoint(tmp_path, cfg, 78)
    model = Zonos.from_local(str(tmp_path / "config.json"), str(tmp_path / "model.safetensors"), device="cpu")
    got = model.state_dict()
    n_backbone = 0
    for k, v in sd.items():
        if k.startswith("backbone."):
            assert k in got and got[k].shape == v.shape and torch.equal(got[k], v), k
            n_backbone += 1
    assert n_backbone == sum(k.startswith("backbone.") for k in got)      # no parameter of the module tree is left unset
    assert got["backbone.layers.0.mixer.conv1d.weight"].shape == (256 + 2 * 64, 1, 4)
    assert "backbone.layers.2.mlp.fc1.weight" in got and "backbone.layers.0.mlp.fc1.weight" not in got


@pytest.mark.gpu
def test_from_local_generates_like_direct_build(tmp_path):
    from zonos_amd.testing import build_model
    cfg = synth.TINY_CFG
    _write_checkpoint(tmp_path, cfg, 77)
    loaded = Zonos.from_local(str(tmp_path / "config.json"), str(tmp_path / "model.safetensors"), device="cuda:0")
    direct, _ = build_model(cfg, 77, "cuda:0")
    cond = synth.conditioning(77, "cond", 2, 6, cfg["d_model"]).to("cuda:0")
    a = loaded.generate(cond, max_new_tokens=16, sampling_params={"temperature": 0.0})
    b = direct.generate(cond, max_new_tokens=16, sampling_params={"temperature": 0.0})
    # rows 1026..1031 of the embedding tables differ (zero padding vs synthetic) but are never indexed (codes <= 1025)
    assert torch.equal(a, b)
    assert a.dtype == torch.int64 and a.shape[:2] == (1, 9)


def test_wav_writer_and_tensor_cache_files(tmp_path):
    """Output-side wire formats: a PCM_S 16-bit WAV like utilities/cache_utils.py:380-390 writes, from
    both the int16 [T, 1] tensor of decode_to_int16 and a float waveform (same clamp / x32767 / truncate conversion,
    autoencoder.py:142-170), and the `.pt` cache files of cache_utils.py:322-362 (plain torch.save of a tensor, loaded
    with weights_only=True)."""
    import wave

    from zonos_amd.utils import load_tensor_cache, save_tensor_cache, save_wav_pcm16

    f = torch.tensor([[0.0, 0.5, -0.5, 1.0, -1.0, 1.5, -2.0, 3.0517578125e-05, 0.99999]])
    expect = (f.clamp(-1, 1) * 32767.0).to(torch.int16)[0]
    p1 = save_wav_pcm16(tmp_path / "f.wav", f, 44100)
    p2 = save_wav_pcm16(tmp_path / "i.wav", expect.unsqueeze(1), 44100)
    for p in (p1, p2):
        with wave.open(p, "rb") as w:
            assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 44100, f.shape[1])
            got = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
        assert np.array_equal(got, expect.numpy())
    emb = synth.conditioning(3, "cache.emb", 1, 1, 128)
    codes = torch.from_numpy(synth.randint(3, "cache.codes", (1, 9, 40), 1024))
    for name, t in (("emb", emb), ("codes", codes)):
        path = save_tensor_cache(tmp_path / f"{name}.pt", t)
        back = load_tensor_cache(path, device="cpu")
        assert back.dtype == t.dtype and torch.equal(back, t)
        assert torch.equal(torch.load(path, map_location="cpu", weights_only=True), t)      # the reference's loader call
    assert load_tensor_cache(tmp_path / "missing.pt") is None
