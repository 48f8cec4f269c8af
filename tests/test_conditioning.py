"""Prefix conditioner numeric path: oracle and HIP vs outputs of the reference's own
`PrefixConditioner` + `make_cond_dict` + `prepare_conditioning_with_cache` (tests/golden/conditioner.npz: transformer
conditioner list of CONDITIONING_README.md, bf16, synthetic weights; phonemisation bypassed with a given phoneme string)."""
import numpy as np
import pytest
import torch

from oracle import zonos_oracle as zo
from zonos_amd import conditioning as zc
from zonos_amd import synth

CASES = [(128, "none"), (2048, "none"), (128, "mlp")]


def _inputs(g, tag, d, seed):
    spk = torch.from_numpy(synth.normal(seed, "cond.speaker", (1, 1, 128))).to(torch.bfloat16)
    return spk, dict(text="ignored", language="en-us", speaker=spk, emotion=[0.5, 0.05, 0.05, 0.05, 0.05, 0.05, 0.1, 0.15],
                     fmax=22050.0, pitch_std=45.0, speaking_rate=13.0)


def test_tokenizer_and_cond_dict_match_reference(golden_dir):
    g = np.load(f"{golden_dir}/conditioner.npz")
    ids, lengths = zc.tokenize_phonemes([str(g["phonemes"])])
    assert np.array_equal(ids.numpy(), g["d128_none_ids"]) and lengths == [ids.shape[1]]
    _, kw = _inputs(g, "d128_none", 128, 77)
    cd = zc.make_cond_dict(device="cpu", **kw)
    assert np.array_equal(cd["emotion"].numpy(), g["d128_none_emotion"]) and np.array_equal(cd["language_id"].numpy(), g["d128_none_langid"])
    assert "vqscore_8" not in cd and "dnsmos_ovrl" not in cd and cd["fmax"].shape == (1, 1, 1)
    with pytest.raises(AssertionError):
        zc.make_cond_dict(language="xx-unknown", device="cpu")
    # LRU cache semantics (conditioning_cache.py:56-136)
    c = zc.ConditioningCache(max_size=2)
    c.put("a", torch.zeros(1)); c.put("b", torch.ones(1)); c.get("a"); c.put("c", torch.ones(1))
    assert c.get("b") is None and c.get("a") is not None and c.size() == 2


@pytest.mark.parametrize("d,projection", CASES)
def test_oracle_prefix_conditioner_matches_reference(golden_dir, d, projection):
    g = np.load(f"{golden_dir}/conditioner.npz")
    tag = f"d{d}_{projection}"
    seed = int(g[tag + "_seed"])
    pw = synth.conditioner_state_dict(synth.TRANSFORMER_CONDITIONERS, d, seed, projection)
    _, kw = _inputs(g, tag, d, seed)
    cd = zc.make_cond_dict(device="cpu", **kw)
    cd["espeak"] = torch.from_numpy(g[tag + "_ids"])
    out = zo.prepare_conditioning(pw, synth.TRANSFORMER_CONDITIONERS, projection, cd, d, cfg_scale=2.0)
    ref = g[tag]
    assert out.shape == ref.shape and out.shape[0] == 2 and out.shape[2] == d
    assert np.array_equal(out.contiguous().view(torch.int16).numpy(), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("d,projection", CASES)
def test_hip_prepare_conditioning_matches_reference(golden_dir, d, projection):
    """Zonos.prepare_conditioning on MI355X vs the reference: gathers and uncond vectors exact; Linear / Fourier / LayerNorm
    follow the reference's rounding points -> >= 99.5 % of the bf16 outputs bit-equal, max error one bf16 ulp."""
    from zonos_amd.testing import build_model
    g = np.load(f"{golden_dir}/conditioner.npz")
    tag = f"d{d}_{projection}"
    seed = int(g[tag + "_seed"])
    cfg = dict(synth.TINY_CFG if d == 128 else synth.FULL_CFG)
    if d != 128:
        cfg["n_layer"] = 1                      # the backbone is irrelevant here; keep the model small
    model, _ = build_model(cfg, seed, "cuda:0", conditioners=synth.TRANSFORMER_CONDITIONERS, projection=projection)
    _, kw = _inputs(g, tag, d, seed)
    cd = zc.make_cond_dict(device="cuda:0", **kw)
    cd["espeak"] = ("ids", torch.from_numpy(g[tag + "_ids"]))
    out = model.prepare_conditioning(cd, cfg_scale=2.0)
    again = model.prepare_conditioning(cd, cfg_scale=2.0, use_cache=True)
    hit = model.prepare_conditioning(cd, cfg_scale=2.0, use_cache=True)
    assert hit is again and torch.equal(out, again)
    ref = torch.from_numpy(g[tag]).view(torch.bfloat16)
    got = out.cpu()
    assert got.shape == ref.shape and got.dtype == torch.bfloat16
    eq = (got.view(torch.int16) == ref.view(torch.int16)).float().mean().item()
    err = (got.float() - ref.float()).abs().max().item()
    print(f"\n[conditioner {tag}] shape {tuple(got.shape)} bit-equal {eq:.5f} max|d| {err:.4g}")
    assert eq >= 0.995 and err <= 0.0625
    assert set(c.name for c in model.prefix_conditioner.conditioners) >= {"espeak", "speaker", "emotion", "language_id"}
    # the conditioning drives generate() end to end
    if d == 128:
        codes = model.generate(out, max_new_tokens=6, sampling_params={"temperature": 0.0})
        assert codes.shape[:2] == (1, 9)
