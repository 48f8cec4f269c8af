"""Pins oracle/zonos_oracle.py against outputs of the reference itself (tests/golden/*.npz, produced by
tests/golden/make_golden.py in the build container).  Integer results must be bit-exact; floating-point
results are bit-exact too when the oracle runs the same torch CPU kernels as the reference did."""
import numpy as np
import pytest
import torch

from oracle import zonos_oracle as zo
from zonos_amd import synth


def _load(golden_dir, name):
    return np.load(f"{golden_dir}/{name}.npz")


@pytest.fixture(scope="module")
def tiny(golden_dir):
    g = _load(golden_dir, "tiny_gen")
    seed = int(g["seed"])
    w = synth.zonos_state_dict(synth.TINY_CFG, seed)
    cond = synth.conditioning(seed, "cond", 2, int(g["l_c"]), synth.TINY_CFG["d_model"])
    return g, w, cond, seed


def test_tiny_generate_matches_reference(tiny):
    g, w, cond, _ = tiny
    tr = zo.GenTrace()
    out = zo.generate(w, synth.TINY_CFG, cond, max_new_tokens=int(g["max_new"]), sampling_params={"temperature": 0.0}, trace=tr)
    assert np.array_equal(out.numpy(), g["out"].astype(np.int64))
    assert np.array_equal(torch.stack(tr.tokens).numpy(), g["tokens"].astype(np.int64))
    ref_logits = g["logits"]
    got = torch.stack(tr.logits).numpy()
    assert got.shape == ref_logits.shape
    assert np.array_equal(got, ref_logits), float(np.nanmax(np.abs(np.where(np.isfinite(ref_logits), got - ref_logits, 0))))


def test_tiny_generate_with_audio_prefix(golden_dir, tiny):
    _, w, cond, seed = tiny
    g = _load(golden_dir, "tiny_gen_prefix")
    pre = torch.from_numpy(synth.randint(seed, "prefix", (1, 9, int(g["prefix_len"])), 1024))
    tr = zo.GenTrace()
    out = zo.generate(w, synth.TINY_CFG, cond, audio_prefix_codes=pre, max_new_tokens=int(g["max_new"]),
                      sampling_params={"temperature": 0.0}, trace=tr)
    assert np.array_equal(out.numpy(), g["out"].astype(np.int64))
    assert np.array_equal(torch.stack(tr.logits).numpy(), g["logits"])


def _force(s):
    def hook(step, logits):
        if step == s:
            logits = logits.clone()
            logits[:, 0, 1024] = 1.0e4
        return logits
    return hook


def test_eos_cadence_and_truncation(golden_dir, tiny):
    """EOS bookkeeping, exit cadence (tensor_ops.py:90-103) and post-hoc truncation (model.py:513-529)."""
    _, w, cond, seed = tiny
    g = _load(golden_dir, "tiny_eos")
    for key in [k for k in g.files if k.startswith("out_")]:
        s = int(key.split("_")[1])
        tr = zo.GenTrace()
        out = zo.generate(w, synth.TINY_CFG, cond, max_new_tokens=int(g["max_new"]), sampling_params={"temperature": 0.0},
                          logits_hook=_force(s), trace=tr)
        assert out.shape == g[key].shape, (s, out.shape, g[key].shape)
        assert np.array_equal(out.numpy(), g[key].astype(np.int64)), s
        assert len(tr.tokens) == int(g[f"calls_{s}"]), s
    pre = torch.from_numpy(synth.randint(seed, "prefix", (1, 9, int(g["prefix_len"])), 1024))
    for key in [k for k in g.files if k.startswith("pout_")]:
        s = int(key.split("_")[1])
        out = zo.generate(w, synth.TINY_CFG, cond, audio_prefix_codes=pre, max_new_tokens=int(g["p_max_new"]),
                          sampling_params={"temperature": 0.0}, logits_hook=_force(s))
        assert np.array_equal(out.numpy(), g[key].astype(np.int64)), s


def test_sampling_transforms(golden_dir):
    g = _load(golden_dir, "sampling")
    seed = int(g["seed"])
    lg = torch.from_numpy(synth.normal(seed, "logits", (2, 9, 1025), 3.0))
    gen = torch.from_numpy(g["gen"])
    assert np.array_equal(zo.repetition_penalty(lg, gen, 3.0, 2).numpy(), g["rep"])
    pr = torch.softmax(lg, -1)
    assert np.array_equal(pr.numpy(), g["softmax"])
    assert np.array_equal(zo.unified(pr, 0.5, 0.4, 0.0).numpy(), g["unified"])
    assert np.array_equal(zo.unified(pr, 0.7, -0.1, 0.2).numpy(), g["unified_q"])
    assert np.array_equal(zo.top_p(pr, 0.8).numpy(), g["top_p"])
    assert np.array_equal(zo.top_k(pr, 50).numpy(), g["top_k"])
    assert np.array_equal(zo.min_p(pr, 0.1).numpy(), g["min_p"])
    got = zo.sample_from_logits(lg, temperature=0.0, generated_tokens=gen).squeeze(-1).numpy()
    assert np.array_equal(got, g["greedy"])


def test_delay_pattern(golden_dir):
    g = _load(golden_dir, "sampling")
    codes = torch.from_numpy(synth.randint(int(g["seed"]), "codes", (2, 9, 13), 1024))
    dl = zo.apply_delay_pattern(codes, 1025)
    assert np.array_equal(dl.numpy(), g["delayed"])
    assert np.array_equal(zo.revert_delay_pattern(dl).numpy(), g["reverted"])
    assert np.array_equal(zo.revert_delay_pattern(dl).numpy(), codes.numpy())


@pytest.fixture(scope="module")
def full_weights(golden_dir):
    g = _load(golden_dir, "full_gen")
    return synth.zonos_state_dict(synth.FULL_CFG, int(g["seed"]))


@pytest.mark.slow
def test_full_dims_generate_matches_reference(golden_dir, full_weights):
    """Zonos-v0.1-transformer dimensions, 64 new tokens, greedy: indices + recorded logits bit-exact."""
    g = _load(golden_dir, "full_gen")
    cond = synth.conditioning(int(g["seed"]), "cond", 2, int(g["l_c"]), synth.FULL_CFG["d_model"])
    tr = zo.GenTrace()
    out = zo.generate(full_weights, synth.FULL_CFG, cond, max_new_tokens=int(g["max_new"]),
                      sampling_params={"temperature": 0.0}, trace=tr)
    assert np.array_equal(out.numpy(), g["out"].astype(np.int64))
    assert np.array_equal(torch.stack(tr.tokens).numpy(), g["tokens"].astype(np.int64))
    for j, s in enumerate(g["logit_steps"]):
        assert np.array_equal(tr.logits[int(s)].numpy(), g["logits"][j]), int(s)


@pytest.mark.slow
def test_full_dims_layer0_decode(golden_dir, full_weights):
    """One decode step of block 0 over a synthetic KV history of length L-1 (reference TransformerBlock)."""
    g = _load(golden_dir, "full_layer0")
    seed, cfg = int(g["seed"]), synth.FULL_CFG
    for L in (1, 17, 900):
        x = synth.conditioning(seed, f"ops.x.{L}", 2, 1, cfg["d_model"])
        kv = torch.from_numpy(synth.normal(seed, f"ops.kv.{L}", (2, 904, 2, 4, 128))).to(torch.bfloat16)
        cache = zo.Cache([kv], 904, L - 1, torch.full((2,), L - 1, dtype=torch.int32), zo.rope_table(16384, 128))
        cs = cache.rope[cache.lengths.long().unsqueeze(-1)]
        y = zo.layer_forward(full_weights, 0, x, cache, cs, cfg)
        assert np.array_equal(y.view(torch.int16).numpy(), g[f"y_{L}"]), L
        assert np.array_equal(kv[:, L - 1, 0].contiguous().view(torch.int16).numpy(), g[f"knew_{L}"]), L


def test_peaky_head_generate_matches_reference(golden_dir, full_weights):
    """Decisive-margin synthetic heads (synth.peaky_heads): the free-running parity cases of the GPU suite."""
    g = _load(golden_dir, "tiny_gen_peaky")
    w = synth.zonos_state_dict(synth.TINY_CFG, int(g["seed"]), peaky=True)
    cond = synth.conditioning(int(g["seed"]), "cond", 2, int(g["l_c"]), synth.TINY_CFG["d_model"])
    out = zo.generate(w, synth.TINY_CFG, cond, max_new_tokens=int(g["max_new"]), sampling_params={"temperature": 0.0})
    assert np.array_equal(out.numpy(), g["out"].astype(np.int64))
    g = _load(golden_dir, "full_gen_peaky")
    w = dict(full_weights)
    w["fused_heads.weight"] = torch.cat([torch.from_numpy(synth.peaky_heads(int(g["seed"]), f"heads.{i}.weight", 1025, 2048)).to(torch.bfloat16)
                                         for i in range(9)], 0)
    cond = synth.conditioning(int(g["seed"]), "cond", 2, int(g["l_c"]), 2048)
    out = zo.generate(w, synth.FULL_CFG, cond, max_new_tokens=int(g["max_new"]), sampling_params={"temperature": 0.0})
    assert np.array_equal(out.numpy(), g["out"].astype(np.int64))


def test_dac_decode_matches_transformers(golden_dir):
    g = _load(golden_dir, "dac")
    seed = int(g["seed"])
    dw = synth.dac_state_dict(seed)
    for T in (16, 40):
        codes = torch.from_numpy(synth.randint(seed, f"codes{T}", (1, 9, T), 1024))
        col = {}
        wav = zo.dac_decode(dw, codes, collect=col)
        assert wav.shape == (1, 1, 512 * T)
        ref = g[f"wav_{T}"]
        err = float(np.sqrt(np.mean((wav[:, 0].numpy() - ref) ** 2)))
        assert err < 1e-6, err
        if T == 16:
            assert abs(float(col["conv1"].pow(2).mean().sqrt()) - float(g["rms_conv1"])) < 1e-5
            for bi in range(4):
                assert abs(float(col[f"block{bi}"].pow(2).mean().sqrt()) - float(g[f"rms_block{bi}"])) < 1e-4 * max(1.0, float(g[f"rms_block{bi}"]))


def test_dac_encode_matches_transformers(golden_dir):
    """oracle dac_encode vs transformers DacModel.encode (the call at zonos/autoencoder.py:117) on synthetic weights and
    waveforms: codes bit-exact, encoder latents equal."""
    g = _load(golden_dir, "dac_encode")
    seed = int(g["seed"])
    dw = synth.dac_state_dict(seed)
    for T in (512 * 6, 512 * 23):
        wav = synth.test_waveform(seed, f"encwav{T}", T)
        col = {}
        codes = zo.dac_encode(dw, wav, collect=col)
        assert codes.shape == (1, 9, T // 512)
        assert np.array_equal(codes.numpy(), g[f"codes_{T}"].astype(np.int64))
        assert np.abs(col["z"].numpy() - g[f"z_{T}"]).max() <= 1e-6


def test_speaker_embedding_matches_reference(golden_dir):
    """oracle speaker_embed vs the reference's ResNet293_based + LDA Linear (zonos/speaker_cloning.py) on synthetic weights
    and features (goldens recorded by importing the reference class): fp32, same torch ops -> equal to 1e-5 relative."""
    g = _load(golden_dir, "speaker")
    seed = int(g["seed"])
    sd, lda = synth.speaker_state_dict(seed)
    T = 64
    feats = synth.speaker_features(seed, f"feats{T}", 1, 80, T)
    emb, ld = zo.speaker_embed(sd, feats, lda)
    for got, ref in ((emb.numpy(), g[f"emb_{T}"]), (ld.numpy(), g[f"lda_{T}"])):
        rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        assert got.shape == ref.shape and rel < 1e-5, rel
