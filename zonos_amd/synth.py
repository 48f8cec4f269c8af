"""Deterministic synthetic weights for parity tests and benchmarks.

No checkpoint exists offline, so every parity/bench run uses seeded synthetic
weights at the real architecture dimensions.  Weights are never committed: both sides of a
comparison (this container, the GPU box, the golden-fixture generator) regenerate them from
this counter-based generator, which uses only exact integer arithmetic and a single IEEE fp32
multiply, so the bits do not depend on the numpy/torch version or the CPU.

Stream definition (all arithmetic mod 2**64):
    key    = splitmix64(seed ^ fnv1a64(tensor_name))
    u64(i) = splitmix64(key + (i + 1) * 0x9E3779B97F4A7C15)
    uniform: top 24 bits k -> (k - 2**23 + 0.5) / 2**23           in (-1, 1), exact in fp32
    normal : four 16-bit fields a,b,c,d -> (a+b+c+d - 2*65535) * (sqrt(3)/65536)   (Irwin-Hall)
"""
from __future__ import annotations

import numpy as np
import torch

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _splitmix64(z: np.ndarray) -> np.ndarray:
    z = (z ^ (z >> np.uint64(30))) * _M1
    z = (z ^ (z >> np.uint64(27))) * _M2
    return z ^ (z >> np.uint64(31))


def fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _raw(seed: int, name: str, n: int, start: int = 0) -> np.ndarray:
    with np.errstate(over="ignore"):
        key = _splitmix64(np.array([(seed ^ fnv1a64(name)) & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64))[0]
        idx = np.arange(start + 1, start + n + 1, dtype=np.uint64)
        return _splitmix64(key + idx * _GAMMA)


_CHUNK = 1 << 24


def uniform(seed: int, name: str, shape, scale: float) -> np.ndarray:
    """fp32 array, U(-scale, scale)."""
    n = int(np.prod(shape))
    out = np.empty(n, dtype=np.float32)
    for s in range(0, n, _CHUNK):
        m = min(_CHUNK, n - s)
        k = (_raw(seed, name, m, s) >> np.uint64(40)).astype(np.int64)
        v = (k - (1 << 23)).astype(np.float32) + np.float32(0.5)
        out[s:s + m] = v * np.float32(scale / float(1 << 23))
    return out.reshape(shape)


_IH = np.float32(np.sqrt(3.0) / 65536.0)


def normal(seed: int, name: str, shape, std: float = 1.0) -> np.ndarray:
    """fp32 array, approximately N(0, std^2) (sum of four 16-bit uniforms)."""
    n = int(np.prod(shape))
    out = np.empty(n, dtype=np.float32)
    for s in range(0, n, _CHUNK):
        m = min(_CHUNK, n - s)
        r = _raw(seed, name, m, s)
        t = ((r & np.uint64(0xFFFF)) + ((r >> np.uint64(16)) & np.uint64(0xFFFF))
             + ((r >> np.uint64(32)) & np.uint64(0xFFFF)) + (r >> np.uint64(48))).astype(np.int64)
        out[s:s + m] = (t - 2 * 65535).astype(np.float32) * np.float32(_IH * np.float32(std))
    return out.reshape(shape)


def randint(seed: int, name: str, shape, high: int) -> np.ndarray:
    """int64 array uniform in [0, high)."""
    n = int(np.prod(shape))
    r = _raw(seed, name, n)
    return ((r >> np.uint64(33)) % np.uint64(high)).astype(np.int64).reshape(shape)


def _t(a: np.ndarray, dtype) -> torch.Tensor:
    return torch.from_numpy(a).to(dtype)


def peaky_heads(seed: int, name: str, rows: int, d: int, nnz: int = 4) -> np.ndarray:
    """Sparse heavy-tailed head rows: `nnz` non-zeros per row at hashed columns, magnitude log-uniform in [1, 20].
    Logits then have decisive top-2 margins (like a trained model) instead of the near-ties of Gaussian logits,
    which makes free-running greedy decode a meaningful bit-exactness test."""
    w = np.zeros((rows, d), dtype=np.float32)
    cols = randint(seed, name + ".cols", (rows, nnz), d)
    mag = np.exp(np.abs(uniform(seed, name + ".mag", (rows, nnz), 1.0)).astype(np.float64) * np.log(20.0)).astype(np.float32)
    sgn = np.where(uniform(seed, name + ".sgn", (rows, nnz), 1.0) < 0, -1.0, 1.0).astype(np.float32)
    for j in range(nnz):
        w[np.arange(rows), cols[:, j]] = mag[:, j] * sgn[:, j]
    return w


def zonos_state_dict(cfg: dict, seed: int = 1234, dtype=torch.bfloat16, peaky: bool = False) -> dict:
    """Synthetic transformer-backbone weights under the reference's safetensors key contract
    (module names at zonos/backbone/_torch.py:154-155,278-281,373-374,453-454,
    zonos/model.py:81-82).  Scales follow PyTorch defaults: Linear U(+-1/sqrt(in)), Embedding
    N(0,1), LayerNorm weight 1+-0.1 / bias +-0.1 (perturbed so that affine bugs show up).

    cfg keys: d_model, n_layer, num_heads, num_heads_kv, d_ff, n_codebooks, vocab_embed, vocab_head
    """
    d, L, H, Hkv, F = cfg["d_model"], cfg["n_layer"], cfg["num_heads"], cfg["num_heads_kv"], cfg["d_ff"]
    hd = d // H
    nq = cfg.get("n_codebooks", 9)
    ve, vh = cfg.get("vocab_embed", 1032), cfg.get("vocab_head", 1025)
    sd = {}
    for i in range(nq):
        sd[f"embeddings.{i}.weight"] = _t(normal(seed, f"embeddings.{i}.weight", (ve, d)), dtype)
    if peaky:
        sd["fused_heads.weight"] = torch.cat([_t(peaky_heads(seed, f"heads.{i}.weight", vh, d), dtype) for i in range(nq)], 0)
    else:
        sd["fused_heads.weight"] = torch.cat(
            [_t(uniform(seed, f"heads.{i}.weight", (vh, d), 1.0 / np.sqrt(d)), dtype) for i in range(nq)], 0)
    ssm = cfg.get("ssm_cfg") or {}
    rms = bool(ssm) and bool(cfg.get("rms_norm"))              # mamba_ssm RMSNorm blocks carry no bias
    ac = cfg.get("attn_cfg") or {}
    qkv_bias = bool(ssm) and bool(ac.get("qkv_proj_bias", True)) and "attn_cfg" in cfg      # mamba_ssm MHA defaults (attn_cfg given)
    out_bias = bool(ssm) and bool(ac.get("out_proj_bias", True)) and "attn_cfg" in cfg
    for l in range(L):
        p = f"backbone.layers.{l}."
        if ssm and l not in cfg["attn_layer_idx"]:
            sd.update(mamba2_layer_state_dict(cfg, seed, p, dtype))
            continue
        for nm in ("norm", "norm2"):
            sd[p + nm + ".weight"] = _t(1.0 + uniform(seed, p + nm + ".weight", (d,), 0.1), dtype)
            if not rms:
                sd[p + nm + ".bias"] = _t(uniform(seed, p + nm + ".bias", (d,), 0.1), dtype)
        sd[p + "mixer.in_proj.weight"] = _t(uniform(seed, p + "mixer.in_proj.weight", ((H + 2 * Hkv) * hd, d), 1.0 / np.sqrt(d)), dtype)
        sd[p + "mixer.out_proj.weight"] = _t(uniform(seed, p + "mixer.out_proj.weight", (d, H * hd), 1.0 / np.sqrt(H * hd)), dtype)
        if qkv_bias:
            sd[p + "mixer.in_proj.bias"] = _t(uniform(seed, p + "mixer.in_proj.bias", ((H + 2 * Hkv) * hd,), 0.2), dtype)
        if out_bias:
            sd[p + "mixer.out_proj.bias"] = _t(uniform(seed, p + "mixer.out_proj.bias", (d,), 0.2), dtype)
        sd[p + "mlp.fc1.weight"] = _t(uniform(seed, p + "mlp.fc1.weight", (2 * F, d), 1.0 / np.sqrt(d)), dtype)
        sd[p + "mlp.fc2.weight"] = _t(uniform(seed, p + "mlp.fc2.weight", (d, F), 1.0 / np.sqrt(F)), dtype)
    sd["backbone.norm_f.weight"] = _t(1.0 + uniform(seed, "backbone.norm_f.weight", (d,), 0.1), dtype)
    sd["backbone.norm_f.bias"] = _t(uniform(seed, "backbone.norm_f.bias", (d,), 0.1), dtype)
    return sd


def mamba2_layer_state_dict(cfg: dict, seed: int, p: str, dtype=torch.bfloat16) -> dict:
    """Synthetic Mamba2 block under mamba_ssm's parameter names (norm + mixer.{in_proj, conv1d, dt_bias, A_log, D, norm,
    out_proj}) with the library's init distributions: dt = exp(U(log 1e-3, log 1e-1)) -> dt_bias = softplus^-1(dt),
    A = U(1, 16) -> A_log, D = 1 (perturbed), conv/Linear U(+-1/sqrt(fan_in))."""
    d = cfg["d_model"]
    sc = cfg["ssm_cfg"]
    d_inner = int(sc.get("expand", 2)) * d
    headdim, d_state, ngroups, d_conv = int(sc.get("headdim", 64)), int(sc.get("d_state", 128)), int(sc.get("ngroups", 1)), int(sc.get("d_conv", 4))
    H = d_inner // headdim
    conv_dim = d_inner + 2 * ngroups * d_state
    sd = {}
    sd[p + "norm.weight"] = _t(1.0 + uniform(seed, p + "norm.weight", (d,), 0.1), dtype)
    if not cfg.get("rms_norm"):
        sd[p + "norm.bias"] = _t(uniform(seed, p + "norm.bias", (d,), 0.1), dtype)
    sd[p + "mixer.in_proj.weight"] = _t(uniform(seed, p + "mixer.in_proj.weight", (2 * d_inner + 2 * ngroups * d_state + H, d), 1.0 / np.sqrt(d)), dtype)
    sd[p + "mixer.conv1d.weight"] = _t(uniform(seed, p + "mixer.conv1d.weight", (conv_dim, 1, d_conv), 1.0 / np.sqrt(d_conv)), dtype)
    sd[p + "mixer.conv1d.bias"] = _t(uniform(seed, p + "mixer.conv1d.bias", (conv_dim,), 1.0 / np.sqrt(d_conv)), dtype)
    u = (uniform(seed, p + "mixer.dt", (H,), 1.0).astype(np.float64) + 1.0) / 2.0
    dt = np.exp(u * (np.log(0.1) - np.log(0.001)) + np.log(0.001))
    sd[p + "mixer.dt_bias"] = _t((dt + np.log(-np.expm1(-dt))).astype(np.float32), dtype)
    a = 1.0 + 15.0 * (uniform(seed, p + "mixer.A", (H,), 1.0).astype(np.float64) + 1.0) / 2.0
    sd[p + "mixer.A_log"] = _t(np.log(a).astype(np.float32), dtype)
    sd[p + "mixer.D"] = _t(1.0 + uniform(seed, p + "mixer.D", (H,), 0.1), dtype)
    sd[p + "mixer.norm.weight"] = _t(1.0 + uniform(seed, p + "mixer.norm.weight", (d_inner,), 0.1), dtype)
    sd[p + "mixer.out_proj.weight"] = _t(uniform(seed, p + "mixer.out_proj.weight", (d, d_inner), 1.0 / np.sqrt(d_inner)), dtype)
    return sd


# Hybrid configurations: tiny for parity tests; full = the recalled Zonos-v0.1-hybrid dimensions (SURVEY.md 0.8,
# "lower confidence": the loader reads the real ones from config.json).
# Without an "attn_cfg" entry the hybrid configurations keep round 1's attention form (interleaved rotary over the whole head,
# no biases); with one, the entry is what config.json's backbone.attn_cfg would hold beside num_heads / num_heads_kv and
# mamba_ssm's MHA defaults apply to the keys it leaves out (rotary_emb_dim 0, half-split pairs, biases on).
HYBRID_TINY_CFG = dict(d_model=128, n_layer=4, num_heads=4, num_heads_kv=2, d_ff=256, ssm_cfg={"layer": "Mamba2", "d_state": 64},
                       attn_layer_idx=[2])
HYBRID_FULL_CFG = dict(d_model=2048, n_layer=46, num_heads=16, num_heads_kv=4, d_ff=8192, ssm_cfg={"layer": "Mamba2"},
                       attn_layer_idx=[9, 19, 29, 39])
# the attention form recalled for the Zonos-v0.1-hybrid checkpoint: rotary over the whole head, half-split pairs, no biases
HYBRID_CKPT_ATTN = dict(causal=True, rotary_emb_dim=32, qkv_proj_bias=False, out_proj_bias=False)
TINY_CFG = dict(d_model=128, n_layer=2, num_heads=4, num_heads_kv=2, d_ff=256)
# smallest shape the persistent decode chain (csrc/zn_chain_kernel.h) serves: d_model 512 = one 512-column chunk per weight
# row, d_ff = 4 d_model, head size 128; three blocks = first, middle (with the next block's in_proj) and last chain launch
CHAIN_CFG = dict(d_model=512, n_layer=3, num_heads=4, num_heads_kv=2, d_ff=2048)
FULL_CFG = dict(d_model=2048, n_layer=26, num_heads=16, num_heads_kv=4, d_ff=8192)


def conditioning(seed: int, name: str, rows: int, l_c: int, d: int, dtype=torch.bfloat16) -> torch.Tensor:
    """Synthetic prefix conditioning [rows, L_c, d]."""
    return _t(normal(seed, name, (rows, l_c, d)), dtype)


# ------------------------------------------------------------------ prefix conditioner
# Conditioner list of Zonos-v0.1-transformer as documented in the reference's CONDITIONING_README.md:3-75
TRANSFORMER_CONDITIONERS = [
    {"type": "EspeakPhonemeConditioner", "name": "espeak"},
    {"type": "PassthroughConditioner", "name": "speaker", "cond_dim": 128, "uncond_type": "learned", "projection": "linear"},
    {"type": "FourierConditioner", "name": "emotion", "input_dim": 8, "uncond_type": "learned"},
    {"type": "FourierConditioner", "name": "fmax", "min_val": 0, "max_val": 24000, "uncond_type": "learned"},
    {"type": "FourierConditioner", "name": "pitch_std", "min_val": 0, "max_val": 400, "uncond_type": "learned"},
    {"type": "FourierConditioner", "name": "speaking_rate", "min_val": 0, "max_val": 40, "uncond_type": "learned"},
    {"type": "IntegerConditioner", "name": "language_id", "min_val": -1, "max_val": 126, "uncond_type": "learned"},
]
N_PHONEME_TOKENS = 4 + 185       # special tokens + symbol table (zonos/conditioning.py:225-240)


def conditioner_state_dict(conditioners: list, d: int, seed: int = 1234, projection: str = "none", dtype=torch.bfloat16) -> dict:
    """Synthetic weights of a PrefixConditioner under the reference's module names (zonos/conditioning.py:506-511)."""
    sd = {}

    def lin(prefix, cin, cout):
        sd[prefix + "weight"] = _t(uniform(seed, "cond." + prefix + "weight", (cout, cin), 1.0 / np.sqrt(cin)), dtype)
        sd[prefix + "bias"] = _t(uniform(seed, "cond." + prefix + "bias", (cout,), 1.0 / np.sqrt(cin)), dtype)

    def proj(prefix, kind, cin):
        if kind == "linear":
            lin(prefix + "project.", cin, d)
        elif kind == "mlp":
            lin(prefix + "project.0.", cin, d)
            lin(prefix + "project.2.", d, d)

    for i, c in enumerate(conditioners):
        p = f"conditioners.{i}."
        if c["type"] == "EspeakPhonemeConditioner":
            sd[p + "phoneme_embedder.weight"] = _t(normal(seed, "cond." + p + "phoneme_embedder", (N_PHONEME_TOKENS, d)), dtype)
        elif c["type"] == "FourierConditioner":
            sd[p + "weight"] = _t(normal(seed, "cond." + p + "weight", (d // 2, c.get("input_dim", 1)), c.get("std", 1.0)), dtype)
        elif c["type"] == "IntegerConditioner":
            sd[p + "int_embedder.weight"] = _t(normal(seed, "cond." + p + "int_embedder", (c.get("max_val", 512) - c.get("min_val", 0) + 1, d)), dtype)
        proj(p, c.get("projection", "none"), c.get("cond_dim") or d)
        if c.get("uncond_type") == "learned":
            sd[p + "uncond_vector"] = _t(normal(seed, "cond." + p + "uncond_vector", (d,)), dtype)
    proj("", projection, d)
    sd["norm.weight"] = _t(1.0 + uniform(seed, "cond.norm.weight", (d,), 0.1), dtype)
    sd["norm.bias"] = _t(uniform(seed, "cond.norm.bias", (d,), 0.1), dtype)
    return sd


# ------------------------------------------------------------------ DAC decoder weights

def dac_decoder_spec(hidden=1024, dec_hidden=1536, ratios=(8, 8, 4, 2), n_codebooks=9,
                     codebook_size=1024, codebook_dim=8):
    """(name, shape, kind) for every tensor autoencoder.decode() touches, in the state-dict naming of
    transformers DacModel (modeling_dac.py:347-371 from_codes, :407-441 DacDecoder, :236-264 block,
    :175-209 residual unit)."""
    spec = []
    for i in range(n_codebooks):
        q = f"quantizer.quantizers.{i}."
        spec += [(q + "codebook.weight", (codebook_size, codebook_dim), "emb"),
                 (q + "out_proj.weight", (hidden, codebook_dim, 1), "conv"),
                 (q + "out_proj.bias", (hidden,), "bias")]
    spec += [("decoder.conv1.weight", (dec_hidden, hidden, 7), "conv"), ("decoder.conv1.bias", (dec_hidden,), "bias")]
    c = dec_hidden
    for bi, s in enumerate(ratios):
        b = f"decoder.block.{bi}."
        co = c // 2
        spec += [(b + "snake1.alpha", (1, c, 1), "alpha"),
                 (b + "conv_t1.weight", (c, co, 2 * s), "convt"), (b + "conv_t1.bias", (co,), "bias")]
        for u in (1, 2, 3):
            r = b + f"res_unit{u}."
            spec += [(r + "snake1.alpha", (1, co, 1), "alpha"),
                     (r + "conv1.weight", (co, co, 7), "conv"), (r + "conv1.bias", (co,), "bias"),
                     (r + "snake2.alpha", (1, co, 1), "alpha"),
                     (r + "conv2.weight", (co, co, 1), "conv"), (r + "conv2.bias", (co,), "bias")]
        c = co
    spec += [("decoder.snake1.alpha", (1, c, 1), "alpha"),
             ("decoder.conv2.weight", (1, c, 7), "conv"), ("decoder.conv2.bias", (1,), "bias")]
    return spec


def dac_encoder_spec(hidden=1024, enc_hidden=64, ratios=(8, 8, 4, 2), n_codebooks=9, codebook_dim=8, **_):
    """(name, shape, kind) for the tensors autoencoder.encode() touches beyond the decoder's: DacEncoder
    (modeling_dac.py:444-473, blocks :212-233 with downsampling_ratios = reversed upsampling ratios) and the quantizers'
    in_proj (:119)."""
    spec = [("encoder.conv1.weight", (enc_hidden, 1, 7), "conv"), ("encoder.conv1.bias", (enc_hidden,), "bias")]
    c = enc_hidden
    for bi, s in enumerate(reversed(ratios)):
        b = f"encoder.block.{bi}."
        for u in (1, 2, 3):
            r = b + f"res_unit{u}."
            spec += [(r + "snake1.alpha", (1, c, 1), "alpha"),
                     (r + "conv1.weight", (c, c, 7), "conv"), (r + "conv1.bias", (c,), "bias"),
                     (r + "snake2.alpha", (1, c, 1), "alpha"),
                     (r + "conv2.weight", (c, c, 1), "conv"), (r + "conv2.bias", (c,), "bias")]
        spec += [(b + "snake1.alpha", (1, c, 1), "alpha"),
                 (b + "conv1.weight", (2 * c, c, 2 * s), "conv"), (b + "conv1.bias", (2 * c,), "bias")]
        c *= 2
    spec += [("encoder.snake1.alpha", (1, c, 1), "alpha"),
             ("encoder.conv2.weight", (hidden, c, 3), "conv"), ("encoder.conv2.bias", (hidden,), "bias")]
    for i in range(n_codebooks):
        q = f"quantizer.quantizers.{i}."
        spec += [(q + "in_proj.weight", (codebook_dim, hidden, 1), "conv"), (q + "in_proj.bias", (codebook_dim,), "bias")]
    return spec


def dac_state_dict(seed: int = 4321, encoder: bool = True, **kw) -> dict:
    """Synthetic fp32 DAC weights (decoder + quantizer, and the encoder unless encoder=False).  Conv weights use the
    PyTorch-default fan-in scale (so activations stay O(1) through ~30 layers), Snake alpha in [0.5, 1.5], codebooks N(0,1)."""
    sd = {}
    dkw = {k: v for k, v in kw.items() if k != "enc_hidden"}
    for name, shape, kind in dac_decoder_spec(**dkw) + (dac_encoder_spec(**kw) if encoder else []):
        if kind == "emb":
            a = normal(seed, name, shape, 1.0)
        elif kind == "conv":
            a = uniform(seed, name, shape, 1.0 / np.sqrt(shape[1] * shape[2]))
        elif kind == "convt":  # [Cin, Cout, k]; each output sees 2 taps * Cin inputs
            a = uniform(seed, name, shape, 1.0 / np.sqrt(shape[0] * 2))
        elif kind == "bias":
            a = uniform(seed, name, shape, 0.05)
        else:
            a = 1.0 + uniform(seed, name, shape, 0.5)
        sd[name] = torch.from_numpy(a)
    return sd


def test_waveform(seed: int, name: str, T: int, batch: int = 1) -> torch.Tensor:
    """Deterministic fp32 test audio [batch, 1, T] in (-1, 1): three sines (220, 1370, 5100 Hz at 44.1 kHz) under a slow
    envelope plus low-level noise from the counter RNG."""
    t = np.arange(T, dtype=np.float64) / 44100.0
    out = np.empty((batch, 1, T), dtype=np.float32)
    for b in range(batch):
        ph = uniform(seed, f"{name}.phase.{b}", (3,), np.pi).astype(np.float64)
        sig = 0.30 * np.sin(2 * np.pi * 220.0 * t + ph[0]) + 0.15 * np.sin(2 * np.pi * 1370.0 * t + ph[1]) + 0.08 * np.sin(2 * np.pi * 5100.0 * t + ph[2])
        env = 0.6 + 0.4 * np.sin(2 * np.pi * 3.0 * t + b)
        out[b, 0] = (sig * env).astype(np.float32) + uniform(seed, f"{name}.noise.{b}", (T,), 0.02)
    return torch.from_numpy(out)


# ------------------------------------------------------------------ speaker embedding (zonos/speaker_cloning.py)
def speaker_state_dict(seed: int = 2468, in_planes: int = 64, num_blocks=(10, 20, 64, 3), n_mels: int = 80, att_dim: int = 128,
                       emb_dim: int = 256, lda_dim: int = 128) -> tuple[dict, dict]:
    """Synthetic fp32 weights of ResNet293_based under the reference's state-dict names (speaker_cloning.py:353-392 ResNet,
    :168-183 SimAMBasicBlock, :115-127 ASP, :460-463) and of the LDA Linear (:860-864).  Conv weights use a He-like scale so
    that activations stay O(1) through the 97 residual blocks; BatchNorm gets non-trivial affine parameters and running
    statistics (so that folding bugs show up).  Returns (model state dict, lda state dict)."""
    sd = {}

    def bn(p, c):
        sd[p + "weight"] = torch.from_numpy(1.0 + uniform(seed, p + "weight", (c,), 0.2))
        sd[p + "bias"] = torch.from_numpy(uniform(seed, p + "bias", (c,), 0.1))
        sd[p + "running_mean"] = torch.from_numpy(uniform(seed, p + "running_mean", (c,), 0.2))
        sd[p + "running_var"] = torch.from_numpy(1.0 + uniform(seed, p + "running_var", (c,), 0.5))
        sd[p + "num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    def conv(name, shape, gain=1.0):
        fan_in = int(np.prod(shape[1:]))
        sd[name] = torch.from_numpy(uniform(seed, name, shape, gain * np.sqrt(3.0 / fan_in)))

    conv("front.conv1.weight", (in_planes, 1, 3, 3), 1.4)
    bn("front.bn1.", in_planes)
    cin = in_planes
    for li, nb in enumerate(num_blocks, start=1):
        planes = in_planes << (li - 1)
        for bi in range(nb):
            p = f"front.layer{li}.{bi}."
            stride = 2 if (bi == 0 and li > 1) else 1
            conv(p + "conv1.weight", (planes, cin, 3, 3), 1.4)
            bn(p + "bn1.", planes)
            conv(p + "conv2.weight", (planes, planes, 3, 3), 0.5)
            bn(p + "bn2.", planes)
            if stride != 1 or cin != planes:
                conv(p + "downsample.0.weight", (planes, cin, 1, 1), 1.0)
                bn(p + "downsample.1.", planes)
            cin = planes
    F = cin * (n_mels // 8)
    conv("pooling.attention.0.weight", (att_dim, F, 1))
    sd["pooling.attention.0.bias"] = torch.from_numpy(uniform(seed, "pooling.attention.0.bias", (att_dim,), 0.1))
    bn("pooling.attention.2.", att_dim)
    conv("pooling.attention.3.weight", (F, att_dim, 1), 2.0)
    sd["pooling.attention.3.bias"] = torch.from_numpy(uniform(seed, "pooling.attention.3.bias", (F,), 0.1))
    conv("bottleneck.weight", (emb_dim, 2 * F))
    sd["bottleneck.bias"] = torch.from_numpy(uniform(seed, "bottleneck.bias", (emb_dim,), 0.1))
    lda = {"weight": torch.from_numpy(uniform(seed, "lda.weight", (lda_dim, emb_dim), 1.0 / np.sqrt(emb_dim))),
           "bias": torch.from_numpy(uniform(seed, "lda.bias", (lda_dim,), 0.1))}
    return sd, lda


def speaker_features(seed: int, name: str, batch: int, n_mels: int, frames: int) -> torch.Tensor:
    """Synthetic mean-normalised log-mel-like features [batch, n_mels, frames] (what logFbankCal returns)."""
    x = torch.from_numpy(normal(seed, name, (batch, n_mels, frames), 1.0))
    return x - x.mean(dim=2, keepdim=True)
