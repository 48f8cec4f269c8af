"""CPU ORACLE for the Zonos hot path.  TEST INFRASTRUCTURE ONLY.

This is a CPU restatement (torch CPU ops, bf16 weights/activations, fp32 logits) of the reference's
autoregressive DAC-token decode path and of DAC `decode()`.  It exists to CHECK the HIP path; only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.  The product
package `zonos_amd` never imports it and has no CPU fallback.

Parity pinning: the reference has no tests or golden vectors of its own, so this oracle
is pinned against outputs of the reference itself, run in the build container by
`tests/golden/make_golden.py` (imports /root/reference with stub modules) and committed
as `tests/golden/*.npz`; `tests/test_oracle_golden.py` checks this file against them bit-for-bit.
DAC arithmetic lives in third-party `transformers` (reference pins >=4.48.1, installed 5.15.0); it is
pinned the same way against `transformers.models.dac.DacModel` built locally with synthetic weights.
The hybrid (Mamba2) backbone has no importable reference here: parity unpinned.

Every function cites the reference file:line it follows (paths relative to /root/reference/).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F

EOS_ID = 1024      # zonos/config.py:123
MASK_ID = 1025     # zonos/config.py:124
N_Q = 9            # zonos/config.py:126


# --------------------------------------------------------------------------- delay pattern
def apply_delay_pattern(codes: torch.Tensor, mask_token: int) -> torch.Tensor:
    """zonos/codebook_pattern.py:31-32 — pad n_q mask columns on the right, rotate codebook k right by k+1."""
    b, nq, t = codes.shape
    out = torch.full((b, nq, t + nq), mask_token, dtype=codes.dtype)
    for k in range(nq):
        out[:, k, k + 1:k + 1 + t] = codes[:, k]
    return out


def revert_delay_pattern(codes: torch.Tensor) -> torch.Tensor:
    """zonos/codebook_pattern.py:60-61 — codebook k keeps columns k+1 .. T-n_q+k."""
    _, nq, t = codes.shape
    return torch.stack([codes[:, k, k + 1:t - nq + k + 1] for k in range(nq)], dim=1)


# --------------------------------------------------------------------------- backbone pieces
def rope_table(n_pos: int, head_dim: int, base: float = 10000.0) -> torch.Tensor:
    """zonos/backbone/_torch.py:29-34 — fp32 [n_pos, head_dim/2, 2] (cos, sin)."""
    inv = 1.0 / (base ** (torch.arange(0, head_dim, 2)[: head_dim // 2].float() / head_dim))
    ang = torch.outer(torch.arange(n_pos), inv)
    z = torch.polar(torch.ones_like(ang), ang)
    return torch.stack([z.real, z.imag], dim=-1)


def rope_apply(x: torch.Tensor, cs: torch.Tensor) -> torch.Tensor:
    """zonos/backbone/_torch.py:57-68 — x [R,S,h,hd] (bf16), cs [R,S,hd/2,2] fp32; interleaved pairs,
    fp32 math with separate mul/sub roundings, cast back to x.dtype."""
    xs = x.float().reshape(*x.shape[:-1], -1, 2)
    c = cs[:, :, None, :, 0]
    s = cs[:, :, None, :, 1]
    re = xs[..., 0] * c - xs[..., 1] * s
    im = xs[..., 1] * c + xs[..., 0] * s
    return torch.stack([re, im], dim=-1).flatten(3).type_as(x)


@dataclass
class Cache:
    """zonos/config.py:9-52 InferenceParams + _torch.py:286-305 per-layer KV [R, maxL, 2, Hkv, hd] bf16."""
    kv: list
    max_len: int
    seqlen_offset: int = 0
    lengths: torch.Tensor = None
    rope: torch.Tensor = None


def find_multiple(n: int, k: int) -> int:
    """zonos/utilities/utils.py:27-29"""
    return n if k == 0 or n % k == 0 else n + k - n % k


def setup_cache(cfg: dict, rows: int, max_len: int) -> Cache:
    """zonos/model.py:333-338, _torch.py:205-211 — length rounded to x8, bf16 cache, 16384-row RoPE table."""
    max_len = find_multiple(max_len, 8)
    hd = cfg["d_model"] // cfg["num_heads"]
    kv = [torch.zeros(rows, max_len, 2, cfg["num_heads_kv"], hd, dtype=torch.bfloat16) for _ in range(cfg["n_layer"])]
    return Cache(kv, max_len, 0, torch.zeros(rows, dtype=torch.int32), rope_table(16384, hd))


def layer_forward(w: dict, li: int, x: torch.Tensor, cache: Cache, cs: torch.Tensor, cfg: dict,
                  double_out_proj: bool = True, eps: float = 1e-5) -> torch.Tensor:
    """One pre-norm block.  zonos/backbone/_torch.py:326-327 (residuals), :397-420 (attention, incl. the
    second out_proj at :420), :473-474 (gated-SiLU MLP)."""
    p = f"backbone.layers.{li}."
    d, H, Hkv = cfg["d_model"], cfg["num_heads"], cfg["num_heads_kv"]
    hd = d // H
    R, S, _ = x.shape
    n1 = F.layer_norm(x, (d,), w[p + "norm.weight"], w[p + "norm.bias"], eps)
    qkv = F.linear(n1, w[p + "mixer.in_proj.weight"])
    q, k, v = qkv.split([H * hd, Hkv * hd, Hkv * hd], dim=-1)
    q = rope_apply(q.view(R, S, H, hd), cs)
    k = rope_apply(k.view(R, S, Hkv, hd), cs)
    v = v.view(R, S, Hkv, hd)
    kvc = cache.kv[li]
    t0 = cache.seqlen_offset
    assert t0 + S <= kvc.shape[1]                                   # _torch.py:103
    kvc[:R, t0:t0 + S, 0] = k                                       # _torch.py:105-106
    kvc[:R, t0:t0 + S, 1] = v
    kk, vv = kvc[:R, :t0 + S].unbind(dim=-3)
    y = F.scaled_dot_product_attention(q.transpose(1, 2), kk.transpose(1, 2), vv.transpose(1, 2),
                                       is_causal=S > 1, enable_gqa=True)          # _torch.py:415
    y = y.transpose(1, 2).contiguous().view(R, S, H * hd)
    y = F.linear(y, w[p + "mixer.out_proj.weight"])
    if double_out_proj:                                             # _torch.py:419-420
        y = F.linear(y, w[p + "mixer.out_proj.weight"])
    x = x + y
    n2 = F.layer_norm(x, (d,), w[p + "norm2.weight"], w[p + "norm2.bias"], eps)
    val, gate = F.linear(n2, w[p + "mlp.fc1.weight"]).chunk(2, dim=-1)
    return x + F.linear(val * F.silu(gate), w[p + "mlp.fc2.weight"])


def backbone_forward(w: dict, x: torch.Tensor, cache: Cache, cfg: dict, double_out_proj: bool = True) -> torch.Tensor:
    """zonos/backbone/_torch.py:232-238 — positions = arange(S) + lengths_per_sample, 26 blocks, final LN."""
    if cfg.get("ssm_cfg"):
        return hybrid_backbone_forward(w, x, cache, cfg)
    pos = torch.arange(x.shape[1])[None, :] + cache.lengths[:x.shape[0], None].long()
    cs = cache.rope[pos]
    for li in range(cfg["n_layer"]):
        x = layer_forward(w, li, x, cache, cs, cfg, double_out_proj)
    d = cfg["d_model"]
    return F.layer_norm(x, (d,), w["backbone.norm_f.weight"], w["backbone.norm_f.bias"], 1e-5)


# --------------------------------------------------------------------------- hybrid backbone (PARITY UNPINNED)
# zonos/backbone/_mamba_ssm.py:8-119 builds its blocks with the third-party mamba_ssm==2.2.5 (requirements.txt:18;
# source not under /root/reference, not installed): `create_block(..., fused_add_norm=True)` -> Block(norm, mixer =
# Mamba2 | MHA, norm2 + GatedMLP on attention layers), Triton layer_norm_fn for the fused residual-add + norm,
# causal_conv1d_update + selective_state_update for the single-token Mamba2 step.  What follows restates the
# PUBLISHED algorithm of those pieces (Mamba-2 paper, arXiv:2405.21060, sec. 7 / mamba_ssm Mamba2.step) with the
# rounding points of the kernels the reference would run: fp32 inside a kernel, one rounding to bf16 at its output,
# bf16 conv/SSM state (Mamba2.allocate_inference_cache with the dtype generate() passes).  No reference output exists
# to pin it: every result that rests on it says "parity unpinned".
def mamba2_dims(cfg: dict) -> dict:
    sc = cfg["ssm_cfg"]
    d_inner = int(sc.get("expand", 2)) * cfg["d_model"]
    headdim, d_state, ngroups = int(sc.get("headdim", 64)), int(sc.get("d_state", 128)), int(sc.get("ngroups", 1))
    return dict(d_inner=d_inner, headdim=headdim, nheads=d_inner // headdim, d_state=d_state, ngroups=ngroups,
                d_conv=int(sc.get("d_conv", 4)), conv_dim=d_inner + 2 * ngroups * d_state,
                d_in_proj=2 * d_inner + 2 * ngroups * d_state + d_inner // headdim)


def hybrid_setup_cache(cfg: dict, rows: int, max_len: int) -> Cache:
    """_mamba_ssm.py:65-86: attention layers get a KV cache, Mamba2 layers (conv_state [R, conv_dim, d_conv],
    ssm_state [R, nheads, headdim, d_state]) in the dtype generate() passes (bf16)."""
    max_len = find_multiple(max_len, 8)
    hd = cfg["d_model"] // cfg["num_heads"]
    m = mamba2_dims(cfg)
    kv = [torch.zeros(rows, max_len, 2, cfg["num_heads_kv"], hd, dtype=torch.bfloat16) if i in cfg["attn_layer_idx"] else
          (torch.zeros(rows, m["conv_dim"], m["d_conv"], dtype=torch.bfloat16),
           torch.zeros(rows, m["nheads"], m["headdim"], m["d_state"], dtype=torch.bfloat16)) for i in range(cfg["n_layer"])]
    return Cache(kv, max_len, 0, torch.zeros(rows, dtype=torch.int32), rope_table(16384, hd))


def add_norm(h: torch.Tensor, res, wgt: torch.Tensor, bias, eps: float, rms: bool = False, res32: bool = False):
    """Fused residual-add + norm (mamba_ssm layer_norm_fn, prenorm=True; _mamba_ssm.py:45-58,111-119): the sum is formed and
    normalised in fp32; the residual stream keeps its bf16 rounding unless residual_in_fp32 (config.py:83), the norm sees the
    unrounded sum.  rms (config.py:82): RMSNorm, y = s * rsqrt(mean(s^2) + eps) * w (+ b when the module has one)."""
    s = h.float() if res is None else h.float() + res.float()
    if rms:
        y = s * torch.rsqrt(s.pow(2).mean(-1, keepdim=True) + eps) * wgt.float()
        if bias is not None:
            y = y + bias.float()
    else:
        y = F.layer_norm(s, (s.shape[-1],), wgt.float(), bias.float(), eps)
    return y.to(h.dtype), (s if res32 else s.to(h.dtype))


def attn_options(cfg: dict) -> dict:
    """attn_cfg as mamba_ssm's MHA reads it (defaults: rotary_emb_dim 0, rotary_emb_interleaved False, biases True); a hybrid
    configuration without an "attn_cfg" entry keeps the first restatement's form (interleaved rotary, no biases)."""
    hd = cfg["d_model"] // cfg["num_heads"]
    if "attn_cfg" not in cfg:
        return dict(mode=0, qkv_bias=False, out_bias=False)
    ac = cfg["attn_cfg"]
    rot = int(ac.get("rotary_emb_dim", 0))
    assert rot in (0, hd)
    mode = 2 if rot == 0 else (0 if ac.get("rotary_emb_interleaved", False) else 1)
    return dict(mode=mode, qkv_bias=bool(ac.get("qkv_proj_bias", True)), out_bias=bool(ac.get("out_proj_bias", True)))


def rotary_half(x: torch.Tensor, cs: torch.Tensor) -> torch.Tensor:
    """flash_attn apply_rotary with interleaved=False (what mamba_ssm's MHA calls): pairs (i, i + hd/2), cos / sin cached in
    the activations' dtype (bf16), fp32 arithmetic, one rounding.  x [R, 1, H, hd], cs [R, 1, hd/2, 2] fp32 table rows."""
    hh = x.shape[-1] // 2
    cos = cs[..., 0].to(torch.bfloat16).float()[:, :, None, :]
    sin = cs[..., 1].to(torch.bfloat16).float()[:, :, None, :]
    x1, x2 = x[..., :hh].float(), x[..., hh:].float()
    return torch.cat([x1 * cos - x2 * sin, x2 * cos + x1 * sin], dim=-1).to(x.dtype)


def mamba2_step(w: dict, p: str, n: torch.Tensor, conv_state: torch.Tensor, ssm_state: torch.Tensor, m: dict, eps: float = 1e-5,
                state32: torch.Tensor | None = None):
    """One token through a Mamba2 mixer (mamba_ssm Mamba2.step; Mamba-2 paper eq. for the SSD recurrence
    h_t = exp(dt A) h_{t-1} + dt B x_t, y_t = C h_t + D x_t).  n [R, d] bf16; states updated in place.
    state32 (fp32, same shape as ssm_state): the state of a PREFILL in progress - the reference's S > 1 path
    (Mamba2.forward -> mamba_chunk_scan_combined) carries the state in fp32 over the whole sequence and casts only the final
    state into the bf16 cache, whereas Mamba2.step reads and writes the bf16 cache every token."""
    R = n.shape[0]
    di, H, P, N, G = m["d_inner"], m["nheads"], m["headdim"], m["d_state"], m["ngroups"]
    zxbcdt = F.linear(n, w[p + "in_proj.weight"])                                   # bf16 [R, 2*di + 2*G*N + H]
    z, xBC, dt = zxbcdt.split([di, di + 2 * G * N, H], dim=-1)
    # causal_conv1d_update: shift the window, fp32 dot with the taps + bias, SiLU, one rounding
    conv_state.copy_(torch.roll(conv_state, shifts=-1, dims=-1))
    conv_state[:, :, -1] = xBC
    wf = w[p + "conv1d.weight"].float().view(-1, m["d_conv"])
    acc = w[p + "conv1d.bias"].float()[None, :].expand(R, -1)
    for i in range(m["d_conv"]):                                                     # out = bias; out += w[i] * win[i], tap order
        acc = acc + wf[None, :, i] * conv_state[:, :, i].float()
    xBC = (acc / (1.0 + torch.exp(-acc))).to(n.dtype)
    x, Bm, Cm = xBC.split([di, G * N, G * N], dim=-1)
    # selective_state_update (fp32 inside; state stored bf16, y from the unrounded new state)
    A = -torch.exp(w[p + "A_log"].float())                                           # [H]
    dtv = dt.float() + w[p + "dt_bias"].float()
    dtv = torch.where(dtv <= 20.0, torch.log1p(torch.exp(dtv)), dtv)                 # softplus
    dA = torch.exp(dtv * A)                                                          # [R, H]
    xh = x.float().view(R, H, P)
    Bg = Bm.float().view(R, G, N).repeat_interleave(H // G, dim=1)                   # [R, H, N]
    Cg = Cm.float().view(R, G, N).repeat_interleave(H // G, dim=1)
    prev = ssm_state.float() if state32 is None else state32
    new_state = prev * dA[:, :, None, None] + (Bg * dtv[:, :, None])[:, :, None, :] * xh[:, :, :, None]
    if state32 is None:
        ssm_state.copy_(new_state.to(ssm_state.dtype))
    else:
        state32.copy_(new_state)
    y = (new_state * Cg[:, :, None, :]).sum(-1) + xh * w[p + "D"].float()[None, :, None]
    y = y.reshape(R, di).to(n.dtype)
    # RMSNormGated(norm_before_gate=False): rmsnorm(y * silu(z)) * weight, per group of d_inner / ngroups
    v = y.float() * (z.float() * torch.sigmoid(z.float()))
    vg = v.view(R, G, di // G)
    vg = vg * torch.rsqrt(vg.pow(2).mean(-1, keepdim=True) + eps)
    o = (vg.reshape(R, di) * w[p + "norm.weight"].float()).to(n.dtype)
    return F.linear(o, w[p + "out_proj.weight"])


def hybrid_backbone_forward(w: dict, x: torch.Tensor, cache: Cache, cfg: dict) -> torch.Tensor:
    """_mamba_ssm.py:106-119: position by position (the chunked-scan prefill of the library is the same recurrence)."""
    d, H, Hkv, eps = cfg["d_model"], cfg["num_heads"], cfg["num_heads_kv"], 1e-5
    hd = d // H
    R, S, _ = x.shape
    m = mamba2_dims(cfg)
    ao = attn_options(cfg)
    rms, res32 = bool(cfg.get("rms_norm")), bool(cfg.get("residual_in_fp32"))
    # flash_attn's RotaryEmbedding (what mamba_ssm's MHA applies) caches cos / sin in the activations' dtype, for the interleaved
    # form as for the half-split one
    rot = ((lambda t, c: rope_apply(t, c.to(torch.bfloat16).float())) if ao["mode"] == 0 else (lambda t, c: rotary_half(t, c)) if ao["mode"] == 1
           else (lambda t, c: t))
    outs = []
    # a prefill (S > 1) carries every Mamba2 layer's SSM state in fp32 across its positions and rounds it into the cache once
    st32 = {li: cache.kv[li][1][:R].float() for li in range(cfg["n_layer"]) if li not in cfg["attn_layer_idx"]} if S > 1 else {}
    for s_i in range(S):
        h, res = x[:, s_i], None
        pos = cache.lengths[:R].long() + s_i
        cs = cache.rope[pos][:, None]                                               # [R, 1, hd/2, 2]
        t0 = cache.seqlen_offset + s_i
        for li in range(cfg["n_layer"]):
            p = f"backbone.layers.{li}."
            n, res = add_norm(h, res, w[p + "norm.weight"], w.get(p + "norm.bias"), eps, rms, res32)
            if li in cfg["attn_layer_idx"]:
                # mamba_ssm MHA (rotary form and biases per attn_cfg), single out_proj
                qkv = F.linear(n, w[p + "mixer.in_proj.weight"], w.get(p + "mixer.in_proj.bias") if ao["qkv_bias"] else None)
                q, k, v = qkv.split([H * hd, Hkv * hd, Hkv * hd], dim=-1)
                q = rot(q.view(R, 1, H, hd), cs)
                k = rot(k.view(R, 1, Hkv, hd), cs)
                kvc = cache.kv[li]
                kvc[:R, t0, 0] = k[:, 0]
                kvc[:R, t0, 1] = v.view(R, Hkv, hd)
                kk, vv = kvc[:R, :t0 + 1].unbind(dim=-3)
                a = F.scaled_dot_product_attention(q.transpose(1, 2), kk.transpose(1, 2), vv.transpose(1, 2), enable_gqa=True)
                h = F.linear(a.transpose(1, 2).reshape(R, H * hd), w[p + "mixer.out_proj.weight"], w.get(p + "mixer.out_proj.bias") if ao["out_bias"] else None)
                n2, res = add_norm(h, res, w[p + "norm2.weight"], w.get(p + "norm2.bias"), eps, rms, res32)
                val, gate = F.linear(n2, w[p + "mlp.fc1.weight"]).chunk(2, dim=-1)
                h = F.linear(val * F.silu(gate), w[p + "mlp.fc2.weight"])
            else:
                conv_state, ssm_state = cache.kv[li]
                h = mamba2_step(w, p + "mixer.", n, conv_state[:R], ssm_state[:R], m, eps, state32=st32.get(li))
        out, _ = add_norm(h, res, w["backbone.norm_f.weight"], w["backbone.norm_f.bias"], eps, rms, res32)      # norm_f keeps its bias
        outs.append(out)
    for li, s32 in st32.items():
        cache.kv[li][1][:R].copy_(s32.to(cache.kv[li][1].dtype))
    return torch.stack(outs, dim=1)


def embed_codes(w: dict, codes: torch.Tensor) -> torch.Tensor:
    """zonos/utilities/codec_utils.py:37 — python sum(): 0 + E0[c0], then sequential bf16 adds."""
    acc = 0
    for i in range(codes.shape[1]):
        acc = acc + F.embedding(codes[:, i], w[f"embeddings.{i}.weight"])
    return acc


def compute_logits(w: dict, hidden: torch.Tensor, cache: Cache, cfg: dict, cfg_scale: float,
                   double_out_proj: bool = True) -> torch.Tensor:
    """zonos/model.py:228-234 + codec_utils.py:68-79 — last position, fused heads [9*1025, d], fp32, CFG mix
    with rows [:B]=cond, [B:]=uncond."""
    last = backbone_forward(w, hidden, cache, cfg, double_out_proj)[:, -1, :].unsqueeze(1)
    out = F.linear(last, w["fused_heads.weight"])
    R = out.shape[0]
    logits = out.view(R, 1, N_Q, -1).transpose(1, 2).squeeze(2).float()
    if cfg_scale != 1.0:
        c, u = logits.chunk(2)
        logits = u + (c - u) * cfg_scale
    logits[..., 1025:].fill_(-torch.inf)
    return logits


# --------------------------------------------------------------------------- sampling (zonos/sampling.py)
def repetition_penalty(logits, generated, penalty: float, window: int):
    """zonos/sampling.py:159-163 — product of `penalty` per occurrence in the last `window` tokens."""
    g = generated[..., -window:].clamp_max(logits.shape[-1] - 1).to(torch.int64)
    fac = torch.ones_like(logits).scatter_reduce(2, g, torch.full_like(logits, penalty), reduce="prod")
    return torch.where(logits <= 0, logits * fac, logits / fac)


def unified(probs, linear: float, conf: float, quad: float):
    """zonos/sampling.py:60-63"""
    lp = torch.log(probs.clamp_min(1e-20))
    ent = -torch.sum(probs * lp, dim=-1, keepdim=True)
    return (lp * (linear + ent * conf) - lp ** 2 * quad).softmax(dim=-1)


def top_p(probs, p: float):
    """zonos/sampling.py:93-99 — keep sorted entries whose exclusive prefix sum is <= p."""
    ps, idx = torch.sort(probs, dim=-1, descending=True)
    cs = torch.cumsum(ps, dim=-1)
    ps = ps * (~(cs - ps > p)).float()
    out = probs.scatter(-1, idx, ps)
    return out / out.sum(dim=-1, keepdim=True)


def top_k(probs, k: int):
    """zonos/sampling.py:77-81"""
    v, _ = torch.topk(probs, min(k, probs.size(-1)))
    out = torch.where(probs < v[..., -1:], 0.0, probs)
    return out / out.sum(dim=-1, keepdim=True)


def min_p(probs, mp: float):
    """zonos/sampling.py:123-127"""
    out = probs.masked_fill(probs < mp * probs.max(dim=-1, keepdim=True).values, 0.0)
    return out / out.sum(dim=-1, keepdim=True)


def filtered_probs(logits, temperature=1.0, top_p_=0.0, top_k_=0, min_p_=0.0, linear=0.0, conf=0.0, quad=0.0):
    """The deterministic part of zonos/sampling.py:216-225 (everything before the Gumbel-max draw)."""
    probs = torch.softmax(logits / temperature, dim=-1)
    if linear > 0.0:
        probs = unified(probs, linear, conf, quad)
    if top_p_ > 0:
        probs = top_p(probs, top_p_)
    if top_k_ > 0:
        probs = top_k(probs, top_k_)
    if min_p_ > 0:
        probs = min_p(probs, min_p_)
    return probs


def sample_from_logits(logits, temperature=1.0, top_p=0.0, top_k=0, min_p=0.0, linear=0.0, conf=0.0, quad=0.0,
                       generated_tokens=None, repetition_penalty_=3.0, repetition_penalty_window=2, generator=None):
    """zonos/sampling.py:166-231; multinomial = Gumbel-max argmax(p / Exp(1)) (:28-30)."""
    if repetition_penalty_ != 1.0 and generated_tokens is not None:
        logits = repetition_penalty(logits, generated_tokens, repetition_penalty_, repetition_penalty_window)
    if temperature > 0:
        probs = filtered_probs(logits, temperature, top_p, top_k, min_p, linear, conf, quad)
        q = torch.empty_like(probs).exponential_(1, generator=generator)
        return torch.argmax(probs / q, dim=-1, keepdim=True).to(torch.int64)
    return torch.argmax(logits, dim=-1, keepdim=True)


# --------------------------------------------------------------------------- generate (zonos/model.py:354-548)
@dataclass
class GenTrace:
    """Optional per-step record for teacher-forced comparisons."""
    logits: list = field(default_factory=list)      # fp32 [B,9,1025] fed to the sampler (after bias), per step
    tokens: list = field(default_factory=list)      # sampled [B,9] per step (before EOS masking)
    inputs: list = field(default_factory=list)      # delayed column [B,9] fed to loop step j (after MASK / prefix / EOS fill)
    final_offset: int = 0
    steps_run: int = 0


def generate(w: dict, cfg: dict, prefix_conditioning: torch.Tensor, audio_prefix_codes=None,
             max_new_tokens: int = 86 * 30, cfg_scale: float = 2.0, batch_size: int = 1,
             sampling_params: dict | None = None, callback=None, double_out_proj: bool = True,
             trace: GenTrace | None = None, logits_hook=None) -> torch.Tensor:
    """Restates Zonos.generate.  `logits_hook(step, logits)->logits` lets tests script the logits stream
    (EOS-cadence cases); step -1 is the prefill.

    Batch semantics: the reference crashes for batch_size>=2 under CFG (generation_utils.py:237-238,
    SURVEY.md §0.6); like the build, this oracle treats B utterances as rows [cond_0..cond_{B-1},
    uncond_0..uncond_{B-1}] and expands the prefix codes per utterance before doubling."""
    sampling_params = dict(min_p=0.1) if sampling_params is None else dict(sampling_params)
    sp = sampling_params
    if "repetition_penalty" in sp:
        sp["repetition_penalty_"] = sp.pop("repetition_penalty")
    assert cfg_scale != 1                                                       # model.py:399
    B = batch_size
    P = 0 if audio_prefix_codes is None else audio_prefix_codes.shape[2]
    L_c = prefix_conditioning.shape[1]
    audio_len = P + max_new_tokens
    cache = (hybrid_setup_cache if cfg.get("ssm_cfg") else setup_cache)(cfg, 2 * B, L_c + audio_len + N_Q)   # model.py:410-413
    codes = torch.full((B, N_Q, audio_len), -1, dtype=torch.int64)
    if audio_prefix_codes is not None:
        codes[..., :P] = audio_prefix_codes
    delayed = apply_delay_pattern(codes, MASK_ID)                               # model.py:419
    # ---- prefill (generation_utils.py:236-244)
    ids = delayed[..., :P + 1]
    hid = torch.cat([prefix_conditioning, embed_codes(w, torch.cat([ids, ids], 0))], dim=1)
    logits = compute_logits(w, hid, cache, cfg, cfg_scale, double_out_proj)
    if logits_hook is not None:
        logits = logits_hook(-1, logits)
    if trace is not None:
        trace.logits.append(logits.clone())
    nxt = sample_from_logits(logits, **sp).squeeze(-1)                          # model.py:423 (no rep. penalty)
    if trace is not None:
        trace.tokens.append(nxt.clone())
    offset = P + 1
    frame = delayed[..., offset:offset + 1]
    frame.copy_(torch.where(frame == -1, nxt.unsqueeze(-1), frame))             # model.py:427-428
    plen = L_c + P + 1
    cache.seqlen_offset += plen
    cache.lengths[:] += plen
    bias = torch.zeros_like(logits)                                             # model.py:433-437
    bias[:, 1:, EOS_ID] = -torch.inf
    bias[:, 0, EOS_ID] -= torch.log(torch.tensor(2.0))
    stopping = torch.zeros(B, dtype=torch.bool)
    max_steps = delayed.shape[2] - offset
    remaining = torch.full((B,), max_steps)
    ctx = min(max_new_tokens, 100)                                              # model.py:463
    cb = torch.arange(N_Q)[None, :]
    cpu_counter = 0
    for step_idx in range(max_steps):
        offset += 1
        cpu_counter += 1
        if offset >= delayed.shape[2]:
            break
        ids = delayed[..., offset - 1:offset]
        if trace is not None:
            trace.inputs.append(ids[..., 0].clone())
        hid = embed_codes(w, ids).repeat(2, 1, 1)                               # generation_utils.py:191-192
        logits = compute_logits(w, hid, cache, cfg, cfg_scale, double_out_proj)
        logits = logits + bias
        if logits_hook is not None:
            logits = logits_hook(step_idx, logits)
        if trace is not None:
            trace.logits.append(logits.clone())
        nxt = sample_from_logits(logits, generated_tokens=delayed[..., max(0, offset - ctx):offset], **sp).squeeze(-1)
        if trace is not None:
            trace.tokens.append(nxt.clone())
        eos0 = nxt[:, 0] == EOS_ID                                              # model.py:483-490
        remaining = torch.where(eos0, torch.minimum(remaining, torch.tensor(N_Q)), remaining)
        stopping |= eos0
        eos_idx = (N_Q - remaining).clamp(max=N_Q - 1)[:, None]
        st = stopping[:, None]
        nxt = torch.where(st & (cb < eos_idx), MASK_ID, torch.where(st & (cb == eos_idx), EOS_ID, nxt))   # tensor_ops.py:190-211
        if offset < delayed.shape[2]:
            col = delayed[:, :, offset]
            col.copy_(torch.where(col == -1, nxt, col))                         # tensor_ops.py:42-49
        cache.seqlen_offset += 1                                                # tensor_ops.py:84-105
        cache.lengths.add_(1)
        remaining = remaining - 1
        if trace is not None:
            trace.steps_run = step_idx + 1
        if step_idx % 16 == 15:
            if bool((remaining <= 0).all()):
                break
        elif step_idx % 8 == 7:
            if max(0, B * 10 - cpu_counter) < 5 and bool((remaining <= 0).all()):
                break
        if callback is not None and not callback(frame, step_idx + 1, max_steps):
            break
    out = revert_delay_pattern(delayed)                                         # model.py:511
    valid = offset - N_Q
    win = min(50, valid // 4)
    for pos in range(max(0, valid - win), valid):                               # model.py:516-528
        if int((out[:, :, pos] == EOS_ID).sum()) >= N_Q // 2:
            valid = pos
            break
    out = torch.where(out > 1024, 512, out)
    out = torch.where(out == 1024, 0, out)
    if trace is not None:
        trace.final_offset = offset
    return torch.clamp(out[..., :valid], 0, 1023)                               # model.py:531-539


# --------------------------------------------------------------------------- prefix conditioner (zonos/conditioning.py)
def _cond_project(pw: dict, prefix: str, kind: str, x: torch.Tensor) -> torch.Tensor:
    """conditioning.py:52-60 Conditioner.project: none | Linear | Linear-SiLU-Linear."""
    if kind == "linear":
        return F.linear(x, pw[prefix + "project.weight"], pw[prefix + "project.bias"])
    if kind == "mlp":
        h = F.silu(F.linear(x, pw[prefix + "project.0.weight"], pw[prefix + "project.0.bias"]))
        return F.linear(h, pw[prefix + "project.2.weight"], pw[prefix + "project.2.bias"])
    return x


def prefix_conditioner_forward(pw: dict, conditioners: list, projection: str, cond_dict: dict, d: int) -> torch.Tensor:
    """conditioning.py:513-522 over :364-365 (phoneme embedding of token ids), :436-441 (Fourier features in the
    weight's dtype), :467 (integer embedding), :476 (passthrough), :104-109 (learned unconditional vector when the key
    is absent).  `cond_dict["espeak"]` holds token ids [B, S] (tokenisation/phonemisation is outside the numeric path)."""
    conds = []
    for i, c in enumerate(conditioners):
        p = f"conditioners.{i}."
        v = cond_dict.get(c["name"])
        if v is None:
            conds.append(pw[p + "uncond_vector"].view(1, 1, -1))
            continue
        if c["type"] == "EspeakPhonemeConditioner":
            e = F.embedding(v, pw[p + "phoneme_embedder.weight"])
        elif c["type"] == "FourierConditioner":
            w = pw[p + "weight"]
            mn, mx = c.get("min_val", 0.0), c.get("max_val", 1.0)
            x = (v - mn) / (mx - mn)
            f = 2 * torch.pi * x.to(w.dtype) @ w.T
            e = torch.cat([f.cos(), f.sin()], dim=-1)
        elif c["type"] == "IntegerConditioner":
            e = F.embedding(v.squeeze(-1) - c.get("min_val", 0), pw[p + "int_embedder.weight"])
        else:
            e = v
        conds.append(_cond_project(pw, p, c.get("projection", "none"), e))
    bsz = max(len(c) for c in conds)
    x = torch.cat([c.expand(bsz, -1, -1) for c in conds], dim=-2)
    return F.layer_norm(_cond_project(pw, "", projection, x), (d,), pw["norm.weight"], pw["norm.bias"], 1e-5)


def prepare_conditioning(pw: dict, conditioners: list, projection: str, cond_dict: dict, d: int, cfg_scale: float = 2.0) -> torch.Tensor:
    """conditioning_cache.py:165-171: [cond ‖ uncond], uncond = only the keys without a learned unconditional vector."""
    if cfg_scale == 1.0:
        return prefix_conditioner_forward(pw, conditioners, projection, cond_dict, d)
    required = {c["name"] for c in conditioners if c.get("uncond_type") != "learned"}
    uncond = {k: cond_dict[k] for k in required}
    return torch.cat([prefix_conditioner_forward(pw, conditioners, projection, cond_dict, d),
                      prefix_conditioner_forward(pw, conditioners, projection, uncond, d)])


# --------------------------------------------------------------------------- DAC decode
def snake(x: torch.Tensor, alpha: torch.Tensor) -> torch.Tensor:
    """modeling_dac.py:98 — x + (alpha + 1e-9)^-1 * sin(alpha x)^2, alpha [1,C,1]."""
    return x + (alpha + 1e-9).reciprocal() * torch.sin(alpha * x).pow(2)


def dac_from_codes(dw: dict, codes: torch.Tensor) -> torch.Tensor:
    """modeling_dac.py:365-371 — sum_i out_proj_i(codebook_i[codes_i]) -> [B, hidden, T]."""
    z = 0.0
    for i in range(codes.shape[1]):
        q = f"quantizer.quantizers.{i}."
        lat = F.embedding(codes[:, i, :], dw[q + "codebook.weight"]).transpose(1, 2)
        z = z + F.conv1d(lat, dw[q + "out_proj.weight"], dw[q + "out_proj.bias"])
    return z


def dac_residual_unit(dw: dict, p: str, x: torch.Tensor, dilation: int) -> torch.Tensor:
    """modeling_dac.py:201-209 — k7 dilated conv (pad 3*dil) then 1x1 conv, skip add."""
    y = F.conv1d(snake(x, dw[p + "snake1.alpha"]), dw[p + "conv1.weight"], dw[p + "conv1.bias"],
                 dilation=dilation, padding=3 * dilation)
    y = F.conv1d(snake(y, dw[p + "snake2.alpha"]), dw[p + "conv2.weight"], dw[p + "conv2.bias"])
    return x + y


def dac_decode(dw: dict, codes: torch.Tensor, ratios=(8, 8, 4, 2), collect: dict | None = None) -> torch.Tensor:
    """zonos/autoencoder.py:138-140 (CPU: autocast disabled -> fp32) -> modeling_dac.py:610-640, :431-441,
    :257-264.  codes int64 [B,9,T] -> wav fp32 [B,1,prod(ratios)*T]."""
    h = F.conv1d(dac_from_codes(dw, codes), dw["decoder.conv1.weight"], dw["decoder.conv1.bias"], padding=3)
    if collect is not None:
        collect["conv1"] = h
    for bi, s in enumerate(ratios):
        b = f"decoder.block.{bi}."
        h = F.conv_transpose1d(snake(h, dw[b + "snake1.alpha"]), dw[b + "conv_t1.weight"], dw[b + "conv_t1.bias"],
                               stride=s, padding=math.ceil(s / 2))
        for u, dil in ((1, 1), (2, 3), (3, 9)):
            h = dac_residual_unit(dw, b + f"res_unit{u}.", h, dil)
        if collect is not None:
            collect[f"block{bi}"] = h
    h = F.conv1d(snake(h, dw["decoder.snake1.alpha"]), dw["decoder.conv2.weight"], dw["decoder.conv2.bias"], padding=3)
    return torch.tanh(h).float()


def dac_decode_to_int16(dw: dict, codes: torch.Tensor) -> torch.Tensor:
    """zonos/autoencoder.py:165-170 — clamp(wav*32767, +-32767) -> int16, shape [512T, 1] (batch 1)."""
    wav = dac_decode(dw, codes).squeeze(1)
    return torch.clamp(wav * 32767.0, -32767.0, 32767.0).to(torch.int16).squeeze(0).unsqueeze(1)


# --------------------------------------------------------------------------- DAC encode
def dac_vq_nearest(dw: dict, q: str, lat: torch.Tensor):
    """modeling_dac.py:156-172 decode_latents — nearest code on L2-normalised vectors; lat [B, dim, T]."""
    B, D, T = lat.shape
    enc = F.normalize(lat.permute(0, 2, 1).reshape(B * T, D))
    cb = F.normalize(dw[q + "codebook.weight"])
    l2 = enc.pow(2).sum(1, keepdim=True)
    dist = -(l2 - 2 * enc @ cb.t()) + cb.pow(2).sum(1, keepdim=True).t()
    idx = dist.max(1)[1].reshape(B, T)
    return F.embedding(idx, dw[q + "codebook.weight"]).transpose(1, 2), idx


def dac_encode(dw: dict, wav: torch.Tensor, ratios=(8, 8, 4, 2), collect: dict | None = None) -> torch.Tensor:
    """zonos/autoencoder.py:117 -> modeling_dac.py:599-602: DacEncoder (:464-473; blocks :227-233 with
    downsampling_ratios = reversed(ratios), strided conv k = 2s, pad ceil(s/2)) then the residual VQ (:310-340):
    wav fp32 [B, 1, T] -> codes int64 [B, 9, T / prod(ratios)]."""
    h = F.conv1d(wav, dw["encoder.conv1.weight"], dw["encoder.conv1.bias"], padding=3)
    for bi, s in enumerate(reversed(ratios)):
        b = f"encoder.block.{bi}."
        for u, dil in ((1, 1), (2, 3), (3, 9)):
            h = dac_residual_unit(dw, b + f"res_unit{u}.", h, dil)
        h = F.conv1d(snake(h, dw[b + "snake1.alpha"]), dw[b + "conv1.weight"], dw[b + "conv1.bias"], stride=s, padding=math.ceil(s / 2))
        if collect is not None:
            collect[f"block{bi}"] = h
    z = F.conv1d(snake(h, dw["encoder.snake1.alpha"]), dw["encoder.conv2.weight"], dw["encoder.conv2.bias"], padding=1)
    if collect is not None:
        collect["z"] = z
    residual, codes = z, []
    n_q = sum(1 for k in dw if k.endswith(".in_proj.weight"))
    for i in range(n_q):
        q = f"quantizer.quantizers.{i}."
        lat = F.conv1d(residual, dw[q + "in_proj.weight"], dw[q + "in_proj.bias"])
        quant, idx = dac_vq_nearest(dw, q, lat)
        residual = residual - F.conv1d(quant, dw[q + "out_proj.weight"], dw[q + "out_proj.bias"])
        codes.append(idx)
    return torch.stack(codes, dim=1)


# --------------------------------------------------------------------------- speaker embedding (zonos/speaker_cloning.py)
def _spk_bn(x: torch.Tensor, sd: dict, p: str) -> torch.Tensor:
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], False, 0.0, 1e-5)


def spk_simam(X: torch.Tensor, lambda_p: float = 1e-4) -> torch.Tensor:
    """speaker_cloning.py:192-215."""
    n = X.shape[2] * X.shape[3] - 1
    d = (X - X.mean(dim=[2, 3], keepdim=True)).pow(2)
    v = d.sum(dim=[2, 3], keepdim=True) / n
    return X * torch.sigmoid(d / (4 * (v + lambda_p)) + 0.5)


def speaker_embed(sd: dict, feats: torch.Tensor, lda: dict | None = None):
    """ResNet293_based.forward after featCal (speaker_cloning.py:465-472): ResNet of SimAMBasicBlocks (:184-190, :385-392),
    ASP (:128-136), bottleneck; optionally the LDA Linear (:879-881).  feats fp32 [B, n_mels, T] -> (emb [B, 256], lda_emb)."""
    x = F.relu(_spk_bn(F.conv2d(feats.unsqueeze(1), sd["front.conv1.weight"], padding=1), sd, "front.bn1."))
    for li in (1, 2, 3, 4):
        bi = 0
        while f"front.layer{li}.{bi}.conv1.weight" in sd:
            p = f"front.layer{li}.{bi}."
            stride = 2 if (bi == 0 and li > 1) else 1
            out = F.relu(_spk_bn(F.conv2d(x, sd[p + "conv1.weight"], stride=stride, padding=1), sd, p + "bn1."))
            out = spk_simam(_spk_bn(F.conv2d(out, sd[p + "conv2.weight"], padding=1), sd, p + "bn2."))
            res = x
            if p + "downsample.0.weight" in sd:
                res = _spk_bn(F.conv2d(x, sd[p + "downsample.0.weight"], stride=stride), sd, p + "downsample.1.")
            x = F.relu(out + res)
            bi += 1
    B = x.shape[0]
    xf = x.reshape(B, -1, x.shape[-1])                                                   # [B, C*H, T']
    a = F.relu(F.conv1d(xf, sd["pooling.attention.0.weight"], sd["pooling.attention.0.bias"]))
    a = _spk_bn(a, sd, "pooling.attention.2.")
    w = torch.softmax(F.conv1d(a, sd["pooling.attention.3.weight"], sd["pooling.attention.3.bias"]), dim=2)
    mu = torch.sum(xf * w, dim=2)
    sg = torch.sqrt((torch.sum((xf ** 2) * w, dim=2) - mu ** 2).clamp(min=1e-5))
    emb = F.linear(torch.cat((mu, sg), 1), sd["bottleneck.weight"], sd["bottleneck.bias"])
    if lda is None:
        return emb, None
    return emb, F.linear(emb, lda["weight"], lda["bias"])
