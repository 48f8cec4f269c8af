"""CPU checks of the hybrid (Mamba2) restatement in oracle/zonos_oracle.py.

PARITY UNPINNED: mamba_ssm 2.2.5 / causal_conv1d 1.5.2 (the third-party code zonos/backbone/_mamba_ssm.py:45-58 builds
its blocks from) are not in the reference tree and not installed, and the reference holds no fixture for this path
(SURVEY.md 8c).  What CAN be checked without them is that the restated single-token step is the published recurrence:
the Mamba-2 SSD form (arXiv:2405.21060, eq. for h_t = exp(dt_t A) h_{t-1} + dt_t B_t x_t, y_t = C_t h_t + D x_t) unrolled
in float64 over a whole sequence must reproduce the step-by-step outputs."""
import numpy as np
import torch

from oracle import zonos_oracle as zo
from zonos_amd import synth


def _fp32_layer(cfg, seed):
    return {k: v.float() for k, v in synth.mamba2_layer_state_dict(cfg, seed, "L.", torch.float32).items()}


def test_mamba2_step_is_the_published_recurrence():
    cfg = dict(synth.HYBRID_TINY_CFG)
    m = zo.mamba2_dims(cfg)
    w = _fp32_layer(cfg, 5)
    R, T, d = 2, 12, cfg["d_model"]
    xs = torch.from_numpy(synth.normal(5, "x", (T, R, d))).float()
    conv = torch.zeros(R, m["conv_dim"], m["d_conv"])
    ssm = torch.zeros(R, m["nheads"], m["headdim"], m["d_state"])
    ys = torch.stack([zo.mamba2_step(w, "L.mixer.", xs[t], conv, ssm, m) for t in range(T)])       # fp32 roundings only
    # float64 closed form
    W = {k: v.double() for k, v in w.items()}
    di, H, P, N, G = m["d_inner"], m["nheads"], m["headdim"], m["d_state"], m["ngroups"]
    zx = xs.double() @ W["L.mixer.in_proj.weight"].T                                               # [T, R, d_in_proj]
    z, xBC, dt = zx.split([di, m["conv_dim"], H], dim=-1)
    pad = torch.cat([torch.zeros(m["d_conv"] - 1, R, m["conv_dim"], dtype=torch.float64), xBC])
    cw = W["L.mixer.conv1d.weight"].view(-1, m["d_conv"])
    conv_out = sum(pad[i:i + T] * cw[:, i] for i in range(m["d_conv"])) + W["L.mixer.conv1d.bias"]
    act = conv_out * torch.sigmoid(conv_out)
    x, B, C = act.split([di, G * N, G * N], dim=-1)
    dtv = torch.nn.functional.softplus(dt + W["L.mixer.dt_bias"])
    A = -torch.exp(W["L.mixer.A_log"])
    la = dtv * A                                                                                   # log decay per step [T, R, H]
    cum = torch.cumsum(la, dim=0)
    xh = x.view(T, R, H, P)
    y = torch.zeros(T, R, H, P, dtype=torch.float64)
    for t in range(T):
        for s_ in range(t + 1):
            decay = torch.exp(cum[t] - cum[s_])                                                    # prod_{r=s+1..t} exp(dt_r A)
            cb = (C[t].view(R, G, N).repeat_interleave(H // G, 1) * B[s_].view(R, G, N).repeat_interleave(H // G, 1)).sum(-1)
            y[t] += (decay * dtv[s_] * cb)[..., None] * xh[s_]
        y[t] += xh[t] * W["L.mixer.D"][None, :, None]
    v = y.view(T, R, di) * (z * torch.sigmoid(z))
    vg = v.view(T, R, G, di // G)
    o = (vg * torch.rsqrt(vg.pow(2).mean(-1, keepdim=True) + 1e-5)).view(T, R, di) * W["L.mixer.norm.weight"]
    ref = o @ W["L.mixer.out_proj.weight"].T
    err = (ys.double() - ref).abs().max().item()
    print(f"\n[mamba2 step vs float64 SSD closed form] max|diff| {err:.3g} (|ref| max {ref.abs().max():.3g})")
    assert err < 2e-4 * max(1.0, ref.abs().max().item())


def test_conv_window_is_causal_conv1d():
    """The rolling window equals torch's depthwise causal conv1d over the sequence (Mamba2.__init__: Conv1d(groups =
    conv_dim, padding = d_conv - 1) truncated to the sequence length)."""
    cfg = dict(synth.HYBRID_TINY_CFG)
    m = zo.mamba2_dims(cfg)
    w = _fp32_layer(cfg, 9)
    T = 9
    seq = torch.from_numpy(synth.normal(9, "xbc", (1, m["conv_dim"], T))).float()
    ref = torch.nn.functional.conv1d(seq, w["L.mixer.conv1d.weight"], w["L.mixer.conv1d.bias"], padding=m["d_conv"] - 1, groups=m["conv_dim"])[..., :T]
    state = torch.zeros(1, m["conv_dim"], m["d_conv"])
    wf = w["L.mixer.conv1d.weight"].view(-1, m["d_conv"])
    for t in range(T):
        state.copy_(torch.roll(state, -1, -1))
        state[:, :, -1] = seq[:, :, t]
        acc = w["L.mixer.conv1d.bias"][None, :].clone()
        for i in range(m["d_conv"]):
            acc = acc + wf[None, :, i] * state[:, :, i]
        assert torch.allclose(acc, ref[:, :, t], atol=1e-5)


def test_hybrid_generate_runs_and_is_deterministic():
    cfg = dict(synth.HYBRID_TINY_CFG)
    sd = synth.zonos_state_dict(cfg, 31)
    cond = synth.conditioning(31, "cond", 2, 5, cfg["d_model"])
    a = zo.generate(sd, cfg, cond, max_new_tokens=10, cfg_scale=2.0, sampling_params={"temperature": 0.0})
    b = zo.generate(sd, cfg, cond, max_new_tokens=10, cfg_scale=2.0, sampling_params={"temperature": 0.0})
    assert a.shape == (1, 9, 10) and torch.equal(a, b) and int(a.min()) >= 0 and int(a.max()) <= 1023
