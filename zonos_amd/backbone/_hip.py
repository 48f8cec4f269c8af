"""HIP transformer / hybrid backbone behind the reference's backbone plugin contract.

Contract (zonos/backbone/__init__.py:24-36, _torch.py:130,157,213): a class with `supported_architectures`,
`__init__(config: BackboneConfig)`, `allocate_inference_cache(batch_size, max_seqlen, dtype) -> {layer: (kv, None)}`
and `forward(hidden_states[R,S,d], inference_params) -> [R,S,d]`, whose parameter names equal the reference's
(`layers.{i}.norm|norm2.{weight,bias}`, `layers.{i}.mixer.{in_proj,out_proj}.weight`, `layers.{i}.mlp.{fc1,fc2}.weight`,
`norm_f.{weight,bias}`) so one safetensors file serves both.  All arithmetic runs in libzonos_hip.so.
"""
from __future__ import annotations

import ctypes as C
import threading

import torch
import torch.nn as nn

from .. import _lib
from ..config import BackboneConfig, InferenceParams

ROPE_POSITIONS = 16384  # zonos/backbone/_torch.py:206


def rope_table(n_pos: int, head_dim: int, base: float = 10000.0) -> torch.Tensor:
    """fp32 [n_pos, head_dim/2, 2] (cos, sin) built on the host CPU with the same torch ops the reference uses
    (zonos/backbone/_torch.py:29-34), so the table bits equal the reference's."""
    inv = 1.0 / (base ** (torch.arange(0, head_dim, 2)[: head_dim // 2].float() / head_dim))
    z = torch.polar(torch.ones(n_pos, head_dim // 2), torch.outer(torch.arange(n_pos), inv))
    return torch.stack([z.real, z.imag], dim=-1).contiguous()


class _Mixer(nn.Module):
    def __init__(self, cfg: BackboneConfig):
        super().__init__()
        h, hkv = cfg.attn_cfg["num_heads"], cfg.attn_cfg["num_heads_kv"]
        hd = cfg.d_model // h
        self.in_proj = nn.Linear(cfg.d_model, (h + 2 * hkv) * hd, bias=False)
        self.out_proj = nn.Linear(h * hd, cfg.d_model, bias=False)


class _Mlp(nn.Module):
    def __init__(self, cfg: BackboneConfig):
        super().__init__()
        self.fc1 = nn.Linear(cfg.d_model, 2 * cfg.attn_mlp_d_intermediate, bias=False)
        self.fc2 = nn.Linear(cfg.attn_mlp_d_intermediate, cfg.d_model, bias=False)


def mamba2_dims(cfg: BackboneConfig) -> dict:
    """Mamba2 hyper-parameters from BackboneConfig.ssm_cfg with the library's defaults (mamba_ssm 2.2.5 Mamba2.__init__,
    the class _mamba_ssm.py:45-58 instantiates through create_block)."""
    sc = cfg.ssm_cfg
    d_inner = int(sc.get("expand", 2)) * cfg.d_model
    headdim, d_state, ngroups, d_conv = int(sc.get("headdim", 64)), int(sc.get("d_state", 128)), int(sc.get("ngroups", 1)), int(sc.get("d_conv", 4))
    return dict(d_inner=d_inner, headdim=headdim, nheads=d_inner // headdim, d_state=d_state, ngroups=ngroups, d_conv=d_conv,
                conv_dim=d_inner + 2 * ngroups * d_state, d_in_proj=2 * d_inner + 2 * ngroups * d_state + d_inner // headdim)


class _RmsWeight(nn.Module):
    def __init__(self, n: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(n))


class _Mamba2Mixer(nn.Module):
    """Parameter container with mamba_ssm Mamba2's names: in_proj, conv1d, dt_bias, A_log, D, norm, out_proj."""
    def __init__(self, cfg: BackboneConfig):
        super().__init__()
        m = mamba2_dims(cfg)
        self.in_proj = nn.Linear(cfg.d_model, m["d_in_proj"], bias=False)
        self.conv1d = nn.Conv1d(m["conv_dim"], m["conv_dim"], m["d_conv"], groups=m["conv_dim"], padding=m["d_conv"] - 1, bias=True)
        self.dt_bias = nn.Parameter(torch.zeros(m["nheads"]))
        self.A_log = nn.Parameter(torch.zeros(m["nheads"]))
        self.D = nn.Parameter(torch.ones(m["nheads"]))
        self.norm = _RmsWeight(m["d_inner"])
        self.out_proj = nn.Linear(m["d_inner"], cfg.d_model, bias=False)


class _Block(nn.Module):
    """Parameter container only (names = _torch.py:278-281 / mamba_ssm Block); the math is the HIP kernels'."""
    def __init__(self, cfg: BackboneConfig, layer_idx: int = 0):
        super().__init__()
        self.norm = nn.LayerNorm(cfg.d_model, eps=cfg.norm_epsilon)
        self.is_mamba = bool(cfg.ssm_cfg) and layer_idx not in cfg.attn_layer_idx
        if self.is_mamba:
            self.mixer = _Mamba2Mixer(cfg)
            if cfg.d_intermediate:
                raise _lib.ZonosHipError("Mamba2 layers with an MLP (d_intermediate > 0) are not supported")
        else:
            self.mixer = _Mixer(cfg)
            self.norm2 = nn.LayerNorm(cfg.d_model, eps=cfg.norm_epsilon)
            self.mlp = _Mlp(cfg)


class HipEngine:
    """Owns one zn_handle bound to a set of device weights."""

    def __init__(self, backbone: "HipZonosBackbone", embeddings=None, heads=None, max_rows: int = 2,
                 double_out_proj: bool = True, n_codebooks: int = 9, vocab_head: int = 1025, vocab_embed: int = 1032,
                 eos_id: int = 1024, mask_id: int = 1025):
        cfg = backbone.config
        dev = backbone.norm_f.weight.device
        if dev.type != "cuda":
            raise _lib.ZonosHipError("zonos_amd runs on MI355X only: move the model to a cuda device (no CPU fallback)")
        for p in backbone.parameters():
            if p.dtype != torch.bfloat16:
                raise _lib.ZonosHipError("weights must be bfloat16 (zonos/model.py:158)")
        self.lib = _lib.load()
        h, hkv = cfg.attn_cfg["num_heads"], cfg.attn_cfg["num_heads_kv"]
        self.head_dim = cfg.d_model // h
        self.rope = rope_table(ROPE_POSITIONS, self.head_dim).to(dev)
        zc = _lib.zn_config(d_model=cfg.d_model, n_layer=cfg.n_layer, n_heads=h, n_heads_kv=hkv, d_ff=cfg.attn_mlp_d_intermediate,
                            n_codebooks=n_codebooks, vocab_head=vocab_head, vocab_embed=vocab_embed, eos_id=eos_id, mask_id=mask_id,
                            rope_positions=ROPE_POSITIONS, double_out_proj=int(double_out_proj), norm_eps=cfg.norm_epsilon)
        if cfg.ssm_cfg:   # hybrid: mamba_ssm Block semantics, Mamba2 mixers outside attn_layer_idx
            m = mamba2_dims(cfg)
            zc.arch, zc.double_out_proj = 1, 0
            zc.m_d_inner, zc.m_headdim, zc.m_d_state, zc.m_ngroups, zc.m_d_conv = m["d_inner"], m["headdim"], m["d_state"], m["ngroups"], m["d_conv"]
        self.zc = zc
        self._keep = [self.rope]
        lw = (_lib.zn_layer_weights * cfg.n_layer)()
        for i, blk in enumerate(backbone.layers):
            if blk.is_mamba:
                mx = blk.mixer
                ts = [blk.norm.weight, blk.norm.bias, mx.in_proj.weight, mx.conv1d.weight, mx.conv1d.bias, mx.dt_bias, mx.A_log, mx.D, mx.norm.weight,
                      mx.out_proj.weight]
                for t in ts:
                    assert t.is_contiguous()
                lw[i].kind = 1
                (lw[i].norm_w, lw[i].norm_b, lw[i].m_in_proj, lw[i].m_conv_w, lw[i].m_conv_b, lw[i].m_dt_bias, lw[i].m_A_log, lw[i].m_D, lw[i].m_norm_w,
                 lw[i].m_out_proj) = [t.data_ptr() for t in ts]
                self._keep += ts
                continue
            ts = [blk.norm.weight, blk.norm.bias, blk.mixer.in_proj.weight, blk.mixer.out_proj.weight, blk.norm2.weight, blk.norm2.bias,
                  blk.mlp.fc1.weight, blk.mlp.fc2.weight]
            for t in ts:
                assert t.is_contiguous()
            (lw[i].norm_w, lw[i].norm_b, lw[i].in_proj, lw[i].out_proj, lw[i].norm2_w, lw[i].norm2_b, lw[i].fc1, lw[i].fc2) = [t.data_ptr() for t in ts]
            self._keep += ts
        w = _lib.zn_weights()
        if embeddings is not None:
            emb = (C.c_void_p * n_codebooks)(*[e.data_ptr() for e in embeddings])
            w.embeddings = emb
            w.heads = heads.data_ptr()
            self._keep += list(embeddings) + [heads]
        w.norm_f_w, w.norm_f_b = backbone.norm_f.weight.data_ptr(), backbone.norm_f.bias.data_ptr()
        w.layers = lw
        w.rope_table = self.rope.data_ptr()
        self.max_rows = max_rows
        self.device = dev
        # One handle carries all per-generation state (KV pointers, captured graphs, tickets) and zn_* forbids overlapping
        # calls on it: callers that span several calls (Zonos.generate) hold this lock for the whole section; the reference
        # serves two concurrent requests per model (utilities/app_constants.py:18), here they queue.
        self.lock = threading.RLock()
        handle = C.c_void_p()
        with torch.cuda.device(dev):
            _lib.check(self.lib.zn_create(C.byref(zc), C.byref(w), max_rows, C.byref(handle)), None, "zn_create")
        self.h = handle

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.zn_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def call(self, name: str, *args):
        """One library call, serialised per handle, with the handle's device current (launches, streams and workspace
        allocations inside the library go to the device the weights live on, whatever the caller's current device)."""
        with self.lock, torch.cuda.device(self.device):
            _lib.check(getattr(self.lib, name)(self.h, *args), self.h, name)

    def stream(self) -> int:
        """The caller's current torch stream on this engine's device."""
        return torch.cuda.current_stream(self.device).cuda_stream


class HipZonosBackbone(nn.Module):
    supported_architectures = ["transformer", "hybrid"]      # _mamba_ssm.py:25

    def __init__(self, config: BackboneConfig):
        if config.ssm_cfg and config.ssm_cfg.get("layer", "Mamba1") != "Mamba2":
            raise _lib.ZonosHipError("only ssm_cfg layer = 'Mamba2' is supported (the Zonos hybrid checkpoints' mixer)")
        if config.rms_norm or config.residual_in_fp32:
            raise _lib.ZonosHipError("rms_norm / residual_in_fp32 backbones are not supported (Zonos checkpoints use neither)")
        super().__init__()
        self.config = config
        self.layers = nn.ModuleList(_Block(config, i) for i in range(config.n_layer))
        self.norm_f = nn.LayerNorm(config.d_model, eps=config.norm_epsilon)
        self.ref_double_out_proj = True   # reproduce zonos/backbone/_torch.py:419-420
        self._engine: HipEngine | None = None

    def engine(self, max_rows: int = 2) -> HipEngine:
        e = self._engine
        if e is None or e.max_rows < max_rows or e.device != self.norm_f.weight.device:
            self._engine = e = HipEngine(self, max_rows=max(2, max_rows), double_out_proj=self.ref_double_out_proj)
        return e

    def allocate_inference_cache(self, batch_size: int, max_seqlen: int, dtype: torch.dtype = torch.bfloat16):
        """_torch.py:157-211: per attention layer (kv [R, maxL, 2, Hkv, hd], None); _mamba_ssm.py:65-86: per Mamba2
        layer (conv_state [R, conv_dim, d_conv], ssm_state [R, nheads, headdim, d_state]), zero-initialised, both views
        of one contiguous buffer (the layout zn_mamba_state_bytes_per_layer describes)."""
        if dtype != torch.bfloat16:
            raise _lib.ZonosHipError("the KV cache is bfloat16 (zonos/model.py:305)")
        cfg = self.config
        hd = cfg.d_model // cfg.attn_cfg["num_heads"]
        dev = self.norm_f.weight.device
        out = {}
        for i, blk in enumerate(self.layers):
            if blk.is_mamba:
                m = mamba2_dims(cfg)
                n_conv = batch_size * m["conv_dim"] * m["d_conv"]
                buf = torch.zeros(n_conv + batch_size * m["d_inner"] * m["d_state"], dtype=dtype, device=dev)
                out[i] = (buf[:n_conv].view(batch_size, m["conv_dim"], m["d_conv"]),
                          buf[n_conv:].view(batch_size, m["nheads"], m["headdim"], m["d_state"]))
            else:
                out[i] = (torch.empty(batch_size, max_seqlen, 2, cfg.attn_cfg["num_heads_kv"], hd, dtype=dtype, device=dev), None)
        return out

    @torch.inference_mode()
    def forward(self, hidden_states: torch.Tensor, inference_params: InferenceParams) -> torch.Tensor:
        """_torch.py:213-238.  Position by position through the decode kernels; KV is appended at
        lengths_per_sample + s (== seqlen_offset + s in every reference call site)."""
        if self.config.ssm_cfg:
            raise _lib.ZonosHipError("HipZonosBackbone.forward: the hybrid stack runs through Zonos.generate (device-resident loop) only")
        R, S, d = hidden_states.shape
        eng = self.engine(R + (R & 1))
        st = eng.stream()
        lengths = inference_params.lengths_per_sample.to(device=hidden_states.device, dtype=torch.int32).clone()
        out = torch.empty_like(hidden_states)
        qb = 256 if S >= 768 else 64 if S >= 192 else 32      # CPU flash-attention query split
        base = int(inference_params.seqlen_offset)
        for s in range(S):
            x = hidden_states[:, s].contiguous()
            ext = torch.full((R,), base + min((s // qb) * qb + qb, S), dtype=torch.int32, device=x.device)
            for li in range(self.config.n_layer):
                kv = inference_params.key_value_memory_dict[li][0]
                eng.call("zn_op_layer_decode", li, x.data_ptr(), kv.data_ptr(), kv.shape[1], lengths.data_ptr(), ext.data_ptr(), R, st)
            y = torch.empty_like(x)
            eng.call("zn_op_layernorm", x.data_ptr(), self.norm_f.weight.data_ptr(), self.norm_f.bias.data_ptr(), y.data_ptr(), R, d, st)
            out[:, s] = y
            lengths += 1
        return out
