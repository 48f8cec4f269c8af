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


def attention_options(cfg: BackboneConfig) -> dict:
    """What `attn_cfg` means for this backbone.  The torch backbone reads only num_heads / num_heads_kv (_torch.py:139-152):
    interleaved-pair RoPE over the whole head, no biases.  The hybrid stack hands attn_cfg to mamba_ssm's MHA
    (create_block(..., attn_cfg=config.attn_cfg), _mamba_ssm.py:45-58), whose defaults apply to every key the
    checkpoint's config.json leaves out: rotary_emb_dim 0 (no rotary), rotary_emb_interleaved False (half-split pairs),
    qkv_proj_bias / out_proj_bias True.  Forms the kernels do not implement are refused here, never run differently."""
    ac = dict(cfg.attn_cfg)
    h, hkv = int(ac["num_heads"]), int(ac.get("num_heads_kv") or ac["num_heads"])
    hd = cfg.d_model // h
    if not cfg.ssm_cfg:
        return dict(num_heads=h, num_heads_kv=hkv, head_dim=hd, rope_mode=0, qkv_bias=False, out_bias=False)
    known = {"num_heads", "num_heads_kv", "head_dim", "mlp_dim", "qkv_proj_bias", "out_proj_bias", "softmax_scale", "causal", "d_conv",
             "rotary_emb_dim", "rotary_emb_base", "rotary_emb_interleaved"}
    unknown = set(ac) - known
    if unknown:
        raise _lib.ZonosHipError(f"attn_cfg keys {sorted(unknown)} are not arguments of mamba_ssm's MHA")
    if ac.get("head_dim") not in (None, hd):
        raise _lib.ZonosHipError(f"attn_cfg head_dim {ac['head_dim']} != d_model / num_heads = {hd} is not supported")
    if ac.get("mlp_dim", 0) or ac.get("d_conv", 0):
        raise _lib.ZonosHipError("attn_cfg mlp_dim / d_conv (MHA with a fused MLP or a conv on qkv) are not supported")
    if not ac.get("causal", False):
        raise _lib.ZonosHipError("attn_cfg causal must be true (mamba_ssm's default is false: the decode path has no non-causal form)")
    scale = ac.get("softmax_scale")
    if scale is not None and abs(float(scale) - hd ** -0.5) > 1e-9:
        raise _lib.ZonosHipError("attn_cfg softmax_scale other than 1/sqrt(head_dim) is not supported")
    rot = int(ac.get("rotary_emb_dim", 0))
    if rot not in (0, hd):
        raise _lib.ZonosHipError(f"attn_cfg rotary_emb_dim {rot}: only 0 or head_dim ({hd}) are supported (no partial rotary)")
    if float(ac.get("rotary_emb_base", 10000.0)) != 10000.0:
        raise _lib.ZonosHipError("attn_cfg rotary_emb_base other than 10000 is not supported")
    mode = 2 if rot == 0 else (0 if ac.get("rotary_emb_interleaved", False) else 1)
    return dict(num_heads=h, num_heads_kv=hkv, head_dim=hd, rope_mode=mode, qkv_bias=bool(ac.get("qkv_proj_bias", True)),
                out_bias=bool(ac.get("out_proj_bias", True)))


class _Mixer(nn.Module):
    def __init__(self, cfg: BackboneConfig):
        super().__init__()
        o = attention_options(cfg)
        h, hkv, hd = o["num_heads"], o["num_heads_kv"], o["head_dim"]
        self.in_proj = nn.Linear(cfg.d_model, (h + 2 * hkv) * hd, bias=o["qkv_bias"])
        self.out_proj = nn.Linear(h * hd, cfg.d_model, bias=o["out_bias"])


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
        rms = bool(cfg.ssm_cfg) and cfg.rms_norm          # mamba_ssm RMSNorm: a weight, no bias (create_block's norm_cls)
        mk_norm = (lambda: _RmsWeight(cfg.d_model)) if rms else (lambda: nn.LayerNorm(cfg.d_model, eps=cfg.norm_epsilon))
        self.norm = mk_norm()
        self.is_mamba = bool(cfg.ssm_cfg) and layer_idx not in cfg.attn_layer_idx
        if self.is_mamba:
            self.mixer = _Mamba2Mixer(cfg)
            if cfg.d_intermediate:
                raise _lib.ZonosHipError("Mamba2 layers with an MLP (d_intermediate > 0) are not supported")
        else:
            self.mixer = _Mixer(cfg)
            self.norm2 = mk_norm()
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
        ao = attention_options(cfg)
        h, hkv = ao["num_heads"], ao["num_heads_kv"]
        self.head_dim = cfg.d_model // h
        rope = rope_table(ROPE_POSITIONS, self.head_dim)
        if cfg.ssm_cfg and ao["rope_mode"] in (0, 1):
            # flash_attn's RotaryEmbedding (what mamba_ssm's MHA applies) caches cos / sin in the activations' dtype, for the
            # interleaved form as for the half-split one (the transformer stack's own RoPE, _torch.py:29-68, stays fp32)
            rope = rope.to(torch.bfloat16).float()
        self.rope = rope.to(dev)
        zc = _lib.zn_config(d_model=cfg.d_model, n_layer=cfg.n_layer, n_heads=h, n_heads_kv=hkv, d_ff=cfg.attn_mlp_d_intermediate,
                            n_codebooks=n_codebooks, vocab_head=vocab_head, vocab_embed=vocab_embed, eos_id=eos_id, mask_id=mask_id,
                            rope_positions=ROPE_POSITIONS, double_out_proj=int(double_out_proj), norm_eps=cfg.norm_epsilon)
        if cfg.ssm_cfg:   # hybrid: mamba_ssm Block semantics, Mamba2 mixers outside attn_layer_idx
            m = mamba2_dims(cfg)
            zc.arch, zc.double_out_proj = 1, 0
            zc.rms_norm, zc.residual_in_fp32, zc.rope_mode = int(cfg.rms_norm), int(cfg.residual_in_fp32), ao["rope_mode"]
            zc.m_d_inner, zc.m_headdim, zc.m_d_state, zc.m_ngroups, zc.m_d_conv = m["d_inner"], m["headdim"], m["d_state"], m["ngroups"], m["d_conv"]
        self.zc = zc
        self._keep = [self.rope]
        lw = (_lib.zn_layer_weights * cfg.n_layer)()
        for i, blk in enumerate(backbone.layers):
            if blk.is_mamba:
                mx = blk.mixer
                nb = getattr(blk.norm, "bias", None)
                ts = [blk.norm.weight, nb if nb is not None else blk.norm.weight, mx.in_proj.weight, mx.conv1d.weight, mx.conv1d.bias, mx.dt_bias, mx.A_log, mx.D, mx.norm.weight,
                      mx.out_proj.weight]
                for t in ts:
                    assert t.is_contiguous()
                lw[i].kind = 1
                (lw[i].norm_w, lw[i].norm_b, lw[i].m_in_proj, lw[i].m_conv_w, lw[i].m_conv_b, lw[i].m_dt_bias, lw[i].m_A_log, lw[i].m_D, lw[i].m_norm_w,
                 lw[i].m_out_proj) = [t.data_ptr() for t in ts]
                if nb is None:
                    lw[i].norm_b = None                      # RMSNorm: no bias
                self._keep += ts
                continue
            nb, nb2 = getattr(blk.norm, "bias", None), getattr(blk.norm2, "bias", None)
            ts = [blk.norm.weight, nb if nb is not None else blk.norm.weight, blk.mixer.in_proj.weight, blk.mixer.out_proj.weight, blk.norm2.weight,
                  nb2 if nb2 is not None else blk.norm2.weight, blk.mlp.fc1.weight, blk.mlp.fc2.weight]
            for t in ts:
                assert t.is_contiguous()
            (lw[i].norm_w, lw[i].norm_b, lw[i].in_proj, lw[i].out_proj, lw[i].norm2_w, lw[i].norm2_b, lw[i].fc1, lw[i].fc2) = [t.data_ptr() for t in ts]
            if nb is None:
                lw[i].norm_b, lw[i].norm2_b = None, None
            for name, lin in (("in_proj_bias", blk.mixer.in_proj), ("out_proj_bias", blk.mixer.out_proj)):
                if lin.bias is not None:
                    setattr(lw[i], name, lin.bias.data_ptr())
                    ts.append(lin.bias)
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

    def counters(self) -> dict:
        """zn_get_counters: hand-off timeouts reported, generations, generations that ran the launches path because of a demotion, ..."""
        buf = (C.c_int64 * 8)()
        self.call("zn_get_counters", buf, 8)
        keys = ("handoff_timeouts", "generations", "fallback_generations", "demoted", "rearms", "clean_since_demotion", "longest_wait_us", "waits_over_200us")
        return dict(zip(keys, (int(v) for v in buf)))

    def stream(self) -> int:
        """The caller's current torch stream on this engine's device."""
        return torch.cuda.current_stream(self.device).cuda_stream


class HipZonosBackbone(nn.Module):
    supported_architectures = ["transformer", "hybrid"]      # _mamba_ssm.py:25

    def __init__(self, config: BackboneConfig):
        if config.ssm_cfg and config.ssm_cfg.get("layer", "Mamba1") != "Mamba2":
            raise _lib.ZonosHipError("only ssm_cfg layer = 'Mamba2' is supported (the Zonos hybrid checkpoints' mixer)")
        if not config.ssm_cfg and (config.rms_norm or config.residual_in_fp32):
            # the torch backbone never reads these fields (_torch.py:130-156 builds nn.LayerNorm blocks unconditionally)
            pass
        attention_options(config)          # refuses attn_cfg forms the kernels do not implement
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
        """_torch.py:213-238 / _mamba_ssm.py:88-119: hidden [R, S, d] -> [R, S, d] after the final norm, the caches advanced by S
        positions written at inference_params.seqlen_offset (== lengths_per_sample in every reference call site).  S > 1
        runs the batched prefill kernels (Mamba2 layers: sequence conv + selective scan), S == 1 the decode kernels."""
        R, S, d = hidden_states.shape
        eng = self.engine(R + (R & 1))
        base = int(inference_params.seqlen_offset)
        lens = inference_params.lengths_per_sample
        if lens is not None and not bool((lens == base).all()):
            raise _lib.ZonosHipError("HipZonosBackbone.forward: lengths_per_sample must equal seqlen_offset for every row")
        caches, max_len = [], int(inference_params.max_seqlen)
        for li, blk in enumerate(self.layers):
            c0 = inference_params.key_value_memory_dict[li][0]
            caches.append(c0.data_ptr())           # KV cache, or the start of the Mamba2 layer's (conv, ssm) state buffer
            if not blk.is_mamba:
                max_len = c0.shape[1]
                if c0.shape[0] != R:
                    raise _lib.ZonosHipError(f"KV cache holds {c0.shape[0]} rows, hidden_states {R}")
            elif c0.shape[0] != R:
                raise _lib.ZonosHipError(f"Mamba2 state holds {c0.shape[0]} rows, hidden_states {R}")
        ptrs = (C.c_void_p * len(caches))(*caches)
        x = hidden_states.to(torch.bfloat16).contiguous()
        out = torch.empty_like(x)
        eng.call("zn_op_backbone_forward", x.data_ptr(), out.data_ptr(), ptrs, max_len, base, S, R, eng.stream())
        return out
