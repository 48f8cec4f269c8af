"""ctypes binding of libzonos_hip.so (C ABI: include/zonos_hip.h).

The HIP library is the product path; there is no CPU fallback.  Loading fails loudly if the shared object is
missing (run `python -m zonos_amd.build`), and every call raises `ZonosHipError` on a non-zero status.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ZONOS_HIP_LIB") or os.path.join(_HERE, "libzonos_hip.so")   # the override selects an experimental build (A/B runs)

ZN_ABI_VERSION = 5


class ZonosHipError(RuntimeError):
    pass


class zn_config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("d_model", "n_layer", "n_heads", "n_heads_kv", "d_ff", "n_codebooks", "vocab_head",
                                          "vocab_embed", "eos_id", "mask_id", "rope_positions", "double_out_proj")] + [("norm_eps", C.c_float)] + \
               [(n, C.c_int32) for n in ("arch", "m_d_inner", "m_headdim", "m_d_state", "m_ngroups", "m_d_conv", "rms_norm", "residual_in_fp32", "rope_mode")]


class zn_layer_weights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("norm_w", "norm_b", "in_proj", "out_proj", "norm2_w", "norm2_b", "fc1", "fc2")] + [("kind", C.c_int32)] + \
               [(n, C.c_void_p) for n in ("m_in_proj", "m_conv_w", "m_conv_b", "m_dt_bias", "m_A_log", "m_D", "m_norm_w", "m_out_proj", "in_proj_bias", "out_proj_bias")]


class zn_weights(C.Structure):
    _fields_ = [("embeddings", C.POINTER(C.c_void_p)), ("heads", C.c_void_p), ("norm_f_w", C.c_void_p), ("norm_f_b", C.c_void_p),
                ("layers", C.POINTER(zn_layer_weights)), ("rope_table", C.c_void_p)]


class zn_sampling(C.Structure):
    _fields_ = [("temperature", C.c_float), ("top_p", C.c_float), ("top_k", C.c_int32), ("min_p", C.c_float),
                ("linear", C.c_float), ("conf", C.c_float), ("quad", C.c_float), ("repetition_penalty", C.c_float),
                ("repetition_penalty_window", C.c_int32), ("seed", C.c_uint64)]


class zn_dac_config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_codebooks", "codebook_size", "codebook_dim", "hidden_size", "decoder_hidden_size", "n_ratios")] + \
               [("ratios", C.c_int32 * 8), ("encoder_hidden_size", C.c_int32)]


class zn_dac_tensor(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data_dev", C.c_void_p), ("numel", C.c_int64)]


# name -> (restype, argtypes); every symbol include/zonos_hip.h declares
SIGNATURES = {
    "zn_abi_version": (C.c_int, []),
    "zn_create": (C.c_int, [C.POINTER(zn_config), C.POINTER(zn_weights), C.c_int32, C.POINTER(C.c_void_p)]),
    "zn_destroy": (C.c_int, [C.c_void_p]),
    "zn_last_error": (C.c_char_p, [C.c_void_p]),
    "zn_kv_bytes_per_layer": (C.c_size_t, [C.POINTER(zn_config), C.c_int32, C.c_int32]),
    "zn_mamba_state_bytes_per_layer": (C.c_size_t, [C.POINTER(zn_config), C.c_int32, C.POINTER(C.c_size_t)]),
    "zn_gen_begin": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                               C.c_int32, C.c_float, C.POINTER(zn_sampling), C.c_void_p]),
    "zn_prefill": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "zn_sample_first": (C.c_int, [C.c_void_p, C.c_void_p]),
    "zn_decode_steps": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "zn_graph_active": (C.c_int, [C.c_void_p]),
    "zn_gen_end": (C.c_int, [C.c_void_p]),
    "zn_tenant_try_claim": (C.c_int, [C.c_int32, C.c_void_p]),
    "zn_tenant_release": (C.c_int, [C.c_int32, C.c_void_p]),
    "zn_decode_path": (C.c_int, [C.c_void_p]),
    "zn_decode_path_detail": (C.c_int, [C.c_void_p]),
    "zn_get_counters": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int32]),
    "zn_all_stopped": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "zn_all_stopped_begin": (C.c_int, [C.c_void_p, C.c_void_p]),
    "zn_all_stopped_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "zn_get_step_outputs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "zn_debug_force_eos": (C.c_int, [C.c_void_p, C.c_int32]),
    "zn_debug_token_override": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    "zn_codes_changed": (C.c_int, [C.c_void_p]),
    "zn_debug_prefill_mode": (C.c_int, [C.c_void_p, C.c_int32]),
    "zn_debug_tune": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "zn_debug_eos_bias": (C.c_int, [C.c_void_p, C.c_float]),
    "zn_debug_chain_stamps": (C.c_int, [C.c_void_p, C.c_void_p]),
    "zn_debug_trace": (C.c_int, [C.c_void_p, C.c_void_p]),
    "zn_bench_kernel": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_double), C.c_void_p]),
    "zn_op_linear": (C.c_int, [C.c_void_p] + [C.c_void_p] * 5 + [C.c_int32] * 3 + [C.c_void_p]),
    "zn_op_linear_bias": (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 3 + [C.c_void_p]),
    "zn_op_gather_rows": (C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 4 + [C.c_void_p]),
    "zn_op_fourier": (C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 3 + [C.c_float, C.c_float, C.c_void_p]),
    "zn_op_silu": (C.c_int, [C.c_void_p] * 3 + [C.c_int64, C.c_void_p]),
    "zn_op_layernorm": (C.c_int, [C.c_void_p] * 5 + [C.c_int32, C.c_int32, C.c_void_p]),
    "zn_op_layer_decode": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "zn_op_backbone_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "zn_op_attn_prefill": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "zn_op_attn_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "zn_op_embed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "zn_op_add_layernorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p]),
    "zn_op_mamba_step": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "zn_op_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.POINTER(zn_sampling), C.c_uint64, C.c_void_p, C.c_void_p,
                               C.c_int32, C.c_void_p]),
    "zn_dac_create": (C.c_int, [C.POINTER(zn_dac_config), C.POINTER(zn_dac_tensor), C.c_int32, C.POINTER(C.c_void_p)]),
    "zn_dac_destroy": (C.c_int, [C.c_void_p]),
    "zn_dac_last_error": (C.c_char_p, [C.c_void_p]),
    "zn_dac_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "zn_dac_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "zn_spk_create": (C.c_int, [C.POINTER(zn_dac_tensor), C.c_int32, C.POINTER(C.c_void_p)]),
    "zn_spk_destroy": (C.c_int, [C.c_void_p]),
    "zn_spk_last_error": (C.c_char_p, [C.c_void_p]),
    "zn_spk_embed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
}

_lib = None


def load() -> C.CDLL:
    """Load the shared object and bind every declared symbol.  Raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("ZONOS_HIP_LIB_VARIANT")      # development only (tools/build_variants.py: A/B of compile-time kernel parameters)
    if path:
        path = os.path.join(os.path.dirname(_HERE), "build", "variants", f"libzonos_hip_{path}.so")
        if not os.path.exists(path):
            raise ZonosHipError(f"{path} not found (ZONOS_HIP_LIB_VARIANT)")
        print(f"[zonos_amd] loading variant library {path}", file=sys.stderr, flush=True)
    else:
        path = LIB_PATH
    if not os.path.exists(path):
        raise ZonosHipError(f"{path} not found: build it with `python -m zonos_amd.build` "
                            "(the HIP library is the only execution path; there is no CPU fallback)")
    import torch  # noqa: F401  -- must come first: the library shares torch's HIP runtime (same libamdhip64 SONAME);
    #                      loading /opt/rocm's copy before torch's leaves two runtimes in the process
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype, fn.argtypes = res, args
    if lib.zn_abi_version() != ZN_ABI_VERSION:
        raise ZonosHipError(f"ABI mismatch: library {lib.zn_abi_version()} vs binding {ZN_ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int, handle=None, what: str = "") -> None:
    if rc != 0:
        msg = load().zn_last_error(handle)
        raise ZonosHipError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")


def check_dac(rc: int, handle=None, what: str = "") -> None:
    if rc != 0:
        msg = load().zn_dac_last_error(handle)
        raise ZonosHipError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")


def stream_ptr() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()
