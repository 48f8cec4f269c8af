"""`DACAutoencoder` — zonos/autoencoder.py:50-170 surface over the HIP DAC decoder.

`decode(codes) -> float32 [B, 1, 512*T]` and `decode_to_int16` run in libzonos_hip.so (fp32 arithmetic: the
reference's CPU path disables autocast, autoencoder.py:139).  Weights use the transformers `DacModel`
state-dict names (the reference loads `descript/dac_44khz`, autoencoder.py:74 — a network fetch, so here they
come from a local safetensors file or a state dict).  `preprocess`/`encode` are the "next" row.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib

# transformers DacConfig defaults = descript/dac_44khz (configuration_dac.py:55-70)
DAC_44KHZ = dict(n_codebooks=9, codebook_size=1024, codebook_dim=8, hidden_size=1024, decoder_hidden_size=1536,
                 upsampling_ratios=(8, 8, 4, 2), sampling_rate=44100)


class DACAutoencoder:
    def __init__(self, state_dict: dict | None = None, config: dict | None = None, device=None):
        self.cfg = dict(DAC_44KHZ, **(config or {}))
        self.codebook_size = self.cfg["codebook_size"]
        self.num_codebooks = self.cfg["n_codebooks"]
        self.sampling_rate = self.cfg["sampling_rate"]
        self.hop = int(np.prod(self.cfg["upsampling_ratios"]))
        self._weights: dict | None = None
        self._h = None
        self.device = torch.device(device) if device is not None else None
        if state_dict is not None:
            self.load_state_dict(state_dict, device)

    @classmethod
    def from_local(cls, path: str, device="cuda") -> "DACAutoencoder":
        """Load a transformers DacModel checkpoint (model.safetensors) from disk."""
        from safetensors.torch import load_file
        return cls(load_file(path), device=device)

    def load_state_dict(self, sd: dict, device=None):
        dev = torch.device(device if device is not None else (self.device or "cuda"))
        keep = {k: v.detach().to(device=dev, dtype=torch.float32).contiguous() for k, v in sd.items()
                if k.startswith("quantizer.") and (".codebook.weight" in k or ".out_proj." in k) or k.startswith("decoder.")}
        self._weights, self.device = keep, dev
        self._destroy()

    def to(self, device):
        if self._weights is not None:
            self.load_state_dict(self._weights, device)
        else:
            self.device = torch.device(device)
        return self

    def _destroy(self):
        if self._h is not None:
            _lib.load().zn_dac_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def _handle(self):
        if self._h is not None:
            return self._h
        if self._weights is None:
            raise _lib.ZonosHipError("DACAutoencoder has no weights: the reference fetches descript/dac_44khz "
                                     "(zonos/autoencoder.py:74); offline, use DACAutoencoder.from_local(path) or pass a state dict")
        if self.device.type != "cuda":
            raise _lib.ZonosHipError("zonos_amd runs on MI355X only (no CPU fallback)")
        lib = _lib.load()
        c = self.cfg
        zc = _lib.zn_dac_config(n_codebooks=c["n_codebooks"], codebook_size=c["codebook_size"], codebook_dim=c["codebook_dim"],
                                hidden_size=c["hidden_size"], decoder_hidden_size=c["decoder_hidden_size"], n_ratios=len(c["upsampling_ratios"]))
        for i, r in enumerate(c["upsampling_ratios"]):
            zc.ratios[i] = r
        names = sorted(self._weights)
        arr = (_lib.zn_dac_tensor * len(names))()
        for i, k in enumerate(names):
            arr[i].name, arr[i].data_dev, arr[i].numel = k.encode(), self._weights[k].data_ptr(), self._weights[k].numel()
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check_dac(lib.zn_dac_create(C.byref(zc), arr, len(names), C.byref(h)), None, "zn_dac_create")
        self._h = h
        return h

    def preprocess(self, wav: torch.Tensor, sr: int) -> torch.Tensor:
        raise NotImplementedError("DAC preprocess/encode is the next row after the hot path (SURVEY.md §8f #2)")

    def encode(self, wav: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError("DAC preprocess/encode is the next row after the hot path (SURVEY.md §8f #2)")

    @torch.inference_mode()
    def decode(self, codes: torch.Tensor) -> torch.Tensor:
        """autoencoder.py:119-140: codes [B, 9, T] (ints in [0, 1023]) -> float32 [B, 1, 512*T]."""
        h = self._handle()
        B, nq, T = codes.shape
        if nq != self.num_codebooks:
            raise ValueError(f"expected {self.num_codebooks} codebooks, got {nq}")
        c32 = codes.to(device=self.device, dtype=torch.int32).contiguous()
        wav = torch.empty(B, 1, self.hop * T, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check_dac(_lib.load().zn_dac_decode(h, c32.data_ptr(), B, T, wav.data_ptr(), _lib.stream_ptr()), h, "zn_dac_decode")
        return wav

    @torch.inference_mode()
    def decode_to_int16(self, codes: torch.Tensor) -> torch.Tensor:
        """autoencoder.py:142-170: clamp(wav * 32767, +-32767) -> int16 [512*T, 1] (batch 1)."""
        wav = self.decode(codes).squeeze(1)
        return torch.clamp(wav * 32767.0, -32767.0, 32767.0).to(torch.int16).squeeze(0).unsqueeze(1)


_GLOBAL_DAC_AUTOENCODER = None


def preload_dac_autoencoder(device=None, warmup: bool = False, state_dict: dict | None = None) -> DACAutoencoder:
    """autoencoder.py:10-47 singleton."""
    global _GLOBAL_DAC_AUTOENCODER
    if _GLOBAL_DAC_AUTOENCODER is None:
        _GLOBAL_DAC_AUTOENCODER = DACAutoencoder(state_dict, device=device)
    return _GLOBAL_DAC_AUTOENCODER
