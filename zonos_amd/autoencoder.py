"""`DACAutoencoder` — zonos/autoencoder.py:50-170 surface over the HIP DAC codec.

`decode(codes) -> float32 [B, 1, 512*T]`, `decode_to_int16` and `encode(wav) -> int64 [B, 9, T/512]` run in
libzonos_hip.so (fp32 arithmetic: the reference's CPU path disables autocast, autoencoder.py:139).  Weights use the
transformers `DacModel` state-dict names (the reference loads `descript/dac_44khz`, autoencoder.py:74 — a network
fetch, so here they come from a local safetensors file or a state dict).  `preprocess` pads on the left to a multiple
of 512 like the reference; its resampler restates torchaudio's `sinc_interp_hann` (torchaudio is not installed here:
that step is unpinned, identity at 44.1 kHz input).
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np
import threading

import torch

from . import _lib

# transformers DacConfig defaults = descript/dac_44khz (configuration_dac.py:55-70)
DAC_44KHZ = dict(n_codebooks=9, codebook_size=1024, codebook_dim=8, hidden_size=1024, decoder_hidden_size=1536,
                 encoder_hidden_size=64, upsampling_ratios=(8, 8, 4, 2), sampling_rate=44100)


def sinc_resample(wav: torch.Tensor, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99) -> torch.Tensor:
    """torchaudio.functional.resample(..., resampling_method="sinc_interp_hann") as published (the call the reference
    makes at zonos/autoencoder.py:98 with the defaults): windowed-sinc polyphase filter applied as a strided conv1d.
    torchaudio is not installed in this environment, so this restatement is unpinned; equal rates return the input."""
    orig_freq, new_freq = int(orig_freq), int(new_freq)
    if orig_freq == new_freq:
        return wav
    g = math.gcd(orig_freq, new_freq)
    o, n = orig_freq // g, new_freq // g
    base = min(o, n) * rolloff
    width = math.ceil(lowpass_filter_width * o / base)
    idx = torch.arange(-width, width + o, dtype=torch.float64, device=wav.device)[None, None] / o
    t = torch.arange(0, -n, -1, dtype=torch.float64, device=wav.device)[:, None, None] / n + idx
    t = (t * base).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernels = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base / o)
    kernels = kernels.to(torch.float32)
    shape = wav.shape
    x = wav.reshape(-1, shape[-1]).to(torch.float32)
    length = x.shape[-1]
    x = torch.nn.functional.pad(x, (width, width + o))
    y = torch.nn.functional.conv1d(x[:, None], kernels, stride=o).transpose(1, 2).reshape(x.shape[0], -1)
    y = y[..., : math.ceil(n * length / o)]
    return y.reshape(shape[:-1] + y.shape[-1:])


class DACAutoencoder:
    def __init__(self, state_dict: dict | None = None, config: dict | None = None, device=None):
        self.cfg = dict(DAC_44KHZ, **(config or {}))
        self.codebook_size = self.cfg["codebook_size"]
        self.num_codebooks = self.cfg["n_codebooks"]
        self.sampling_rate = self.cfg["sampling_rate"]
        self.hop = int(np.prod(self.cfg["upsampling_ratios"]))
        self._weights: dict | None = None
        self._h = None
        self._lock = threading.Lock()      # one handle = one workspace: concurrent encode/decode calls queue
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
                if k.startswith(("quantizer.quantizers.", "decoder.", "encoder."))}
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
                                hidden_size=c["hidden_size"], decoder_hidden_size=c["decoder_hidden_size"], n_ratios=len(c["upsampling_ratios"]),
                                encoder_hidden_size=c["encoder_hidden_size"] if "encoder.conv1.weight" in self._weights else 0)
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
        """autoencoder.py:80-101: resample to 44.1 kHz, zero-pad on the LEFT to a multiple of 512 samples."""
        wav = sinc_resample(wav, sr, self.sampling_rate)
        left_pad = math.ceil(wav.shape[-1] / self.hop) * self.hop - wav.shape[-1]
        return torch.nn.functional.pad(wav, (left_pad, 0), value=0)

    @torch.inference_mode()
    def encode(self, wav: torch.Tensor) -> torch.Tensor:
        """autoencoder.py:103-117: preprocessed audio [B, 1, T] (or [B, T]) -> int64 codes [B, 9, T / 512]."""
        h = self._handle()
        if "encoder.conv1.weight" not in self._weights:
            raise _lib.ZonosHipError("this DACAutoencoder was built from decoder-only weights: encode() needs encoder.* and in_proj tensors")
        if wav.dim() == 3:
            if wav.shape[1] != 1:
                raise ValueError(f"expected mono audio [B, 1, T], got {tuple(wav.shape)}")
            wav = wav[:, 0]
        B, T = wav.shape
        if T < self.hop or T % self.hop:
            raise ValueError(f"audio length {T} is not a positive multiple of {self.hop}: call preprocess() first (autoencoder.py:99-100)")
        x = wav.to(device=self.device, dtype=torch.float32).contiguous()
        codes = torch.empty(B, self.num_codebooks, T // self.hop, dtype=torch.int32, device=self.device)
        with self._lock, torch.cuda.device(self.device):
            _lib.check_dac(_lib.load().zn_dac_encode(h, x.data_ptr(), B, T, codes.data_ptr(), _lib.stream_ptr()), h, "zn_dac_encode")
        return codes.to(torch.int64)

    @torch.inference_mode()
    def decode(self, codes: torch.Tensor) -> torch.Tensor:
        """autoencoder.py:119-140: codes [B, 9, T] (ints in [0, 1023]) -> float32 [B, 1, 512*T]."""
        h = self._handle()
        B, nq, T = codes.shape
        if nq != self.num_codebooks:
            raise ValueError(f"expected {self.num_codebooks} codebooks, got {nq}")
        c32 = codes.to(device=self.device, dtype=torch.int32).contiguous()
        wav = torch.empty(B, 1, self.hop * T, dtype=torch.float32, device=self.device)
        with self._lock, torch.cuda.device(self.device):
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
