"""Speaker embedding — zonos/speaker_cloning.py surface (`SpeakerEmbedding`, `SpeakerEmbeddingLDA`, `logFbankCal`) over
the HIP network in libzonos_hip.so (zn_spk_create / zn_spk_embed: ResNet293 of SimAM blocks -> ASP -> bottleneck -> LDA,
fp32).  SURVEY.md 8f row 4: runs once per speaker, outside the decode loop.

The reference downloads its two checkpoints (speaker_cloning.py:848-857); offline they come from local files or state
dicts.  The feature front end restates torchaudio's published `MelSpectrogram` / `Resample` with torch ops (torch.stft):
torchaudio is not installed in this environment, so the front end is unpinned; the network itself is checked against
the reference's own `ResNet293_based` class (tests/golden/speaker.npz).
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib
from .autoencoder import sinc_resample


def melscale_fbanks(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> torch.Tensor:
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale="htk") as published: triangular filters [n_freqs, n_mels]."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    f_pts = 700.0 * (10 ** (torch.linspace(m_min, m_max, n_mels + 2) / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


class logFbankCal:
    """speaker_cloning.py:39-87: MelSpectrogram(sample_rate, n_fft, win_length, hop_length, n_mels) -> log1p -> subtract the
    mean over time.  MelSpectrogram defaults restated: periodic Hann window, centred reflect-padded frames, power 2,
    HTK mel scale without normalisation, f_max = sample_rate / 2."""

    def __init__(self, sample_rate: int = 16_000, n_fft: int = 512, win_length: float = 0.025, hop_length: float = 0.01, n_mels: int = 80):
        self.sample_rate, self.n_fft, self.n_mels = sample_rate, n_fft, n_mels
        self.win_length, self.hop_length = int(win_length * sample_rate), int(hop_length * sample_rate)
        self._fb = melscale_fbanks(n_fft // 2 + 1, 0.0, sample_rate / 2.0, n_mels, sample_rate)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        window = torch.hann_window(self.win_length, periodic=True, device=x.device, dtype=torch.float32)
        spec = torch.stft(x.to(torch.float32), self.n_fft, hop_length=self.hop_length, win_length=self.win_length, window=window, center=True,
                          pad_mode="reflect", normalized=False, onesided=True, return_complex=True).abs().pow(2.0)
        mel = torch.matmul(spec.transpose(-1, -2), self._fb.to(x.device)).transpose(-1, -2)     # [B, n_mels, T]
        out = torch.log1p(mel)
        return out - out.mean(dim=2, keepdim=True)


class SpeakerEmbedding:
    """speaker_cloning.py:692-798.  `state_dict` = the reference checkpoint's keys (ResNet293_based: front.*, pooling.*,
    bottleneck.*); `lda` = the LDA file's {"weight", "bias"} (optional)."""

    def __init__(self, state_dict: dict | str, device="cuda", lda: dict | None = None):
        self.device = torch.device(device)
        if isinstance(state_dict, str):
            state_dict = torch.load(state_dict, weights_only=True, map_location="cpu")      # speaker_cloning.py:735-740
        if self.device.type != "cuda":
            raise _lib.ZonosHipError("zonos_amd runs on MI355X only (no CPU fallback)")
        keep = {k: v.detach().to(device=self.device, dtype=torch.float32).contiguous() for k, v in state_dict.items()
                if k.startswith(("front.", "pooling.", "bottleneck.")) and not k.endswith("num_batches_tracked")}
        if lda is not None:
            keep["lda.weight"] = lda["weight"].detach().to(device=self.device, dtype=torch.float32).contiguous()
            keep["lda.bias"] = lda["bias"].detach().to(device=self.device, dtype=torch.float32).contiguous()
        self._weights = keep
        self.emb_dim = keep["bottleneck.bias"].numel()
        self.lda_dim = keep["lda.bias"].numel() if lda is not None else 0
        self.featCal = logFbankCal()
        self.dtype = torch.float32
        lib = _lib.load()
        names = sorted(keep)
        arr = (_lib.zn_dac_tensor * len(names))()
        for i, k in enumerate(names):
            arr[i].name, arr[i].data_dev, arr[i].numel = k.encode(), keep[k].data_ptr(), keep[k].numel()
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = lib.zn_spk_create(arr, len(names), C.byref(h))
        if rc != 0:
            raise _lib.ZonosHipError(f"zn_spk_create failed ({rc}): {(lib.zn_spk_last_error(None) or b'').decode()}")
        self._h = h

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.load().zn_spk_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def prepare_input(self, wav: torch.Tensor, sample_rate: int) -> torch.Tensor:
        """speaker_cloning.py:756-782: mono, 16 kHz, shape [1, samples]."""
        assert wav.ndim < 3
        if wav.ndim == 2:
            wav = wav.mean(0, keepdim=True)
        return sinc_resample(wav, sample_rate, 16_000)

    @torch.inference_mode()
    def embed_features(self, feats: torch.Tensor, with_lda: bool = False):
        """feats [B, 80, T] (what logFbankCal returns) -> emb [B, 256] (and the LDA projection)."""
        B, _, T = feats.shape
        x = feats.to(device=self.device, dtype=torch.float32).contiguous()
        emb = torch.empty(B, self.emb_dim, dtype=torch.float32, device=self.device)
        lda = torch.empty(B, self.lda_dim, dtype=torch.float32, device=self.device) if with_lda else None
        lib = _lib.load()
        with torch.cuda.device(self.device):
            rc = lib.zn_spk_embed(self._h, x.data_ptr(), B, T, emb.data_ptr(), lda.data_ptr() if with_lda else None, _lib.stream_ptr())
        if rc != 0:
            raise _lib.ZonosHipError(f"zn_spk_embed failed ({rc}): {(lib.zn_spk_last_error(self._h) or b'').decode()}")
        return (emb, lda) if with_lda else emb

    def forward(self, wav: torch.Tensor, sample_rate: int) -> torch.Tensor:
        """speaker_cloning.py:784-797."""
        wav = self.prepare_input(wav.to(self.device), sample_rate).to(torch.float32)
        return self.embed_features(self.featCal(wav))

    __call__ = forward


class SpeakerEmbeddingLDA:
    """speaker_cloning.py:800-883: forward(wav, sr) -> (emb [1, 256], lda_emb [1, 128])."""

    def __init__(self, spk_state_dict: dict | str, lda_state_dict: dict | str, device="cuda"):
        if isinstance(lda_state_dict, str):
            lda_state_dict = torch.load(lda_state_dict, weights_only=True, map_location="cpu")   # speaker_cloning.py:861
        self.device = torch.device(device)
        self.model = SpeakerEmbedding(spk_state_dict, device, lda=lda_state_dict)

    def forward(self, wav: torch.Tensor, sample_rate: int):
        wav = self.model.prepare_input(wav.to(self.device), sample_rate).to(torch.float32)
        return self.model.embed_features(self.model.featCal(wav), with_lda=True)

    __call__ = forward
