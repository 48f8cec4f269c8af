"""Prefix conditioning (zonos/conditioning.py:14-109,338-522,545-644; zonos/utilities/conditioning_cache.py:139-193).

Same surface as the reference: `make_cond_dict`, `PrefixConditioner` (with `.conditioners[*].name`, `.required_keys`),
the four conditioner types, `supported_language_codes`, and `prepare_conditioning_with_cache`.  Parameters live in
`nn.Module`s under the reference's names (`prefix_conditioner.conditioners.{i}.phoneme_embedder.weight`, `.project.*`,
`.weight` (Fourier buffer), `.int_embedder.weight`, `.uncond_vector`, `prefix_conditioner.norm.*`) so checkpoints load
unchanged; the arithmetic runs in libzonos_hip.so (gathers, Fourier features, Linear+bias, SiLU, LayerNorm) on the device.

Text -> phonemes is an external dependency in the reference too (phonemizer + espeak-ng).  It is imported lazily; callers
that already have phoneme strings or token ids can pass `("phonemes", [...])` / `("ids", tensor)` as the `espeak` entry.
"""
from __future__ import annotations

import hashlib
import json
import os
from typing import Any, Iterable, Literal

import torch
import torch.nn as nn

from . import _lib
from .config import PrefixConditionerConfig
from .utils import DEFAULT_DEVICE

_VOCAB = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "phoneme_vocab.json"), encoding="utf-8"))
PAD_ID, UNK_ID, BOS_ID, EOS_ID = 0, 1, 2, 3
SPECIAL_TOKEN_IDS = [PAD_ID, UNK_ID, BOS_ID, EOS_ID]
symbols = list(_VOCAB["symbols"])
_symbol_to_id = {s: i for i, s in enumerate(symbols, start=len(SPECIAL_TOKEN_IDS))}
supported_language_codes = list(_VOCAB["language_codes"])


def tokenize_phonemes(phonemes: list[str]) -> tuple[torch.Tensor, list[int]]:
    """conditioning.py:243-248: BOS + symbol ids (unknown -> UNK) + EOS, left-padded with PAD to the longest."""
    ids = [[BOS_ID, *[_symbol_to_id.get(ch, UNK_ID) for ch in p], EOS_ID] for p in phonemes]
    lengths = [len(i) for i in ids]
    longest = max(lengths)
    return torch.tensor([[PAD_ID] * (longest - len(i)) + i for i in ids]), lengths


_backends: dict = {}


def get_backend(language: str):
    """conditioning.py:291-304: one EspeakBackend per language, kept (building one starts espeak and loads the voice)."""
    be = _backends.get(language)
    if be is None:
        try:
            from phonemizer.backend import EspeakBackend
        except ImportError as e:      # pragma: no cover - dependency absent offline
            raise _lib.ZonosHipError("text -> phoneme conversion needs `phonemizer` + espeak-ng (as the reference does); "
                                     "pass espeak=('phonemes', [...]) or ('ids', tensor) instead") from e
        be = _backends[language] = EspeakBackend(language, preserve_punctuation=True, with_stress=True,   # pragma: no cover
                                                 punctuation_marks=_VOCAB["punctuation"])
    return be


def phonemize(texts: list[str], languages: list[str]) -> list[str]:
    """conditioning.py:307-335: clean() (numbers spelled out; Japanese normalised) and then espeak through `phonemizer` — the same
    third-party stack as the reference."""
    from .text_cleaning import clean
    texts = clean(texts, languages)
    return [get_backend(lang).phonemize([text], strip=True)[0] for text, lang in zip(texts, languages)]   # pragma: no cover


def _engine_call(mod: nn.Module, name: str, *args):
    eng = mod._zn_engine() if hasattr(mod, "_zn_engine") else None
    if eng is None:
        raise _lib.ZonosHipError("conditioner is not attached to a Zonos model (model.prefix_conditioner)")
    eng.call(name, *args, eng.stream())


class Conditioner(nn.Module):
    """conditioning.py:14-109."""

    def __init__(self, output_dim: int, name: str, cond_dim: int | None = None, projection: Literal["none", "linear", "mlp"] = "none",
                 uncond_type: Literal["learned", "none"] = "none", **kwargs):
        super().__init__()
        self.name, self.output_dim = name, output_dim
        self.cond_dim = cond_dim = cond_dim or output_dim
        if projection == "linear":
            self.project = nn.Linear(cond_dim, output_dim)
        elif projection == "mlp":
            self.project = nn.Sequential(nn.Linear(cond_dim, output_dim), nn.SiLU(), nn.Linear(output_dim, output_dim))
        else:
            self.project = nn.Identity()
        self.uncond_vector = nn.Parameter(torch.zeros(output_dim)) if uncond_type == "learned" else None
        self._zn_engine = lambda: None

    def apply_cond(self, *inputs: Any) -> torch.Tensor:
        raise NotImplementedError()

    # ---- device arithmetic
    def _linear(self, lin: nn.Linear, x: torch.Tensor) -> torch.Tensor:
        lead, K = x.shape[:-1], x.shape[-1]          # forward() unpacks [1, 1, n] inputs into 2-D tensors (conditioning.py:107)
        x2 = x.reshape(-1, K).to(device=lin.weight.device, dtype=torch.bfloat16).contiguous()
        out = torch.empty(x2.shape[0], lin.out_features, dtype=torch.bfloat16, device=lin.weight.device)
        _engine_call(self, "zn_op_linear_bias", x2.data_ptr(), lin.weight.data_ptr(), _lib.ptr(lin.bias), out.data_ptr(), x2.shape[0], lin.out_features, K)
        return out.view(*lead, -1)

    def _project(self, cond: torch.Tensor) -> torch.Tensor:
        if isinstance(self.project, nn.Linear):
            return self._linear(self.project, cond)
        if isinstance(self.project, nn.Sequential):
            h = self._linear(self.project[0], cond)
            a = torch.empty_like(h)
            _engine_call(self, "zn_op_silu", h.data_ptr(), a.data_ptr(), h.numel())
            return self._linear(self.project[2], a)
        return cond

    def _gather(self, table: torch.Tensor, ids: torch.Tensor, id_offset: int = 0) -> torch.Tensor:
        lead = ids.shape
        i32 = ids.reshape(-1).to(device=table.device, dtype=torch.int32).contiguous()
        out = torch.empty(i32.numel(), table.shape[1], dtype=table.dtype, device=table.device)
        _engine_call(self, "zn_op_gather_rows", table.data_ptr(), i32.data_ptr(), out.data_ptr(), i32.numel(), table.shape[1], table.shape[0], id_offset)
        return out.view(*lead, -1)

    def forward(self, inputs: tuple[Any, ...] | None) -> torch.Tensor:
        if inputs is None:
            assert self.uncond_vector is not None
            return self.uncond_vector.data.view(1, 1, -1)
        return self._project(self.apply_cond(*inputs))


class EspeakPhonemeConditioner(Conditioner):
    """conditioning.py:338-382."""

    def __init__(self, output_dim: int, **kwargs):
        super().__init__(output_dim, **kwargs)
        self.phoneme_embedder = nn.Embedding(len(SPECIAL_TOKEN_IDS) + len(symbols), output_dim)

    def apply_cond(self, texts, languages) -> torch.Tensor:
        if isinstance(texts, str) and texts == "ids":
            ids = languages
        elif isinstance(texts, str) and texts == "phonemes":
            ids, _ = tokenize_phonemes(list(languages))
        else:
            ids, _ = tokenize_phonemes(phonemize(texts, languages))
        return self._gather(self.phoneme_embedder.weight, ids)


class FourierConditioner(Conditioner):
    """conditioning.py:388-441."""

    def __init__(self, output_dim: int, input_dim: int = 1, std: float = 1.0, min_val: float = 0.0, max_val: float = 1.0, **kwargs):
        assert output_dim % 2 == 0
        super().__init__(output_dim, **kwargs)
        self.register_buffer("weight", torch.randn([output_dim // 2, input_dim]) * std)
        self.input_dim, self.min_val, self.max_val = input_dim, min_val, max_val

    def apply_cond(self, x: torch.Tensor) -> torch.Tensor:
        assert x.shape[-1] == self.input_dim
        lead = x.shape[:-1]
        xf = x.reshape(-1, self.input_dim).to(device=self.weight.device, dtype=torch.float32).contiguous()
        half = self.weight.shape[0]
        out = torch.empty(xf.shape[0], 2 * half, dtype=torch.bfloat16, device=self.weight.device)
        w = (self.weight if self.weight.dtype == torch.bfloat16 else self.weight.to(torch.bfloat16)).contiguous()
        _engine_call(self, "zn_op_fourier", xf.data_ptr(), w.data_ptr(), out.data_ptr(), xf.shape[0], self.input_dim, half,
                     float(self.min_val), float(self.max_val))
        return out.view(*lead, -1)


class IntegerConditioner(Conditioner):
    """conditioning.py:444-468."""

    def __init__(self, output_dim: int, min_val: int = 0, max_val: int = 512, **kwargs):
        super().__init__(output_dim, **kwargs)
        self.min_val, self.max_val = min_val, max_val
        self.int_embedder = nn.Embedding(max_val - min_val + 1, output_dim)

    def apply_cond(self, x: torch.Tensor) -> torch.Tensor:
        assert x.shape[-1] == 1
        return self._gather(self.int_embedder.weight, x.squeeze(-1), id_offset=self.min_val)


class PassthroughConditioner(Conditioner):
    """conditioning.py:471-477."""

    def apply_cond(self, x: torch.Tensor) -> torch.Tensor:
        assert x.shape[-1] == self.cond_dim
        return x


_cond_cls_map = {"PassthroughConditioner": PassthroughConditioner, "EspeakPhonemeConditioner": EspeakPhonemeConditioner,
                 "FourierConditioner": FourierConditioner, "IntegerConditioner": IntegerConditioner}


def build_conditioners(conditioners: list[dict], output_dim: int) -> list[Conditioner]:
    return [_cond_cls_map[c["type"]](output_dim, **c) for c in conditioners]


class PrefixConditioner(Conditioner):
    """conditioning.py:506-522: per-conditioner embeddings concatenated along the sequence, projected, LayerNorm."""

    def __init__(self, config: PrefixConditionerConfig, output_dim: int):
        super().__init__(output_dim, "prefix", projection=config.projection)
        self.conditioners = nn.ModuleList(build_conditioners(config.conditioners, output_dim))
        self.norm = nn.LayerNorm(output_dim)
        self.required_keys = {c.name for c in self.conditioners if c.uncond_vector is None}

    def attach(self, engine_fn) -> None:
        self._zn_engine = engine_fn
        for c in self.conditioners:
            c._zn_engine = engine_fn

    def forward(self, cond_dict: dict) -> torch.Tensor:
        if not set(cond_dict).issuperset(self.required_keys):
            raise ValueError(f"Missing required keys: {self.required_keys - set(cond_dict)}")
        dev = self.norm.weight.device
        conds = [c(cond_dict.get(c.name)).to(device=dev, dtype=torch.bfloat16) for c in self.conditioners]
        max_bsz = max(map(len, conds))
        assert all(c.shape[0] in (max_bsz, 1) for c in conds)
        x = self._project(torch.cat([c.expand(max_bsz, -1, -1) for c in conds], dim=-2).contiguous()).contiguous()
        B, S, d = x.shape
        out = torch.empty_like(x)
        _engine_call(self, "zn_op_layernorm", x.data_ptr(), self.norm.weight.data_ptr(), self.norm.bias.data_ptr(), out.data_ptr(), B * S, d)
        return out


def _get_language_id(language: str) -> int:
    lid = {lang: i for i, lang in enumerate(supported_language_codes)}.get(language.lower(), -1)
    assert lid != -1, f"Unsupported language: {language}. Please pick from {supported_language_codes}"
    return lid


def make_cond_dict(text: str = "It would be nice to have time for testing, indeed.", language: str = "en-us", speaker: torch.Tensor = None,
                   emotion: list[float] = [0.3077, 0.0256, 0.0256, 0.0256, 0.0256, 0.0256, 0.2564, 0.3077], fmax: float = 22050.0,
                   pitch_std: float = 20.0, speaking_rate: float = 15.0, vqscore_8: list[float] = [0.78] * 8, ctc_loss: float = 0.0,
                   dnsmos_ovrl: float = 4.0, speaker_noised: bool = False, unconditional_keys: Iterable[str] = {"vqscore_8", "dnsmos_ovrl"},
                   device: torch.device | str = DEFAULT_DEVICE) -> dict:
    """conditioning.py:545-644: scalars/lists -> tensors [1, 1, n]; emotion normalised to sum 1; language -> id."""
    cond = {"espeak": ([text], [language]), "speaker": speaker, "emotion": emotion, "fmax": fmax, "pitch_std": pitch_std,
            "speaking_rate": speaking_rate, "language_id": _get_language_id(language), "vqscore_8": vqscore_8, "ctc_loss": ctc_loss,
            "dnsmos_ovrl": dnsmos_ovrl, "speaker_noised": int(speaker_noised)}
    for k in unconditional_keys:
        cond.pop(k, None)
    for k, v in list(cond.items()):
        if isinstance(v, (float, int, list)):
            v = torch.tensor(v)
        if isinstance(v, torch.Tensor):
            cond[k] = v.view(1, 1, -1).to(device)
        if k == "emotion":
            cond[k] /= cond[k].sum(dim=-1)
    return cond


# ------------------------------------------------------------------ conditioning cache (conditioning_cache.py:13-193)
def create_conditioning_cache_key(cond_dict: dict, uncond_dict: dict | None) -> str:
    h = hashlib.sha512()
    for d in (cond_dict, uncond_dict or {}):
        for k in sorted(d):
            v = d[k]
            h.update(k.encode())
            if isinstance(v, torch.Tensor):
                h.update(str(tuple(v.shape)).encode() + str(v.dtype).encode() + v.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes())
            else:
                h.update(repr(v).encode())
        h.update(b"|")
    return h.hexdigest()


class ConditioningCache:
    """LRU (insertion-ordered dict) of conditioning tensors, capacity 32 (conditioning_cache.py:56-136)."""

    def __init__(self, max_size: int = 32):
        self.max_size, self._cache = max_size, {}

    def get(self, key: str):
        if key in self._cache:
            self._cache[key] = self._cache.pop(key)
            return self._cache[key]
        return None

    def put(self, key: str, tensor: torch.Tensor) -> None:
        if key in self._cache:
            self._cache.pop(key)
        elif len(self._cache) >= self.max_size:
            del self._cache[next(iter(self._cache))]
        self._cache[key] = tensor

    def clear(self) -> None:
        self._cache.clear()

    def size(self) -> int:
        return len(self._cache)


def prepare_conditioning_with_cache(prefix_conditioner, cond_dict: dict, uncond_dict: dict | None = None, use_cache: bool = False,
                                    cfg_scale: float = 1.0, cache: ConditioningCache | None = None) -> torch.Tensor:
    """conditioning_cache.py:139-193: cfg_scale == 1 -> cond only; else cat([cond, uncond]) with uncond = required keys."""
    key = create_conditioning_cache_key(cond_dict, uncond_dict) if (use_cache and cache is not None) else None
    if key is not None:
        hit = cache.get(key)
        if hit is not None:
            return hit
    if cfg_scale == 1.0:
        out = prefix_conditioner(cond_dict)
    else:
        if uncond_dict is None:
            uncond_dict = {k: cond_dict[k] for k in prefix_conditioner.required_keys}
        out = torch.cat([prefix_conditioner(cond_dict), prefix_conditioner(uncond_dict)])
    if key is not None:
        cache.put(key, out)
    return out
