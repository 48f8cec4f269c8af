"""`Zonos` — the reference's model surface (zonos/model.py:43-548) over the MI355X HIP path.

Kept API: `Zonos.from_pretrained / from_local / setup_cache / embed_codes / apply_heads / generate`, attributes
`config, backbone, embeddings, fused_heads, autoencoder, device`.  The hot loop (model.py:467-502) runs as one
hipGraph replay per step inside libzonos_hip.so; the host only mirrors the reference's stop-check cadence
(tensor_ops.py:90-103) and the post-processing (model.py:511-539).
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import threading
import json
from typing import Callable

import torch
import torch.nn as nn

from . import _lib
from .autoencoder import DACAutoencoder
from .backbone import BACKBONES, HipEngine
from .codebook_pattern import apply_delay_pattern, revert_delay_pattern
from .conditioning import ConditioningCache, PrefixConditioner, prepare_conditioning_with_cache
from .config import InferenceParams, ZonosConfig
from .utils import DEFAULT_DEVICE, find_multiple

DEFAULT_BACKBONE_CLS = next(iter(BACKBONES.values()))

_SAMPLING_DEFAULTS = dict(temperature=1.0, top_p=0.0, top_k=0, min_p=0.0, linear=0.0, conf=0.0, quad=0.0,
                          repetition_penalty=3.0, repetition_penalty_window=2)   # zonos/sampling.py:166-178


def _sampling_struct(params: dict, seed: int) -> _lib.zn_sampling:
    unknown = set(params) - set(_SAMPLING_DEFAULTS)
    if unknown:
        raise TypeError(f"sample_from_logits() got unexpected keyword arguments {sorted(unknown)}")
    p = {**_SAMPLING_DEFAULTS, **params}
    return _lib.zn_sampling(temperature=p["temperature"], top_p=p["top_p"], top_k=int(p["top_k"]), min_p=p["min_p"],
                            linear=p["linear"], conf=p["conf"], quad=p["quad"], repetition_penalty=p["repetition_penalty"],
                            repetition_penalty_window=int(p["repetition_penalty_window"]), seed=seed & (2 ** 64 - 1))


class Zonos(nn.Module):
    def __init__(self, config: ZonosConfig, backbone_cls=DEFAULT_BACKBONE_CLS, autoencoder: DACAutoencoder | None = None):
        super().__init__()
        self.config = config
        dim = config.backbone.d_model
        self.eos_token_id = config.eos_token_id
        self.masked_token_id = config.masked_token_id
        self.autoencoder = autoencoder if autoencoder is not None else DACAutoencoder()
        self.backbone = backbone_cls(config.backbone)
        self.prefix_conditioner = PrefixConditioner(config.prefix_conditioner, dim)
        self.prefix_conditioner.attach(lambda: self.engine(1))
        self._conditioning_cache = ConditioningCache(max_size=32)
        vocab_size = find_multiple(1026, 8)  # 1024 codes + EOS + MASK, padded to 1032 (model.py:79-80)
        self.embeddings = nn.ModuleList([nn.Embedding(vocab_size, dim) for _ in range(self.autoencoder.num_codebooks)])
        self.fused_heads = nn.Linear(dim, self.autoencoder.num_codebooks * 1025, bias=False)
        self._engine: HipEngine | None = None
        self._spare: HipEngine | None = None          # second handle over the same weights (a concurrent generate() call)
        self._spare_guard = threading.RLock()
        self._repeats = 0                             # generations repeated on the launches path after a reported hand-off timeout

    # ------------------------------------------------------------------ loading
    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    @classmethod
    def from_pretrained(cls, repo_id: str, revision: str | None = None, device: str = DEFAULT_DEVICE, **kwargs) -> "Zonos":
        """model.py:103-126.  Resolves config.json / model.safetensors through huggingface_hub's local cache only
        semantics are the caller's (no network on the GPU box: pass a local snapshot via from_local)."""
        from huggingface_hub import hf_hub_download
        config_path = hf_hub_download(repo_id=repo_id, filename="config.json", revision=revision)
        model_path = hf_hub_download(repo_id=repo_id, filename="model.safetensors", revision=revision)
        return cls.from_local(config_path, model_path, device, **kwargs)

    @classmethod
    def from_local(cls, config_path: str, model_path: str, device: str = DEFAULT_DEVICE, backbone: str | None = None) -> "Zonos":
        """model.py:128-176: bf16 model, embeddings zero-padded to the 1032-row tables, heads.{i} fused."""
        import safetensors
        config = ZonosConfig.from_dict(json.load(open(config_path)))
        backbone_cls = BACKBONES[backbone] if backbone else DEFAULT_BACKBONE_CLS
        model = cls(config, backbone_cls).to(device, torch.bfloat16)
        sd = model.state_dict()
        expected, seen = set(sd), set()
        with safetensors.safe_open(model_path, framework="pt") as f:
            for k in f.keys():
                t = f.get_tensor(k)
                if k.startswith("embeddings.") and k.endswith(".weight") and k in sd and sd[k].shape[0] != t.shape[0] and sd[k].shape[1] == t.shape[1]:
                    padded = torch.zeros(sd[k].shape, dtype=t.dtype)
                    padded[: t.shape[0]] = t
                    t = padded
                sd[k] = t
                seen.add("fused_heads.weight" if k.startswith("heads.") and k.endswith(".weight") else k)
        # The reference loads strictly (model.py:174: a tensor the model does not know is an error).  A tensor the file lacks
        # would stay at its random initialisation there; here it is an error as well: either way a checkpoint that does not
        # match the configuration (e.g. bias tensors of an attn_cfg this backbone does not implement) never runs silently.
        unexpected, missing = sorted(seen - expected), sorted(expected - seen)
        if unexpected or missing:
            raise _lib.ZonosHipError(f"checkpoint {model_path} does not match the model built from {config_path}: "
                                     f"unexpected tensors {unexpected[:8]}{'...' if len(unexpected) > 8 else ''}, "
                                     f"missing tensors {missing[:8]}{'...' if len(missing) > 8 else ''}")
        model.load_state_dict(sd, strict=True)
        return model

    def _load_from_state_dict(self, state_dict, prefix, *args):
        """model.py:208-223: per-codebook heads.{i}.weight [1025,d] -> fused_heads.weight [9*1025,d]."""
        if f"{prefix}heads.0.weight" in state_dict:
            ws, i = [], 0
            while f"{prefix}heads.{i}.weight" in state_dict:
                ws.append(state_dict.pop(f"{prefix}heads.{i}.weight"))
                i += 1
            state_dict[f"{prefix}fused_heads.weight"] = torch.cat(ws, dim=0)
        super()._load_from_state_dict(state_dict, prefix, *args)

    def _apply(self, fn, *a, **k):
        self._engine = None       # device/dtype moves invalidate bound pointers
        self._spare = None
        if getattr(self.backbone, "_engine", None) is not None:
            self.backbone._engine = None
        return super()._apply(fn, *a, **k)

    def _new_engine(self, batch_size: int) -> HipEngine:
        return HipEngine(self.backbone, [m.weight for m in self.embeddings], self.fused_heads.weight,
                         max_rows=2 * batch_size, double_out_proj=getattr(self.backbone, "ref_double_out_proj", True),
                         n_codebooks=self.autoencoder.num_codebooks, vocab_head=1025, vocab_embed=self.embeddings[0].weight.shape[0],
                         eos_id=self.eos_token_id, mask_id=self.masked_token_id)

    def engine(self, batch_size: int = 1) -> HipEngine:
        # (both handles are created and replaced under one lock: two threads arriving together must not each build an engine,
        # nor drop the spare another thread is about to take)
        with self._spare_guard:
            e = self._engine
            if e is None or e.max_rows < 2 * batch_size or e.device != self.device:
                self._engine = e = self._new_engine(batch_size)
                self._spare = None
            return e

    def handoff_counters(self) -> dict:
        """Hand-off timeouts are never silent: per engine, what the library counted (include/zonos_hip.h zn_get_counters) plus the
        generations this model repeated after a reported timeout."""
        out = dict(repeated_generations=self._repeats)
        for name, e in (("engine", self._engine), ("spare", self._spare)):
            if e is not None:
                out[name] = e.counters()
        return out

    def _acquire_engine(self, batch_size: int) -> HipEngine:
        """The engine a generate() call runs on, with its lock held.  The reference serves two requests per model at a time
        (utilities/app_constants.py:18); one library handle carries one generation, so a second handle over the SAME weight tensors
        (a few MB of workspace) takes the second request instead of making it wait for the first.  The device's persistent-kernel
        tenancy (include/zonos_hip.h) stays with whichever generation began first; the other runs the launches path."""
        def take(e, blocking):
            # (the engine lock is re-entrant for the calls a generation makes on its own thread; `generating` keeps a generate()
            # nested in a callback from re-entering the handle that is mid-generation)
            if not e.lock.acquire(blocking=blocking):
                return False
            if getattr(e, "generating", False):
                e.lock.release()
                return False
            e.generating = True
            return True
        eng = self.engine(batch_size)
        if take(eng, False):
            return eng
        with self._spare_guard:
            sp = self._spare
            if sp is None or sp.max_rows < 2 * batch_size or sp.device != self.device:
                self._spare = sp = self._new_engine(batch_size)
        if take(sp, False):
            return sp
        if not take(eng, True):                             # both busy: queue on the first
            raise _lib.ZonosHipError("generate() re-entered from a callback while both of the model's engines are generating")
        return eng

    # ------------------------------------------------------------------ embed / heads
    @torch.inference_mode()
    def embed_codes(self, codes: torch.Tensor, _eng: HipEngine | None = None) -> torch.Tensor:
        """codec_utils.py:15-37: codes [B, n_q, T] -> bf16 [B, T, d]."""
        B, nq, T = codes.shape
        eng = _eng if _eng is not None else self.engine(1)
        flat = codes.permute(0, 2, 1).reshape(B * T, nq).to(device=self.device, dtype=torch.int32).contiguous()
        out = torch.empty(B * T, self.config.backbone.d_model, dtype=torch.bfloat16, device=self.device)
        eng.call("zn_op_embed", flat.data_ptr(), out.data_ptr(), B * T, eng.stream())
        return out.view(B, T, -1)

    @torch.inference_mode()
    def apply_heads(self, hidden_states: torch.Tensor) -> torch.Tensor:
        """codec_utils.py:40-79: [B, S, d] -> [B, n_q, S, 1025] (bf16)."""
        B, S, d = hidden_states.shape
        nq = self.autoencoder.num_codebooks
        eng = self.engine(1)
        x = hidden_states.reshape(B * S, d).contiguous()
        out = torch.empty(B * S, nq * 1025, dtype=torch.bfloat16, device=x.device)
        eng.call("zn_op_linear", x.data_ptr(), None, None, self.fused_heads.weight.data_ptr(), out.data_ptr(), B * S, nq * 1025, d, eng.stream())
        return out.view(B, S, nq, 1025).transpose(1, 2)

    @torch.inference_mode()
    def prepare_conditioning(self, cond_dict: dict, uncond_dict: dict | None = None, use_cache: bool = False, cfg_scale: float = 1.0) -> torch.Tensor:
        """model.py:237-265 -> conditioning_cache.py:139-193: [2B or B, L_c, d] bf16 ([cond ‖ uncond] when cfg_scale != 1)."""
        return prepare_conditioning_with_cache(self.prefix_conditioner, cond_dict=cond_dict, uncond_dict=uncond_dict, use_cache=use_cache,
                                               cfg_scale=cfg_scale, cache=self._conditioning_cache if use_cache else None)

    def setup_cache(self, batch_size: int, max_seqlen: int, dtype: torch.dtype = torch.bfloat16) -> InferenceParams:
        """model.py:305-338: length rounded to x8, bf16 KV per layer, lengths_per_sample int32 zeros."""
        max_seqlen = find_multiple(max_seqlen, 8)
        kv = self.backbone.allocate_inference_cache(batch_size, max_seqlen, dtype=dtype)
        lengths = torch.zeros(batch_size, dtype=torch.int32, device=self.device)
        return InferenceParams(max_seqlen, batch_size, 0, 0, kv, lengths)

    def can_use_cudagraphs(self) -> bool:
        return self.device.type == "cuda"   # the step is always replayed as a hipGraph

    # ------------------------------------------------------------------ generate
    @torch.inference_mode()
    def generate(self, prefix_conditioning: torch.Tensor, audio_prefix_codes: torch.Tensor = None, max_new_tokens: int = 86 * 30,
                 cfg_scale: float = 2.0, batch_size: int = 1, sampling_params: dict = dict(min_p=0.1),
                 disable_torch_compile: bool = False, callback: Callable[[torch.Tensor, int, int], bool] | None = None,
                 seed: int | None = None, _trace: dict | None = None):
        """zonos/model.py:354-548.  prefix_conditioning bf16 [2B, L_c, d] = [cond ‖ uncond]; returns int64
        [B, 9, T_out] with values in [0, 1023].  Batch semantics for B > 1 (the reference crashes there,
        SURVEY.md §0.6): B independent utterances, rows [cond_0..cond_{B-1}, uncond_0..uncond_{B-1}].
        `seed` seeds the device Gumbel-max stream (default: drawn from torch's generator)."""
        assert cfg_scale != 1, "TODO: add support for cfg_scale=1"
        dev = self.device
        if dev.type != "cuda":
            raise _lib.ZonosHipError("zonos_amd runs on MI355X only: move the model to a cuda device (no CPU fallback)")
        B = batch_size
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        # an engine's handle holds this generation's state: a third concurrent generate() call on one model queues here
        eng = self._acquire_engine(B)
        try:
            return self._generate_on(eng, dev, prefix_conditioning, audio_prefix_codes, max_new_tokens, cfg_scale, B, sampling_params, callback, seed, _trace)
        finally:
            eng.generating = False
            eng.lock.release()

    def _generate_on(self, eng, dev, prefix_conditioning, audio_prefix_codes, max_new_tokens, cfg_scale, B, sampling_params, callback, seed, _trace):
        with torch.cuda.device(dev):
            run = lambda: self._generate_locked(eng, prefix_conditioning, audio_prefix_codes, max_new_tokens, cfg_scale, B, sampling_params, callback,
                                                seed, _trace)
            return self._with_timeout_policy(run, caller_saw_frames=callback is not None or _trace is not None)

    def _with_timeout_policy(self, run, caller_saw_frames: bool):
        """A bounded in-kernel hand-off wait gave up (include/zonos_hip.h, INTEGRATION.md "Single tenant per device"): the results are void
        and the library has demoted this handle to the launches path, which has no in-launch hand-offs.  Nothing has been returned yet, so
        the generation is run once more there - unless the caller has already seen frames of it, or ZONOS_HIP_NO_TIMEOUT_RETRY=1 asks for
        the error itself (tests/conftest.py sets it for every test, so that no timeout can hide behind a repeated generation; bench.py
        counts repeats and fails on one).  Every repeat is counted (`handoff_counters`) and announced on stderr."""
        try:
            return run()
        except _lib.ZonosHipError as e:
            if "hand-off wait" not in str(e) or caller_saw_frames or os.environ.get("ZONOS_HIP_NO_TIMEOUT_RETRY") == "1":
                raise
            self._repeats += 1
            print(f"[zonos_amd] {e}\n[zonos_amd] repeating the generation on the launches path", file=sys.stderr, flush=True)
            return run()

    def _generate_locked(self, eng, prefix_conditioning, audio_prefix_codes, max_new_tokens, cfg_scale, batch_size, sampling_params, callback, seed,
                         _trace):
        dev = self.device
        B, nq = batch_size, self.config.codebook_dimension
        if prefix_conditioning.shape[0] != 2 * B:
            raise ValueError(f"prefix_conditioning must have 2*batch_size={2 * B} rows, got {prefix_conditioning.shape[0]}")
        P = 0 if audio_prefix_codes is None else audio_prefix_codes.shape[2]
        L_c = prefix_conditioning.shape[1]
        audio_len = P + max_new_tokens
        seq_len = L_c + audio_len + nq
        ip = self.setup_cache(batch_size=2 * B, max_seqlen=seq_len)
        codes = torch.full((B, nq, audio_len), -1, dtype=torch.int32, device=dev)
        if audio_prefix_codes is not None:
            codes[..., :P] = audio_prefix_codes.to(device=dev, dtype=torch.int32)
        delayed = apply_delay_pattern(codes, self.masked_token_id).contiguous()       # [B, nq, audio_len + nq]
        t_total = delayed.shape[2]
        offset = P + 1
        st = eng.stream()
        sp = _sampling_struct(sampling_params, seed)
        kv_ptrs = (C.c_void_p * self.config.backbone.n_layer)(*[ip.key_value_memory_dict[i][0].data_ptr() for i in range(self.config.backbone.n_layer)])
        eng.call("zn_gen_begin", B, kv_ptrs, ip.max_seqlen, ip.lengths_per_sample.data_ptr(), delayed.data_ptr(), t_total, offset,
                 max_new_tokens, float(cfg_scale), C.byref(sp), st)
        try:
            offset = self._decode_loop(eng, ip, delayed, prefix_conditioning, offset, t_total, B, nq, callback, _trace, st)
        finally:
            # the device's persistent-kernel tenancy goes back once this generation's kernels have drained (include/zonos_hip.h)
            torch.cuda.current_stream(dev).synchronize()
            eng.call("zn_gen_end")
        out = revert_delay_pattern(delayed.to(torch.int64)).cpu()     # one device->host copy (model.py:511)
        valid_length = offset - nq
        window = min(50, valid_length // 4)
        for pos in range(max(0, valid_length - window), valid_length):   # model.py:516-528
            if int((out[:, :, pos] == self.eos_token_id).sum()) >= nq // 2:
                valid_length = pos
                break
        out = torch.where(out > 1024, 512, out)
        out = torch.where(out == 1024, 0, out)
        return torch.clamp(out[..., :valid_length], 0, 1023).to(dev)

    def _decode_loop(self, eng, ip, delayed, prefix_conditioning, offset, t_total, B, nq, callback, _trace, st) -> int:
        """Prefill, first frame and the hot loop (model.py:421-509); returns the final column offset."""
        dev = self.device
        # prefill (generation_utils.py:236-244): [cond ‖ uncond] conditioning + embed(delayed[..., :P+1]) for both halves
        emb = self.embed_codes(delayed[..., :offset], _eng=eng)
        hidden = torch.cat([prefix_conditioning.to(device=dev, dtype=torch.bfloat16), emb.repeat(2, 1, 1)], dim=1).contiguous()
        S = hidden.shape[1]
        eng.call("zn_prefill", hidden.data_ptr(), S, st)
        eng.call("zn_sample_first", st)
        ip.seqlen_offset += S
        if _trace is not None:
            _trace.setdefault("logits", []).append(self._step_logits(eng, B, nq))
            if _trace.get("after_step") is not None:
                _trace["after_step"](-1, delayed, offset)
        # hot loop with the reference's stop-check cadence (tensor_ops.py:84-105)
        max_steps = t_total - offset
        frame = delayed[..., offset:offset + 1]
        pending, done = 0, ctypes_int()
        cpu_step_counter = 0
        # Without a callback the stop flag of check k is read after the steps up to check k + 1 have been enqueued (the device ->
        # host round trip leaves the critical path); a stop seen late rolls `offset` back to the check that saw it, and the
        # over-run steps have only written columns beyond that cut.
        deferred = callback is None and _trace is None
        begun_at = None                                            # offset at the check whose read-back is in flight
        for step_idx in range(max_steps):
            offset += 1
            cpu_step_counter += 1
            if offset >= t_total:
                break
            pending += 1
            ip.seqlen_offset += 1
            check = (step_idx % 16 == 15) or (step_idx % 8 == 7 and max(0, B * 10 - cpu_step_counter) < 5)
            if check or callback is not None or _trace is not None:
                eng.call("zn_decode_steps", pending, st)
                pending = 0
                if _trace is not None:
                    _trace["logits"].append(self._step_logits(eng, B, nq))
                    hook = _trace.get("after_step")
                    if hook is not None:
                        hook(step_idx, delayed, offset)
                        eng.call("zn_codes_changed")        # the hook may rewrite the column the next step embeds
            if check and deferred:
                if begun_at is not None:
                    eng.call("zn_all_stopped_end", C.byref(done))
                    if done.value:
                        offset, begun_at = begun_at, None
                        break
                eng.call("zn_all_stopped_begin", st)
                begun_at = offset
            elif check:
                eng.call("zn_all_stopped", C.byref(done), st)
                if done.value:
                    break
            if callback is not None and not callback(frame, step_idx + 1, max_steps):
                break
        if pending:
            eng.call("zn_decode_steps", pending, st)
        if begun_at is not None:
            eng.call("zn_all_stopped_end", C.byref(done))
            if done.value:
                offset = begun_at
        eng.call("zn_all_stopped", C.byref(done), st)      # also surfaces a timed-out in-kernel hand-off of the last steps
        return offset

    def _step_logits(self, eng: HipEngine, B: int, nq: int) -> torch.Tensor:
        buf = torch.empty(B, nq, 1025, dtype=torch.float32, device=self.device)
        eng.call("zn_get_step_outputs", buf.data_ptr(), None, eng.stream())
        return buf


def ctypes_int():
    return C.c_int32(0)
