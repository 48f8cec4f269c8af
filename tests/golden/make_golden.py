"""Generate the golden fixtures in this directory by running the REFERENCE itself.

Runs only in the build container (needs /root/reference, which never travels).  It imports the
reference's hot-path modules with stub modules for absent third-party packages (recipe: SURVEY.md §8c),
builds `zonos.model.Zonos` around `TorchZonosBackbone` with the build's synthetic weights
(zonos_amd/synth.py, regenerated from the seed on both sides, never committed) and records outputs of the
real `Zonos.generate()`, of `zonos.sampling`, `zonos.codebook_pattern`, `zonos.backbone._torch` and of
`transformers.models.dac.DacModel.decode`.  Only data (inputs' seeds + expected outputs) is written.

    python tests/golden/make_golden.py [--only tiny,full,ops,sampling,eos,dac,dacenc,spk,longfull,tinyb]
"""
import argparse
import importlib.machinery
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from zonos_amd import synth  # noqa: E402

REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    class _Log:
        def __getattr__(self, _):
            return lambda *a, **k: None
    _stub("loguru", logger=_Log())
    ta = _stub("torchaudio")
    ta.functional = _stub("torchaudio.functional")
    ph = _stub("phonemizer")
    ph.backend = _stub("phonemizer.backend", EspeakBackend=object)
    _stub("inflect", engine=lambda: None)
    _stub("kanjize", number2kanji=lambda *a, **k: "")
    _stub("sudachipy", Dictionary=lambda *a, **k: SimpleNamespace(create=lambda *a, **k: None), SplitMode=SimpleNamespace(A=None))
    sys.path.insert(0, REF)
    import zonos.model as zm
    return zm


def build_reference_model(zm, cfg, seed, peaky=False):
    from zonos.backbone._torch import TorchZonosBackbone
    from zonos.config import BackboneConfig, PrefixConditionerConfig, ZonosConfig
    from zonos.utilities.generation_utils import CUDAGraphManager
    import torch.nn as nn
    zc = ZonosConfig(BackboneConfig(d_model=cfg["d_model"], n_layer=cfg["n_layer"],
                                    attn_mlp_d_intermediate=cfg["d_ff"], attn_layer_idx=list(range(cfg["n_layer"])),
                                    attn_cfg=dict(num_heads=cfg["num_heads"], num_heads_kv=cfg["num_heads_kv"])),
                     PrefixConditionerConfig([], "none"))
    m = zm.Zonos.__new__(zm.Zonos)
    nn.Module.__init__(m)
    m.config = zc
    m.eos_token_id, m.masked_token_id = 1024, 1025
    m.autoencoder = SimpleNamespace(num_codebooks=9)
    with torch.device("meta"):
        bb = TorchZonosBackbone(zc.backbone)
        emb = nn.ModuleList([nn.Embedding(1032, cfg["d_model"]) for _ in range(9)])
        heads = nn.Linear(cfg["d_model"], 9 * 1025, bias=False)
    m.backbone, m.embeddings, m.fused_heads = bb, emb, heads
    m._cuda_graph_manager = CUDAGraphManager()
    sd = synth.zonos_state_dict(cfg, seed, peaky=peaky)
    m.load_state_dict(sd, assign=True, strict=True)
    return m.eval(), sd


class Recorder:
    """Wraps zonos.model.sample_from_logits: records logits/tokens per call, optionally forces cb0 EOS."""
    def __init__(self, zm, keep_logits=None, force_eos_at=None):
        self.zm, self.real = zm, zm.sample_from_logits
        self.keep, self.force = keep_logits, force_eos_at
        self.logits, self.tokens, self.inputs, self.margin, self.calls = {}, [], [], [], 0

    def __call__(self, logits, **kw):
        i = self.calls          # 0 = prefill sample, i>=1 = loop step i-1
        self.calls += 1
        if self.force is not None and i - 1 == self.force:
            logits = logits.clone()
            logits[:, 0, 1024] = 1.0e4
        if self.keep is None or i in self.keep:
            self.logits[i] = logits.clone().numpy()
        pen = logits
        if kw.get("generated_tokens") is not None:
            self.inputs.append(kw["generated_tokens"][..., -1].clone().numpy())
            from zonos.sampling import modify_logit_for_repetition_penalty
            pen = modify_logit_for_repetition_penalty(logits, kw["generated_tokens"], kw.get("repetition_penalty", 3.0),
                                                      kw.get("repetition_penalty_window", 2))
        t2 = torch.topk(pen, 2, dim=-1).values
        self.margin.append((t2[..., 0] - t2[..., 1]).numpy())
        tok = self.real(logits, **kw)
        self.tokens.append(tok.squeeze(-1).clone().numpy())
        return tok

    def __enter__(self):
        self.zm.sample_from_logits = self
        return self

    def __exit__(self, *a):
        self.zm.sample_from_logits = self.real


def gen_case(zm, model, cond, max_new, prefix=None, keep=None, force=None, sampling=None):
    with Recorder(zm, keep, force) as rec:
        out = model.generate(cond, audio_prefix_codes=prefix, max_new_tokens=max_new, cfg_scale=2.0, batch_size=1,
                             sampling_params=sampling or {"temperature": 0.0}, disable_torch_compile=True)
    steps = sorted(rec.logits)
    return dict(out=out.numpy().astype(np.int16), tokens=np.stack(rec.tokens).astype(np.int16),
                inputs=np.stack(rec.inputs).astype(np.int16) if rec.inputs else np.zeros((0, 1, 9), np.int16),
                margin=np.stack(rec.margin).astype(np.float32), logit_steps=np.array(steps, np.int32),
                logits=(np.stack([rec.logits[s] for s in steps]) if steps else np.zeros((0,), np.float32)).astype(np.float32), n_calls=np.int32(rec.calls))


def bf16_bits(t):
    return t.contiguous().view(torch.int16).numpy().copy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="tiny,full,ops,sampling,eos,dac,peaky,cond")
    args = ap.parse_args()
    only = set(args.only.split(","))
    torch.manual_seed(0)
    zm = import_reference()

    if "tiny" in only or "eos" in only:
        cfg, seed = synth.TINY_CFG, 77
        model, _ = build_reference_model(zm, cfg, seed)
        cond = synth.conditioning(seed, "cond", 2, 6, cfg["d_model"])
    if "tiny" in only:
        c = gen_case(zm, model, cond, 24)
        np.savez_compressed(f"{HERE}/tiny_gen.npz", seed=seed, l_c=6, max_new=24, **c)
        pre = torch.from_numpy(synth.randint(seed, "prefix", (1, 9, 5), 1024))
        c = gen_case(zm, model, cond, 16, prefix=pre)
        np.savez_compressed(f"{HERE}/tiny_gen_prefix.npz", seed=seed, l_c=6, max_new=16, prefix_len=5, **c)
        print("tiny done")
    if "eos" in only:
        cases = {}
        for s in (0, 3, 6, 7, 10, 14, 15, 17, 22, 23, 30, 38, 45):
            c = gen_case(zm, model, cond, 48, keep=(), force=s)
            cases[f"out_{s}"] = c["out"]
            cases[f"calls_{s}"] = c["n_calls"]
            cases[f"tokens_{s}"] = c["tokens"]
        pre = torch.from_numpy(synth.randint(seed, "prefix", (1, 9, 5), 1024))
        for s in (2, 9):
            c = gen_case(zm, model, cond, 32, prefix=pre, keep=(), force=s)
            cases[f"pout_{s}"] = c["out"]
            cases[f"pcalls_{s}"] = c["n_calls"]
            cases[f"ptokens_{s}"] = c["tokens"]
        np.savez_compressed(f"{HERE}/tiny_eos.npz", seed=seed, l_c=6, max_new=48, p_max_new=32, prefix_len=5, **cases)
        print("eos done")

    if "full" in only or "ops" in only:
        cfg, seed = synth.FULL_CFG, 1234
        model, sd = build_reference_model(zm, cfg, seed)
    if "full" in only:
        cond = synth.conditioning(seed, "cond", 2, 24, cfg["d_model"])
        c = gen_case(zm, model, cond, 64, keep=(0, 1, 2, 32, 64))
        np.savez_compressed(f"{HERE}/full_gen.npz", seed=seed, l_c=24, max_new=64, **c)
        print("full done")
    if "ops" in only:
        # one decode step of reference TransformerBlock 0 at full dims over a synthetic KV history
        from zonos.config import InferenceParams
        out = {}
        for L in (1, 17, 900):
            R, d = 2, cfg["d_model"]
            x = synth.conditioning(seed, f"ops.x.{L}", R, 1, d)
            kv = torch.from_numpy(synth.normal(seed, f"ops.kv.{L}", (R, 904, 2, 4, 128))).to(torch.bfloat16)
            ip = InferenceParams(904, R, L - 1, 0, {0: (kv, None)}, torch.full((R,), L - 1, dtype=torch.int32))
            model.backbone.freqs_cis = model.backbone.allocate_inference_cache_pure(1, 8)[1]
            fc = model.backbone.freqs_cis[ip.lengths_per_sample.long().unsqueeze(-1)]
            # per-op intermediates of the reference block (forward hooks on ITS modules; data only): n1 = norm(x), qkv = in_proj(n1),
            # a = the SDPA output entering out_proj, o1 / o2 = the two out_proj calls (_torch.py:419-420), n2 = norm2(x1), u = fc1(n2),
            # m = y * silu(gate) entering fc2, f = fc2(m); x1 = x + o2 follows from them
            blk = model.backbone.layers[0]
            rec, hooks = {}, []
            def keep(name, multi=False):
                def hook(mod, inp, outp):
                    if multi:
                        rec.setdefault(name, []).append((inp[0].detach().clone(), outp.detach().clone()))
                    else:
                        rec[name] = (inp[0].detach().clone(), outp.detach().clone())
                return hook
            hooks.append(blk.norm.register_forward_hook(keep("norm")))
            hooks.append(blk.mixer.in_proj.register_forward_hook(keep("in_proj")))
            hooks.append(blk.mixer.out_proj.register_forward_hook(keep("out_proj", multi=True)))
            hooks.append(blk.norm2.register_forward_hook(keep("norm2")))
            hooks.append(blk.mlp.fc1.register_forward_hook(keep("fc1")))
            hooks.append(blk.mlp.fc2.register_forward_hook(keep("fc2")))
            with torch.inference_mode():
                y = blk(x, ip, fc)
            for h_ in hooks:
                h_.remove()
            out[f"y_{L}"] = bf16_bits(y)
            out[f"knew_{L}"] = bf16_bits(kv[:, L - 1, 0])
            out[f"vnew_{L}"] = bf16_bits(kv[:, L - 1, 1])
            out[f"n1_{L}"] = bf16_bits(rec["norm"][1])
            out[f"qkv_{L}"] = bf16_bits(rec["in_proj"][1])
            out[f"a_{L}"] = bf16_bits(rec["out_proj"][0][0])
            out[f"o1_{L}"] = bf16_bits(rec["out_proj"][0][1])
            out[f"o2_{L}"] = bf16_bits(rec["out_proj"][1][1])
            out[f"x1_{L}"] = bf16_bits(rec["norm2"][0])
            out[f"n2_{L}"] = bf16_bits(rec["norm2"][1])
            out[f"u_{L}"] = bf16_bits(rec["fc1"][1])
            out[f"m_{L}"] = bf16_bits(rec["fc2"][0])
            out[f"f_{L}"] = bf16_bits(rec["fc2"][1])
        np.savez_compressed(f"{HERE}/full_layer0.npz", seed=seed, **out)
        print("ops done")

    if "longfull" in only:
        # BASELINE config 5's prefill at full dims: a 30 s audio prefix (P = 2584 synthetic codes) + 8 new tokens through the
        # reference's generate(): prefill logits over 24 + 2585 positions (the batched causal SDPA path), the first loop logits,
        # tokens.  Takes a few minutes on the build container's CPU cores.
        cfg, seed = synth.FULL_CFG, 1234
        if "full" not in only and "ops" not in only:
            model, sd = build_reference_model(zm, cfg, seed)
        P = int(os.environ.get("ZN_GOLDEN_PREFIX", "2584"))
        cond = synth.conditioning(seed, "cond", 2, 24, cfg["d_model"])
        pre = torch.from_numpy(synth.randint(seed, "longprefix", (1, 9, P), 1024))
        c = gen_case(zm, model, cond, 8, prefix=pre, keep=(0, 1, 2, 8))
        c["out"] = c["out"][..., -16:]          # the prefix part of the output is the input
        np.savez_compressed(f"{HERE}/full_gen_longprefix.npz", seed=seed, l_c=24, max_new=8, prefix_len=P, **c)
        print("longfull done")

    if "peaky" in only:
        # decisive-margin ("peaky" head) variants for free-running greedy parity
        cfg, seed = synth.TINY_CFG, 77
        model, _ = build_reference_model(zm, cfg, seed, peaky=True)
        cond = synth.conditioning(seed, "cond", 2, 6, cfg["d_model"])
        c = gen_case(zm, model, cond, 96, keep=(0, 1, 50))
        np.savez_compressed(f"{HERE}/tiny_gen_peaky.npz", seed=seed, l_c=6, max_new=96, **c)
        cfg, seed = synth.FULL_CFG, 1234
        model, _ = build_reference_model(zm, cfg, seed, peaky=True)
        cond = synth.conditioning(seed, "cond", 2, 24, cfg["d_model"])
        c = gen_case(zm, model, cond, 160, keep=(0, 1, 100))
        np.savez_compressed(f"{HERE}/full_gen_peaky.npz", seed=seed, l_c=24, max_new=160, **c)
        print("peaky done")

    if "cond" in only:
        # reference PrefixConditioner (zonos/conditioning.py) with the transformer's conditioner list, bf16, synthetic weights;
        # phonemisation (external espeak) is bypassed: the phoneme strings are given
        import zonos.conditioning as zc
        from zonos.config import PrefixConditionerConfig
        out = {}
        for d, seed, projection in ((128, 77, "none"), (2048, 1234, "none"), (128, 78, "mlp")):
            pc = zc.PrefixConditioner(PrefixConditionerConfig(list(synth.TRANSFORMER_CONDITIONERS), projection), d).to(torch.bfloat16)
            sd = synth.conditioner_state_dict(synth.TRANSFORMER_CONDITIONERS, d, seed, projection)
            pc.load_state_dict(sd, strict=True)
            phon = "h@l'oU w'3:ld, tEst!"
            zc.phonemize = lambda texts, languages, _p=phon: [_p for _ in texts]
            spk = torch.from_numpy(synth.normal(seed, "cond.speaker", (1, 1, 128))).to(torch.bfloat16)
            cd = zc.make_cond_dict(text="ignored", language="en-us", speaker=spk, emotion=[0.5, 0.05, 0.05, 0.05, 0.05, 0.05, 0.1, 0.15],
                                   fmax=22050.0, pitch_std=45.0, speaking_rate=13.0, device="cpu")
            from zonos.utilities.conditioning_cache import prepare_conditioning_with_cache
            with torch.inference_mode():
                full = prepare_conditioning_with_cache(pc, cd, None, False, 2.0, None)
            tag = f"d{d}_{projection}"
            out[tag] = bf16_bits(full)
            out[tag + "_ids"] = zc.tokenize_phonemes([phon])[0].numpy()
            out[tag + "_seed"] = seed
            out[tag + "_emotion"] = cd["emotion"].numpy()
            out[tag + "_langid"] = cd["language_id"].numpy()
        out["phonemes"] = np.array(phon)
        np.savez_compressed(f"{HERE}/conditioner.npz", **out)
        print("cond done")

    if "sampling" in only:
        import zonos.sampling as zs
        seed = 99
        lg = torch.from_numpy(synth.normal(seed, "logits", (2, 9, 1025), 3.0))
        gen = torch.from_numpy(synth.randint(seed, "gen", (2, 9, 7), 1025))
        gen[0, 0, -1] = gen[0, 0, -2]          # duplicate -> factor 9
        gen[1, 3, -1] = 1025                   # MASK clamps to 1024
        out = dict(seed=seed)
        out["rep"] = zs.modify_logit_for_repetition_penalty(lg, gen, 3.0, 2).numpy()
        pr = torch.softmax(lg, -1)
        out["softmax"] = pr.numpy()
        out["unified"] = zs.apply_unified(pr, 0.5, 0.4, 0.0).numpy()
        out["unified_q"] = zs.apply_unified(pr, 0.7, -0.1, 0.2).numpy()
        out["top_p"] = zs.apply_top_p(pr.clone(), 0.8).numpy()
        out["top_k"] = zs.apply_top_k(pr.clone(), 50).numpy()
        out["min_p"] = zs.apply_min_p(pr.clone(), 0.1).numpy()
        out["greedy"] = zs.sample_from_logits(lg, temperature=0.0, generated_tokens=gen).squeeze(-1).numpy()
        out["gen"] = gen.numpy()
        codes = torch.from_numpy(synth.randint(seed, "codes", (2, 9, 13), 1024))
        from zonos.codebook_pattern import apply_delay_pattern, revert_delay_pattern
        dl = apply_delay_pattern(codes, 1025)
        out["delayed"] = dl.numpy()
        out["reverted"] = revert_delay_pattern(dl).numpy()
        np.savez_compressed(f"{HERE}/sampling.npz", **out)
        print("sampling done")

    if "dac" in only:
        from transformers.models.dac import DacConfig, DacModel
        seed = 4321
        dm = DacModel(DacConfig(sampling_rate=44100)).eval()
        dsd = synth.dac_state_dict(seed)
        full = dm.state_dict()
        missing = [k for k in dsd if k not in full]
        assert not missing, missing
        full.update(dsd)
        dm.load_state_dict(full)
        out = dict(seed=seed)
        for T in (16, 40):
            codes = torch.from_numpy(synth.randint(seed, f"codes{T}", (1, 9, T), 1024))
            with torch.no_grad():
                wav = dm.decode(audio_codes=codes).audio_values
            out[f"wav_{T}"] = wav.numpy().astype(np.float32)
        codes = torch.from_numpy(synth.randint(seed, "codes16", (1, 9, 16), 1024))
        with torch.no_grad():
            h = dm.decoder.conv1(dm.quantizer.from_codes(codes)[0])
            out["rms_conv1"] = np.float32(h.pow(2).mean().sqrt())
            for bi, blk in enumerate(dm.decoder.block):
                h = blk(h)
                out[f"rms_block{bi}"] = np.float32(h.pow(2).mean().sqrt())
        np.savez_compressed(f"{HERE}/dac.npz", **out)
        print("dac done")

    if "dacenc" in only:
        # DacModel.encode (the call zonos/autoencoder.py:117 makes) on synthetic waveforms: codes + encoder latents
        from transformers.models.dac import DacConfig, DacModel
        seed = 4321
        dm = DacModel(DacConfig(sampling_rate=44100)).eval()
        dsd = synth.dac_state_dict(seed)
        full = dm.state_dict()
        missing = [k for k in dsd if k not in full]
        assert not missing, missing
        full.update(dsd)
        dm.load_state_dict(full)
        out = dict(seed=seed)
        for T in (512 * 6, 512 * 23):
            wav = synth.test_waveform(seed, f"encwav{T}", T)
            with torch.no_grad():
                enc = dm.encode(wav)
                z = dm.encoder(wav)
            out[f"codes_{T}"] = enc.audio_codes.numpy().astype(np.int16)
            out[f"z_{T}"] = z.numpy().astype(np.float32)
        np.savez_compressed(f"{HERE}/dac_encode.npz", **out)
        print("dacenc done")

    if "spk" in only:
        # the reference's own ResNet293_based (zonos/speaker_cloning.py:419-472) + the LDA Linear, synthetic weights and
        # features; the feature front end (torchaudio MelSpectrogram) cannot run here (torchaudio absent) and is not part
        # of these vectors
        import importlib.machinery
        import types
        for name, attrs in (("torchaudio", {}), ("torchaudio.functional", {}),
                            ("torchaudio.transforms", dict(MelSpectrogram=lambda **k: None, Resample=lambda *a, **k: None))):
            if name not in sys.modules:
                m = types.ModuleType(name)
                m.__spec__ = importlib.machinery.ModuleSpec(name, None)
                for k, v in attrs.items():
                    setattr(m, k, v)
                sys.modules[name] = m
        sys.modules["torchaudio"].transforms = sys.modules["torchaudio.transforms"]
        from zonos.speaker_cloning import ResNet293_based
        seed = 2468
        sd, lda = synth.speaker_state_dict(seed)
        net = ResNet293_based().eval()
        net.load_state_dict(sd, strict=True)
        out = dict(seed=seed)
        for T in (64, 104):
            feats = synth.speaker_features(seed, f"feats{T}", 1, 80, T)
            with torch.no_grad():
                h = net.front(feats.unsqueeze(1))
                emb = net.bottleneck(net.pooling(h))
                ld = torch.nn.functional.linear(emb, lda["weight"], lda["bias"])
            out[f"emb_{T}"] = emb.numpy().astype(np.float32)
            out[f"lda_{T}"] = ld.numpy().astype(np.float32)
            out[f"front_rms_{T}"] = np.float32(h.pow(2).mean().sqrt())
        np.savez_compressed(f"{HERE}/speaker.npz", **out)
        print("spk done")


if __name__ == "__main__":
    main()
