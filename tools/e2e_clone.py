"""BASELINE config 5 end to end on one GPU, every stage on the HIP path: a 30 s reference waveform (synthetic) ->
speaker embedding (ResNet293-SimAM + LDA) and DAC preprocess + encode (audio prefix codes) -> make_cond_dict /
prepare_conditioning (phoneme ids given directly: text -> phonemes needs espeak, absent here) -> generate 30 s with the
prefix -> DAC decode.  Synthetic weights everywhere; EOS suppressed so that the full length is generated.
    python tools/e2e_clone.py [prefix_seconds=30] [new_seconds=30]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zonos_amd import synth  # noqa: E402
from zonos_amd.autoencoder import DACAutoencoder  # noqa: E402
from zonos_amd.conditioning import make_cond_dict  # noqa: E402
from zonos_amd.speaker_cloning import SpeakerEmbeddingLDA  # noqa: E402
from zonos_amd.testing import build_model  # noqa: E402

PS = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
NS = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
dev = "cuda:0"
dac = DACAutoencoder(synth.dac_state_dict(4321), device=dev)
model, _ = build_model(synth.FULL_CFG, 1234, dev, dac=dac, conditioners=synth.TRANSFORMER_CONDITIONERS)
spk = SpeakerEmbeddingLDA(*synth.speaker_state_dict(2468), device=dev)
model.engine(1).call("zn_debug_eos_bias", float("-inf"))
wav = synth.test_waveform(5, "clone", int(PS * 44100))[0].to(dev)      # [1, samples] at 44.1 kHz
N = int(round(NS * 86.1328))
ids = torch.from_numpy(synth.randint(5, "phonemes", (1, 40), synth.N_PHONEME_TOKENS))


def stage(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return out, dt


for it in range(2):
    times = {}
    (emb, lda), times["speaker embedding"] = stage("spk", lambda: spk(wav, 44100))
    prefix, times["DAC preprocess + encode"] = stage("enc", lambda: dac.encode(dac.preprocess(wav, 44100).unsqueeze(0)))
    cd = make_cond_dict(speaker=lda.to(torch.bfloat16), language="en-us", device=dev)
    cd["espeak"] = ("ids", ids)
    cond, times["prepare_conditioning"] = stage("cond", lambda: model.prepare_conditioning(cd, cfg_scale=2.0))
    codes, times["generate"] = stage("gen", lambda: model.generate(cond, audio_prefix_codes=prefix, max_new_tokens=N, sampling_params={"temperature": 0.0}))
    out, times["DAC decode"] = stage("dec", lambda: dac.decode(codes))
    total = sum(times.values())
    if it == 1:
        print(f"config 5 end to end: {PS:.0f} s reference audio ({prefix.shape[-1]} prefix frames), {N} new tokens, conditioning {tuple(cond.shape)}, codes {tuple(codes.shape)}")
        for k, v in times.items():
            print(f"  {k:26s} {v * 1e3:9.1f} ms")
        print(f"  total {total:.3f} s for {N / 86.1328:.1f} s of new audio -> {N / 86.1328 / total:.2f}x real-time")
