"""Ties the oracle's batched generate() back to the pinned batch-1 oracle.

The reference crashes for batch_size >= 2 under CFG (zonos/utilities/generation_utils.py:237-238), so it
holds no batched fixture: the batch oracle the GPU tests compare with (tests/test_gpu_decode.py,
test_batched_generate_vs_batched_oracle / test_full_dims_batch8_vs_batched_oracle) is only as good as its agreement with B
independent batch-1 runs of the restatement that tests/test_oracle_golden.py pins bit for bit against the reference.  Rows of
a batch never interact (attention, LayerNorm and the sampler are per row; only the GEMMs see the other rows, through their
blocking), so per utterance: sampled tokens and output codes equal, logits equal up to the summation order of the CPU GEMM
at another row count (a few bf16 ulps of a hidden value; measured 0 on this container's oneDNN build)."""
import numpy as np
import pytest
import torch

from oracle import zonos_oracle as zo
from zonos_amd import synth

GREEDY = {"temperature": 0.0}


@pytest.mark.parametrize("B,with_prefix", [(2, False), (3, True), (5, True)])
def test_batched_oracle_equals_solo_oracle_runs(B, with_prefix):
    cfg = synth.TINY_CFG
    w = synth.zonos_state_dict(cfg, 77)
    conds = [synth.conditioning(300 + i, "cond", 2, 6, cfg["d_model"]) for i in range(B)]
    cond = torch.cat([c[0:1] for c in conds] + [c[1:2] for c in conds], 0)
    pre = torch.from_numpy(synth.randint(9, "bo.prefix", (B, 9, 5), 1024)) if with_prefix else None
    N = 24
    noeos = lambda s_, l: l.index_fill(2, torch.tensor([1024]), -float("inf"))
    btr = zo.GenTrace()
    out_b = zo.generate(w, cfg, cond, audio_prefix_codes=pre, max_new_tokens=N, batch_size=B, sampling_params=GREEDY, trace=btr, logits_hook=noeos)
    worst = 0.0
    for i in range(B):
        str_ = zo.GenTrace()
        out_s = zo.generate(w, cfg, conds[i], audio_prefix_codes=None if pre is None else pre[i:i + 1], max_new_tokens=N, batch_size=1,
                            sampling_params=GREEDY, trace=str_, logits_hook=noeos)
        assert out_s.shape[-1] == out_b.shape[-1]
        assert torch.equal(out_b[i], out_s[0]), f"utterance {i}: codes differ"
        assert len(btr.tokens) == len(str_.tokens)
        for k in range(len(str_.tokens)):
            assert torch.equal(btr.tokens[k][i], str_.tokens[k][0]), (i, k)
            a, b = btr.logits[k][i].numpy(), str_.logits[k][0].numpy()
            fin = np.isfinite(b)
            assert np.array_equal(np.isfinite(a), fin)
            d = float(np.abs(np.where(fin, a, 0.0) - np.where(fin, b, 0.0)).max())
            worst = max(worst, d)
            # one bf16 ulp of a logit of magnitude < 8 is 2^-5: a GEMM blocked for another row count may flip a few roundings
            assert d <= 0.0625, (i, k, d)
    print(f"\n[batched oracle B={B} vs solo oracle runs] worst |dlogit| {worst:.4g}")
