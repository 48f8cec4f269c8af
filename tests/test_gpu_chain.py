"""The persistent per-block decode chain (csrc/zn_chain_kernel.h: out_proj -> out_proj + residual -> LayerNorm + fc1 + SiLU gate
-> fc2 + residual -> next block's LayerNorm + in_proj + RoPE + KV append in ONE launch, in-kernel hand-offs) against
(a) the CPU oracle and (b) the launches path, which it must reproduce bit for bit: same tiles, same summation order.

Every batch-1 test of tests/test_gpu_decode.py at the Zonos-v0.1 dimensions runs this path too (it is the default there);
this file adds the smallest configuration the kernel serves, where the oracle can follow a whole generation."""
import time

import numpy as np
import pytest
import torch

from oracle import zonos_oracle as zo
from zonos_amd import synth
from zonos_amd.testing import build_model

pytestmark = pytest.mark.gpu
GREEDY = {"temperature": 0.0}


def _run(model, cond, max_new, toks=None, chain=True, prefix=None):
    eng = model.engine(1)
    eng.call("zn_debug_tune", 8, 1 if chain else 2)
    eng.call("zn_debug_eos_bias", float("-inf"))
    if toks is not None:
        tk = torch.from_numpy(toks.astype(np.int32)).to("cuda:0").contiguous()
        eng.call("zn_debug_token_override", tk.data_ptr(), tk.shape[0])
    try:
        tr = {"logits": []}
        out = model.generate(cond.to("cuda:0"), audio_prefix_codes=prefix, max_new_tokens=max_new, sampling_params=GREEDY, _trace=tr)
        path = eng.lib.zn_decode_path(eng.h)
    finally:
        eng.call("zn_debug_token_override", None, 0)
        eng.call("zn_debug_eos_bias", 0.0)
        eng.call("zn_debug_tune", 8, 1)
    return out.cpu(), torch.stack(tr["logits"]).cpu(), path


def test_chain_generate_vs_oracle():
    """d_model 512, three blocks: prefill (batched) + 48 decode steps through the chain kernel, the oracle's token stream
    fed through the override hook: codes bit-equal, every call's logits within 0.06, decisive argmax equal."""
    cfg = synth.CHAIN_CFG
    model, w = build_model(cfg, 55, "cuda:0")
    cond = synth.conditioning(55, "cond", 2, 9, cfg["d_model"])
    pre = torch.from_numpy(synth.randint(55, "prefix", (1, 9, 7), 1024))
    N = 40
    otr = zo.GenTrace()
    noeos = lambda s_, l: l.index_fill(2, torch.tensor([1024]), -float("inf"))
    ref_out = zo.generate(w, cfg, cond, audio_prefix_codes=pre, max_new_tokens=N, sampling_params=GREEDY, trace=otr, logits_hook=noeos)
    toks = torch.stack(otr.tokens).numpy()
    out, logits, path = _run(model, cond, N, toks=toks, prefix=pre.to("cuda:0"))
    assert path == 1, "the chain kernel did not serve this configuration"
    assert torch.equal(out, ref_out)
    worst = 0.0
    for k in range(len(otr.logits)):
        a, b = logits[k].numpy(), otr.logits[k].numpy()
        fin = np.isfinite(b)
        worst = max(worst, float(np.abs(np.where(fin, a - b, 0.0)).max()))
        t2 = np.sort(np.where(fin, b, -1e30), -1)[..., -2:]
        dec = (t2[..., 1] - t2[..., 0]) > 0.15
        assert (np.where(fin, a, -1e30).argmax(-1) == np.where(fin, b, -1e30).argmax(-1))[dec].all(), k
    print(f"\n[chain, d 512 x 3 blocks, vs oracle] {len(otr.logits)} calls, worst |dlogit| {worst:.4g}")
    assert worst <= 0.06


@pytest.mark.parametrize("which", ["chain512", "full-chain", "full-step", "full-step-nopre"])
def test_chain_is_bit_identical_to_the_launches_path(which):
    """Free-running greedy generation through the persistent kernels and through the per-op launches (zn_debug_tune(8, 2)).  At the
    Zonos-v0.1-transformer dimensions every path cuts every dot product the same way (fc2's K = 8192 in four quarters): equal
    codes and bit-equal logits at every step (200 steps: 8-step graphs, the fused attention arithmetic, hand-offs replayed
    26 x 6 x 200 times), for one chain launch per block (zn_debug_tune(15, 2)), for the whole-step kernel (the default: block 0's in_proj
    inside the launch, on the helper and communication waves) and for the whole-step kernel behind an in_proj launch (zn_debug_tune(18, 2)).  At
    d_model 512 the launches path keeps fc2's K = 2048 in one wave while the chain splits it in quarters - another summation
    order: equal codes, logits within one bf16 ulp of a hidden value."""
    cfg, seed, n = (synth.CHAIN_CFG, 55, 60) if which == "chain512" else (synth.FULL_CFG, 1234, 200)
    model, _ = build_model(cfg, seed, "cuda:0")
    eng = model.engine(1)
    cond = synth.conditioning(seed, "cond", 2, 24, cfg["d_model"])
    eng.call("zn_debug_tune", 15, {"full-chain": 2}.get(which, 1))
    eng.call("zn_debug_tune", 18, 2 if which == "full-step-nopre" else 1)
    try:
        a, la, pa = _run(model, cond, n, chain=True)
        b, lb, pb = _run(model, cond, n, chain=False)
        assert (pa, pb) == (1, 0)
        assert eng.lib.zn_decode_path_detail(eng.h) in (0, 1, 2, 3)
        assert torch.equal(a, b)
        assert la.shape == lb.shape
        if which != "chain512":
            assert torch.equal(la.view(torch.int32), lb.view(torch.int32))
        else:
            fin = torch.isfinite(lb)
            same = float((la.view(torch.int32) == lb.view(torch.int32))[fin].float().mean())
            worst = float((la - lb)[fin].abs().max())
            print(f"\n[chain vs launches, d 512] logits bit-equal {same:.5f}, max |d| {worst:.4g}")
            assert same > 0.98 and worst <= 0.04
        a2, la2, _ = _run(model, cond, n, chain=True)           # replay: no state survives a generation (counters, timeout word)
        assert torch.equal(a, a2) and torch.equal(la.view(torch.int32), la2.view(torch.int32))
        assert eng.counters()["handoff_timeouts"] == 0
    finally:
        eng.call("zn_debug_tune", 15, 1)
        eng.call("zn_debug_tune", 18, 1)


def test_whole_step_kernel_is_bit_identical_to_the_chain_path():
    """The whole-step kernel (csrc/zn_step_kernel.h: every block of the decode step in ONE launch; the default at batch 1) against one
    attention launch (two beyond 512 keys) + one chain launch per block (zn_debug_tune(15, 2)) at the Zonos-v0.1-transformer dimensions:
    free-running greedy codes equal and the logits of every step bit-equal, over single-step launches (trace mode, 40 steps) and over
    8-step graphs (300 steps: contexts 26 .. 333), with an audio prefix (contexts 426 .. 700, across the 512-key boundary where a second
    attention workgroup per (row, kv head) joins) and through the second and third block (contexts 650 .. 1050, logits of every step).  A hand-off timeout is an error (the bounded waits describe themselves: zn_last_error)."""
    cfg, seed = synth.FULL_CFG, 1234
    model, _ = build_model(cfg, seed, "cuda:0")
    eng = model.engine(1)
    cond = synth.conditioning(seed, "cond", 2, 24, cfg["d_model"])
    try:
        eng.call("zn_debug_tune", 15, 2)
        a, la, pa = _run(model, cond, 40)
        assert eng.lib.zn_decode_path_detail(eng.h) == 1
        eng.call("zn_debug_tune", 15, 1)
        b, lb, pb = _run(model, cond, 40)
        assert eng.lib.zn_decode_path_detail(eng.h) == 2, "the whole-step kernel did not serve this configuration"
        assert torch.equal(a, b)
        assert torch.equal(la.view(torch.int32), lb.view(torch.int32))
        eng.call("zn_debug_eos_bias", float("-inf"))
        outs = []
        for t15 in (2, 1):
            eng.call("zn_debug_tune", 15, t15)
            outs.append(model.generate(cond.to("cuda:0"), max_new_tokens=300, sampling_params=GREEDY).cpu())
        assert torch.equal(outs[0], outs[1])
        pre = torch.from_numpy(synth.randint(seed, "prefix", (1, 9, 400), 1024)).to("cuda:0")
        outs = []
        for t15 in (2, 1):
            eng.call("zn_debug_tune", 15, t15)
            outs.append(model.generate(cond.to("cuda:0"), audio_prefix_codes=pre, max_new_tokens=268, sampling_params=GREEDY).cpu())
            assert eng.lib.zn_decode_path_detail(eng.h) == (1 if t15 == 2 else 2)
        assert torch.equal(outs[0], outs[1])
        pre = torch.from_numpy(synth.randint(seed, "prefix2", (1, 9, 620), 1024)).to("cuda:0")
        outs = []
        for t15 in (2, 1):
            eng.call("zn_debug_tune", 15, t15)
            tr = {"logits": []}
            o = model.generate(cond.to("cuda:0"), audio_prefix_codes=pre, max_new_tokens=400, sampling_params=GREEDY, _trace=tr)
            outs.append((o.cpu(), torch.stack(tr["logits"]).cpu()))
            assert eng.lib.zn_decode_path_detail(eng.h) == (1 if t15 == 2 else 2)
        assert torch.equal(outs[0][0], outs[1][0])
        assert torch.equal(outs[0][1].view(torch.int32), outs[1][1].view(torch.int32))
        assert eng.counters()["handoff_timeouts"] == 0
    finally:
        eng.call("zn_debug_tune", 15, 1)
        eng.call("zn_debug_eos_bias", 0.0)


def test_whole_step_kernel_at_long_contexts_is_bit_identical_to_the_per_block_path():
    """Contexts beyond 1024 keys (the reference's default call is 30 s = 2.6 k keys, `zonos/model.py:359`; BASELINE config 5 runs at
    2.6 - 5.2 k): the whole-step kernel's attention role (one workgroup per (row, kv head, 512-key block), K and full-width V of
    the block in registers a block ahead, block maxima and partials exchanged as granules, the in-order combine of the split P.V pass)
    against the per-block path (attn_scores_kernel + attn_block_kernel + one chain launch per block, zn_debug_tune(15, 2)) and, for
    the traced steps, against the launches path (zn_debug_tune(8, 2)): equal codes over the whole run (8-step graphs; contexts 1025 ..
    2725, 2585 .. 5225 = config 5, and across the 3072- and 4096-key changes of instantiation, up to the kernel's 6144-key limit and
    past it, where both runs take the per-block path) and bit-equal logits at every traced step."""
    cfg, seed = synth.FULL_CFG, 1234
    model, _ = build_model(cfg, seed, "cuda:0")
    eng = model.engine(1)
    cond = synth.conditioning(seed, "cond", 2, 24, cfg["d_model"]).to("cuda:0")
    for prefix, new, traced in ((1000, 1700, 40), (2560, 2640, 24), (3060, 40, 40), (4080, 40, 40), (5590, 640, 16)):
        _long_context_case(model, eng, cond, seed, prefix, new, traced)


def _long_context_case(model, eng, cond, seed, prefix, new, traced):
    pre = torch.from_numpy(synth.randint(seed, f"longprefix{prefix}", (1, 9, prefix), 1024)).to("cuda:0")
    try:
        eng.call("zn_debug_eos_bias", float("-inf"))
        outs = []
        for t15 in (2, 1):
            eng.call("zn_debug_tune", 15, t15)
            outs.append(model.generate(cond, audio_prefix_codes=pre, max_new_tokens=new, sampling_params=GREEDY).cpu())
            want = 1 if (t15 == 2 or 24 + prefix + new + 8 > 6144) else 2
            assert eng.lib.zn_decode_path_detail(eng.h) == want, (t15, eng.lib.zn_decode_path_detail(eng.h))
        assert outs[0].shape[-1] == prefix + new and torch.equal(outs[0], outs[1])
        logs = []
        for t15, t8 in ((2, 1), (1, 1), (1, 2)):
            eng.call("zn_debug_tune", 15, t15)
            eng.call("zn_debug_tune", 8, t8)
            tr = {"logits": []}
            o = model.generate(cond, audio_prefix_codes=pre, max_new_tokens=traced, sampling_params=GREEDY, _trace=tr)
            logs.append((o.cpu(), torch.stack(tr["logits"]).cpu()))
        for o, lg in logs[1:]:
            assert torch.equal(o, logs[0][0])
            assert torch.equal(lg.view(torch.int32), logs[0][1].view(torch.int32))
        assert eng.counters()["handoff_timeouts"] == 0
    finally:
        eng.call("zn_debug_tune", 8, 1)
        eng.call("zn_debug_tune", 15, 1)
        eng.call("zn_debug_eos_bias", 0.0)


def test_second_generation_on_the_device_takes_the_launches_path():
    """One generation per device owns the persistent kernels (their hand-offs need every workgroup of the grid resident): a
    generation that begins on ANOTHER handle while the first is still running gets the launches path, with the same codes; once
    the first has ended (zn_gen_end) the device is free again."""
    cfg = synth.CHAIN_CFG
    m1, _ = build_model(cfg, 55, "cuda:0")
    m2, _ = build_model(cfg, 55, "cuda:0")
    cond = synth.conditioning(55, "cond", 2, 9, cfg["d_model"]).to("cuda:0")
    e1, e2 = m1.engine(1), m2.engine(1)
    alone = m2.generate(cond, max_new_tokens=12, sampling_params=GREEDY).cpu()
    assert e2.lib.zn_decode_path(e2.h) == 1
    seen = {}

    def nested(frame, step, max_steps):
        if step == 3:
            seen["codes"] = m2.generate(cond, max_new_tokens=12, sampling_params=GREEDY).cpu()
            seen["path"] = e2.lib.zn_decode_path(e2.h)
        return True
    out1 = m1.generate(cond, max_new_tokens=12, sampling_params=GREEDY, callback=nested).cpu()
    assert e1.lib.zn_decode_path(e1.h) == 1, "the first generation keeps the persistent kernels"
    assert seen["path"] == 0, "the generation begun meanwhile must run the launches path"
    assert torch.equal(seen["codes"][..., :8], alone[..., :8])       # d 512: the two paths cut fc2's K differently (one ulp of a logit late in the run)
    assert torch.equal(out1, alone)
    again = m2.generate(cond, max_new_tokens=12, sampling_params=GREEDY).cpu()
    assert e2.lib.zn_decode_path(e2.h) == 1 and torch.equal(again, alone)


def test_handoff_tags_restart_before_they_can_wrap():
    """The hand-off tags are 32-bit and advance by n_layer per decode step (about 41 hours of continuous batch-1 decoding).  Between two
    generations, long before a wrap, zn_gen_begin restarts the epoch at 1 over zeroed granule buffers; zn_debug_tune(14, 7) makes the next
    zn_gen_begin take that branch: the generation after it must be unaffected (same codes, same logits, no stale tag accepted)."""
    cfg, seed = synth.FULL_CFG, 1234
    model, _ = build_model(cfg, seed, "cuda:0")
    eng = model.engine(1)
    cond = synth.conditioning(seed, "cond", 2, 24, cfg["d_model"])
    a, la, _ = _run(model, cond, 40)
    assert eng.lib.zn_decode_path_detail(eng.h) == 2
    eng.call("zn_debug_tune", 14, 7)
    b, lb, _ = _run(model, cond, 40)
    assert torch.equal(a, b) and torch.equal(la.view(torch.int32), lb.view(torch.int32))
    c, lc, _ = _run(model, cond, 40)                       # and the epoch counts on from 1 afterwards
    assert torch.equal(a, c) and torch.equal(la.view(torch.int32), lc.view(torch.int32))


def test_a_reported_handoff_timeout_is_survived(capfd, monkeypatch):
    """A bounded hand-off wait that gives up voids the generation and demotes the handle to the launches path.  Under the test suite's
    setting (ZONOS_HIP_NO_TIMEOUT_RETRY=1, tests/conftest.py) `Zonos.generate` raises; without it - and when the caller has not seen any
    frame yet - it repeats the generation on the launches path, says so on stderr and counts it.  zn_debug_tune(14, 9) sets the sticky
    timeout word for the next generation (every wait gives up at once).  The demotion is visible (zn_get_counters) and temporary: four
    clean generations on the launches path, or zn_debug_tune(8, 1), re-arm the persistent kernels."""
    cfg, seed = synth.FULL_CFG, 1234
    model, _ = build_model(cfg, seed, "cuda:0")
    eng = model.engine(1)
    cond = synth.conditioning(seed, "cond", 2, 24, cfg["d_model"]).to("cuda:0")
    eng.call("zn_debug_eos_bias", float("-inf"))
    try:
        ref = model.generate(cond, max_new_tokens=40, sampling_params=GREEDY).cpu()
        assert eng.lib.zn_decode_path_detail(eng.h) == 2 and eng.counters()["handoff_timeouts"] == 0
        eng.call("zn_debug_tune", 14, 9)                        # the suite's setting: the timeout is an error
        with pytest.raises(Exception, match="hand-off wait"):
            model.generate(cond, max_new_tokens=40, sampling_params=GREEDY)
        c = eng.counters()
        assert c["handoff_timeouts"] == 1 and c["demoted"] == 1 and model.handoff_counters()["repeated_generations"] == 0
        eng.call("zn_debug_tune", 8, 1)                         # re-arm at once
        assert eng.counters()["demoted"] == 0 and eng.counters()["rearms"] == 1
        monkeypatch.delenv("ZONOS_HIP_NO_TIMEOUT_RETRY")        # production behaviour: repeat once, loudly
        eng.call("zn_debug_tune", 14, 9)
        out = model.generate(cond, max_new_tokens=40, sampling_params=GREEDY).cpu()
        err = capfd.readouterr().err
        assert "hand-off wait" in err and "repeating the generation" in err
        assert torch.equal(out, ref)
        assert eng.lib.zn_decode_path(eng.h) == 0, "after a reported timeout the handle runs the launches path"
        c = eng.counters()
        assert c["handoff_timeouts"] == 2 and c["demoted"] == 1 and c["fallback_generations"] == 1 and model.handoff_counters()["repeated_generations"] == 1
        for k in range(3):                                      # clean generations 2 .. 4 on the launches path (the repeat was the first)
            assert torch.equal(model.generate(cond, max_new_tokens=40, sampling_params=GREEDY).cpu(), ref)
            assert eng.lib.zn_decode_path(eng.h) == 0
        again = model.generate(cond, max_new_tokens=40, sampling_params=GREEDY).cpu()     # the fifth: re-armed
        c = eng.counters()
        assert torch.equal(again, ref) and eng.lib.zn_decode_path_detail(eng.h) == 2 and c["demoted"] == 0 and c["rearms"] == 2 and c["fallback_generations"] == 4
        eng.call("zn_debug_tune", 14, 9)                        # with a callback the caller has seen frames: the error is raised, not hidden
        with pytest.raises(Exception, match="hand-off wait"):
            model.generate(cond, max_new_tokens=40, sampling_params=GREEDY, callback=lambda f, s_, m: True)
        eng.call("zn_debug_tune", 8, 1)
        again = model.generate(cond, max_new_tokens=40, sampling_params=GREEDY).cpu()
        assert torch.equal(again, ref) and eng.lib.zn_decode_path_detail(eng.h) == 2
    finally:
        eng.call("zn_debug_tune", 8, 1)
        eng.call("zn_debug_eos_bias", 0.0)


def test_two_concurrent_requests_on_one_model():
    """The reference serves two requests per model at a time (utilities/app_constants.py:18).  Two threads call generate() on ONE model
    at once: the second finds the model's engine busy and runs on a second handle over the same weights (launches path: the first holds
    the device's persistent-kernel tenancy); both get the codes a solo run gives."""
    import threading
    cfg = synth.CHAIN_CFG
    model, _ = build_model(cfg, 55, "cuda:0")
    conds = [synth.conditioning(55 + i, "cond", 2, 9, cfg["d_model"]).to("cuda:0") for i in range(2)]
    solo = [model.generate(c, max_new_tokens=120, sampling_params=GREEDY).cpu() for c in conds]
    assert model._spare is None
    outs, errs = [None, None], []
    gate = threading.Barrier(2)

    def work(i):
        try:
            torch.cuda.set_device(0)
            gate.wait()
            outs[i] = model.generate(conds[i], max_new_tokens=120, sampling_params=GREEDY).cpu()
        except Exception as e:                                  # surfaced below
            errs.append(e)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert model._spare is not None, "the second request did not get a handle of its own"
    for i in range(2):
        n = min(outs[i].shape[-1], solo[i].shape[-1])
        assert n >= 100 and torch.equal(outs[i][..., :8], solo[i][..., :8])    # d 512: chain and launches paths cut fc2's K differently
        assert (outs[i][..., :n] == solo[i][..., :n]).float().mean() > 0.9


def test_single_workgroup_sampler_matches_the_ticketed_sampler():
    """Batch 1: sample1_kernel (one workgroup: a wave per codebook, bookkeeping and the next embedding behind a barrier) against
    sample_kernel + its ticketed tail (zn_debug_tune(16, 2)): same codes and same logits over greedy decoding, temperature + min_p
    sampling with a repetition window, and a run that ends through the EOS bookkeeping (forced EOS: masks, remaining counters)."""
    cfg, seed = synth.FULL_CFG, 1234
    model, _ = build_model(cfg, seed, "cuda:0")
    eng = model.engine(1)
    cond = synth.conditioning(seed, "cond", 2, 24, cfg["d_model"]).to("cuda:0")
    cases = [({"temperature": 0.0}, None), ({"temperature": 1.0, "min_p": 0.1}, None),
             ({"temperature": 0.7, "min_p": 0.0, "repetition_penalty": 2.0, "repetition_penalty_window": 5}, None), ({"temperature": 0.0}, 21)]
    try:
        for sp, force in cases:
            outs = []
            for t16 in (2, 3):                              # 3: sample1_kernel also for the sampled cases (the default keeps it to greedy decoding)
                eng.call("zn_debug_tune", 16, t16)
                eng.call("zn_debug_eos_bias", float("-inf") if force is None else 0.0)
                eng.call("zn_debug_force_eos", -1 if force is None else force)
                tr = {"logits": []}
                o = model.generate(cond, max_new_tokens=64, sampling_params=sp, seed=77, _trace=tr)
                outs.append((o.cpu(), torch.stack(tr["logits"]).cpu()))
                o2 = model.generate(cond, max_new_tokens=64, sampling_params=sp, seed=77)       # 8-step graphs
                assert torch.equal(o2.cpu(), outs[-1][0]), (sp, t16)
            assert torch.equal(outs[0][0], outs[1][0]), sp
            assert torch.equal(outs[0][1].view(torch.int32), outs[1][1].view(torch.int32)), sp
            if force is not None:
                assert outs[0][0].shape[-1] < 64, "the forced EOS did not end the generation"
    finally:
        eng.call("zn_debug_tune", 16, 1)
        eng.call("zn_debug_force_eos", -1)
        eng.call("zn_debug_eos_bias", 0.0)


def test_a_device_pause_inside_the_attention_wait_is_survived(capfd):
    """zn_debug_tune(14, 11): every whole-step launch of the next generation stops all its waves for 30 ms in block 2 - the attention
    workgroups inside the wait their pacer measures, the streaming workgroups outside theirs - which is how a pause of the device
    (queue preemption) looks from inside the kernel.  The pacer must not turn the 30 ms "wait" into the next block's sleep (uncapped it
    did: the attention workgroups slept 22 ms while the streaming ones polled past the 20 ms / 4096-pass bound; the second record in
    profiles/r03_handoff_timeout_record.txt): same codes as an undisturbed run, no timeout reported, the handle stays on the kernel."""
    cfg, seed = synth.FULL_CFG, 1234
    model, _ = build_model(cfg, seed, "cuda:0")
    eng = model.engine(1)
    cond = synth.conditioning(seed, "cond", 2, 24, cfg["d_model"]).to("cuda:0")
    try:
        eng.call("zn_debug_eos_bias", float("-inf"))
        ref = model.generate(cond, max_new_tokens=12, sampling_params=GREEDY).cpu()
        eng.call("zn_debug_tune", 14, 11)
        t0 = time.perf_counter()
        out = model.generate(cond, max_new_tokens=12, sampling_params=GREEDY).cpu()
        dt = time.perf_counter() - t0
        assert eng.lib.zn_decode_path_detail(eng.h) == 2, "the generation left the whole-step kernel"
        again = model.generate(cond, max_new_tokens=12, sampling_params=GREEDY).cpu()
        assert eng.lib.zn_decode_path_detail(eng.h) == 2
    finally:
        eng.call("zn_debug_eos_bias", 0.0)
    err = capfd.readouterr().err
    assert "hand-off" not in err, err
    assert dt > 19 * 0.030, f"the pause hook did not run ({dt:.3f} s)"
    assert torch.equal(out, ref) and torch.equal(again, ref)
    # ... and the pause was OBSERVED: the attention workgroups' waits contained it, the launch left the longest one in the handle's
    # diagnostic words (zn_get_counters [6], [7]; tools/soak.py logs real pauses of the device this way)
    c = eng.counters()
    assert c["handoff_timeouts"] == 0 and c["longest_wait_us"] >= 29000 and c["waits_over_200us"] >= 8, c
    eng.call("zn_debug_tune", 14, 13)
    assert eng.counters()["longest_wait_us"] == 0
