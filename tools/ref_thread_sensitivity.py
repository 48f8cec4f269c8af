"""How much does the REFERENCE's own CPU path vary with the host's thread count?  (build container only: needs /root/reference)

The parity bar "bit-exact codebook indices under greedy decode" presumes that the reference CPU path is one function of its inputs.  At
the Zonos-v0.1 dimensions it is not: oneDNN cuts the bf16 GEMV contractions by thread count, so fp32 summation order - and with it one
bf16 ulp of a hidden value here and there - changes with `torch.set_num_threads`.  This script runs the real `Zonos.generate()` (the
recipe of tests/golden/make_golden.py: synthetic weights, seed 1234, 64 greedy steps; Gaussian and decisive-margin "peaky" heads) with
1, 2, 3, 5 and 8 threads and records, against the 8-thread run (the one the goldens were recorded with):

  free run        identical-frame prefix, token match
  teacher-forced  (the 8-thread token stream fed back) per-step max |dlogit|, argmax disagreements and the reference's own top-2 margin
                  (after the repetition penalty, the quantity the sampler's argmax sees) at every disagreeing (step, codebook) pair

Output: tests/golden/ref_thread_sensitivity.json (data only).  tests/test_gpu_decode.py takes its `decisive` margins and its bound on
the teacher-forced |dlogit| from that file.

    python tools/ref_thread_sensitivity.py [--steps 64] [--threads 1,2,3,5,8]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_golden as mg  # noqa: E402
from zonos_amd import synth  # noqa: E402


class Forced(mg.Recorder):
    """Recorder that keeps every call's logits and, when given a token stream, feeds THAT back instead of its own argmax."""
    def __init__(self, zm, forced=None):
        super().__init__(zm, keep_logits=None)
        self.forced = forced

    def __call__(self, logits, **kw):
        i = self.calls
        tok = super().__call__(logits, **kw)
        if self.forced is not None:
            tok = torch.from_numpy(self.forced[i].astype(np.int64)).reshape(tok.shape)
        return tok


def run(zm, model, cond, steps, forced=None):
    rec = Forced(zm, forced)
    with rec:
        out = model.generate(cond, max_new_tokens=steps, cfg_scale=2.0, batch_size=1, sampling_params={"temperature": 0.0},
                             disable_torch_compile=True)
    n = rec.calls
    return dict(out=out.numpy(), tokens=np.stack(rec.tokens), logits=np.stack([rec.logits[i] for i in range(n)]), margin=np.stack(rec.margin))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--threads", default="1,2,3,5,8")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "ref_thread_sensitivity.json"))
    args = ap.parse_args()
    threads = [int(t) for t in args.threads.split(",")]
    base_t = 8
    zm = mg.import_reference()
    cfg, seed = synth.FULL_CFG, 1234
    cond = synth.conditioning(seed, "cond", 2, 24, cfg["d_model"])
    report = dict(config="Zonos-v0.1-transformer dims, synthetic weights seed 1234, L_c 24, greedy, cfg_scale 2.0", steps=args.steps,
                  base_threads=base_t, torch=torch.__version__, cpu_count=os.cpu_count(), heads={})
    for head in ("gaussian", "peaky"):
        model, _ = mg.build_reference_model(zm, cfg, seed, peaky=(head == "peaky"))
        torch.set_num_threads(base_t)
        base = run(zm, model, cond, args.steps)
        again = run(zm, model, cond, args.steps)
        rows = []
        hd = dict(base_run_to_run_identical=bool(np.array_equal(base["tokens"], again["tokens"]) and np.array_equal(base["logits"], again["logits"])),
                  base_margin_quantiles={q: float(np.quantile(base["margin"], float(q))) for q in ("0.01", "0.05", "0.25", "0.5")})
        flips = []          # (threads, step, codebook, reference margin) of every teacher-forced argmax disagreement
        for t in threads:
            if t == base_t:
                continue
            torch.set_num_threads(t)
            free = run(zm, model, cond, args.steps)
            tf = run(zm, model, cond, args.steps, forced=base["tokens"])
            same_frame = (free["tokens"] == base["tokens"]).all(axis=(1, 2))
            prefix = int(np.argmin(same_frame)) if not same_frame.all() else int(same_frame.shape[0])
            fin = np.isfinite(base["logits"]) & np.isfinite(tf["logits"])
            d = np.where(fin, np.abs(tf["logits"] - base["logits"]), 0.0)
            per_step = d.reshape(d.shape[0], -1).max(axis=1)
            dis = tf["tokens"] != base["tokens"]                      # [calls, 1, 9] (tokens recorded before the override)
            margins = base["margin"][dis]
            for (s, b, q) in zip(*np.nonzero(dis)):
                flips.append((t, int(s), int(q), float(base["margin"][s, b, q])))
            rows.append(dict(threads=t, free_identical_frame_prefix=prefix, free_identical_frames=int(same_frame.sum()), calls=int(same_frame.shape[0]),
                             free_token_match=float((free["tokens"] == base["tokens"]).mean()),
                             free_output_equal=bool(free["out"].shape == base["out"].shape and np.array_equal(free["out"], base["out"])),
                             tf_logits_bit_equal=float((tf["logits"][fin] == base["logits"][fin]).mean()),
                             tf_max_abs_dlogit=float(per_step.max()), tf_max_abs_dlogit_step1=float(per_step[1]) if len(per_step) > 1 else 0.0,
                             tf_max_abs_dlogit_prefill=float(per_step[0]), tf_mean_step_max_abs_dlogit=float(per_step.mean()),
                             tf_argmax_disagreements=int(dis.sum()), tf_pairs=int(dis.size),
                             tf_largest_margin_of_a_disagreement=float(margins.max()) if margins.size else 0.0))
            print(head, rows[-1], flush=True)
        hd["runs"] = rows
        hd["flips"] = flips
        hd["largest_margin_of_any_disagreement"] = max([f[3] for f in flips], default=0.0)
        hd["largest_tf_abs_dlogit"] = max([r["tf_max_abs_dlogit"] for r in rows], default=0.0)
        report["heads"][head] = hd
        del model
    torch.set_num_threads(base_t)
    with open(args.out, "w") as f:
        json.dump(report, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
