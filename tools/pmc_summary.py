"""Summarise rocprofv3 --pmc passes (separate FETCH_SIZE and WRITE_SIZE runs, --output-format csv) into the JSON files bench.py and
DESIGN.md quote.

    python tools/pmc_summary.py stepkernel <fetch_dir> <write_dir> [out.json] # dominant kernel (tools/pmc_kernel.py 6) -> profiles/pmc_step_kernel.json
    python tools/pmc_summary.py kernel <fetch_dir> <write_dir> [out.json]     # the per-block chain launch (tools/pmc_kernel.py 5) -> profiles/pmc_chain.json
    python tools/pmc_summary.py step <fetch_dir> <write_dir> <decode_steps> [out.json]   # whole decode steps (tools/pmc_step.py)
    python tools/pmc_summary.py dacmfma <dir> [out.json]   # matrix-core utilisation of the DAC kernels (one pass: --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE)

Counter units are KB; FETCH_SIZE is doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md, HBM).
The kernel summary records the kernel's name and a hash of the kernel sources it was measured on: bench.py refuses it once
the sources change."""
import collections
import csv
import glob
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def read_counter(d, counter):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] == counter:
                    rows.append((r["Kernel_Name"], float(r["Counter_Value"]), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    if not rows:
        raise SystemExit(f"no {counter} rows under {d}")
    return rows


def kernel_summary(fetch_dir, write_dir, out, which=5):
    from bench import STEP_KERNEL_CTX, kernel_source_hash
    if which == 6:
        want, drop = "step_kernel<4, 2, 10, 5, 6, 6>", 8
        d, F, nqkv, nl, kvpos = 2048, 8192, 3072, 26, 2 * 512 * 2
        alg = int(nl * (d * d + 3 * F * d) * 2 + nl * nqkv * d * 2 + 9 * 1025 * d * 2 + 2 * nl * kvpos * STEP_KERNEL_CTX + 2 * nl * kvpos)      # (in_proj of block 0 inside the launch since the pre-block)
        what = f"whole decode step at {STEP_KERNEL_CTX} keys of context: 26 x (attention, out_proj x2, LayerNorm+fc1+SiLU-gate, fc2, next LayerNorm+in_proj+RoPE+KV append), norm_f + heads"
    else:
        want, drop = "chain_kernel<4, 1, 8, 4, 2>", 8
        alg = int((2048 * 2048 + 3 * 8192 * 2048 + 3072 * 2048) * 2)          # out_proj (once) + fc1 + fc2 + next in_proj, bf16
        what = "out_proj x2, LayerNorm+fc1+SiLU-gate, fc2, next LayerNorm+in_proj+RoPE+KV append"
    f = [v for n, v, *_ in read_counter(fetch_dir, "FETCH_SIZE") if want in n][drop:]     # warm-up launches dropped
    w = [v for n, v, *_ in read_counter(write_dir, "WRITE_SIZE") if want in n][drop:]
    fm, wm = sum(f) / len(f), sum(w) / len(w)
    rec = {"kernel_name": f"void {want}(ChainArgs)  ({what})", "kernel_source_sha256_16": kernel_source_hash(),
           "date": time.strftime("%Y-%m-%d"), "launches": len(f),
           "command": f"rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python3 tools/pmc_kernel.py {which}   (second pass: --pmc WRITE_SIZE)",
           "notes": f"counter units KB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads); {drop} warm-up launches dropped",
           "fetch_size_kb_mean": round(fm, 1), "write_size_kb_mean": round(wm, 1), "hbm_read_bytes_per_launch": int(2 * fm * 1024),
           "hbm_write_bytes_per_launch": int(wm * 1024), "traffic_bytes_per_launch": int(2 * fm * 1024 + wm * 1024), "algorithmic_bytes_per_launch": alg}
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))


def dac_mfma_summary(d, out):
    """SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs: utilisation = busy / (gui / 8 * 1024)."""
    busy, gui, ns = collections.defaultdict(float), collections.defaultdict(float), collections.defaultdict(float)
    cnt = collections.Counter()
    for n, v, t0, t1 in read_counter(d, "SQ_VALU_MFMA_BUSY_CYCLES"):
        if "dac_" in n:
            busy[n] += v; ns[n] += t1 - t0; cnt[n] += 1
    for n, v, *_ in read_counter(d, "GRBM_GUI_ACTIVE"):
        if "dac_" in n:
            gui[n] += v
    rec = {"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -- python3 tools/dacprof.py 10 1 2   (4 decodes of 10 s)",
           "interpretation": "SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs; MFMA pipe utilisation = busy / (gui / 8 * 1024); "
                             "the three-term kernels issue 6 bf16 MFMAs (32 cycles each) where the fp32 kernels issued 8 fp32 MFMAs (64 cycles each)",
           "date": time.strftime("%Y-%m-%d"), "kernels": {}}
    for n in sorted(busy):
        if gui[n] > 0 and busy[n] > 0:
            rec["kernels"][n.split("(")[0]] = {"launches": cnt[n], "total_ms": round(ns[n] / 1e6, 3), "SQ_VALU_MFMA_BUSY_CYCLES": busy[n], "GRBM_GUI_ACTIVE": gui[n],
                                               "mfma_utilisation": round(busy[n] / (gui[n] / 8 * 1024), 4)}
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


def step_summary(fetch_dir, write_dir, steps, out):
    fetch, write = collections.defaultdict(list), collections.defaultdict(list)
    for n, v, *_ in read_counter(fetch_dir, "FETCH_SIZE"):
        fetch[n].append(v)
    for n, v, *_ in read_counter(write_dir, "WRITE_SIZE"):
        write[n].append(v)
    decode = [n for n in fetch if any(k in n for k in ("gemv_kernel", "attn_", "embed_kernel", "sample_kernel", "frame_update_kernel", "chain_kernel", "step_kernel"))]
    per_kernel = {}
    total_r = total_w = 0.0
    for n in sorted(decode, key=lambda k: -sum(fetch[k])):
        r, w = 2 * sum(fetch[n]) * 1024, sum(write.get(n, [0.0])) * 1024
        total_r += r
        total_w += w
        per_kernel[n] = {"launches": len(fetch[n]), "hbm_read_bytes_per_launch": int(r / len(fetch[n])), "hbm_write_bytes_per_launch": int(w / max(1, len(write.get(n, [0]))))}
    alg = 3249278976
    rec = {"date": time.strftime("%Y-%m-%d"), "decode_steps": steps, "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python3 tools/pmc_step.py <steps>   (second pass: --pmc WRITE_SIZE)",
           "notes": "all launches of the decode-step kernels in the run (prefill of 25 positions uses other kernels and is excluded; the first-frame sample + frame update are included: < 0.01 %); FETCH_SIZE doubled (gfx950)",
           "hbm_read_bytes_per_step": int(total_r / steps), "hbm_write_bytes_per_step": int(total_w / steps),
           "traffic_bytes_per_step": int((total_r + total_w) / steps), "algorithmic_bytes_per_step": alg,
           "traffic_over_algorithmic": round((total_r + total_w) / steps / alg, 4), "per_kernel": per_kernel}
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in rec.items() if k != "per_kernel"}))
    for n, v in per_kernel.items():
        print(f"  {v['launches']:6d} x {v['hbm_read_bytes_per_launch'] / 1e6:9.3f} MB read {v['hbm_write_bytes_per_launch'] / 1e6:8.3f} MB written  {n[:100]}")


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[1] == "dacmfma":
    dac_mfma_summary(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "r03_pmc_dac_mfma.json"))
    sys.exit(0)
if __name__ == "__main__":
    if sys.argv[1] == "kernel":
        kernel_summary(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "profiles", "pmc_chain.json"))
    elif sys.argv[1] == "stepkernel":
        kernel_summary(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "profiles", "pmc_step_kernel.json"), which=6)
    else:
        step_summary(sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5] if len(sys.argv) > 5 else os.path.join(ROOT, "profiles", "r02_pmc_step.json"))
