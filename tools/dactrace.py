"""Per-launch table of one DAC decode from a rocprofv3 kernel trace (csv): the last decode's conv launches in order.
    python3 tools/dactrace.py <dir with *_kernel_trace.csv> [launches per decode=30]"""
import csv
import glob
import sys

d = sys.argv[1]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 30
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = [r for r in csv.DictReader(open(f)) if "dac_conv" in r["Kernel_Name"] or "dac_final" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-per:]
tot = 0
for r in last:
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += dur
    g = [int(r[k]) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")]
    wg = [int(r[k]) for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z")]
    print(f"{r['Kernel_Name'][:40]:40s} grid {g[0] // wg[0]:5d} x {g[1] // wg[1]:3d} x {g[2] // wg[2]:2d}  {dur:9.1f} us")
print(f"total {tot / 1e3:.3f} ms over {len(last)} launches")
