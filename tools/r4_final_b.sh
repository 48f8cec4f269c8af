# Round-4 measurement batch, part B (run by gpurun from the repo root, after part A's profiles/pmc_step_kernel.json is in the tree): bench lines,
# kernel stats, long-form calls, hybrid, prefill, a short soak.
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4fb
mkdir -p $O
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.log; echo "bench rc=$?"; cut -c1-220 $O/bench.json
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/kt_bench.json 2> $O/kt.log; echo "kt rc=$?"; cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
timeout -k 10 200 python bench.py --batch-per-gpu 8 --no-cpu-baseline > $O/bench_b8.json 2> $O/bench_b8.log; echo "b8 rc=$?"; cut -c1-220 $O/bench_b8.json
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt8 -- python3 bench.py --batch-per-gpu 8 --steps 1 --warmup 1 --no-cpu-baseline > $O/kt_b8.json 2> $O/kt_b8.log; echo "kt8 rc=$?"; cp $(find /tmp/kt8 -name "*kernel_stats.csv" | head -1) $O/bench_b8_kernel_stats.csv
timeout -k 10 200 python tools/longform.py 0 2580 > $O/default30.txt 2>&1; tail -3 $O/default30.txt
timeout -k 10 300 python tools/longform.py > $O/longform.txt 2>&1; tail -3 $O/longform.txt
timeout -k 10 200 python tools/hybridbench.py 8 > $O/hybrid8.txt 2>&1; tail -2 $O/hybrid8.txt
timeout -k 10 200 python tools/hybridbench.py 1 > $O/hybrid1.txt 2>&1; tail -2 $O/hybrid1.txt
timeout -k 10 200 python tools/prefillbench.py > $O/prefill.txt 2>&1; tail -1 $O/prefill.txt
timeout -k 10 300 python tools/soak.py 100 > $O/soak.txt 2>&1; tail -4 $O/soak.txt
