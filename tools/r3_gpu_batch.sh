# Round-3 measurement batch (run by gpurun from the repo root): GPU tests, bench, kernel stats and PMC passes of the sources as they are.
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3v
mkdir -p $O
timeout -k 10 700 python -m pytest tests -x -q -s -m gpu > $O/gpu_tests.txt 2>&1; echo "tests rc=$?"; tail -2 $O/gpu_tests.txt
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python3 tools/pmc_kernel.py 6 > $O/pmc_f.txt 2>&1; echo "pmc f rc=$?"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python3 tools/pmc_kernel.py 6 > $O/pmc_w.txt 2>&1; echo "pmc w rc=$?"
python tools/pmc_summary.py stepkernel $O/pmc_f $O/pmc_w profiles/pmc_step_kernel.json | tail -1 | cut -c1-200; cp profiles/pmc_step_kernel.json $O/
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.log; echo "bench rc=$?"; cut -c1-200 $O/bench.json
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/kt_bench.json 2> $O/kt.log; echo "kt rc=$?"
timeout -k 10 200 python tools/stacksweep.py 400 > $O/sweep.txt 2>&1; grep -v amdgpu $O/sweep.txt | tail -4
timeout -k 10 300 python tools/stackbench.py 400 > $O/stackbench.txt 2>&1; grep -v amdgpu $O/stackbench.txt | tail -9
timeout -k 10 200 python bench.py --batch-per-gpu 8 --no-cpu-baseline > $O/bench_b8.json 2> $O/bench_b8.log; echo "b8 rc=$?"; cut -c1-200 $O/bench_b8.json
timeout -k 10 300 python tools/longform.py > $O/longform.txt 2>&1; tail -3 $O/longform.txt
timeout -k 10 120 python tools/dacbench.py 10 > $O/dacbench.txt 2>&1; tail -2 $O/dacbench.txt
timeout -k 10 120 python tools/dacbench.py 10 8 > $O/dacbench_b8.txt 2>&1; tail -1 $O/dacbench_b8.txt
timeout -k 10 200 python tools/prefillbench.py > $O/prefill.txt 2>&1; tail -1 $O/prefill.txt
find $O -name "*kernel_trace.csv" -size +3M -delete
