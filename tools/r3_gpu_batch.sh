set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3e
mkdir -p $O
timeout -k 10 500 python -m pytest tests -x -q -s -m gpu > $O/gpu_tests.txt 2>&1; echo "tests rc=$?"; tail -3 $O/gpu_tests.txt
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.log; echo "bench rc=$?"; cat $O/bench.json
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/kt_bench.json 2> $O/kt.log; echo "kt rc=$?"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python3 tools/pmc_kernel.py 6 > $O/pmc_f.txt 2>&1; echo "pmc f rc=$?"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python3 tools/pmc_kernel.py 6 > $O/pmc_w.txt 2>&1; echo "pmc w rc=$?"
python tools/pmc_summary.py stepkernel $O/pmc_f $O/pmc_w $O/pmc_step_kernel.json | tail -2
find $O/kt -name "*kernel_stats.csv" | head -2
# keep the merged output small: drop the raw traces except stats and counters
find $O -name "*kernel_trace.csv" -size +3M -delete
du -sh $O
