# PMC passes of the whole-step kernel + the bench line + its kernel stats on the sources as they are (run by gpurun from the repo root).
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3r
mkdir -p $O
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python3 tools/pmc_kernel.py 6 > $O/pmc_f.txt 2>&1; echo "pmc f rc=$?"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python3 tools/pmc_kernel.py 6 > $O/pmc_w.txt 2>&1; echo "pmc w rc=$?"
python tools/pmc_summary.py stepkernel $O/pmc_f $O/pmc_w profiles/pmc_step_kernel.json | tail -1 | cut -c1-200; cp profiles/pmc_step_kernel.json $O/
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.log; echo "bench rc=$?"; cut -c1-200 $O/bench.json
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > $O/kt_bench.json 2> $O/kt.log; echo "kt rc=$?"
find $O -name "*kernel_trace.csv" -size +3M -delete
