# Round-4 measurement batch, part A (run by gpurun from the repo root): GPU tests and the PMC passes of the whole-step kernel on the sources as they are.
set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4fa
mkdir -p $O
timeout -k 10 800 python -m pytest tests -x -q -s -m gpu > $O/gpu_tests.txt 2>&1; echo "tests rc=$?"; tail -2 $O/gpu_tests.txt
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python3 tools/pmc_kernel.py 6 > $O/pmc_f.txt 2>&1; echo "pmc f rc=$?"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python3 tools/pmc_kernel.py 6 > $O/pmc_w.txt 2>&1; echo "pmc w rc=$?"
python tools/pmc_summary.py stepkernel $O/pmc_f $O/pmc_w profiles/pmc_step_kernel.json | tail -1 | cut -c1-200; cp profiles/pmc_step_kernel.json $O/
find $O -name "*kernel_trace.csv" -size +3M -delete
