# DAC measurements on the sources as they are (run by gpurun from the repo root): tests, per-launch trace, MFMA PMC, timings.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3u; mkdir -p $O; cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_dac.py -x -q -s -m gpu > $O/dac_tests.txt 2>&1; echo "tests rc=$?"; tail -1 $O/dac_tests.txt
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/dt; timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/dt -- python3 $R/tools/dacprof.py 10 1 2 > /dev/null 2>&1
python3 $R/tools/dactrace.py /tmp/dt 30 > $O/dactrace.txt; tail -1 $O/dactrace.txt
rm -rf /tmp/dp; timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/dp -- python3 $R/tools/dacprof.py 10 1 2 > /dev/null 2>&1
cd $R; python3 tools/pmc_summary.py dacmfma /tmp/dp $O/r03_pmc_dac_mfma.json | grep -E "void|util"
timeout -k 10 120 python tools/dacbench.py 10 2>&1 | grep code; timeout -k 10 120 python tools/dacbench.py 10 8 2>&1 | grep decode
