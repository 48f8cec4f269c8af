mkdir -p gpurun_out/r3l
timeout -k 10 300 python -m pytest tests/test_gpu_dac.py -x -q -s -m gpu > gpurun_out/r3l/dac_tests.txt 2>&1; echo "tests rc=$?"; grep -E "RMS|passed|failed|Error" gpurun_out/r3l/dac_tests.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "" k1rb2 nfast; do
  export ZONOS_HIP_LIB_VARIANT=$v; [ -z "$v" ] && unset ZONOS_HIP_LIB_VARIANT
  rm -rf /tmp/dt; timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/dt -- python3 $R/tools/dacprof.py 10 1 2 > $R/gpurun_out/r3l/dacprof_$v.log 2>&1
  echo "== variant [$v]"; python3 $R/tools/dactrace.py /tmp/dt 30 > $R/gpurun_out/r3l/dactrace_$v.txt; tail -1 $R/gpurun_out/r3l/dactrace_$v.txt
done
unset ZONOS_HIP_LIB_VARIANT
cd $R && timeout -k 10 120 python tools/dacbench.py 10 2>&1 | tail -2
