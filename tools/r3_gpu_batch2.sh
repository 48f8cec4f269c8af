set -o pipefail
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3f
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_b8 -- python3 bench.py --batch-per-gpu 8 --steps 1 --warmup 1 --no-cpu-baseline > $O/bench_b8.json 2> $O/kt_b8.log; echo "b8 rc=$?"; cat $O/bench_b8.json | cut -c1-600
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_hy -- python3 tools/hybridbench.py 8 > $O/hybrid_b8.txt 2> $O/kt_hy.log; echo "hy rc=$?"; tail -3 $O/hybrid_b8.txt
timeout -k 10 200 python tools/hybridbench.py 1 > $O/hybrid_b1.txt 2>&1; tail -2 $O/hybrid_b1.txt
timeout -k 10 200 python tools/prefillbench.py > $O/prefill.txt 2>&1; tail -3 $O/prefill.txt
timeout -k 10 300 python tools/longform.py > $O/longform.txt 2>&1; tail -4 $O/longform.txt
find $O -name "*kernel_trace.csv" -size +3M -delete
find $O -name "*.csv" | head
