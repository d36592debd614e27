O=$GRAFT_REPO_ROOT/gpurun_out/r04j; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr2 -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 39 --warmup 13 --min-iterations 3 --no-cpu-baseline --other-configs 0 > /tmp/tr2.log 2>&1
gzip -c /tmp/tr2/t_kernel_trace.csv > $O/trace_g2.csv.gz
python3 $GRAFT_REPO_ROOT/tools/rollout_timeline.py $O/trace_g2.csv.gz
