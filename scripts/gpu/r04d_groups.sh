set -x
O=gpurun_out/r04d; mkdir -p $O
J='import json,sys; d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); print(sys.argv[1], round(d["value"]), "rollout", round(d["rollout_only_env_steps_per_s"]), "update_s", round(d["update_s_per_iteration"],5), "substep_ms", round(d["roofline"]["kernel_ms"],3), "post_ms", round(d["roofline"]["poststep_kernel_ms"],3))'
for q in 8 16; do for g in 2 3 4; do GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --groups $g --no-cpu-baseline > $O/bench_q${q}_g$g.json 2>$O/err.txt; python -c "$J" $O/bench_q${q}_g$g.json; done; done
for g in 3 4; do GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --groups $g --async-reward 0 --no-cpu-baseline > $O/bench_q8_noasync_g$g.json 2>$O/err.txt; python -c "$J" $O/bench_q8_noasync_g$g.json; done
cd /tmp; export TMPDIR=/tmp
GPU_MAX_HW_QUEUES=8 timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr3 -o t -- python3 $GRAFT_REPO_ROOT/bench.py --groups 3 --steps 26 --warmup 13 --min-iterations 2 --no-cpu-baseline > /tmp/tr3.log 2>&1
ls -la /tmp/tr3/*; F=$(ls /tmp/tr3/*/*kernel_trace.csv /tmp/tr3/*kernel_trace.csv 2>/dev/null | head -1); gzip -c $F > $GRAFT_REPO_ROOT/$O/trace_g3_q8.csv.gz; ls -la $GRAFT_REPO_ROOT/$O/
