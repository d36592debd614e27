set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05r; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -6 $O/pytest_gpu.log
J='import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], round(d["value"]), "rollout", round(d["rollout_only_env_steps_per_s"]), "update_s", round(d["update_s_per_iteration"],5), "substep_ms", round(d["roofline"]["kernel_ms"],3))'
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --other-configs 0 > $O/bench_new_$i.json 2>$O/err.txt; python -c "$J" $O/bench_new_$i.json
timeout 300 python bench.py --no-cpu-baseline --other-configs 0 --side-stream 0 --prepack 0 --pack-in-rollout 0 > $O/bench_old_$i.json 2>>$O/err.txt; python -c "$J" $O/bench_old_$i.json
done
timeout 300 python bench.py --no-cpu-baseline --other-configs 0 --side-stream 0 > $O/bench_no_side.json 2>>$O/err.txt; python -c "$J" $O/bench_no_side.json
tail -3 $O/err.txt
