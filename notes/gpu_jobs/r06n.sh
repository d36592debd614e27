R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06n; mkdir -p $O
J='import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], round(d["value"]), "rollout", round(d["rollout_only_env_steps_per_s"]), "update_s", round(d["update_s_per_iteration"],5), d.get("workload_stats"))'
for e in 32 256 512 1024; do
timeout 300 python bench.py --sample-mode episodes --envs $e --no-cpu-baseline > $O/ep$e.json 2>$O/err_$e.txt || tail -8 $O/err_$e.txt; python -c "$J" $O/ep$e.json
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "episodes or whole_episode or reference_flags or entry_script" 2>&1 | tail -3
