R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06j; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
J='import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], round(d["value"]), "rollout", round(d["rollout_only_env_steps_per_s"]), "with tail", round(d["rollout_with_tail_env_steps_per_s"]), "update_s", round(d["update_s_per_iteration"],5), "substep_ms", round(d["roofline"]["kernel_ms"],3))'
timeout 300 python bench.py --no-cpu-baseline --other-configs 0 > $O/bench.json 2>$O/bench.err || tail -5 $O/bench.err; python -c "$J" $O/bench.json
