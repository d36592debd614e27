O=gpurun_out/r04o; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "early_forward or agent_iteration or forks" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
J='import json,sys; d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); print(sys.argv[1], round(d["value"]), "rollout", round(d["rollout_only_env_steps_per_s"]), "update_s", round(d["update_s_per_iteration"],5), "rollout_s", round(d["rollout_s_per_iteration"],5), "substep_ms", round(d["roofline"]["kernel_ms"],3))'
for i in 1 2; do for e in 0 1; do timeout 300 python bench.py --no-cpu-baseline --other-configs 0 --min-iterations 20 --early-forward $e > $O/bench_e${e}_$i.json 2>$O/err.txt; python -c "$J" $O/bench_e${e}_$i.json; done; done
