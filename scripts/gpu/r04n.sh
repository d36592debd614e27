O=gpurun_out/r04n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
J='import json,sys; d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); print(sys.argv[1], round(d["value"]), "rollout", round(d["rollout_only_env_steps_per_s"]), "update_s", round(d["update_s_per_iteration"],5), "substep_ms", round(d["roofline"]["kernel_ms"],3), "post_ms", round(d["roofline"]["poststep_kernel_ms"],3))'
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --other-configs 0 --min-iterations 20 > $O/bench_$i.json 2>$O/err.txt; python -c "$J" $O/bench_$i.json; done
timeout 300 python bench.py --no-cpu-baseline --other-configs 0 --min-iterations 10 --workload closed-grasp --pretrain 100 > $O/bench_grasp.json 2>$O/err.txt; python -c "$J" $O/bench_grasp.json
HOIC_SHOW_DUR=1 timeout 100 python tools/sim_only.py 4096 12 | tail -3
