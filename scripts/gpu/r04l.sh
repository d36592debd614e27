O=gpurun_out/r04l; mkdir -p $O
for o in box; do timeout 120 python tools/phase_timing.py 2048 $o > $O/phase_$o.log 2>&1; tail -27 $O/phase_$o.log; done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
J='import json,sys; d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); print(sys.argv[1], round(d["value"]), "rollout", round(d["rollout_only_env_steps_per_s"]), "update_s", round(d["update_s_per_iteration"],5), "substep_ms", round(d["roofline"]["kernel_ms"],3), "post_ms", round(d["roofline"]["poststep_kernel_ms"],3))'
timeout 300 python bench.py --no-cpu-baseline --other-configs 0 --min-iterations 20 > $O/bench.json 2>$O/err.txt; python -c "$J" $O/bench.json
