O=gpurun_out/r04g; mkdir -p $O
J='import json,sys; d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); print(sys.argv[1], round(d["value"]), "rollout", round(d["rollout_only_env_steps_per_s"]), "update_s", round(d["update_s_per_iteration"],5), "rollout_s", round(d["rollout_s_per_iteration"],5), "substep_ms", round(d["roofline"]["kernel_ms"],3), "post_ms", round(d["roofline"]["poststep_kernel_ms"],3))'
for ov in 0 1; do for i in 1 2; do timeout 300 python bench.py --overlap $ov --no-cpu-baseline > $O/bench_ov${ov}_$i.json 2>$O/err.txt; python -c "$J" $O/bench_ov${ov}_$i.json; done; done
GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --overlap 1 --no-cpu-baseline > $O/bench_ov1_q8.json 2>$O/err.txt; python -c "$J" $O/bench_ov1_q8.json
tail -3 $O/err.txt
