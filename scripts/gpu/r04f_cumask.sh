O=gpurun_out/r04f; mkdir -p $O
J='import json,sys; d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); print(sys.argv[1], round(d["value"]), "rollout", round(d["rollout_only_env_steps_per_s"]), "update_s", round(d["update_s_per_iteration"],5), "substep_ms", round(d["roofline"]["kernel_ms"],3), "post_ms", round(d["roofline"]["poststep_kernel_ms"],3))'
export GPU_MAX_HW_QUEUES=16
for g in 2 3 4; do for r in 0 16 32 64; do timeout 300 python bench.py --groups $g --reserve-cus $r --no-cpu-baseline > $O/bench_g${g}_r$r.json 2>$O/err.txt; python -c "$J" $O/bench_g${g}_r$r.json; done; done
tail -3 $O/err.txt
