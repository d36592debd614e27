O=gpurun_out/r04k; mkdir -p $O
J='import json,sys; d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); print(sys.argv[1], round(d["value"]), "rollout", round(d["rollout_only_env_steps_per_s"]), "update_s", round(d["update_s_per_iteration"],5), "substep_ms", round(d["roofline"]["kernel_ms"],3), "post_ms", round(d["roofline"]["poststep_kernel_ms"],3))'
for i in 1 2; do for l in libhoic_hip_noprio.so libhoic_hip.so; do HOIC_LIB=$l timeout 300 python bench.py --no-cpu-baseline --other-configs 0 --min-iterations 20 > $O/bench_${l}_$i.json 2>$O/err.txt; python -c "$J" $O/bench_${l}_$i.json; done; done
for l in libhoic_hip_noprio.so libhoic_hip.so; do HOIC_LIB=$l HOIC_SHOW_DUR=1 timeout 100 python tools/sim_only.py 2048 12 | tail -3; done
