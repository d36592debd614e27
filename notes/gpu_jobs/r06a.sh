R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06a; mkdir -p $O
J='import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], round(d["value"]), "rollout", round(d["rollout_only_env_steps_per_s"]), "update_s", round(d["update_s_per_iteration"],5), "substep_ms", round(d["roofline"]["kernel_ms"],3))'
timeout 300 python bench.py --no-cpu-baseline --other-configs 0 > $O/bench_base.json 2>$O/bench_base.err || tail -5 $O/bench_base.err; python -c "$J" $O/bench_base.json
timeout 200 python tools/gemm_bench.py --reps 9 --no-update --out $O/gemm_53248.json > $O/gemm_53248.log 2>&1; tail -16 $O/gemm_53248.log | cut -c1-150
timeout 200 python tools/gemm_bench.py --reps 9 --no-update --rows 106496 --out $O/gemm_106496.json > $O/gemm_106496.log 2>&1; tail -16 $O/gemm_106496.log | cut -c1-150
