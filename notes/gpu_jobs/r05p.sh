set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05p; mkdir -p $O
timeout 900 python -m pytest tests/test_mlp.py tests/test_dist_gpu.py -m gpu -q > $O/pytest_mlp.log 2>&1; tail -5 $O/pytest_mlp.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "band or agent or sampler or rollout or update or gae or run_ahead or learner or ppo" > $O/pytest_agent.log 2>&1; tail -5 $O/pytest_agent.log
J='import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], round(d["value"]), "rollout", round(d["rollout_only_env_steps_per_s"]), "update_s", round(d["update_s_per_iteration"],5), "substep_ms", round(d["roofline"]["kernel_ms"],3))'
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --other-configs 0 > $O/bench_new_$i.json 2>$O/err.txt; python -c "$J" $O/bench_new_$i.json
timeout 300 python bench.py --no-cpu-baseline --other-configs 0 --side-stream 0 --prepack 0 --pack-in-rollout 0 > $O/bench_old_$i.json 2>>$O/err.txt; python -c "$J" $O/bench_old_$i.json
done
timeout 300 python bench.py --no-cpu-baseline --other-configs 0 --side-stream 1 --prepack 0 --pack-in-rollout 0 > $O/bench_side_only.json 2>>$O/err.txt; python -c "$J" $O/bench_side_only.json
timeout 300 python bench.py --no-cpu-baseline --other-configs 0 --side-stream 0 --prepack 1 --pack-in-rollout 0 > $O/bench_prepack_only.json 2>>$O/err.txt; python -c "$J" $O/bench_prepack_only.json
timeout 300 python bench.py --no-cpu-baseline --other-configs 0 --side-stream 0 --prepack 0 --pack-in-rollout 1 > $O/bench_packroll_only.json 2>>$O/err.txt; python -c "$J" $O/bench_packroll_only.json
tail -5 $O/err.txt
