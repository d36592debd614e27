R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06u; mkdir -p $O
J='import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], round(d["value"]), "rollout", round(d["rollout_only_env_steps_per_s"]), "update_s", round(d["update_s_per_iteration"],5), "warmup", d["warmup"], d["warmup_requested"])'
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --other-configs 0 > $O/first.json 2>$O/err.txt || tail -5 $O/err.txt; python -c "$J" $O/first.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --other-configs 0 --min-warmup-iterations 1 > $O/second_min1.json 2>$O/err.txt || tail -5 $O/err.txt; python -c "$J" $O/second_min1.json
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --other-configs 0 > $O/third.json 2>$O/err.txt || tail -5 $O/err.txt; python -c "$J" $O/third.json
