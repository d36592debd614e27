R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06t; mkdir -p $O
timeout 600 python -m pytest tests/test_mlp.py -m gpu -q -x 2>&1 | tail -2
timeout 300 python tools/probe/update_copies.py 2>&1 | grep -E "ms  x" | head -5
J='import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], round(d["value"]), "rollout", round(d["rollout_only_env_steps_per_s"]), "update_s", round(d["update_s_per_iteration"],5))'
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --other-configs 0 > $O/b_$i.json 2>$O/err.txt || tail -5 $O/err.txt; python -c "$J" $O/b_$i.json; done
