R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05ai; mkdir -p $O
timeout 600 python -m pytest tests/test_mlp.py -m gpu -q > $O/pytest_mlp.log 2>&1; tail -2 $O/pytest_mlp.log
J='import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], round(d["value"]), "rollout", round(d["rollout_only_env_steps_per_s"]), "update_s", round(d["update_s_per_iteration"],5))'
for i in 1 2; do
for m in 2 3; do
timeout 300 python bench.py --no-cpu-baseline --other-configs 0 --update-streams $m > $O/s${m}_$i.json 2>$O/err_$m.txt || tail -5 $O/err_$m.txt; python -c "$J" $O/s${m}_$i.json
done; done
timeout 200 python tools/update_only.py f16x3 4 2>&1 | tail -2
