R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05ak; mkdir -p $O
J='import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], round(d["value"]), "rollout", round(d["rollout_only_env_steps_per_s"]), "update_s", round(d["update_s_per_iteration"],5))'
for i in 1 2 3; do
for m in 2 3; do
timeout 300 python bench.py --no-cpu-baseline --other-configs 0 --update-streams $m > $O/s${m}_$i.json 2>$O/err_$m.txt || tail -5 $O/err_$m.txt; python -c "$J" $O/s${m}_$i.json
done; done
