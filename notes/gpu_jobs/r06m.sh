R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06m; mkdir -p $O
J='import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], round(d["value"]), "rollout", round(d["rollout_only_env_steps_per_s"]), "update_s", round(d["update_s_per_iteration"],5), "ar_ms", d.get("allreduce_exposed_ms_per_iteration"))'
export HOIC_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for i in 1 2; do
for s in 3 1 2; do
timeout 300 python bench.py --no-cpu-baseline --other-configs 0 --update-streams $s > $O/dist_s${s}_$i.json 2>$O/err.txt || tail -5 $O/err.txt; python -c "$J" $O/dist_s${s}_$i.json
done
timeout 300 python bench.py --no-cpu-baseline --other-configs 0 --side-stream 0 > $O/dist_noside_$i.json 2>$O/err.txt || tail -5 $O/err.txt; python -c "$J" $O/dist_noside_$i.json
done
