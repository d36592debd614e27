R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06zd; mkdir -p $O
J='import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], round(d["value"]), "rollout", round(d["rollout_only_env_steps_per_s"]), "update_s", round(d["update_s_per_iteration"],5))'
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --other-configs 0 > $O/base_$i.json 2>$O/err.txt || tail -5 $O/err.txt; python -c "$J" $O/base_$i.json
HIP_FORCE_DEV_KERNARG=1 timeout 300 python bench.py --no-cpu-baseline --other-configs 0 > $O/devkernarg_$i.json 2>$O/err.txt || tail -5 $O/err.txt; python -c "$J" $O/devkernarg_$i.json
HSA_ENABLE_SDMA=0 timeout 300 python bench.py --no-cpu-baseline --other-configs 0 > $O/nosdma_$i.json 2>$O/err.txt || tail -5 $O/err.txt; python -c "$J" $O/nosdma_$i.json
done
