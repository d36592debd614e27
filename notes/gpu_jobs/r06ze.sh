R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06ze; mkdir -p $O
J='import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], round(d["value"]), "rollout", round(d["rollout_only_env_steps_per_s"]), "update_s", round(d["update_s_per_iteration"],5))'
run() { n=$1; shift; timeout 300 "$@" > $O/$n.json 2>$O/err.txt || tail -3 $O/err.txt; python -c "$J" $O/$n.json; }
B="python bench.py --no-cpu-baseline --other-configs 0"
run g2 $B
GPU_MAX_HW_QUEUES=8 run g2_q8 $B
GPU_MAX_HW_QUEUES=8 run g3_q8 $B --groups 3
GPU_MAX_HW_QUEUES=6 run g3_q6 $B --groups 3
run g3_q4 $B --groups 3
GPU_MAX_HW_QUEUES=8 run g4_q8 $B --groups 4
