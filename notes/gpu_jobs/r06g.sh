R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06g; mkdir -p $O
J='import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], round(d["value"]), "rollout", round(d["rollout_only_env_steps_per_s"]), "update_s", round(d["update_s_per_iteration"],5), "substep_ms", round(d["roofline"]["kernel_ms"],3))'
run() { name=$1; shift; timeout 300 "$@" > $O/$name.json 2>$O/$name.err || tail -5 $O/$name.err; python -c "$J" $O/$name.json; }
B="python bench.py --no-cpu-baseline --other-configs 0"
run base $B
run overlap $B --overlap 1
for c in 32 64 96; do
HOIC_VALUE_CUS=$c run ov_v${c}_r${c} $B --overlap 1 --reserve-cus $c
done
HOIC_VALUE_CUS=64 run ov_v64_r0 $B --overlap 1
run base2 $B
