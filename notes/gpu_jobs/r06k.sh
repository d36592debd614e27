R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06k; mkdir -p $O
nproc
timeout 1500 python tools/reward_curve.py --arms cpu_fixed_online --seeds 5 --iters 100 --eval-every 5 --workers 32 --out $O/r06_reward_curve_box_cpu_fixed_online.json --tmp $O/curves --time-limit 1300 > $O/curve.log 2>&1; tail -4 $O/curve.log | cut -c1-250
python - <<'PY'
import json, os
d = json.load(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06k/r06_reward_curve_box_cpu_fixed_online.json"))
for arm, b in d["bands"].items():
    print(arm, [(e["iter"], round(e["reward_per_step_mean"], 4), round(e["reward_per_step_std"], 4), e["seeds"]) for e in b["eval"] if e["iter"] in (5, 10, 20, 30, 60, 100)])
print([ (r["seed"], round(r["wall_s"])) for r in d["runs"]])
PY
