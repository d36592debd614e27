R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05t; mkdir -p $O
J='import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], round(d["value"]), "rollout", round(d["rollout_only_env_steps_per_s"]), "update_s", round(d["update_s_per_iteration"],5), "substep_ms", round(d["roofline"]["kernel_ms"],3))'
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --other-configs 0 > $O/bench_new_$i.json 2>$O/err.txt; python -c "$J" $O/bench_new_$i.json
timeout 300 python bench.py --no-cpu-baseline --other-configs 0 --side-stream 0 --prepack 0 --pack-in-rollout 0 > $O/bench_old_$i.json 2>>$O/err.txt; python -c "$J" $O/bench_old_$i.json
done
timeout 300 python bench.py --no-cpu-baseline --other-configs 0 --prepack 0 > $O/bench_no_prepack.json 2>>$O/err.txt; python -c "$J" $O/bench_no_prepack.json
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_trace -o t -- python3 $R/bench.py --steps 39 --warmup 13 --min-iterations 3 --no-cpu-baseline --other-configs 0 > /tmp/prof_trace.log 2>&1
python3 - <<'PY'
import csv, glob, gzip, os
f = sorted(glob.glob("/tmp/prof_trace/**/*kernel_trace.csv", recursive=True))[-1]
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r05t", "trace.csv.gz")
with gzip.open(out, "wt") as g:
    cols = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Stream_Id", "Queue_Id", "Grid_Size_X"]
    w = csv.writer(g); w.writerow(cols)
    for r in csv.DictReader(open(f)):
        w.writerow([r["Kernel_Name"][:80]] + [r.get(c, "") for c in cols[1:]])
print("wrote", out)
PY
