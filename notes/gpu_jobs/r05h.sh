set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05h; mkdir -p $O
BID=$(python3 -c "from hoic_amd import lib; print(lib.build_id())"); echo $BID > $O/build_id.txt
J='import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], round(d["value"]), "rollout", round(d["rollout_only_env_steps_per_s"]), "update_s", round(d["update_s_per_iteration"],5), "substep_ms", round(d["roofline"]["kernel_ms"],3), "post_ms", round(d["roofline"]["poststep_kernel_ms"],3))'
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; rc=$?; tail -3 $O/pytest.txt
if [ $rc -ne 0 ]; then echo "GPU tests failed"; exit 1; fi
for i in 1 2; do
timeout 400 python bench.py --no-cpu-baseline --other-configs 0 > $O/bench_box_fused_$i.json 2>$O/bench_err.txt; python -c "$J" $O/bench_box_fused_$i.json
timeout 400 python bench.py --no-cpu-baseline --other-configs 0 --fused-filter 0 > $O/bench_box_unfused_$i.json 2>>$O/bench_err.txt; python -c "$J" $O/bench_box_unfused_$i.json
done
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_trace -o t -- python3 $R/bench.py --steps 39 --warmup 13 --min-iterations 3 --no-cpu-baseline --other-configs 0 > /tmp/prof_trace.log 2>&1
python3 - <<'PY'
import csv, glob, gzip, os
f = sorted(glob.glob("/tmp/prof_trace/**/*kernel_trace.csv", recursive=True))[-1]
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r05h", "rollout_trace_2ranges.csv.gz")
with gzip.open(out, "wt") as g:
    cols = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Stream_Id", "Queue_Id", "Grid_Size_X"]
    w = csv.writer(g); w.writerow(cols)
    for r in csv.DictReader(open(f)):
        w.writerow([r["Kernel_Name"][:80]] + [r.get(c, "") for c in cols[1:]])
PY
cd $R; python3 tools/rollout_timeline.py $O/rollout_trace_2ranges.csv.gz > $O/rollout_timeline.txt 2>&1; head -9 $O/rollout_timeline.txt
