R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05aj; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_trace -o t -- python3 $R/bench.py --steps 39 --warmup 13 --min-iterations 3 --no-cpu-baseline --other-configs 0 --update-streams 3 > /tmp/prof_trace.log 2>&1
python3 - <<'PY'
import csv, glob, gzip, os
f = sorted(glob.glob("/tmp/prof_trace/**/*kernel_trace.csv", recursive=True))[-1]
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r05aj", "trace3.csv.gz")
with gzip.open(out, "wt") as g:
    cols = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Stream_Id", "Queue_Id", "Grid_Size_X"]
    w = csv.writer(g); w.writerow(cols)
    for r in csv.DictReader(open(f)):
        w.writerow([r["Kernel_Name"][:80]] + [r.get(c, "") for c in cols[1:]])
PY
cd $R; python3 tools/seam_timeline.py gpurun_out/r05aj/trace3.csv.gz | head -12
