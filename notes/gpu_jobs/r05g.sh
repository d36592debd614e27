set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05g; mkdir -p $O
BID=$(python3 -c "from hoic_amd import lib; print(lib.build_id())"); echo $BID > $O/build_id.txt
J='import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], round(d["value"]), "rollout", round(d["rollout_only_env_steps_per_s"]), "update_s", round(d["update_s_per_iteration"],5), "substep_ms", round(d["roofline"]["kernel_ms"],3), "post_ms", round(d["roofline"]["poststep_kernel_ms"],3)); [print(" ", k, round(v.get("value",0)), v.get("kernel_ms"), v.get("error")) for k,v in d.get("other_configs",{}).items()]'
timeout 1500 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; rc=$?; tail -3 $O/pytest.txt
if [ $rc -ne 0 ]; then echo "GPU tests failed"; exit 1; fi
for i in 1 2; do timeout 400 python bench.py --no-cpu-baseline --other-configs 0 > $O/bench_box_$i.json 2>$O/bench_err.txt; python -c "$J" $O/bench_box_$i.json; done
HOIC_LIB=libhoic_hip_prev.so timeout 400 python bench.py --no-cpu-baseline --other-configs 0 > $O/bench_box_prevlib.json 2>>$O/bench_err.txt; python -c "$J" $O/bench_box_prevlib.json
timeout 400 python bench.py --no-cpu-baseline --other-configs 0 --fused-filter 0 > $O/bench_box_unfused.json 2>>$O/bench_err.txt; python -c "$J" $O/bench_box_unfused.json
timeout 400 python bench.py --no-cpu-baseline --other-configs 0 > $O/bench_box_3.json 2>>$O/bench_err.txt; python -c "$J" $O/bench_box_3.json
HOIC_LIB=libhoic_hip_prev.so timeout 400 python bench.py --no-cpu-baseline --other-configs 0 > $O/bench_box_prevlib2.json 2>>$O/bench_err.txt; python -c "$J" $O/bench_box_prevlib2.json
GPU_MAX_HW_QUEUES=8 timeout 400 python bench.py --groups 3 --no-cpu-baseline --other-configs 0 > $O/bench_box_3ranges_q8.json 2>>$O/bench_err.txt; python -c "$J" $O/bench_box_3ranges_q8.json
timeout 120 python tools/phase_timing.py 2048 box > $O/phase_box.txt 2>&1; grep -E "kernel ms|total cycles|pre-hsolve|kin:levels|mass_matrix|bias " $O/phase_box.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_trace -o t -- python3 $R/bench.py --steps 39 --warmup 13 --min-iterations 3 --no-cpu-baseline --other-configs 0 > /tmp/prof_trace.log 2>&1
python3 - <<'PY'
import csv, glob, gzip, os
f = sorted(glob.glob("/tmp/prof_trace/**/*kernel_trace.csv", recursive=True))[-1]
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r05g", "rollout_trace_2ranges.csv.gz")
with gzip.open(out, "wt") as g:
    cols = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Stream_Id", "Queue_Id", "Grid_Size_X"]
    w = csv.writer(g); w.writerow(cols)
    for r in csv.DictReader(open(f)):
        w.writerow([r["Kernel_Name"][:80]] + [r.get(c, "") for c in cols[1:]])
PY
cd $R; python3 tools/rollout_timeline.py $O/rollout_trace_2ranges.csv.gz > $O/rollout_timeline.txt 2>&1; head -9 $O/rollout_timeline.txt
