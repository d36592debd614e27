set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05a; mkdir -p $O
# 1. what the chip sustains with the vendor library at the f16x3 kernels' MFMA flop counts (idle GPU), our kernels beside it
timeout 300 python tools/gemm_bench.py --calibrate --reps 9 --out $O/gemm_calibrate.json > $O/calibrate.log 2>&1; tail -3 $O/calibrate.log
timeout 300 python tools/gemm_bench.py --pipeline 3 --reps 9 --no-update --out $O/gemm_bench_own.json > $O/gemm_own.log 2>&1; tail -3 $O/gemm_own.log
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/cal_pmc -- python3 $R/tools/gemm_bench.py --calibrate --reps 3 > /tmp/cal_pmc.log 2>&1
python3 - > $O/calibrate_clock.txt <<'PY'
import csv,glob,collections
acc=collections.defaultdict(list); dur=collections.defaultdict(list)
for f in glob.glob("/tmp/cal_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"]=="GRBM_GUI_ACTIVE": acc[(r["Kernel_Name"][:90], r.get("Grid_Size"))].append(float(r["Counter_Value"]))
for f in glob.glob("/tmp/cal_pmc/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[(r["Kernel_Name"][:90], r.get("Grid_Size"))].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k in acc:
    if len(acc[k])<3 or k not in dur: continue
    g=sum(acc[k])/len(acc[k]); d=sum(dur[k])/len(dur[k])
    if d<100: continue
    print(k, "launches",len(acc[k]),"dur_us",round(d,1),"GRBM_GUI_ACTIVE",round(g),"GHz(8 XCD)",round(g/8/d/1e3,3))
PY
cat $O/calibrate_clock.txt
cd $R
# 2. matched-filter control on the headline sampler (5 seeds x 100), then the CPU arms of the mesh objects beside each other
timeout 600 python tools/reward_curve.py --arms hip_fixed_f16x3_frozen --seeds 5 --iters 100 --out $O/curve_box_frozen.json --tmp $O/runs_box > $O/curve_box.log 2>&1; tail -3 $O/curve_box.log
(timeout 2400 python tools/reward_curve.py --arms cpu_fixed --seeds 3 --iters 60 --obj bottle --workers 40 --time-limit 2300 --out $O/curve_bottle_cpu.json --tmp $O/runs_bottle > $O/curve_bottle.log 2>&1 &)
timeout 2400 python tools/reward_curve.py --arms cpu_fixed --seeds 3 --iters 60 --obj banana --workers 40 --time-limit 2300 --out $O/curve_banana_cpu.json --tmp $O/runs_banana > $O/curve_banana.log 2>&1
sleep 20
tail -3 $O/curve_bottle.log $O/curve_banana.log
