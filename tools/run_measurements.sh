# Round-6 measurement pass (run on the GPU box from the repo root: gpurun -- 'bash tools/run_measurements.sh [part]').
# part 1: tests + bench lines; part 2: rocprofv3 kernel statistics and counter passes; part 3: GEMM counters + vendor-library
# calibration with its clock; part 4: reward curves on the final library.  Results land in
# gpurun_out/r06_*; the ones quoted in DESIGN.md are copied to profiles/.
PART=${1:-1}
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
BID=$(python3 -c "from hoic_amd import lib; print(lib.build_id())")
J='import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], round(d["value"]), "rollout", round(d["rollout_only_env_steps_per_s"]), "update_s", round(d["update_s_per_iteration"],5), "substep_ms", round(d["roofline"]["kernel_ms"],3), d["workload_stats"])'
if [ "$PART" = 1 ]; then
for i in 1 2 3; do timeout 600 python bench.py > gpurun_out/r06_bench_box_$i.json 2> gpurun_out/r06_bench_box.err; python -c "$J" gpurun_out/r06_bench_box_$i.json; done
timeout 300 python bench.py --pretrain 60 --no-cpu-baseline --min-iterations 10 > gpurun_out/r06_bench_box_tracking.json 2>/dev/null; python -c "$J" gpurun_out/r06_bench_box_tracking.json
timeout 400 python bench.py --workload closed-grasp --pretrain 100 --no-cpu-baseline --min-iterations 10 > gpurun_out/r06_bench_box_closed_grasp.json 2>/dev/null; python -c "$J" gpurun_out/r06_bench_box_closed_grasp.json
for o in bottle banana; do timeout 300 python bench.py --obj $o --no-cpu-baseline --min-iterations 10 > gpurun_out/r06_bench_$o.json 2>/dev/null; python -c "$J" gpurun_out/r06_bench_$o.json; done
timeout 300 python bench.py --sample-mode episodes --envs 32 --no-cpu-baseline > gpurun_out/r06_bench_box_episodes32.json 2>/dev/null; python -c "$J" gpurun_out/r06_bench_box_episodes32.json
timeout 300 python bench.py --sample-mode episodes --envs 256 --no-cpu-baseline > gpurun_out/r06_bench_box_episodes256.json 2>/dev/null; python -c "$J" gpurun_out/r06_bench_box_episodes256.json
timeout 300 python bench.py --sample-mode episodes --envs 512 --no-cpu-baseline > gpurun_out/r06_bench_box_episodes512.json 2>/dev/null; python -c "$J" gpurun_out/r06_bench_box_episodes512.json
HOIC_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --no-cpu-baseline --other-configs 0 > gpurun_out/r06_bench_box_rccl_one_rank.json 2>/dev/null; python -c "$J" gpurun_out/r06_bench_box_rccl_one_rank.json
timeout 200 python tools/gemm_bench.py --pipeline 3 --reps 9 --out gpurun_out/r06_gemm_bench.json > gpurun_out/gemm_bench.log 2>&1; tail -2 gpurun_out/gemm_bench.log
timeout 200 python tools/gemm_bench.py --pipeline 3 --reps 9 --zero --no-update --out gpurun_out/r06_gemm_bench_zero.json > gpurun_out/gemm_bench_zero.log 2>&1
for o in box bottle banana; do timeout 120 python tools/phase_timing.py 2048 $o > gpurun_out/r06_phase_$o.log 2>&1; done
fi
if [ "$PART" = 2 ]; then
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o b -- python3 $R/bench.py --steps 52 --warmup 13 --min-iterations 4 --no-cpu-baseline --other-configs 0 > $R/gpurun_out/r06_bench_box_under_rocprof.json 2>/tmp/prof_bench.log; python3 $R/tools/pmc_summary.py stats --dir /tmp/prof_bench --out $R/gpurun_out/r06_bench_kernel_stats.csv
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sim -o s -- python3 $R/tools/sim_only.py 4096 12 > /tmp/prof_sim.log 2>&1; python3 $R/tools/pmc_summary.py stats --dir /tmp/prof_sim --out $R/gpurun_out/r06_simonly_kernel_stats.csv
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d /tmp/pmc_a -- python3 $R/tools/sim_only.py 4096 6 > /tmp/pmc_a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pmc_b -- python3 $R/tools/sim_only.py 4096 6 > /tmp/pmc_b.log 2>&1; tail -2 /tmp/pmc_b.log
python3 $R/tools/pmc_summary.py counters --dir /tmp/pmc_a /tmp/pmc_b --kernel hoic_substep_kernel --envs 4096 --build-id $BID --out $R/gpurun_out/r06_substep_sq_counters.json --command "rocprofv3 --kernel-trace --pmc <8 SQ counters per pass, two passes> -- python3 tools/sim_only.py 4096 6 (mean over launches 3..6, divided by 4096 = per wavefront = per env-step)"
for o in box banana; do
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f_$o -- python3 $R/tools/sim_only.py 2048 12 $o > /tmp/pmc_f.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w_$o -- python3 $R/tools/sim_only.py 2048 12 $o > /tmp/pmc_w.log 2>&1
python3 $R/tools/pmc_summary.py traffic --fetch-dir /tmp/pmc_f_$o --write-dir /tmp/pmc_w_$o --kernel hoic_substep_kernel --envs 2048 --obj $o --build-id $BID --out $R/gpurun_out/r06_hbm_traffic_$o.json --command "separate passes: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE -- python3 tools/sim_only.py 2048 12 $o"
done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_upd -o u -- python3 $R/tools/update_only.py f16x3 4 > /tmp/prof_upd.log 2>&1; python3 $R/tools/pmc_summary.py stats --dir /tmp/prof_upd --out $R/gpurun_out/r06_update_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_trace -o t -- python3 $R/bench.py --steps 39 --warmup 13 --min-iterations 3 --no-cpu-baseline --other-configs 0 > /tmp/prof_trace.log 2>&1
python3 - <<'PY'
import csv, glob, gzip, os
f = sorted(glob.glob("/tmp/prof_trace/**/*kernel_trace.csv", recursive=True))[-1]
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r06_rollout_trace_2ranges.csv.gz")
with gzip.open(out, "wt") as g:
    cols = ["Kernel_Name", "Start_Timestamp", "End_Timestamp", "Stream_Id", "Queue_Id", "Grid_Size_X"]
    w = csv.writer(g); w.writerow(cols)
    for r in csv.DictReader(open(f)):
        w.writerow([r["Kernel_Name"][:80]] + [r.get(c, "") for c in cols[1:]])
print("wrote", out)
PY
cd $R; python3 tools/rollout_timeline.py gpurun_out/r06_rollout_trace_2ranges.csv.gz > gpurun_out/r06_rollout_timeline.txt 2>&1; tail -12 gpurun_out/r06_rollout_timeline.txt
python3 tools/seam_timeline.py gpurun_out/r06_rollout_trace_2ranges.csv.gz > gpurun_out/r06_update_seams.txt 2>&1; head -9 gpurun_out/r06_update_seams.txt
ls -la gpurun_out | grep r06_ | tail -12
fi
if [ "$PART" = 3 ]; then
bash tools/gemm_investigate.sh > gpurun_out/r06_gemm_pmc.txt 2>&1; tail -15 gpurun_out/r06_gemm_pmc.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/cal_pmc -- python3 $R/tools/gemm_bench.py --calibrate --reps 3 > /tmp/cal_pmc.log 2>&1
python3 $R/tools/pmc_summary.py clock --dir /tmp/cal_pmc --out $R/gpurun_out/r06_gemm_calibrate_clock.json
cd $R
fi
if [ "$PART" = 4 ]; then
timeout 600 python tools/reward_curve.py --arms hip_fixed_f16x3 --seeds 5 --iters 100 --out gpurun_out/r06_reward_curve_hip_fixed_f16x3.json --tmp gpurun_out/r06_curves/box > gpurun_out/r06_curve_box.log 2>&1; tail -2 gpurun_out/r06_curve_box.log
for o in bottle banana; do
timeout 400 python tools/reward_curve.py --arms hip_fixed_f16x3,hip_fixed_f16x3_frozen --obj $o --seeds 3 --iters 60 --out gpurun_out/r06_reward_curve_${o}_hip.json --tmp gpurun_out/r06_curves/$o > gpurun_out/r06_curve_$o.log 2>&1; tail -3 gpurun_out/r06_curve_$o.log
done
fi
