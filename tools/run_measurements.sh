mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/pytest_gpu.log
timeout 400 python bench.py > gpurun_out/r02_bench_box.json 2> gpurun_out/bench_box.err; tail -c 600 gpurun_out/r02_bench_box.json; echo
timeout 300 python bench.py --pretrain 60 --no-cpu-baseline > gpurun_out/r02_bench_box_tracking.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r02_bench_box_tracking.json')); print('tracking', d['value'], d['rollout_only_env_steps_per_s'], d['workload_stats'], d['roofline']['kernel_ms'])"
timeout 200 python tools/gemm_bench.py --pipeline 3 --reps 9 --out gpurun_out/r02_gemm_bench.json > gpurun_out/gemm_bench.log 2>&1; tail -2 gpurun_out/gemm_bench.log
timeout 200 python tools/gemm_bench.py --pipeline 3 --reps 9 --zero --no-update --out gpurun_out/r02_gemm_bench_zero.json > gpurun_out/gemm_bench_zero.log 2>&1
for o in bottle banana; do timeout 300 python bench.py --obj $o --no-cpu-baseline > gpurun_out/r02_bench_$o.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r02_bench_$o.json')); print('$o', d['value'], d['rollout_only_env_steps_per_s'], d['workload_stats'], d['roofline']['kernel_ms'])"; done
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o b -- python3 $R/bench.py --steps 52 --warmup 13 --no-cpu-baseline > $R/gpurun_out/r02_bench_box_under_rocprof.json 2>/tmp/prof_bench.log; python3 $R/tools/pmc_summary.py stats --dir /tmp/prof_bench --out $R/gpurun_out/r02_bench_kernel_stats.csv
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sim -o s -- python3 $R/tools/sim_only.py 4096 12 > /tmp/prof_sim.log 2>&1; python3 $R/tools/pmc_summary.py stats --dir /tmp/prof_sim --out $R/gpurun_out/r02_simonly_kernel_stats.csv
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d /tmp/pmc_a -- python3 $R/tools/sim_only.py 4096 6 > /tmp/pmc_a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pmc_b -- python3 $R/tools/sim_only.py 4096 6 > /tmp/pmc_b.log 2>&1; tail -2 /tmp/pmc_b.log
python3 $R/tools/pmc_summary.py counters --dir /tmp/pmc_a /tmp/pmc_b --kernel hoic_substep_kernel --envs 4096 --out $R/gpurun_out/r02_substep_sq_counters.json --command "rocprofv3 --kernel-trace --pmc <8 SQ counters per pass, two passes> -- python3 tools/sim_only.py 4096 6 (mean over launches 3..6, divided by 4096 = per wavefront = per env-step)"
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -- python3 $R/tools/sim_only.py 2048 12 > /tmp/pmc_f.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -- python3 $R/tools/sim_only.py 2048 12 > /tmp/pmc_w.log 2>&1
python3 $R/tools/pmc_summary.py traffic --fetch-dir /tmp/pmc_f --write-dir /tmp/pmc_w --kernel hoic_substep_kernel --envs 2048 --obj box --out $R/gpurun_out/r02_hbm_traffic.json --command "separate passes: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE -- python3 tools/sim_only.py 2048 12"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_upd -o u -- python3 $R/tools/update_only.py f16x3 4 > /tmp/prof_upd.log 2>&1; python3 $R/tools/pmc_summary.py stats --dir /tmp/prof_upd --out $R/gpurun_out/r02_update_kernel_stats.csv
cd $R; ls -la gpurun_out | tail -15
