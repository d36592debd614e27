R=$GRAFT_REPO_ROOT
BID=$(python3 -c "from hoic_amd import lib; print(lib.build_id())"); echo "build id $BID"
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r04_pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r04_pytest_gpu.log
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d /tmp/pmc_a -- python3 $R/tools/sim_only.py 4096 6 > /tmp/pmc_a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pmc_b -- python3 $R/tools/sim_only.py 4096 6 > /tmp/pmc_b.log 2>&1
python3 $R/tools/pmc_summary.py counters --dir /tmp/pmc_a /tmp/pmc_b --kernel hoic_substep_kernel --envs 4096 --build-id $BID --out $R/gpurun_out/profiles_r04/r04_substep_sq_counters.json --command "rocprofv3 --kernel-trace --pmc <8 SQ counters per pass, two passes> -- python3 tools/sim_only.py 4096 6 (mean over launches 3..6, divided by 4096 = per wavefront = per env-step)" > /dev/null
for o in box banana; do
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f_$o -- python3 $R/tools/sim_only.py 2048 12 $o > /tmp/pmc_f.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w_$o -- python3 $R/tools/sim_only.py 2048 12 $o > /tmp/pmc_w.log 2>&1
python3 $R/tools/pmc_summary.py traffic --fetch-dir /tmp/pmc_f_$o --write-dir /tmp/pmc_w_$o --kernel hoic_substep_kernel --envs 2048 --obj $o --build-id $BID --out $R/gpurun_out/profiles_r04/r04_hbm_traffic_$o.json --command "separate passes: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE -- python3 tools/sim_only.py 2048 12 $o" > /dev/null
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -o b -- python3 $R/bench.py --steps 52 --warmup 13 --min-iterations 4 --no-cpu-baseline --other-configs 0 > $R/gpurun_out/profiles_r04/r04_bench_box_under_rocprof.json 2>/tmp/prof_bench.log; python3 $R/tools/pmc_summary.py stats --dir /tmp/prof_bench --out $R/gpurun_out/profiles_r04/r04_bench_kernel_stats.csv
cd $R
cp gpurun_out/profiles_r04/r04_substep_sq_counters.json gpurun_out/profiles_r04/r04_hbm_traffic_*.json profiles/
for i in 1 2 3; do timeout 400 python bench.py > gpurun_out/profiles_r04/r04_bench_box_$i.json 2> gpurun_out/r04_bench_box.err; python3 -c "
import json; d=json.loads(open('gpurun_out/profiles_r04/r04_bench_box_$i.json').read().strip().split('\n')[-1]); print(round(d['value']), 'rollout', round(d['rollout_only_env_steps_per_s']), 'upd', round(d['update_s_per_iteration']*1e3,2), 'k_ms', round(d['roofline']['kernel_ms'],3), 'traffic', d['roofline']['traffic'], 'valu frac', d['roofline_valu'].get('frac'), {k:(round(v['value']), round(v['rollout_only_env_steps_per_s']), round(v['kernel_ms'],2)) for k,v in d['other_configs'].items()})"; done
