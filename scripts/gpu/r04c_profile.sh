set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04c; mkdir -p $O
for o in box; do timeout 120 python tools/phase_timing.py 2048 $o > $O/phase_$o.log 2>&1; tail -28 $O/phase_$o.log; done
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d /tmp/pmc_a -- python3 $R/tools/sim_only.py 4096 6 > /tmp/pmc_a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT --output-format csv -d /tmp/pmc_b -- python3 $R/tools/sim_only.py 4096 6 > /tmp/pmc_b.log 2>&1; tail -2 /tmp/pmc_b.log
python3 $R/tools/pmc_summary.py counters --dir /tmp/pmc_a /tmp/pmc_b --kernel hoic_substep_kernel --envs 4096 --out $O/substep_sq_counters.json --command "rocprofv3 --kernel-trace --pmc <8 SQ counters per pass, two passes> -- python3 tools/sim_only.py 4096 6 (mean over launches 3..6, divided by 4096 = per wavefront = per env-step)"
cat $O/substep_sq_counters.json
cd $R
# the fixed-horizon arm next to three concurrent GPU processes (round 3's final-check set-up), with and without round 3's fork order
timeout 900 python tools/reward_curve.py --arms "hip_episodes_frozen,hip_fixed_f16x3+racy" --seeds 3 --iters 30 --eval-every 10 --out $O/contended_racy.json --tmp $O/contended_racy_runs > $O/contended_racy.log 2>&1; tail -3 $O/contended_racy.log
timeout 900 python tools/reward_curve.py --arms "hip_episodes_frozen,hip_fixed_f16x3" --seeds 3 --iters 30 --eval-every 10 --out $O/contended_fixed.json --tmp $O/contended_fixed_runs > $O/contended_fixed.log 2>&1; tail -3 $O/contended_fixed.log
