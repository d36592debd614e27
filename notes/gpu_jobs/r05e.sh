set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05e; mkdir -p $O
BID=$(python3 -c "from hoic_amd import lib; print(lib.build_id())"); echo $BID > $O/build_id.txt
J='import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], round(d["value"]), "rollout", round(d["rollout_only_env_steps_per_s"]), "update_s", round(d["update_s_per_iteration"],5), "substep_ms", round(d["roofline"]["kernel_ms"],3), "post_ms", round(d["roofline"]["poststep_kernel_ms"],3), d["workload_stats"]); [print(" ", k, round(v.get("value",0)), v.get("kernel_ms"), v.get("poststep_kernel_ms"), v.get("hand_object_contact_env_fraction"), v.get("error")) for k,v in d.get("other_configs",{}).items()]'
timeout 1800 python -m pytest tests -m gpu -q -s > $O/pytest.txt 2>&1; echo "pytest exit $?"; tail -6 $O/pytest.txt
grep -E "worst deviation|episode parity|envs with contacts|FAILED|Error" $O/pytest.txt | cut -c1-400 | head -40
timeout 400 python bench.py --no-cpu-baseline --other-configs 0 > $O/bench_box_1.json 2>$O/bench_err.txt; python -c "$J" $O/bench_box_1.json
timeout 400 python bench.py --no-cpu-baseline --other-configs 0 > $O/bench_box_2.json 2>>$O/bench_err.txt; python -c "$J" $O/bench_box_2.json
timeout 900 python bench.py --no-cpu-baseline > $O/bench_box_full.json 2>>$O/bench_err.txt; python -c "$J" $O/bench_box_full.json
for o in box bottle banana; do timeout 120 python tools/phase_timing.py 2048 $o > $O/phase_$o.txt 2>&1; done; grep -E "kernel ms|total cycles" $O/phase_*.txt
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d /tmp/pmc_a -- python3 $R/tools/sim_only.py 4096 6 > /tmp/pmc_a.log 2>&1
python3 $R/tools/pmc_summary.py counters --dir /tmp/pmc_a --kernel hoic_substep_kernel --envs 4096 --build-id $BID --out $O/substep_sq_counters_a.json --command "pass a" | cut -c1-700
