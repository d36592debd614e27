set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05n; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "mesh or obb or bottle or banana or Bottle or Banana or episode" > $O/pytest.log 2>&1; tail -4 $O/pytest.log
J='import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], round(d["value"]), "rollout", round(d["rollout_only_env_steps_per_s"]), "update_s", round(d["update_s_per_iteration"],5), "substep_ms", round(d["roofline"]["kernel_ms"],3))'
for o in banana bottle; do
for L in libhoic_hip.so libhoic_hip_prev.so; do
HOIC_LIB=$L timeout 300 python bench.py --obj $o --no-cpu-baseline --other-configs 0 --min-iterations 8 > $O/bench_${o}_$L.json 2>$O/err.txt; python -c "$J" $O/bench_${o}_$L.json
done
HOIC_LIB=libhoic_hip_timing.so timeout 120 python tools/phase_timing.py 2048 $o 2>&1 | grep -E "kernel ms|collision|total"
HOIC_LIB=libhoic_hip_timing_prev.so timeout 120 python tools/phase_timing.py 2048 $o 2>&1 | grep -E "kernel ms|collision|total"
done
