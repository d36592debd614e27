mkdir -p gpurun_out/r04fw3
for v in head new head new head new; do
  if [ $v = head ]; then export HOIC_LIB=$PWD/hoic_amd/libhoic_hip_head.so; unset HOIC_FWD_TILE32; else unset HOIC_LIB; export HOIC_FWD_TILE32=1; fi
  timeout 300 python bench.py --other-configs 0 --no-cpu-baseline > gpurun_out/r04fw3/bench_${v}_$RANDOM.json 2> gpurun_out/r04fw3/err.log || tail -5 gpurun_out/r04fw3/err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04fw3/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d['value']), round(d['rollout_only_env_steps_per_s']), d['update_s_per_iteration'], d['roofline']['kernel_ms'])
PY
