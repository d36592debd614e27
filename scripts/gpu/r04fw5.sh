mkdir -p gpurun_out/r04fw5
timeout 600 python -m pytest tests/test_mlp.py -x -q -m gpu -k tiled 2>&1 | tail -2
for v in head new head new head new; do
  if [ $v = head ]; then export HOIC_LIB=$PWD/hoic_amd/libhoic_hip_head.so; else unset HOIC_LIB; fi
  timeout 300 python bench.py --other-configs 0 --no-cpu-baseline > gpurun_out/r04fw5/bench_${v}_$RANDOM.json 2> gpurun_out/r04fw5/err.log || tail -5 gpurun_out/r04fw5/err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04fw5/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d['value']), round(d['rollout_only_env_steps_per_s']), d['update_s_per_iteration'], d['roofline']['kernel_ms'])
PY
