mkdir -p gpurun_out/r04fw
timeout 600 python -m pytest tests/test_mlp.py -x -q -m gpu 2>&1 | tail -3
HOIC_FWD_TILE32=1 timeout 600 python -m pytest tests/test_mlp.py -x -q -m gpu -k tiled 2>&1 | tail -2
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export HOIC_FWD_TILE32=1; else unset HOIC_FWD_TILE32; fi
  timeout 300 python bench.py --other-configs 0 --no-cpu-baseline > gpurun_out/r04fw/bench_t32_${v}_$RANDOM.json 2> gpurun_out/r04fw/err.log || tail -5 gpurun_out/r04fw/err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04fw/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d['value']), round(d['rollout_only_env_steps_per_s']), d['update_s_per_iteration'], d['roofline']['kernel_ms'])
PY
