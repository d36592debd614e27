set -x
mkdir -p gpurun_out/r04w
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
for i in 1 2; do timeout 300 python bench.py > gpurun_out/r04w/bench_$i.json 2> gpurun_out/r04w/err.log || tail -20 gpurun_out/r04w/err.log; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04w/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d['value']), round(d['rollout_only_env_steps_per_s']), d['update_s_per_iteration'], d['roofline']['kernel_ms'], {k:(round(v['value']), v['kernel_ms']) for k,v in d['other_configs'].items()})
PY
