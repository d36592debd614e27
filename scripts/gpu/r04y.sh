set -x
mkdir -p gpurun_out/r04y
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
for o in box bottle banana; do
  timeout 300 python tools/phase_timing.py 2048 $o > gpurun_out/r04y/phase_$o.txt 2>&1
done
grep -h "kernel ms\|hs:\|total cycles" gpurun_out/r04y/phase_*.txt
for i in 1 2; do timeout 300 python bench.py > gpurun_out/r04y/bench_$i.json 2> gpurun_out/r04y/err.log || tail -20 gpurun_out/r04y/err.log; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04y/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d['value']), round(d['rollout_only_env_steps_per_s']), d['update_s_per_iteration'], d['roofline']['kernel_ms'], {k:(round(v['value']), v['kernel_ms']) for k,v in d['other_configs'].items()})
PY
