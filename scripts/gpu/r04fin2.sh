mkdir -p gpurun_out/r04fin2
for i in 1 2 3; do timeout 300 python bench.py > gpurun_out/r04fin2/bench_$i.json 2> gpurun_out/r04fin2/err.log || tail -20 gpurun_out/r04fin2/err.log; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04fin2/bench_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d['value']), round(d['rollout_only_env_steps_per_s']), d['update_s_per_iteration'], d['roofline']['kernel_ms'], d['roofline']['traffic'], d['roofline_valu'].get('frac'), d['roofline_valu'].get('frac_chip'), {k:(round(v['value']), v['kernel_ms']) for k,v in d['other_configs'].items()})
PY
