timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/drv.json 2> gpurun_out/drv.err; echo rc $?
python3 - <<'PY'
import json
l=open('gpurun_out/drv.json').read().strip().splitlines()
print(len(l), 'line(s)')
d=json.loads(l[-1])
for k in ("metric","value","unit","n_gpus","steps","warmup","ms_per_step","higher_is_better","scaling","vs_baseline","dtype","data"): print(k, d[k])
print(d['config']['workload'][:100], d['config']['host_runs_ahead'])
print(d['roofline']); print(d['cpu_baseline']['value'], d['cpu_baseline']['kind'], d['cpu_baseline']['cores'])
PY
tail -3 gpurun_out/drv.err
