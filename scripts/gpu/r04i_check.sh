O=gpurun_out/r04i; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04i/bench.json').read().strip().split('\n')[-1])
print(round(d['value']), d['steps'], d['config']['timed_iterations'], 'rollout', round(d['rollout_only_env_steps_per_s']), 'upd', d['update_s_per_iteration'], 'k_ms', d['roofline']['kernel_ms'])
print(d['roofline_valu']); print(d['roofline']['traffic']); print(json.dumps(d.get('other_configs'), indent=1))
PY
timeout 100 python bench.py --gpus 2 --no-cpu-baseline; echo "exit code of --gpus 2 on a 1-GPU box: $?"
