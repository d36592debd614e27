set -x
mkdir -p gpurun_out/r04s
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_dist_gpu.py -x -q -m gpu -k "agent or loop or rollout or curve or whole or optimize or fork" 2>&1 | tail -8
for ra in 0 1 0 1; do
  timeout 300 python bench.py --run-ahead $ra --other-configs 0 > gpurun_out/r04s/bench_ra${ra}_$RANDOM.json 2> gpurun_out/r04s/err.log || tail -20 gpurun_out/r04s/err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04s/bench_ra*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d['value']), round(d['rollout_only_env_steps_per_s']), d['update_s_per_iteration'], d['rollout_s_per_iteration'], d.get('host_s_per_iteration'), d['avg_c_reward'])
PY
