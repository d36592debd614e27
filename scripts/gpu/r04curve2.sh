mkdir -p gpurun_out/r04curve2
for o in bottle banana; do
timeout 1000 python tools/reward_curve.py --arms hip_fixed_f16x3 --obj $o --seeds 3 --iters 60 --out gpurun_out/r04curve2/curve_$o.json --tmp gpurun_out/r04curve2/runs_$o > gpurun_out/r04curve2/log_$o.txt 2>&1
tail -2 gpurun_out/r04curve2/log_$o.txt
python - $o <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r04curve2/curve_{sys.argv[1]}.json'))
for e in d['bands']['hip_fixed_f16x3']['eval']:
    if e['iter'] % 10 == 0: print(sys.argv[1], e['iter'], round(e['reward_per_step_mean'],4), round(e['reward_per_step_std'],4), round(e['tracked_mean'],3), e['seeds'])
print('wall', d['wall_s'])
PY
done
