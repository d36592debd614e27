mkdir -p gpurun_out/r04curve
timeout 2000 python tools/reward_curve.py --arms hip_fixed_f16x3 --seeds 5 --iters 100 --out gpurun_out/r04curve/curve.json --tmp gpurun_out/r04curve/runs > gpurun_out/r04curve/log.txt 2>&1
tail -5 gpurun_out/r04curve/log.txt
python - <<'PY'
import json
d=json.load(open('gpurun_out/r04curve/curve.json'))
for e in d['bands']['hip_fixed_f16x3']['eval']:
    if e['iter'] in (0,10,20,30,50,75,100,99): print(e['iter'], round(e['reward_per_step_mean'],4), round(e['reward_per_step_std'],4), round(e['tracked_mean'],3), e['seeds'])
print('wall', d['wall_s'])
PY
