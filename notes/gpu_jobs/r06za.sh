R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
timeout 1200 python tools/reward_curve.py --arms hip_fixed_f16x3 --seeds 2 --iters 1000 --eval-every 100 --out gpurun_out/r06_reward_curve_box_1000_iterations.json --tmp gpurun_out/r06_curves/box1000 > gpurun_out/r06_curve_box1000.log 2>&1; tail -1 gpurun_out/r06_curve_box1000.log | cut -c1-200
for o in bottle banana; do
timeout 900 python tools/reward_curve.py --arms hip_fixed_f16x3 --obj $o --seeds 2 --iters 300 --eval-every 50 --out gpurun_out/r06_reward_curve_${o}_300_iterations.json --tmp gpurun_out/r06_curves/${o}300 > gpurun_out/r06_curve_${o}300.log 2>&1; tail -1 gpurun_out/r06_curve_${o}300.log | cut -c1-200
done
grep -il "error\|overflow\|Traceback" gpurun_out/r06_curve_*.log
