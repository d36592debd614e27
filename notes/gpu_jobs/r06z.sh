bash tools/run_measurements.sh 3 2>&1 | grep -v "^+" | tail -12
timeout 900 python tools/reward_curve.py --arms hip_fixed_f16x3 --seeds 5 --iters 100 --out gpurun_out/r06_reward_curve_box_hip_fixed_f16x3.json --tmp gpurun_out/r06_curves/box > gpurun_out/r06_curve_box.log 2>&1; tail -2 gpurun_out/r06_curve_box.log | cut -c1-200
