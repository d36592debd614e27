bash tools/run_measurements.sh 2
bash tools/run_measurements.sh 3
mkdir -p gpurun_out/r04_curves
# the headline arm on the final code: 5 seeds x 100 iterations, GPU to itself; then the fixed-horizon CPU arm's seeds 3..5
timeout 1500 python tools/reward_curve.py --arms "hip_fixed_f16x3" --seeds 5 --iters 100 --out gpurun_out/r04_curves/hip_fixed_f16x3.json --tmp gpurun_out/r04_curves/runs_hip > gpurun_out/r04_curves/hip.log 2>&1; tail -3 gpurun_out/r04_curves/hip.log
timeout 1700 python tools/reward_curve.py --arms "cpu_fixed" --seeds 3 --seed0 3 --iters 100 --workers 64 --out gpurun_out/r04_curves/cpu_fixed_seeds3to5.json --tmp gpurun_out/r04_curves/runs_cpu > gpurun_out/r04_curves/cpu.log 2>&1; tail -3 gpurun_out/r04_curves/cpu.log
