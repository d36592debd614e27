set -x
mkdir -p gpurun_out/r04a
./tools/probe/lds_occupancy.bin > gpurun_out/r04a/lds_occupancy.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04a/pytest.txt 2>&1; tail -5 gpurun_out/r04a/pytest.txt
# attribution of the headline arm's curve: 3 seeds x 30 iterations per switch
timeout 1500 python tools/reward_curve.py --arms "hip_fixed_f16x3,hip_fixed_f16x3+g1,hip_fixed_f16x3+torchfwd,hip_fixed_f16x3+autograd,hip_fixed_f16x3+racy,hip_fixed" --seeds 3 --iters 30 --out gpurun_out/r04a/bisect.json --tmp gpurun_out/r04a/bisect_runs > gpurun_out/r04a/bisect.log 2>&1
tail -8 gpurun_out/r04a/bisect.log
