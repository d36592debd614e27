set -x
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
timeout 300 python -m pytest tests/test_mlp.py -m gpu -q -x -k "fork_merge or one_launch" > gpurun_out/r05_pytest_quick.log 2>&1 && timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r05_pytest_gpu_gate.log 2>&1; rc=$?; tail -3 gpurun_out/r05_pytest_gpu_gate.log
if [ $rc -ne 0 ]; then echo "GPU tests failed: no measurements"; exit 1; fi
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.log 2>&1; tail -2 gpurun_out/r05_smoke.log
bash tools/run_measurements.sh 1 2>&1 | grep -v "^+" | tail -40
bash tools/run_measurements.sh 2 2>&1 | grep -v "^+" | tail -30
