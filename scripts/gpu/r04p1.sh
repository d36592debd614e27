bash tools/run_measurements.sh 1
