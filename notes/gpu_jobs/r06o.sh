bash tools/run_measurements.sh 1 2>&1 | grep -v "^+" | tail -16
