bash tools/run_measurements.sh 2 2>&1 | grep -v "^+" | tail -30
