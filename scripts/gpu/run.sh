#!/bin/bash
# helper for this build container: write a GPU job script from stdin into scripts/gpu/<name>.sh and run it on an MI355X box
# through gpurun, always from the repository root.   usage: scripts/gpu/run.sh <name> [timeout_s] < script
set -e
cd /root/repo
name=$1; to=${2:-1800}
cat > scripts/gpu/$name.sh
test -s scripts/gpu/$name.sh
exec /usr/local/graft/bin/gpurun --timeout $to -- "bash scripts/gpu/$name.sh"
