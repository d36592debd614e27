#!/bin/bash
# helper for this build container: write a GPU job script from stdin into notes/gpu_jobs/<name>.sh and run it on an MI355X box
# through gpurun, always from the repository root.   usage: notes/gpu_jobs/run.sh <name> [timeout_s] < script
set -e
cd /root/repo
name=$1; to=${2:-1800}
cat > notes/gpu_jobs/$name.sh
test -s notes/gpu_jobs/$name.sh
exec /usr/local/graft/bin/gpurun --timeout $to -- "bash notes/gpu_jobs/$name.sh"
