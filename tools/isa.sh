#!/bin/bash
# development aid: gfx950 assembly of hoic_mlp.hip -> /tmp/mlp.s, one kernel (mangled-name regex $1) -> /tmp/k.s, loop summary
set -e
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=fast -S --cuda-device-only -o /tmp/mlp.s /root/repo/hoic_amd/csrc/hoic_mlp.hip 2>&1 | grep -E "error" -A3 || true
awk -v pat="$1" 'index($0, pat":")==1 {p=1} p{print} p&&/s_endpgm/{exit}' /tmp/mlp.s > /tmp/k.s
echo "lines $(wc -l < /tmp/k.s) scratch $(grep -c scratch_ /tmp/k.s || true)"
grep -E "\.vgpr_count|scratch_en|\.private_segment_fixed_size" /tmp/mlp.s | head -0
