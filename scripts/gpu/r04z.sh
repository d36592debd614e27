mkdir -p gpurun_out/r04z
for cfg in "0 0" "4 0" "8 0" "16 0" "8 1" "8 2" "4 2"; do
  set -- $cfg
  HOIC_GEMM_STAGGER=$1 HOIC_GEMM_STAGGER_MODE=$2 timeout 120 python tools/gemm_bench.py --pipeline 3 --reps 9 --no-update --ops fwd,fwd_nostore,bwd_data --out gpurun_out/r04z/g_$1_$2.json > /dev/null 2>&1
  python - $1 $2 <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r04z/g_{sys.argv[1]}_{sys.argv[2]}.json'))
print('stagger',sys.argv[1],'mode',sys.argv[2],' '.join(f"{g['op']}{g['layer']}:{g['f16x3_ms']:.3f}" for g in d['gemms']))
PY
done
