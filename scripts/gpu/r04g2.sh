mkdir -p gpurun_out/r04g2
for d in 0 1 2 4 3 6 7; do
  HOIC_GEMM_DBG=$d timeout 120 python tools/gemm_bench.py --pipeline 3 --reps 9 --no-update --ops fwd,fwd_nostore,fwd_plain --out gpurun_out/r04g2/g_$d.json > /dev/null 2>&1
  python - $d <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r04g2/g_{sys.argv[1]}.json'))
print('dbg',sys.argv[1],' '.join(f"{g['op']}{g['layer']}:{g['f16x3_ms']:.3f}" for g in d['gemms']))
PY
done
