for l in libhoic_hip.so libhoic_o1.so libhoic_o2.so libhoic_hip.so; do echo "== $l"; HOIC_LIB=$l python tools/probe/epi_variants.py 2>&1 | grep -E "none|P\+gout " ; done
for l in libhoic_hip.so libhoic_o1.so libhoic_o2.so; do echo "== $l"; HOIC_LIB=$l timeout 200 python tools/gemm_bench.py --reps 9 --no-update --ops bwd_weight,bwd_data 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        g=json.loads(l); print(g['layer'],g['op'],round(g['f16x3_ms'],4))"; done
HOIC_LIB=libhoic_o1.so timeout 300 python -m pytest tests/test_mlp.py -m gpu -q -x 2>&1 | tail -2
