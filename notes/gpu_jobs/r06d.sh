timeout 300 python -m pytest tests/test_mlp.py -m gpu -q -x 2>&1 | tail -2
for v in "HOIC_GEMM_PERSIST=0" "HOIC_GEMM_PERSIST=1" "HOIC_GEMM_PERSIST=1 HOIC_GEMM_STAGGER=1" "HOIC_GEMM_PERSIST=0 HOIC_GEMM_STAGGER=1"; do echo "== $v"; env $v python tools/probe/epi_variants.py 2>&1 | grep -E "none|P\+gout " ; done
