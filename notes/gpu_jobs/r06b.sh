R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06b; mkdir -p $O
timeout 900 python -m pytest tests/test_mlp.py -m gpu -q -x > $O/pytest_mlp.log 2>&1; tail -3 $O/pytest_mlp.log
timeout 200 python tools/gemm_bench.py --reps 9 --out $O/gemm.json > $O/gemm.log 2>&1; python - <<'PY'
import json,os
d=json.load(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r06b/gemm.json"))
for g in d["gemms"]: print(g["layer"], g["op"], round(g["f16x3_ms"],4))
print({k:v for k,v in d.items() if "update" in k})
PY
