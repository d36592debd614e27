R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06e; mkdir -p $O
(sleep 8; for i in $(seq 1 40); do rocm-smi --showpower --showclocks --csv 2>/dev/null | tail -2 | head -1; sleep 0.2; done > $O/smi_update.csv) &
SM=$!
timeout 120 python tools/update_only.py f16x3 500 2>&1 | tail -1
wait $SM 2>/dev/null
cat $O/smi_update.csv | cut -d, -f6,10 | tr '\n' ' '
echo
(sleep 8; for i in $(seq 1 30); do rocm-smi --showpower --showclocks --csv 2>/dev/null | tail -2 | head -1; sleep 0.2; done > $O/smi_gemm.csv) &
SM=$!
python - <<'PY'
import os, sys, torch, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from hoic_amd import mlp as M
dev = torch.device("cuda"); Mr, K, N = 53248, 2048, 1024
g = torch.Generator(device=dev).manual_seed(0); t = M.ScaleTable(dev)
x = torch.randn(Mr, K, device=dev, generator=g); w = torch.randn(N, K, device=dev, generator=g) * 0.03; bias = torch.zeros(N, device=dev)
Xp, _ = M.pack(x, t, 0, Mr, K); Wp, _ = M.pack(w, t, 1, N, K)
with torch.no_grad(): t.exps[3] = 4
G = torch.empty(Mr, N, device=dev); Hp = torch.empty(Mr, 2 * N, dtype=torch.float16, device=dev)
t0 = time.time(); n = 0
while time.time() - t0 < 16:
    for _ in range(200): M.gemm(M.EPI_FWD, Mr, N, K, Xp, Wp, t, 0, 1, 3, bias=bias, P=Hp, gout=G)
    torch.cuda.synchronize(); n += 200
print("fwd1 sustained ms", (time.time() - t0) / n * 1e3)
PY
wait $SM 2>/dev/null
cat $O/smi_gemm.csv | cut -d, -f6,10 | tr '\n' ' '
