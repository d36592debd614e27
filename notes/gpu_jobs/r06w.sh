R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06w; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d /tmp/pmc_fwd -- python3 $R/tools/fwd_bench.py 2048 > /tmp/pmc_fwd.log 2>&1; tail -2 /tmp/pmc_fwd.log
timeout 200 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d /tmp/pmc_fwd2 -- python3 $R/tools/fwd_bench.py 2048 > /tmp/pmc_fwd2.log 2>&1; tail -1 /tmp/pmc_fwd2.log
python3 - <<'PY'
import csv, glob, collections
for d in ("/tmp/pmc_fwd", "/tmp/pmc_fwd2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r.get("Kernel_Name", "")
            if "fwd_tiled" not in n: continue
            key = (n[:40], r.get("Grid_Size") or r.get("Grid_Size_X"))
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(d, k, {c: round(sum(x[3:]) / max(len(x[3:]), 1)) for c, x in v.items()})
PY
