R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06x; mkdir -p $O
timeout 600 python -m pytest tests/test_mlp.py -m gpu -q -x 2>&1 | tail -2
for i in 1 2 3; do python tools/fwd_bench.py 2048 2>&1 | tail -1; done
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d /tmp/pmc_fwd -- python3 $R/tools/fwd_bench.py 2048 > /tmp/pmc_fwd.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d /tmp/pmc_fwd2 -- python3 $R/tools/fwd_bench.py 2048 > /tmp/pmc_fwd2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("/tmp/pmc_fwd", "/tmp/pmc_fwd2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r.get("Kernel_Name", "")
            if "fwd_tiled" not in n: continue
            acc[(n[:40], r.get("Grid_Size") or r.get("Grid_Size_X"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k[1], {c: round(sum(x[3:]) / max(len(x[3:]), 1)) for c, x in v.items()})
PY
cd $R
J='import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1].split("/")[-1], round(d["value"]), "rollout", round(d["rollout_only_env_steps_per_s"]), "update_s", round(d["update_s_per_iteration"],5))'
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --other-configs 0 > $O/b_$i.json 2>$O/err.txt || tail -5 $O/err.txt; python -c "$J" $O/b_$i.json; done
