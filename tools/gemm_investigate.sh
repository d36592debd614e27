# development aid: what bounds the f16x3 GEMM main loop (run on the GPU box from the repo root)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
echo "== stride experiment (fwd_nostore: main loop only)"
python tools/gemm_bench.py --no-update --pipeline 3 --ops fwd_nostore --dims 2048x1024,2112x1024,2080x1024,1024x512,1056x512,640x2048,2048x2048 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: r=json.loads(l); print(r['K'],r['N'],r['op'],round(r['f16x3_ms'],3),round(r['mfma_frac_of_2500'],3))
    except Exception: pass
"
cd /tmp; export TMPDIR=/tmp
for p in 1 2 3; do
  case $p in
    1) C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA";;
    2) C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU";;
    3) C="GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum";;
  esac
  timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/gpmc$p -- python3 $R/tools/gemm_bench.py --no-update --pipeline 3 --ops fwd_nostore,fwd,bwd_weight --dims 2048x1024 --reps 3 > /tmp/gpmc$p.log 2>&1
  tail -2 /tmp/gpmc$p.log
done
python3 - <<'PY'
import csv,glob,collections
for p in (1,2,3):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"/tmp/gpmc{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:60]
            if "gemm" not in k: continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,c in acc.items():
        print(p,k,{n:round(sum(v)/len(v)) for n,v in c.items()}, "launches", max(len(v) for v in c.values()))
PY
for f in /tmp/gpmc1/*/*kernel_trace.csv; do python3 - "$f" <<'PY'
import csv,sys,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm" in r["Kernel_Name"]: d[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in d.items(): print("dur_us",k,round(sum(v)/len(v),1),len(v))
PY
done
