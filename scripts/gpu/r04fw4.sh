R=$PWD
cd /tmp; export TMPDIR=/tmp
for v in head new wide; do
  if [ $v = head ]; then export HOIC_LIB=$R/hoic_amd/libhoic_hip_head.so; unset HOIC_FWD_TILE32; elif [ $v = new ]; then unset HOIC_LIB; export HOIC_FWD_TILE32=1; else unset HOIC_LIB; unset HOIC_FWD_TILE32; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_$v -o p -- python3 $R/bench.py --steps 26 --warmup 13 --min-iterations 2 --no-cpu-baseline --other-configs 0 > /tmp/pf_$v.json 2>/tmp/pf_$v.log
  echo "== $v"; python3 - $v <<'PY'
import csv,glob,sys
f=glob.glob(f'/tmp/pf_{sys.argv[1]}/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'fwd_tiled' in r['Name'] or 'substep' in r['Name'] or 'zfilter' in r['Name'] or 'head_kernel' in r['Name'] or 'pack_tiled' in r['Name']:
        print(r['Name'][:48], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us  min', round(float(r['MinNs'])/1e3,1))
PY
done
