R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
t0=$(date +%s.%N); python bench.py > gpurun_out/r05_bench_box_driver_form.json 2> gpurun_out/r05_bench_driver_form.err; t1=$(date +%s.%N); echo "wall $(echo "$t1 - $t0" | bc) s"; tail -3 gpurun_out/r05_bench_driver_form.err
python -c "
import json; d=json.load(open('gpurun_out/r05_bench_box_driver_form.json')); print(round(d['value']), d['roofline']['traffic'], d['roofline_valu']['frac'], d['roofline']['kernel_ms'], {k:round(v['value']) for k,v in d.get('other_configs',{}).items() if isinstance(v,dict) and 'value' in v})"
