set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05k; mkdir -p $O
J='import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[1], round(d["value"]), "n_gpus", d["n_gpus"], d["config"]["parallelism"], "rollout", round(d["rollout_only_env_steps_per_s"]), "update_s", round(d["update_s_per_iteration"],5), "traffic", d["roofline"]["traffic"], "valu frac_chip", d["roofline_valu"].get("frac_chip"))'
# the driver's two launch forms: plain, and one rank under torch.distributed.run with the RCCL calls exercised
( time timeout 900 python bench.py > $O/bench_driver_form.json 2> $O/bench_driver_form.err ) 2> $O/time_plain.txt; python -c "$J" $O/bench_driver_form.json; tail -3 $O/time_plain.txt
HOIC_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 130 --warmup 26 --no-cpu-baseline --other-configs 0 > $O/bench_rccl_one_rank.json 2> $O/bench_rccl.err; python -c "$J" $O/bench_rccl_one_rank.json; tail -2 $O/bench_rccl.err
timeout 300 python scripts/train_hand_mimic.py --cfg box_future5_light_add_geom --num_threads 32 --no_log --num_epoch 6 > $O/train_script.log 2>&1; tail -4 $O/train_script.log
