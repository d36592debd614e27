#!/usr/bin/env python3
"""Merge per-run result files of tools/reward_curve.py (several invocations, `--tmp` directories; a `.partial` file stands
in for a run that was cut by a time limit) into one JSON with per-arm bands.

    python3 tools/reward_curve_merge.py --out profiles/r02_reward_curve.json gpurun_out/rc2 gpurun_out/rc3 \
        [--log-json profiles/r02a_reward_curve_run1_partial.json:cpu_fixed,hip_fixed_long]

`--log-json FILE:arm,arm` adds the runs of the named arms from a file recovered from a run's log (evaluation points only).
A later directory overrides an earlier one for the same (arm, seed) when it holds more iterations."""
import argparse
import glob
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="*")
    ap.add_argument("--log-json", action="append", default=[])
    ap.add_argument("--out", required=True)
    ap.add_argument("--note", default="")
    args = ap.parse_args()
    runs = {}
    for d in args.dirs:
        files = sorted(glob.glob(os.path.join(d, "*.json"))) + sorted(glob.glob(os.path.join(d, "*.json.partial")))
        for f in files:
            if f.endswith(".partial") and os.path.exists(f[:-len(".partial")]):
                continue
            r = json.load(open(f))
            r["source"] = f
            for e in r["eval"]:
                e.get("on_hip", {}).pop("per_seq_len", None)
            k = (r["arm"], r["seed"])
            if k not in runs or len(r["eval"]) > len(runs[k]["eval"]):
                runs[k] = r
    for spec in args.log_json:
        f, arms = spec.split(":")
        for r in json.load(open(f))["runs"]:
            if r["arm"] in arms.split(","):
                k = (r["arm"], r["seed"])
                r = dict(r, source=f, curve=r.get("curve_every_10", []), from_log=True)
                if k not in runs or len(r["eval"]) > len(runs[k]["eval"]):
                    runs[k] = r
    runs = [runs[k] for k in sorted(runs)]
    bands = {}
    for arm in sorted({r["arm"] for r in runs}):
        rs = [r for r in runs if r["arm"] == arm]
        its = sorted({e["iter"] for r in rs for e in r["eval"]})
        rows = []
        for it in its:
            rp = np.array([e["on_hip"]["reward_per_step"] for r in rs for e in r["eval"] if e["iter"] == it])
            pc = np.array([e["on_hip"]["mean_percent"] for r in rs for e in r["eval"] if e["iter"] == it])
            rows.append({"iter": it, "seeds": int(len(rp)), "reward_per_step_mean": float(rp.mean()), "reward_per_step_std": float(rp.std(ddof=1)) if len(rp) > 1 else 0.0,
                         "tracked_mean": float(pc.mean()), "tracked_std": float(pc.std(ddof=1)) if len(pc) > 1 else 0.0})
        bands[arm] = {"eval": rows, "seeds": sorted(r["seed"] for r in rs), "iterations_reached": [max(e["iter"] for e in r["eval"]) for r in rs]}
    out = {"what": "deterministic (mean-action) episodes from frame 0 of all 17 sequences on the HIP simulator: reward per step and tracked "
                   "fraction every 5 PPO iterations, mean +- std over seeds; the same PPOLearner / schedule / synthetic motions in every arm "
                   "(tools/reward_curve.py)",
           "arms": {"cpu_episodes": "float64 CPU-oracle envs, whole episodes per sampler thread (reference-shaped)",
                    "cpu_fixed": "float64 CPU-oracle envs, the batched sampler's fixed-horizon scheme (4096 envs x 13 steps, value bootstrap)",
                    "hip_fixed": "HIP simulator, fixed-horizon batches (4096 envs x 13 steps, value bootstrap): product default",
                    "hip_episodes": "HIP simulator, whole-episode batches (sample_mode='episodes')",
                    "hip_fixed_long": "HIP simulator, fixed horizon with 1024 envs x 49 steps",
                    "hip_fixed_f16x3": "hip_fixed with the update's GEMMs on the f16x3 matrix-core path (bench.py's default update)"},
           "note": args.note, "bands": bands, "runs": runs}
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    for arm, b in bands.items():
        last = b["eval"][-1]
        print(f"{arm:15s} seeds {b['seeds']} reached {b['iterations_reached']}  iter {last['iter']}: {last['reward_per_step_mean']:.4f} +- {last['reward_per_step_std']:.4f}")


if __name__ == "__main__":
    main()
