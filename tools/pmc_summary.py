#!/usr/bin/env python3
"""Summaries of rocprofv3 passes for profiles/ (no GPU needed: reads the CSVs rocprofv3 wrote).

  counters  python3 tools/pmc_summary.py counters --dir gpurun_out/pmc_sq --kernel hoic_substep_kernel --envs 4096 \
                --out profiles/r02_substep_sq_counters.json [--skip 2] [--command "..."]
            mean over the kernel's launches (after --skip) of every counter in <dir>/**/*counter_collection.csv, divided by
            the env count (one wavefront per env: per wavefront = per env-step)
  traffic   python3 tools/pmc_summary.py traffic --fetch-dir D1 --write-dir D2 --kernel hoic_substep_kernel --envs 2048 \
                --obj box --out profiles/r02_hbm_traffic.json
            FETCH_SIZE / WRITE_SIZE (KB, separate passes) -> bytes per launch, FETCH_SIZE doubled as MI355X_MICROARCH.md
            prescribes for gfx950
  stats     python3 tools/pmc_summary.py stats --dir gpurun_out/prof --out profiles/r02_bench_kernel_stats.csv
            copies the kernel_stats CSV of a --kernel-trace --stats run
"""
import argparse
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict


def read_counters(d, kernel, skip):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no counter_collection.csv under {d}")
    per_dispatch = defaultdict(dict)          # (file, dispatch id) -> {counter: value}
    order = []
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name") or row.get("Kernel Name") or ""
                if kernel not in name:
                    continue
                key = (f, int(row.get("Dispatch_Id") or row.get("Dispatch Id") or 0))
                if key not in per_dispatch:
                    order.append(key)
                c = row.get("Counter_Name") or row.get("Counter Name")
                per_dispatch[key][c] = per_dispatch[key].get(c, 0.0) + float(row.get("Counter_Value") or row.get("Counter Value") or 0)
    order.sort()
    keep = order[skip:] if len(order) > skip else order
    sums, n = defaultdict(float), defaultdict(int)
    for k in keep:
        for c, v in per_dispatch[k].items():
            sums[c] += v; n[c] += 1
    return {c: sums[c] / n[c] for c in sums}, len(keep)


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    a = sub.add_parser("counters")
    a.add_argument("--dir", required=True, nargs="+"); a.add_argument("--kernel", required=True); a.add_argument("--envs", type=int, required=True)
    a.add_argument("--skip", type=int, default=2); a.add_argument("--out", required=True); a.add_argument("--command", default=""); a.add_argument("--obj", default="box")
    a.add_argument("--build-id", default=None, help="hoic_build_id() of the library the pass ran on (python3 -c 'from hoic_amd import lib; print(lib.build_id())')")
    b = sub.add_parser("traffic")
    b.add_argument("--fetch-dir", required=True); b.add_argument("--write-dir", required=True); b.add_argument("--kernel", required=True)
    b.add_argument("--envs", type=int, required=True); b.add_argument("--obj", default="box"); b.add_argument("--skip", type=int, default=2)
    b.add_argument("--out", required=True); b.add_argument("--command", default=""); b.add_argument("--build-id", default=None)
    c = sub.add_parser("stats")
    c.add_argument("--dir", required=True); c.add_argument("--out", required=True)
    e = sub.add_parser("clock", help="GRBM_GUI_ACTIVE / duration per kernel: the shader clock a kernel ran at (GRBM_GUI_ACTIVE sums the 8 XCDs)")
    e.add_argument("--dir", required=True); e.add_argument("--out", required=True); e.add_argument("--min-us", type=float, default=100.0)
    args = ap.parse_args()
    if args.cmd == "clock":
        acc = defaultdict(list)
        for f in glob.glob(os.path.join(args.dir, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if row["Counter_Name"] != "GRBM_GUI_ACTIVE":
                    continue
                us = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
                if us >= args.min_us:
                    acc[(row["Kernel_Name"][:100], row["Grid_Size"])].append((float(row["Counter_Value"]), us))
        out = {"what": "per kernel (name, grid size): launches, mean duration (us), mean GRBM_GUI_ACTIVE, GHz = GRBM_GUI_ACTIVE / 8 XCDs / duration", "kernels": []}
        for (name, grid), v in sorted(acc.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
            g, us = sum(x[0] for x in v) / len(v), sum(x[1] for x in v) / len(v)
            out["kernels"].append({"kernel": name, "grid": grid, "launches": len(v), "dur_us": round(us, 1), "GRBM_GUI_ACTIVE": round(g), "GHz": round(g / 8 / us / 1e3, 3)})
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(out, open(args.out, "w"), indent=1)
        for k in out["kernels"][:30]:
            print(k)
        return
    if args.cmd == "counters":
        per, launches = {}, 0
        for d in args.dir:
            m, n = read_counters(d, args.kernel, args.skip)
            per.update({k: v / args.envs for k, v in m.items()}); launches = max(launches, n)
        out = {"kernel": args.kernel, "envs": args.envs, "obj": args.obj, "launches_averaged": launches, "command": args.command,
               "per_env_step": {k: per[k] for k in sorted(per)}}
    elif args.cmd == "traffic":
        f, n1 = read_counters(args.fetch_dir, args.kernel, args.skip)
        w, n2 = read_counters(args.write_dir, args.kernel, args.skip)
        fetch_kb, write_kb = f["FETCH_SIZE"], w["WRITE_SIZE"]
        out = {"kernel": args.kernel, "envs": args.envs, "obj": args.obj, "command": args.command, "launches_averaged": min(n1, n2),
               "FETCH_SIZE_KB_raw": fetch_kb, "WRITE_SIZE_KB_raw": write_kb, "fetch_bytes_corrected": 2 * fetch_kb * 1024, "write_bytes": write_kb * 1024,
               "traffic_bytes_per_launch": 2 * fetch_kb * 1024 + write_kb * 1024,
               "note": "FETCH_SIZE doubled (gfx950: the counter reports half of the bytes of wide coalesced reads, MI355X_MICROARCH.md section "
                       "HBM); WRITE_SIZE taken as is (uncalibrated per the guide)"}
    else:
        files = glob.glob(os.path.join(args.dir, "**", "*kernel_stats.csv"), recursive=True)
        if not files:
            raise SystemExit(f"no kernel_stats.csv under {args.dir}")
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        shutil.copy(sorted(files)[-1], args.out)
        print("copied", sorted(files)[-1], "->", args.out)
        return
    out["build_id"] = args.build_id
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)
    print(json.dumps(out)[:600])


if __name__ == "__main__":
    main()
