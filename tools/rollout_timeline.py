#!/usr/bin/env python3
"""Where a rollout range's cycle goes: per stream of a `rocprofv3 --kernel-trace` CSV (gzip ok), the kernels between two
consecutive substep launches -- their durations and the gaps between them -- averaged over the cycles of the trace.
usage: python3 tools/rollout_timeline.py trace.csv[.gz] [skip_cycles]"""
import csv, gzip, sys, collections
import numpy as np
f = sys.argv[1]; skip = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rows = list(csv.DictReader(gzip.open(f, "rt") if f.endswith(".gz") else open(f)))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
by = collections.defaultdict(list)
for r in rows:
    by[r["Stream_Id"]].append(r)
short = lambda n: n.split("(")[0].replace("void ", "")[:34]
for st, rs in sorted(by.items()):
    rs.sort(key=lambda r: r["s"])
    idx = [i for i, r in enumerate(rs) if "hoic_substep" in r["Kernel_Name"]]
    if len(idx) < skip + 4:
        continue
    cyc = []
    for a, b in zip(idx[skip:-1], idx[skip + 1:]):
        seq = rs[a:b + 1]
        if seq[-1]["s"] - seq[0]["s"] > 20e6:      # an update sits in between
            continue
        cyc.append(seq)
    if not cyc:
        continue
    period = np.mean([c[-1]["s"] - c[0]["s"] for c in cyc]) / 1e3
    print(f"stream {st}: {len(cyc)} cycles, period {period:.1f} us  (substep grid {cyc[0][0]['Grid_Size_X']})")
    n = min(len(c) for c in cyc)
    for k in range(n - 1):
        dur = np.mean([c[k]["e"] - c[k]["s"] for c in cyc]) / 1e3
        gap = np.mean([c[k + 1]["s"] - c[k]["e"] for c in cyc]) / 1e3
        print(f"   {short(cyc[0][k]['Kernel_Name']):36s} {dur:8.1f} us   then gap {gap:7.1f} us")
