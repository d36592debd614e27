#!/usr/bin/env python3
"""The seams between rollout and update in a `rocprofv3 --kernel-trace` CSV (gzip ok) of bench.py: for the last complete update
window (last substep end -> first substep start of the next rollout) the time with 0 / 1 / 2 update GEMMs running, and the
kernels of the first gap (rollout end -> first GEMM), the gap around the advantages and the last gap (last GEMM -> first substep).
usage: python3 tools/seam_timeline.py trace.csv[.gz] [verbose]"""
import collections, csv, gzip, sys
f = sys.argv[1]; verbose = len(sys.argv) > 2
rows = list(csv.DictReader(gzip.open(f, "rt") if f.endswith(".gz") else open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Stream_Id"], r.get("Queue_Id", "")) for r in rows)
sub = [i for i, e in enumerate(ev) if "hoic_substep" in e[2]]
wins = [(a, b) for a, b in zip(sub[:-1], sub[1:]) if ev[b][0] - ev[a][1] > 20e6]
if not wins:
    sys.exit("no update window in this trace")
a, b = wins[-1]
W, t0, t1 = ev[a + 1:b], ev[a][1], ev[b][0]
isg = lambda n: "gemm_f16x3" in n
print(f"update window (last substep end -> next rollout's first substep start): {(t1 - t0) / 1e6:.3f} ms, {len(W)} kernels")
pts = sorted([(s, 0, isg(n)) for s, e, n, st, q in W] + [(e, 1, isg(n)) for s, e, n, st, q in W])
g = sm = 0; last = t0; acc = collections.Counter()
for t, kind, ig in pts:
    acc[(min(g, 2), min(sm, 1))] += t - last; last = t
    d = 1 if kind == 0 else -1
    if ig: g += d
    else: sm += d
acc[(0, 0)] += t1 - last
for k, v in sorted(acc.items()):
    print(f"   {k[0]} GEMM(s) running, {'a' if k[1] else 'no'} small kernel beside: {v / 1e6:7.3f} ms")
gi = sorted((s, e) for s, e, n, st, q in W if isg(n))
merged = []
for s, e in gi:
    if merged and s <= merged[-1][1]: merged[-1][1] = max(merged[-1][1], e)
    else: merged.append([s, e])
gaps = [(t0, merged[0][0])] + [(merged[i][1], merged[i + 1][0]) for i in range(len(merged) - 1)] + [(merged[-1][1], t1)]
gaps = [(s, e) for s, e in gaps if e > s]
print(f"time without any GEMM: {sum(e - s for s, e in gaps) / 1e6:.3f} ms in {len(gaps)} gaps; first {(gaps[0][1] - gaps[0][0]) / 1e3:.0f} us, last {(gaps[-1][1] - gaps[-1][0]) / 1e3:.0f} us")
queues = collections.defaultdict(set)
for s, e, n, st, q in ev: queues[st].add(q)
print("stream -> hardware queue:", {k: sorted(v) for k, v in sorted(queues.items())})
show = [gaps[0]] + [g_ for g_ in gaps[1:4] if g_[1] - g_[0] > 60e3][:1] + [gaps[-1]]
for s, e in show:
    print(f"--- gap at {(s - t0) / 1e3:.0f} us, {(e - s) / 1e3:.0f} us long:")
    ks = [(ks_, ke, n, st) for ks_, ke, n, st, q in W if ks_ < e and ke > s and not isg(n)]
    if not verbose:
        c = collections.Counter(); d = collections.Counter()
        for ks_, ke, n, st in ks:
            key = (st, n.split("(")[0].replace("void ", "")[:48]); c[key] += 1; d[key] += ke - ks_
        for key, cnt in c.most_common():
            print(f"      stream {key[0]}  {cnt:3d} x {key[1]:50s} {d[key] / 1e3:7.1f} us")
    else:
        for ks_, ke, n, st in ks:
            print(f"      {(ks_ - t0) / 1e3:9.1f} {(ke - ks_) / 1e3:7.1f} us  stream {st}  {n[:70]}")
