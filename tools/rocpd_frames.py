#!/usr/bin/env python3
"""Per-kernel and per-frame accounting from a rocprofv3 rocpd database (the default output of `rocprofv3 --kernel-trace`):
kernel totals (calls, mean, min, max in us) and, for the LAST frames of the run, the timeline -- every dispatch with its
start relative to the frame's first kernel, its duration and the gap before it -- so that what a frame costs beyond its
kernels' own time is visible.   python tools/rocpd_frames.py results.db [frame-start-kernel-substring] [frames]"""
import collections, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
names = {r[0]: r[1] for r in cur.execute(f"select id, kernel_name from {ks}")}
rows = [(s, e, names[k]) for s, e, k in cur.execute(f"select start, end, kernel_id from {kd} order by start")]
def short(n):
    n = n.replace("ptd::", "").replace("void ", "")
    return n.split("(")[0][:44]
agg = collections.defaultdict(list)
for s, e, n in rows:
    agg[short(n)].append((e - s) / 1e3)
print(f"{'kernel':46s} {'calls':>6s} {'mean us':>9s} {'min':>8s} {'max':>8s} {'total ms':>9s}")
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f"{n:46s} {len(v):6d} {sum(v) / len(v):9.1f} {min(v):8.1f} {max(v):8.1f} {sum(v) / 1e3:9.2f}")
key = sys.argv[2] if len(sys.argv) > 2 else "k_generate"
nfr = int(sys.argv[3]) if len(sys.argv) > 3 else 1
starts = [i for i, r in enumerate(rows) if key in r[2]]
if len(starts) >= nfr + 1:
    for f in range(nfr):
        a, b = starts[-(nfr + 1) + f], starts[-(nfr + 1) + f + 1]
        t0 = rows[a][0]
        busy = sum(e - s for s, e, _ in rows[a:b]) / 1e3
        print(f"\nframe of dispatches {a}..{b - 1}: span {(rows[b][0] - t0) / 1e3:.1f} us, kernels' own time {busy:.1f} us (sum; overlapping kernels count twice)")
        prev_end = t0
        for s, e, n in rows[a:b]:
            print(f"  +{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f}  {short(n)}")
            prev_end = max(prev_end, e)
