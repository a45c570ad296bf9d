"""Idle time between consecutive kernels of the bench's train steps, from a rocprofv3 kernel trace.

    python tools/gap_analysis.py <kernel_trace.csv> [skip_fraction]

Gaps are start[i] - end[i-1] over the last (1 - skip_fraction) of the trace (the timed steps).
Printed: total busy / idle time, the gap histogram, and the idle time grouped by the kernel that
FOLLOWS the gap (the one that started late).
"""
import csv, sys, collections
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from pmc_summary import short_name

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short_name(r["Kernel_Name"])))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * skip):]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = []
end = rows[0][1]
for s, e, n in rows[1:]:
    gaps.append((max(0, s - end), n))
    end = max(end, e)
idle = sum(g for g, _ in gaps)
print(f"kernels {len(rows)}  span {span/1e6:.2f} ms  busy {busy/1e6:.2f} ms  idle {idle/1e6:.2f} ms ({100*idle/span:.1f} %)  "
      f"mean gap {idle/len(gaps)/1e3:.2f} us")
edges = [0, 1000, 2000, 4000, 8000, 16000, 64000, 10**12]
for lo, hi in zip(edges, edges[1:]):
    sel = [g for g, _ in gaps if lo <= g < hi]
    print(f"  gap {lo/1e3:5.0f}..{hi/1e3 if hi < 10**11 else float('inf'):5.0f} us: n={len(sel):6d}  total {sum(sel)/1e6:7.3f} ms")
by = collections.defaultdict(lambda: [0, 0])
for g, n in gaps:
    by[n][0] += 1; by[n][1] += g
for n, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {t/1e6:7.3f} ms  n={c:5d}  mean {t/c/1e3:6.2f} us  before {n[:80]}")
