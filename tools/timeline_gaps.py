#!/usr/bin/env python3
"""Idle time of the GPU inside the multi-stream benchmark step, from a rocprofv3 kernel trace: union of all kernel intervals against the
wall time between two optimizer steps, the largest gaps, and the sum of the kernel durations (average concurrency = sum / union).
Usage (GPU box): rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python bench.py --no-cpu-baseline --no-roofline --steps 4
                 python tools/timeline_gaps.py $(find /tmp/tl -name '*kernel_trace.csv')"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Stream_Id") or r.get("Queue_Id")) for r in rows)
sg = [e for e in ev if "sgd_step" in e[2]]
if len(sg) < 3:
    raise SystemExit("need at least three optimizer steps in the trace")
a, b = sg[-3][1], sg[-1][1]                      # the last two full steps
sel = [e for e in ev if e[0] >= a and e[1] <= b]
busy, cur_s, cur_e, gaps = 0, None, None, []
for s, e, n, _ in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _, _ in sel)
print("per step: wall %.2f ms, at least one kernel running %.2f ms, sum of kernel durations %.2f ms (average concurrency %.2f)"
      % ((b - a) / 2e6, busy / 2e6, tot / 2e6, tot / busy))
gaps.sort(reverse=True)
print("idle gaps: %.2f ms per step in %d gaps; largest (us, kernel that ended the gap): %s"
      % (sum(g for g, _ in gaps) / 2e6, len(gaps), [(round(g / 1e3, 1), n[:36]) for g, n in gaps[:6]]))
print("launches per queue:", collections.Counter(e[3] for e in sel).most_common(8))
