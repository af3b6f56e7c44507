"""HBM traffic per device kernel from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d A -o f -- python bench.py ...
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d B -o w -- python bench.py ...
  python tools/pmc_traffic.py A/f_counter_collection.csv B/w_counter_collection.csv profiles/r02_pmc_hbm_traffic.json

The output carries `_source_stamp` (bench.source_stamp(): sha256 of the kernel sources) and `_steps`: bench.py reports
`roofline.traffic` from this file only when the stamp equals the build it is running.

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read, so it is
doubled (MI355X_MICROARCH.md, HBM section).  Kernels are grouped by their base name (template arguments dropped)."""
import collections
import csv
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def base(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"[<(].*", "", n).strip()
    return n or "?"


def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = agg[base(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


f, w, out = sys.argv[1:4]
fa, wa = load(f, "FETCH_SIZE"), load(w, "WRITE_SIZE")
res = {}
for k in sorted(set(fa) | set(wa), key=lambda k: -(2 * fa.get(k, [0, 0])[1] + wa.get(k, [0, 0])[1])):
    n = max(fa.get(k, [0, 0])[0], wa.get(k, [0, 0])[0])
    fk, wk = fa.get(k, [0, 0.0])[1], wa.get(k, [0, 0.0])[1]
    tot = (2 * fk + wk) * 1024
    res[k] = {"launches": n, "fetch_kib_raw": round(fk, 1), "write_kib": round(wk, 1), "hbm_bytes_corrected": tot,
              "per_launch_bytes": tot / max(n, 1)}
import bench  # noqa: E402
res["_source_stamp"] = bench.source_stamp()
res["_command"] = "bench.py --single-stream --steps 2 --warmup 1 --no-cpu-baseline --no-roofline (3 steps per pass)"
json.dump(res, open(out, "w"), indent=1)
for k, v in [kv for kv in res.items() if not kv[0].startswith("_")][:14]:
    print("%-32s launches %6d  %.1f GB total, %.1f MB / launch" % (k, v["launches"], v["hbm_bytes_corrected"] / 1e9, v["per_launch_bytes"] / 1e6))
