"""Per-kernel averages of rocprofv3 --pmc counter_collection CSVs (one row per dispatch and counter): prints, for every kernel,
the mean of each counter over its dispatches (the first dispatch of a kernel is dropped as warm-up when there are several)."""
import csv, re, sys
from collections import defaultdict, OrderedDict
vals = defaultdict(lambda: defaultdict(dict))     # kernel -> counter -> {dispatch: value}
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        k = re.sub(r"\(.*", "", k)[:64]
        d = vals[k][r["Counter_Name"]]
        key = int(r["Dispatch_Id"])
        d[key] = d.get(key, 0.0) + float(r["Counter_Value"])
for k, cs in vals.items():
    out = OrderedDict()
    n = 0
    for c, d in cs.items():
        v = [d[i] for i in sorted(d)]
        if len(v) > 1:
            v = v[1:]
        n = len(v)
        out[c] = sum(v) / len(v)
    print("%-64s n=%d " % (k, n) + " ".join("%s=%.4g" % (c.replace("_sum", ""), v) for c, v in out.items()))
