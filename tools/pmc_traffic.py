"""HBM traffic per device kernel from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950).

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d A -o f -- python bench.py ...
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d B -o w -- python bench.py ...
  python tools/pmc_traffic.py A/f_counter_collection.csv B/w_counter_collection.csv profiles/r02_pmc_hbm_traffic.json

The output carries `_source_stamp` (bench.source_stamp(): sha256 of the kernel sources) and `_steps`: bench.py reports
`roofline.traffic` from this file only when the stamp equals the build it is running.  `_roles`: the same traffic per step grouped by roofline role
(runtime.R_*: plain 1x1 streaming / fused 1x1 streaming / KxK MFMA-bound / weight gradient), from the kernels' template arguments.

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


def role(name):
    """Roofline role of an MFMA kernel instance (adamml_amd/runtime.py R_*), from its template arguments:
    conv_gemm_kernel<BC, MODE, PD, RES, DUAL, CAT, FADD, GLDS, EID, LZF, EPI, PF>."""
    b = base(name)
    if b == "conv_gemm_kernel":
        a = [t.strip() for t in re.search(r"conv_gemm_kernel<([^>]*)>", name).group(1).split(",")]
        a += ["false"] * (12 - len(a))
        flag = lambda i: a[i] in ("true", "1")
        if int(a[1]) != 0:
            return "convKxK_mfma"
        if flag(3) or flag(4) or flag(5) or flag(6) or a[10].lstrip("(int)") == "1":
            return "conv1x1_fused_streaming"
        return "conv1x1_streaming"
    if b == "conv1x1_narrow_fwd_kernel":
        return "conv1x1_streaming"
    if b == "conv1x1_narrow_dgrad_kernel":                  # <KS, COUT, DUAL, EPI>: DUAL loader or BatchNorm-fused epilogue -> fused
        a = [t.strip() for t in re.search(r"conv1x1_narrow_dgrad_kernel<([^>]*)>", name).group(1).split(",")]
        return "conv1x1_fused_streaming" if (a[2] in ("true", "1") or a[3].lstrip("(int)") == "1") else "conv1x1_streaming"
    if b == "conv1x1_narrow_wgrad_kernel":
        return "weight_gradient"
    if b in ("conv3x3_c64_kernel", "conv_stem_kernel"):
        return "convKxK_mfma"
    if b in ("alg_stream_kernel", "conv1x1_fadd_next_kernel", "conv1x1_fadd_tpool_kernel", "res_prod_stream_kernel", "conv1x1_fadd_stream_kernel",
             "conv1x1_fadd_tpool_stream_kernel"):
        return "conv1x1_fused_streaming"
    if b == "wide_all_kernel":                              # (csrc/conv1x1_wide.hip: plain forward / data gradient of the wide expanding 1x1 convs)
        return "conv1x1_streaming"
    if b in ("conv_wgrad_kernel", "conv_wgrad_glds_kernel", "conv3x3_wgrad_kernel", "conv3x3_c64_wgrad_kernel", "conv_stem_wgrad_kernel",
             "wgrad_reduce_kernel", "stem_wgrad_reduce_kernel", "gram_colsum_kernel", "gram_reduce_kernel", "tpool_bwd_prod_kernel"):
        return "weight_gradient"
    return None


def load(path, counter, roles=None):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = agg[base(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        if roles is not None:
            ro = role(r["Kernel_Name"])
            if ro:
                roles[ro][0] += 1
                roles[ro][1] += float(r["Counter_Value"])
    return agg


f, w, out = sys.argv[1:4]
rf, rw = collections.defaultdict(lambda: [0, 0.0]), collections.defaultdict(lambda: [0, 0.0])
fa, wa = load(f, "FETCH_SIZE", rf), load(w, "WRITE_SIZE", rw)
res = {}
for k in sorted(set(fa) | set(wa), key=lambda k: -(2 * fa.get(k, [0, 0])[1] + wa.get(k, [0, 0])[1])):
    n = max(fa.get(k, [0, 0])[0], wa.get(k, [0, 0])[0])
    fk, wk = fa.get(k, [0, 0.0])[1], wa.get(k, [0, 0.0])[1]
    tot = (2 * fk + wk) * 1024
    res[k] = {"launches": n, "fetch_kib_raw": round(fk, 1), "write_kib": round(wk, 1), "hbm_bytes_corrected": tot,
              "per_launch_bytes": tot / max(n, 1)}
import bench  # noqa: E402
STEPS = 3          # steps per pass of the command below (1 warm-up + 2)
res["_roles"] = {k: {"launches_per_step": max(rf[k][0], rw[k][0]) / STEPS, "hbm_bytes_per_step": (2 * rf[k][1] + rw[k][1]) * 1024 / STEPS}
                 for k in sorted(set(rf) | set(rw))}
res["_steps"] = STEPS
res["_total_hbm_bytes_per_step"] = sum(v["hbm_bytes_corrected"] for k, v in res.items() if not k.startswith("_")) / STEPS
res["_source_stamp"] = bench.source_stamp()
res["_command"] = "bench.py --single-stream --steps 2 --warmup 1 --no-cpu-baseline --no-roofline (3 steps per pass)"
json.dump(res, open(out, "w"), indent=1)
for k, v in [kv for kv in res.items() if not kv[0].startswith("_")][:14]:
    print("%-32s launches %6d  %.1f GB total, %.1f MB / launch" % (k, v["launches"], v["hbm_bytes_corrected"] / 1e9, v["per_launch_bytes"] / 1e6))
