"""Regenerates profiles/README.md from the committed rocprofv3 kernel statistics, PMC traffic and bench lines of a round.

    python tools/profile_readme.py [r03]"""
import collections
import csv
import json
import re
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r03"
P = "profiles/%s_" % R


def kname(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"conv_gemm_kernel<(\d+), (\d), (\d), (true|false), (true|false)(?:, (true|false))?(?:, (true|false))?[^>]*>", n)
    if m:
        tag = (" FADD (forward conv3 + BatchNorm + residual add + ReLU epilogue)" if m.group(7) == "true" else
               " RES (residual-backward epilogue)" if m.group(4) == "true" else
               " DUAL (BatchNorm-backward loader)" if m.group(5) == "true" else
               " CAT (algebraic BatchNorm backward: [g' | a] data gradient)" if m.group(6) == "true" else
               " 3x3 / strided (MODE %s)" % m.group(2) if m.group(2) != "0" else " 1x1 forward / data gradient")
        return "conv_gemm_kernel" + tag
    return re.sub(r"[<(].*", "", n).strip()


agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(P + "bench_single_stream_kernel_stats.csv")):
    a = agg[kname(r["Name"])]
    a[0] += int(r["Calls"])
    a[1] += float(r["TotalDurationNs"]) / 1e6
T = sum(v[1] for v in agg.values())
ss = json.load(open(P + "bench_single_stream_under_rocprof.json"))
df = json.load(open(P + "bench_default.json"))
tr = json.load(open(P + "pmc_hbm_traffic.json"))
steps = ss["steps"] + ss["warmup"] + 2
cg = [v for k, v in agg.items() if k.startswith("conv_gemm_kernel")]
cg_calls, cg_ms = sum(v[0] for v in cg), sum(v[1] for v in cg)
out = []
out.append("# Round %s profile summary (MI355X, B=72 x 5 segments, single-stream run so that kernel durations are not inflated by overlap)\n" % R[1:].lstrip("0"))
out.append("Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --single-stream --no-cpu-baseline` (`tools/gpu_round.sh`;")
out.append("%d steps: %d warm-up + %d timed + 2 of the roofline leg).  Full per-kernel table: `%s_bench_single_stream_kernel_stats.csv`;" % (steps, ss["warmup"], ss["steps"], R))
out.append("the JSON line that run printed: `%s_bench_single_stream_under_rocprof.json` (%.1f ms/step); the default multi-stream bench line of the same box:" % (R, ss["ms_per_step"]))
out.append("`%s_bench_default.json` (**%.0f clips/s, %.1f ms/step**, HIP-event median step %.1f ms, peak memory %.0f GiB); GPU test log of the same call: `%s_gpu_tests.txt`.\n"
           % (R, df["value"], df["ms_per_step"], df["ms_per_step_median_hipevent"], df["peak_mem_gib"], R))
out.append("| device kernel (template instances merged by role) | launches / step | ms / step | share | avg us |")
out.append("|---|---|---|---|---|")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:32]:
    out.append("| %s | %.1f | %.2f | %.1f %% | %.1f |" % (k, v[0] / steps, v[1] / steps, 100 * v[1] / T, 1e3 * v[1] / v[0]))
out.append("")
rf = df["roofline"]
out.append("All device kernels: %.1f ms per step back to back (single stream); the multi-stream step takes %.1f ms.\n" % (T / steps, df["ms_per_step"]))
out.append("conv_gemm_kernel (all instances): rocprofv3 average %.1f us over %d launches; bench.py's HIP-event average for the same kernel %.1f us over %d launches of one step"
           % (1e3 * cg_ms / cg_calls, cg_calls, rf["avg_launch_us"], rf["launches_per_step"]))
out.append("(the events bracket the launch on its stream and include ~2-3 us of launch overhead).  Roofline: %.0f GB/s = %.2f of the 8 TB/s HBM peak over %.1f GB of"
           % (rf["achieved"], rf["frac"], rf["algorithmic_gb_per_step"]))
out.append("algorithmic traffic per step (%.3f GB per launch); step level (SURVEY.md section 8d figures): %.3f of the HBM roof, %.3f of the MFMA roof.\n"
           % (rf["algorithmic_gb_per_launch"], df["step_roofline"]["hbm_frac"], df["step_roofline"]["mfma_frac"]))
t = tr["conv_gemm_kernel"]
tot = sum(v["hbm_bytes_corrected"] for k, v in tr.items() if not k.startswith("_")) / 3e9
out.append("HBM traffic (`%s_pmc_hbm_traffic.json`, `tools/gpu_pmc.sh`: two `--pmc` passes FETCH_SIZE / WRITE_SIZE over `bench.py --single-stream --steps 2 --warmup 1`," % R)
out.append("FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md; stamped with the kernel-source hash `%s`, bench.py reports it only for that build):" % tr.get("_source_stamp"))
out.append("conv_gemm_kernel %.1f MB per launch = %.1f GB per step measured vs %.1f GB algorithmic (%.2fx); **all kernels %.0f GB per step = %.2fx the 403 GB of SURVEY.md section 8(d)**"
           % (t["per_launch_bytes"] / 1e6, t["per_launch_bytes"] * rf["launches_per_step"] / 1e9, rf["algorithmic_gb_per_step"],
              t["per_launch_bytes"] * rf["launches_per_step"] / 1e9 / rf["algorithmic_gb_per_step"], tot, tot / 403.0))
out.append("(round 1: 587 GB, 1.46x).  Largest contributors per step: " + ", ".join(
    "%s %.0f GB" % (k, v["hbm_bytes_corrected"] / 3e9) for k, v in sorted(((k, v) for k, v in tr.items() if not k.startswith("_")),
                                                                          key=lambda kv: -kv[1]["hbm_bytes_corrected"])[:8]) + ".\n")
import os


def line(n):
    return json.load(open(P + n + ".json"))


if rf.get("per_role"):
    out.append("The same single-stream measurement split by ROLE (bench.py `roofline.per_role`; each role against the roof that bounds it, PMC traffic per role from the template arguments):\n")
    out.append("| role | C-ABI launches / step | ms / step | bound | fraction of that roof | HBM fraction | MFMA fraction | PMC / algorithmic bytes |")
    out.append("|---|---|---|---|---|---|---|---|")
    for k, v in rf["per_role"].items():
        out.append("| %s | %d | %.2f | %s | %.3f | %.3f | %.3f | %s |" % (k, v["c_abi_launches_per_step"], v["ms_per_step"], v["bound"], v["frac"], v["hbm_frac"], v["mfma_frac"],
                                                                        ("%.2f" % v["traffic_over_algorithmic"]) if v.get("traffic_over_algorithmic") else "-"))
    out.append("")
others = [n for n in ("bench_policy_stage", "bench_inference_skipping", "bench_c4_rgb_flow_rgbdiff_b72", "bench_c5_four_modalities_b48") if os.path.exists(P + n + ".json")]
out.append("Other bench lines of this round (one MI355X): " + "; ".join("`%s_%s.json` %.0f clips/s" % (R, n, line(n)["value"]) for n in others) + ".\n")
b9 = [("bench_b9", "B = 9 (the per-GPU share of the reference recipe: global batch 72 over 8 GPUs, train_adamml.py:122), eager"),
      ("bench_b9_launch_plan", "B = 9, launch plans (`--launch-plan`)"),
      ("bench_b9_forced_collectives", "B = 9 with the configs[2] choreography forced on a one-rank RCCL communicator (`--force-collectives`: SyncBN rounds + bucketed all-reduce)"),
      ("bench_b9_forced_collectives_launch_plan", "the same with launch plans"),
      ("bench_b72_forced_collectives", "B = 72 with the forced choreography (alternating exchange groups: the default)"),
      ("bench_b72_forced_collectives_one_group", "B = 72, forced, `ADAMML_SYNC_GROUPS=1` (one coalesced collective per BatchNorm depth: the form before)")]
if all(os.path.exists(P + n + ".json") for n, _ in b9):
    out.append("configs[2] as far as one GPU allows (host issue time = wall time `step()` takes to return, nothing in it synchronises):\n")
    out.append("| line | clips/s | ms / step | host issue ms | peak GiB |")
    out.append("|---|---|---|---|---|")
    for n, what in b9:
        d = line(n)
        out.append("| `%s_%s.json`: %s | %.0f | %.2f | %.2f | %.1f |" % (R, n, what, d["value"], d["ms_per_step"], d["host_issue_ms"], d["peak_mem_gib"]))
    out.append("")
strong = [("bench_b72_forced_collectives", 1, 72), ("bench_b36_forced_collectives", 2, 36), ("bench_b18_forced_collectives", 4, 18),
          ("bench_b9_forced_collectives_launch_plan", 8, 9)]
if all(os.path.exists(P + n + ".json") for n, _, _ in strong):
    out.append("The per-GPU shares of the reference's GLOBAL batch of 72 (`train_adamml.py:122`: `-b` is split over the ranks), each run on this one GPU with the "
               "configs[2] choreography forced on a one-rank RCCL communicator (launch plans where the share is <= 16 videos).  What the table contains: everything "
               "a rank does per step -- kernels, host issue, 106 SyncBN rounds, bucketed all-reduce launches; what it cannot contain: the time the messages spend on xGMI "
               "(every collective is an identity here) and waiting for slower ranks:\n")
    out.append("| GPUs of the recipe | videos per GPU | ms / step on one GPU | videos/s per GPU | projected global clips/s (x GPUs, transfers excluded) | per-GPU efficiency vs B = 72 |")
    out.append("|---|---|---|---|---|---|")
    base = None
    for n, g, b in strong:
        d = line(n)
        vps = b / d["ms_per_step"] * 1e3
        base = base or vps
        out.append("| %d | %d | %.2f | %.0f | %.0f | %.2f |" % (g, b, d["ms_per_step"], vps, vps * 5 * g, vps / base))
    out.append("")
out.append("Per-layer tables of the same build (B = 72 shapes): `%s_per_layer_bench_conv.txt` (every ResNet-50 conv: forward / data gradient / weight gradient, GB/s and TFLOP/s;"
           " `tools/bench_conv.py`), `%s_per_layer_bench_fused.txt` (RES / DUAL forms), `%s_per_layer_bench_dw.txt` (depthwise), `%s_bench_elementwise.txt` (BatchNorm / residual passes"
           " against a plain copy), `%s_launch_table_resnet.txt` / `%s_launch_table_sound.txt` (every launch of one backbone step with its excess over a 5.3 TB/s / 800 TFLOP/s floor;"
           " `tools/launch_table.py`), `%s_bench_nets.txt` (each backbone alone), `%s_kernel_resources.txt` (registers / spills / LDS of every kernel instance, `tools/kernel_resources.py`)." % ((R,) * 8))
if os.path.exists("profiles/%s_per_layer_bench_dw_bwd.txt" % R):
    out.append("")
    out.append("Round-5 fused forms against the launches they replace, same build: `%s_per_layer_bench_dw_bwd.txt` (one-pass depthwise backward, stride 1 and 2;"
               " `tools/bench_dw_bwd.py`), `%s_per_layer_bench_fadd_next.txt` (conv3 + add + ReLU + the next block's conv1; `tools/bench_fadd_next.py`)."
               "  `tools/bench_fadd_tpool.py` and `tools/bench_tpool_bwd.py` time the two temporal-pool kernels the same way." % (R, R))
if os.path.exists("profiles/%s_bench_deterministic.json" % R):
    ab = [json.loads(l) for l in open("profiles/%s_bench_deterministic.json" % R) if l.strip()]
    out.append("")
    out.append("`%s_bench_deterministic.json`: the benchmark step with reproducible per-channel sums (order-fixed in the workgroup, exact integer bins across "
               "workgroups; `deterministic: true`) against the fp64-atomic form it replaced (`false`), back to back on one box, on the last build that "
               "carried both forms: %s ms per step." % (R, ", ".join("%s %.2f" % ("true" if d["deterministic"] else "false", d["ms_per_step"]) for d in ab)))
    out.append("`%s_launch_table_policy_rgb.txt` / `_policy_sound.txt`: every launch of the frozen policy MobileNetV2s' forward; `%s_gpu_tests.txt`: `pytest -m gpu` "
               "of the same build." % (R, R))
open("profiles/README.md", "w").write("\n".join(out) + "\n")
print("\n".join(out))
