"""Regenerates profiles/README.md from the committed rocprofv3 kernel statistics, PMC traffic and bench lines."""
import collections, csv, json, re
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open("profiles/r01_bench_single_stream_kernel_stats.csv")):
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"conv_gemm_kernel<(\d+), (\d), (\d), (true|false), (true|false)(?:, (true|false))?>", n)
    if m:
        n = "conv_gemm_kernel" + (" RES (residual epilogue)" if m.group(4) == "true" else " DUAL (BN-backward loader)" if m.group(5) == "true"
                                  else " CAT (algebraic BN backward: [g' | a] data gradient)" if m.group(6) == "true" else "")
    else:
        n = re.sub(r"[<(].*", "", n).strip()
    agg[n][0] += int(r["Calls"]); agg[n][1] += float(r["TotalDurationNs"]) / 1e6
T = sum(v[1] for v in agg.values())
ss = json.load(open("profiles/r01_bench_single_stream_under_rocprof.json"))
df = json.load(open("profiles/r01_bench_default.json"))
pol = json.load(open("profiles/r01_bench_policy_stage.json"))
tr = json.load(open("profiles/r01_pmc_hbm_traffic.json"))
steps = ss["steps"] + ss["warmup"] + 2
cg = [v for k, v in agg.items() if k.startswith("conv_gemm_kernel")]
cg_calls, cg_ms = sum(v[0] for v in cg), sum(v[1] for v in cg)
out = []
out.append("# Round 1 profile summary (MI355X, B=72 x 5 segments, single-stream run so that kernel durations are not inflated by overlap)\n")
out.append("Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --single-stream --no-cpu-baseline` (`tools/gpu_refresh.sh`;")
out.append("%d steps: %d warm-up + %d timed + 2 of the roofline leg).  Full per-kernel table: `r01_bench_single_stream_kernel_stats.csv`;" % (steps, ss["warmup"], ss["steps"]))
out.append("the JSON line that run printed: `r01_bench_single_stream_under_rocprof.json` (%.1f ms/step); the default multi-stream bench line:" % ss["ms_per_step"])
out.append("`r01_bench_default.json` (**%.0f clips/s, %.1f ms/step**); policy stage: `r01_bench_policy_stage.json` (%.0f clips/s).\n" % (df["value"], df["ms_per_step"], pol["value"]))
out.append("| device kernel (template instances merged) | launches | total ms | share | avg us |")
out.append("|---|---|---|---|---|")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:26]:
    out.append("| %s | %d | %.1f | %.1f %% | %.1f |" % (k, v[0], v[1], 100 * v[1] / T, 1e3 * v[1] / v[0]))
out.append("")
rf = df["roofline"]
out.append("conv_gemm_kernel (all instances): rocprofv3 average %.1f us over %d launches; bench.py's HIP-event average for the same kernel %.1f us over %d launches of one step"
           % (1e3 * cg_ms / cg_calls, cg_calls, rf["avg_launch_us"], rf["launches_per_step"]))
out.append("(the events bracket the launch on its stream and include ~2-3 us of launch overhead).  Roofline: %.0f GB/s = %.2f of the 8 TB/s HBM peak over %.1f GB of" % (rf["achieved"], rf["frac"], rf["algorithmic_gb_per_step"]))
out.append("algorithmic traffic per step; the RES / DUAL instances carry the residual-backward and BatchNorm-backward traffic that used to be separate elementwise kernels.\n")
t = tr["conv_gemm_kernel"]
out.append("HBM traffic (`r01_pmc_hbm_traffic.json`, `tools/gpu_pmc.sh`: two `--pmc` passes FETCH_SIZE / WRITE_SIZE over `bench.py --single-stream --steps 2 --warmup 1`,")
out.append("FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md): conv_gemm_kernel %.1f MB per launch = %.1f GB per step measured vs %.1f GB algorithmic (%.2fx)."
           % (t["per_launch_bytes"] / 1e6, t["per_launch_bytes"] * rf["launches_per_step"] / 1e9, rf["algorithmic_gb_per_step"],
              t["per_launch_bytes"] * rf["launches_per_step"] / 1e9 / rf["algorithmic_gb_per_step"]))
out.append("(`roofline.achieved` = algorithmic bytes per launch / average launch duration; `roofline.traffic` = PMC bytes per launch.)")
out.append("\nPer-layer micro-benchmarks behind DESIGN.md section 4: `tools/bench_conv.py`, `tools/bench_fused.py`, `tools/bench_dw.py`, `tools/bench_elementwise.py`.")
open("profiles/README.md", "w").write("\n".join(out) + "\n")
print("\n".join(out))
