"""Register / spill / LDS census of every device kernel (hipcc -Rpass-analysis=kernel-resource-usage, device-only compile of each
csrc/*.hip for gfx950; runs in the build container, no GPU).  Usage: python tools/kernel_resources.py [file.hip ...] > profiles/r03_kernel_resources.txt"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "adamml_amd", "csrc", "*.hip")))
rows = []
for f in files:
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "--cuda-device-only", "-c", f,
                        "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    cur = None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: .*Function Name: (\S+)", line)
        if m:
            cur = {"name": subprocess.run(["/usr/bin/c++filt", m.group(1)], capture_output=True, text=True).stdout.strip(), "file": os.path.basename(f)}
            rows.append(cur)
            continue
        for key, pat in (("vgpr", r"\bVGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"SGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
rows = [r for r in rows if "vgpr" in r]
print("%d kernel instances; %d with VGPR spills, %d with scratch" % (len(rows), sum(1 for r in rows if r.get("vspill", 0)), sum(1 for r in rows if r.get("scratch", 0))))
for r in sorted(rows, key=lambda r: (-r.get("vspill", 0), -r.get("scratch", 0), r["file"], r["name"])):
    name = re.sub(r"\(anonymous namespace\)::", "", r["name"])
    name = re.sub(r"\((anonymous namespace::)?\w+P\)$|\(.*\)$", "", name)
    print("%-22s vgpr %3d agpr %3d sgpr %3d occ %d lds %6d scratch %4d vspill %3d  %s" % (r["file"], r["vgpr"], r.get("agpr", 0), r.get("sgpr", 0), r.get("occ", 0),
                                                                                       r.get("lds", 0), r.get("scratch", 0), r.get("vspill", 0), name[:150]))
