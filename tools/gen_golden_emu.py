"""bf16-STORAGE emulation of the full-size golden cases (build container or any CPU box; imports the ORACLE only, never the reference).

The fp32 goldens (tools/gen_golden.py) say where the reference's arithmetic lands; a pipeline that STORES activations in bf16 cannot
land there, and how far it lands away is a property of the network (random-weight stacks amplify a last-bit difference), not of a kernel.
This script runs the oracle with its bf16-storage hook (oracle.QUANT: every tensor the HIP path keeps in bf16 -- dense-conv operands,
conv outputs, residual / pool outputs -- is rounded, all arithmetic stays fp32) on the four full-size cases and stores what the whole-
network parity tests compare: logits, policy logits, every running statistic, the classifier-head gradients.  The GPU tests then gate
on |HIP - emulation| (tight, stable: both round at the same points) and on |HIP - fp32| <= 1.25 x |emulation - fp32| (the intrinsic
bf16 distance), tests/test_parity_fullsize_gpu.py.

    python tools/gen_golden_emu.py [case ...]        # default: resnet50_c1 adamml_c2 adamml_c4 adamml_c5 -> tests/golden/<case>_bf16emu.npz
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.golden_cases import CASES  # noqa: E402
from tests.oracle_harness import oracle_case, GOLDEN_DIR  # noqa: E402

FULL = ["resnet50_c1", "adamml_c2", "adamml_c4", "adamml_c5"]
KEEP = (".logits", ".policy_logits", ".stats_full", ".stats_full_names", ".decisions", ".ce")


def main():
    torch.set_num_threads(int(os.environ.get("ADAMML_CPU_THREADS", "8")))
    for name in (sys.argv[1:] or FULL):
        c = CASES[name]
        modes = [m for m in c["modes"] if m.startswith("train")]
        t0 = time.time()
        out = oracle_case(c, emulate_bf16=True, modes=modes)
        rec = {k: v for k, v in out.items() if k.endswith(KEEP) or ".grad." in k}
        path = os.path.join(GOLDEN_DIR, name + "_bf16emu.npz")
        np.savez_compressed(path, **rec)
        print("%s: %d arrays, %.0f s -> %s (%d KiB)" % (name, len(rec), time.time() - t0, path, os.path.getsize(path) >> 10), flush=True)


if __name__ == "__main__":
    main()
