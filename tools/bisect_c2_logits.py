"""Which kernel family moves the full-size C2 logit figure?  (round-5 review: c2.train_main.logits went 3.09e-2 -> 3.99e-2 in one round
although every forward kernel added is bit-identical in its OUTPUT to the form it replaced -- only the summation order of BatchNorm
statistics changed.)  Runs the adamml_c2 train_main forward of tests/test_parity_fullsize_gpu.py once per environment setting given on the
command line (each in its own process: the switches are read at import / first use) and prints |HIP - fp32 golden|, |HIP - emulation| for
the logits and the policy logits.

    python tools/bisect_c2_logits.py "" ADAMML_NARROW_STREAM=0 ADAMML_FADD_NEXT=0 ADAMML_GEMM_MFMA=0 ADAMML_FADD_TPOOL_STREAM=0 ADAMML_WIDE_STREAM=0
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import sys
sys.path.insert(0, %r)
import numpy as np, torch
from adamml_amd import synth
from tests.golden_cases import CASES
from tests.oracle_harness import manifest, load_golden
from tests.test_parity_fullsize_gpu import build, hip_train_step, rel_max
c = CASES["adamml_c2"]
gold, emu = load_golden("adamml_c2"), load_golden("adamml_c2_bf16emu")
model = build(c)
sd = synth.synth_state_dict(manifest(c), seed=1234)
logits, sel, plog, grads, state, _ = hip_train_step(model, c, "train_main", sd)
m = "train_main"
print("RESULT logits |HIP-fp32| %%.4e |HIP-emu| %%.4e   policy logits |HIP-fp32| %%.4e |HIP-emu| %%.4e   decisions equal %%s" %% (
    rel_max(logits.numpy(), gold[m + ".logits"]), rel_max(logits.numpy(), emu[m + ".logits"]),
    rel_max(plog.numpy(), gold[m + ".policy_logits"]), rel_max(plog.numpy(), emu[m + ".policy_logits"]),
    np.array_equal(np.round(sel.numpy()), np.round(gold[m + ".decisions"]))))
''' % ROOT


def main():
    for setting in (sys.argv[1:] or [""]):
        env = dict(os.environ)
        for kv in setting.split():
            k, v = kv.split("=", 1)
            env[k] = v
        out = subprocess.run([sys.executable, "-c", CHILD], env=env, cwd=ROOT, capture_output=True, text=True)
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("RESULT")]
        print("[%-28s] %s" % (setting or "default", line[0][7:] if line else "FAILED: " + out.stderr[-400:]), flush=True)


if __name__ == "__main__":
    main()
