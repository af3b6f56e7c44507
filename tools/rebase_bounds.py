"""Re-base the parity bounds from ONE run (GPU box): runs the whole-network parity tests with ADAMML_REBASE set -- tests/parity_bounds.check
then RECORDS every figure instead of asserting it --, prints every measured figure next to the table's, and writes the new table
(bound = min(1.3 x measured, stated tolerance of the category): tests/parity_bounds.py) to tests/parity_bounds.json and, for gpurun,
to gpurun_out/parity_bounds.json (the only directory that travels back).  A figure above its category's stated tolerance is a
regression, not a re-base: the tool exits non-zero and leaves the table alone.  Exact (bit-identity, decisions, integer) asserts of the
same tests stay hard asserts during the run.

Every entry carries its OWN ceiling (min(category ceiling, 2 x the figure it was created with)); a figure above it, or one that grew by more
than 10 % over the committed figure without --allow-growth, is refused (round-5 advisor finding: a category-wide ceiling let a tight entry
drift).

usage: python tools/rebase_bounds.py [--allow-growth] [pytest selection ...]      (default: the files that read the table)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import parity_bounds as pb  # noqa: E402

FILES = ["tests/test_parity_fullsize_gpu.py", "tests/test_syncbn_gpu.py", "tests/test_train_trajectory_gpu.py"]


def main():
    args = [a for a in sys.argv[1:] if a != "--allow-growth"]
    allow_growth = "--allow-growth" in sys.argv[1:]
    sel = args or FILES
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    rec_path = os.path.join(out_dir, "rebase_measured.json")
    if os.path.exists(rec_path):
        os.remove(rec_path)
    measured = {}
    for f in sel:                                   # (one process per file: each writes its own record file)
        env = dict(os.environ, ADAMML_REBASE=rec_path)
        rc = subprocess.call([sys.executable, "-m", "pytest", f, "-x", "-q", "-s", "-m", "gpu", "-p", "no:cacheprovider"], cwd=ROOT, env=env)
        if rc != 0:
            raise SystemExit("rebase: %s failed (an exact assert, not a bound): table unchanged" % f)
        if os.path.exists(rec_path):
            measured.update(json.load(open(rec_path)))
            os.remove(rec_path)
    old = pb.table()
    new = dict(old)
    print("%-46s %12s %12s %12s %12s" % ("entry", "was", "measured", "bound", "ceiling"))
    for k in sorted(measured):
        cat = measured[k]["cat"] or (old[k]["cat"] if k in old else pb_category(k))
        ent = pb.rebased_entry(cat, measured[k]["value"], old.get(k), allow_growth, k)
        print("%-46s %12s %12.4e %12.3e %12.3e" % (k, "%.4e" % old[k]["measured"] if k in old else "new", measured[k]["value"], ent["bound"],
                                                    ent["ceil"]))
        new[k] = ent
    stale = sorted(set(old) - set(measured))
    if stale and sel == FILES:
        print("entries not measured by this run (kept): " + ", ".join(stale))
    for path in (pb.TABLE_PATH, os.path.join(out_dir, "parity_bounds.json")):
        with open(path, "w") as f:
            json.dump(new, f, indent=1, sort_keys=True)
            f.write("\n")
    print("re-based %d entries -> %s" % (len(measured), pb.TABLE_PATH))


def pb_category(key):
    """Category of a NEW entry that did not name one: from its name (<case>.<mode>.<figure>)."""
    fig = key.split(".")[-1]
    if ".eval." in key:
        return "eval_" + fig
    if fig == "head" and "policy" in key:
        return "head_policy"
    return fig


if __name__ == "__main__":
    main()
