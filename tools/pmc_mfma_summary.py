"""MFMA utilisation per kernel from one rocprofv3 --pmc pass of bench.py.

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
        -d gpurun_out/pmc_mfma -- python3 bench.py --no-cpu-baseline
    python tools/pmc_mfma_summary.py gpurun_out/pmc_mfma profiles/pmc_mfma.json

SQ_VALU_MFMA_BUSY_CYCLES is summed over every SIMD of the chip (a v_mfma_f32_32x32x2_f32 keeps its
SIMD's matrix unit busy for 64 cycles); GRBM_GUI_ACTIVE is summed over the 8 XCDs, so one eighth of it
is the dispatch's length in shader cycles.  utilisation = busy / (cycles * 1024 SIMDs): the fraction of
the chip's matrix-unit cycles that did MFMA work while the kernel ran.  Kernels are serialised under
counter collection, so these are stand-alone figures (in the real step two streams share the chip)."""
import csv
import glob
import json
import os
import re
import subprocess
import sys

SIMDS = 256 * 4


def git_sha():
    """HEAD of the tree the summary is written in (the runs are made from that same tree)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        return subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True,
                              timeout=5).stdout.strip() or None
    except (OSError, subprocess.SubprocessError):
        return None


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.split(r"[<(]", name, maxsplit=1)[0].strip()


def main(directory, out_path):
    per_dispatch = {}
    for path in glob.glob(os.path.join(directory, "**", "*_counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                d = per_dispatch.setdefault(row["Dispatch_Id"], {"name": short(row["Kernel_Name"])})
                d[row["Counter_Name"]] = float(row["Counter_Value"])
    table = {}
    for d in per_dispatch.values():
        acc = table.setdefault(d["name"], [0, 0.0, 0.0])
        acc[0] += 1
        acc[1] += d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        acc[2] += d.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    out = {"_git_sha": git_sha(), "_note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; utilisation = busy cycles / "
                    "(dispatch cycles * 1024 SIMDs); kernels run one at a time under counter collection"}
    for name, (n, busy, cycles) in table.items():
        if busy > 0:
            out[name] = {"launches": n, "mfma_busy_cycles_per_launch": busy / n, "cycles_per_launch": cycles / n,
                         "mfma_utilisation": busy / (cycles * SIMDS) if cycles else None}
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for name, row in sorted(((k, v) for k, v in out.items() if not k.startswith("_")), key=lambda kv: -kv[1]["launches"] * kv[1]["cycles_per_launch"]):
        print("%-28s n=%5d  cycles/launch %.3e  MFMA utilisation %.1f %%" % (name, row["launches"], row["cycles_per_launch"],
                                                                         100 * row["mfma_utilisation"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
