"""Kernels of one traced minibatch in start order, from a `rocprofv3 --kernel-trace --output-format csv` directory:

    python tools/step_kernels.py <trace dir> [minibatch index, default 5] [shortest kernel listed in us, default 30] [from us, default 0]

Minibatches are delimited by the fused Adam launch; every listed line is start (us after the previous minibatch's
Adam ended), duration, hardware queue, kernel.  The chain, seam and Adam launches are listed whatever their length."""
import csv
import glob
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.split(r"[<(]", name, maxsplit=1)[0].strip()


path = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 5
shortest = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
first = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]) == "adam_kernel"]
a, b = adam[which], adam[which + 1]
t0 = int(rows[a]["End_Timestamp"])
print("minibatch %d: %.3f ms from Adam to Adam" % (which, (int(rows[b]["End_Timestamp"]) - t0) / 1e6))
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = short(r["Kernel_Name"])
    if (s - t0) / 1e3 >= first and ((e - s) / 1e3 >= shortest or name.startswith(("seg", "adam", "seam"))):
        print("%8.1f %8.1f  q%s %-40s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Queue_Id"], name[:40]))
