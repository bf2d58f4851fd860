"""What the device does around a step boundary, from a rocprofv3 kernel trace of bench.py:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 8 --warmup 2 --soak 0 --no-cpu-baseline
    python tools/boundary_gap.py gpurun_out/trace [step index, default 5] [us before] [us after]
lists every launch that runs between `before` microseconds ahead of the optimiser launch that ends the chosen
minibatch and `after` microseconds behind it (start and end relative to that launch's start, queue, name), then, for
every boundary of the trace, the time from the optimiser launch's end to the start of the next minibatch's first
matrix product."""
import csv, glob, os, re, sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.split(r"[<(]", n, maxsplit=1)[0].strip()


def main(directory, which=5, before=400.0, after=600.0):
    path = glob.glob(os.path.join(directory, "**", "*_kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    adam = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]) == "adam_kernel"]
    a = adam[which]
    t0 = int(rows[a]["Start_Timestamp"])
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if e >= t0 - before * 1e3 and s <= t0 + after * 1e3:
            print("%9.1f .. %9.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, r["Queue_Id"], short(r["Kernel_Name"])[:40]))
    print()
    for k, i in enumerate(adam[:-1]):
        end = int(rows[i]["End_Timestamp"])
        nxt = next((r for r in rows[i + 1:] if short(r["Kernel_Name"]).startswith("gemm")), None)
        busy_until = max(int(r["End_Timestamp"]) for r in rows[max(0, i - 40): i + 1])
        first = int(rows[i + 1]["Start_Timestamp"]) if i + 1 < len(rows) else end
        if nxt is not None:
            print("boundary %2d: optimiser launch %.1f us; next launch starts %.1f us after it ends, first matrix product %.1f us after"
                  % (k, (end - int(rows[i]["Start_Timestamp"])) / 1e3, (first - busy_until) / 1e3,
                     (int(nxt["Start_Timestamp"]) - end) / 1e3))


if __name__ == "__main__":
    args = sys.argv[1:]
    main(args[0], int(args[1]) if len(args) > 1 else 5, float(args[2]) if len(args) > 2 else 400.0,
         float(args[3]) if len(args) > 3 else 600.0)
