"""Per-queue timeline of one timed minibatch from a rocprofv3 kernel trace of bench.py:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 8 --warmup 2 --soak 0 --no-cpu-baseline
    python tools/step_timeline.py gpurun_out/trace [min_gap_us]
lists, for every hardware queue, the launches of the step between the 6th and 7th optimiser launch that follow a
gap (queue idle) of at least min_gap_us, and each queue's busy time: a gap on a queue whose next launch does not
wait for another queue is the HOST being late."""
import csv, glob, os, re, sys


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return re.split(r"[<(]", n, maxsplit=1)[0].strip()


def main(directory, min_gap=15.0):
    path = glob.glob(os.path.join(directory, "**", "*_kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    adam = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]) == "adam_kernel"]
    a, b = adam[5], adam[6]
    t0 = int(rows[a]["End_Timestamp"])
    print("step length %.2f ms, %d launches" % ((int(rows[b]["End_Timestamp"]) - t0) / 1e6, b - a))
    queues = {}
    for r in rows[a + 1: b + 1]:
        queues.setdefault(r["Queue_Id"], []).append(r)
    for q, lst in sorted(queues.items()):
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in lst)
        print("queue %s: %d launches, busy %.2f ms" % (q, len(lst), busy / 1e6))
        prev_end = t0
        for r in lst:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            if (s - prev_end) / 1e3 >= min_gap:
                print("   at %8.1f us: idle %7.1f us before %-28s (%.1f us)" % ((s - t0) / 1e3, (s - prev_end) / 1e3,
                                                                               short(r["Kernel_Name"])[:28], (e - s) / 1e3))
            prev_end = max(prev_end, e)


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 15.0)
