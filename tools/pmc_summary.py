"""Aggregate two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py into HBM bytes per launch.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/pmc_traffic.json

The counters are in KiB.  On gfx950 FETCH_SIZE tallies each 128-byte request of a wide coalesced read
as 64 bytes, so it is doubled; WRITE_SIZE is taken as reported.  Kernel names are cut at the first
'(' or '<' and the anonymous-namespace prefix is dropped, so all tile shapes of a template fold
into one row."""
import csv
import glob
import json
import os
import re
import subprocess
import sys


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


def collect(directory, counter):
    out = {}
    for path in glob.glob(os.path.join(directory, "**", "*_counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                acc = out.setdefault(short(row["Kernel_Name"]), [0, 0.0])
                acc[0] += 1
                acc[1] += float(row["Counter_Value"]) * 1024.0
    return out


def main(fetch_dir, write_dir, out_path):
    fetch, write = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    table = {}
    for name in sorted(set(fetch) | set(write)):
        nf, bf = fetch.get(name, (0, 0.0))
        nw, bw = write.get(name, (0, 0.0))
        table[name] = {"launches": max(nf, nw),
                       "fetch_bytes_per_launch": 2.0 * bf / nf if nf else 0.0,
                       "write_bytes_per_launch": bw / nw if nw else 0.0}
    with open(out_path, "w") as f:
        json.dump({"_git_sha": git_sha(), "_note": "HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                            "(separate passes; FETCH_SIZE doubled for gfx950)", **table}, f, indent=1, sort_keys=True)
    for name, row in sorted(table.items(), key=lambda kv: -kv[1]["launches"] * (kv[1]["fetch_bytes_per_launch"] + kv[1]["write_bytes_per_launch"])):
        print(f"{name:32s} n={row['launches']:6d} fetch/launch={row['fetch_bytes_per_launch'] / 1e6:9.2f} MB "
              f"write/launch={row['write_bytes_per_launch'] / 1e6:9.2f} MB")


if __name__ == "__main__":
    main(*sys.argv[1:4])
