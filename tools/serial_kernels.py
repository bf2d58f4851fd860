"""One A2C minibatch with everything on ONE stream (no overlap), for a rocprofv3 kernel trace:
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/serial -- python3 tools/serial_kernels.py
then  python tools/serial_kernels.py --report gpurun_out/serial  lists the last step's kernels in launch
order with their durations, merged where the same kernel repeats back to back."""
import glob
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import numpy as np
    import torch
    import bench
    pkg = bench.PKG
    syn = importlib.import_module(pkg + ".synthetic")
    models = importlib.import_module(pkg + ".models")
    optim = importlib.import_module(pkg + ".optim")
    trainers = importlib.import_module(pkg + ".trainers")
    dev = torch.device("cuda", 0)
    n, vocab, caplen, fdim = 512, 10000, 20, 2048
    a2c, rew, flat, params = bench.build_networks(syn, models, optim, vocab, fdim, 0, dev)
    engine = trainers.build_engine(a2c, rew, concurrent=False)
    feats, caps = syn.make_minibatch(n, vocab, caplen, fdim, 0)
    fd, pd = torch.from_numpy(feats).to(dev), torch.from_numpy(caps[:, :1]).to(dev)
    u = torch.from_numpy(np.random.RandomState(1).random_sample((3, 19 * n))).to(dev)
    for i in range(3):
        engine.step(fd, pd, 19, uniforms=u[i]).check()
        torch.cuda.synchronize()


def report(directory):
    import csv
    import re
    path = glob.glob(os.path.join(directory, "**", "*_kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    # the last step starts at the last cnn2linear GEMM ... simpler: the last third of the launches
    third = len(rows) // 3
    rows = rows[len(rows) - third:]
    out, total = [], 0.0
    for r in rows:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = re.sub(r"\(.*", "", name)[:48]
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) // 256
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        total += us
        if out and out[-1][0] == name and out[-1][1] == grid:
            out[-1][2] += 1
            out[-1][3] += us
        else:
            out.append([name, grid, 1, us])
    for name, grid, cnt, us in out:
        print("%-50s wgs %6d  x%-3d %9.1f us" % (name, grid, cnt, us))
    print("total %.2f ms over %d launches" % (total / 1e3, len(rows)))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--report":
        report(sys.argv[2])
    else:
        run()
