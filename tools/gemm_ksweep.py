"""Fixed and per-k cost of the f32 GEMM at one output shape: time over a range of k, least-squares line.

    python tools/gemm_ksweep.py M N tile [kA kB]
"""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

hip = importlib.import_module(bench.PKG + ".hipabi")
m, n, tile = (int(x) for x in sys.argv[1:4])
ka, kb = (int(x) for x in sys.argv[4:6]) if len(sys.argv) > 5 else (0, 0)
dev = torch.device("cuda", 0)
ks, ts = [32, 64, 128, 256, 512, 1024, 2048], []
for k in ks:
    a = torch.randn((k, m) if ka else (m, k), device=dev)
    b = torch.randn((k, n) if kb else (n, k), device=dev)
    out = torch.empty(m, n, device=dev)
    for _ in range(3):
        hip.gemm(a, b, out, a_kmajor=bool(ka), b_kmajor=bool(kb), tile=tile)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        hip.gemm(a, b, out, a_kmajor=bool(ka), b_kmajor=bool(kb), tile=tile)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) * 1e3 / 20)
    print("k %5d  %7.1f us" % (k, ts[-1]))
slope, icpt = np.polyfit(ks[3:], ts[3:], 1)
print("fit over k >= 256: %.1f us fixed + %.4f us per k (%.2f of f32 peak in the k loop)" % (icpt, slope, 2.0 * m * n / (slope * 1e-6) / 157.3e12))
