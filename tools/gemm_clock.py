"""Is the f32 GEMM held back by the clock the chip sustains under load?  The same launches on random and on
all-zero operands (same instruction stream, same cycles; zero operands toggle no multiplier bits and draw less
power): a gap between the two rates is clock, not stalls.

    python tools/gemm_clock.py
"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

hip = importlib.import_module(bench.PKG + ".hipabi")
dev = torch.device("cuda", 0)
for m, n, k, ka, kb, tile in [(9728, 512, 10000, 0, 1, 3), (512, 12048, 512, 0, 0, 3), (10000, 2048, 512, 0, 0, 3),
                              (9728, 512, 10000, 0, 1, 1)]:
    for fill in ("random", "zeros", "random"):
        mk = torch.randn if fill == "random" else torch.zeros
        a = mk((k, m) if ka else (m, k), device=dev)
        b = mk((k, n) if kb else (n, k), device=dev)
        out = torch.empty(m, n, device=dev)
        reps = max(20, int(2e-2 / (2.0 * m * n * k / 100e12)))
        for _ in range(reps):   # long enough for the clock to settle
            hip.gemm(a, b, out, a_kmajor=bool(ka), b_kmajor=bool(kb), tile=tile)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            hip.gemm(a, b, out, a_kmajor=bool(ka), b_kmajor=bool(kb), tile=tile)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        print("%6d %6d %6d tile %d %-7s %8.1f us  %6.1f TFLOP/s  %.2f of peak" % (m, n, k, tile, fill, us, 2.0 * m * n * k / us / 1e6,
                                                                               2.0 * m * n * k / us / 1e6 / 157.3))
