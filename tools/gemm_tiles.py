"""The f32 GEMM's tile shapes on the products of one A2C minibatch: each (shape, tile code) alone on the device,
HIP events around 20 launches, results compared bit for bit with the 64x64 tiles' (unsplit products sum
over k in the same order whatever the tile).

    python tools/gemm_tiles.py
"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

hip = importlib.import_module(bench.PKG + ".hipabi")
dev = torch.device("cuda", 0)
TILES = {1: "128x128", 2: "128x64", 3: "64x64"}
# (M, N, K, a_kmajor, b_kmajor, what)
SHAPES = [(512, 12048, 512, 0, 0, "logits | h.W_hh, per rollout step"),
          (512, 10000, 512, 0, 0, "logits, first rollout steps"),
          (10000, 2048, 512, 0, 0, "per-word table"),
          (9728, 512, 512, 0, 0, "chain input products"),
          (9728, 512, 10000, 0, 1, "G0 . W"),
          (10000, 512, 9728, 1, 1, "G0^T . H"),
          (9728, 512, 2048, 0, 1, "dgates . W_ih"),
          (2048, 512, 9728, 1, 1, "dgates^T . X")]
g = torch.Generator(device="cpu").manual_seed(0)
print("%6s %6s %6s kA kB  " % ("M", "N", "K") + "  ".join("%9s" % TILES[t] for t in sorted(TILES)) + "   (us; of f32 peak)")
for m, n, k, ka, kb, what in SHAPES:
    a = torch.randn((k, m) if ka else (m, k), generator=g).to(dev)
    b = torch.randn((k, n) if kb else (n, k), generator=g).to(dev)
    want = torch.empty(m, n, device=dev)
    hip.gemm(a, b, want, a_kmajor=bool(ka), b_kmajor=bool(kb), tile=3)
    cells = []
    for t in sorted(TILES):
        out = torch.zeros(m, n, device=dev)
        for _ in range(3):
            hip.gemm(a, b, out, a_kmajor=bool(ka), b_kmajor=bool(kb), tile=t)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            hip.gemm(a, b, out, a_kmajor=bool(ka), b_kmajor=bool(kb), tile=t)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        same = bool(torch.equal(out, want))
        cells.append("%6.1f %.2f%s" % (us, 2.0 * m * n * k / (us * 1e-6) / 157.3e12, "" if same else "!"))
    print("%6d %6d %6d %2d %2d  " % (m, n, k, ka, kb) + "  ".join(cells) + "   " + what)
