"""The three-piece bf16 GEMM against the exact-f32 kernel: time alone on the device and error against float64.

    python tools/gemm_f32x3_check.py
"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

hip = importlib.import_module(bench.PKG + ".hipabi")
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(0)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


# (M, N, K, a_kmajor, b_kmajor, f32 plan (tile, splits), what)
for m, n, k, ka, kb, plan32, what in [(9728, 512, 10000, 0, 1, (3, 2), "G0 . W"), (10000, 512, 9728, 1, 1, (3, 2), "G0^T . Hs"),
                                      (2048, 512, 9728, 1, 1, (3, 4), "dgates^T . X"), (9728, 512, 2048, 0, 1, (3, 2), "dgates . W_ih"),
                                      (512, 2048, 9728, 1, 1, (3, 4), "X^T . dgates"),
                                      (512, 12048, 512, 0, 0, (3, 1), "logits | h.W_hh"), (10000, 2048, 512, 0, 0, (3, 1), "per-word table"),
                                      (9728, 512, 512, 0, 0, (3, 1), "chain input products"),
                                      (1000, 300, 777, 0, 0, (3, 1), "edges"), (130, 260, 4100, 1, 0, (3, 1), "edges, a k-major")]:
    a = torch.randn((k, m) if ka else (m, k), generator=g).to(dev)
    b = torch.randn((k, n) if kb else (n, k), generator=g).to(dev)
    if what.startswith("G0"):   # the real operand: onehot - softmax rows
        p = torch.softmax(3.0 * a, dim=1)   # a is stored [rows, V] in both products
        tok = torch.multinomial(p, 1)
        a = -p
        a.scatter_add_(1, tok, torch.ones_like(tok, dtype=torch.float32))
    ref = ((a.t() if ka else a).double() @ (b if kb else b.t()).double())
    scratch = torch.empty(16 * m * n, device=dev)
    out32, outx3 = torch.empty(m, n, device=dev), torch.empty(m, n, device=dev)

    def run32():
        t, s = plan32
        if s > 1:
            slabs = scratch[:s * m * n].view(s, m, n)
            hip.gemm_splitk(a, b, slabs, s, a_kmajor=bool(ka), b_kmajor=bool(kb), tile=t)
            hip.sum_slabs(slabs, out32)
        else:
            hip.gemm(a, b, out32, a_kmajor=bool(ka), b_kmajor=bool(kb), tile=t)

    t32 = timed(run32)
    scale = ref.abs().max().item()
    e32 = (out32.double() - ref).abs().max().item() / scale
    cells = []
    for tile in (1,):
        best = None
        for s in (1, 2, 3, 4, 5, 6, 8):
            if k // s < 256:
                continue

            def runx3():
                if s > 1:
                    slabs = scratch[:s * m * n].view(s, m, n)
                    hip.gemm_splitk(a, b, slabs, s, a_kmajor=bool(ka), b_kmajor=bool(kb), tile=tile, dtype="f32x3")
                    hip.sum_slabs(slabs, outx3)
                else:
                    hip.gemm(a, b, outx3, a_kmajor=bool(ka), b_kmajor=bool(kb), tile=tile, dtype="f32x3")

            tx = timed(runx3)
            ex = (outx3.double() - ref).abs().max().item() / scale
            if best is None or tx < best[0]:
                best = (tx, s, ex)
        cells.append("best k/%d %7.1f us err %.1e" % (best[1], best[0], best[2]))
    print("%6d %6d %6d %d %d  f32 %7.1f us err %.1e | f32x3 %s   %s" % (m, n, k, ka, kb, t32, e32, "  ".join(cells), what))
