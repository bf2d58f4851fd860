"""One GEMM shape, a few launches, for counter collection:
    python tools/gemm_one.py M N K kA kB tile [launches] [f32|bf16|f32x3] [k splits]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

hip = importlib.import_module(bench.PKG + ".hipabi")
m, n, k, ka, kb, tile = (int(x) for x in sys.argv[1:7])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 5
dev = torch.device("cuda", 0)
a = torch.randn((k, m) if ka else (m, k), device=dev)
b = torch.randn((k, n) if kb else (n, k), device=dev)
dtype = sys.argv[8] if len(sys.argv) > 8 else "f32"
splits = int(sys.argv[9]) if len(sys.argv) > 9 else 1
out = torch.empty(m, n, device=dev)
slabs = torch.empty(splits, m, n, device=dev)
for _ in range(reps):
    if splits > 1:
        hip.gemm_splitk(a, b, slabs, splits, a_kmajor=bool(ka), b_kmajor=bool(kb), tile=tile, dtype=dtype)
    else:
        hip.gemm(a, b, out, a_kmajor=bool(ka), b_kmajor=bool(kb), tile=tile, dtype=dtype)
torch.cuda.synchronize()
