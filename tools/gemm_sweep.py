"""Sweep output tile and split-K for the A2C step's large products; prints the best few per shape."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = importlib.import_module("image-captioning-through-rl_amd.hipabi")
lib = hip.load()


def time_us(f, reps=5):
    for _ in range(2):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def sweep(name, m, n, k, a_km, b_km, splits_list=(1, 2, 3, 4, 5, 6, 8, 12, 16)):
    a = torch.randn((k, m) if a_km else (m, k), device="cuda")
    b = torch.randn((k, n) if b_km else (n, k), device="cuda")
    out = torch.empty(m, n, device="cuda")
    slabs = torch.empty(max(splits_list), m, n, device="cuda")
    res = []
    for tile in (1, 2, 3):
        for sp in splits_list:
            if sp > 1 and k // sp < 256:
                continue
            if sp == 1:
                f = lambda tile=tile: hip.gemm(a, b, out, a_kmajor=a_km, b_kmajor=b_km, tile=tile)
            else:
                def f(sp=sp, tile=tile):
                    hip.gemm_splitk(a, b, slabs[:sp], sp, a_kmajor=a_km, b_kmajor=b_km, tile=tile)
                    hip.sum_slabs(slabs[:sp], out)
            res.append((time_us(f), tile, sp))
    res.sort()
    cur = hip.plan_gemm(m, n, k)
    cur_us = [r[0] for r in res if (r[1], r[2]) == cur]
    best = " | ".join("t%d/s%d %.0f us %.0f TF" % (t, s, us, 2.0 * m * n * k / us / 1e6) for us, t, s in res[:4])
    print("%-28s plan t%d/s%d %s us || %s" % (name, cur[0], cur[1], "%.0f" % cur_us[0] if cur_us else "?", best), flush=True)


sweep("dWv 10000x512x9728 TN", 10000, 512, 9728, True, True)
sweep("dHl 9728x512x10000 NN", 9728, 512, 10000, False, True)
sweep("dWih 2048x512x9728 TN", 2048, 512, 9728, True, True)
sweep("dX 9728x512x2048 NN", 9728, 512, 2048, False, True)
sweep("XPv 9728x2048x512 NT", 9728, 2048, 512, False, False, (1,))
sweep("XPr 10752x1536x512 NT", 10752, 1536, 512, False, False, (1,))
sweep("vocab 512x10000x512 NT", 512, 10000, 512, False, False, (1,))
sweep("dWhh_v 2048x512x97280 TN", 2048, 512, 97280, True, True, (4, 6, 8, 12, 16))
sweep("se 9728x512x512 NT", 9728, 512, 512, False, False, (1,))
sweep("ve 512x512x2048 NT", 512, 512, 2048, False, False, (1, 2, 4, 8))
