"""Time icrl_gemm_f32 on the shapes of the A2C step for each output-tile / split-K choice."""
import sys, importlib, torch
sys.path.insert(0, '/root/repo')
hip = importlib.import_module("image-captioning-through-rl_amd.hipabi")
lib = hip.load()
def bench(name, m, n, k, a_km, b_km, splits=1):
    a = torch.randn((k, m) if a_km else (m, k), device="cuda")
    b = torch.randn((k, n) if b_km else (n, k), device="cuda")
    out = torch.empty(m, n, device="cuda"); slabs = torch.empty(max(splits,1), m, n, device="cuda")
    res = []
    for tile in (1, 2, 3):
        for sp in ([1] if splits == 1 else [1, splits]):
            f = (lambda: hip.gemm(a, b, out, a_kmajor=a_km, b_kmajor=b_km, tile=tile)) if sp == 1 else (lambda: hip.gemm_splitk(a, b, slabs[:sp], sp, a_kmajor=a_km, b_kmajor=b_km, tile=tile))
            for _ in range(3): f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            res.append("t%d/s%d: %7.1f us %5.1f TF" % (tile, sp, us, 2.0 * m * n * k / us / 1e6))
    print("%-34s %s" % (name, " | ".join(res)), flush=True)
bench("gates 512x2048x512 NT", 512, 2048, 512, False, False)
bench("vocab 512x10000x512 NT", 512, 10000, 512, False, False)
bench("XPv 9728x2048x512 NT", 9728, 2048, 512, False, False)
bench("dHl 9728x512x10000 NN", 9728, 512, 10000, False, True, 2)
bench("dWv 10000x512x9728 TN", 10000, 512, 9728, True, True, 3)
bench("dWih 2048x512x9728 TN", 2048, 512, 9728, True, True, 4)
bench("dWhh_v 2048x512x97280 TN", 2048, 512, 97280, True, True, 8)
bench("bptt 512x512x2048 NN", 512, 512, 2048, False, True, 4)
bench("dXv 9728x512x2048 NN", 9728, 512, 2048, False, True)
bench("cnn 512x512x2048 NT", 512, 512, 2048, False, False, 4)
bench("cube 4096^3 NT", 4096, 4096, 4096, False, False)
bench("cube 4096^3 TN", 4096, 4096, 4096, True, True)
