import sys, importlib, torch
sys.path.insert(0, '/root/repo')
hip = importlib.import_module("image-captioning-through-rl_amd.hipabi")
def bench(name, m, n, k, a_km, b_km):
    a = torch.randn((k, m) if a_km else (m, k), device="cuda")
    b = torch.randn((k, n) if b_km else (n, k), device="cuda")
    out = torch.empty(m, n, device="cuda")
    res = []
    for tile in (1, 2, 3):
        f = lambda: hip.gemm(a, b, out, a_kmajor=a_km, b_kmajor=b_km, tile=tile)
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        res.append("t%d: %7.1f us %5.1f TF" % (tile, us, 2.0 * m * n * k / us / 1e6))
    print("%-34s %s" % (name, " | ".join(res)), flush=True)
for m in (512, 1024, 2048, 4096, 16384):
    bench("logits %dx12048x512 NT" % m, m, 12048, 512, False, False)
for k in (512, 1024, 2048, 4096):
    bench("4096x4096x%d NT" % k, 4096, 4096, k, False, False)
bench("dWv 10000x512x9728 TN", 10000, 512, 9728, True, True)
bench("10000x512x9728 NT", 10000, 512, 9728, False, False)
bench("bptt 512x512x2048 NN", 512, 512, 2048, False, True)
