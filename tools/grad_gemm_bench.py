"""The backward pass's product shapes on two scaled f16 pieces (dtype "f16x2") and on three bf16 pieces ("f32x3"), with the
passes that compute the scales, each alone on the device (10 launches, HIP events).

    python tools/grad_gemm_bench.py [path of another build of libicrl_a2c.so]
"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
hip = importlib.import_module(bench.PKG + ".hipabi")
if len(sys.argv) > 1:   # another build of the library (kernel variants under test)
    hip.LIB_PATH = os.path.abspath(sys.argv[1])
dev = torch.device("cuda", 0)
def run(m, n, k, akm, bkm):
    a = torch.randn((k, m) if akm else (m, k), device=dev); b = torch.randn((k, n) if bkm else (n, k), device=dev)
    out = torch.empty(m, n, device=dev); scratch = torch.empty((64 if m * n < 4e6 else 8) * m * n, device=dev)
    work = torch.zeros(max(m, n) + k, dtype=torch.int32, device=dev)
    sa, sb = torch.zeros(m, device=dev), torch.zeros(n, device=dev)
    ka = {"col_scale": sa} if akm else {"row_scale": sa}; kb = {"col_scale": sb} if bkm else {"row_scale": sb}
    hip.absmax_scales(a, work, **ka); hip.absmax_scales(b, work, **kb)
    res = []
    for what, fn in (("f16x2", lambda: hip.gemm_auto(a, b, out, scratch, a_kmajor=akm, b_kmajor=bkm, shared=True, dtype="f16x2", a_scale=sa, b_scale=sb)),
                     ("f16x2 on 256x256 tiles", lambda: hip.gemm_auto(a, b, out, scratch, a_kmajor=akm, b_kmajor=bkm, shared=True, dtype="f16x2", a_scale=sa, b_scale=sb, big_tile=True)),
                     ("f32x3", lambda: hip.gemm_auto(a, b, out, scratch, a_kmajor=akm, b_kmajor=bkm, shared=True, dtype="f32x3")),
                     ("scales a", lambda: hip.absmax_scales(a, work, **ka)), ("scales b", lambda: hip.absmax_scales(b, work, **kb))):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        res.append("%s %.0f us" % (what, e0.elapsed_time(e1) * 100))
    print("%d x %d x %d km %d%d: %s" % (m, n, k, akm, bkm, ", ".join(res)), flush=True)
run(9728, 512, 10000, False, True); run(10000, 512, 9728, True, True); run(2048, 512, 9728, True, True); run(9728, 512, 2048, False, True); run(512, 512, 9728, True, True)
