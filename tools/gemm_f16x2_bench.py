"""The pre-cut f16 product (hipabi.gemm_f16x2) against the exact-f32 kernel on the rollout's, a batch-64 and the per-word
tables' shapes: error of both against the float64 product, time per launch (back-to-back launches, HIP events), and the
time of the two cutters.

    python tools/gemm_f16x2_bench.py
"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
hip = importlib.import_module(bench.PKG + ".hipabi")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
def run(m, n, k, time_it):
    a = (torch.rand(m, k, device=dev) * 2 - 1) * torch.rand(m, 1, device=dev)
    b = torch.randn(n, k, device=dev) * 0.05 * torch.exp(torch.randn(n, 1, device=dev))
    bias = torch.randn(n, device=dev)
    out = torch.empty(m, n, device=dev); ref32 = torch.empty(m, n, device=dev)
    ap, ad = hip.f16x2_pieces(m, k, dev); bp, bd = hip.f16x2_pieces(n, k, dev)
    hip.cut_f16x2(a, ap, ad, fixed_scale=4096.0); hip.cut_f16x2(b, bp, bd)
    hip.gemm_f16x2(ap, ad, bp, bd, out, k, bias=bias)
    hip.gemm(a, b, ref32, bias=bias)
    want = a.double() @ b.double().t() + bias.double()
    scale = (a.double().abs() @ b.double().abs().t()) + bias.double().abs()
    e16 = ((out.double() - want).abs() / scale).max().item(); e32 = ((ref32.double() - want).abs() / scale).max().item()
    print("%d x %d x %d: max |err| / sum|a||b|: f16x2 %.3g, f32 kernel %.3g" % (m, n, k, e16, e32), flush=True)
    if time_it:
        for what, fn in (("f16x2", lambda: hip.gemm_f16x2(ap, ad, bp, bd, out, k, bias=bias)),
                         ("cut a", lambda: hip.cut_f16x2(a, ap, ad, fixed_scale=4096.0)),
                         ("cut b", lambda: hip.cut_f16x2(b, bp, bd)),
                         ("f32", lambda: hip.gemm(a, b, ref32, bias=bias, tile=3))):
            for _ in range(5): fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(50): fn()
            e1.record(); torch.cuda.synchronize()
            print("   %s %.1f us" % (what, e0.elapsed_time(e1) * 20), flush=True)
run(512, 12096, 512, True)
run(64, 12096, 512, True)
run(4, 100 + 2048, 512, False)
run(70, 1000, 48, False)
run(10000, 2048, 512, True)
