"""One backward-pass product alone, four launches (for counter passes: tools/gemm_counters.sh):

    python tools/one_gemm.py [path of another build of the library]
"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
hip = importlib.import_module(bench.PKG + ".hipabi")
if len(sys.argv) > 1: hip.LIB_PATH = os.path.abspath(sys.argv[1])
dev = torch.device("cuda", 0)
m, n, k = 9728, 512, 10000
a = torch.randn(m, k, device=dev); b = torch.randn(k, n, device=dev)
out = torch.empty(m, n, device=dev); scratch = torch.empty(8 * m * n, device=dev)
work = torch.zeros(max(m, n) + k, dtype=torch.int32, device=dev)
sa, sb = torch.zeros(m, device=dev), torch.zeros(n, device=dev)
hip.absmax_scales(a, work, row_scale=sa); hip.absmax_scales(b, work, col_scale=sb)
for _ in range(4):
    hip.gemm_auto(a, b, out, scratch, b_kmajor=True, shared=True, dtype="f16x2", a_scale=sa, b_scale=sb)
torch.cuda.synchronize()
