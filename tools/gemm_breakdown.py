"""The GEMM launches of one A2C minibatch by shape: every launch of the minibatch on ONE stream (nothing beside
it), HIP events around each, grouped by (M, N, K, operand layouts, tile, k splits).

    python tools/gemm_breakdown.py [f32|bf16]
"""
import importlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

PKG = bench.PKG
syn = importlib.import_module(PKG + ".synthetic")
models = importlib.import_module(PKG + ".models")
optim = importlib.import_module(PKG + ".optim")
trainers = importlib.import_module(PKG + ".trainers")
hip = importlib.import_module(PKG + ".hipabi")

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
dev = torch.device("cuda", 0)
n, vocab, caplen, fdim = 512, 10000, 20, 2048
a2c, rew, flat, params = bench.build_networks(syn, models, optim, vocab, fdim, 0, dev)
opt = optim.FusedAdam(flat, lr=1e-4)
engine = trainers.build_engine(a2c, rew, concurrent=False, policy_dtype=dtype)
feats, caps = syn.make_minibatch(n, vocab, caplen, fdim, 0)
fd, pd = torch.from_numpy(feats).to(dev), torch.from_numpy(caps[:, :1]).to(dev)
u = torch.from_numpy(np.random.RandomState(1).random_sample((4, 19 * n))).to(dev)

shapes = []   # one entry per GEMM launch, in launch order
_gemm, _splitk = hip.gemm, hip.gemm_splitk


def gemm(a, b, out, **kw):
    m = kw.get("m") or (a.shape[1] if kw.get("a_kmajor") else (kw["a_rows"].numel() if kw.get("a_rows") is not None else a.shape[0]))
    k = kw.get("k") or (a.shape[0] if kw.get("a_kmajor") else a.shape[1])
    nn = kw.get("n") or (b.shape[1] if kw.get("b_kmajor") else b.shape[0])
    if hip.PROFILE is not None:
        shapes.append((m, nn, k, int(bool(kw.get("a_kmajor"))), int(bool(kw.get("b_kmajor"))), kw.get("tile", 0), 1))
    return _gemm(a, b, out, **kw)


def gemm_splitk(a, b, slabs, splits, **kw):
    m = a.shape[1] if kw.get("a_kmajor") else a.shape[0]
    k = a.shape[0] if kw.get("a_kmajor") else a.shape[1]
    nn = b.shape[1] if kw.get("b_kmajor") else b.shape[0]
    if hip.PROFILE is not None:
        shapes.append((m, nn, k, int(bool(kw.get("a_kmajor"))), int(bool(kw.get("b_kmajor"))), kw.get("tile", 0), splits))
    return _splitk(a, b, slabs, splits, **kw)


hip.gemm, hip.gemm_splitk = gemm, gemm_splitk
for i in range(4):
    if i == 3:
        hip.PROFILE = []
    engine.step(fd, pd, 19, uniforms=u[i]).check()
    opt.step()
    torch.cuda.synchronize()
prof, hip.PROFILE = hip.PROFILE, None
launches = [r for r in prof if r[0].startswith("gemm")]
assert len(launches) == len(shapes), (len(launches), len(shapes))
peak = 157.3 if dtype == "f32" else 2500.0
groups = {}
for shape, r in zip(shapes, launches):
    g = groups.setdefault(shape, [0, 0.0, 0.0])
    g[0] += 1
    g[1] += r[2].elapsed_time(r[3]) * 1e-3
    g[2] += r[1]
print("%6s %6s %6s  kA kB tile split   x   us each   total us   TFLOP/s  of f32 peak" % ("M", "N", "K"))
tot_s = tot_f = 0.0
for shape, (cnt, secs, flops) in sorted(groups.items(), key=lambda kv: -kv[1][1]):
    m, nn, k, ka, kb, tile, splits = shape
    print("%6d %6d %6d  %2d %2d %4d %5d %3d %9.1f %10.1f %9.1f %8.2f" % (m, nn, k, ka, kb, tile, splits, cnt, 1e6 * secs / cnt,
                                                                    1e6 * secs, flops / secs / 1e12,
                                                                    flops / secs / 1e12 / 157.3))
    tot_s += secs
    tot_f += flops
print("all %d launches: %.2f ms, %.1f GFLOP, %.1f TFLOP/s" % (len(launches), 1e3 * tot_s, tot_f / 1e9, tot_f / tot_s / 1e12))
others = {}
for r in prof:
    if not r[0].startswith("gemm"):
        o = others.setdefault(r[0], [0, 0.0, 0])
        o[0] += 1
        o[1] += r[2].elapsed_time(r[3]) * 1e-3
        o[2] += r[5]
for kind, (cnt, secs, steps) in others.items():
    print("%-20s x%d %.2f ms (%d steps)" % (kind, cnt, 1e3 * secs, steps))
