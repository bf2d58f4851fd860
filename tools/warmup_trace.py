"""Per-minibatch trace of the warm-up ladders on the bench workload: accepted warm-up lengths, seam errors
as fractions of their tolerances, redone minibatches and step time.

    python tools/warmup_trace.py [steps] [first warm-up length]
"""
import importlib
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

PKG = bench.PKG
syn = importlib.import_module(PKG + ".synthetic")
models = importlib.import_module(PKG + ".models")
optim = importlib.import_module(PKG + ".optim")
trainers = importlib.import_module(PKG + ".trainers")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
n, vocab, caplen, fdim = 512, 10000, 20, 2048
a2c, rew, flat, params = bench.build_networks(syn, models, optim, vocab, fdim, 0, dev)
opt = optim.FusedAdam(flat, lr=1e-4)
engine = trainers.build_engine(a2c, rew, **({'first_warmup': int(sys.argv[2])} if len(sys.argv) > 2 else {}))
feats, caps = syn.make_minibatch(n, vocab, caplen, fdim, 0)
fd, pd = torch.from_numpy(feats).to(dev), torch.from_numpy(caps[:, :1]).to(dev)
u = torch.from_numpy(np.random.RandomState(1000).random_sample((steps, 19 * n))).to(dev)
total = 0.0
for i in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    before = engine.fallbacks
    res = engine.step(fd, pd, 19, uniforms=u[i])
    opt.step()
    res.check()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0)
    total += ms if i >= 2 else 0.0
    s = engine.last_seam
    print("step %3d  W value fwd %4d (%.2f)  reward fwd %4d (%.2f)  value bwd %4d (%.2f)  redone %d  %6.2f ms" % (
        i, res.chain_plan[0][2], s[0] / engine.SEAM_TOL_FWD, res.reward_plan[2], s[2] / engine.SEAM_TOL_FWD,
        res.chain_plan[1][2], (s[4] / s[5] if s[5] > 0 else 0.0) / engine.SEAM_TOL_BWD, engine.fallbacks - before, ms) + "  fc %d replay %s warm16 %s" % (res.full_calls, engine.last_replay, engine.last_cheap),
        flush=True)
print("mean over steps 2.. : %.2f ms" % (total / max(1, steps - 2)))
