"""Forty A2C minibatches of freshly initialised networks at the bench workload, one line each: step time, the warm-up
lengths the three chains ran with, redone attempts so far, the seam errors.  With an argument the chains' arithmetic:

    python tools/warmup_drift.py [f32|f16x2]
"""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
PKG = bench.PKG
syn = importlib.import_module(PKG + ".synthetic"); models = importlib.import_module(PKG + ".models")
optim = importlib.import_module(PKG + ".optim"); trainers = importlib.import_module(PKG + ".trainers")
dev = torch.device("cuda", 0)
n, vocab, caplen, fdim = 512, 10000, 20, 2048
feats, caps = syn.make_minibatch(n, vocab, caplen, fdim, 0)
fd, pd = torch.from_numpy(feats).to(dev), torch.from_numpy(caps[:, :1]).to(dev)
steps = 40
u = torch.from_numpy(np.random.RandomState(1000).random_sample((steps, 19 * n))).to(dev)
a2c, rew, flat, params = bench.build_networks(syn, models, optim, vocab, fdim, 0, dev)
opt = optim.FusedAdam(flat, lr=1e-4)
engine = trainers.build_engine(a2c, rew, **({"chain_arith": sys.argv[1]} if len(sys.argv) > 1 else {}))
for i in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = engine.step(fd, pd, 19, uniforms=u[i]); opt.step(); res.check()
    torch.cuda.synchronize()
    print("step %2d %.2f ms  fwd %s rew %s bwd %s fallbacks %d seam %s" % (i, 1e3 * (time.perf_counter() - t0), res.chain_plan[0][2], res.reward_plan[2], res.chain_plan[1][2], engine.fallbacks, ["%.2g" % x for x in engine.last_seam]), flush=True)
