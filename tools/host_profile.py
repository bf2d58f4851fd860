"""Where the HOST's time goes while it queues one A2C minibatch (cProfile over a few steps)."""
import cProfile, importlib, os, pstats, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
PKG = bench.PKG
syn = importlib.import_module(PKG + ".synthetic"); models = importlib.import_module(PKG + ".models")
optim = importlib.import_module(PKG + ".optim"); trainers = importlib.import_module(PKG + ".trainers")
dev = torch.device("cuda", 0)
n, vocab, caplen, fdim = 512, 10000, 20, 2048
a2c, rew, flat, params = bench.build_networks(syn, models, optim, vocab, fdim, 0, dev)
opt = optim.FusedAdam(flat, lr=1e-4)
engine = trainers.build_engine(a2c, rew)
feats, caps = syn.make_minibatch(n, vocab, caplen, fdim, 0)
fd, pd = torch.from_numpy(feats).to(dev), torch.from_numpy(caps[:, :1]).to(dev)
u = torch.from_numpy(np.random.RandomState(1).random_sample((12, 19 * n))).to(dev)
for i in range(4):
    engine.step(fd, pd, 19, uniforms=u[i]); opt.step()
torch.cuda.synchronize()
# host time of the queueing alone: time each phase's enqueue with the device idle-waiting excluded
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for i in range(4, 12):
    engine.step(fd, pd, 19, uniforms=u[i]); opt.step()
pr.disable()
torch.cuda.synchronize()
print("wall per step %.2f ms" % ((time.perf_counter() - t0) / 8 * 1e3))
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
