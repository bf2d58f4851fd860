"""Per-phase GPU timeline of one A2C minibatch at the bench workload (HIP events on the launching streams).

    python tools/phase_timeline.py [f32|bf16] [minibatches before the traced one, default 5] [f32|f32x3 gradient products] [late policy BPTT 1|0]
"""
import sys, importlib, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
PKG = bench.PKG
syn = importlib.import_module(PKG + ".synthetic"); models = importlib.import_module(PKG + ".models")
optim = importlib.import_module(PKG + ".optim"); trainers = importlib.import_module(PKG + ".trainers")
dev = torch.device("cuda", 0)
n, vocab, caplen, fdim = 512, 10000, 20, 2048
a2c, rew, flat, params = bench.build_networks(syn, models, optim, vocab, fdim, 0, dev)
opt = optim.FusedAdam(flat, lr=1e-4)
opts = {"grad_products": sys.argv[3]} if len(sys.argv) > 3 else {}
if len(sys.argv) > 4:   # 0: the policy LSTM's BPTT as 3 T launches on the side stream (the earlier schedule)
    opts["late_policy_bptt"] = bool(int(sys.argv[4]))
engine = trainers.build_engine(a2c, rew, policy_dtype=sys.argv[1] if len(sys.argv) > 1 else "f32", **opts)
feats, caps = syn.make_minibatch(n, vocab, caplen, fdim, 0)
fd = torch.from_numpy(feats).to(dev); pd = torch.from_numpy(caps[:, :1]).to(dev)
before = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rs = np.random.RandomState(1)
u = [torch.from_numpy(rs.random_sample(19 * n)).to(dev) for _ in range(before + 1)]
import time
for i in range(before + 1):
    if i == before: engine.trace, engine.host_trace = [], []
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = engine.step(fd, pd, 19, uniforms=u[i]); t1 = time.perf_counter()
    opt.step(); res.check(); torch.cuda.synchronize(); t2 = time.perf_counter()
    if i < 5 or i >= before - 2: print("step %d: host enqueue+seam sync %.2f ms, total %.2f ms, plan %s, fallbacks %d" % (i, 1e3 * (t1 - t0), 1e3 * (t2 - t0), res.chain_plan, engine.fallbacks))
tr = engine.trace
t0 = tr[0][1]
h0 = engine.host_trace[0][1]
print("%-28s %11s %14s" % ("", "device", "host got there"))
for (label, ev), (_, h) in zip(tr, engine.host_trace):
    print("%-28s %8.2f ms %11.2f ms" % (label, t0.elapsed_time(ev), 1e3 * (h - h0)))
