"""Host against device through consecutive A2C minibatches run back to back, as the bench runs them (no device-wide
synchronisation between minibatches): for every mark of the engine's trace, when the HOST queued it and when the
DEVICE got there, on one clock.  (The device's clock is tied to the host's at the end: an event recorded when the
device has drained.)  Shows whether a stretch of the step is bound by the host's launch rate or by the device.

    python tools/pipe_timeline.py [minibatches before the traced ones, default 8] [traced minibatches, default 2]
"""
import sys, os, importlib, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
PKG = bench.PKG
syn = importlib.import_module(PKG + ".synthetic"); models = importlib.import_module(PKG + ".models")
optim = importlib.import_module(PKG + ".optim"); trainers = importlib.import_module(PKG + ".trainers")
dev = torch.device("cuda", 0)
n, vocab, caplen, fdim = 512, 10000, 20, 2048
a2c, rew, flat, params = bench.build_networks(syn, models, optim, vocab, fdim, 0, dev, 0)
opt = optim.FusedAdam(flat, lr=1e-4)
engine = trainers.build_engine(a2c, rew)
feats, caps = syn.make_minibatch(n, vocab, caplen, fdim, 0)
fd = torch.from_numpy(feats).to(dev); pd = torch.from_numpy(caps[:, :1]).to(dev)
before = int(sys.argv[1]) if len(sys.argv) > 1 else 8
traced = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rs = np.random.RandomState(1)
u = [torch.from_numpy(rs.random_sample(19 * n)).to(dev) for _ in range(before + traced)]
torch.cuda.synchronize()
for i in range(before + traced):
    if i == before:
        engine.trace, engine.host_trace = [], []
    res = engine.step(fd, pd, 19, uniforms=u[i])
    opt.step()
    if engine.trace is not None:
        engine._mark("adam_queued")
    res.check()
    if engine.trace is not None:
        engine.trace.append(("returned_to_caller", None)); engine.host_trace.append(("returned_to_caller", time.perf_counter()))
tr, ht = engine.trace, engine.host_trace
engine.trace = engine.host_trace = None
torch.cuda.synchronize()
end = torch.cuda.Event(enable_timing=True); end.record(); end.synchronize(); h_end = time.perf_counter()
h0 = ht[0][1]
print("%-30s %12s %12s %10s" % ("mark", "host queued", "device there", "host lead"))
for (label, ev), (_, h) in zip(tr, ht):
    if ev is None:
        print("%-30s %9.3f ms" % (label, 1e3 * (h - h0)))
        continue
    d = h_end - 1e-3 * ev.elapsed_time(end)
    print("%-30s %9.3f ms %9.3f ms %7.3f ms" % (label, 1e3 * (h - h0), 1e3 * (d - h0), 1e3 * (d - h)))
