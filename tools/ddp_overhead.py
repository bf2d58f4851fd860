"""Cost of the gradient exchange code path on ONE GPU (single-rank process group, every collective issued):
step time without any exchange, with one all-reduce after the step, and with the two-piece exchange
whose policy part overlaps the value chain's backward pass.  The last mode puts a hostile neighbour on
the card: a fourth stream that runs device-filling elementwise kernels back to back for the whole step
(a single-rank collective launches no kernel of note, so this stands in for the collective's kernels on a
multi-GPU node, and is harsher: it asks for every CU).  The persistent chain kernels need all their
workgroups resident at once; the count of launches that gave up waiting is printed per mode.

    ICRL_FORCE_COLLECTIVES=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
        --master-addr 127.0.0.1 --master-port 29511 tools/ddp_overhead.py
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
ddp = importlib.import_module(PKG + ".dist")

rank, world, local = ddp.init_from_env("nccl")
dev = torch.device("cuda", local)
torch.cuda.set_device(local)
n, vocab, caplen, fdim = 512, 10000, 20, 2048
feats, caps = syn.make_minibatch(n, vocab, caplen, fdim, 0)
fd, pd = torch.from_numpy(feats).to(dev), torch.from_numpy(caps[:, :1]).to(dev)
steps = 10
u = torch.from_numpy(np.random.RandomState(1000).random_sample((steps, 19 * n))).to(dev)
neighbour_stream = torch.cuda.Stream(device=dev)
neighbour_buf = torch.ones(64 << 20, device=dev)   # 256 MB read + written per kernel: ~0.1 ms of every CU
for mode in ("none", "single", "two-piece", "two-piece+neighbour"):
    a2c, rew, flat, params = bench.build_networks(syn, models, optim, vocab, fdim, 0, dev)
    opt = optim.FusedAdam(flat, lr=1e-4)
    engine = trainers.build_engine(a2c, rew)
    averager = ddp.GradientAverager(flat.flat_grad, flat.span("policy_network."))
    if mode.startswith("two-piece"):
        engine.on_policy_grads = averager.start_early
    times = []
    for i in range(steps):
        torch.cuda.synchronize()
        if i == 6 and os.environ.get("ICRL_TRACE") == "1":
            engine.trace = []
        t0 = time.perf_counter()
        if mode.endswith("neighbour"):
            with torch.cuda.stream(neighbour_stream):
                for _ in range(120):
                    neighbour_buf.mul_(1.0)
        res = engine.step(fd, pd, 19, uniforms=u[i])
        if mode != "none":
            averager.finish()
        opt.step()
        res.check()
        torch.cuda.synchronize()
        times.append(1e3 * (time.perf_counter() - t0))
        if engine.trace:
            t_first = engine.trace[0][1]
            print("   ".join("%s %.2f" % (label, t_first.elapsed_time(ev)) for label, ev in engine.trace), flush=True)
            engine.trace = None
    print("%-20s ms/step (steps 2..): %.2f   launches that gave up waiting: %d, redone: %d   %s"
          % (mode, float(np.mean(times[2:])), engine.launch_timeouts, engine.fallbacks, " ".join("%.1f" % t for t in times)),
          flush=True)
if torch.distributed.is_initialized():
    torch.distributed.destroy_process_group()
