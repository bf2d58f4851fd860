"""Time the beam look-ahead decoder (trainers.GenerateCaptionsWithActorCriticLookAhead) on one MI355X
at the reference's validation batch (127 rows, trainers.py:642), vocab 10k, 2048-d features, next to the
float64 CPU restatement on a few rows (its cost is linear in the rows: one batch-1 chain step per row,
column and value call)."""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

pkg = bench.PKG
syn = importlib.import_module(pkg + ".synthetic")
models = importlib.import_module(pkg + ".models")
optim = importlib.import_module(pkg + ".optim")
trainers = importlib.import_module(pkg + ".trainers")

n, vocab, fdim = 127, 10000, 2048
dev = torch.device("cuda", 0)
a2c, rew, flat, params = bench.build_networks(syn, models, optim, vocab, fdim, 0, dev)
feats, caps = syn.make_minibatch(n, vocab, 17, fdim, 0)
for rep in range(2):
    a2c.value_network.valrnn.init_hidden()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    best = trainers.GenerateCaptionsWithActorCriticLookAhead(feats, caps, a2c.policy_network, a2c.value_network,
                                                             most_likely=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("HIP: %d rows decoded in %.3f s = %.1f images/s (run %d)" % (n, dt, n / dt, rep), flush=True)

if "--cpu" in sys.argv:
    from oracle import a2c_oracle as orc
    rows = 4
    torch.set_num_threads(16)
    t0 = time.perf_counter()
    orc.lookahead_decode(params[0], params[1], feats[:rows], caps[:rows])
    dt = time.perf_counter() - t0
    print("CPU float64 restatement: %d rows in %.1f s = %.2f images/s (%d threads)" % (rows, dt, rows / dt,
                                                                                      torch.get_num_threads()))
