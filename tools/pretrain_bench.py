"""Time one minibatch of each pre-training loop (policy cross entropy, value MSE, reward VSE) on one
MI355X: batch 512, vocab 10k, 2048-d features, captions of 17 columns."""
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
engine_mod = importlib.import_module(pkg + ".engine")

n, vocab, fdim = 512, 10000, 2048
dev = torch.device("cuda", 0)
a2c, rew, flat, params = bench.build_networks(syn, models, optim, vocab, fdim, 0, dev)
rew.requires_grad_(True)
data = syn.make_ragged_train_data(n, vocab, fdim, 0)
caps = torch.from_numpy(data["train_captions"]).to(dev)
feats = torch.from_numpy(data["train_features"]).to(dev)
valid, weight = trainers._ce_weights(data["train_captions"], 16)
valid, weight = torch.from_numpy(valid).to(dev), torch.from_numpy(weight).to(dev)
pol, val = a2c.policy_network, a2c.value_network
e_pol = engine_mod.A2CEngine(trainers._params_of(pol), None, None, trainers._grads_of(pol))
e_val = engine_mod.A2CEngine(None, trainers._params_of(val), None, None, trainers._grads_of(val))
e_rew = engine_mod.A2CEngine(None, None, trainers._params_of(rew), reward_grads=trainers._grads_of(rew))


def value_step():
    gen = trainers.GenerateCaptionsGreedy(feats, caps, pol)
    rewards = trainers.GetRewards(feats, gen, rew)
    rew.rewrnn.init_hidden()
    return e_val.value_mse_step(feats, gen[:, :9].contiguous(), rewards.view(-1))[0]


for name, fn in (("policy cross-entropy step", lambda: e_pol.policy_ce_step(feats, caps, valid, weight)),
                 ("value MSE step (greedy decode + reward + 9-column value chain)", value_step),
                 ("reward VSE step (17-column GRU chain fwd + bwd)", lambda: e_rew.reward_vse_step(feats, caps))):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        loss = fn()
    loss = float(loss)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print("%-66s %8.2f ms  %8.0f images/s  (loss %.4f)" % (name, 1e3 * dt, n / dt, loss), flush=True)
