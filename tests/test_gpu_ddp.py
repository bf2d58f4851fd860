"""Two ranks on ONE GPU (gloo moves the CUDA gradient buffers through the host): the whole data-parallel
minibatch - per-rank rollout, two-piece gradient average started from the engine's policy-gradients
callback, fused Adam - against the same thing computed rank by rank in one process.  Needs an MI355X."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from helpers import ROOT, params_for, synthetic

pytestmark = pytest.mark.gpu

CFG = dict(n=6, vocab=120, caplen=7, fdim=512, seed=11, level=None)


def _build(models, optim, trainers):
    w2i = synthetic.make_word_to_idx(CFG["vocab"])
    pol = models.PolicyNetwork(w2i, input_dim=CFG["fdim"])
    val = models.ValueNetwork(w2i, feature_dim=CFG["fdim"])
    rew = models.RewardNetwork(w2i, feature_dim=CFG["fdim"])
    for net, params in zip((pol, val, rew), params_for(CFG)):
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
        net.cuda()
    rew.requires_grad_(False)
    a2c = models.AdvantageActorCriticNetwork(val, pol).cuda()
    flat = optim.flatten_module(a2c)
    return a2c, rew, flat, optim.FusedAdam(flat, lr=1e-4), trainers.build_engine(a2c, rew)


def _rank_inputs(rank):
    n, steps = CFG["n"], CFG["caplen"] - 1
    feats, caps = synthetic.make_minibatch(n, CFG["vocab"], CFG["caplen"], CFG["fdim"], 100 + rank)
    u = synthetic.make_uniforms(200 + rank, steps * n)
    return (torch.from_numpy(feats).cuda(), torch.from_numpy(caps[:, :1]).cuda(), steps, torch.from_numpy(u).cuda())


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    pkg = "image-captioning-through-rl_amd."
    ddp, models = importlib.import_module(pkg + "dist"), importlib.import_module(pkg + "models")
    optim, trainers = importlib.import_module(pkg + "optim"), importlib.import_module(pkg + "trainers")
    ddp.init_from_env("gloo")
    torch.cuda.set_device(0)
    a2c, rew, flat, opt, engine = _build(models, optim, trainers)
    if rank == 1:   # the broadcast must undo this
        flat.flat.mul_(1.5)
    ddp.broadcast_parameters(flat.flat)
    averager = ddp.GradientAverager(flat.flat_grad, flat.span("policy_network."))
    fired = []
    engine.on_policy_grads = lambda ev: (fired.append(1), averager.start_early(ev))
    feats, prefix, steps, u = _rank_inputs(rank)
    res = engine.step(feats, prefix, steps, uniforms=u).check()
    averager.finish()
    opt.step()
    torch.cuda.synchronize()
    torch.save(dict(flat=flat.flat.cpu(), grad=flat.flat_grad.cpu(), fired=len(fired), loss=float(res.scalars[0])),
               os.path.join(out_dir, "r%d.pt" % rank))
    torch.distributed.destroy_process_group()


def test_two_ranks_average_their_gradients_and_stay_in_step(tmp_path):
    world, port = 2, 29600 + os.getpid() % 300
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = [torch.load(tmp_path / ("r%d.pt" % r)) for r in range(world)]
    assert got[0]["fired"] == got[1]["fired"] == 1
    assert torch.equal(got[0]["flat"], got[1]["flat"]), "ranks diverged"
    assert torch.equal(got[0]["grad"], got[1]["grad"])
    # the same in one process: each shard's gradients from the common parameters, their mean, one Adam step
    pkg = "image-captioning-through-rl_amd."
    models, optim = importlib.import_module(pkg + "models"), importlib.import_module(pkg + "optim")
    trainers = importlib.import_module(pkg + "trainers")
    a2c, rew, flat, opt, engine = _build(models, optim, trainers)
    grads, losses = [], []
    for rank in range(world):
        feats, prefix, steps, u = _rank_inputs(rank)
        res = engine.step(feats, prefix, steps, uniforms=u).check()
        grads.append(flat.flat_grad.clone())
        losses.append(float(res.scalars[0]))
    flat.flat_grad.copy_((grads[0] + grads[1]) / 2)
    opt.step()
    torch.cuda.synchronize()
    assert got[0]["loss"] == pytest.approx(losses[0], rel=1e-5) and got[1]["loss"] == pytest.approx(losses[1], rel=1e-5)
    scale = flat.flat_grad.abs().max().item()
    assert (got[0]["grad"] - flat.flat_grad.cpu()).abs().max().item() <= 1e-5 * scale
    np.testing.assert_allclose(got[0]["flat"].numpy(), flat.flat.cpu().numpy(), rtol=1e-5, atol=2e-6)
