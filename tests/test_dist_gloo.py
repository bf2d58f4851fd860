"""The N > 1 path on CPU: two gloo ranks, flat-gradient averaging, parameter broadcast,
row sharding, max-over-ranks timing.  No GPU needed."""
import importlib
import os
import sys

import torch
import torch.multiprocessing as mp

from helpers import ROOT


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    ddp = importlib.import_module("image-captioning-through-rl_amd.dist")
    optim = importlib.import_module("image-captioning-through-rl_amd.optim")
    r, w, _ = ddp.init_from_env("gloo")
    assert (r, w) == (rank, world) and ddp.world_size() == world
    # every rank starts from rank 0's parameters
    torch.manual_seed(100 + rank)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    flat = optim.flatten_module(net)
    ddp.broadcast_parameters(flat.flat)
    # rank-dependent gradients, written through the per-parameter views
    for i, p in enumerate(net.parameters()):
        p.grad.fill_(float(rank + 1) * (i + 1))
    ddp.average_gradients(flat.flat_grad)
    # the same average in two pieces (the second layer's gradients first), as the training loop does it
    span = flat.span("1.")
    pad = flat.flat_grad.numel()
    two = torch.zeros(pad)
    for i, (p, off) in enumerate(zip(net.parameters(), flat.offsets)):
        two[off:off + p.numel()] = float(rank + 1) * (i + 1)
    avg = ddp.GradientAverager(two, span)
    avg.start_early()
    avg.finish()
    assert torch.equal(two, flat.flat_grad), "two-piece average differs from the single all-reduce"
    whole = ddp.GradientAverager(two.clone(), span)
    whole.finish()                                   # start_early never called: one all-reduce of everything
    assert torch.allclose(whole.flat_grad, two)      # already equal on all ranks: the mean changes nothing
    # the training loops' data side: each rank its own rows of every global minibatch, decisions taken together
    import numpy as np
    utilities = importlib.import_module("image-captioning-through-rl_amd.utilities")
    data = {"train_captions": np.arange(11 * 3).reshape(11, 3), "train_image_idxs": np.arange(11)[::-1].copy(),
            "train_features": np.arange(11, dtype=np.float32)[:, None] * np.ones((1, 4), dtype=np.float32),
            "train_urls": np.array(["u%d" % i for i in range(11)])}
    torch.manual_seed(7 + rank)                      # different generators: rank 0's permutation must win
    rows = [(c[:, 0] // 3, f[:, 0], u) for c, f, u in utilities.minibatches(data, batch_size=2)]
    assert [len(r[0]) for r in rows] == [2, 2, 1]    # 11 rows = global minibatches of 4, 4 and 3 (split 1 + 1, one row dropped)
    for cap_rows, feats, urls in rows:               # features and urls follow image_idxs, as in get_coco_minibatches
        assert np.array_equal(feats, 10.0 - cap_rows) and list(urls) == ["u%d" % (10 - r) for r in cap_rows]
    assert ddp.rank() == rank and ddp.is_main() == (rank == 0)
    assert ddp.agree_min(5 - 4 * rank) == 1          # rank 1 holds the smallest value
    np.random.seed(3)
    ddp.decorrelate_numpy_stream()
    draw = float(np.random.random_sample())
    lo, hi = ddp.shard_rows(1024)
    t = ddp.max_over_ranks(1.0 + rank, torch.device("cpu"))
    torch.save(dict(flat=flat.flat.clone(), grads=[p.grad.clone() for p in net.parameters()], shard=(lo, hi), t=t,
                    keys=list(net.state_dict().keys()), rows=np.concatenate([r[0] for r in rows]), draw=draw),
               os.path.join(out_dir, "r%d.pt" % rank))
    torch.distributed.destroy_process_group()


def test_two_rank_gradient_average_and_broadcast(tmp_path):
    world, port = 2, 29500 + os.getpid() % 1000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "r0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "r1.pt", weights_only=False)
    assert torch.equal(r0["flat"], r1["flat"]), "parameters differ after broadcast"
    for i, (g0, g1) in enumerate(zip(r0["grads"], r1["grads"])):
        want = (1 + 2) / 2.0 * (i + 1)  # mean over ranks of (rank+1)*(i+1)
        assert torch.allclose(g0, torch.full_like(g0, want)) and torch.equal(g0, g1)
    assert r0["shard"] == (0, 512) and r1["shard"] == (512, 1024)
    assert r0["t"] == r1["t"] == 2.0
    assert r0["keys"] == ["0.weight", "0.bias", "1.weight", "1.bias"]
    # ten of the eleven rows, each on exactly one rank, in the order of rank 0's permutation
    both = list(r0["rows"]) + list(r1["rows"])
    assert len(both) == 10 and len(set(both)) == 10
    torch.manual_seed(7)
    perm = torch.randperm(11).tolist()
    assert list(r0["rows"]) == perm[0:2] + perm[4:6] + perm[8:9] and list(r1["rows"]) == perm[2:4] + perm[6:8] + perm[9:10]
    assert r0["draw"] != r1["draw"]
