"""Chains whose later calls leave out the replay of the earlier columns (full_calls < n_calls), and the check
that goes with them.

Call s of the reference's value / reward network walks caption columns 0 .. p0+s-1 (models.py:168-169,
254-255): call s+1 replays every input of call s before it reads its new column.  The kernels can walk calls
0 .. fc-1 in full and then one column per later call.  Here: (a) that shorter chain, forward and backward, against
a float64 restatement of exactly that chain; (b) the replay check against the exact chain kernels run over the
replays it stands for; (c) a whole A2C minibatch with replays left out against the same minibatch with every
call walked in full on the exact kernels, and the fall-back when a network does not forget."""
import importlib

import numpy as np
import pytest
import torch

from helpers import params_for, synthetic
from oracle import kernel_refs as ref

hip = importlib.import_module("image-captioning-through-rl_amd.hipabi")
models = importlib.import_module("image-captioning-through-rl_amd.models")
trainers = importlib.import_module("image-captioning-through-rl_amd.trainers")
pytestmark = pytest.mark.gpu
H = 512


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _inputs(rs, rows, gates, forget_shift=0.0):
    xproj = (0.5 * rs.standard_normal((rows, gates * H))).astype(np.float32)
    if forget_shift:
        xproj[:, H:2 * H] += forget_shift      # LSTM forget gate / GRU update gate towards 1: a long memory
    w_hh = (rs.uniform(-1, 1, size=(gates * H, H)) / np.sqrt(H)).astype(np.float32)
    return xproj, w_hh


def _positions(n, p0, calls, fc):
    """[(call, column, row)] of the chain with fc calls in full."""
    return [(s, t, i) for s in range(calls) for t in ref.call_columns(p0, s, fc) for i in range(n)]


@pytest.mark.parametrize("n,p0,calls,fc,segs,warm", [(16, 2, 5, 1, 3, 40), (16, 2, 5, 2, 3, 40), (16, 1, 6, 3, 4, 40),
                                                      (8, 3, 4, 4, 2, 40), (32, 1, 7, 1, 33, 56)])
def test_chains_with_replays_left_out_match_their_float64_restatement(n, p0, calls, fc, segs, warm):
    rs = np.random.RandomState(100 * n + 10 * calls + fc)
    cols = p0 + calls
    xproj, w_hh = _inputs(rs, cols * n, 4)
    d_emit = (rs.standard_normal((calls, n, H))).astype(np.float32)
    positions = hip.chain_positions(n, p0, calls, fc)
    assert positions == len(_positions(n, p0, calls, fc))
    h_w, saved_w, dg_w = ref.lstm_chain(xproj, w_hh, n, p0, calls, want_grad_of=d_emit, full_calls=fc)
    xd, wd, dd = dev(xproj), dev(w_hh), dev(d_emit)
    persist_ok = hip.chain_pair_supported(segs)
    outs = []
    for persistent in ([False, True] if persist_ok else [False]):
        work, err = torch.zeros(hip.chain_seg_workspace_floats(segs), device="cuda"), torch.zeros(2, device="cuda")
        h = torch.zeros(calls, n, H, device="cuda")
        saved = torch.zeros(positions, 6, H, device="cuda")
        hip.lstm_chain_fwd_seg(xd, wd, n, p0, calls, segs, warm, h, saved, work, err, persistent=persistent, full_calls=fc)
        torch.cuda.synchronize()
        assert err[0].item() < 5e-6
        np.testing.assert_allclose(saved.cpu().numpy(), saved_w, rtol=2e-5, atol=2e-6)
        np.testing.assert_allclose(h.cpu().numpy(), h_w, rtol=2e-5, atol=2e-6)
        dg = torch.zeros(positions, 4 * H, device="cuda")
        hip.lstm_chain_bwd_seg(wd, saved, dd, n, p0, calls, segs, warm, dg, work, err, persistent=persistent, full_calls=fc)
        torch.cuda.synchronize()
        assert err[0].item() <= 2e-5 * err[1].item() + 1e-12
        np.testing.assert_allclose(dg.cpu().numpy(), dg_w, rtol=2e-4, atol=2e-5 * np.abs(dg_w).max())
        outs.append((h, saved, dg))
    if len(outs) == 2:   # one launch per step and the persistent launch: the same bits
        for a, b in zip(*outs):
            assert torch.equal(a, b)
    # the gate gradients summed over the calls that hold a column
    dg = outs[0][2]
    folded = torch.full(((p0 + calls - 1) * n, 4 * H), 7.0, device="cuda")
    hip.chain_fold_calls(dg, n, p0, calls, 4 * H, folded, full_calls=fc)
    want = np.zeros(((p0 + calls - 1) * n, 4 * H))
    for pos, (s, t, i) in enumerate(_positions(n, p0, calls, fc)):
        want[t * n + i] += dg_w[pos]
    np.testing.assert_allclose(folded.cpu().numpy(), want, rtol=2e-4, atol=4e-5 * np.abs(want).max())
    # the reward GRU's chain (one more column per call), alone and in the pair launch
    xg, wg = _inputs(rs, (cols + 1) * n, 3)
    bg = (0.1 * rs.standard_normal(3 * H)).astype(np.float32)
    g_w = ref.gru_chain(xg, wg, bg, n, p0, calls, full_calls=fc)
    work, err = torch.zeros(hip.chain_seg_workspace_floats(segs), device="cuda"), torch.zeros(2, device="cuda")
    g = torch.zeros(calls, n, H, device="cuda")
    pos_r = hip.chain_positions(n, p0 + 1, calls, fc)
    saved_r = torch.zeros(pos_r, 5, H, device="cuda")
    hip.gru_chain_fwd_seg(dev(xg), dev(wg), dev(bg), n, p0 + 1, calls, segs, warm, g, work, err, saved=saved_r,
                          full_calls=fc)
    torch.cuda.synchronize()
    assert err[0].item() < 5e-6
    np.testing.assert_allclose(g.cpu().numpy(), g_w, rtol=2e-5, atol=2e-6)
    if persist_ok:
        work_v, work_r = (torch.zeros(hip.chain_seg_workspace_floats(segs), device="cuda") for _ in range(2))
        err_v, err_r = torch.zeros(2, device="cuda"), torch.zeros(2, device="cuda")
        h2, g2 = torch.zeros(calls, n, H, device="cuda"), torch.zeros(calls, n, H, device="cuda")
        saved2, saved_r2 = torch.zeros(positions, 6, H, device="cuda"), torch.zeros(pos_r, 5, H, device="cuda")
        hip.chain_pair_fwd_seg(xd, wd, dev(xg), dev(wg), dev(bg), n, p0, p0 + 1, calls, segs, warm, h2, saved2, g2,
                               work_v, work_r, err_v, err_r, full_calls=fc, saved_r=saved_r2)
        torch.cuda.synchronize()
        assert torch.equal(h2, outs[0][0]) and torch.equal(saved2, outs[0][1])
        assert torch.equal(g2, g) and torch.equal(saved_r2, saved_r)


@pytest.mark.parametrize("arith", ["f32", "f16x2"])
@pytest.mark.parametrize("forget_shift", [0.0, 5.0])
def test_replay_check_measures_what_the_exact_kernels_compute(forget_shift, arith):
    """fc = 2 of 6 calls left in full: for calls 2..5 the check replays the first `steps` inputs from the state the
    chain had before the call.  The same replays, one at a time, on the exact chain kernels give the error it
    must report; with a long memory (forget / update gates near 1) that error is far from rounding."""
    rs = np.random.RandomState(5)
    n, p0, calls, fc, segs, warm, steps = 48, 1, 6, 2, 8, 64, 80
    xv, wv = _inputs(rs, (p0 + calls) * n, 4, forget_shift)
    xr, wr = _inputs(rs, (p0 + calls + 1) * n, 3, forget_shift)
    br = (0.1 * rs.standard_normal(3 * H)).astype(np.float32)
    xv, wv, xr, wr, br = dev(xv), dev(wv), dev(xr), dev(wr), dev(br)
    pos_v, pos_r = hip.chain_positions(n, p0, calls, fc), hip.chain_positions(n, p0 + 1, calls, fc)
    saved_v, saved_r = torch.zeros(pos_v, 6, H, device="cuda"), torch.zeros(pos_r, 5, H, device="cuda")
    hv, hr = torch.zeros(calls, n, H, device="cuda"), torch.zeros(calls, n, H, device="cuda")
    work_v, work_r = (torch.zeros(hip.chain_seg_workspace_floats(segs), device="cuda") for _ in range(2))
    err_v, err_r = torch.zeros(2, device="cuda"), torch.zeros(2, device="cuda")
    hip.chain_pair_fwd_seg(xv, wv, xr, wr, br, n, p0, p0 + 1, calls, segs, warm, hv, saved_v, hr, work_v, work_r,
                           err_v, err_r, full_calls=fc, saved_r=saved_r)
    rows = calls - fc
    cw_v, cw_r = (torch.zeros(hip.chain_seg_workspace_floats(rows), device="cuda") for _ in range(2))
    err = torch.full((2,), 7.0, device="cuda")
    hip.chain_pair_replay_check(xv, wv, saved_v, xr, wr, br, saved_r, n, p0, p0 + 1, calls, fc, steps, cw_v, cw_r, err,
                                arith=arith)
    torch.cuda.synchronize()
    # the same on the exact kernels: chain of `steps` positions (one call of one column of `steps` rows) from state_in
    sync, status = torch.zeros(hip.chain_sync_bytes(), dtype=torch.uint8, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda")
    want = [0.0, 0.0]
    for j in range(rows):
        for which, (x, w, saved, p0x, slots) in enumerate(((xv, wv, saved_v, p0, 6), (xr, wr, saved_r, p0 + 1, 5))):
            c0 = fc * p0x + fc * (fc - 1) // 2
            start = (c0 + j) * n - 1                                  # end of the call before call fc + j
            target = (c0 - (p0x + fc - 1)) * n + steps - 1            # the last full call, `steps` inputs in
            out = torch.zeros(1, steps, H, device="cuda")
            if which == 0:
                state = torch.cat((saved[start, 5], saved[start, 4])).contiguous()
                state_out = torch.zeros(2 * H, device="cuda")
                hip.lstm_chain_fwd(x, w, steps, 1, 1, out, None, sync, status, state_in=state, state_out=state_out)
                tgt = torch.cat((saved[target, 5], saved[target, 4]))
            else:
                state = saved[start, 4].contiguous()
                state_out = torch.zeros(H, device="cuda")
                hip.gru_chain_fwd(x, w, br, steps, 1, 1, out, sync, status, state_in=state, state_out=state_out)
                tgt = saved[target, 4]
            torch.cuda.synchronize()
            assert status.item() == 0
            want[which] = max(want[which], ((state_out - tgt).abs() / tgt.abs().clamp(min=1.0)).max().item())
    got = err.cpu().tolist()
    if forget_shift:
        assert min(want) > 1e-3, "the long-memory chains were meant not to forget in %d steps" % steps
        # (the f16 launch's products differ from the exact kernels' at float32 rounding level, and a long memory
        # carries such differences along: the same error to a few per cent)
        np.testing.assert_allclose(got, want, rtol=2e-3 if arith == "f32" else 5e-2)
    else:
        assert max(want) < 2e-6 and max(got) < (2e-6 if arith == "f32" else 4e-6), (got, want)


# --------------------------------------------------------------------------- whole minibatches
def _modules(cfg):
    w2i = synthetic.make_word_to_idx(cfg["vocab"])
    nets = (models.PolicyNetwork(w2i, input_dim=cfg["fdim"]), models.ValueNetwork(w2i, feature_dim=cfg["fdim"]),
            models.RewardNetwork(w2i, feature_dim=cfg["fdim"]))
    for net, params in zip(nets, params_for(cfg)):
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
        net.cuda()
    nets[2].requires_grad_(False)
    return models.AdvantageActorCriticNetwork(nets[1], nets[0]).cuda(), nets[2]


def _step(a2c, rew, feats, caps, p0, steps, u, before=None, **opts):
    engine = trainers.build_engine(a2c, rew, **opts)
    if before is not None:
        before(engine)
    res = engine.step(torch.from_numpy(feats).cuda(), torch.from_numpy(caps[:, :p0]).cuda().contiguous(), steps,
                      uniforms=torch.from_numpy(u).cuda()).check()
    torch.cuda.synchronize()
    return engine, res, {k: p.grad.clone() for k, p in a2c.named_parameters()}


def _same_step(res, grads, exact, g_exact, grad_tol=1e-3):
    assert torch.equal(res.tokens, exact.tokens)
    np.testing.assert_allclose(res.values.cpu().numpy(), exact.values.cpu().numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(res.rewards.cpu().numpy(), exact.rewards.cpu().numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(res.scalars.cpu().numpy()[:5], exact.scalars.cpu().numpy()[:5], rtol=1e-5, atol=1e-7)
    for k in g_exact:
        scale = g_exact[k].pow(2).mean().sqrt().item() + 1e-12
        assert (grads[k] - g_exact[k]).abs().max().item() <= grad_tol * scale + 1e-9, k


@pytest.mark.parametrize("n,caplen,p0", [(256, 9, 1), (192, 12, 4)])
def test_minibatch_with_replays_left_out_equals_every_call_in_full_on_the_exact_kernels(n, caplen, p0):
    """Plain A2C (p0 = 1) and a curriculum level (ground-truth prefix of 4 columns): values, rewards, scalars and
    every gradient of the step that walks one column per call against the step that walks every call in full."""
    cfg = dict(n=n, vocab=300, caplen=caplen, fdim=512, seed=31, level=None)
    a2c, rew = _modules(cfg)
    feats, caps = synthetic.make_minibatch(n, 300, caplen, 512, 31)
    steps = caplen - p0
    u = synthetic.make_uniforms(31, steps * n)
    _, exact, g_exact = _step(a2c, rew, feats, caps, p0, steps, u, chain_mode="exact", collapse_replays=False)
    assert exact.full_calls == steps and exact.chain_plan == (("exact",), ("exact",))
    eng, res, grads = _step(a2c, rew, feats, caps, p0, steps, u)
    # (a seam check may ask for a longer warm-up first, and twice that warm-up for a second call in full)
    assert res.full_calls <= 2 and res.chain_plan[0][0] == "segmented"
    fc, checked, err_v, err_r = eng.last_replay
    assert fc == res.full_calls and checked >= 128 and max(err_v, err_r) <= eng.REPLAY_TOL
    _same_step(res, grads, exact, g_exact)
    # the same with every call in full on the segmented kernels: what leaving the replays out must reproduce
    eng2, full, g_full = _step(a2c, rew, feats, caps, p0, steps, u, collapse_replays=False)
    assert full.full_calls == steps and eng2.last_replay == (steps, 0, 0.0, 0.0)
    _same_step(full, g_full, exact, g_exact)


def test_rejected_replay_check_walks_more_calls_in_full_until_it_stands():
    """A tolerance no replay can meet: every attempt that leaves replays out is rejected, more calls are walked
    in full each time, and the step that finally stands (every call in full) equals the exact kernels'."""
    cfg = dict(n=256, vocab=300, caplen=8, fdim=512, seed=33, level=None)
    a2c, rew = _modules(cfg)
    feats, caps = synthetic.make_minibatch(256, 300, 8, 512, 33)
    u = synthetic.make_uniforms(33, 7 * 256)
    _, exact, g_exact = _step(a2c, rew, feats, caps, 1, 7, u, chain_mode="exact", collapse_replays=False)

    def no_replay_passes(engine):
        engine.REPLAY_TOL = -1.0
    eng, res, grads = _step(a2c, rew, feats, caps, 1, 7, u, before=no_replay_passes)
    assert res.full_calls == 7 and eng.fallbacks >= 2 and eng._replay_need > 256
    _same_step(res, grads, exact, g_exact)


def test_long_memory_is_caught_by_the_replay_check():
    """A reward GRU whose update gates sit near 1 keeps its state for thousands of steps.  With the seam
    tolerance opened wide (so that only the replay check can object) the first attempt leaves replays out,
    the check measures that they do not return to their start state, and more calls are walked in full."""
    cfg = dict(n=256, vocab=300, caplen=8, fdim=512, seed=35, level=None)
    a2c, rew = _modules(cfg)
    with torch.no_grad():
        rew.rewrnn.gru.bias_ih_l0[512:1024] += 6.0
    feats, caps = synthetic.make_minibatch(256, 300, 8, 512, 35)
    u = synthetic.make_uniforms(35, 7 * 256)

    def seams_never_object(engine):
        engine.SEAM_TOL_FWD = float("inf")
    eng, res, _ = _step(a2c, rew, feats, caps, 1, 7, u, before=seams_never_object)
    assert eng.fallbacks >= 1 and res.full_calls > 1


def test_policy_bptt_launch_that_gave_up_is_walked_again_step_by_step(monkeypatch):
    """The policy LSTM's one-launch BPTT needs every CU; when it gives up waiting for its workgroups (another process
    holds some) it leaves its status word set and its outputs unfinished.  Simulated here: the launch runs, then its
    gate gradients are overwritten with NaN and the status word is set.  The step must notice at the backward seam
    read-back, walk the BPTT again with a launch per step, form the policy network's weight gradients again, count
    the give-up, and end on the gradients of the exact kernels."""
    cfg = dict(n=256, vocab=300, caplen=9, fdim=512, seed=37, level=None)
    a2c, rew = _modules(cfg)
    feats, caps = synthetic.make_minibatch(256, 300, 9, 512, 37)
    u = synthetic.make_uniforms(37, 8 * 256)
    _, exact, g_exact = _step(a2c, rew, feats, caps, 1, 8, u, chain_mode="exact", collapse_replays=False)
    real, calls = hip.policy_lstm_bptt_persist, []

    def gives_up(gates, c, dh_all, w_hh, dgp, dh0, work, status, **kw):
        real(gates, c, dh_all, w_hh, dgp, dh0, work, status, **kw)
        dgp.fill_(float("nan"))
        dh0.fill_(float("nan"))
        status.fill_(1)
        calls.append(1)
    monkeypatch.setattr(hip, "policy_lstm_bptt_persist", gives_up)
    eng, res, grads = _step(a2c, rew, feats, caps, 1, 8, u)
    assert len(calls) == 1, "the step took the one-launch BPTT (else this test checks nothing)"
    assert eng.launch_timeouts == 1 and res.status_host[3] == 0
    _same_step(res, grads, exact, g_exact)


def test_scales_kept_in_a_workspace_do_not_outlive_their_minibatch():
    """The two-piece products of the backward pass take some operands' power-of-two scales from the workspace (parameter
    columns, the policy LSTM's outputs), where the A2C step leaves them once per minibatch.  The teacher-forced
    pre-training step shares that workspace for the same shape: after the parameters have grown eightfold, its
    gradients must equal those of an engine that never ran an A2C step (with the old scales the cut operands would
    overflow f16)."""
    n, caplen = 256, 9
    cfg = dict(n=n, vocab=300, caplen=caplen, fdim=512, seed=39, level=None)
    a2c, rew = _modules(cfg)
    feats, caps = synthetic.make_minibatch(n, 300, caplen, 512, 39)
    u = synthetic.make_uniforms(39, (caplen - 1) * n)
    eng, _, _ = _step(a2c, rew, feats, caps, 1, caplen - 1, u)
    with torch.no_grad():
        a2c.policy_network.caption_embedding.weight.mul_(8.0)
        a2c.policy_network.lstm.weight_ih_l0.mul_(8.0)
    f, c = torch.from_numpy(feats).cuda(), torch.from_numpy(caps).cuda()
    valid = torch.full((n,), caplen - 1, dtype=torch.int32, device="cuda")
    weight = torch.full((n,), 1.0 / n, dtype=torch.float32, device="cuda")

    def ce(engine):
        loss = engine.policy_ce_step(f, c, valid, weight).clone()
        torch.cuda.synchronize()
        return loss, {k: p.grad.clone() for k, p in a2c.policy_network.named_parameters()}
    loss, grads = ce(eng)
    fresh = trainers.build_engine(a2c, rew)
    loss_f, grads_f = ce(fresh)
    assert torch.isfinite(loss).all() and torch.equal(loss, loss_f)
    for k in grads_f:
        assert torch.isfinite(grads[k]).all(), k
        scale = grads_f[k].pow(2).mean().sqrt().item() + 1e-12
        assert (grads[k] - grads_f[k]).abs().max().item() <= 1e-5 * scale, k


@pytest.mark.parametrize("variant", ["value_bptt_first", "long_replay", "four_streams"])
def test_schedule_variants_give_the_gradients_of_the_exact_kernels(variant):
    """The backward pass's other queueing orders - the value chain's BPTT ahead of the policy LSTM's (the schedule before
    the reordering), the policy network's weight gradients on the third stream (what a long replay check switches
    to) and a stream of its own for the replay check - against the exact kernels: same tokens, values, rewards,
    scalars and gradients.  Every order has to carry the same dependencies; only their timing differs."""
    cfg = dict(n=256, vocab=300, caplen=9, fdim=512, seed=41, level=None)
    a2c, rew = _modules(cfg)
    feats, caps = synthetic.make_minibatch(256, 300, 9, 512, 41)
    u = synthetic.make_uniforms(41, 8 * 256)
    _, exact, g_exact = _step(a2c, rew, feats, caps, 1, 8, u, chain_mode="exact", collapse_replays=False)
    opts, before = {}, None
    if variant == "value_bptt_first":
        opts["policy_bptt_first"] = False
    elif variant == "long_replay":
        def before(engine):
            engine.LONG_REPLAY = 1
    else:
        def before(engine):
            engine.max_streams = 4
    eng, res, grads = _step(a2c, rew, feats, caps, 1, 8, u, before=before, **opts)
    assert res.chain_plan[0][0] == "segmented" and res.full_calls < 8, "the step took the schedule under test"
    _same_step(res, grads, exact, g_exact)
    # and a second minibatch on the same engine (workspaces, events and kept scales of the first are in place)
    res2 = eng.step(torch.from_numpy(feats).cuda(), torch.from_numpy(caps[:, :1]).cuda().contiguous(), 8,
                    uniforms=torch.from_numpy(u).cuda()).check()
    torch.cuda.synchronize()
    grads2 = {k: p.grad.clone() for k, p in a2c.named_parameters()}
    _same_step(res2, grads2, exact, g_exact)
