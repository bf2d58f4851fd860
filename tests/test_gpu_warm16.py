"""The first steps of the chains' warm-ups at 16-bit split precision (icrl_chain_pair_fwd_seg_warm16 forward,
icrl_lstm_chain_bwd_seg_warm16 backward): a warm-up's results are discarded, so all that may change is the state
it leaves, and that is measured by the same seam check.  Needs an MI355X: -m gpu."""
import importlib

import numpy as np
import pytest
import torch

from helpers import params_for, synthetic

pytestmark = pytest.mark.gpu

hip = importlib.import_module("image-captioning-through-rl_amd.hipabi")
models = importlib.import_module("image-captioning-through-rl_amd.models")
trainers = importlib.import_module("image-captioning-through-rl_amd.trainers")
H = 512


def _pair(n, p0, s, k, fc, w_v, w_r, cheap, seed=0, scale=1.0, phase_split=False):
    tv, tr = hip.chain_positions(n, p0, s, fc), hip.chain_positions(n, p0 + 1, s, fc)
    g = torch.Generator(device="cpu").manual_seed(seed)
    u = lambda *shape, sc=1.0: ((torch.rand(*shape, generator=g) * 2 - 1) * sc).cuda()
    w_hv, w_hr, b_r = u(4 * H, H, sc=scale * H ** -0.5), u(3 * H, H, sc=scale * H ** -0.5), u(3 * H, sc=H ** -0.5)
    xp_v, xp_r = u((p0 + s - 1) * n, 4 * H), u((p0 + s) * n, 3 * H)
    hv, hr = torch.zeros(s, n, H, device="cuda"), torch.zeros(s, n, H, device="cuda")
    saved, saved_r = torch.zeros(tv, 6, H, device="cuda"), torch.zeros(tr, 5, H, device="cuda")
    work_v = torch.zeros(hip.chain_seg_workspace_floats(k), device="cuda")
    work_r = torch.zeros_like(work_v)
    seam = torch.zeros(4, device="cuda")
    args = (xp_v, w_hv, xp_r, w_hr, b_r, n, p0, p0 + 1, s, k, w_v, hv, saved, hr, work_v, work_r, seam[0:1], seam[2:3])
    opts = dict(warmup_r=w_r, full_calls=fc, saved_r=saved_r, cheap=cheap)
    if phase_split:
        hip.chain_pair_fwd_seg(*args, phase=1, **opts)
        hip.chain_pair_fwd_seg(*args, phase=2, **opts)
    else:
        hip.chain_pair_fwd_seg(*args, **opts)
    torch.cuda.synchronize()
    return hv, hr, saved, saved_r, seam.tolist()


@pytest.mark.parametrize("case", [
    dict(n=512, p0=1, s=19, k=256, fc=1, w_v=96, w_r=96, cheap=(80, 80)),      # the A2C minibatch, replays left out
    dict(n=512, p0=1, s=19, k=256, fc=1, w_v=256, w_r=80, cheap=(224, 64)),    # a longer value warm-up
    dict(n=64, p0=2, s=10, k=64, fc=0, w_v=128, w_r=96, cheap=(100, 0)),       # every call in full, two row groups, value chain only
    dict(n=40, p0=1, s=12, k=96, fc=2, w_v=64, w_r=64, cheap=(1, 63)),         # the shortest 16-bit and float32 parts
])
def test_warm16_leaves_the_state_the_float32_warm_up_leaves(case):
    if not (hip.chain_pair_supported(case["k"]) and hip.chain_warm16_supported(case["k"])):
        pytest.skip("the persistent launches do not fit this device")
    c = dict(case)
    cheap = c.pop("cheap")
    want = _pair(cheap=(0, 0), **c)
    for split in (False, True):
        got = _pair(cheap=cheap, phase_split=split, **c)
        assert max(got[4]) <= 2e-6, got[4]               # seam errors at the rounding floor, as without it
        for a, b in zip(got[:4], want[:4]):
            assert float((a - b).abs().max()) <= 2e-6


def test_warm16_is_measured_by_the_seam_check():
    """A recurrence with a long memory and a warm-up too short for it: the seam error says so with and without
    the 16-bit part, i.e. the check still measures what it measured."""
    if not hip.chain_warm16_supported(128):
        pytest.skip("the persistent launches do not fit this device")
    kw = dict(n=256, p0=1, s=19, k=128, fc=1, w_v=48, w_r=48, scale=3.0)
    plain = _pair(cheap=(0, 0), **kw)
    mixed = _pair(cheap=(32, 32), **kw)
    assert plain[4][2] > 1e-5 or plain[4][0] > 1e-5      # too short: rejected ...
    assert mixed[4][2] > 1e-5 or mixed[4][0] > 1e-5      # ... either way
    with pytest.raises(RuntimeError, match="shorter than the warm-up"):
        _pair(cheap=(48, 0), **kw)


def _bwd(n, p0, s, k, fc, w_f, w_b, cheap, seed=0, scale=1.0):
    """Value chain forward (float32, to have real activations), then its BPTT with `cheap` 16-bit warm-up steps."""
    tv = hip.chain_positions(n, p0, s, fc)
    g = torch.Generator(device="cpu").manual_seed(seed)
    u = lambda *shape, sc=1.0: ((torch.rand(*shape, generator=g) * 2 - 1) * sc).cuda()
    w_hv = u(4 * H, H, sc=scale * H ** -0.5)
    xp_v = u((p0 + s - 1) * n, 4 * H)
    d_emit = u(s, n, H, sc=1e-3)
    hv, saved = torch.zeros(s, n, H, device="cuda"), torch.zeros(tv, 6, H, device="cuda")
    work = torch.zeros(hip.chain_seg_workspace_floats(k), device="cuda")
    seam = torch.zeros(4, device="cuda")
    hip.lstm_chain_fwd_seg(xp_v, w_hv, n, p0, s, k, w_f, hv, saved, work, seam[0:2], persistent=True, full_calls=fc)
    dgv = torch.zeros(tv, 4 * H, device="cuda")
    hip.lstm_chain_bwd_seg(w_hv, saved, d_emit, n, p0, s, k, w_b, dgv, work, seam[2:4], persistent=True, full_calls=fc,
                           cheap=cheap)
    torch.cuda.synchronize()
    return dgv, seam.tolist()


@pytest.mark.parametrize("case", [
    dict(n=512, p0=1, s=19, k=256, fc=1, w_f=96, w_b=96, cheap=80),     # the A2C minibatch, replays left out
    dict(n=512, p0=1, s=19, k=256, fc=1, w_f=96, w_b=192, cheap=192),   # the whole warm-up at 16 bits
    dict(n=64, p0=2, s=10, k=64, fc=0, w_f=128, w_b=128, cheap=100),    # every call in full, two row groups
    dict(n=40, p0=1, s=12, k=96, fc=2, w_f=64, w_b=64, cheap=1),        # the shortest 16-bit part
    dict(n=40, p0=1, s=12, k=96, fc=2, w_f=64, w_b=64, cheap=2),
])
def test_backward_warm16_leaves_the_state_the_float32_warm_up_leaves(case):
    if not (hip.chain_pair_supported(case["k"]) and hip.chain_bwd_warm16_supported(case["k"])):
        pytest.skip("the persistent launches do not fit this device")
    c = dict(case)
    cheap = c.pop("cheap")
    want, seam_w = _bwd(cheap=0, **c)
    got, seam_g = _bwd(cheap=cheap, **c)
    assert seam_w[2] <= 1e-4 * seam_w[3] and seam_g[2] <= 1e-4 * seam_g[3], (seam_w, seam_g)
    # the segments' own positions are walked in float32 from a (dh, dc) that differs by what the seam allows
    assert float((got - want).abs().max()) <= 1e-4 * float(want.abs().max())
    assert float((got - want).norm()) <= 2e-5 * float(want.norm())


def test_backward_warm16_is_measured_by_the_seam_check_and_needs_whole_row_groups():
    if not hip.chain_bwd_warm16_supported(128):
        pytest.skip("the persistent launches do not fit this device")
    kw = dict(n=256, p0=1, s=19, k=128, fc=1, w_f=512, w_b=16, scale=3.0)   # a long memory, a short backward warm-up
    _, plain = _bwd(cheap=0, **kw)
    _, mixed = _bwd(cheap=12, **kw)
    assert plain[2] > 1e-4 * plain[3] and mixed[2] > 1e-4 * mixed[3]      # too short: rejected either way
    assert not hip.chain_bwd_warm16_supported(80)
    with pytest.raises(RuntimeError, match="multiple of 32"):
        _bwd(n=40, p0=1, s=12, k=80, fc=2, w_f=64, w_b=64, cheap=8)
    with pytest.raises(RuntimeError, match="longer than the warm-up"):
        _bwd(n=40, p0=1, s=12, k=96, fc=2, w_f=64, w_b=64, cheap=65)


def test_engine_uses_warm16_from_128_warm_up_steps_and_gets_the_same_minibatch():
    cfg = dict(n=512, vocab=1000, caplen=20, fdim=512, seed=3, level=None)
    w2i = synthetic.make_word_to_idx(cfg["vocab"])
    nets = (models.PolicyNetwork(w2i, input_dim=cfg["fdim"]), models.ValueNetwork(w2i, feature_dim=cfg["fdim"]),
            models.RewardNetwork(w2i, feature_dim=cfg["fdim"]))
    for net, params in zip(nets, params_for(cfg)):
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
        net.cuda()
    nets[2].requires_grad_(False)
    a2c = models.AdvantageActorCriticNetwork(nets[1], nets[0]).cuda()
    feats, caps = synthetic.make_minibatch(cfg["n"], cfg["vocab"], cfg["caplen"], cfg["fdim"], cfg["seed"])
    u = torch.from_numpy(synthetic.make_uniforms(cfg["seed"], 19 * cfg["n"])).cuda()
    args = (torch.from_numpy(feats).cuda(), torch.from_numpy(caps[:, :1]).cuda(), 19)
    out = {}
    for warm16 in (True, False, "backward too"):
        # the 16-bit warm-up part belongs to the float32 launch (the default f16 launch walks every step alike)
        engine = trainers.build_engine(a2c, nets[2], first_warmup=128, warm16=bool(warm16), chain_arith="f32",
                                       warm16_backward=warm16 == "backward too")
        res = engine.step(*args, uniforms=u).check()
        torch.cuda.synchronize()
        assert res.chain_plan[0][:1] == ("segmented",) and res.chain_plan[0][2] >= 128
        assert (engine.last_cheap[0] > 0) == bool(warm16) and engine.fallbacks == 0
        assert (engine.last_cheap_bwd > 0) == (warm16 == "backward too")   # the backward launch only when asked for
        out[warm16] = (res.tokens.clone(), res.values.clone(), res.rewards.clone(),
                       {k: p.grad.clone() for k, p in a2c.named_parameters()})
    # and the default engine (both forward chains on the f16 matrix instruction, every step)
    engine = trainers.build_engine(a2c, nets[2], first_warmup=128)
    res = engine.step(*args, uniforms=u).check()
    torch.cuda.synchronize()
    assert engine.chain_arith == "f16x2" and engine.last_cheap == (0, 0) and engine.fallbacks == 0
    out["f16x2"] = (res.tokens.clone(), res.values.clone(), res.rewards.clone(),
                    {k: p.grad.clone() for k, p in a2c.named_parameters()})
    for variant in (True, "backward too", "f16x2"):
        assert torch.equal(out[variant][0], out[False][0])
        for a, b in zip(out[variant][1:3], out[False][1:3]):
            assert float((a - b).abs().max()) <= 1e-5
        for k, g in out[False][3].items():
            assert float((out[variant][3][k] - g).norm()) <= 1e-4 * float(g.norm()) + 1e-9, k
