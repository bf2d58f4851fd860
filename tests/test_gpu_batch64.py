"""The batch-64 configuration (2048-d features, vocabulary 10 000, T = 20): the float32 step against the
reference-shaped float32 CPU port (tokens, scalars, every value / reward / log-prob, gradient norms), and the
opt-in bf16 policy arithmetic against the float32 step on the same tokens.  Needs an MI355X: -m gpu.

Tolerances of the bf16 mode (operands of the policy network's products rounded to 8 significant bits, float32
sums; chains, heads and reductions unchanged): log-probs rtol 2e-2 + atol 2e-2, policy gradients 5e-2 of the
tensor's L2 norm; values and rewards are bit-equal to the float32 step (they never touch the policy network)."""
import importlib

import numpy as np
import pytest
import torch

from helpers import params_for, synthetic
from oracle import a2c_oracle as orc
from oracle import cpu_port

pytestmark = pytest.mark.gpu

hip = importlib.import_module("image-captioning-through-rl_amd.hipabi")
models = importlib.import_module("image-captioning-through-rl_amd.models")
trainers = importlib.import_module("image-captioning-through-rl_amd.trainers")

CFG = dict(n=64, vocab=10000, caplen=20, fdim=2048, seed=2, level=None)
STEPS = CFG["caplen"] - 1


def _modules():
    w2i = synthetic.make_word_to_idx(CFG["vocab"])
    nets = (models.PolicyNetwork(w2i, input_dim=CFG["fdim"]), models.ValueNetwork(w2i, feature_dim=CFG["fdim"]),
            models.RewardNetwork(w2i, feature_dim=CFG["fdim"]))
    for net, params in zip(nets, params_for(CFG)):
        net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
        net.cuda()
    nets[2].requires_grad_(False)
    return models.AdvantageActorCriticNetwork(nets[1], nets[0]).cuda(), nets[2]


@pytest.fixture(scope="module")
def batch64():
    feats, caps = synthetic.make_minibatch(CFG["n"], CFG["vocab"], CFG["caplen"], CFG["fdim"], CFG["seed"])
    u = synthetic.make_uniforms(CFG["seed"], STEPS * CFG["n"])
    a2c, rew = _modules()
    engine = trainers.build_engine(a2c, rew)
    res = engine.step(torch.from_numpy(feats).cuda(), torch.from_numpy(caps[:, :1]).cuda(), STEPS,
                      uniforms=torch.from_numpy(u).cuda()).check()
    torch.cuda.synchronize()
    return dict(feats=feats, caps=caps, u=u, a2c=a2c, rew=rew, res=res, engine=engine,
                grads={k: p.grad.clone() for k, p in a2c.named_parameters()})


def test_float32_step_at_batch_64_against_the_cpu_port(batch64):
    b = batch64
    pol, val, rew = params_for(CFG)
    nets = cpu_port.PortNets(pol, val, rew, CFG["vocab"], CFG["fdim"])
    out = cpu_port.a2c_minibatch(nets, cpu_port.make_optimizer(nets), b["feats"], b["caps"], uniforms=b["u"])
    res = b["res"]
    same = out["tokens"] == res.tokens.cpu().numpy()
    assert same.mean() > 0.998, "sampled tokens differ from the CPU port's"
    rows = same.all(axis=1)
    np.testing.assert_allclose(res.values.cpu().numpy()[rows], out["values"][rows], rtol=5e-4, atol=5e-5)
    np.testing.assert_allclose(res.rewards.cpu().numpy()[rows], out["rewards"][rows], rtol=5e-4, atol=5e-5)
    np.testing.assert_allclose(res.log_probs.cpu().numpy()[rows], out["log_probs"][rows], rtol=5e-4, atol=5e-5)
    if same.all():
        sc = res.scalars.cpu().numpy()
        for got, want in ((sc[0], out["loss"]), (sc[3], out["mean_reward"]), (sc[4], out["mean_advantage"])):
            assert abs(got - want) <= 5e-4 * abs(want) + 1e-5, (got, want)
        for k, g in b["grads"].items():
            want = out["grad_norms"][k]
            assert abs(float(g.norm()) - want) <= 2e-3 * want + 1e-9, (k, float(g.norm()), want)


def test_bf16_policy_arithmetic_against_the_float32_step_on_the_same_tokens(batch64):
    b = batch64
    f32 = b["res"]
    engine = trainers.build_engine(b["a2c"], b["rew"], policy_dtype="bf16")
    res = engine.step(torch.from_numpy(b["feats"]).cuda(), torch.from_numpy(b["caps"][:, :1]).cuda(), STEPS,
                      mode=hip.FORCED, forced_tokens=f32.tokens).check()
    torch.cuda.synchronize()
    assert torch.equal(res.tokens, f32.tokens)
    assert torch.equal(res.values, f32.values) and torch.equal(res.rewards, f32.rewards)   # chains and heads: float32
    got, want = res.log_probs.cpu().numpy(), f32.log_probs.cpu().numpy()
    assert not np.array_equal(got, want)                  # the policy products really ran in bf16
    np.testing.assert_allclose(got, want, rtol=2e-2, atol=2e-2)
    assert abs(float(res.scalars[0]) - float(f32.scalars[0])) <= 2e-2 * abs(float(f32.scalars[0])) + 1e-3
    for k, p in b["a2c"].named_parameters():
        ref = b["grads"][k]
        tol = 5e-2 if k.startswith("policy_network.") else 2e-2
        assert float((p.grad - ref).norm()) <= tol * float(ref.norm()) + 1e-9, (k, float((p.grad - ref).norm()), float(ref.norm()))
    # a sampled rollout in bf16 draws valid tokens and finite log-probs
    free = engine.step(torch.from_numpy(b["feats"]).cuda(), torch.from_numpy(b["caps"][:, :1]).cuda(), STEPS,
                       uniforms=torch.from_numpy(b["u"]).cuda()).check()
    toks = free.tokens.cpu().numpy()
    assert toks.min() >= 0 and toks.max() < CFG["vocab"] and np.isfinite(free.log_probs.cpu().numpy()).all()
    assert (toks == f32.tokens.cpu().numpy()).mean() > 0.9    # most draws are far from a CDF edge


def test_bf16_policy_arithmetic_against_the_operand_rounding_oracle(batch64):
    """The bf16 step's log-probabilities against the float64 oracle with both operands of every policy product
    (models.py:75-82) rounded to bf16 and wide sums: the arithmetic the mode claims, not merely "near float32".
    Teacher-forced on the float32 step's tokens.  Tolerance: rtol 1e-3 of the log-probability is the stated bar;
    what is measured is far inside it (float32 against float64 sums, and a rare element of h that the two round
    to different bf16 neighbours), and several times closer than the unrounded oracle is."""
    b = batch64
    tokens = b["res"].tokens
    engine = trainers.build_engine(b["a2c"], b["rew"], policy_dtype="bf16")
    res = engine.step(torch.from_numpy(b["feats"]).cuda(), torch.from_numpy(b["caps"][:, :1]).cuda(), STEPS,
                      mode=hip.FORCED, forced_tokens=tokens, backward=False).check()
    got = res.log_probs.cpu().numpy().astype(np.float64)
    P = orc.to_t(params_for(CFG)[0])
    forced = tokens.cpu().numpy()[:, 1:]
    with torch.no_grad():
        _, _, want = orc.policy_rollout(P, b["feats"], b["caps"][:, :1], STEPS, mode="forced", forced=forced,
                                        operand_dtype=torch.bfloat16)
        _, _, exact = orc.policy_rollout(P, b["feats"], b["caps"][:, :1], STEPS, mode="forced", forced=forced)
    want, exact = want.numpy(), exact.numpy()
    err_model, err_exact = np.abs(got - want).max(), np.abs(got - exact).max()
    print("bf16 step log-probs: max |diff| %.3g against the bf16-rounding oracle, %.3g against the unrounded one"
          % (err_model, err_exact))
    np.testing.assert_allclose(got, want, rtol=1e-3, atol=0)
    assert err_model <= 2e-4, "further from the operand-rounding model than float32 sums explain"
    assert err_model < 0.25 * err_exact, "no closer to the bf16 model than to exact arithmetic"


def test_greedy_ids_at_vocabulary_10k_against_the_float64_oracle(batch64):
    """GenerateCaptionsGreedy (trainers.py:57-70) at the batch-64 configuration's size: ids equal to the float64
    oracle's wherever the oracle's top-2 logit margin exceeds 1e-5 (below it the arg-max is not well posed)."""
    b = batch64
    want, margins = orc.greedy_decode(params_for(CFG)[0], b["feats"], b["caps"])
    got = trainers.GenerateCaptionsGreedy(b["feats"], b["caps"], b["a2c"].policy_network).cpu().numpy()
    assert got.dtype == np.int64 and got.shape == want.shape == (CFG["n"], 17)
    fragile = margins < 1e-5
    first = np.where(fragile.any(axis=1), fragile.argmax(axis=1), 16)   # first step whose arg-max is not well posed
    print("rows with a top-2 margin under 1e-5: %d of %d (smallest margin %.3g)" % (fragile.any(axis=1).sum(), CFG["n"], margins.min()))
    for i in range(CFG["n"]):
        assert np.array_equal(got[i, :first[i] + 1], want[i, :first[i] + 1]), i
    assert fragile.any(axis=1).mean() <= 0.05
