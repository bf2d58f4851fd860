"""Pins the CPU oracle against vectors produced by running the reference itself
(oracle/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from helpers import (GOLDEN_A2C, GOLDEN_CURRICULUM, GOLDEN_LOOKAHEAD, GOLDEN_PER_SAMPLE, GOLDEN_PRETRAIN, check_summary, load_golden, load_lookahead,
                     params_for, synthetic)
from oracle import a2c_oracle as orc

# float64 oracle vs the reference's float32 arithmetic
RTOL, ATOL = 2e-4, 2e-6


@pytest.mark.parametrize("name", GOLDEN_A2C + GOLDEN_CURRICULUM)
def test_sampled_tokens_match_reference(name):
    g = load_golden(name)
    cfg = g["cfg"]
    pol, val, rew = params_for(cfg)
    out = orc.a2c_step(pol, val, rew, g["features"], g["captions"], uniforms=g["uniforms"],
                       level=cfg["level"], want_grads=False)
    p0, steps = out["p0"], out["n_steps"]
    same = out["tokens"][:, p0:] == g["tokens"][:, p0:]
    # a draw may only differ where its uniform sat within rounding of a CDF edge
    fragile = g["cdf_edge"].T < 1e-6
    assert np.all(same | fragile), "token mismatch away from a CDF edge"
    assert same.mean() > 0.99
    assert np.array_equal(out["tokens"][:, :p0], g["captions"][:, :p0])


@pytest.mark.parametrize("name", GOLDEN_A2C + GOLDEN_CURRICULUM)
def test_step_outputs_and_gradients(name):
    g = load_golden(name)
    cfg = g["cfg"]
    pol, val, rew = params_for(cfg)
    out = orc.a2c_step(pol, val, rew, g["features"], g["captions"], level=cfg["level"],
                       forced_tokens=g["tokens"])
    np.testing.assert_allclose(out["logits"], g["logits"], rtol=RTOL, atol=2e-5)
    np.testing.assert_allclose(out["log_probs"], g["log_probs"], rtol=RTOL, atol=2e-5)
    np.testing.assert_allclose(out["values"], g["values"], rtol=RTOL, atol=2e-5)
    np.testing.assert_allclose(out["rewards"], g["rewards"], rtol=RTOL, atol=2e-5)
    loss, mean_reward, mean_adv = g["scalars_step1"]
    assert abs(out["loss"] - loss) <= RTOL * abs(loss) + 2e-6
    assert abs(out["mean_reward"] - mean_reward) <= RTOL * abs(mean_reward) + 2e-6
    assert abs(out["mean_advantage"] - mean_adv) <= RTOL * abs(mean_adv) + 2e-6
    for k, grad in out["policy_grads"].items():
        check_summary(g, "grad", "policy_network." + k, grad, 1e-3, 1e-7)
    for k, grad in out["value_grads"].items():
        check_summary(g, "grad", "value_network." + k, grad, 1e-3, 1e-7)
    # one Adam step on those gradients reproduces the reference's new parameters
    params = {("policy_network." + k): v for k, v in pol.items()}
    params.update({("value_network." + k): v for k, v in val.items()})
    grads = {("policy_network." + k): v for k, v in out["policy_grads"].items()}
    grads.update({("value_network." + k): v for k, v in out["value_grads"].items()})
    new = orc.adam_update(params, grads, {})
    for k, v in new.items():
        check_summary(g, "param1", k, v, 1e-5, 1e-6)


@pytest.mark.parametrize("name", GOLDEN_A2C)
def test_greedy_decode(name):
    g = load_golden(name)
    cfg = g["cfg"]
    pol, _, _ = params_for(cfg)
    feats, caps = synthetic.make_minibatch(cfg["n"], cfg["vocab"], cfg["caplen"], cfg["fdim"], cfg["seed"])
    tokens, margins = orc.greedy_decode(pol, feats, caps)
    assert tokens.shape == (cfg["n"], 17)
    assert g["greedy_margins"].min() > 1e-5, "fixture has a near-tie; regenerate with another seed"
    assert np.array_equal(tokens, g["greedy_tokens"])
    np.testing.assert_allclose(margins, g["greedy_margins"], rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("name", GOLDEN_LOOKAHEAD)
def test_lookahead_decode(name):
    """The restatement of the beam look-ahead decoder against what the reference itself decoded: ids
    exact, scores and the value chain's final state to float32 rounding."""
    g = load_lookahead(name)
    cfg = g["cfg"]
    pol, val, _ = params_for(cfg)
    feats, caps = synthetic.make_minibatch(cfg["n"], cfg["vocab"], cfg["caplen"], cfg["fdim"], cfg["seed"])
    got = orc.lookahead_decode(pol, val, feats, caps)
    assert got["tokens"].shape == (5, cfg["n"], 17)
    assert g["logit_margin"] > 5e-6 and g["score_margin"] > 1e-4, "fixture has a near-tie; regenerate with another seed"
    assert np.array_equal(got["tokens"], g["tokens"])
    np.testing.assert_allclose(got["scores"], g["scores"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(got["state"][0], g["final_h"], rtol=1e-3, atol=2e-5)
    np.testing.assert_allclose(got["state"][1], g["final_c"], rtol=1e-3, atol=2e-5)


def test_two_step_training_matches_reference():
    """Two consecutive A2C minibatches + Adam: step-2 scalars and final parameters."""
    g = load_golden("a2c_n4_v100_l10_f512_s0")
    cfg = g["cfg"]
    pol, val, rew = params_for(cfg)
    state = {}
    out1 = orc.a2c_step(pol, val, rew, g["features"], g["captions"], forced_tokens=g["tokens"])
    params = {("policy_network." + k): v for k, v in pol.items()}
    params.update({("value_network." + k): v for k, v in val.items()})
    grads = {("policy_network." + k): v for k, v in out1["policy_grads"].items()}
    grads.update({("value_network." + k): v for k, v in out1["value_grads"].items()})
    params = orc.adam_update(params, grads, state)
    pol2 = {k: params["policy_network." + k] for k in pol}
    val2 = {k: params["value_network." + k] for k in val}
    out2 = orc.a2c_step(pol2, val2, rew, g["features_step2"], g["captions_step2"],
                        forced_tokens=g["tokens_step2"])
    loss, mean_reward, mean_adv = g["scalars_step2"]
    assert abs(out2["loss"] - loss) <= 1e-3 * abs(loss) + 1e-5
    assert abs(out2["mean_reward"] - mean_reward) <= 1e-3 * abs(mean_reward) + 1e-5
    grads = {("policy_network." + k): v for k, v in out2["policy_grads"].items()}
    grads.update({("value_network." + k): v for k, v in out2["value_grads"].items()})
    params = orc.adam_update(params, grads, state)
    for k, v in params.items():
        check_summary(g, "param2", k, v, 1e-5, 2e-6)


def test_pretraining_steps_match_reference():
    """Two minibatches of the reference's policy and value pre-training loops (length-weighted cross
    entropy; MSE against the greedy rollout's reward) against the float64 restatements + Adam(1e-3)."""
    g = load_lookahead(GOLDEN_PRETRAIN)
    cfg = g["cfg"]
    data = synthetic.make_ragged_train_data(cfg["n"], cfg["vocab"], cfg["fdim"], cfg["seed"])
    pol, val, _ = params_for(cfg)
    state = {}
    for step in range(2):   # the policy loss does not depend on the order of the rows
        loss, grads = orc.policy_ce_step(pol, data["train_features"], data["train_captions"])
        assert abs(loss - g["policy_losses"][step]) <= 1e-4 * abs(loss), (step, loss)
        pol = orc.adam_update(pol, grads, state, lr=1e-3)
    for k, v in pol.items():
        check_summary(g, "policy_after", k, v, 1e-5, 2e-6)
    state = {}
    for step in range(2):
        rows = g["value_rows"][step]
        caps = g["value_gen_caps"][step][:, :int(g["value_cols"][step])]
        loss, _, grads = orc.value_mse_step(val, data["train_features"][rows], caps, g["value_rewards"][step])
        assert abs(loss - g["value_losses"][step]) <= 2e-3 * abs(loss) + 1e-6, (step, loss)
        val = orc.adam_update(val, grads, state, lr=1e-3)
    for k, v in val.items():
        check_summary(g, "value_after", k, v, 1e-5, 2e-6)
    # reward: visual-semantic embedding loss through the batch-1 GRU chain, Adam(1e-4)
    _, _, rew = params_for(cfg)
    state = {}
    for step in range(2):
        rows = g["reward_rows"][step]
        loss, grads = orc.reward_vse_step(rew, data["train_features"][rows], data["train_captions"][rows])
        assert abs(loss - g["reward_losses"][step]) <= 1e-3 * abs(loss), (step, loss)
        rew = orc.adam_update(rew, grads, state, lr=1e-4)
    for k, v in rew.items():
        check_summary(g, "reward_after", k, v, 1e-5, 2e-6)


@pytest.mark.parametrize("name", GOLDEN_PER_SAMPLE)
def test_per_sample_chains_match_reference_at_batch_size_one(name):
    """per_sample=True on a batch == the reference's loop fed one image per minibatch: same tokens from
    the same draws, per-image values / rewards / logged scalars, and the mean of the per-image gradients."""
    g = load_lookahead(name)
    cfg = g["cfg"]
    pol, val, rew = params_for(cfg)
    steps = cfg["caplen"] - 1
    free = orc.a2c_step(pol, val, rew, g["features"], g["captions"], uniforms=g["uniforms"].reshape(-1),
                        want_grads=False, per_sample=True)
    assert np.array_equal(free["tokens"], g["tokens"])
    out = orc.a2c_step(pol, val, rew, g["features"], g["captions"], forced_tokens=g["tokens"], per_sample=True)
    np.testing.assert_allclose(out["values"], g["values"], rtol=RTOL, atol=2e-5)
    np.testing.assert_allclose(out["rewards"], g["rewards"], rtol=RTOL, atol=2e-5)
    loss, mean_reward, mean_adv = g["scalars"].mean(axis=0)   # equal caption lengths: mean of per-image means
    assert abs(out["loss"] - loss) <= RTOL * abs(loss) + 2e-6
    assert abs(out["mean_reward"] - mean_reward) <= RTOL * abs(mean_reward) + 2e-6
    assert abs(out["mean_advantage"] - mean_adv) <= RTOL * abs(mean_adv) + 2e-6
    for k, grad in out["policy_grads"].items():
        check_summary(g, "grad", "policy_network." + k, grad, 1e-3, 1e-7)
    for k, grad in out["value_grads"].items():
        check_summary(g, "grad", "value_network." + k, grad, 1e-3, 1e-7)
    # and it is a different computation from the batch-1 chain through all rows
    batch = orc.a2c_step(pol, val, rew, g["features"], g["captions"], forced_tokens=g["tokens"], want_grads=False)
    assert np.abs(batch["values"][1:] - out["values"][1:]).max() > 1e-3
    np.testing.assert_allclose(batch["values"][0, 0], out["values"][0, 0], rtol=1e-12)
    assert steps == out["n_steps"]


def test_curriculum_skips_short_captions():
    cfg = dict(n=4, vocab=100, caplen=5, fdim=512, seed=0)
    pol, val, rew = params_for(cfg)
    feats, caps = synthetic.make_minibatch(4, 100, 5, 512, 0)
    assert orc.a2c_step(pol, val, rew, feats, caps, uniforms=np.zeros(64), level=6) is None


@pytest.mark.parametrize("name", ["a2c_n4_v100_l10_f512_s0", "a2c_n4_v100_l10_f4096_s0"])
def test_cpu_port_matches_reference(name):
    """The reference-shaped float32 port that bench.py times as the CPU baseline."""
    from oracle import cpu_port
    g = load_golden(name)
    cfg = g["cfg"]
    pol, val, rew = params_for(cfg)
    nets = cpu_port.PortNets(pol, val, rew, cfg["vocab"], cfg["fdim"])
    opt = cpu_port.make_optimizer(nets)
    np.random.seed(cfg["seed"])
    out = cpu_port.a2c_minibatch(nets, opt, g["features"], g["captions"])
    same = out["tokens"] == g["tokens"]
    assert np.all(same[:, 1:] | (g["cdf_edge"].T < 1e-6))
    loss, mean_reward, mean_adv = g["scalars_step1"]
    assert abs(out["loss"] - loss) <= 1e-4 * abs(loss) + 1e-6
    assert abs(out["mean_reward"] - mean_reward) <= 1e-4 * abs(mean_reward) + 1e-6
    for k, p in nets.named_trainable():
        check_summary(g, "grad", k, p.grad.numpy(), 1e-3, 1e-7)
        check_summary(g, "param1", k, p.detach().numpy(), 1e-5, 1e-6)


def test_operand_rounding_switch_of_the_policy_rollout():
    """operand_dtype rounds both operands of the policy products and nothing else: float32 operands leave float32
    parameters' results where they were (only h is rounded: ~1e-7), bf16 operands move the log-probabilities by
    bf16's 2^-9 per factor - well under a percent, well over float32 rounding."""
    import torch
    from helpers import synthetic
    n, vocab, caplen, fdim = 4, 100, 10, 512
    P = orc.to_t(synthetic.make_params("policy", vocab, fdim, 0))
    feats, caps = synthetic.make_minibatch(n, vocab, caplen, fdim, 0)
    forced = np.random.RandomState(0).randint(4, vocab, size=(n, caplen - 1))
    run = lambda od: orc.policy_rollout(P, feats, caps[:, :1], caplen - 1, mode="forced", forced=forced,
                                        operand_dtype=od)[2].numpy()
    with torch.no_grad():
        exact, f32, bf16 = run(None), run(torch.float32), run(torch.bfloat16)
    assert np.abs(f32 - exact).max() < 2e-6
    assert 1e-5 < np.abs(bf16 - exact).max() < 5e-2
