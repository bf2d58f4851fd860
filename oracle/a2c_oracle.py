"""CPU oracle for the A2C caption-training step.  TEST INFRASTRUCTURE ONLY.

This is a from-scratch restatement (torch-CPU tensor algebra, float64 by default)
of what the reference computes on its hot path; it is the checker for the HIP
kernels and is never on the product path.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function here
against tests/golden/*.npz, which oracle/gen_golden.py produced by importing and
running the reference's own models.py / trainers.py on CPU.

What is restated (all citations are reference file:line):
  * PolicyNetwork.forward, models.py:71-84 - run incrementally: the LSTM state
    is carried from one rollout step to the next and only the newest position
    is projected to the vocabulary.  Same values and gradients as the
    reference's full-prefix recompute (models.py:286 keeps only [:, -1:, :]).
  * ValueNetworkRNN / ValueNetwork.forward, models.py:130-135, 166-180 - the
    nn.LSTM is built without batch_first and fed (N, 1, E), so the batch is the
    sequence: ONE batch-1 recurrence over rows, columns and successive calls,
    its state reset only by init_hidden() at the end of a minibatch
    (trainers.py:495-496).
  * RewardNetworkRNN / RewardNetwork.forward, models.py:223-228, 253-262 - the
    same with a GRU; GetRewards, trainers.py:108-121, cosine of the two embeds.
  * a2c_training rollout, loss and update, trainers.py:436-480, and the
    curriculum variant, trainers.py:547-594.
  * numpy.random.choice(V, p=...) (trainers.py:449): float64 cumsum of p,
    divide by the last entry, searchsorted(..., side="right") of ONE uniform.
  * torch.optim.Adam defaults (trainers.py:378): lr 1e-4, betas (0.9, 0.999),
    eps 1e-8, no weight decay, bias-corrected.
"""
import math

import numpy as np
import torch

H = 512


def to_t(params, dtype=torch.float64, requires_grad=False):
    out = {}
    for k, v in params.items():
        t = torch.tensor(np.asarray(v), dtype=dtype)
        if requires_grad:
            t.requires_grad_(True)
        out[k] = t
    return out


def caption_length(captions):
    """trainers.py:436: right-most column holding an <END> (id 2), plus one."""
    cols = np.nonzero(np.asarray(captions) == 2)[1]
    return int(cols.max()) + 1


def inverse_cdf_sample(p_row, u):
    """numpy.random.choice(len(p), p=p) given its one uniform draw u."""
    cdf = np.cumsum(np.asarray(p_row).astype(np.float64))
    cdf /= cdf[-1]
    return int(np.searchsorted(cdf, u, side="right"))


def cdf_edge_distance(p_row, u):
    """Distance from u to the nearest CDF edge: small means the draw is fragile."""
    cdf = np.cumsum(np.asarray(p_row).astype(np.float64))
    cdf /= cdf[-1]
    return float(np.min(np.abs(cdf - u)))


def _rounded(x, operand_dtype):
    """x with every element rounded to operand_dtype (round to nearest even) and widened again: the value a
    matrix unit multiplies when its operands are stored in that type and its sums are kept wide."""
    return x if operand_dtype is None else x.to(operand_dtype).to(x.dtype)


def _lstm_step(xp, h, c, whh_t, operand_dtype=None):
    g = xp + _rounded(h, operand_dtype) @ whh_t
    i, f, gg, o = g.split(g.shape[-1] // 4, dim=-1)
    c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    h = torch.sigmoid(o) * torch.tanh(c)
    return h, c


def _gru_step(xp, h, whh_t, bhh):
    hp = h @ whh_t + bhh
    xr, xz, xn = xp.split(xp.shape[-1] // 3, dim=-1)
    hr, hz, hn = hp.split(hp.shape[-1] // 3, dim=-1)
    r = torch.sigmoid(xr + hr)
    z = torch.sigmoid(xz + hz)
    n = torch.tanh(xn + r * hn)
    return (1.0 - z) * n + z * h


def policy_rollout(P, features, prefix, n_steps, uniforms=None, mode="sample", forced=None, operand_dtype=None):
    """Incremental policy rollout.

    operand_dtype (e.g. torch.bfloat16): both operands of every product of the policy network
    (models.py:75, 80, 82) are rounded to that type before they are multiplied; sums, biases, the cell
    and the softmax stay in the parameters' dtype.  With float64 parameters this is the exact result
    of reduced-precision operands with wide accumulation, the arithmetic of the opt-in bf16 mode.

    prefix: int64 [N, p0] ground-truth start columns (p0 = 1 for plain A2C).
    mode: "sample" (inverse CDF with uniforms[n_steps, N]), "greedy" (argmax on the
    logits, first index wins, trainers.py:69) or "forced" (tokens given in forced
    [N, n_steps]; used to check downstream numbers on the reference's own tokens).
    Returns tokens [N, p0+n_steps], logits [n_steps, N, V], log_probs [N, n_steps].
    """
    n, p0 = prefix.shape
    od = operand_dtype
    emb = _rounded(P["caption_embedding.weight"], od)
    wih_t = _rounded(P["lstm.weight_ih_l0"], od).t()
    whh_t = _rounded(P["lstm.weight_hh_l0"], od).t()
    wvocab_t = _rounded(P["linear2vocab.weight"], od).t()
    bias = P["lstm.bias_ih_l0"] + P["lstm.bias_hh_l0"]
    feats = torch.as_tensor(features, dtype=emb.dtype)
    h = _rounded(feats, od) @ _rounded(P["cnn2linear.weight"], od).t() + P["cnn2linear.bias"]
    c = torch.zeros_like(h)
    tokens = torch.as_tensor(np.asarray(prefix), dtype=torch.int64).clone()
    for t in range(p0 - 1):  # prime on the ground-truth prefix, no outputs needed
        h, c = _lstm_step(emb[tokens[:, t]] @ wih_t + bias, h, c, whh_t, od)
    all_logits, log_probs = [], []
    for k in range(n_steps):
        col = p0 - 1 + k
        h, c = _lstm_step(emb[tokens[:, col]] @ wih_t + bias, h, c, whh_t, od)
        logits = _rounded(h, od) @ wvocab_t + P["linear2vocab.bias"]
        all_logits.append(logits)
        if mode == "greedy":
            act = logits.detach().argmax(dim=1)
            log_probs.append(torch.zeros(n, dtype=emb.dtype))
        else:
            probs = torch.softmax(logits, dim=1)
            if mode == "sample":
                # the reference hands float32 probabilities to numpy (trainers.py:445)
                dist = probs.detach().to(torch.float32).numpy()
                act = torch.tensor([inverse_cdf_sample(dist[i], uniforms[k][i]) for i in range(n)],
                                   dtype=torch.int64)
            else:
                act = torch.as_tensor(np.asarray(forced)[:, k], dtype=torch.int64)
            log_probs.append(torch.log(probs.gather(1, act.view(-1, 1)))[:, 0])
        tokens = torch.cat((tokens, act.view(-1, 1)), dim=1)
    return tokens, torch.stack(all_logits, 0), torch.stack(log_probs, 1)


def value_chain_positions(n, p0, n_calls, extra=0):
    """Token-stream order of the batch-1 recurrence: for call s, for column t in
    [0, p0+s+extra), for row i.  Returns int arrays (call, column, row)."""
    calls, cols, rows = [], [], []
    for s in range(n_calls):
        for t in range(p0 + s + extra):
            for i in range(n):
                calls.append(s)
                cols.append(t)
                rows.append(i)
    return np.array(calls), np.array(cols), np.array(rows)


def value_forward(Vp, features, tokens, p0, n_calls, per_sample=False):
    """values [N, n_calls] of ValueNetwork called on tokens[:, :p0+s], s = 0..n_calls-1,
    with ONE hidden state carried through rows, columns and calls.  per_sample: what the same
    classes compute when called with one image at a time (batch_size 1): every row carries its own
    state through its columns and calls, from zeros."""
    n = tokens.shape[0]
    emb = Vp["valrnn.caption_embedding.weight"]
    feats = torch.as_tensor(features, dtype=emb.dtype)
    xp_tab = emb[tokens] @ Vp["valrnn.lstm.weight_ih_l0"].t() \
        + Vp["valrnn.lstm.bias_ih_l0"] + Vp["valrnn.lstm.bias_hh_l0"]  # [N, L, 4H]
    whh_t = Vp["valrnn.lstm.weight_hh_l0"].t()
    rows = n if per_sample else 1
    h = torch.zeros(rows, H, dtype=emb.dtype)
    c = torch.zeros(rows, H, dtype=emb.dtype)
    values = []
    for s in range(n_calls):
        outs = None
        for t in range(p0 + s):
            if per_sample:
                h, c = _lstm_step(xp_tab[:, t], h, c, whh_t)
                continue
            outs = []
            for i in range(n):
                h, c = _lstm_step(xp_tab[i, t][None], h, c, whh_t)
                outs.append(h[0])
        hs = h if per_sample else torch.stack(outs, 0)  # hidden outputs of the last column's N positions
        state = torch.cat((feats, hs), dim=1)
        y = state @ Vp["linear1.weight"].t() + Vp["linear1.bias"]
        values.append((y @ Vp["linear2.weight"].t() + Vp["linear2.bias"])[:, 0])
    return torch.stack(values, 1)


def reward_forward(Rp, features, tokens, p0, n_calls, per_sample=False):
    """rewards [N, n_calls]: GetRewards on tokens[:, :p0+s+1] (the newly sampled token is
    included, trainers.py:454-459), one GRU state carried through everything (per_sample: one
    state per row, as in value_forward)."""
    n = tokens.shape[0]
    emb = Rp["rewrnn.caption_embedding.weight"]
    feats = torch.as_tensor(features, dtype=emb.dtype)
    xp_tab = emb[tokens] @ Rp["rewrnn.gru.weight_ih_l0"].t() + Rp["rewrnn.gru.bias_ih_l0"]
    whh_t = Rp["rewrnn.gru.weight_hh_l0"].t()
    bhh = Rp["rewrnn.gru.bias_hh_l0"]
    ve = feats @ Rp["visual_embed.weight"].t() + Rp["visual_embed.bias"]
    ve = ve / ve.norm(dim=1, keepdim=True).clamp_min(1e-12)  # F.normalize, eps 1e-12
    h = torch.zeros(n if per_sample else 1, H, dtype=emb.dtype)
    rewards = []
    for s in range(n_calls):
        outs = None
        for t in range(p0 + s + 1):
            if per_sample:
                h = _gru_step(xp_tab[:, t], h, whh_t, bhh)
                continue
            outs = []
            for i in range(n):
                h = _gru_step(xp_tab[i, t][None], h, whh_t, bhh)
                outs.append(h[0])
        hs = h if per_sample else torch.stack(outs, 0)
        se = hs @ Rp["semantic_embed.weight"].t() + Rp["semantic_embed.bias"]
        se = se / se.norm(dim=1, keepdim=True).clamp_min(1e-12)
        rewards.append((ve * se).sum(dim=1))
    return torch.stack(rewards, 1)


def a2c_loss(values, rewards, log_probs):
    """trainers.py:471-475.  advantage = value - reward, NOT detached."""
    adv = values - rewards
    actor = (-log_probs * adv).mean()
    critic = 0.5 * adv.pow(2).mean()
    return actor + critic, actor, critic, adv


def a2c_step(policy, value, reward, features, captions, uniforms=None, level=None,
             forced_tokens=None, dtype=torch.float64, want_grads=True, per_sample=False):
    """One A2C minibatch: rollout, rewards, loss and gradients (no optimizer step).

    level=None is a2c_training (trainers.py:436-480): start from captions[:, :1],
    caplen-1 steps.  level=k is a2c_curriculum_training (trainers.py:547-594): start
    from captions[:, :caplen-k], k steps; returns None when caplen-k < 1.
    uniforms: flat array consumed in (step, row) order.  forced_tokens [N, p0+steps]
    replaces sampling (teacher forcing on the reference's own tokens).  per_sample: every row
    owns its value / reward recurrent state; the result then equals the mean over rows of the
    reference's minibatches of one image each.
    """
    captions = np.asarray(captions)
    n = captions.shape[0]
    caplen = caption_length(captions)
    if level is None:
        p0, n_steps = 1, caplen - 1
    else:
        p0, n_steps = caplen - level, level
        if p0 < 1:
            return None
    P = to_t(policy, dtype, want_grads)
    Vp = to_t(value, dtype, want_grads)
    Rp = to_t(reward, dtype, False)
    prefix = captions[:, :p0]
    if forced_tokens is not None:
        forced = np.asarray(forced_tokens)[:, p0:p0 + n_steps]
        tokens, logits, logp = policy_rollout(P, features, prefix, n_steps, mode="forced", forced=forced)
    else:
        u = np.asarray(uniforms)[:n_steps * n].reshape(n_steps, n)
        tokens, logits, logp = policy_rollout(P, features, prefix, n_steps, uniforms=u)
    values = value_forward(Vp, features, tokens, p0, n_steps, per_sample)
    with torch.no_grad():
        rewards = reward_forward(Rp, features, tokens, p0, n_steps, per_sample)
    loss, actor, critic, adv = a2c_loss(values, rewards, logp)
    out = {
        "tokens": tokens.numpy(), "logits": logits.detach().numpy(),
        "log_probs": logp.detach().numpy(), "values": values.detach().numpy(),
        "rewards": rewards.numpy(), "loss": float(loss.detach()), "actor_loss": float(actor.detach()),
        "critic_loss": float(critic.detach()), "mean_reward": float(rewards.mean()),
        "mean_advantage": float(adv.detach().mean()), "p0": p0, "n_steps": n_steps,
    }
    if want_grads:
        loss.backward()
        out["policy_grads"] = {k: (v.grad.numpy() if v.grad is not None else np.zeros(v.shape))
                               for k, v in P.items()}
        out["value_grads"] = {k: (v.grad.numpy() if v.grad is not None else np.zeros(v.shape))
                              for k, v in Vp.items()}
    return out


def greedy_decode(policy, features, captions, max_seq_len=17, dtype=torch.float64):
    """GenerateCaptionsGreedy, trainers.py:57-70: max_seq_len-1 argmax steps from
    captions[:, 0:1].  Returns tokens [N, max_seq_len] and the top-2 logit margins
    [N, max_seq_len-1] (bit-exact ids are only well-posed when the margin exceeds the
    arithmetic's rounding error)."""
    P = to_t(policy, dtype)
    with torch.no_grad():
        tokens, logits, _ = policy_rollout(P, features, np.asarray(captions)[:, :1],
                                           max_seq_len - 1, mode="greedy")
        top2 = logits.topk(2, dim=2).values
        margins = (top2[..., 0] - top2[..., 1]).t().numpy()
    return tokens.numpy(), margins


def policy_last_logits(P, features, caption):
    """PolicyNetwork.forward(features[None], caption)[:, -1, :] (models.py:71-84): the LSTM from
    h0 = cnn2linear(features), c0 = 0 over every column of `caption` [N, len], last position projected."""
    emb = P["caption_embedding.weight"]
    wih_t, whh_t = P["lstm.weight_ih_l0"].t(), P["lstm.weight_hh_l0"].t()
    bias = P["lstm.bias_ih_l0"] + P["lstm.bias_hh_l0"]
    feats = torch.as_tensor(features, dtype=emb.dtype)
    h = feats @ P["cnn2linear.weight"].t() + P["cnn2linear.bias"]
    c = torch.zeros_like(h)
    for t in range(caption.shape[1]):
        h, c = _lstm_step(emb[caption[:, t]] @ wih_t + bias, h, c, whh_t)
    return h @ P["linear2vocab.weight"].t() + P["linear2vocab.bias"]


def value_call(Vp, feats, caption, state):
    """One ValueNetwork.forward(features, caption) (models.py:166-180) continuing the batch-1 chain
    from state = (h, c); returns (values [N], new state)."""
    h, c = state
    emb = Vp["valrnn.caption_embedding.weight"]
    xp = emb[caption] @ Vp["valrnn.lstm.weight_ih_l0"].t() + Vp["valrnn.lstm.bias_ih_l0"] + Vp["valrnn.lstm.bias_hh_l0"]
    whh_t = Vp["valrnn.lstm.weight_hh_l0"].t()
    outs = None
    for t in range(caption.shape[1]):
        outs = []
        for i in range(caption.shape[0]):
            h, c = _lstm_step(xp[i, t][None], h, c, whh_t)
            outs.append(h[0])
    y = torch.cat((feats, torch.stack(outs, 0)), dim=1) @ Vp["linear1.weight"].t() + Vp["linear1.bias"]
    return (y @ Vp["linear2.weight"].t() + Vp["linear2.bias"])[:, 0], (h, c)


def lookahead_decode(policy, value, features, captions, beam=5, max_seq_len=17, dtype=torch.float64):
    """GenerateCaptionsWithActorCriticLookAhead (trainers.py:73-105) restated: whole-batch candidates,
    top-`beam` LOGITS per row as next words, score -= 0.6 * value + 0.4 * log(logit), stable sort by the
    batch-mean score, keep `beam`; the value chain's state runs on through every call in call order.
    Returns dict(tokens [beam, N, max_seq_len], scores [beam, N], state (h, c) after the last call,
    logit_margin = smallest gap between consecutive entries of any row's top beam+1 logits,
    score_margin = smallest gap between consecutive sorted candidate means among the first beam+1):
    the decoded ids are only well-posed when both margins exceed the arithmetic's rounding error."""
    P, Vp = to_t(policy, dtype), to_t(value, dtype)
    feats = torch.as_tensor(features, dtype=dtype)
    state = (torch.zeros(1, H, dtype=dtype), torch.zeros(1, H, dtype=dtype))
    cands = [(torch.as_tensor(np.asarray(captions)[:, :1], dtype=torch.int64), 0.0)]
    logit_margin, score_margin = float("inf"), float("inf")
    with torch.no_grad():
        for _ in range(max_seq_len - 1):
            nxt = []
            for cap, score in cands:
                logits = policy_last_logits(P, feats, cap)
                top_v, top_i = logits.topk(min(beam + 1, logits.shape[1]), dim=1)
                logit_margin = min(logit_margin, float((top_v[:, :-1] - top_v[:, 1:]).min()))
                for i in range(beam):
                    cap_i = torch.cat((cap, top_i[:, i:i + 1]), dim=1)
                    v, state = value_call(Vp, feats, cap_i, state)
                    nxt.append((cap_i, score - (0.6 * v + 0.4 * torch.log(top_v[:, i]))))
            means = [float(c[1].mean()) for c in nxt]
            order = sorted(range(len(nxt)), key=lambda j: means[j])
            ranked = [means[j] for j in order[:beam + 1]]
            if len(ranked) > 1:
                score_margin = min(score_margin, float(np.diff(ranked).min()))
            cands = [nxt[j] for j in order[:beam]]
    return {"tokens": np.stack([c[0].numpy() for c in cands]), "scores": np.stack([c[1].numpy() for c in cands]),
            "state": (state[0][0].numpy(), state[1][0].numpy()),
            "logit_margin": logit_margin, "score_margin": score_margin}


def policy_ce_step(policy, features, captions, dtype=torch.float64):
    """One minibatch of train_policy_network (trainers.py:228-242): teacher-forced logits of
    captions[:, :-1], loss = sum_i (caplen_i / N) * CrossEntropy(output[i][:caplen_i], captions[i, 1:][:caplen_i])
    with caplen_i = index of the first <END> (2) + 1 and the slices clipped to the L-1 positions that
    exist.  Returns (loss float, gradients dict)."""
    P = to_t(policy, dtype, requires_grad=True)
    caps = np.asarray(captions)
    n, l_cols = caps.shape
    _, _, logp = policy_rollout(P, features, caps[:, :1], l_cols - 1, mode="forced", forced=caps[:, 1:])
    loss = 0.0
    for i in range(n):
        caplen = int(np.nonzero(caps[i] == 2)[0][0]) + 1
        valid = min(caplen, l_cols - 1)
        loss = loss + (caplen / n) * (-logp[i, :valid]).mean()
    loss.backward()
    return float(loss), {k: v.grad.numpy() for k, v in P.items() if v.grad is not None}


def value_mse_step(value, features, captions, targets, dtype=torch.float64):
    """One minibatch of train_value_network after the greedy rollout and its reward
    (trainers.py:175-190): ValueNetwork on captions [N, r] from a zero chain state, nn.MSELoss against
    targets [N].  Returns (loss float, values [N], gradients dict)."""
    Vp = to_t(value, dtype, requires_grad=True)
    caps = np.asarray(captions)
    values = value_forward(Vp, features, torch.as_tensor(caps, dtype=torch.int64), caps.shape[1], 1)[:, 0]
    loss = ((values - torch.as_tensor(np.asarray(targets), dtype=dtype)) ** 2).mean()
    loss.backward()
    return float(loss), values.detach().numpy(), {k: v.grad.numpy() for k, v in Vp.items() if v.grad is not None}


def vse_loss(visuals, semantics, beta=0.2):
    """VisualSemanticEmbeddingLoss, trainers.py:21-54."""
    n = visuals.shape[0]
    margin = (beta / n) * (torch.ones(n, n, dtype=visuals.dtype) - torch.eye(n, dtype=visuals.dtype))
    vis = visuals @ semantics.t()
    vis = torch.relu(vis - torch.diag(vis).unsqueeze(1) + margin).sum() / n
    sem = semantics @ visuals.t()
    sem = torch.relu(sem - torch.diag(sem).unsqueeze(1) + margin).sum() / n
    return vis + sem


def reward_vse_step(reward, features, captions, dtype=torch.float64):
    """One minibatch of train_reward_network (trainers.py:286-300): RewardNetwork on the whole captions
    [N, L] from a zero chain state (one batch-1 GRU recurrence over columns, then rows), the VSE loss of
    the un-normalised embeddings.  Returns (loss float, gradients dict)."""
    Rp = to_t(reward, dtype, requires_grad=True)
    caps = torch.as_tensor(np.asarray(captions), dtype=torch.int64)
    n, l_cols = caps.shape
    feats = torch.as_tensor(features, dtype=dtype)
    xp = Rp["rewrnn.caption_embedding.weight"][caps] @ Rp["rewrnn.gru.weight_ih_l0"].t() + Rp["rewrnn.gru.bias_ih_l0"]
    whh_t, bhh = Rp["rewrnn.gru.weight_hh_l0"].t(), Rp["rewrnn.gru.bias_hh_l0"]
    h = torch.zeros(1, H, dtype=dtype)
    outs = None
    for t in range(l_cols):
        outs = []
        for i in range(n):
            h = _gru_step(xp[i, t][None], h, whh_t, bhh)
            outs.append(h[0])
    se = torch.stack(outs, 0) @ Rp["semantic_embed.weight"].t() + Rp["semantic_embed.bias"]
    ve = feats @ Rp["visual_embed.weight"].t() + Rp["visual_embed.bias"]
    loss = vse_loss(ve, se)
    loss.backward()
    return float(loss.detach()), {k: v.grad.numpy() for k, v in Rp.items() if v.grad is not None}


def adam_update(params, grads, state, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam (no amsgrad, no weight decay), in place on float64 copies.
    state: dict with 'step', 'm', 'v' (created on first use)."""
    if not state:
        state.update(step=0, m={k: np.zeros_like(v, dtype=np.float64) for k, v in params.items()},
                     v={k: np.zeros_like(v, dtype=np.float64) for k, v in params.items()})
    state["step"] += 1
    t = state["step"]
    bc1 = 1.0 - beta1 ** t
    bc2 = 1.0 - beta2 ** t
    out = {}
    for k, p in params.items():
        g = np.asarray(grads[k], dtype=np.float64)
        state["m"][k] = beta1 * state["m"][k] + (1 - beta1) * g
        state["v"][k] = beta2 * state["v"][k] + (1 - beta2) * g * g
        denom = np.sqrt(state["v"][k]) / math.sqrt(bc2) + eps
        out[k] = np.asarray(p, dtype=np.float64) - (lr / bc1) * state["m"][k] / denom
    return out
