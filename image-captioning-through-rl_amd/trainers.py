"""The reference's A2C entry points (trainers.py) on the HIP path.

``GenerateCaptionsGreedy``, ``GetRewards``, ``a2c_training``, ``a2c_curriculum_training``,
``train_a2c_network``, the three supervised pre-training loops (``train_policy_network``,
``train_value_network``, ``train_reward_network``), the beam look-ahead decoder
``GenerateCaptionsWithActorCriticLookAhead`` and the evaluation loop ``test_a2c_network``: same names,
argument order and meaning, same logged scalar tags, same NumPy / torch RNG consumption (so the same
seeds give the same minibatch order and the same sampled tokens as the reference).

With more than one process (torch.distributed initialised, one rank per GPU) every loop here is
data-parallel: each rank takes its own rows of every global minibatch (utilities.sharded_minibatches),
draws its sampling uniforms from its own NumPy stream, starts from rank 0's parameters, averages the
flat gradient buffer before every optimizer step, takes the "skip this minibatch" decision of the
curriculum loop together with the other ranks, and only rank 0 writes checkpoints, result files and
logged scalars.
"""
import math
import os

import numpy as np
import torch

from . import dist as ddp
from . import hipabi as hip
from .engine import A2CEngine
from .models import (MAX_SEQ_LEN, AdvantageActorCriticNetwork, PolicyNetwork, RewardNetwork, ValueNetwork,
                     device)
from .optim import FusedAdam, flatten_module
from .utilities import (decode_captions, get_coco_batch, get_coco_minibatches, global_minibatch_number, minibatches,
                        print_green, save_a2c_model)

H = 512


class ScalarLog:
    """Stand-in with SummaryWriter's add_scalar when tensorboard is not installed;
    keeps (tag, value, step) triples in ``records``."""

    def __init__(self, log_dir=None):
        self.log_dir = log_dir
        self.records = []

    def add_scalar(self, tag, value, step):
        self.records.append((tag, float(value), int(step)))


class _NoLog:
    def add_scalar(self, tag, value, step):
        pass


def make_writer(plot_dir):
    if not ddp.is_main():   # one rank logs
        return _NoLog()
    if hasattr(plot_dir, "add_scalar"):  # a writer object may be passed instead of a directory
        return plot_dir
    try:
        from torch.utils.tensorboard import SummaryWriter
        return SummaryWriter(log_dir=os.path.join(plot_dir, 'runs'))
    except Exception:
        return ScalarLog(os.path.join(str(plot_dir), 'runs'))


def GenerateCaptionsGreedy(features, captions, policy_network):
    """MAX_SEQ_LEN-1 argmax steps from captions[:, 0:1] -> [N, 17] int64
    (trainers.py:57-70), with the LSTM state carried instead of recomputing the prefix."""
    w = policy_network.cnn2linear.weight
    dev = w.device
    feats = torch.as_tensor(features, device=dev).float().contiguous()
    caps = torch.as_tensor(np.asarray(captions)[:, 0:1] if not torch.is_tensor(captions) else captions[:, 0:1],
                           device=dev).long()
    n = feats.shape[0]
    vocab = policy_network.linear2vocab.weight.shape[0]
    tok_t = torch.zeros(MAX_SEQ_LEN, n, dtype=torch.int64, device=dev)
    tok_t[0] = caps[:, 0]
    h = torch.empty(n, H, device=dev)
    c_prev, c_next = torch.zeros(n, H, device=dev), torch.empty(n, H, device=dev)
    gates, logits = torch.empty(n, 4 * H, device=dev), torch.empty(n, vocab, device=dev)
    lstm = policy_network.lstm
    with torch.no_grad():
        hip.gemm(feats, w, h, bias=policy_network.cnn2linear.bias)
        for t in range(MAX_SEQ_LEN - 1):
            hip.gemm(policy_network.caption_embedding.weight, lstm.weight_ih_l0, gates, a_rows=tok_t[t],
                     bias=lstm.bias_ih_l0, bias2=lstm.bias_hh_l0)
            hip.gemm(h, lstm.weight_hh_l0, gates, accumulate=True)
            hip.lstm_cell_fwd(gates, c_prev, c_next, h)
            c_prev, c_next = c_next, c_prev
            hip.gemm(h, policy_network.linear2vocab.weight, logits, bias=policy_network.linear2vocab.bias)
            hip.sample_tokens(logits, hip.GREEDY, tok_t[t + 1], 1)
    return tok_t.t().contiguous()


@torch.no_grad()
def GenerateCaptionsWithActorCriticLookAhead(features, captions, policy_network, value_network, beamSize=5,
                                             most_likely=False):
    """Beam look-ahead decoding (trainers.py:73-105): every step each kept candidate is extended by the
    beamSize highest-scoring next words of the policy network; an extension is scored
    0.6 * value(features, caption) + 0.4 * log(its logit) per row, scores accumulate with a minus sign
    and the beamSize candidates with the lowest batch-mean score survive.  Candidates are whole-batch
    captions [N, len].  As in the reference the "probabilities" handed to the logarithm are the raw
    logits (a negative one gives NaN), the value network's recurrent state is never reset inside this
    function (the order of its calls matters) and the ranking is Python's stable sort of the means.
    Returns the list of (captions [N,17] int64, score [N,1]) pairs, or the first caption matrix."""
    dev = policy_network.cnn2linear.weight.device
    feats = torch.as_tensor(features, device=dev).float().contiguous()
    caps0 = captions[:, 0:1] if torch.is_tensor(captions) else np.asarray(captions)[:, 0:1]
    gen_caps = torch.as_tensor(caps0, device=dev).long()
    candidates = [(gen_caps, 0)]
    for _ in range(MAX_SEQ_LEN - 1):
        # the policy network is stateless, so all extensions of a step are known before the value network
        # sees the first of them; it then scores them in the reference's order (candidate by candidate,
        # word by word, its state running through), as one chain
        extensions = []
        # ... and row-wise, so the kept candidates go through it stacked along the batch dimension
        n_rows, k_c = gen_caps.shape[0], len(candidates)
        stacked = policy_network(feats.repeat(k_c, 1).unsqueeze(0), torch.cat([c for c, _ in candidates], dim=0),
                                 last_only=True)[:, -1, :]
        for j, (cand, cand_score) in enumerate(candidates):
            logits = stacked[j * n_rows:(j + 1) * n_rows].contiguous()
            top_vals, top_words = hip.topk_rows(logits, beamSize)
            for i in range(beamSize):
                extensions.append((torch.cat((cand, top_words[:, i:i + 1]), dim=1), cand_score, top_vals[:, i:i + 1]))
        values = value_network.forward_many(feats, [e[0] for e in extensions])
        # the score update of all extensions at once (the same elementwise expressions, so the same bits)
        prev = torch.stack([s if torch.is_tensor(s) else torch.full_like(values[0], float(s)) for _, s, _ in extensions])
        score_delta = 0.6 * torch.stack(values) + 0.4 * torch.log(torch.stack([t for _, _, t in extensions]))
        scores = prev - score_delta
        next_candidates = [(ext[0], scores[j]) for j, ext in enumerate(extensions)]
        means = torch.stack([c[1].mean() for c in next_candidates]).tolist()   # one read-back per step
        order = sorted(range(len(next_candidates)), key=lambda j: means[j])
        candidates = [next_candidates[j] for j in order[:beamSize]]
    if most_likely:
        return candidates[0][0]
    return candidates


def test_a2c_network(a2c_network, test_data, image_caption_data, data_size, validation_batch_size=128):
    """Decode a random validation sample with the look-ahead beam and append real captions, generated
    captions and image urls to the three result files (trainers.py:619-665; the slices of
    validation_batch_size - 1 rows and the missing newline between batches are the reference's)."""
    a2c_network.train(False)
    real_f = open(image_caption_data["real_captions_path"], "a")
    gen_f = open(image_caption_data["generated_captions_path"], "a")
    url_f = open(image_caption_data["image_urls_path"], "a")
    try:
        captions_all, features_all, urls_all = get_coco_batch(test_data, batch_size=data_size, split='val')
        for i in range(0, len(captions_all), validation_batch_size):
            features_real = features_all[i:i + validation_batch_size - 1]
            captions_real = captions_all[i:i + validation_batch_size - 1]
            urls = urls_all[i:i + validation_batch_size - 1]
            gen_cap = GenerateCaptionsWithActorCriticLookAhead(features_real, captions_real,
                                                               a2c_network.policy_network,
                                                               a2c_network.value_network, most_likely=True)
            real_f.write("\n".join(decode_captions(captions_real, idx_to_word=test_data["idx_to_word"])))
            gen_f.write("\n".join(decode_captions(gen_cap, idx_to_word=test_data["idx_to_word"])))
            url_f.write("\n".join(urls))
            for f in (real_f, gen_f, url_f):
                f.flush()
            a2c_network.value_network.valrnn.init_hidden()
    finally:
        for f in (real_f, gen_f, url_f):
            f.close()


test_a2c_network.__test__ = False  # a reference entry point, not a pytest case


def GetRewards(features, captions, reward_network):
    """Cosine similarity of the reward network's two embeddings -> [N,1]
    (trainers.py:108-121)."""
    vis, sem = reward_network(features, captions)
    n = vis.shape[0]
    rewards = torch.empty(n, device=vis.device)
    hip.cosine_reward(vis, sem, n, n, rewards)
    return rewards.unsqueeze(1)


def _params_of(module):
    return {k: p.data for k, p in module.named_parameters()}


def _grads_of(module):
    grads = {}
    for k, p in module.named_parameters():
        if not p.requires_grad:
            continue
        if p.grad is None:
            p.grad = torch.zeros_like(p.data)
        grads[k] = p.grad
    return grads


def build_engine(a2c_network, reward_network, **engine_options):
    """Bind an A2CEngine to the live parameter / gradient storage of the modules.  Modules built with
    per_sample=True get the engine's per-image recurrent state too, unless chain_semantics is given."""
    if getattr(a2c_network.value_network.valrnn, "per_sample", False):
        engine_options.setdefault("chain_semantics", "per_sample")
    # a reward network without trainable parameters (trainers.py:372-373 freezes it) is not written to between
    # minibatches; whoever loads new weights into it afterwards calls engine.parameters_changed()
    engine_options.setdefault("frozen_reward", not any(p.requires_grad for p in reward_network.parameters()))
    return A2CEngine(_params_of(a2c_network.policy_network), _params_of(a2c_network.value_network),
                     _params_of(reward_network), _grads_of(a2c_network.policy_network),
                     _grads_of(a2c_network.value_network), **engine_options)


def caption_length(captions):
    """Right-most column holding an <END> (id 2), plus one (trainers.py:436)."""
    return int(np.nonzero(np.asarray(captions) == 2)[1].max()) + 1


def _a2c_minibatch(engine, optimizer, captions, features, p0, n_steps):
    """Rollout + loss + gradients + update for one minibatch; returns the step result."""
    dev = engine.device
    caps_np = np.asarray(captions)
    n = caps_np.shape[0]
    feats = torch.as_tensor(features).float().to(dev).contiguous()
    prefix = torch.as_tensor(caps_np[:, :p0]).long().to(dev)
    # one double per (step, row), drawn from NumPy's global stream exactly as the reference's
    # N calls of np.random.choice per step do (trainers.py:447-449)
    uniforms = torch.from_numpy(np.random.random_sample(n_steps * n)).to(dev)
    flat = getattr(optimizer, "fp", None)
    averager = None
    if flat is not None and ddp.world_size() > 1:
        # one all-reduce of the flat buffer after the step - or, with the value chain's BPTT queued in front of the
        # policy LSTM's (policy_bptt_first=False), where the policy network's gradients are complete well before the
        # value network's, their all-reduce started as soon as the engine says so, beside the rest of the backward
        # pass.  (The switch is a constructor argument, the same on every rank: every rank issues the same collectives.)
        averager = getattr(engine, "_averager", None)
        if averager is None or averager.flat_grad is not flat.flat_grad:
            averager = engine._averager = ddp.GradientAverager(flat.flat_grad, flat.span("policy_network."),
                                                               stream=engine.side_stream())
        engine.on_policy_grads = None if engine.policy_bptt_first else averager.start_early
    res = engine.step(feats, prefix, n_steps, uniforms=uniforms)
    if averager is not None:
        averager.finish()
    optimizer.step()
    return res


def a2c_training(train_data, a2c_network, reward_network, optimizer, plot_dir, save_paths, batch_size, epochs,
                 chain_semantics="batch", policy_dtype="f32"):
    """Advantage-weighted policy-gradient training (trainers.py:402-500).  chain_semantics="per_sample"
    gives every image its own recurrent state in the value / reward networks (what the reference
    computes at batch_size 1) instead of one recurrence through all rows of the minibatch.
    policy_dtype="bf16": the policy network's products in bf16 with float32 sums (an opt-in; the default is
    exact float32)."""
    writer = make_writer(plot_dir)
    print_green('[Training] Training Advantage Actor-Critic Network')
    engine = build_engine(a2c_network, reward_network, chain_semantics=chain_semantics, policy_dtype=policy_dtype)
    ddp.decorrelate_numpy_stream()
    best_loss = float('inf')
    for epoch in range(epochs):
        for minibatch_id, (captions, features, _) in enumerate(
                minibatches(train_data, batch_size=batch_size, split='train')):
            caplen = caption_length(captions)
            res = _a2c_minibatch(engine, optimizer, captions, features, 1, caplen - 1).check()
            loss, _, _, mean_reward, mean_adv = res.scalars[:5].tolist()
            best_loss = min(best_loss, loss)
            step = global_minibatch_number(epoch, minibatch_id, batch_size)
            writer.add_scalar('A2C Network-episodic-loss', loss, step)
            writer.add_scalar('A2C Network-episodic-mean-rewards', mean_reward, step)
            writer.add_scalar('A2C Network-episodic-mean-advantage', mean_adv, step)
            reward_network.rewrnn.init_hidden()
            a2c_network.value_network.valrnn.init_hidden()
        save_a2c_model(a2c_network, save_paths)
    return a2c_network


def a2c_curriculum_training(train_data, a2c_network, reward_network, optimizer, plot_dir, save_paths, batch_size,
                            epochs, curriculum, chain_semantics="batch", policy_dtype="f32"):
    """Curriculum variant (trainers.py:503-616): for each level, start from the
    ground-truth prefix captions[:, :caplen-level] and roll out `level` steps; minibatches
    whose captions are too short for the level are skipped."""
    writer = make_writer(plot_dir)
    print_green('[Training] Training Advantage Actor-Critic Network')
    print_green(f'[Training] mode set to curriculum training using levels: {curriculum}')
    engine = build_engine(a2c_network, reward_network, chain_semantics=chain_semantics, policy_dtype=policy_dtype)
    ddp.decorrelate_numpy_stream()
    for level in curriculum:
        print_green(f'[Training] Training curriculum level: {level}')
        for epoch in range(epochs):
            for minibatch_id, (captions, features, _) in enumerate(
                    minibatches(train_data, batch_size=batch_size, split='train')):
                curr_seq_len = caption_length(captions) - level
                # every rank or none: a rank that skipped alone would leave the others waiting in the all-reduce
                if ddp.agree_min(curr_seq_len) >= 1:
                    res = _a2c_minibatch(engine, optimizer, captions, features, curr_seq_len, level).check()
                    loss, _, _, mean_reward, mean_adv = res.scalars[:5].tolist()
                    step = global_minibatch_number(epoch, minibatch_id, batch_size)
                    tag = 'A2C Curriculum' + ' Level-' + str(level)
                    writer.add_scalar(tag + '-loss', loss, step)
                    writer.add_scalar(tag + '-mean-rewards', mean_reward, step)
                    writer.add_scalar(tag + '-mean-advantage', mean_adv, step)
                reward_network.rewrnn.init_hidden()
                a2c_network.value_network.valrnn.init_hidden()
            save_a2c_model(a2c_network, save_paths)
    return a2c_network


def _ce_weights(captions, steps):
    """Per row: how many of the `steps` teacher-forced positions count, and the weight of each
    (trainers.py:238-242: (caplen / N) * mean over the first caplen positions, slices clipped to the
    positions that exist)."""
    caps = np.asarray(captions)
    n = caps.shape[0]
    caplen = np.array([np.nonzero(caps[i] == 2)[0][0] + 1 for i in range(n)])
    valid = np.minimum(caplen, steps)
    return valid.astype(np.int32), (caplen / (n * valid)).astype(np.float32)


def _save_if_best(loss, best_loss, network, path):
    if loss < best_loss:
        if ddp.is_main():
            torch.save(network.state_dict(), path)
        return loss
    return best_loss


def _after_checkpoints():
    """Checkpoints are written by rank 0: the others wait here before anything reads them."""
    if ddp.world_size() > 1:
        torch.distributed.barrier()


def train_policy_network(train_data, network_paths, plot_dir, bidirectional, epochs=100, batch_size=512):
    """Supervised pre-training of the policy network (trainers.py:202-257): teacher forcing, cross
    entropy weighted by caption length, Adam(lr=1e-3); the parameters are saved whenever a minibatch
    improves on the best loss seen (before that minibatch's update, as in the reference)."""
    fdim = int(np.asarray(train_data["train_features"]).shape[1])
    policy_network = PolicyNetwork(train_data["word_to_idx"], input_dim=fdim,
                                   pretrained_embeddings=train_data["embeddings"], bidirectional=bidirectional).to(device)
    flat = flatten_module(policy_network)
    ddp.broadcast_parameters(flat.flat)
    optimizer = FusedAdam(flat, lr=0.001)
    engine = A2CEngine(_params_of(policy_network), None, None, _grads_of(policy_network))
    writer = make_writer(plot_dir)
    best_loss = float("inf")
    print_green('[Training] Training Policy Network')
    for epoch in range(epochs):
        for minibatch_id, (captions, features, _) in enumerate(
                minibatches(train_data, batch_size=batch_size, split='train')):
            caps = torch.as_tensor(np.asarray(captions), device=device).long().contiguous()
            feats = torch.as_tensor(features, device=device).float().contiguous()
            valid, weight = _ce_weights(captions, caps.shape[1] - 1)
            loss = engine.policy_ce_step(feats, caps, torch.from_numpy(valid).to(device),
                                         torch.from_numpy(weight).to(device)).item()
            best_loss = _save_if_best(loss, best_loss, policy_network, network_paths["policy_network"])
            writer.add_scalar('Policy Network-loss', loss, global_minibatch_number(epoch, minibatch_id, batch_size))
            ddp.average_gradients(flat.flat_grad)
            optimizer.step()
    _after_checkpoints()
    return policy_network


def train_value_network(train_data, network_paths, plot_dir, bidirectional, epochs=50, batch_size=512):
    """Supervised pre-training of the value network (trainers.py:125-199): for each minibatch the
    frozen policy network decodes greedily, the frozen reward network scores the result, and the value
    network's estimate for a random-length prefix (random.randint(1, MAX_SEQ_LEN) columns) is regressed
    on that reward with nn.MSELoss, Adam(lr=1e-3)."""
    import random
    w2i, emb = train_data["word_to_idx"], train_data["embeddings"]
    fdim = int(np.asarray(train_data["train_features"]).shape[1])
    reward_network = RewardNetwork(w2i, pretrained_embeddings=emb, bidirectional=bidirectional, feature_dim=fdim).to(device)
    reward_network.load_state_dict(torch.load(network_paths["reward_network"], map_location=device, weights_only=True),
                                   strict=False)
    reward_network.train(False).requires_grad_(False)
    policy_network = PolicyNetwork(w2i, input_dim=fdim, pretrained_embeddings=emb, bidirectional=bidirectional).to(device)
    policy_network.load_state_dict(torch.load(network_paths["policy_network"], map_location=device, weights_only=True),
                                   strict=False)
    policy_network.train(False).requires_grad_(False)
    value_network = ValueNetwork(w2i, pretrained_embeddings=emb, bidirectional=bidirectional, feature_dim=fdim).to(device)
    flat = flatten_module(value_network)
    ddp.broadcast_parameters(flat.flat)
    optimizer = FusedAdam(flat, lr=0.001)
    engine = A2CEngine(None, _params_of(value_network), None, None, _grads_of(value_network))
    writer = make_writer(plot_dir)
    best_loss = float('inf')
    print_green('[Training] Training Value Network')
    for epoch in range(epochs):
        for minibatch_id, (captions, features, _) in enumerate(
                minibatches(train_data, batch_size=batch_size, split='train')):
            feats = torch.as_tensor(features, device=device).float().contiguous()
            gen_caps = GenerateCaptionsGreedy(feats, captions, policy_network)
            rewards = GetRewards(feats, gen_caps, reward_network)
            columns = random.randint(1, MAX_SEQ_LEN)
            loss, _ = engine.value_mse_step(feats, gen_caps[:, :columns].contiguous(), rewards.view(-1))
            loss = loss.item()
            if int(engine._buffers(feats.shape[0], columns, 1)["status"].abs().sum()) != 0:
                raise RuntimeError("a persistent chain kernel timed out")
            best_loss = _save_if_best(loss, best_loss, value_network, network_paths["value_network"])
            writer.add_scalar('Value Network-loss', loss, global_minibatch_number(epoch, minibatch_id, batch_size))
            ddp.average_gradients(flat.flat_grad)
            optimizer.step()
            value_network.valrnn.init_hidden()
            reward_network.rewrnn.init_hidden()
    _after_checkpoints()
    return value_network


def train_reward_network(train_data, network_paths, plot_dir, bidirectional, epochs=50, batch_size=512):
    """Supervised pre-training of the reward network (trainers.py:260-309): caption and image are
    embedded into one space and trained with VisualSemanticEmbeddingLoss (trainers.py:21-54),
    Adam(lr=1e-4); saved whenever a minibatch improves on the best loss."""
    fdim = int(np.asarray(train_data["train_features"]).shape[1])
    reward_network = RewardNetwork(train_data["word_to_idx"], pretrained_embeddings=train_data["embeddings"],
                                   bidirectional=bidirectional, feature_dim=fdim).to(device)
    flat = flatten_module(reward_network)
    ddp.broadcast_parameters(flat.flat)
    optimizer = FusedAdam(flat, lr=0.0001)
    engine = A2CEngine(None, None, _params_of(reward_network), reward_grads=_grads_of(reward_network))
    writer = make_writer(plot_dir)
    best_loss = float('inf')
    print_green('[Training] Training Reward Network')
    for epoch in range(epochs):
        for minibatch_id, (captions, features, _) in enumerate(
                minibatches(train_data, batch_size=batch_size, split='train')):
            caps = torch.as_tensor(np.asarray(captions), device=device).long().contiguous()
            feats = torch.as_tensor(features, device=device).float().contiguous()
            loss = engine.reward_vse_step(feats, caps).item()
            if int(engine._vse_status.abs().sum()) != 0:
                raise RuntimeError("a persistent chain kernel timed out")
            best_loss = _save_if_best(loss, best_loss, reward_network, network_paths["reward_network"])
            writer.add_scalar('Reward Network-loss', loss, global_minibatch_number(epoch, minibatch_id, batch_size))
            ddp.average_gradients(flat.flat_grad)
            optimizer.step()
            reward_network.rewrnn.init_hidden()
    _after_checkpoints()
    return reward_network


def _load_or_train(make, path, what, train):
    """A pre-trained network from its checkpoint, or - when the file is missing - from its
    pre-training loop, as trainers.py:345-370 does."""
    try:
        net = make()
        net.load_state_dict(torch.load(path, map_location=device, weights_only=True), strict=False)
        print(f'[Training] loaded {what}')
        return net
    except FileNotFoundError:
        print(f'[Training] {what} not found')
        return train()


def train_a2c_network(train_data, save_paths, network_paths, plot_dir, bidirectional, epochs, batch_size,
                      retrain_all=False, curriculum=None, chain_semantics="batch", policy_dtype="f32"):
    """Entry point (trainers.py:312-399): the three pre-trained networks are loaded from their
    checkpoints (or trained by their pre-training loops when a file is missing, or all of them with
    retrain_all), the reward network is frozen, Adam(lr=1e-4) runs over value + policy, the plain or
    the curriculum loop is dispatched and the network description is appended to the results file."""
    model_save_path = save_paths["model_path"]
    results_save_path = save_paths["results_path"]
    w2i, emb = train_data["word_to_idx"], train_data["embeddings"]
    fdim = int(np.asarray(train_data["train_features"]).shape[1])
    per_sample = chain_semantics == "per_sample"

    def train_reward():
        return train_reward_network(train_data, network_paths, plot_dir, bidirectional, batch_size=batch_size)

    def train_policy():
        return train_policy_network(train_data, network_paths, plot_dir, bidirectional, batch_size=batch_size)

    def train_value():
        return train_value_network(train_data, network_paths, plot_dir, bidirectional, batch_size=batch_size)

    if retrain_all:
        print_green('[Training] Training all the networks')
        reward_network, policy_network, value_network = train_reward(), train_policy(), train_value()
        print_green('[Training] All networks trained')
    else:
        reward_network = _load_or_train(
            lambda: RewardNetwork(w2i, pretrained_embeddings=emb, bidirectional=bidirectional, feature_dim=fdim,
                                  per_sample=per_sample).to(device),
            network_paths["reward_network"], "reward network", train_reward)
        policy_network = _load_or_train(
            lambda: PolicyNetwork(w2i, input_dim=fdim, pretrained_embeddings=emb, bidirectional=bidirectional).to(device),
            network_paths["policy_network"], "policy network", train_policy)
        value_network = _load_or_train(
            lambda: ValueNetwork(w2i, pretrained_embeddings=emb, bidirectional=bidirectional, feature_dim=fdim,
                                 per_sample=per_sample).to(device),
            network_paths["value_network"], "value network", train_value)
    # the pre-training loops always run the reference's batch-1 recurrences; what the trained modules
    # compute from here on (A2C steps, decoding) follows chain_semantics
    value_network.valrnn.per_sample = reward_network.rewrnn.per_sample = per_sample
    reward_network.requires_grad_(False)
    reward_network.train(False)
    a2c_network = AdvantageActorCriticNetwork(value_network, policy_network).to(device)
    a2c_network.train(True)
    flat = flatten_module(a2c_network)
    ddp.broadcast_parameters(flat.flat)
    optimizer = FusedAdam(flat, lr=0.0001)

    print(f'[Training] train_data len = {len(train_data["train_captions"])}')
    print(f'[Training] episodes = {batch_size}')
    print(f'[Training] epochs = {epochs}')

    save_paths = [model_save_path, network_paths["a2c_network"]]
    if curriculum is None:
        a2c_network = a2c_training(train_data, a2c_network, reward_network, optimizer, plot_dir, save_paths,
                                   batch_size, epochs, chain_semantics=chain_semantics, policy_dtype=policy_dtype)
    else:
        if 16 not in curriculum:
            curriculum.append(16)  # a curriculum always ends with the whole 16-step rollout (trainers.py:389-390)
        a2c_network = a2c_curriculum_training(train_data, a2c_network, reward_network, optimizer, plot_dir,
                                              save_paths, batch_size, epochs, curriculum,
                                              chain_semantics=chain_semantics, policy_dtype=policy_dtype)
    if ddp.is_main():
        with open(results_save_path, 'a') as f:
            f.write('\n' + '-' * 10 + ' network ' + '-' * 10 + '\n')
            f.write(str(a2c_network))
            f.write('\n' + '-' * 10 + ' network ' + '-' * 10 + '\n')
    return a2c_network
