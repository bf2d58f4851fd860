"""Reference-shaped CPU port of one A2C minibatch, for the cpu_baseline leg of bench.py.
TEST / MEASUREMENT INFRASTRUCTURE ONLY - never on the product path.

Unlike a2c_oracle.py (which restates the arithmetic in its cheapest order), this port
keeps the reference's COST structure on purpose, because it stands in for "the reference
CPU path" in the reported baseline:
  * the policy LSTM is rerun over the whole prefix and every position is projected to
    the vocabulary at every rollout step (models.py:71-84, 286);
  * the value LSTM and reward GRU are torch.nn modules without batch_first fed (N,1,E),
    i.e. batch-1 recurrences over the rows, called once per prefix column per step with
    the hidden state carried along (models.py:130-135, 166-169, 223-228, 253-255);
  * sampling is a host loop of numpy.random.choice (trainers.py:447-449);
  * autograd backward through everything, then torch.optim.Adam (trainers.py:478-480).
It is checked against the reference's golden vectors in tests/test_oracle_golden.py.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HID = 512


class PortNets:
    """float32 torch-CPU modules holding the seeded parameters."""

    def __init__(self, policy, value, reward, vocab, feature_dim, wordvec=512):
        self.p_emb = nn.Embedding(vocab, wordvec)
        self.p_in = nn.Linear(feature_dim, HID)
        self.p_lstm = nn.LSTM(wordvec, HID, batch_first=True)
        self.p_out = nn.Linear(HID, vocab)
        self.v_emb = nn.Embedding(vocab, wordvec)
        self.v_lstm = nn.LSTM(wordvec, HID)
        self.v_l1 = nn.Linear(feature_dim + HID, 512)
        self.v_l2 = nn.Linear(512, 1)
        self.r_emb = nn.Embedding(vocab, wordvec)
        self.r_gru = nn.GRU(wordvec, HID)
        self.r_vis = nn.Linear(feature_dim, 512)
        self.r_sem = nn.Linear(HID, 512)
        self._load(policy, {"caption_embedding.weight": self.p_emb.weight, "cnn2linear": self.p_in,
                            "lstm": self.p_lstm, "linear2vocab": self.p_out})
        self._load(value, {"valrnn.caption_embedding.weight": self.v_emb.weight, "valrnn.lstm": self.v_lstm,
                           "linear1": self.v_l1, "linear2": self.v_l2})
        self._load(reward, {"rewrnn.caption_embedding.weight": self.r_emb.weight, "rewrnn.gru": self.r_gru,
                            "visual_embed": self.r_vis, "semantic_embed": self.r_sem})
        for m in (self.r_emb, self.r_gru, self.r_vis, self.r_sem):
            m.requires_grad_(False)
        self.trainable = [p for m in (self.v_emb, self.v_lstm, self.v_l1, self.v_l2, self.p_emb, self.p_in,
                                      self.p_lstm, self.p_out) for p in m.parameters()]

    @staticmethod
    def _load(params, targets):
        with torch.no_grad():
            for name, arr in params.items():
                t = torch.from_numpy(np.asarray(arr, dtype=np.float32).copy())
                if name in targets:
                    targets[name].copy_(t)
                    continue
                owner, attr = name.rsplit(".", 1)
                getattr(targets[owner], attr).copy_(t)

    def named_trainable(self):
        """(reference state_dict name, tensor) pairs of the A2C network's parameters."""
        out = [("value_network.valrnn.caption_embedding.weight", self.v_emb.weight)]
        out += [("value_network.valrnn.lstm." + k, p) for k, p in self.v_lstm.named_parameters()]
        out += [("value_network.linear1." + k, p) for k, p in self.v_l1.named_parameters()]
        out += [("value_network.linear2." + k, p) for k, p in self.v_l2.named_parameters()]
        out += [("policy_network.caption_embedding.weight", self.p_emb.weight)]
        out += [("policy_network.cnn2linear." + k, p) for k, p in self.p_in.named_parameters()]
        out += [("policy_network.lstm." + k, p) for k, p in self.p_lstm.named_parameters()]
        out += [("policy_network.linear2vocab." + k, p) for k, p in self.p_out.named_parameters()]
        return out


def a2c_minibatch(nets, optimizer, features, captions, uniforms=None):
    """One minibatch in the reference's order.  uniforms=None draws from NumPy's global
    stream through numpy.random.choice, as the reference does; otherwise the given
    doubles are consumed in (step, row) order by the equivalent inverse-CDF lookup."""
    feats = torch.as_tensor(features).float()
    caps = torch.as_tensor(np.asarray(captions)).long()
    n = caps.shape[0]
    caplen = int(np.nonzero(np.asarray(captions) == 2)[1].max()) + 1
    seq = caps[:, :1]
    v_state = (torch.zeros(1, 1, HID), torch.zeros(1, 1, HID))
    r_state = torch.zeros(1, 1, HID)
    values, rewards, logps = [], [], []
    draw = 0
    for _ in range(caplen - 1):
        # critic: one batch-1 LSTM pass of N steps per prefix column
        for t in range(seq.shape[1]):
            v_out, v_state = nets.v_lstm(nets.v_emb(seq[:, t]).view(n, 1, -1), v_state)
        joint = torch.cat((feats, v_out.squeeze(1)), dim=1)
        values.append(nets.v_l2(nets.v_l1(joint)))
        # actor: whole prefix again, all positions projected, last one kept
        h0 = nets.p_in(feats.unsqueeze(0))
        out, _ = nets.p_lstm(nets.p_emb(seq), (h0, torch.zeros_like(h0)))
        probs = F.softmax(nets.p_out(out)[:, -1:, :], dim=2)
        dist = probs.detach().numpy()[:, 0]
        picks = []
        for i in range(n):
            if uniforms is None:
                picks.append(np.random.choice(dist.shape[1], p=dist[i]))
            else:
                cdf = np.cumsum(dist[i].astype(np.float64))
                cdf /= cdf[-1]
                picks.append(int(np.searchsorted(cdf, uniforms[draw], side="right")))
                draw += 1
        act = torch.tensor(picks, dtype=torch.int64)
        seq = torch.cat((seq, act.unsqueeze(1)), dim=1)
        logps.append(torch.log(probs[:, 0, :].gather(1, act.view(-1, 1))))
        # reward: batch-1 GRU over the extended prefix, cosine of the two embeddings
        with torch.no_grad():
            for t in range(seq.shape[1]):
                r_out, r_state = nets.r_gru(nets.r_emb(seq[:, t]).view(n, 1, -1), r_state)
            sem = F.normalize(nets.r_sem(r_out.squeeze(1)), p=2, dim=1)
            vis = F.normalize(nets.r_vis(feats), p=2, dim=1)
            rewards.append((vis * sem).sum(dim=1, keepdim=True))
    values = torch.stack(values, 1).squeeze()
    rewards = torch.stack(rewards, 1).squeeze()
    logps = torch.stack(logps, 1).squeeze()
    adv = values - rewards
    loss = (-logps * adv).mean() + 0.5 * adv.pow(2).mean()
    optimizer.zero_grad()
    loss.backward()
    grad_norms = {name: float(p.grad.norm()) for name, p in nets.named_trainable() if p.grad is not None}
    optimizer.step()
    return dict(tokens=seq.numpy(), loss=float(loss.detach()), mean_reward=float(rewards.mean()),
                mean_advantage=float(adv.detach().mean()), values=values.detach().numpy(), rewards=rewards.numpy(),
                log_probs=logps.detach().numpy(), grad_norms=grad_norms)


def make_optimizer(nets):
    return torch.optim.Adam(nets.trainable, lr=1e-4)
