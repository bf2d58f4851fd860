"""Per-kernel CPU references (float64 torch / numpy).  TEST INFRASTRUCTURE ONLY.

Each function restates one entry point of include/icrl_a2c.h on the CPU so that a
GPU test can check the HIP kernel in isolation; the end-to-end oracle they are
consistent with is a2c_oracle.py (pinned against the reference's golden vectors)."""
import numpy as np
import torch

from . import a2c_oracle as orc

H = 512


def lstm_cell(gates_pre, c_prev):
    """PyTorch gate order i,f,g,o (models.py:80).  Returns activated gates, c, h."""
    g = torch.as_tensor(gates_pre, dtype=torch.float64)
    c_prev = torch.as_tensor(c_prev, dtype=torch.float64)
    i, f, gg, o = g.split(H, dim=1)
    i, f, gg, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)
    c = f * c_prev + i * gg
    h = o * torch.tanh(c)
    return torch.cat((i, f, gg, o), 1), c, h


def lstm_cell_grads(gates_pre, c_prev, dh, dc):
    g = torch.tensor(np.asarray(gates_pre), dtype=torch.float64, requires_grad=True)
    cp = torch.tensor(np.asarray(c_prev), dtype=torch.float64, requires_grad=True)
    _, c, h = lstm_cell(g, cp)
    loss = (h * torch.as_tensor(dh, dtype=torch.float64)).sum()
    if dc is not None:
        loss = loss + (c * torch.as_tensor(dc, dtype=torch.float64)).sum()
    loss.backward()
    return g.grad.numpy(), cp.grad.numpy()


def sample_row(p_row, u):
    return orc.inverse_cdf_sample(p_row, u)


def call_columns(p0, s, full_calls):
    """Columns call s contributes to a chain: all of 0 .. p0+s-1, or (calls past the first `full_calls`, whose
    replay of the earlier columns is left out) the last one only.  full_calls = 0: every call in full."""
    return range(p0 + s) if (not full_calls or s < full_calls) else range(p0 + s - 1, p0 + s)


def lstm_chain(xproj, w_hh, n, p0, n_calls, want_grad_of=None, full_calls=0):
    """Flattened batch-1 LSTM chain (models.py:133 with seq = N, batch = 1).

    xproj [(cols*n), 4H] already holds W_ih x + b_ih + b_hh.  Returns h_emit
    [n_calls, n, H] and saved [positions, 6, H] = (i, f, g, o, c, h).  With
    want_grad_of = d_emit [n_calls, n, H] also returns dLoss/d(gate pre-activations)
    [positions, 4H] for Loss = sum(h_emit * d_emit)."""
    xp = torch.tensor(np.asarray(xproj), dtype=torch.float64)
    whh_t = torch.tensor(np.asarray(w_hh), dtype=torch.float64).t()
    h = torch.zeros(1, H, dtype=torch.float64)
    c = torch.zeros(1, H, dtype=torch.float64)
    emit, saved, pres = [], [], []
    for s in range(n_calls):
        outs = None
        for t in call_columns(p0, s, full_calls):
            outs = []
            for i in range(n):
                pre = xp[t * n + i][None] + h @ whh_t
                if want_grad_of is not None:
                    pre.retain_grad() if pre.requires_grad else pre.requires_grad_(True)
                    pres.append(pre)
                gi, gf, gg, go = pre.split(H, dim=1)
                gi, gf, gg, go = torch.sigmoid(gi), torch.sigmoid(gf), torch.tanh(gg), torch.sigmoid(go)
                c = gf * c + gi * gg
                h = go * torch.tanh(c)
                outs.append(h[0])
                saved.append(torch.stack((gi[0], gf[0], gg[0], go[0], c[0], h[0]), 0).detach())
        emit.append(torch.stack(outs, 0))
    h_emit = torch.stack(emit, 0)
    saved = torch.stack(saved, 0)
    if want_grad_of is None:
        return h_emit.numpy(), saved.numpy()
    loss = (h_emit * torch.as_tensor(want_grad_of, dtype=torch.float64)).sum()
    loss.backward()
    dg = torch.cat([p.grad for p in pres], 0)
    return h_emit.detach().numpy(), saved.numpy(), dg.numpy()


def gru_chain(xproj, w_hh, b_hh, n, p0, n_calls, full_calls=0):
    """Flattened batch-1 GRU chain (models.py:226); call s covers p0+s+1 columns.
    xproj [(cols*n), 3H] holds W_ih x + b_ih; gate order r,z,n."""
    xp = torch.tensor(np.asarray(xproj), dtype=torch.float64)
    whh_t = torch.tensor(np.asarray(w_hh), dtype=torch.float64).t()
    bhh = torch.tensor(np.asarray(b_hh), dtype=torch.float64)
    h = torch.zeros(1, H, dtype=torch.float64)
    emit = []
    for s in range(n_calls):
        outs = None
        for t in call_columns(p0 + 1, s, full_calls):
            outs = []
            for i in range(n):
                h = orc._gru_step(xp[t * n + i][None], h, whh_t, bhh)
                outs.append(h[0])
        emit.append(torch.stack(outs, 0))
    return torch.stack(emit, 0).numpy()


def adam(p, g, m, v, lr, b1, b2, eps, step):
    p, g, m, v = [np.asarray(x, dtype=np.float64) for x in (p, g, m, v)]
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    denom = np.sqrt(v) / np.sqrt(1 - b2 ** step) + eps
    return p - (lr / (1 - b1 ** step)) * m / denom, m, v
