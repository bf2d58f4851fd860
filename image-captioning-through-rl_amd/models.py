"""The reference's models.py class surface, computing on the HIP kernels.

Same class names, constructor arguments, attribute names and ``state_dict`` keys as the
reference (models.py:33-287), so checkpoints written by either side load in the other and
``from models import *`` call sites keep working.  Differences, all additive:
  * ``feature_dim`` keyword on ValueNetwork / RewardNetwork (the reference hard-wires
    512-d PCA features, models.py:160, 250; its own PolicyNetwork already takes
    ``input_dim``);
  * ``forward`` runs hand-written gfx950 kernels through the C ABI and builds no autograd
    graph - gradients of the A2C step come from the fused backward in ``engine.py``;
  * ``bidirectional=True`` is accepted for signature compatibility and raises
    NotImplementedError (off by default in the reference, image_captioner.py:172-173).
"""
import numpy as np
import torch
import torch.nn as nn

from . import hipabi as hip

device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
MAX_SEQ_LEN = 17  # the max captions len in the dataset (models.py:18)
H = 512


# segment-parallel evaluation of a recurrence that starts from zeros (see RewardNetworkRNN._forward_segmented)
SEGMENTED_MIN_POSITIONS = 4096
SEGMENTED_WARMUPS = (96, 192, 384, 768)
SEGMENTED_SEAM_TOL = 1e-5


def repackage_hidden(h):
    """Detach hidden states from their history (models.py:20-30)."""
    if isinstance(h, torch.Tensor):
        return h.detach().to(device)
    return tuple(repackage_hidden(v) for v in h)


def _no_bidirectional(flag):
    if flag:
        raise NotImplementedError("bidirectional recurrent layers are not built on the HIP path")


def _embedding(vocab_size, wordvec_dim, pretrained):
    if pretrained is not None:
        emb = nn.Embedding.from_pretrained(torch.FloatTensor(np.asarray(pretrained)), freeze=True)
        return emb, pretrained.shape[1]
    return nn.Embedding(vocab_size, wordvec_dim), wordvec_dim


def _dev_f32(x, dev):
    return torch.as_tensor(x, device=dev).float().contiguous()


class _ChainWorkspace:
    """Granule buffer + status word of the persistent chain kernels, per device."""

    def __init__(self):
        self.sync = None
        self.status = None

    def get(self, dev):
        if self.sync is None or self.sync.device != dev:
            self.sync = torch.zeros(hip.chain_sync_bytes(), dtype=torch.uint8, device=dev)
            self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        return self.sync, self.status


class PolicyNetwork(nn.Module):
    """Actor: LSTM over caption embeddings, hidden state initialised from the image
    features, every position projected to the vocabulary (models.py:33-84)."""

    def __init__(self, word_to_idx, input_dim=512, wordvec_dim=512, hidden_dim=512,
                 pretrained_embeddings=None, bidirectional=False):
        super().__init__()
        _no_bidirectional(bidirectional)
        assert hidden_dim == H, "the HIP kernels are built for hidden_dim = 512"
        self.bidirectional = bidirectional
        self.word_to_idx = word_to_idx
        self.idx_to_word = {i: w for w, i in word_to_idx.items()}
        vocab_size = len(word_to_idx)
        self.caption_embedding, wordvec_dim = _embedding(vocab_size, wordvec_dim, pretrained_embeddings)
        self.cnn2linear = nn.Linear(input_dim, hidden_dim)
        self.lstm = nn.LSTM(wordvec_dim, hidden_dim, batch_first=True, bidirectional=False)
        self.linear2vocab = nn.Linear(hidden_dim, vocab_size)

    @torch.no_grad()
    def forward(self, features, captions, last_only=False):
        """features [1,N,F], captions [N,p] -> logits [N,p,V] (models.py:71-84).  last_only (not in the
        reference): project only the newest position, logits [N,1,V] - what a caller that slices
        [:, -1, :] needs, without the p - 1 other vocabulary products."""
        w = self.cnn2linear.weight
        feats = _dev_f32(features, w.device).reshape(-1, w.shape[1])
        caps = torch.as_tensor(captions, device=w.device).long()
        n, p = caps.shape
        tok_t = caps.t().contiguous()
        hs = torch.empty(p + 1, n, H, device=w.device)
        c_prev, c_next = torch.zeros(n, H, device=w.device), torch.empty(n, H, device=w.device)
        # the input side of every position in one product (it does not depend on the state) ...
        gates = torch.empty(p, n, 4 * H, device=w.device)
        hip.gemm(self.caption_embedding.weight, self.lstm.weight_ih_l0, gates.view(p * n, 4 * H), a_rows=tok_t.view(-1),
                 bias=self.lstm.bias_ih_l0, bias2=self.lstm.bias_hh_l0)
        hip.gemm(feats, w, hs[0], bias=self.cnn2linear.bias)
        for t in range(p):   # ... then per position the recurrent product on top of it and the cell
            hip.gemm(hs[t], self.lstm.weight_hh_l0, gates[t], accumulate=True)
            hip.lstm_cell_fwd(gates[t], c_prev, c_next, hs[t + 1])
            c_prev, c_next = c_next, c_prev
        vocab = self.linear2vocab.weight.shape[0]
        if last_only:
            logits = torch.empty(n, vocab, device=w.device)
            hip.gemm(hs[p], self.linear2vocab.weight, logits, bias=self.linear2vocab.bias)
            return logits.view(n, 1, vocab)
        logits = torch.empty(p * n, vocab, device=w.device)
        hip.gemm(hs[1:].view(p * n, H), self.linear2vocab.weight, logits, bias=self.linear2vocab.bias)
        return logits.view(p, n, vocab).permute(1, 0, 2)


class ValueNetworkRNN(nn.Module):
    """Embedding + LSTM fed (N,1,E) without batch_first: the rows of a minibatch are
    consecutive time steps of one batch-1 recurrence whose state persists in
    ``hidden_cell`` until ``init_hidden()`` (models.py:87-135)."""

    def __init__(self, word_to_idx, input_dim=512, wordvec_dim=512, hidden_dim=512,
                 pretrained_embeddings=None, bidirectional=False, per_sample=False):
        super().__init__()
        _no_bidirectional(bidirectional)
        assert hidden_dim == H
        self.bidirectional = bidirectional
        self.hidden_dim = hidden_dim
        # per_sample: every row of a batch keeps its own (h, c) across calls until init_hidden() - what the
        # reference computes when it is called with one image at a time.  Never a default.
        self.per_sample = per_sample
        self._row_state = None
        self._seg_work, self._seg_err, self._seg_warmup = None, None, 0
        self.word_to_idx = word_to_idx
        self.idx_to_word = {i: w for w, i in word_to_idx.items()}
        self.caption_embedding, wordvec_dim = _embedding(len(word_to_idx), wordvec_dim, pretrained_embeddings)
        self.init_hidden()
        self.lstm = nn.LSTM(wordvec_dim, hidden_dim, bidirectional=False)
        self._chain_ws = _ChainWorkspace()

    def init_hidden(self):
        self.hidden_cell = (torch.zeros(1, 1, self.hidden_dim).to(device), torch.zeros(1, 1, self.hidden_dim).to(device))
        self._row_state = None

    @torch.no_grad()
    def forward_columns(self, tok_t):
        """tok_t [p, N] int64 (column-major captions): runs the p*N-step chain from
        ``hidden_cell`` and returns the hidden outputs of the LAST column, [N, H]."""
        dev = self.lstm.weight_hh_l0.device
        p, n = tok_t.shape
        xproj = torch.empty(p * n, 4 * H, device=dev)
        hip.gemm(self.caption_embedding.weight, self.lstm.weight_ih_l0, xproj, a_rows=tok_t.reshape(-1),
                 bias=self.lstm.bias_ih_l0, bias2=self.lstm.bias_hh_l0)
        if self.per_sample:
            # N chains side by side: one [N,512] x W_hh^T product and one cell launch per column
            if self._row_state is None or self._row_state[0].shape[0] != n:
                self._row_state = (torch.zeros(n, H, device=dev), torch.zeros(n, H, device=dev))
            h, c = self._row_state
            for t in range(p):
                gates = xproj[t * n:(t + 1) * n]
                hip.gemm(h, self.lstm.weight_hh_l0, gates, accumulate=True)
                h_next, c_next = torch.empty_like(h), torch.empty_like(c)
                hip.lstm_cell_fwd(gates, c, c_next, h_next)
                h, c = h_next, c_next
            self._row_state = (h, c)
            self.hidden_cell = (h.view(1, n, H), c.view(1, n, H))
            return h
        state = torch.cat((self.hidden_cell[0].reshape(-1), self.hidden_cell[1].reshape(-1))).to(dev).contiguous()
        h_emit = torch.empty(1, n, H, device=dev)
        if p * n >= SEGMENTED_MIN_POSITIONS:
            # a long chain runs segment-parallel from the carried state (the first segment picks it up; the
            # others warm up from zeros and are checked at their seams), exact walk if no warm-up passes
            saved = torch.empty(p * n, 6, H, device=dev)
            if self._forward_segmented(xproj, n, p, h_emit, saved, state):
                self.hidden_cell = (saved[-1, 5].view(1, 1, H).clone(), saved[-1, 4].view(1, 1, H).clone())
                return h_emit[0]
        state_out = torch.empty_like(state)
        sync, status = self._chain_ws.get(dev)
        hip.lstm_chain_fwd(xproj, self.lstm.weight_hh_l0, n, p, 1, h_emit, None, sync, status, state_in=state,
                           state_out=state_out)
        if status.item() != 0:
            raise RuntimeError("value LSTM chain kernel timed out")
        self.hidden_cell = (state_out[:H].view(1, 1, H), state_out[H:].view(1, 1, H))
        return h_emit[0]

    def _forward_segmented(self, xproj, n, columns, h_emit, saved, state):
        """One call's chain (columns * n positions) cut into segments that run side by side after a warm-up
        (icrl_lstm_chain_fwd_seg with the carried state as state_in).  Accepted only if every seam agrees to
        rounding; the warm-up that passed is tried first next time.  False: the caller walks the chain exactly."""
        dev = xproj.device
        positions = columns * n
        segments = max(32, min(512, positions // 16 // 32 * 32))
        if self._seg_work is None or self._seg_work.numel() < hip.chain_seg_workspace_floats(segments):
            self._seg_work = torch.empty(hip.chain_seg_workspace_floats(segments), device=dev)
            self._seg_err = torch.zeros(2, device=dev)
        for warmup in (w for w in SEGMENTED_WARMUPS if w >= self._seg_warmup):
            hip.lstm_chain_fwd_seg(xproj, self.lstm.weight_hh_l0, n, columns, 1, segments, warmup, h_emit, saved,
                                   self._seg_work, self._seg_err, state_in=state,
                                   persistent=hip.chain_pair_supported(segments))
            if self._seg_err[0].item() <= SEGMENTED_SEAM_TOL:
                self._seg_warmup = warmup
                return True
        return False

    @torch.no_grad()
    def forward_columns_many(self, tok_list):
        """Several forward_columns calls in a row (each tok [p, N]; the state runs through all of them, as
        it does when the decoder scores one candidate extension after the other) as ONE chain: the calls'
        columns are stacked, the chain keeps every position's h, and each call's last column is read back
        from there.  Long chains run segment-parallel from the carried state, under the seam check.
        Returns the [calls, N, H] outputs (indexing it gives the list the separate calls would return)."""
        if self.per_sample:
            return torch.stack([self.forward_columns(tok) for tok in tok_list])
        dev = self.lstm.weight_hh_l0.device
        p, n = tok_list[0].shape
        calls = len(tok_list)
        tok_all = torch.cat([tok.reshape(-1) for tok in tok_list])
        positions = calls * p * n
        xproj = torch.empty(positions, 4 * H, device=dev)
        vocab = self.caption_embedding.weight.shape[0]
        if positions > 2 * vocab:
            # more positions than words (the decoder's 25 extensions repeat their prefixes): the input side of
            # every WORD once, then a row gather - row by row the same sums as the direct product
            table = torch.empty(vocab, 4 * H, device=dev)
            hip.gemm(self.caption_embedding.weight, self.lstm.weight_ih_l0, table, bias=self.lstm.bias_ih_l0,
                     bias2=self.lstm.bias_hh_l0)
            hip.gather_rows(table, tok_all, 1, positions, xproj)
        else:
            hip.gemm(self.caption_embedding.weight, self.lstm.weight_ih_l0, xproj, a_rows=tok_all,
                     bias=self.lstm.bias_ih_l0, bias2=self.lstm.bias_hh_l0)
        state = torch.cat((self.hidden_cell[0].reshape(-1), self.hidden_cell[1].reshape(-1))).to(dev).contiguous()
        saved = torch.empty(positions, 6, H, device=dev)
        h_emit = torch.empty(1, n, H, device=dev)
        done = positions >= SEGMENTED_MIN_POSITIONS and self._forward_segmented(xproj, n, calls * p, h_emit, saved, state)
        if not done:
            sync, status = self._chain_ws.get(dev)
            hip.lstm_chain_fwd(xproj, self.lstm.weight_hh_l0, n, calls * p, 1, h_emit, saved, sync, status,
                               state_in=state, state_out=torch.empty_like(state))
            if status.item() != 0:
                raise RuntimeError("value LSTM chain kernel timed out")
        self.hidden_cell = (saved[-1, 5].view(1, 1, H).clone(), saved[-1, 4].view(1, 1, H).clone())
        return saved.view(calls, p, n, 6, H)[:, p - 1, :, 5, :].contiguous()

    def forward(self, captions):
        """captions [N] -> [N,1,H] (models.py:130-135)."""
        caps = torch.as_tensor(captions, device=self.lstm.weight_hh_l0.device).long()
        return self.forward_columns(caps.view(1, -1)).unsqueeze(1)


class ValueNetwork(nn.Module):
    """Critic: last column's chain outputs concatenated with the image features, two
    linear layers with no activation between them (models.py:138-180)."""

    def __init__(self, word_to_idx, pretrained_embeddings=None, bidirectional=False, feature_dim=512,
                 per_sample=False):
        super().__init__()
        _no_bidirectional(bidirectional)
        self.bidirectional = bidirectional
        self.valrnn = ValueNetworkRNN(word_to_idx, pretrained_embeddings=pretrained_embeddings, bidirectional=False,
                                      per_sample=per_sample)
        self.linear1 = nn.Linear(feature_dim + 512, 512)
        self.linear2 = nn.Linear(512, 1)

    @torch.no_grad()
    def forward(self, features, captions):
        """features [N,F], captions [N,p] -> [N,1] (models.py:166-180)."""
        dev = self.linear1.weight.device
        feats = _dev_f32(features, dev)
        caps = torch.as_tensor(captions, device=dev).long()
        n = caps.shape[0]
        f = feats.shape[1]
        hs = self.valrnn.forward_columns(caps.t().contiguous())
        u, y = torch.empty(n, 512, device=dev), torch.empty(n, 512, device=dev)
        hip.gemm(feats, self.linear1.weight[:, :f], u, bias=self.linear1.bias)
        hip.gemm(hs, self.linear1.weight[:, f:], y)
        values = torch.empty(n, device=dev)
        hip.value_head_fwd(y, u, self.linear2.weight, self.linear2.bias, n, n, values)
        return values.view(n, 1)


    @torch.no_grad()
    def forward_many(self, features, captions_list):
        """forward(features, c) for every c of captions_list (equal shapes [N,p]) in that order, the
        recurrent state running through all of them, with the calls' chains joined into one launch
        sequence (ValueNetworkRNN.forward_columns_many).  Returns the list of [N,1] values."""
        dev = self.linear1.weight.device
        feats = _dev_f32(features, dev)
        n, f = feats.shape
        toks = [torch.as_tensor(c, device=dev).long().t().contiguous() for c in captions_list]
        u = torch.empty(n, 512, device=dev)
        hip.gemm(feats, self.linear1.weight[:, :f], u, bias=self.linear1.bias)
        hs = self.valrnn.forward_columns_many(toks)            # [calls, N, 512]
        calls = hs.shape[0]
        # the head for all calls at once: its rows do not depend on each other
        y, values = torch.empty(calls * n, 512, device=dev), torch.empty(calls * n, device=dev)
        hip.gemm(hs.view(calls * n, 512), self.linear1.weight[:, f:], y)
        hip.value_head_fwd(y, u, self.linear2.weight, self.linear2.bias, calls * n, n, values)
        return [values[c * n:(c + 1) * n].view(n, 1) for c in range(calls)]


class RewardNetworkRNN(nn.Module):
    """Embedding + GRU, batch-as-sequence like ValueNetworkRNN (models.py:183-228)."""

    def __init__(self, word_to_idx, input_dim=512, wordvec_dim=512, hidden_dim=512,
                 pretrained_embeddings=None, bidirectional=False, per_sample=False):
        super().__init__()
        _no_bidirectional(bidirectional)
        assert hidden_dim == H
        self.bidirectional = bidirectional
        self.hidden_dim = hidden_dim
        self.per_sample = per_sample   # one recurrent state per row, as in ValueNetworkRNN
        self._row_state = None
        self._seg_work, self._seg_err, self._seg_warmup = None, None, 0
        self.word_to_idx = word_to_idx
        self.idx_to_word = {i: w for w, i in word_to_idx.items()}
        self.caption_embedding, wordvec_dim = _embedding(len(word_to_idx), wordvec_dim, pretrained_embeddings)
        self.init_hidden()
        self.gru = nn.GRU(wordvec_dim, hidden_dim, bidirectional=False)
        self._chain_ws = _ChainWorkspace()

    @property
    def hidden_cell(self):
        return self._hidden_cell

    @hidden_cell.setter
    def hidden_cell(self, value):   # a state assigned from outside is not known to be zeros
        self._hidden_cell = value
        self._fresh = False

    def init_hidden(self):
        self.hidden_cell = torch.zeros(1, 1, self.hidden_dim).to(device)
        self._row_state = None
        self._fresh = True   # the state is zeros: a long chain may run segment-parallel

    @torch.no_grad()
    def forward_columns(self, tok_t):
        dev = self.gru.weight_hh_l0.device
        p, n = tok_t.shape
        xproj = torch.empty(p * n, 3 * H, device=dev)
        hip.gemm(self.caption_embedding.weight, self.gru.weight_ih_l0, xproj, a_rows=tok_t.reshape(-1),
                 bias=self.gru.bias_ih_l0)
        if self.per_sample:
            if self._row_state is None or self._row_state.shape[0] != n:
                self._row_state = torch.zeros(n, H, device=dev)
            h = self._row_state
            gh = torch.empty(n, 3 * H, device=dev)
            for t in range(p):
                hip.gemm(h, self.gru.weight_hh_l0, gh, bias=self.gru.bias_hh_l0)
                h_next = torch.empty_like(h)
                hip.gru_cell_fwd(xproj[t * n:(t + 1) * n], gh, h, h_next)
                h = h_next
            self._row_state = h
            self.hidden_cell = h.view(1, n, H)
            return h
        h_emit = torch.empty(1, n, H, device=dev)
        fresh, self._fresh = self._fresh, False
        if fresh and p * n >= SEGMENTED_MIN_POSITIONS and self._forward_segmented(xproj, n, p, h_emit):
            self.hidden_cell = h_emit[0, n - 1].view(1, 1, H).clone()
            return h_emit[0]
        state = self.hidden_cell.reshape(-1).to(dev).contiguous()
        state_out = torch.empty_like(state)
        sync, status = self._chain_ws.get(dev)
        hip.gru_chain_fwd(xproj, self.gru.weight_hh_l0, self.gru.bias_hh_l0, n, p, 1, h_emit, sync, status,
                          state_in=state, state_out=state_out)
        if status.item() != 0:
            raise RuntimeError("reward GRU chain kernel timed out")
        self.hidden_cell = state_out.view(1, 1, H)
        return h_emit[0]

    def _forward_segmented(self, xproj, n, p, h_emit):
        """A chain that starts from zeros, cut into segments that run side by side after a warm-up
        (icrl_gru_chain_fwd_seg).  Accepted only if every segment's warmed-up state met the state its
        neighbour really ended in to rounding; the warm-up that passed is tried first next time.
        False: no warm-up length passed, the caller walks the chain exactly."""
        dev = xproj.device
        segments = max(32, min(512, p * n // 16 // 32 * 32))
        if self._seg_work is None or self._seg_work.numel() < hip.chain_seg_workspace_floats(segments):
            self._seg_work = torch.empty(hip.chain_seg_workspace_floats(segments), device=dev)
            self._seg_err = torch.zeros(2, device=dev)
        for warmup in (w for w in SEGMENTED_WARMUPS if w >= self._seg_warmup):
            hip.gru_chain_fwd_seg(xproj, self.gru.weight_hh_l0, self.gru.bias_hh_l0, n, p, 1, segments, warmup, h_emit,
                                  self._seg_work, self._seg_err, persistent=hip.chain_pair_supported(segments))
            if self._seg_err[0].item() <= SEGMENTED_SEAM_TOL:
                self._seg_warmup = warmup
                return True
        return False

    def forward(self, captions):
        caps = torch.as_tensor(captions, device=self.gru.weight_hh_l0.device).long()
        return self.forward_columns(caps.view(1, -1)).unsqueeze(1)


class RewardNetwork(nn.Module):
    """Projects caption (GRU chain) and image features into one embedding space
    (models.py:231-262)."""

    def __init__(self, word_to_idx, pretrained_embeddings=None, bidirectional=False, feature_dim=512,
                 per_sample=False):
        super().__init__()
        _no_bidirectional(bidirectional)
        self.bidirectional = bidirectional
        self.rewrnn = RewardNetworkRNN(word_to_idx, pretrained_embeddings=pretrained_embeddings, bidirectional=False,
                                       per_sample=per_sample)
        self.visual_embed = nn.Linear(feature_dim, 512)
        self.semantic_embed = nn.Linear(512, 512)

    @torch.no_grad()
    def forward(self, features, captions):
        """-> (visual embeds [N,512], semantic embeds [N,512]) (models.py:253-262)."""
        dev = self.visual_embed.weight.device
        feats = _dev_f32(features, dev)
        caps = torch.as_tensor(captions, device=dev).long()
        n = caps.shape[0]
        hs = self.rewrnn.forward_columns(caps.t().contiguous())
        se, ve = torch.empty(n, 512, device=dev), torch.empty(n, 512, device=dev)
        hip.gemm(hs, self.semantic_embed.weight, se, bias=self.semantic_embed.bias)
        hip.gemm(feats, self.visual_embed.weight, ve, bias=self.visual_embed.bias)
        return ve, se


class AdvantageActorCriticNetwork(nn.Module):
    """Wraps the value and policy networks (models.py:265-287)."""

    def __init__(self, value_network, policy_network):
        super().__init__()
        self.value_network = value_network
        self.policy_network = policy_network

    def forward(self, features, captions):
        values = self.value_network(features, captions)
        probs = self.policy_network(torch.as_tensor(features).unsqueeze(0), captions)[:, -1:, :]
        return values, probs
