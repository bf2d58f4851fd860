"""Seeded synthetic parameters and minibatches for the A2C caption-training step.

NumPy-only, so that the golden-vector generator, the CPU oracle, the tests and
bench.py all rebuild bit-identical inputs from a seed instead of storing them
(MT19937 streams of ``numpy.random.RandomState`` are stable across versions).

Parameter names and shapes follow the reference's ``state_dict`` layout
(models.py:61-69 policy, models.py:113-120 + 158-161 value,
models.py:208-215 + 248-251 reward), with the feature width generalised from
the reference's hard-wired 512 (models.py:160, 250) to ``feature_dim``.
"""
import math

import numpy as np

HIDDEN = 512  # models.py:41, 94, 189: hidden_dim default, never overridden by callers
WORDVEC = 512  # wordvec_dim default when no pretrained vectors are given


def policy_param_shapes(vocab, feature_dim, wordvec=WORDVEC, hidden=HIDDEN):
    return [
        ("caption_embedding.weight", (vocab, wordvec)),
        ("cnn2linear.weight", (hidden, feature_dim)),
        ("cnn2linear.bias", (hidden,)),
        ("lstm.weight_ih_l0", (4 * hidden, wordvec)),
        ("lstm.weight_hh_l0", (4 * hidden, hidden)),
        ("lstm.bias_ih_l0", (4 * hidden,)),
        ("lstm.bias_hh_l0", (4 * hidden,)),
        ("linear2vocab.weight", (vocab, hidden)),
        ("linear2vocab.bias", (vocab,)),
    ]


def value_param_shapes(vocab, feature_dim, wordvec=WORDVEC, hidden=HIDDEN):
    return [
        ("valrnn.caption_embedding.weight", (vocab, wordvec)),
        ("valrnn.lstm.weight_ih_l0", (4 * hidden, wordvec)),
        ("valrnn.lstm.weight_hh_l0", (4 * hidden, hidden)),
        ("valrnn.lstm.bias_ih_l0", (4 * hidden,)),
        ("valrnn.lstm.bias_hh_l0", (4 * hidden,)),
        ("linear1.weight", (512, feature_dim + hidden)),
        ("linear1.bias", (512,)),
        ("linear2.weight", (1, 512)),
        ("linear2.bias", (1,)),
    ]


def reward_param_shapes(vocab, feature_dim, wordvec=WORDVEC, hidden=HIDDEN):
    return [
        ("rewrnn.caption_embedding.weight", (vocab, wordvec)),
        ("rewrnn.gru.weight_ih_l0", (3 * hidden, wordvec)),
        ("rewrnn.gru.weight_hh_l0", (3 * hidden, hidden)),
        ("rewrnn.gru.bias_ih_l0", (3 * hidden,)),
        ("rewrnn.gru.bias_hh_l0", (3 * hidden,)),
        ("visual_embed.weight", (512, feature_dim)),
        ("visual_embed.bias", (512,)),
        ("semantic_embed.weight", (512, hidden)),
        ("semantic_embed.bias", (512,)),
    ]


def _draw(rs, name, shape):
    """Embedding tables ~ N(0,1); everything else ~ U(-k, k), k = 1/sqrt(fan)."""
    if name.endswith("caption_embedding.weight"):
        return rs.standard_normal(shape).astype(np.float32)
    if ".lstm." in name or ".gru." in name or name.startswith("lstm."):
        k = 1.0 / math.sqrt(HIDDEN)
    elif len(shape) == 2:
        k = 1.0 / math.sqrt(shape[1])
    else:
        k = 1.0 / math.sqrt(512)
    return rs.uniform(-k, k, size=shape).astype(np.float32)


def make_params(kind, vocab, feature_dim, seed):
    """Ordered dict name -> float32 array for kind in {policy, value, reward}."""
    shapes = {"policy": policy_param_shapes, "value": value_param_shapes,
              "reward": reward_param_shapes}[kind](vocab, feature_dim)
    offset = {"policy": 1, "value": 2, "reward": 3}[kind]
    rs = np.random.RandomState(seed * 10 + offset)
    return {name: _draw(rs, name, shape) for name, shape in shapes}


def make_word_to_idx(vocab):
    return {"w%d" % i: i for i in range(vocab)}


def make_minibatch(n, vocab, caplen, feature_dim, seed, max_len=None):
    """Features ~ N(0,1) float32; captions uniform in [4, vocab) with column 0 = 1
    (<START>) and column caplen-1 = 2 (<END>); columns past caplen are 0 (<NULL>).
    trainers.py:436 derives the rollout length from the right-most <END>."""
    max_len = caplen if max_len is None else max_len
    rs = np.random.RandomState(seed * 10 + 7)
    features = rs.standard_normal((n, feature_dim)).astype(np.float32)
    captions = np.zeros((n, max_len), dtype=np.int64)
    captions[:, :caplen] = rs.randint(4, vocab, size=(n, caplen))
    captions[:, 0] = 1
    captions[:, caplen - 1] = 2
    return features, captions


def make_train_data(n_captions, vocab, caplen, feature_dim, seed, max_len=None):
    """A ``train_data`` dict with the keys the A2C path reads
    (utilities.py:168-178, trainers.py:340-362)."""
    features, captions = make_minibatch(n_captions, vocab, caplen, feature_dim, seed, max_len)
    return {
        "train_captions": captions,
        "train_image_idxs": np.arange(n_captions, dtype=np.int64),
        "train_features": features,
        "train_urls": np.array(["url%d" % i for i in range(n_captions)]),
        "word_to_idx": make_word_to_idx(vocab),
        "idx_to_word": ["w%d" % i for i in range(vocab)],
        "embeddings": None,
    }


def make_ragged_train_data(n_captions, vocab, feature_dim, seed, max_len=17):
    """make_train_data with captions of different lengths (6 .. max_len tokens, <END> last, <NULL>
    after it), as the pre-training loops meet them (trainers.py:238-242 weights rows by length)."""
    data = make_train_data(n_captions, vocab, max_len, feature_dim, seed, max_len)
    caps = data["train_captions"]
    for i in range(n_captions):
        length = max_len if i == 0 else 6 + (5 * i + seed) % (max_len - 6)
        caps[i, length - 1] = 2
        caps[i, length:] = 0
    return data


def make_uniforms(seed, count):
    """The doubles ``numpy.random.choice`` would consume after ``np.random.seed(seed)``:
    one ``random_sample()`` per call, in (step, row) order (trainers.py:447-449)."""
    return np.random.RandomState(seed).random_sample(count)
