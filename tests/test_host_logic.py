"""Host-side logic of the drop-in surface that needs no GPU."""
import importlib

import numpy as np
import pytest
import torch

from helpers import synthetic

models = importlib.import_module("image-captioning-through-rl_amd.models")
trainers = importlib.import_module("image-captioning-through-rl_amd.trainers")
utilities = importlib.import_module("image-captioning-through-rl_amd.utilities")
optim = importlib.import_module("image-captioning-through-rl_amd.optim")
hip = importlib.import_module("image-captioning-through-rl_amd.hipabi")


def test_state_dict_keys_and_shapes_match_the_reference_layout():
    vocab, fdim = 37, 2048
    w2i = synthetic.make_word_to_idx(vocab)
    nets = dict(policy=models.PolicyNetwork(w2i, input_dim=fdim), value=models.ValueNetwork(w2i, feature_dim=fdim),
                reward=models.RewardNetwork(w2i, feature_dim=fdim))
    for kind, net in nets.items():
        want = synthetic.make_params(kind, vocab, fdim, 0)
        sd = net.state_dict()
        assert list(sd.keys()) == list(want.keys())
        assert all(tuple(sd[k].shape) == want[k].shape for k in sd)
    a2c = models.AdvantageActorCriticNetwork(nets["value"], nets["policy"])
    keys = list(a2c.state_dict().keys())
    assert keys[0] == "value_network.valrnn.caption_embedding.weight"
    assert "policy_network.linear2vocab.bias" in keys and len(keys) == 18
    assert models.MAX_SEQ_LEN == 17
    assert nets["value"].valrnn.hidden_cell[0].shape == (1, 1, 512)
    assert nets["reward"].rewrnn.hidden_cell.shape == (1, 1, 512)


def test_pretrained_embeddings_are_frozen_and_set_the_word_vector_width():
    emb = np.random.RandomState(0).standard_normal((20, 300)).astype(np.float32)
    pol = models.PolicyNetwork(synthetic.make_word_to_idx(20), pretrained_embeddings=emb)
    assert not pol.caption_embedding.weight.requires_grad
    assert pol.lstm.weight_ih_l0.shape == (2048, 300)
    with pytest.raises(NotImplementedError):
        models.ValueNetwork(synthetic.make_word_to_idx(20), bidirectional=True)


def test_minibatch_iterator_order_and_caption_length():
    data = synthetic.make_train_data(10, 50, 7, 16, 3, max_len=9)
    torch.manual_seed(11)
    perm = torch.randperm(10).numpy()
    torch.manual_seed(11)
    batches = list(utilities.get_coco_minibatches(data, batch_size=4))
    assert [b[0].shape[0] for b in batches] == [4, 4, 2]
    got = np.concatenate([b[0] for b in batches])
    assert np.array_equal(got, data["train_captions"][perm])
    assert np.array_equal(np.concatenate([b[1] for b in batches]), data["train_features"][perm])
    assert trainers.caption_length(data["train_captions"]) == 7
    assert utilities.global_minibatch_number(3, 5, 512) == 3 * 512 + 5


def test_flatten_module_keeps_values_and_aliases_one_buffer():
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(3, 5), torch.nn.Linear(5, 2))
    before = {k: v.clone() for k, v in net.state_dict().items()}
    flat = optim.flatten_module(net)
    for k, v in net.state_dict().items():
        assert torch.equal(v, before[k])
    assert flat.flat.numel() % 64 == 0
    flat.flat.zero_()
    assert all(float(p.detach().abs().sum()) == 0 for p in net.parameters())
    flat.flat_grad.fill_(2.0)
    assert all(float(p.grad.min()) == 2.0 for p in net.parameters())


def test_scalar_log_and_missing_library_message(monkeypatch, tmp_path):
    log = trainers.make_writer(trainers.ScalarLog())
    log.add_scalar("x", torch.tensor(1.5), 3)
    assert log.records == [("x", 1.5, 3)]
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(hip.HipExtensionMissing, match="no CPU fallback"):
        hip.load()


def test_train_a2c_network_falls_back_to_pretraining(tmp_path, monkeypatch):
    """A missing checkpoint sends train_a2c_network to that network's pre-training loop, and retrain_all
    to all three, as trainers.py:337-370 does (the loops themselves need the GPU: they are stubbed here)."""
    data = synthetic.make_train_data(4, 30, 5, 512, 0)
    paths = dict(reward_network=str(tmp_path / "r.pt"), policy_network=str(tmp_path / "p.pt"),
                 value_network=str(tmp_path / "v.pt"), a2c_network=str(tmp_path / "a.pt"))
    save = dict(model_path=str(tmp_path / "m.pt"), results_path=str(tmp_path / "res.txt"))
    called = []

    class Stop(Exception):
        pass

    def stub(name):
        def run(*a, **k):
            called.append(name)
            raise Stop(name)
        return run
    for name in ("train_reward_network", "train_policy_network", "train_value_network"):
        monkeypatch.setattr(trainers, name, stub(name))
    with pytest.raises(Stop):
        trainers.train_a2c_network(data, save, paths, str(tmp_path), False, 1, 4)
    assert called == ["train_reward_network"]
    with pytest.raises(Stop):
        trainers.train_a2c_network(data, save, paths, str(tmp_path), False, 1, 4, retrain_all=True)
    assert called == ["train_reward_network", "train_reward_network"]


def test_warmup_ladder_moves_up_before_failing_and_probes_down_with_a_cooldown():
    engine = importlib.import_module("image-captioning-through-rl_amd.engine")
    warmups = (64, 128, 192, 256, 320, 384, 448)       # rungs of 64 steps: six calm minibatches before a probe
    lad, last, tol = engine._WarmupLadder(warmups), 6, 1e-5
    lad.accepted(0, [(0.6 * tol, tol)], last)          # passed with less than 2x margin
    assert lad.first == 1
    lad.rejected(3)                                    # a rejected length: the estimate is taken
    assert lad.first == 3
    for _ in range(6):                                 # error at the rounding floor: probe one rung down
        assert lad.first == 3
        lad.accepted(3, [(0.01 * tol, tol)], last)
    assert lad.first == 2 and lad.probing
    lad.rejected(3)                                    # the probe did not hold
    assert lad.first == 3 and lad.cooldown == 32 and lad.penalty == 64
    for _ in range(31):
        lad.accepted(3, [(0.01 * tol, tol)], last)
    assert lad.first == 3                              # still cooling down
    lad.accepted(3, [(0.01 * tol, tol)], last)
    assert lad.first == 2 and lad.probing
    lad.accepted(2, [(0.3 * tol, tol)], last)          # the probe holds with margin: stay
    assert lad.first == 2 and not lad.probing
    for _ in range(20):                                # neither at the floor nor close to failing: stay
        lad.accepted(2, [(0.3 * tol, tol)], last)
    assert lad.first == 2
    lad.accepted(5, [(0.9 * tol, tol)], last)          # a result from another rung changes nothing
    assert lad.first == 2
    # an error that doubles from minibatch to minibatch moves the ladder up long before it fails
    lad = engine._WarmupLadder(warmups, 2)
    for frac in (0.11, 0.22):
        lad.accepted(2, [(frac * tol, tol)], last)
    assert lad.first == 2
    lad.accepted(2, [(0.44 * tol, tol)], last)         # growth ~2 per minibatch: 0.44 * 2^2 > 1
    assert lad.first == 3
    # short rungs wait longer before a probe
    lad = engine._WarmupLadder((80, 96, 128), 1)
    for _ in range(63):
        lad.accepted(1, [(0.05 * tol, tol)], 2)
    assert lad.first == 1
    lad.accepted(1, [(0.05 * tol, tol)], 2)
    assert lad.first == 0 and lad.probing


def test_step_result_check_uses_the_status_words_read_back_with_the_seam():
    engine = importlib.import_module("image-captioning-through-rl_amd.engine")
    dev_status = torch.zeros(4, dtype=torch.int32)
    ok = engine.A2CStepResult(None, None, None, None, None, None, dev_status, status_host=[0, 0, 0, 0])
    assert ok.check() is ok
    bad = engine.A2CStepResult(None, None, None, None, None, None, dev_status, status_host=[0, 0, 1, 0])
    with pytest.raises(RuntimeError, match="timed out"):
        bad.check()
    # without a host copy the device words are read
    dev_status[1] = 1
    with pytest.raises(RuntimeError, match="timed out"):
        engine.A2CStepResult(None, None, None, None, None, None, dev_status).check()


def test_full_calls_follow_the_warmups_and_the_failed_replay_checks():
    """How many calls the chains walk in full: the fewest whose left-out replays are long enough for the replay
    check's steps, the backward warm-up and whatever a failed check asked for; every call when that does not
    shorten the chain by a quarter."""
    engine = importlib.import_module("image-captioning-through-rl_amd.engine")
    eng = object.__new__(engine.A2CEngine)
    eng.collapse, eng._replay_need = True, 0
    assert eng._full_calls(512, 1, 19, 96, 96) == 1            # replays of >= 512 positions, 128 checked
    assert eng._full_calls(512, 1, 19, 448, 96) == 1           # 448 steps to check still fit a 512-position replay
    assert eng._full_calls(512, 1, 19, 640, 96) == 2           # 640 do not: one more call in full
    assert eng._full_calls(512, 1, 19, 96, 600) == 2           # a gradient that reaches 600 positions back
    assert eng._full_calls(64, 1, 19, 96, 96) == 2             # short columns: two of them per replay
    assert eng._full_calls(512, 4, 16, 96, 96) == 1            # a curriculum level's longer prefix
    assert eng._full_calls(4, 1, 9, 96, 96) == 9               # 4 rows: no replay is ever long enough
    eng._replay_need = 5000
    assert eng._full_calls(512, 1, 19, 96, 96) == 10
    eng._replay_need = 9000
    assert eng._full_calls(512, 1, 19, 96, 96) == 19           # from 18 calls in full on nothing is gained
    eng.collapse = False
    assert eng._full_calls(512, 1, 19, 96, 96) == 19
    assert hip.chain_positions(512, 1, 19) == 97280 and hip.chain_positions(512, 1, 19, 1) == 9728
    assert hip.chain_positions(512, 1, 19, 2) == 512 * (1 + 2 + 17)


def test_load_data_reads_the_coco_layout(tmp_path, monkeypatch):
    """load_data (reference utilities.py:45-113) against files of the reference's layout.  h5py is not
    installed here, so a stand-in module that serves arrays from memory takes its place: the file names,
    keys, the max_train draw from NumPy's global stream and the caption lengths are what is checked; real
    HDF5 decoding is h5py's business (parity unpinned for it)."""
    import json
    import os
    import sys
    import types
    utilities = importlib.import_module("image-captioning-through-rl_amd.utilities")
    rs = np.random.RandomState(0)
    caps = rs.randint(3, 50, size=(9, 7))
    caps[:, 0] = 1
    ends = rs.randint(2, 7, size=9)
    for i, e in enumerate(ends):
        caps[i, e] = 2
        caps[i, e + 1:] = 0
    files = {
        "coco2014_captions.h5": {"train_captions": caps[:6], "train_image_idxs": np.arange(6) % 4,
                                 "val_captions": caps[6:], "val_image_idxs": np.arange(3) % 2},
        "train2014_vgg16_fc7_pca.h5": {"features": rs.randn(4, 8).astype(np.float32)},
        "val2014_vgg16_fc7_pca.h5": {"features": rs.randn(2, 8).astype(np.float32)},
        "train2014_vgg16_fc7.h5": {"features": rs.randn(4, 16).astype(np.float32)},
        "val2014_vgg16_fc7.h5": {"features": rs.randn(2, 16).astype(np.float32)},
    }

    class FakeFile(dict):
        def __init__(self, path, mode):
            assert mode == "r"
            super().__init__(files[os.path.basename(path)])

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            return False

    monkeypatch.setitem(sys.modules, "h5py", types.SimpleNamespace(File=FakeFile))
    (tmp_path / "coco2014_vocab.json").write_text(json.dumps({"idx_to_word": ["<NULL>", "<START>", "<END>"],
                                                              "word_to_idx": {"<NULL>": 0, "<START>": 1, "<END>": 2}}))
    (tmp_path / "train2014_urls.txt").write_text("a\nb\nc\nd\n")
    (tmp_path / "val2014_urls.txt").write_text("e\nf\n")
    data = utilities.load_data(str(tmp_path))
    assert data["train_features"].shape == (4, 8) and data["val_features"].shape == (2, 8)
    assert list(data["train_urls"]) == ["a", "b", "c", "d"] and data["word_to_idx"]["<END>"] == 2
    np.testing.assert_array_equal(data["train_captions_lens"], ends[:6] + 1.0)
    np.testing.assert_array_equal(data["val_captions_lens"], ends[6:] + 1.0)
    assert data["train_captions_lens"].dtype == np.float64
    full = utilities.load_data(str(tmp_path), pca_features=False)
    assert full["train_features"].shape == (4, 16)
    np.random.seed(5)
    want = np.random.randint(6, size=4)
    np.random.seed(5)
    sub = utilities.load_data(str(tmp_path), max_train=4)
    np.testing.assert_array_equal(sub["train_captions"], caps[:6][want])
    np.testing.assert_array_equal(sub["train_image_idxs"], (np.arange(6) % 4)[want])
    np.testing.assert_array_equal(sub["train_captions_lens"], (ends[:6] + 1.0)[want])
    caps_bad = caps.copy()
    caps_bad[1][caps_bad[1] == 2] = 5
    files["coco2014_captions.h5"]["train_captions"] = caps_bad[:6]
    with pytest.raises(ValueError, match="row 1 has no <END>"):
        utilities.load_data(str(tmp_path))
