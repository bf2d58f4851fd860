"""The parts of the reference's utilities.py that the A2C training path and the look-ahead test
loop call: minibatch iteration, caption decoding, checkpoint save / load, logging helpers.  Dataset readers, word
embedding tooling, metrics and post-processing are not part of this path."""
import os

import numpy as np
import torch
from torch import randperm, save

from .models import AdvantageActorCriticNetwork, PolicyNetwork, ValueNetwork, device


def print_green(text):
    print('\033[32m', text, '\033[0m', sep='')


def print_red(text):
    print('\033[31m', text, '\033[0m', sep='')


def decode_captions(captions, idx_to_word):
    """Token ids -> strings, dropping <NULL> and stopping after <END> (utilities.py:116-140)."""
    captions = np.asarray(captions.cpu() if torch.is_tensor(captions) else captions)
    singleton = captions.ndim == 1
    if singleton:
        captions = captions[None]
    decoded = []
    for row in captions:
        words = []
        for tok in row:
            word = idx_to_word[int(tok)]
            if word != '<NULL>':
                words.append(word)
            if word == '<END>':
                break
        decoded.append(' '.join(words))
    return decoded[0] if singleton else decoded


def get_coco_batch(data, batch_size=100, split='train'):
    """One random batch (with replacement, NumPy's global stream): (captions, features, urls)
    (utilities.py:143-157)."""
    split_total_size = data['%s_captions' % split].shape[0]
    mask = np.random.choice(split_total_size, batch_size)
    captions = data['%s_captions' % split][mask]
    image_idxs = data['%s_image_idxs' % split][mask]
    return captions, data['%s_features' % split][image_idxs], data['%s_urls' % split][image_idxs]


def get_coco_minibatches(data, batch_size=100, split='train'):
    """Shuffled pass over the split: yields (captions [n,L], image_features [n,F], urls)
    per minibatch (utilities.py:160-178).  Same torch.randperm draw as the reference, so
    the same torch seed gives the same minibatch order."""
    split_total_size = data['%s_captions' % split].shape[0]
    permutation = randperm(split_total_size)
    for i in range(0, split_total_size, batch_size):
        mask = permutation[i: i + batch_size]
        captions = data['%s_captions' % split][mask]
        image_idxs = data['%s_image_idxs' % split][mask]
        image_features = data['%s_features' % split][image_idxs]
        urls = data['%s_urls' % split][image_idxs]
        yield captions, image_features, urls


class DeviceFeatureStore:
    """The feature matrix resident in HBM: a minibatch's features are a coalesced device row gather
    (icrl_gather_rows) instead of the host gather + PCIe copy per step of get_coco_minibatches
    (utilities.py:160-178).  Captions stay on the host (the training loops read their length there)."""

    def __init__(self, data, split='train', dev=None):
        dev = device if dev is None else dev
        self.split = split
        self.features = torch.as_tensor(np.asarray(data['%s_features' % split])).float().to(dev).contiguous()
        self.image_idxs = np.asarray(data['%s_image_idxs' % split])
        self.captions = np.asarray(data['%s_captions' % split])
        self.urls = data.get('%s_urls' % split)

    def minibatches(self, batch_size):
        """Same order and rows as get_coco_minibatches under the same torch seed (one randperm draw);
        yields (captions [n,L] host array, features [n,F] float32 device tensor, urls or None)."""
        from . import hipabi as hip
        total = self.captions.shape[0]
        permutation = randperm(total)
        for i in range(0, total, batch_size):
            mask = permutation[i: i + batch_size].numpy()
            idx_np = self.image_idxs[mask]
            idx = torch.as_tensor(idx_np).long().to(self.features.device)
            feats = torch.empty(idx.numel(), self.features.shape[1], dtype=torch.float32, device=self.features.device)
            hip.gather_rows(self.features, idx, 1, idx.numel(), feats)
            urls = None if self.urls is None else np.asarray(self.urls)[idx_np]
            yield self.captions[mask], feats, urls


def minibatches(data, batch_size=100, split='train'):
    """get_coco_minibatches, or the device-resident store when the data dict carries one under
    'feature_store' (built once by the caller: data['feature_store'] = DeviceFeatureStore(data))."""
    store = data.get('feature_store') if hasattr(data, 'get') else None
    if store is not None and store.split == split:
        return store.minibatches(batch_size)
    return get_coco_minibatches(data, batch_size=batch_size, split=split)


def global_minibatch_number(epoch, batch_id, batch_size):
    """x-axis of the logged scalars (utilities.py:204-212; epoch * batch_size, as there)."""
    return epoch * batch_size + batch_id


def save_a2c_model(model, save_paths):
    """utilities.py:286-296."""
    if isinstance(save_paths, list):
        for path in save_paths:
            save(model.state_dict(), path)
    elif isinstance(save_paths, str):
        save(model.state_dict(), save_paths)


def load_a2c_models(model_path, train_data, network_paths, bidirectional, feature_dim=512):
    """utilities.py:299-323: rebuild the three modules and load their state_dicts."""
    policy_network = PolicyNetwork(train_data["word_to_idx"], input_dim=feature_dim,
                                   pretrained_embeddings=train_data["embeddings"],
                                   bidirectional=bidirectional).to(device)
    policy_network.load_state_dict(torch.load(network_paths["policy_network"], map_location=device), strict=False)
    value_network = ValueNetwork(train_data["word_to_idx"], pretrained_embeddings=train_data["embeddings"],
                                 bidirectional=bidirectional, feature_dim=feature_dim).to(device)
    value_network.load_state_dict(torch.load(network_paths["value_network"], map_location=device), strict=False)
    a2c_network = AdvantageActorCriticNetwork(value_network, policy_network).to(device)
    a2c_network.load_state_dict(torch.load(model_path, map_location=device), strict=False)
    a2c_network.policy_network.train(mode=False)
    a2c_network.value_network.train(mode=False)
    return a2c_network


def get_filename(base_name, bidirectional, curriculum=None):
    """utilities.py:326-338."""
    name, ext = os.path.splitext(base_name)
    if bidirectional:
        name += "_bidirectional"
    if curriculum is not None:
        if curriculum:
            name += "_curriculum"
    name += ext
    return name
