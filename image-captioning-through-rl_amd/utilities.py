"""The parts of the reference's utilities.py that the A2C training path calls:
minibatch iteration, checkpoint save / load, logging helpers.  Dataset readers, word
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
    """The feature matrix resident in HBM, so a minibatch's features are a coalesced
    device row gather instead of a host gather + PCIe copy per step."""

    def __init__(self, data, split='train', dev=None):
        dev = device if dev is None else dev
        self.features = torch.as_tensor(data['%s_features' % split]).float().to(dev).contiguous()
        self.image_idxs = torch.as_tensor(np.asarray(data['%s_image_idxs' % split])).long().to(dev)
        self.captions = torch.as_tensor(np.asarray(data['%s_captions' % split])).long().to(dev)

    def minibatches(self, batch_size):
        """Same order as get_coco_minibatches under the same torch seed; yields device
        tensors (captions [n,L] int64, features [n,F] float32)."""
        from . import hipabi as hip
        total = self.captions.shape[0]
        permutation = randperm(total).to(self.captions.device)
        for i in range(0, total, batch_size):
            mask = permutation[i: i + batch_size]
            caps = self.captions[mask]
            idx = self.image_idxs[mask].contiguous()
            feats = torch.empty(idx.numel(), self.features.shape[1], device=self.features.device)
            hip.gather_rows(self.features, idx, 1, idx.numel(), feats)
            yield caps, feats


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
