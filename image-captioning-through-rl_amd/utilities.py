"""The parts of the reference's utilities.py that the A2C training path and the look-ahead test
loop call: minibatch iteration, caption decoding, checkpoint save / load, logging helpers.  Dataset readers, word
embedding tooling, metrics and post-processing are not part of this path."""
import numpy as np
import torch
from torch import randperm, save

from .models import device


def print_green(text):
    print('\033[32m', text, '\033[0m', sep='')


def print_red(text):
    print('\033[31m', text, '\033[0m', sep='')


def _h5_arrays(path, only=None):
    """Datasets of an HDF5 file as NumPy arrays ({name: array}; `only`: that one dataset).  h5py is imported
    here, on first use: nothing else in this package needs it."""
    try:
        import h5py
    except ImportError as exc:   # pragma: no cover - depends on the installation
        raise ImportError("load_data reads the COCO captioning set from HDF5 files and needs h5py (%s)" % exc) from exc
    with h5py.File(path, "r") as f:
        if only is not None:
            return np.asarray(f[only])
        return {name: np.asarray(ds) for name, ds in f.items()}


def _caption_lengths(captions, end_token=2):
    """Tokens up to and including the first <END> (id 2) of every caption row, as float64 (utilities.py:98-103)."""
    is_end = np.asarray(captions) == end_token
    if captions.shape[0] and not is_end.any(axis=1).all():
        raise ValueError("caption row %d has no <END> token" % int(np.flatnonzero(~is_end.any(axis=1))[0]))
    return (is_end.argmax(axis=1) + 1).astype(np.float64)


def load_data(base_dir, max_train=None, pca_features=True, print_keys=False):
    """The COCO captioning set of the reference (utilities.py:45-113) as one dict: every dataset of
    coco2014_captions.h5, train/val fc7 features (PCA-reduced or full), the vocabulary JSON's entries, the URL
    lists, and `{train,val}_captions_lens`.  max_train draws that many training captions WITH replacement from
    NumPy's global stream (one np.random.randint call, as the reference)."""
    import json
    import os
    data = dict(_h5_arrays(os.path.join(base_dir, "coco2014_captions.h5")))
    suffix = "_pca.h5" if pca_features else ".h5"
    for split in ("train", "val"):
        data["%s_features" % split] = _h5_arrays(os.path.join(base_dir, "%s2014_vgg16_fc7%s" % (split, suffix)), "features")
    with open(os.path.join(base_dir, "coco2014_vocab.json")) as f:
        data.update(json.load(f))
    for split in ("train", "val"):
        with open(os.path.join(base_dir, "%s2014_urls.txt" % split)) as f:
            data["%s_urls" % split] = np.asarray([line.strip() for line in f])
    if max_train is not None:
        pick = np.random.randint(data["train_captions"].shape[0], size=max_train)
        data["train_captions"] = data["train_captions"][pick]
        data["train_image_idxs"] = data["train_image_idxs"][pick]
    for split in ("train", "val"):
        data["%s_captions_lens" % split] = _caption_lengths(data["%s_captions" % split])
    if print_keys:
        for key, value in data.items():
            if isinstance(value, np.ndarray):
                print(key, type(value), value.shape, value.dtype)
            else:
                print(key, type(value), len(value))
    return data


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
    'feature_store' (built once by the caller: data['feature_store'] = DeviceFeatureStore(data)).
    With more than one rank every rank gets its own `batch_size` rows of each global minibatch of
    world * batch_size rows (sharded_minibatches)."""
    from . import dist as ddp
    if ddp.world_size() > 1:
        return sharded_minibatches(data, batch_size, split)
    store = data.get('feature_store') if hasattr(data, 'get') else None
    if store is not None and store.split == split:
        return store.minibatches(batch_size)
    return get_coco_minibatches(data, batch_size=batch_size, split=split)


def sharded_minibatches(data, batch_size=100, split='train'):
    """One shuffled pass over the split for data-parallel training: the permutation is rank 0's (one
    torch.randperm draw on every rank, as in get_coco_minibatches), it is cut into global minibatches of
    world * batch_size rows, and rank r takes rows [r * b, (r + 1) * b) of each, b = batch_size (the last,
    shorter global minibatch is split evenly and its remainder dropped, so that every rank runs the same
    number of equally sized minibatches: the mean of the ranks' means is then the global mean, and no rank
    waits for another in a collective).  Yields (captions, features, urls) like get_coco_minibatches."""
    from . import dist as ddp
    world, rank = ddp.world_size(), ddp.rank()
    captions_all = data['%s_captions' % split]
    total = captions_all.shape[0]
    permutation = ddp.same_permutation(randperm(total))
    for i in range(0, total, world * batch_size):
        chunk = permutation[i: i + world * batch_size]
        per = min(batch_size, len(chunk) // world)
        if per == 0:
            return
        mask = chunk[rank * per: (rank + 1) * per].numpy()   # an index array keeps a one-row shard two-dimensional
        image_idxs = data['%s_image_idxs' % split][mask]
        urls = data.get('%s_urls' % split) if hasattr(data, 'get') else data['%s_urls' % split]
        yield captions_all[mask], data['%s_features' % split][image_idxs], None if urls is None else urls[image_idxs]


def global_minibatch_number(epoch, batch_id, batch_size):
    """x-axis of the logged scalars (utilities.py:204-212; epoch * batch_size, as there)."""
    return epoch * batch_size + batch_id


def save_a2c_model(model, save_paths):
    """The A2C network's state_dict to one path or to each path of a list (utilities.py:286-296); with
    several ranks only the first one writes."""
    from . import dist as ddp
    if not ddp.is_main():
        return
    for path in ([save_paths] if isinstance(save_paths, str) else list(save_paths or [])):
        save(model.state_dict(), path)
