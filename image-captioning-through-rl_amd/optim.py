"""Flat parameter / gradient storage and the fused Adam step.

``flatten_module`` re-homes every trainable parameter of a module (and its ``.grad``)
as a view into ONE float32 buffer each, so that the optimizer is a single kernel launch
and the data-parallel gradient exchange is a single RCCL all-reduce.
``FusedAdam`` is torch.optim.Adam's default configuration (trainers.py:378: lr 1e-4,
betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad) on that flat buffer.
"""
import torch

from . import hipabi as hip


class FlatParams:
    def __init__(self, params, flat, flat_grad, names, offsets):
        self.params, self.flat, self.flat_grad, self.names, self.offsets = params, flat, flat_grad, names, offsets

    def span(self, prefix):
        """[lo, hi) of the flat buffers covering the parameters whose names start with `prefix`
        (they are neighbours: named_parameters() walks one submodule after the other)."""
        idx = [i for i, name in enumerate(self.names) if name.startswith(prefix)]
        assert idx and idx == list(range(idx[0], idx[-1] + 1)), "no such parameters, or not contiguous: %r" % prefix
        hi = self.offsets[idx[-1] + 1] if idx[-1] + 1 < len(self.offsets) else self.flat.numel()
        return self.offsets[idx[0]], hi


def flatten_module(module):
    """Move the trainable parameters of `module` into one contiguous buffer (values are
    preserved; ``state_dict`` keys and shapes do not change).  Offsets are padded to 64
    floats so that every view stays 256-byte aligned for 16-byte kernel loads."""
    named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
    assert named, "no trainable parameters"
    dev = named[0][1].device
    offsets, total = [], 0
    for _, p in named:
        offsets.append(total)
        total += (p.numel() + 63) // 64 * 64
    flat = torch.zeros(total, dtype=torch.float32, device=dev)
    flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
    for (_, p), off in zip(named, offsets):
        view = flat[off:off + p.numel()].view(p.shape)
        view.copy_(p.data)
        p.data = view
        p.grad = flat_grad[off:off + p.numel()].view(p.shape)
    return FlatParams([p for _, p in named], flat, flat_grad, [n for n, _ in named], offsets)


class FusedAdam(torch.optim.Optimizer):
    """Adam over a FlatParams buffer: one HIP kernel per step."""

    def __init__(self, flat_params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(flat_params.params, dict(lr=lr, betas=betas, eps=eps))
        self.fp = flat_params
        self.exp_avg = torch.zeros_like(flat_params.flat)
        self.exp_avg_sq = torch.zeros_like(flat_params.flat)
        self.step_count = 0

    def zero_grad(self, set_to_none=False):
        # the fused backward overwrites every gradient, and the views must stay attached
        # to the flat buffer, so never drop them
        self.fp.flat_grad.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        group = self.param_groups[0]
        self.step_count += 1
        hip.adam_step(self.fp.flat, self.fp.flat_grad, self.exp_avg, self.exp_avg_sq, group["lr"],
                      group["betas"][0], group["betas"][1], group["eps"], self.step_count)
