"""Data-parallel exchange for the A2C step: one process per GPU, the flat gradient buffer averaged
once per step.

Each rank runs the whole rollout / loss / backward on its own rows (the value and reward
recurrences run over the rank's local rows only), then the flat float32 gradient buffer of
policy + value network is summed across ranks and divided by the world size (RCCL over xGMI
with backend "nccl"; gloo on CPU for tests): as one all-reduce (average_gradients), or as two
(GradientAverager) so that the policy network's part, complete milliseconds before the value
network's, crosses the links beside the value chain's backward pass.  Every rank then
applies the identical Adam update.  This equals `world` independent reference minibatches
with averaged gradients, not one reference minibatch of world*N rows, because the
reference's value / reward recurrences chain the rows of a minibatch together
(models.py:133, 226).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun's env).
    Returns (rank, world, local_rank); a no-op (0, 1, 0) when WORLD_SIZE is unset or 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1 and not _force_collectives():
        return 0, 1, 0
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def _force_collectives():
    """ICRL_FORCE_COLLECTIVES=1 makes a single-rank launch (under torch.distributed.run) initialise
    the process group and issue every collective anyway: the N > 1 code path on one GPU."""
    return os.environ.get("ICRL_FORCE_COLLECTIVES", "0") == "1" and "MASTER_ADDR" in os.environ


def _active():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _force_collectives())


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _through_host(t):
    """Whether a collective on tensor t is staged through host memory by this module: device tensors under a
    backend other than RCCL (gloo: rehearsals of the N > 1 path with several ranks on one GPU).  The staging is
    explicit - a synchronous copy to the host, the collective on the host tensor, a copy back - rather than
    left to the backend's own handling of device tensors."""
    return t.is_cuda and dist.get_backend() != "nccl"


def _sum_over_ranks(t):
    """In-place sum over ranks of t (any device)."""
    if _through_host(t):
        h = t.detach().cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


def average_gradients(flat_grad):
    """In-place mean over ranks of the flat gradient buffer (sum, then divide)."""
    if _active():
        _sum_over_ranks(flat_grad)
        flat_grad.div_(world_size())
    return flat_grad


class GradientAverager:
    """In-place mean over ranks of a flat gradient buffer in two pieces: [lo, hi) as soon as the
    caller knows it is complete (start_early), the rest in finish().  Every rank must make the same
    calls in the same order, once per step each."""

    def __init__(self, flat_grad, early_span, stream=None):
        """stream: the CUDA stream the early collective is queued on (e.g. A2CEngine.side_stream(), where the early
        range is completed); None: a stream of this object's own.  HIP multiplexes a process's streams onto four
        hardware queues, so every further stream may end up sharing a queue with one that should run beside it."""
        self.flat_grad = flat_grad
        self.lo, self.hi = early_span
        assert 0 <= self.lo < self.hi <= flat_grad.numel()
        self._work, self._stream = None, stream

    def start_early(self, ready_event=None):
        """Start averaging flat_grad[lo:hi].  ready_event (CUDA): the event after which that range is
        complete; the collective waits for it on a stream of its own."""
        if not _active() or self._work is not None:
            return
        piece = self.flat_grad[self.lo:self.hi]
        if _through_host(piece):
            # host-staged: nothing can start before the range is complete; finish() waits for that and reduces
            # this piece first, so every rank still issues the same collectives in the same order
            self._work = ("host", ready_event)
        elif piece.is_cuda:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=piece.device)
            if ready_event is not None:
                self._stream.wait_event(ready_event)
            else:
                self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                self._work = dist.all_reduce(piece, op=dist.ReduceOp.SUM, async_op=True)
        else:
            self._work = dist.all_reduce(piece, op=dist.ReduceOp.SUM, async_op=True)

    def finish(self):
        """Average what start_early did not cover (everything, if it was not called), wait for the
        early piece, divide by the world size.  The buffer is ready on the current stream afterwards."""
        if not _active():
            return self.flat_grad
        if self._work is None:
            _sum_over_ranks(self.flat_grad)
        else:
            staged = isinstance(self._work, tuple)
            if staged:
                if self._work[1] is not None:
                    self._work[1].synchronize()
                _sum_over_ranks(self.flat_grad[self.lo:self.hi])
            for lo, hi in ((0, self.lo), (self.hi, self.flat_grad.numel())):
                if hi > lo:
                    _sum_over_ranks(self.flat_grad[lo:hi])
            if not staged:
                self._work.wait()   # CUDA: the current stream waits for the collective's stream
            self._work = None
        self.flat_grad.div_(world_size())
        return self.flat_grad


def broadcast_parameters(flat, src=0):
    """Make every rank start from rank `src`'s parameters."""
    if _active():
        if _through_host(flat):
            h = flat.detach().cpu()
            dist.broadcast(h, src=src)
            flat.copy_(h)
        else:
            dist.broadcast(flat, src=src)
    return flat


def max_over_ranks(value, device):
    """Max of a python float over ranks (bench timing)."""
    if not _active():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def same_on_all_ranks(flat):
    """True when every rank holds bit-identical contents of the float32 buffer `flat`: three 64-bit integer
    checksums of its bit patterns (sum, sum of squares of the low halves, position-weighted sum), MIN and
    MAX over ranks equal."""
    if not _active():
        return True
    bits = flat.detach().contiguous().view(torch.int32).to(torch.int64)
    weights = torch.arange(bits.numel(), dtype=torch.int64, device=bits.device) % 65521 + 1
    low = bits & 0xFFFF
    sums = torch.stack([bits.sum(), (low * low).sum(), (bits * weights).sum()])
    lo, hi = sums.clone(), sums.clone()
    if dist.get_backend() != "nccl":
        lo, hi = lo.cpu(), hi.cpu()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return bool(torch.equal(lo, hi))


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def is_main():
    """True on the rank that writes checkpoints, result files and logged scalars."""
    return rank() == 0


def agree_min(value):
    """The smallest `value` (an int) any rank holds: a decision every rank must take alike - skipping a
    minibatch, say - is taken on this, so that no rank leaves the others waiting in a collective."""
    if not _active():
        return int(value)
    t = torch.tensor([int(value)], dtype=torch.int64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


def same_permutation(perm):
    """Rank 0's copy of an index permutation on every rank (the ranks draw it from their own torch
    generators, which agree only if they were seeded alike)."""
    if not _active():
        return perm
    t = perm.clone()
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.broadcast(t, src=0)
    return t.cpu()


def decorrelate_numpy_stream():
    """Give every rank its own NumPy global stream (the sampling uniforms are drawn from it): one draw from
    the stream as it stands, then a reseed that depends on the rank.  A no-op for a single process, so a
    single-process run consumes the stream exactly as the reference does."""
    if world_size() > 1:
        import numpy as np
        base = int(np.random.randint(0, 2 ** 31 - 1))
        np.random.seed((base + 7919 * rank()) % (2 ** 32))


def shard_rows(total_rows, rank=None, world=None):
    """Contiguous, equal row shard [lo, hi) of a global minibatch for this rank."""
    world = world_size() if world is None else world
    rank = (dist.get_rank() if world > 1 else 0) if rank is None else rank
    assert total_rows % world == 0, "global batch must divide evenly so that mean-of-means is the global mean"
    per = total_rows // world
    return rank * per, (rank + 1) * per
