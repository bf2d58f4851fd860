"""The entry point that allocates nothing (icrl_lstm_chain_fwd_seg_static: the segment workspace is a static
device array inside the library) against icrl_lstm_chain_fwd_seg on a caller-owned workspace: same kernels,
so the results must be bit-equal.  Two calls in a row, the second on a smaller input (the shared array
still holds the first call's contents)."""
import importlib

import numpy as np
import pytest
import torch

from helpers import pkg  # noqa: F401  (puts the package on the path)

hip = importlib.import_module("image-captioning-through-rl_amd.hipabi")
pytestmark = pytest.mark.gpu
H = 512


def _case(seed, n, p0, calls, segs, warm, with_state):
    rs = np.random.RandomState(seed)
    cols = p0 + calls - 1
    positions = n * sum(p0 + s for s in range(calls))
    xproj = torch.from_numpy((0.5 * rs.standard_normal((cols * n, 4 * H))).astype(np.float32)).cuda()
    w_hh = torch.from_numpy((rs.uniform(-1, 1, size=(4 * H, H)) / np.sqrt(H)).astype(np.float32)).cuda()
    state = torch.from_numpy(rs.uniform(-0.5, 0.5, size=2 * H).astype(np.float32)).cuda() if with_state else None
    out = []
    for static in (False, True):
        h_emit = torch.full((calls, n, H), 7.0, device="cuda")
        saved = torch.full((positions, 6, H), 7.0, device="cuda")
        err = torch.full((2,), 7.0, device="cuda")
        if static:
            hip.lstm_chain_fwd_seg_static(xproj, w_hh, n, p0, calls, segs, warm, h_emit, saved, err, state_in=state)
        else:
            work = torch.zeros(hip.chain_seg_workspace_floats(segs), device="cuda")
            hip.lstm_chain_fwd_seg(xproj, w_hh, n, p0, calls, segs, warm, h_emit, saved, work, err, state_in=state)
        torch.cuda.synchronize()
        out.append((h_emit, saved, err[0].item()))
    (h_a, s_a, e_a), (h_b, s_b, e_b) = out
    assert torch.equal(h_a, h_b) and torch.equal(s_a, s_b) and e_a == e_b
    assert np.isfinite(e_a) and e_a < 1e-4, "seam check of the reference call itself"
    assert not (s_b == 7.0).any(), "every position's activations are written"


def test_static_workspace_entry_point_equals_the_caller_workspace_one():
    _case(1, 64, 1, 12, 128, 48, False)     # the largest segment count the static array holds
    _case(2, 24, 2, 5, 20, 32, True)        # smaller, right behind it: the array still holds the first call's state
    _case(3, 64, 1, 12, 96, 64, True)


def test_static_workspace_entry_point_allocates_nothing():
    """Free device memory is the same before and after a call (a first call has already looked the array up)."""
    rs = np.random.RandomState(5)
    n, p0, calls, segs, warm = 64, 1, 12, 128, 48
    positions = n * sum(p0 + s for s in range(calls))
    xproj = torch.from_numpy((0.5 * rs.standard_normal(((p0 + calls - 1) * n, 4 * H))).astype(np.float32)).cuda()
    w_hh = torch.from_numpy((rs.uniform(-1, 1, size=(4 * H, H)) / np.sqrt(H)).astype(np.float32)).cuda()
    h_emit, saved = torch.zeros(calls, n, H, device="cuda"), torch.zeros(positions, 6, H, device="cuda")
    err = torch.zeros(2, device="cuda")
    hip.lstm_chain_fwd_seg_static(xproj, w_hh, n, p0, calls, segs, warm, h_emit, saved, err)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    hip.lstm_chain_fwd_seg_static(xproj, w_hh, n, p0, calls, segs, warm, h_emit, saved, err)
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before


def test_static_workspace_entry_point_rejects_more_segments_than_it_holds():
    x = torch.zeros(64 * 19, 4 * H, device="cuda")
    w = torch.zeros(4 * H, H, device="cuda")
    h, s, e = torch.zeros(19, 64, H, device="cuda"), torch.zeros(64 * 190, 6, H, device="cuda"), torch.zeros(2, device="cuda")
    with pytest.raises(RuntimeError, match="exceed the static workspace"):
        hip.lstm_chain_fwd_seg_static(x, w, 64, 1, 19, 129, 32, h, s, e)
