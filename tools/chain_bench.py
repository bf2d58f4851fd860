"""Time the segment-parallel chain kernels alone and side by side (us per step launch).

    python tools/chain_bench.py [segments] [warmup]
"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = importlib.import_module("image-captioning-through-rl_amd.hipabi")

H = 512
n, p0, s = 512, 1, 19
K = int(sys.argv[1]) if len(sys.argv) > 1 else 512
W = int(sys.argv[2]) if len(sys.argv) > 2 else 96
dev = "cuda"
tv = n * sum(p0 + k for k in range(s))
tr = n * sum(p0 + 1 + k for k in range(s))
g = torch.Generator(device="cpu").manual_seed(0)


def u(*shape, scale=1.0):
    return ((torch.rand(*shape, generator=g) * 2 - 1) * scale).to(dev)


w_v, w_r, b_r = u(4 * H, H, scale=H ** -0.5), u(3 * H, H, scale=H ** -0.5), u(3 * H, scale=H ** -0.5)
xp_v, xp_r = u((p0 + s - 1) * n, 4 * H), u((p0 + s) * n, 3 * H)
hv, hr = torch.empty(s, n, H, device=dev), torch.empty(s, n, H, device=dev)
saved, dgv = torch.empty(tv, 6, H, device=dev), torch.empty(tv, 4 * H, device=dev)
d_emit = u(s, n, H, scale=1e-3)
work_v = torch.empty(hip.chain_seg_workspace_floats(K), device=dev)
work_r = torch.empty(hip.chain_seg_workspace_floats(K), device=dev)
seam = torch.zeros(6, device=dev)
steps_v = W + (tv + K - 1) // K
steps_r = W + (tr + K - 1) // K
side = torch.cuda.Stream()


def fwd_v():
    hip.lstm_chain_fwd_seg(xp_v, w_v, n, p0, s, K, W, hv, saved, work_v, seam[0:2])


def fwd_r():
    hip.gru_chain_fwd_seg(xp_r, w_r, b_r, n, p0 + 1, s, K, W, hr, work_r, seam[2:4])


def bwd_v():
    hip.lstm_chain_bwd_seg(w_v, saved, d_emit, n, p0, s, K, W, dgv, work_v, seam[4:6])


def both():
    ev = torch.cuda.Event()
    ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        fwd_r()
        done = torch.cuda.Event()
        done.record(side)
    fwd_v()
    torch.cuda.current_stream().wait_event(done)


hv2, hr2, saved2 = torch.empty_like(hv), torch.empty_like(hr), torch.empty_like(saved)
seam2 = torch.zeros(6, device=dev)


def pair():
    hip.chain_pair_fwd_seg(xp_v, w_v, xp_r, w_r, b_r, n, p0, p0 + 1, s, K, W, hv2, saved2, hr2, work_v, work_r,
                           seam2[0:1], seam2[2:3])


def timeit(name, fn, steps, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%-28s %7.3f ms  %6.2f us/step (%d steps)" % (name, ms, 1e3 * ms / steps, steps), flush=True)


print("segments %d warmup %d" % (K, W))
timeit("lstm fwd alone", fwd_v, steps_v)
timeit("gru fwd alone", fwd_r, steps_r)
timeit("lstm+gru fwd, two streams", both, max(steps_v, steps_r))
if hip.chain_pair_supported(K):
    timeit("lstm+gru fwd, persistent pair", pair, max(steps_v, steps_r))
    fwd_v(); fwd_r(); torch.cuda.synchronize()
    print("pair == per-step launches:", torch.equal(hv, hv2), torch.equal(hr, hr2), torch.equal(saved, saved2),
          seam2.tolist())
    d = (saved - saved2).abs()
    bad = (d > 0).nonzero()
    print("saved mismatches", bad.shape[0], "max", float(d.max()), "first", bad[:5].tolist(), "last", bad[-3:].tolist())
    d = (hv - hv2).abs()
    bad = (d > 0).nonzero()
    print("hv mismatches", bad.shape[0], "max", float(d.max()), "first", bad[:5].tolist())
timeit("lstm bwd alone", bwd_v, steps_v + 1)
print("seam", seam.tolist())
