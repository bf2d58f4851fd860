"""Time the persistent chain launches alone: float32 against the f16 matrix-instruction launches (us per step).

    python tools/chain_bench_f16.py [segments] [warmup] [full_calls]
"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = importlib.import_module("image-captioning-through-rl_amd.hipabi")

H = 512
n, p0, s = 512, 1, 19
K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
W = int(sys.argv[2]) if len(sys.argv) > 2 else 96
FC = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = "cuda"
tv = hip.chain_positions(n, p0, s, FC)
tr = hip.chain_positions(n, p0 + 1, s, FC)
g = torch.Generator(device="cpu").manual_seed(0)


def u(*shape, scale=1.0):
    return ((torch.rand(*shape, generator=g) * 2 - 1) * scale).to(dev)


w_v, w_r, b_r = u(4 * H, H, scale=H ** -0.5), u(3 * H, H, scale=H ** -0.5), u(3 * H, scale=H ** -0.5)
xp_v, xp_r = u((p0 + s - 1) * n, 4 * H), u((p0 + s) * n, 3 * H)
hv, hr = torch.empty(s, n, H, device=dev), torch.empty(s, n, H, device=dev)
saved, dgv = torch.empty(tv, 6, H, device=dev), torch.empty(tv, 4 * H, device=dev)
saved_r = torch.empty(tr, 5, H, device=dev)
d_emit = u(s, n, H, scale=1e-3)
work_v = torch.empty(hip.chain_seg_workspace_floats(K), device=dev)
work_r = torch.empty(hip.chain_seg_workspace_floats(K), device=dev)
work_b = torch.empty(hip.chain_bwd_f16x2_workspace_floats(K), device=dev)
seam = torch.zeros(6, device=dev)
steps_v = W + (tv + K - 1) // K
steps_r = W + (tr + K - 1) // K


def timeit(name, fn, steps, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%-44s %7.3f ms  %6.2f us/step (%d steps)" % (name, ms, 1e3 * ms / steps, steps), flush=True)


def pair(arith):
    hip.chain_pair_fwd_seg(xp_v, w_v, xp_r, w_r, b_r, n, p0, p0 + 1, s, K, W, hv, saved, hr, work_v, work_r,
                           seam[0:1], seam[2:3], full_calls=FC, saved_r=saved_r, arith=arith)


print("segments %d warmup %d full_calls %d (%d / %d positions)" % (K, W, FC, tv, tr))
timeit("forward pair, float32", lambda: pair("f32"), max(steps_v, steps_r))
timeit("forward pair, f16 pieces", lambda: pair("f16x2"), max(steps_v, steps_r))
timeit("backward, float32 (gathers dgates)", lambda: hip.lstm_chain_bwd_seg(
    w_v, saved, d_emit, n, p0, s, K, W, dgv, work_v, seam[4:6], persistent=True, full_calls=FC), steps_v + 1)
timeit("backward, f16 pieces (owner computes)", lambda: hip.lstm_chain_bwd_seg(
    w_v, saved, d_emit, n, p0, s, K, W, dgv, work_b, seam[4:6], full_calls=FC, arith="f16x2"), steps_v + 1)
# the policy LSTM's BPTT: 512 rows, 19 steps
t_steps = 19
gates, c_all = u(t_steps, n, 4 * H).sigmoid(), u(t_steps + 1, n, H)
d_h, dgp, dh0 = u(t_steps, n, H, scale=1e-3), torch.empty(t_steps, n, 4 * H, device=dev), torch.empty(n, H, device=dev)
work_p = torch.empty(max(hip.chain_seg_workspace_floats(n), hip.chain_bwd_f16x2_workspace_floats(n)), device=dev)
status = torch.zeros(1, dtype=torch.int32, device=dev)
for arith in ("f32", "f16x2"):
    timeit("policy BPTT, one launch, %s" % arith, lambda: hip.policy_lstm_bptt_persist(
        gates, c_all, d_h, w_v, dgp, dh0, work_p, status, arith=arith), t_steps + 1)
print("seam", seam.tolist(), "status", int(status[0]))
