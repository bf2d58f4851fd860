"""Each HIP entry point, called through the C ABI, against its CPU reference.
Needs an MI355X: run with -m gpu."""
import numpy as np
import pytest
import torch

from helpers import pkg  # noqa: F401  (puts the package on the path)
import importlib

hip = importlib.import_module("image-captioning-through-rl_amd.hipabi")
from oracle import kernel_refs as ref

pytestmark = pytest.mark.gpu
H = 512


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.asarray(a)).to(dtype).cuda()


def rnd(rs, *shape, scale=1.0):
    return (rs.standard_normal(shape) * scale).astype(np.float32)


# --------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("a_km", [False, True])
@pytest.mark.parametrize("b_km", [False, True])
@pytest.mark.parametrize("shape", [(70, 130, 77), (512, 2048, 1024), (256, 1000, 512), (33, 64, 5)])
def test_gemm_layouts(a_km, b_km, shape):
    m, n, k = shape
    rs = np.random.RandomState(m + n + k)
    a, b = rnd(rs, m, k), rnd(rs, k, n)
    bias, bias2 = rnd(rs, n), rnd(rs, n)
    c0 = rnd(rs, m, n)
    a_st = dev(a.T.copy() if a_km else a)
    b_st = dev(b if b_km else b.T.copy())
    out = dev(c0)
    hip.gemm(a_st, b_st, out, a_kmajor=a_km, b_kmajor=b_km, bias=dev(bias), bias2=dev(bias2), accumulate=True)
    want = c0.astype(np.float64) + a.astype(np.float64) @ b.astype(np.float64) + bias + bias2
    # exact-f32 MFMA = a k-ordered fmaf chain: error ~ 1e-7 * sum|a*b|
    tol = 4e-7 * (np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)) + 1e-6
    assert np.all(np.abs(out.cpu().numpy() - want) <= tol)


def test_gemm_gather_strided_and_plain_store():
    rs = np.random.RandomState(5)
    table, w = rnd(rs, 300, 512), rnd(rs, 2048, 512 + 64)
    idx = rs.randint(0, 300, size=100).astype(np.int64)
    out = torch.full((100, 2048), 7.0, device="cuda")
    w_dev = dev(w)
    hip.gemm(dev(table), w_dev[:, 64:], out, a_rows=dev(idx, torch.int64))  # column slice: ldb = 576
    want = table[idx].astype(np.float64) @ w[:, 64:].astype(np.float64).T
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-5, atol=2e-4)


def test_gemm_rejects_bad_arguments():
    a = torch.zeros(4, 8, device="cuda")
    with pytest.raises(RuntimeError, match="leading dimension"):
        hip.gemm(a, a, torch.zeros(4, 4, device="cuda"), lda=2)


# --------------------------------------------------------------------------- pointwise
def test_gather_scatter_colsum():
    rs = np.random.RandomState(1)
    table = rnd(rs, 50, 512)
    toks = rs.randint(0, 50, size=(16, 7)).astype(np.int64)
    toks_d = dev(toks, torch.int64)
    out = torch.empty(16, 512, device="cuda")
    hip.gather_rows(dev(table), toks_d[:, 3:], 7, 16, out)
    assert np.array_equal(out.cpu().numpy(), table[toks[:, 3]])
    grad = torch.zeros(50, 512, device="cuda")
    rows = rnd(rs, 16, 512)
    hip.scatter_add_rows(grad, toks_d[:, 3:], 7, 16, dev(rows))
    want = np.zeros((50, 512))
    np.add.at(want, toks[:, 3], rows)
    np.testing.assert_allclose(grad.cpu().numpy(), want, rtol=1e-6, atol=1e-6)
    x = rnd(rs, 1000, 130)
    cs = torch.ones(130, device="cuda")
    hip.colsum(dev(x), cs, accumulate=True)
    np.testing.assert_allclose(cs.cpu().numpy(), 1 + x.astype(np.float64).sum(0), rtol=1e-5, atol=1e-4)


def test_lstm_cell_forward_backward():
    rs = np.random.RandomState(2)
    n = 37
    pre, c_prev = rnd(rs, n, 4 * H, scale=2.0), rnd(rs, n, H)
    dh, dc = rnd(rs, n, H), rnd(rs, n, H)
    gates = dev(pre)
    c, h = torch.empty(n, H, device="cuda"), torch.empty(n, H, device="cuda")
    hip.lstm_cell_fwd(gates, dev(c_prev), c, h)
    act_w, c_w, h_w = ref.lstm_cell(pre, c_prev)
    np.testing.assert_allclose(gates.cpu().numpy(), act_w.numpy(), rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(c.cpu().numpy(), c_w.numpy(), rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(h.cpu().numpy(), h_w.numpy(), rtol=2e-6, atol=2e-6)
    dgates, dc_prev = torch.empty(n, 4 * H, device="cuda"), torch.empty(n, H, device="cuda")
    hip.lstm_cell_bwd(gates, dev(c_prev), c, dev(dh), dev(dc), dgates, dc_prev)
    dg_w, dcp_w = ref.lstm_cell_grads(pre, c_prev, dh, dc)
    np.testing.assert_allclose(dgates.cpu().numpy(), dg_w, rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(dc_prev.cpu().numpy(), dcp_w, rtol=1e-5, atol=2e-6)


# --------------------------------------------------------------------------- sampler
@pytest.mark.parametrize("v", [100, 257, 10000])
def test_sampler_matches_numpy_choice(v):
    rs = np.random.RandomState(v)
    n = 64
    logits = rnd(rs, n, v, scale=2.0)
    u = rs.random_sample(n)
    u[0], u[1] = 0.0, 1.0 - 2 ** -53  # extreme draws
    toks = torch.zeros(n, 5, dtype=torch.int64, device="cuda")
    logp = torch.zeros(n, device="cuda")
    probs = torch.zeros(n, v, device="cuda")
    ent = torch.zeros(n, device="cuda")
    hip.sample_tokens(dev(logits), hip.SAMPLE, toks[:, 2:], 5, uniforms=dev(u, torch.float64), log_prob=logp,
                      probs=probs, entropy=ent)
    p = probs.cpu().numpy()
    p_want = torch.softmax(torch.tensor(logits, dtype=torch.float64), 1).numpy()
    np.testing.assert_allclose(p, p_want, rtol=3e-6, atol=1e-12)
    got = toks[:, 2].cpu().numpy()
    # bit-exact on the kernel's own float32 probabilities: numpy's algorithm, same draw
    want = np.array([ref.sample_row(p[i], u[i]) for i in range(n)])
    fragile = np.array([ref.orc.cdf_edge_distance(p[i], u[i]) < 1e-13 for i in range(n)])
    assert np.all((got == want) | fragile)
    assert np.all(toks[:, [0, 1, 3, 4]].cpu().numpy() == 0)
    np.testing.assert_allclose(logp.cpu().numpy(), np.log(p_want[np.arange(n), got]), rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(ent.cpu().numpy(), -(p_want * np.log(p_want)).sum(1), rtol=1e-4)


def test_sampler_greedy_first_index_wins_and_forced():
    rs = np.random.RandomState(3)
    n, v = 32, 1000
    logits = rnd(rs, n, v)
    logits[0, [17, 900]] = 50.0  # exact tie: the lower id must win
    logits[1, v - 1] = 60.0
    toks = torch.zeros(n, dtype=torch.int64, device="cuda")
    hip.sample_tokens(dev(logits), hip.GREEDY, toks, 1)
    assert np.array_equal(toks.cpu().numpy(), torch.tensor(logits).argmax(1).numpy())
    assert toks[0].item() == 17 and toks[1].item() == v - 1
    forced = dev(rs.randint(0, v, size=n), torch.int64)
    logp = torch.zeros(n, device="cuda")
    hip.sample_tokens(dev(logits), hip.FORCED, forced, 1, log_prob=logp)
    want = torch.log_softmax(torch.tensor(logits, dtype=torch.float64), 1).numpy()[np.arange(n), forced.cpu().numpy()]
    np.testing.assert_allclose(logp.cpu().numpy(), want, rtol=2e-5, atol=2e-6)


def test_policy_dlogits_loss_value_head_cosine_adam():
    rs = np.random.RandomState(4)
    rows, v, n = 24, 300, 8
    p = torch.softmax(torch.tensor(rnd(rs, rows, v)), 1).numpy()
    toks = rs.randint(0, v, size=rows).astype(np.int64)
    coef = rnd(rs, rows)
    g = dev(p)
    hip.policy_dlogits(g, g, dev(toks, torch.int64), 1, dev(coef))
    hot = np.zeros((rows, v)); hot[np.arange(rows), toks] = 1
    np.testing.assert_allclose(g.cpu().numpy(), coef[:, None] * (hot - p), rtol=1e-6, atol=1e-8)

    values, rewards, logp = rnd(rs, rows), rnd(rs, rows), -np.abs(rnd(rs, rows))
    scal = torch.zeros(8, device="cuda")
    dv, dl = torch.zeros(rows, device="cuda"), torch.zeros(rows, device="cuda")
    hip.a2c_loss(dev(values), dev(rewards), dev(logp), scal, dv, dl)
    adv = values.astype(np.float64) - rewards
    actor, critic = (-logp * adv).mean(), 0.5 * (adv ** 2).mean()
    np.testing.assert_allclose(scal[:5].cpu().numpy(), [actor + critic, actor, critic, rewards.mean(), adv.mean()],
                               rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dv.cpu().numpy(), (adv - logp) / rows, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(dl.cpu().numpy(), -adv / rows, rtol=1e-5, atol=1e-7)

    yh, u, w2, b2 = rnd(rs, rows, 512), rnd(rs, n, 512), rnd(rs, 512), rnd(rs, 1)
    y = dev(yh)
    vals = torch.zeros(rows, device="cuda")
    hip.value_head_fwd(y, dev(u), dev(w2), dev(b2), rows, n, vals)
    y_w = yh.astype(np.float64) + u[np.arange(rows) % n]
    np.testing.assert_allclose(y.cpu().numpy(), y_w, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(vals.cpu().numpy(), y_w @ w2 + b2, rtol=1e-5, atol=2e-5)

    ve, se = rnd(rs, n, 512), rnd(rs, rows, 512)
    se[3] = 0.0  # zero vector: F.normalize clamps the norm at 1e-12, cosine 0
    rew = torch.zeros(rows, device="cuda")
    hip.cosine_reward(dev(ve), dev(se), rows, n, rew)
    ven = ve / np.maximum(np.linalg.norm(ve, axis=1, keepdims=True), 1e-12)
    sen = se / np.maximum(np.linalg.norm(se, axis=1, keepdims=True), 1e-12)
    np.testing.assert_allclose(rew.cpu().numpy(), (ven[np.arange(rows) % n] * sen).sum(1), rtol=1e-5, atol=1e-6)

    cnt = 100003
    pp, gg, mm, vv = rnd(rs, cnt), rnd(rs, cnt, scale=0.01), rnd(rs, cnt, scale=0.01), np.abs(rnd(rs, cnt, scale=1e-4))
    pd, md, vd = dev(pp), dev(mm), dev(vv)
    hip.adam_step(pd, dev(gg), md, vd, 1e-4, 0.9, 0.999, 1e-8, 3)
    p_w, m_w, v_w = ref.adam(pp, gg, mm, vv, 1e-4, 0.9, 0.999, 1e-8, 3)
    np.testing.assert_allclose(pd.cpu().numpy(), p_w, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(md.cpu().numpy(), m_w, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(vd.cpu().numpy(), v_w, rtol=5e-6, atol=1e-12)


# --------------------------------------------------------------------------- chains
def _chain_inputs(rs, n, cols, gates):
    xproj = rnd(rs, cols * n, gates * H, scale=0.5)
    w_hh = (rs.uniform(-1, 1, size=(gates * H, H)) / np.sqrt(H)).astype(np.float32)
    return xproj, w_hh


def _ws():
    return (torch.zeros(hip.chain_sync_bytes(), dtype=torch.uint8, device="cuda"),
            torch.zeros(1, dtype=torch.int32, device="cuda"))


@pytest.mark.parametrize("n,p0,calls", [(5, 1, 4), (8, 3, 3), (1, 1, 2)])
def test_lstm_chain_forward_backward(n, p0, calls):
    rs = np.random.RandomState(10 * n + p0)
    cols = p0 + calls - 1
    xproj, w_hh = _chain_inputs(rs, n, cols, 4)
    positions = n * sum(p0 + s for s in range(calls))
    d_emit = rnd(rs, calls, n, H)
    h_w, saved_w, dg_w = ref.lstm_chain(xproj, w_hh, n, p0, calls, want_grad_of=d_emit)

    ws, status = _ws()
    h_emit = torch.zeros(calls, n, H, device="cuda")
    saved = torch.zeros(positions, 6, H, device="cuda")
    hip.lstm_chain_fwd(dev(xproj), dev(w_hh), n, p0, calls, h_emit, saved, ws, status)
    torch.cuda.synchronize()
    assert status.item() == 0, "chain kernel timed out waiting for a peer workgroup"
    np.testing.assert_allclose(saved.cpu().numpy(), saved_w, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(h_emit.cpu().numpy(), h_w, rtol=2e-5, atol=2e-6)

    dgates = torch.zeros(positions, 4 * H, device="cuda")
    hip.lstm_chain_bwd(dev(w_hh), saved, dev(d_emit), n, p0, calls, dgates, ws, status)
    torch.cuda.synchronize()
    assert status.item() == 0
    np.testing.assert_allclose(dgates.cpu().numpy(), dg_w, rtol=2e-4, atol=2e-6)

    # fold over calls + previous-h stream, as the weight-gradient GEMMs consume them
    folded = torch.zeros(cols * n, 4 * H, device="cuda")
    hip.chain_fold_calls(dgates, n, p0, calls, 4 * H, folded)
    want = np.zeros((cols * n, 4 * H))
    off = 0
    for s in range(calls):
        ln = (p0 + s) * n
        want[:ln] += dg_w[off:off + ln]
        off += ln
    np.testing.assert_allclose(folded.cpu().numpy(), want, rtol=2e-4, atol=4e-6)
    h_prev = torch.ones(positions, H, device="cuda")
    hip.chain_hprev(saved, positions, h_prev)
    want_hp = np.concatenate((np.zeros((1, H)), saved_w[:-1, 5]), 0)
    np.testing.assert_allclose(h_prev.cpu().numpy(), want_hp, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("n,p0,calls", [(5, 1, 4), (8, 3, 3)])
def test_gru_chain_forward(n, p0, calls):
    rs = np.random.RandomState(100 + n)
    cols = p0 + calls
    xproj, w_hh = _chain_inputs(rs, n, cols, 3)
    b_hh = rnd(rs, 3 * H, scale=0.1)
    h_w = ref.gru_chain(xproj, w_hh, b_hh, n, p0, calls)
    ws, status = _ws()
    h_emit = torch.zeros(calls, n, H, device="cuda")
    hip.gru_chain_fwd(dev(xproj), dev(w_hh), dev(b_hh), n, p0 + 1, calls, h_emit, ws, status)
    torch.cuda.synchronize()
    assert status.item() == 0
    np.testing.assert_allclose(h_emit.cpu().numpy(), h_w, rtol=2e-5, atol=2e-6)


def test_chain_long_run_is_stable_and_timed():
    """12k dependent steps: state stays finite and bounded; prints us/step for the log."""
    rs = np.random.RandomState(77)
    n, p0, calls = 64, 1, 19
    cols = p0 + calls - 1
    xproj, w_hh = _chain_inputs(rs, n, cols, 4)
    positions = n * sum(p0 + s for s in range(calls))
    ws, status = _ws()
    h_emit = torch.zeros(calls, n, H, device="cuda")
    saved = torch.zeros(positions, 6, H, device="cuda")
    xd, wd = dev(xproj), dev(w_hh)
    hip.lstm_chain_fwd(xd, wd, n, p0, calls, h_emit, saved, ws, status)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    hip.lstm_chain_fwd(xd, wd, n, p0, calls, h_emit, saved, ws, status)
    e1.record()
    torch.cuda.synchronize()
    assert status.item() == 0
    hs = saved[:, 5].cpu().numpy()
    assert np.isfinite(hs).all() and np.abs(hs).max() <= 1.0
    print("\nlstm_chain_fwd: %d steps, %.3f us/step" % (positions, 1e3 * e0.elapsed_time(e1) / positions))
    # same inputs, second run: bitwise identical (fixed summation order, no atomics)
    saved2 = saved.clone()
    hip.lstm_chain_fwd(xd, wd, n, p0, calls, h_emit, saved, ws, status)
    torch.cuda.synchronize()
    assert torch.equal(saved, saved2)


# --------------------------------------------------------------------------- segment-parallel chains
def _seg_ws(k):
    return (torch.zeros(hip.chain_seg_workspace_floats(k), device="cuda"), torch.zeros(2, device="cuda"))


@pytest.mark.parametrize("n,p0,calls,segs,warm", [(64, 1, 19, 32, 64), (16, 3, 6, 7, 40), (64, 1, 19, 512, 64)])
def test_segmented_chains_match_the_exact_kernels(n, p0, calls, segs, warm):
    rs = np.random.RandomState(segs)
    cols = p0 + calls
    positions = n * sum(p0 + s for s in range(calls))
    xproj, w_hh = _chain_inputs(rs, n, cols, 4)
    xd, wd = dev(xproj), dev(w_hh)
    d_emit = dev(rnd(rs, calls, n, H))
    ws, status = _ws()
    h_x, saved_x = torch.zeros(calls, n, H, device="cuda"), torch.zeros(positions, 6, H, device="cuda")
    dg_x = torch.zeros(positions, 4 * H, device="cuda")
    hip.lstm_chain_fwd(xd, wd, n, p0, calls, h_x, saved_x, ws, status)
    hip.lstm_chain_bwd(wd, saved_x, d_emit, n, p0, calls, dg_x, ws, status)
    work, err = _seg_ws(segs)
    h_s, saved_s = torch.zeros_like(h_x), torch.zeros_like(saved_x)
    dg_s = torch.zeros_like(dg_x)
    hip.lstm_chain_fwd_seg(xd, wd, n, p0, calls, segs, warm, h_s, saved_s, work, err)
    torch.cuda.synchronize()
    assert status.item() == 0
    assert err[0].item() < 2e-6, "seam check: warm-up state differs from the true state"
    np.testing.assert_allclose(saved_s.cpu().numpy(), saved_x.cpu().numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(h_s.cpu().numpy(), h_x.cpu().numpy(), rtol=1e-5, atol=2e-6)
    hip.lstm_chain_bwd_seg(wd, saved_x, d_emit, n, p0, calls, segs, warm, dg_s, work, err)
    torch.cuda.synchronize()
    assert err[0].item() <= 1e-5 * err[1].item() + 1e-12
    scale = dg_x.abs().max().item()
    assert (dg_s - dg_x).abs().max().item() <= 2e-5 * scale
    # GRU (reward network): p0+1 columns in the first call
    xg, wg = _chain_inputs(rs, n, cols, 3)
    bg = dev(rnd(rs, 3 * H, scale=0.1))
    g_x, g_s = torch.zeros(calls, n, H, device="cuda"), torch.zeros(calls, n, H, device="cuda")
    hip.gru_chain_fwd(dev(xg), dev(wg), bg, n, p0 + 1, calls, g_x, ws, status)
    hip.gru_chain_fwd_seg(dev(xg), dev(wg), bg, n, p0 + 1, calls, segs, warm, g_s, work, err)
    torch.cuda.synchronize()
    assert status.item() == 0 and err[0].item() < 2e-6
    np.testing.assert_allclose(g_s.cpu().numpy(), g_x.cpu().numpy(), rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("n,p0,calls,segs,warm", [(32, 1, 10, 8, 24), (16, 3, 6, 7, 12), (48, 2, 12, 96, 16), (64, 1, 19, 512, 8)])
def test_persistent_chains_equal_the_per_step_launches(n, p0, calls, segs, warm):
    """One launch for both forward chains, with in-launch hand-offs of the hidden state between
    workgroups: every output must equal the per-step kernels' bit for bit, twice in a row (the second
    call reuses the flag words and state buffers of the first)."""
    assert hip.chain_pair_supported(segs)
    rs = np.random.RandomState(segs + 1)
    cols = p0 + calls
    pos_v = n * sum(p0 + s for s in range(calls))
    xv, wv = _chain_inputs(rs, n, cols, 4)
    xg, wg = _chain_inputs(rs, n, cols + 1, 3)
    xv, wv, xg, wg = dev(xv), dev(wv), dev(xg), dev(wg)
    bg = dev(rnd(rs, 3 * H, scale=0.1))
    work_v, err_v = _seg_ws(segs)
    work_r, err_r = _seg_ws(segs)
    h_s, saved_s = torch.zeros(calls, n, H, device="cuda"), torch.zeros(pos_v, 6, H, device="cuda")
    g_s = torch.zeros(calls, n, H, device="cuda")
    hip.lstm_chain_fwd_seg(xv, wv, n, p0, calls, segs, warm, h_s, saved_s, work_v, err_v)
    warm_r = warm + 5   # each chain has its own warm-up length
    hip.gru_chain_fwd_seg(xg, wg, bg, n, p0 + 1, calls, segs, warm_r, g_s, work_r, err_r)
    torch.cuda.synchronize()
    ref_err = (err_v[0].item(), err_r[0].item())
    for _ in range(2):
        h_p, saved_p, g_p = torch.zeros_like(h_s), torch.zeros_like(saved_s), torch.zeros_like(g_s)
        hip.chain_pair_fwd_seg(xv, wv, xg, wg, bg, n, p0, p0 + 1, calls, segs, warm, h_p, saved_p, g_p, work_v, work_r,
                               err_v, err_r, warmup_r=warm_r)
        torch.cuda.synchronize()
        assert (err_v[0].item(), err_r[0].item()) == ref_err
        assert torch.equal(h_p, h_s) and torch.equal(saved_p, saved_s) and torch.equal(g_p, g_s)
    # each forward chain alone in the persistent launch, the LSTM one also from a given start state
    pos_r = n * sum(p0 + 1 + s for s in range(calls))
    gsaved_s = torch.zeros(pos_r, 5, H, device="cuda")
    hip.gru_chain_fwd_seg(xg, wg, bg, n, p0 + 1, calls, segs, warm_r, torch.zeros_like(g_s), work_r, err_r, saved=gsaved_s)
    state = dev(rnd(rs, 2 * H, scale=0.3))
    h_i, saved_i = torch.zeros_like(h_s), torch.zeros_like(saved_s)
    hip.lstm_chain_fwd_seg(xv, wv, n, p0, calls, segs, warm, h_i, saved_i, work_v, err_v, state_in=state)
    torch.cuda.synchronize()
    err_i = err_v[0].item()
    assert not torch.equal(saved_i[:4], saved_s[:4])   # the first positions feel the start state
    for _ in range(2):
        h_p, saved_p, g_p = torch.zeros_like(h_s), torch.zeros_like(saved_s), torch.zeros_like(g_s)
        hip.lstm_chain_fwd_seg(xv, wv, n, p0, calls, segs, warm, h_p, saved_p, work_v, err_v, persistent=True)
        gsaved_p = torch.zeros_like(gsaved_s)
        hip.gru_chain_fwd_seg(xg, wg, bg, n, p0 + 1, calls, segs, warm_r, g_p, work_r, err_r, saved=gsaved_p,
                              persistent=True)
        torch.cuda.synchronize()
        assert (err_v[0].item(), err_r[0].item()) == ref_err
        assert torch.equal(h_p, h_s) and torch.equal(saved_p, saved_s) and torch.equal(g_p, g_s)
        assert torch.equal(gsaved_p, gsaved_s)
        hip.lstm_chain_fwd_seg(xv, wv, n, p0, calls, segs, warm, h_p, saved_p, work_v, err_v, state_in=state,
                               persistent=True)
        torch.cuda.synchronize()
        assert err_v[0].item() == err_i
        assert torch.equal(h_p, h_i) and torch.equal(saved_p, saved_i)
    # the backward chain: persistent launch against the per-step launches
    d_emit = dev(rnd(rs, calls, n, H, scale=0.01))
    dg_s = torch.zeros(pos_v, 4 * H, device="cuda")
    err_b, err_p = torch.zeros(2, device="cuda"), torch.zeros(2, device="cuda")
    hip.lstm_chain_bwd_seg(wv, saved_s, d_emit, n, p0, calls, segs, warm, dg_s, work_v, err_b)
    for _ in range(2):
        dg_p = torch.full_like(dg_s, float("nan"))
        hip.lstm_chain_bwd_seg(wv, saved_s, d_emit, n, p0, calls, segs, warm, dg_p, work_v, err_p, persistent=True)
        torch.cuda.synchronize()
        assert torch.equal(err_p, err_b)
        assert torch.equal(dg_p, dg_s)


def test_seam_check_detects_a_chain_that_has_not_forgotten():
    """Forget gates pinned near 1 and a warm-up of 2 steps: the seam error must say so."""
    rs = np.random.RandomState(9)
    n, p0, calls, segs = 32, 1, 8, 8
    xproj, w_hh = _chain_inputs(rs, n, p0 + calls, 4)
    xproj[:, H:2 * H] += 8.0  # forget-gate pre-activation
    work, err = _seg_ws(segs)
    positions = n * sum(p0 + s for s in range(calls))
    h_s, saved_s = torch.zeros(calls, n, H, device="cuda"), torch.zeros(positions, 6, H, device="cuda")
    hip.lstm_chain_fwd_seg(dev(xproj), dev(w_hh), n, p0, calls, segs, 2, h_s, saved_s, work, err)
    torch.cuda.synchronize()
    assert err[0].item() > 1e-2


def test_splitk_gemm_and_slab_sum():
    rs = np.random.RandomState(12)
    m, n, k, splits = 200, 96, 5000, 7
    a, b = rnd(rs, k, m), rnd(rs, k, n)
    slabs = torch.zeros(splits, m, n, device="cuda")
    hip.gemm_splitk(dev(a), dev(b), slabs, splits, a_kmajor=True, b_kmajor=True)
    out = torch.zeros(m, n, device="cuda")
    hip.sum_slabs(slabs, out)
    want = a.astype(np.float64).T @ b.astype(np.float64)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("n,e", [(45, 512), (512, 512), (7, 320)])
def test_fused_policy_lstm_step_matches_the_unfused_kernels(n, e):
    rs = np.random.RandomState(n + e)
    vocab = 300
    emb, w_ih, w_hh = rnd(rs, vocab, e), rnd(rs, 4 * H, e, scale=0.05), rnd(rs, 4 * H, H, scale=0.05)
    b_ih, b_hh = rnd(rs, 4 * H, scale=0.1), rnd(rs, 4 * H, scale=0.1)
    h0, c0 = rnd(rs, n, H, scale=0.5), rnd(rs, n, H)
    tok = rs.randint(0, vocab, size=n).astype(np.int64)
    d = {k: dev(v) for k, v in dict(emb=emb, w_ih=w_ih, w_hh=w_hh, b_ih=b_ih, b_hh=b_hh, h0=h0, c0=c0).items()}
    tok_d = dev(tok, torch.int64)
    gates, c1, h1 = (torch.zeros(n, 4 * H, device="cuda"), torch.zeros(n, H, device="cuda"),
                     torch.zeros(n, H, device="cuda"))
    x_out = torch.zeros(n, e, device="cuda")
    hip.policy_lstm_step(d["emb"], tok_d, d["w_ih"], d["w_hh"], d["b_ih"], d["b_hh"], d["h0"], d["c0"], gates, c1, h1,
                         x_out)
    pre = (emb[tok].astype(np.float64) @ w_ih.T.astype(np.float64) + h0.astype(np.float64) @ w_hh.T.astype(np.float64)
           + b_ih + b_hh)
    act_w, c_w, h_w = ref.lstm_cell(pre, c0)
    assert np.array_equal(x_out.cpu().numpy(), emb[tok])
    np.testing.assert_allclose(gates.cpu().numpy(), act_w.numpy(), rtol=1e-5, atol=3e-6)
    np.testing.assert_allclose(c1.cpu().numpy(), c_w.numpy(), rtol=1e-5, atol=3e-6)
    np.testing.assert_allclose(h1.cpu().numpy(), h_w.numpy(), rtol=1e-5, atol=3e-6)


# --------------------------------------------------------------------------- top-k
@pytest.mark.parametrize("n,v,k", [(5, 100, 5), (64, 10000, 5), (3, 257, 8), (2, 16384, 1)])
def test_topk_rows_matches_torch(n, v, k):
    rs = np.random.RandomState(v + k)
    x = dev(rnd(rs, n, v))
    vals, idx = hip.topk_rows(x, k)
    want_v, want_i = torch.topk(x, k, dim=1)
    assert torch.equal(vals, want_v) and torch.equal(idx, want_i)
    # ties: lowest index first
    x[:, 7] = x[:, 3] = 100.0
    vals, idx = hip.topk_rows(x, min(k, 2))
    assert idx[:, 0].tolist() == [3] * n and (k == 1 or idx[:, 1].tolist() == [7] * n)


# --------------------------------------------------------------------------- reward pre-training kernels
def test_gru_chain_backward_and_vse_loss_match_autograd():
    """GRU chain forward (activations kept) + backward against float64 autograd of the same batch-1
    recurrence, and the visual-semantic embedding loss with its gradient against the formula."""
    rs = np.random.RandomState(21)
    n, p0, calls = 5, 3, 1
    cols = p0
    xproj, w_hh = _chain_inputs(rs, n, cols, 3)
    b_hh = rnd(rs, 3 * H, scale=0.1)
    d_emit = rnd(rs, calls, n, H)
    pos = n * cols
    xd, wd, bd = dev(xproj), dev(w_hh), dev(b_hh)
    ws, status = _ws()
    h_emit, saved = torch.zeros(calls, n, H, device="cuda"), torch.zeros(pos, 5, H, device="cuda")
    dgx, dgh = torch.zeros(pos, 3 * H, device="cuda"), torch.zeros(pos, 3 * H, device="cuda")
    hip.gru_chain_fwd(xd, wd, bd, n, p0, calls, h_emit, ws, status, saved=saved)
    hip.gru_chain_bwd(wd, saved, dev(d_emit), n, p0, calls, dgx, dgh, ws, status)
    torch.cuda.synchronize()
    assert status.item() == 0
    xt = torch.tensor(xproj, dtype=torch.float64, requires_grad=True)
    wt = torch.tensor(w_hh, dtype=torch.float64, requires_grad=True)
    bt = torch.tensor(b_hh, dtype=torch.float64, requires_grad=True)
    h = torch.zeros(H, dtype=torch.float64)
    hs = []
    for t in range(cols):
        for i in range(n):
            hp = wt @ h + bt
            xr, xz, xn = xt[t * n + i].split(H)
            hr, hz, hn = hp.split(H)
            r, z = torch.sigmoid(xr + hr), torch.sigmoid(xz + hz)
            nn_ = torch.tanh(xn + r * hn)
            h = (1 - z) * nn_ + z * h
            hs.append(h)
    last = torch.stack(hs[-n:])
    np.testing.assert_allclose(h_emit[0].cpu().numpy(), last.detach().numpy(), rtol=1e-5, atol=2e-6)
    (last * torch.tensor(d_emit[0], dtype=torch.float64)).sum().backward()
    np.testing.assert_allclose(dgx.cpu().numpy(), xt.grad.numpy(), rtol=1e-4, atol=1e-6)       # d/d x-side pre-activations
    np.testing.assert_allclose(dgh.sum(0).cpu().numpy(), bt.grad.numpy(), rtol=1e-4, atol=1e-5)  # sum over positions = d/d b_hh
    hprev = torch.zeros(pos, H, device="cuda")
    hip.chain_hprev(saved, pos, hprev, gru=True)
    gw = (dgh.double().t() @ hprev.double()).cpu().numpy()
    np.testing.assert_allclose(gw, wt.grad.numpy(), rtol=1e-4, atol=1e-5)

    m = 37
    ve, se = rnd(rs, m, 64), rnd(rs, m, 64)
    S = dev((ve @ se.T).astype(np.float32))
    loss, dS, rows = torch.zeros(1, device="cuda"), torch.zeros(m, m, device="cuda"), torch.zeros(m, device="cuda")
    hip.vse_loss(S, 0.2, rows, loss, dS)
    St = S.cpu().double().requires_grad_(True)
    margin = (0.2 / m) * (torch.ones(m, m, dtype=torch.float64) - torch.eye(m, dtype=torch.float64))
    want = (torch.relu(St - torch.diag(St).unsqueeze(1) + margin).sum()
            + torch.relu(St.t() - torch.diag(St).unsqueeze(1) + margin).sum()) / m
    want.backward()
    assert abs(loss.item() - want.item()) <= 1e-5 * abs(want.item())
    np.testing.assert_allclose(dS.cpu().numpy(), St.grad.numpy(), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("n,p0,calls,segs,warm", [(64, 17, 1, 32, 64), (48, 2, 7, 9, 48), (512, 17, 1, 512, 64)])
def test_segmented_gru_forward_keeps_activations_and_backward_matches_the_exact_kernels(n, p0, calls, segs, warm):
    """The reward GRU chain segment-parallel, forward with its activations kept and backward, against the
    exact kernels on the same inputs: {r, z, n, hn, h} per position, emitted h, both gate-gradient streams."""
    rs = np.random.RandomState(100 + segs)
    cols = p0 + calls - 1
    pos = n * sum(p0 + s for s in range(calls))
    xproj, w_hh = _chain_inputs(rs, n, cols, 3)
    xd, wd, bd = dev(xproj), dev(w_hh), dev(rnd(rs, 3 * H, scale=0.1))
    d_emit = dev(rnd(rs, calls, n, H, scale=0.01))
    ws, status = _ws()
    h_x, saved_x = torch.zeros(calls, n, H, device="cuda"), torch.zeros(pos, 5, H, device="cuda")
    gx_x, gh_x = torch.zeros(pos, 3 * H, device="cuda"), torch.zeros(pos, 3 * H, device="cuda")
    hip.gru_chain_fwd(xd, wd, bd, n, p0, calls, h_x, ws, status, saved=saved_x)
    hip.gru_chain_bwd(wd, saved_x, d_emit, n, p0, calls, gx_x, gh_x, ws, status)
    work, err_f = _seg_ws(segs)
    err_b = torch.zeros(2, device="cuda")
    h_s, saved_s = torch.zeros_like(h_x), torch.full_like(saved_x, float("nan"))
    gx_s, gh_s = torch.full_like(gx_x, float("nan")), torch.full_like(gh_x, float("nan"))
    hip.gru_chain_fwd_seg(xd, wd, bd, n, p0, calls, segs, warm, h_s, work, err_f, saved=saved_s)
    hip.gru_chain_bwd_seg(wd, saved_s, d_emit, n, p0, calls, segs, warm, gx_s, gh_s, work, err_b)
    torch.cuda.synchronize()
    assert status.item() == 0 and err_f[0].item() < 2e-6
    assert err_b[0].item() <= 1e-4 * err_b[1].item()
    np.testing.assert_allclose(h_s.cpu().numpy(), h_x.cpu().numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(saved_s.cpu().numpy(), saved_x.cpu().numpy(), rtol=1e-5, atol=2e-6)
    for got, want in ((gx_s, gx_x), (gh_s, gh_x)):
        scale = want.abs().max().item()
        assert (got - want).abs().max().item() <= 1e-4 * scale, ((got - want).abs().max().item(), scale)


def test_reproducible_embedding_gradient_and_column_sums():
    """icrl_embedding_grad (rows ordered by token, summed in ascending order) and icrl_colsum with a scratch
    buffer (row blocks' partials added in a fixed order) against float64 sums, and bit-equal from run to run;
    repeated tokens, unused tokens and a strided index included."""
    rs = np.random.RandomState(3)
    vocab, width, rows = 700, 300, 5000
    idx = rs.randint(0, 40, size=rows) * 17 % vocab          # few distinct tokens: long per-token lists
    idx[::7] = rs.randint(0, vocab, size=len(idx[::7]))
    idx2 = np.stack((idx, np.zeros_like(idx)), 1).reshape(-1)  # stride 2
    x = rnd(rs, rows, width)
    want = np.zeros((vocab, width))
    np.add.at(want, idx, x.astype(np.float64))
    work = torch.zeros(hip.embedding_grad_work_ints(vocab, rows), dtype=torch.int32, device="cuda")
    xd, id2 = dev(x), torch.from_numpy(idx2).cuda()
    outs = []
    for _ in range(3):
        g = torch.full((vocab, width), 7.0, device="cuda")
        hip.token_order(id2, 2, rows, vocab, work)
        hip.embedding_grad(g, rows, xd, work)
        torch.cuda.synchronize()
        outs.append(g)
    np.testing.assert_allclose(outs[0].cpu().numpy(), want, rtol=1e-5, atol=1e-5)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert float(outs[0][np.setdiff1d(np.arange(vocab), idx)].abs().max()) == 0.0
    # column sums: 9728 x 2048 (many row blocks) and a one-column matrix
    for r, c in ((9728, 2048), (300, 1), (19, 70000)):
        m = rnd(rs, r, c)
        md = dev(m)
        scratch = torch.zeros(hip.colsum_scratch_floats(r, c), device="cuda")
        got = []
        for _ in range(2):
            o = torch.full((c,), 3.0, device="cuda")
            hip.colsum(md, o, scratch=scratch)
            got.append(o)
        np.testing.assert_allclose(got[0].cpu().numpy(), m.astype(np.float64).sum(0), rtol=1e-4, atol=1e-3)
        assert torch.equal(got[0], got[1])
        hip.colsum(md, got[1], accumulate=True, scratch=scratch)
        np.testing.assert_allclose(got[1].cpu().numpy(), 2 * m.astype(np.float64).sum(0), rtol=1e-4, atol=2e-3)


@pytest.mark.parametrize("a_km", [False, True])
@pytest.mark.parametrize("b_km", [False, True])
@pytest.mark.parametrize("shape", [(70, 130, 77), (512, 2048, 1024), (256, 1000, 300), (33, 64, 5), (200, 10000, 512)])
def test_bf16_gemm_layouts(a_km, b_km, shape):
    """icrl_gemm_bf16: operands rounded to bf16 (nearest even), exact products, float32 sums - against the same
    rounding followed by a float64 product; bias, accumulate, ragged edges, both tile sizes."""
    m, n, k = shape
    rs = np.random.RandomState(m + n + k)
    a = rnd(rs, *((k, m) if a_km else (m, k)))
    b = rnd(rs, *((k, n) if b_km else (n, k)))
    bias = rnd(rs, n)
    a64 = torch.from_numpy(a).bfloat16().double()
    b64 = torch.from_numpy(b).bfloat16().double()
    want = ((a64.t() if a_km else a64) @ (b64 if b_km else b64.t())).numpy()
    scale = np.sqrt(k)
    for tile in (1, 3):
        out = torch.full((m, n), 9.0, device="cuda")
        hip.gemm(dev(a), dev(b), out, a_kmajor=a_km, b_kmajor=b_km, bias=dev(bias), tile=tile, dtype="bf16")
        np.testing.assert_allclose(out.cpu().numpy(), want + bias, rtol=1e-5, atol=2e-6 * scale)
        hip.gemm(dev(a), dev(b), out, a_kmajor=a_km, b_kmajor=b_km, accumulate=True, tile=tile, dtype="bf16")
        np.testing.assert_allclose(out.cpu().numpy(), 2 * want + bias, rtol=1e-5, atol=4e-6 * scale)
    # split-K slabs through the planner
    scratch = torch.zeros(8 * m * n, device="cuda")
    out = torch.zeros(m, n, device="cuda")
    if not a_km and not b_km:
        slabs = scratch[:3 * m * n].view(3, m, n)
        hip.gemm_splitk(dev(a), dev(b), slabs, 3, dtype="bf16")
        hip.sum_slabs(slabs, out)
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-5, atol=2e-6 * scale)


def test_bf16_gemm_with_row_gather_and_strided_output():
    rs = np.random.RandomState(8)
    table, w = rnd(rs, 50, 96), rnd(rs, 40, 96)
    idx = rs.randint(0, 50, size=77)
    out = torch.zeros(77, 64, device="cuda")
    hip.gemm(dev(table), dev(w), out[:, 8:48], a_rows=torch.from_numpy(idx).cuda(), dtype="bf16")
    want = torch.from_numpy(table[idx]).bfloat16().double() @ torch.from_numpy(w).bfloat16().double().t()
    np.testing.assert_allclose(out[:, 8:48].cpu().numpy(), want.numpy(), rtol=1e-5, atol=2e-5)
    assert float(out[:, :8].abs().max()) == 0.0 and float(out[:, 48:].abs().max()) == 0.0


@pytest.mark.parametrize("mode", ["sample", "forced"])
def test_sampler_with_next_cell_update(mode):
    """icrl_sample_tokens_lstm_cell against icrl_sample_tokens followed by the cell on gates = pre + table[token]:
    tokens, log-probs and probabilities bit-equal (the sampling part is the same code), gates / c / h against a
    float64 cell; logits and pre are column ranges of one wide matrix, as the rollout passes them."""
    rs = np.random.RandomState(11)
    n, v, hh = 37, 1000, 512
    ext = dev(rnd(rs, n, v + 4 * hh))
    table = dev(rnd(rs, v, 4 * hh))
    c_prev = dev(rnd(rs, n, hh))
    u = torch.from_numpy(rs.random_sample(n)).cuda()
    m = hip.SAMPLE if mode == "sample" else hip.FORCED
    forced = torch.from_numpy(rs.randint(0, v, size=n)).cuda()
    tok_a, tok_b = forced.clone(), forced.clone()
    lp_a, lp_b = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    pr_a, pr_b = torch.zeros(n, v, device="cuda"), torch.zeros(n, v, device="cuda")
    logits = ext[:, :v].contiguous()
    hip.sample_tokens(logits, m, tok_a, 1, uniforms=u if mode == "sample" else None, log_prob=lp_a, probs=pr_a)
    gates, c_out, h_out = (torch.zeros(n, w, device="cuda") for w in (4 * hh, hh, hh))
    h_pieces, _ = hip.f16x2_pieces(n, hh, "cuda")
    hip.sample_tokens_lstm_cell(ext[:, :v], m, tok_b, 1, ext[:, v:], table, gates, c_prev, c_out, h_out,
                                uniforms=u if mode == "sample" else None, log_prob=lp_b, probs=pr_b, h_pieces=h_pieces)
    torch.cuda.synchronize()
    assert torch.equal(tok_a, tok_b) and torch.equal(lp_a, lp_b) and torch.equal(pr_a, pr_b)
    # the pieces the launch leaves for the next round's product are those the cutter makes of its h, bit for bit
    want_pieces, want_descale = hip.f16x2_pieces(n, hh, "cuda")
    hip.cut_f16x2(h_out, want_pieces, want_descale, fixed_scale=4096.0)
    torch.cuda.synchronize()
    assert torch.equal(h_pieces.view(torch.int16), want_pieces.view(torch.int16))
    assert torch.equal(want_descale, torch.full((n,), 2.0 ** -12, device="cuda"))
    pre = ext[:, v:].double().cpu() + table.double().cpu()[tok_b.cpu()]
    i, f, g, o = (pre[:, q * hh:(q + 1) * hh] for q in range(4))
    i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
    c = f * c_prev.double().cpu() + i * g
    h = o * torch.tanh(c)
    np.testing.assert_allclose(gates.cpu().numpy(), torch.cat((i, f, g, o), 1).numpy(), rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(c_out.cpu().numpy(), c.numpy(), rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(h_out.cpu().numpy(), h.numpy(), rtol=2e-6, atol=2e-6)


# --------------------------------------------------------------------------- pre-cut f16 pieces
def _pieces_to_matrix(pieces, descale, rows, k):
    """The value a piece buffer stands for, (hi + mid) * descale, as a float64 [rows, k] matrix."""
    rb, kb = -(-rows // 32), k // 16
    p = pieces.cpu().double().view(rb, kb, 2, 2, 32, 8)          # row block, k block, piece, k half, row, k
    x = (p[:, :, 0] + p[:, :, 1]).permute(0, 3, 1, 2, 4).reshape(rb * 32, k)   # -> [row block, row, k block, half, k]
    return x[:rows] * descale.cpu().double()[:, None], x[rows:]


@pytest.mark.parametrize("rows,k,fixed", [(70, 48, 0.0), (512, 512, 4096.0), (1000, 512, 0.0), (33, 2048, 0.0)])
def test_f16x2_cutter_pieces_add_up_to_the_element(rows, k, fixed):
    """hi + mid, descaled, is the float32 element to 2^-22 of the row's largest entry (exactly that relative to the
    element itself wherever both pieces are normal f16 numbers); the padding rows stay zero; a fixed scale is used as
    given; rows may be slices of a wider matrix."""
    rs = np.random.RandomState(rows + k)
    wide = dev(rnd(rs, rows, k + 16) * np.exp(rs.standard_normal((rows, 1)) * 3).astype(np.float32))
    if fixed:
        wide = torch.tanh(wide)
    x = wide[:, 4:4 + k] if not fixed else wide[:, :k]
    if not fixed:
        x[min(5, rows - 1)] = 0.
    pieces, descale = hip.f16x2_pieces(rows, k, "cuda")
    hip.cut_f16x2(x, pieces, descale, fixed_scale=fixed)
    torch.cuda.synchronize()
    got, padding = _pieces_to_matrix(pieces, descale, rows, k)
    assert not padding.any()
    xd = x.cpu().double()
    top = xd.abs().max(dim=1, keepdim=True).values
    if fixed:
        assert torch.equal(descale.cpu(), torch.full((rows,), 1.0 / fixed))
        top = torch.ones_like(top)
    else:
        scaled_top = top[:, 0] / descale.cpu().double()
        nz = top[:, 0] > 0
        assert ((scaled_top[nz] >= 1024) & (scaled_top[nz] < 2048)).all()
    assert ((got - xd).abs() <= 2.0 ** -22 * top).all()
    assert torch.equal(got[xd == 0], xd[xd == 0])


@pytest.mark.parametrize("m,n,k", [(512, 12096, 512), (64, 12096, 512), (4, 2148, 512), (70, 1001, 48), (130, 200, 1040)])
def test_f16x2_gemm_against_float64(m, n, k):
    """icrl_gemm_f16x2 on operands cut by icrl_cut_f16x2 against the float64 product of the float32 operands: the
    error bound of an f32 product (a few 2^-24 of sum |a||b| per element) with room for the pieces' 2^-21; ragged
    tiles, a k that is not a multiple of the pipeline depth, a vocabulary-shaped and an unaligned n (single-float
    stores) included.  Rows of A span several orders of magnitude (each has its own scale)."""
    rs = np.random.RandomState(m + n + k)
    a = dev(rnd(rs, m, k) * np.exp(rs.standard_normal((m, 1)) * 2).astype(np.float32))
    b = dev(rnd(rs, n, k, scale=0.05) * np.exp(rs.standard_normal((n, 1))).astype(np.float32))
    bias = dev(rnd(rs, n))
    ap, ad = hip.f16x2_pieces(m, k, "cuda")
    bp, bd = hip.f16x2_pieces(n, k, "cuda")
    hip.cut_f16x2(a, ap, ad)
    hip.cut_f16x2(b, bp, bd)
    wide = torch.full((m, n + 3), 7.0, device="cuda")
    out = wide[:, :n] if n % 4 else torch.empty(m, n, device="cuda")
    hip.gemm_f16x2(ap, ad, bp, bd, out, k, bias=bias)
    plain = torch.empty(m, n, device="cuda")
    hip.gemm_f16x2(ap, ad, bp, bd, plain, k)
    torch.cuda.synchronize()
    want = a.double() @ b.double().t()
    bound = a.double().abs() @ b.double().abs().t()
    err = (plain.double() - want).abs() / bound
    print("%d x %d x %d: max |error| / sum|a||b| = %.3g" % (m, n, k, err.max().item()))
    assert err.max().item() < 1e-6
    np.testing.assert_array_equal(out.cpu().numpy(), (plain + bias).cpu().numpy())
    if n % 4:
        assert (wide[:, n:] == 7.0).all()


def test_f16x2_gemm_rejects_what_it_does_not_do():
    pieces, descale = hip.f16x2_pieces(32, 32, "cuda")
    out = torch.empty(32, 32, device="cuda")
    with pytest.raises(AssertionError):
        hip.f16x2_pieces(32, 24, "cuda")                       # k not a multiple of 16
    with pytest.raises(AssertionError):
        hip.gemm_f16x2(pieces, descale, pieces, descale, torch.empty(64, 32, device="cuda"), 32)   # A buffer too small
    rc = hip.load().icrl_gemm_f16x2(pieces.data_ptr(), descale.data_ptr(), pieces.data_ptr(), descale.data_ptr(),
                                    out.data_ptr(), 16, 32, 32, 32, None, None)
    assert rc != 0 and b"ldc" in hip.load().icrl_last_error()


# --------------------------------------------------------------------------- scaled two-piece f16 products
def _scales_for(x_rows):
    """Reference for absmax_scales along the last axis: 2^(15 - e) with max|row| = f 2^e, f in [0.5, 1)."""
    top = np.abs(x_rows).max(axis=1)
    _, e = np.frexp(top)
    return np.where(top > 0, np.ldexp(1.0, 15 - e), 2.0 ** 15).astype(np.float32)


@pytest.mark.parametrize("rows,cols", [(64, 256), (9728, 2048), (70, 1000), (300, 12)])
def test_absmax_scales_rows_and_columns(rows, cols):
    rs = np.random.RandomState(rows + cols)
    x = rnd(rs, rows, cols + 4) * np.exp(rs.standard_normal((rows, 1)) * 4).astype(np.float32)
    x[:, 8] *= 1e-6
    x[min(3, rows - 1)] = 0.
    xd = dev(x)[:, :cols]
    work = torch.zeros(rows + cols, dtype=torch.int32, device="cuda")
    rscale, cscale = torch.zeros(rows, device="cuda"), torch.zeros(cols, device="cuda")
    hip.absmax_scales(xd, work, row_scale=rscale, col_scale=cscale)
    np.testing.assert_array_equal(rscale.cpu().numpy(), _scales_for(x[:, :cols]))
    np.testing.assert_array_equal(cscale.cpu().numpy(), _scales_for(x[:, :cols].T))
    only_c = torch.zeros(cols, device="cuda")
    hip.absmax_scales(xd, work, col_scale=only_c)
    assert torch.equal(only_c, cscale)
    scaled = np.abs(x[:, :cols]).max(axis=1) * rscale.cpu().numpy()
    assert ((scaled[scaled > 0] >= 2 ** 14) & (scaled[scaled > 0] < 2 ** 15)).all()


def test_g0_row_scales_bound_the_row():
    """The scale computed from log p_token keeps every entry of onehot - softmax below f16's largest number and the
    token entry at 2^13 or above (the estimate may sit one binade off the exact row maximum, never more)."""
    rs = np.random.RandomState(3)
    logits = rnd(rs, 400, 1000) * np.linspace(0.1, 30, 400, dtype=np.float32)[:, None]
    p = torch.softmax(torch.from_numpy(logits).double(), 1)
    tok = torch.from_numpy(rs.randint(0, 1000, size=400))
    tok[:200] = p[:200].argmax(1)
    g0 = -p.clone()
    g0[torch.arange(400), tok] += 1.0
    logp = torch.log(p[torch.arange(400), tok]).float().cuda()
    scale = torch.zeros(400, device="cuda")
    hip.g0_row_scales(logp, scale)
    scaled = (g0.abs().max(1).values * scale.cpu().double()).numpy()
    assert (scaled < 65504).all()
    exact = g0.abs().max(1).values.numpy() > 2.0 ** -21
    assert (scaled[exact] >= 2 ** 13).all()


@pytest.mark.parametrize("a_km", [False, True])
@pytest.mark.parametrize("b_km", [False, True])
@pytest.mark.parametrize("shape", [(128, 128, 64), (200, 130, 333), (77, 300, 2050), (2048, 512, 9728)])
def test_f16x2s_gemm_layouts(a_km, b_km, shape):
    """icrl_gemm_f16x2s with the scales icrl_absmax_scales makes, every layout, split-K included, against float64;
    rows of both operands span many orders of magnitude (what the scales are for).  Error bound of an f32 product
    with room for the pieces' 2^-21."""
    m, n, k = shape
    rs = np.random.RandomState(m + n + k)
    a = rnd(rs, m, k) * np.exp(rs.standard_normal((m, 1)) * 4).astype(np.float32)
    b = rnd(rs, n, k) * np.exp(rs.standard_normal((n, 1)) * 4).astype(np.float32)
    a[:, ::7] *= 1e-4
    a_st, b_st = dev(a.T.copy() if a_km else a), dev(b.T.copy() if b_km else b)
    work = torch.zeros(max(m, n) + k, dtype=torch.int32, device="cuda")
    sa, sb = torch.zeros(m, device="cuda"), torch.zeros(n, device="cuda")
    hip.absmax_scales(a_st, work, **({"col_scale": sa} if a_km else {"row_scale": sa}))
    hip.absmax_scales(b_st, work, **({"col_scale": sb} if b_km else {"row_scale": sb}))
    bias = dev(rnd(rs, n))
    out = torch.empty(m, n, device="cuda")
    hip.gemm(a_st, b_st, out, a_kmajor=a_km, b_kmajor=b_km, bias=bias, dtype="f16x2", a_scale=sa, b_scale=sb)
    want = a.astype(np.float64) @ b.astype(np.float64).T
    bound = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64).T
    err = np.abs(out.cpu().numpy().astype(np.float64) - bias.cpu().numpy() - want) / (bound + np.abs(bias.cpu().numpy()))
    print("%s km %d%d: max |error| / sum|a||b| = %.3g" % (shape, a_km, b_km, err.max()))
    assert err.max() < 1e-6
    scratch = torch.empty(8 * m * n, device="cuda")
    auto = torch.empty(m, n, device="cuda")
    hip.gemm_auto(a_st, b_st, auto, scratch, a_kmajor=a_km, b_kmajor=b_km, shared=True, dtype="f16x2", a_scale=sa, b_scale=sb)
    err = np.abs(auto.cpu().numpy().astype(np.float64) - want) / bound
    assert err.max() < 1e-6
    acc = out.clone()
    hip.gemm(a_st, b_st, acc, a_kmajor=a_km, b_kmajor=b_km, accumulate=True, dtype="f16x2", a_scale=sa, b_scale=sb)
    twice = np.abs(acc.cpu().numpy().astype(np.float64) - bias.cpu().numpy() - 2 * want) / (bound + np.abs(bias.cpu().numpy()))
    assert twice.max() < 2e-6


@pytest.mark.parametrize("a_km", [False, True])
@pytest.mark.parametrize("b_km", [False, True])
@pytest.mark.parametrize("shape,splits", [((256, 256, 64), 1), ((300, 700, 333), 2), ((1000, 512, 2050), 3),
                                          ((2560, 512, 9728), 6)])
def test_f16x2s_gemm_on_256_tiles(a_km, b_km, shape, splits):
    """icrl_gemm_f16x2s_splitk_tile with tile 2 (256 x 256, 512 threads): every layout, ragged tile rows and columns,
    ragged k, against float64 - and bit for bit the slabs of the 128 x 128 tiles for the same splits (the k order
    within a split, the pieces and their order are the same; only who computes which 32 x 32 block differs)."""
    m, n, k = shape
    rs = np.random.RandomState(m + n + k + 1)
    a = rnd(rs, m, k) * np.exp(rs.standard_normal((m, 1)) * 4).astype(np.float32)
    b = rnd(rs, n, k) * np.exp(rs.standard_normal((n, 1)) * 4).astype(np.float32)
    a_st, b_st = dev(a.T.copy() if a_km else a), dev(b.T.copy() if b_km else b)
    work = torch.zeros(max(m, n) + k, dtype=torch.int32, device="cuda")
    sa, sb = torch.zeros(m, device="cuda"), torch.zeros(n, device="cuda")
    hip.absmax_scales(a_st, work, **({"col_scale": sa} if a_km else {"row_scale": sa}))
    hip.absmax_scales(b_st, work, **({"col_scale": sb} if b_km else {"row_scale": sb}))
    big, std = torch.full((splits, m, n), float("nan"), device="cuda"), torch.empty(splits, m, n, device="cuda")
    hip.gemm_splitk(a_st, b_st, big, splits, a_kmajor=a_km, b_kmajor=b_km, tile=2, dtype="f16x2", a_scale=sa, b_scale=sb)
    hip.gemm_splitk(a_st, b_st, std, splits, a_kmajor=a_km, b_kmajor=b_km, tile=1, dtype="f16x2", a_scale=sa, b_scale=sb)
    assert torch.equal(big, std)
    want = a.astype(np.float64) @ b.astype(np.float64).T
    bound = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64).T
    err = np.abs(big.sum(0).cpu().numpy().astype(np.float64) - want) / bound
    assert err.max() < 1e-6
    # through gemm_auto, as the engine calls it
    scratch = torch.empty(8 * m * n, device="cuda")
    auto = torch.empty(m, n, device="cuda")
    hip.gemm_auto(a_st, b_st, auto, scratch, a_kmajor=a_km, b_kmajor=b_km, shared=True, dtype="f16x2", a_scale=sa, b_scale=sb,
                  big_tile=True)
    assert (np.abs(auto.cpu().numpy().astype(np.float64) - want) / bound).max() < 1e-6


# --------------------------------------------------------------------------- policy LSTM backward-through-time
@pytest.mark.parametrize("n,t_steps,dtype", [(512, 19, "f32"), (48, 5, "f32"), (64, 3, "bf16")])
def test_policy_lstm_bptt_equals_the_launches_issued_one_by_one_and_autograd(n, t_steps, dtype):
    """icrl_policy_lstm_bptt queues cell backward + dgates . W_hh per step itself: bit for bit what the same
    launches give when issued one by one, and (float32) the gradients autograd gives for models.py:80."""
    rs = np.random.RandomState(5)
    w_hh = dev(rnd(rs, 4 * H, H, scale=H ** -0.5))
    xw = rnd(rs, t_steps, n, 4 * H)                   # the input side of every step's gates, held fixed
    c_all = torch.zeros(t_steps + 1, n, H, device="cuda")
    h_all = torch.zeros(t_steps + 1, n, H, device="cuda")
    c_all[0], h_all[0] = dev(rnd(rs, n, H)), dev(rnd(rs, n, H))
    gates = torch.empty(t_steps, n, 4 * H, device="cuda")
    for k in range(t_steps):                          # forward with the kernels, keeping what the backward pass reads
        gates[k] = dev(xw[k])
        hip.gemm(h_all[k], w_hh, gates[k], accumulate=True)
        hip.lstm_cell_fwd(gates[k], c_all[k], c_all[k + 1], h_all[k + 1])
    d_out = dev(rnd(rs, t_steps, n, H, scale=0.1))    # gradient reaching every step's output h
    scratch = torch.empty(16 * n * H, device="cuda")

    def one_by_one():
        d_h, dgp = d_out.clone(), torch.empty(t_steps, n, 4 * H, device="cuda")
        dc, dh0 = [torch.empty(n, H, device="cuda") for _ in range(2)], torch.empty(n, H, device="cuda")
        for k in range(t_steps - 1, -1, -1):
            hip.lstm_cell_bwd(gates[k], c_all[k], c_all[k + 1], d_h[k], None if k == t_steps - 1 else dc[(k + 1) & 1],
                              dgp[k], dc[k & 1])
            out = d_h[k - 1] if k > 0 else dh0
            hip.gemm_auto(dgp[k], w_hh, out, scratch, b_kmajor=True, accumulate=k > 0, shared=True, dtype=dtype)
        return dgp, dh0, dc[0]

    d_h, dgp = d_out.clone(), torch.empty(t_steps, n, 4 * H, device="cuda")
    dc0, dc1, dh0 = (torch.empty(n, H, device="cuda") for _ in range(3))
    hip.policy_lstm_bptt(gates, c_all, d_h, w_hh, dgp, dc0, dc1, dh0, scratch, shared=True, dtype=dtype)
    want = one_by_one()
    torch.cuda.synchronize()
    assert torch.equal(dgp, want[0]) and torch.equal(dh0, want[1]) and torch.equal(dc0, want[2])
    if dtype != "f32":
        return
    # autograd on the CPU over the same recurrence
    w = w_hh.cpu().double().requires_grad_(True)
    h = h_all[0].cpu().double().requires_grad_(True)
    c = c_all[0].cpu().double().requires_grad_(True)
    h0, c0, loss = h, c, 0.0
    for k in range(t_steps):
        g = torch.from_numpy(xw[k]).double() + h @ w.t()
        i, f, gg, o = g.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        loss = loss + (h * d_out[k].cpu().double()).sum()
    loss.backward()
    np.testing.assert_allclose(dh0.cpu().numpy(), h0.grad.numpy(), rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(dc0.cpu().numpy(), c0.grad.numpy(), rtol=2e-4, atol=2e-6)
    dw = (dgp.view(-1, 4 * H).cpu().double().t() @ h_all[:t_steps].reshape(-1, H).cpu().double())
    np.testing.assert_allclose(dw.numpy(), w.grad.numpy(), rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("v,mode", [(10000, "sample"), (10000, "forced"), (257, "sample")])
def test_sampler_emits_onehot_minus_softmax_bit_equal_to_the_separate_pass(v, mode):
    """mode | EMIT_DLOGITS: the probs output holds what icrl_policy_dlogits makes of the softmax (coef = 1)."""
    rs = np.random.RandomState(11)
    n = 96
    logits = dev(rnd(rs, n, v, scale=2.0))
    u = dev(rs.random_sample(n), torch.float64)
    m = hip.SAMPLE if mode == "sample" else hip.FORCED
    tok = torch.zeros(n, dtype=torch.int64, device="cuda")
    if mode == "forced":
        tok.copy_(dev(rs.randint(0, v, size=n), torch.int64))
    tok2 = tok.clone()
    lp, lp2 = torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
    probs, g0 = torch.empty(n, v, device="cuda"), torch.empty(n, v, device="cuda")
    hip.sample_tokens(logits, m, tok, 1, uniforms=u if mode == "sample" else None, log_prob=lp, probs=probs)
    hip.sample_tokens(logits, m | hip.EMIT_DLOGITS, tok2, 1, uniforms=u if mode == "sample" else None, log_prob=lp2,
                      probs=g0)
    want = torch.empty(n, v, device="cuda")
    hip.policy_dlogits(probs, want, tok, 1, None)
    torch.cuda.synchronize()
    assert torch.equal(tok, tok2) and torch.equal(lp, lp2) and torch.equal(g0, want)
    with pytest.raises(RuntimeError, match="EMIT_DLOGITS"):
        hip.sample_tokens(logits, hip.GREEDY | hip.EMIT_DLOGITS, tok2, 1, probs=g0)


# --------------------------------------------------------------------------- GEMM from three bf16 pieces per element
@pytest.mark.parametrize("a_km", [False, True])
@pytest.mark.parametrize("b_km", [False, True])
@pytest.mark.parametrize("shape", [(128, 128, 64), (200, 130, 333), (77, 300, 2050), (256, 512, 37)])
def test_f32x3_gemm_layouts(a_km, b_km, shape):
    """icrl_gemm_f32x3: every element cut into three bf16 pieces, six piece products per element product, float32
    sums.  Against a float64 product: within the exact-f32 kernel's own error bound (1e-7 * sum|a*b| per rounding,
    here with the three dropped piece products on top); bias, accumulate, ragged edges, split-K slabs."""
    m, n, k = shape
    rs = np.random.RandomState(m + n + k + 1)
    a = rnd(rs, *((k, m) if a_km else (m, k)))
    b = rnd(rs, *((k, n) if b_km else (n, k)))
    bias = rnd(rs, n)
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    a64, b64 = (a64.T if a_km else a64), (b64 if b_km else b64.T)
    want = a64 @ b64
    tol = 6e-7 * (np.abs(a64) @ np.abs(b64)) + 1e-6
    for tile in (0, 1):   # (one tile shape: 128x128)
        out = torch.full((m, n), 9.0, device="cuda")
        hip.gemm(dev(a), dev(b), out, a_kmajor=a_km, b_kmajor=b_km, bias=dev(bias), tile=tile, dtype="f32x3")
        assert np.all(np.abs(out.cpu().numpy() - (want + bias)) <= tol)
        hip.gemm(dev(a), dev(b), out, a_kmajor=a_km, b_kmajor=b_km, accumulate=True, tile=tile, dtype="f32x3")
        assert np.all(np.abs(out.cpu().numpy() - (2 * want + bias)) <= 2 * tol)
        if k >= 300:
            slabs = torch.zeros(3, m, n, device="cuda")
            hip.gemm_splitk(dev(a), dev(b), slabs, 3, a_kmajor=a_km, b_kmajor=b_km, tile=tile, dtype="f32x3")
            hip.sum_slabs(slabs, out)
            assert np.all(np.abs(out.cpu().numpy() - want) <= tol)
    # the same product twice: the same bits (no atomics, fixed summation order)
    again = torch.empty(m, n, device="cuda")
    hip.gemm(dev(a), dev(b), again, a_kmajor=a_km, b_kmajor=b_km, dtype="f32x3")
    first = torch.empty(m, n, device="cuda")
    hip.gemm(dev(a), dev(b), first, a_kmajor=a_km, b_kmajor=b_km, dtype="f32x3")
    assert torch.equal(first, again)


@pytest.mark.parametrize("b_km", [False, True])
def test_f32x3_pieces_add_up_to_the_element_exactly(b_km):
    """hi + mid + lo is the float32 value bit for bit: a product with the identity returns the matrix itself,
    over 60 binades of magnitudes, for both staging paths (k contiguous / rows contiguous) of each operand."""
    rs = np.random.RandomState(3)
    m, k = 150, 96
    a = (rs.standard_normal((m, k)) * np.exp2(rs.randint(-30, 30, size=(m, k)))).astype(np.float32)
    eye = dev(np.eye(k, dtype=np.float32))
    for a_km, tile in ((False, 1), (True, 1)):
        out = torch.empty(m, k, device="cuda")
        hip.gemm(dev(a.T.copy() if a_km else a), eye, out, a_kmajor=a_km, b_kmajor=b_km, tile=tile, dtype="f32x3")
        assert np.array_equal(out.cpu().numpy(), a)
        out_t = torch.empty(k, m, device="cuda")   # the matrix as the B operand
        hip.gemm(eye, dev(a.T.copy() if b_km else a), out_t, a_kmajor=a_km, b_kmajor=b_km, tile=tile, dtype="f32x3")
        assert np.array_equal(out_t.cpu().numpy(), a.T)


def test_f32x3_gemm_rejects_what_it_does_not_do():
    a = torch.zeros(8, 8, device="cuda")
    with pytest.raises(RuntimeError, match="no row gather"):
        hip.gemm(a, a, torch.zeros(8, 8, device="cuda"), bias=a[0], bias2=a[0], dtype="f32x3")


@pytest.mark.parametrize("n,t_steps", [(512, 19), (48, 5), (7, 3)])
def test_persistent_policy_lstm_bptt_against_the_launches_issued_one_by_one(n, t_steps):
    """icrl_policy_lstm_bptt_persist (the backward chain kernel walking N per-row chains with a gradient injected at
    every step, one launch) against icrl_policy_lstm_bptt (cell backward + product per step): the same gate gradients
    and initial-state gradient to float32 rounding (different summation orders), twice in a row."""
    assert hip.chain_pair_supported(n)
    rs = np.random.RandomState(11)
    w_hh = dev(rnd(rs, 4 * H, H, scale=H ** -0.5))
    xw = rnd(rs, t_steps, n, 4 * H)
    c_all = torch.zeros(t_steps + 1, n, H, device="cuda")     # the policy network starts from c = 0
    h_all = torch.zeros(t_steps + 1, n, H, device="cuda")
    h_all[0] = dev(rnd(rs, n, H))
    gates = torch.empty(t_steps, n, 4 * H, device="cuda")
    for k in range(t_steps):
        gates[k] = dev(xw[k])
        hip.gemm(h_all[k], w_hh, gates[k], accumulate=True)
        hip.lstm_cell_fwd(gates[k], c_all[k], c_all[k + 1], h_all[k + 1])
    d_out = dev(rnd(rs, t_steps, n, H, scale=0.1))
    scratch = torch.empty(16 * n * H, device="cuda")
    d_h, dgp = d_out.clone(), torch.empty(t_steps, n, 4 * H, device="cuda")
    dc0, dc1, dh0 = (torch.empty(n, H, device="cuda") for _ in range(3))
    hip.policy_lstm_bptt(gates, c_all, d_h, w_hh, dgp, dc0, dc1, dh0, scratch, shared=True)
    work = torch.zeros(hip.chain_seg_workspace_floats(n), device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    for _ in range(2):
        dgp2, dh02 = torch.full_like(dgp, float("nan")), torch.full_like(dh0, float("nan"))
        hip.policy_lstm_bptt_persist(gates, c_all, d_out, w_hh, dgp2, dh02, work, status)
        torch.cuda.synchronize()
        assert int(status[0]) == 0
        scale = dgp.abs().max().item()
        assert (dgp2 - dgp).abs().max().item() <= 2e-5 * scale
        assert (dh02 - dh0).abs().max().item() <= 2e-5 * dh0.abs().max().item()
    # the owner-computes launch on the f16 matrix instruction
    assert hip.chain_bwd_f16x2_supported(n)
    work_b = torch.zeros(hip.chain_bwd_f16x2_workspace_floats(n), device="cuda")
    for _ in range(2):
        dgp3, dh03 = torch.full_like(dgp, float("nan")), torch.full_like(dh0, float("nan"))
        hip.policy_lstm_bptt_persist(gates, c_all, d_out, w_hh, dgp3, dh03, work_b, status, arith="f16x2")
        torch.cuda.synchronize()
        assert int(status[0]) == 0
        assert (dgp3 - dgp).abs().max().item() <= 2e-5 * scale
        assert (dh03 - dh0).abs().max().item() <= 2e-5 * dh0.abs().max().item()


@pytest.mark.parametrize("n,p0,calls,segs,warm", [(64, 1, 19, 32, 64), (48, 2, 12, 96, 16), (512, 1, 19, 256, 96)])
def test_forward_chain_pair_on_the_f16_matrix_instruction(n, p0, calls, segs, warm):
    """icrl_chain_pair_fwd_seg_f16x2 (operands as two scaled f16 pieces, W_hh in registers) against the exact-float32
    persistent pair launch on the same inputs: every kept activation, emitted h and both seam errors to float32
    rounding level (1e-6 of the activations' unit scale; a recurrence carries a rounding difference along until it is
    forgotten, so not bit for bit), twice in a row; and closer to a float64 evaluation of the same segment than 4e-6."""
    assert hip.chain_pair_supported(segs) and hip.chain_f16x2_supported(segs)
    rs = np.random.RandomState(segs + 3)
    cols = p0 + calls
    pos_v = n * sum(p0 + s for s in range(calls))
    pos_r = n * sum(p0 + 1 + s for s in range(calls))
    xv, wv = _chain_inputs(rs, n, cols, 4)
    xg, wg = _chain_inputs(rs, n, cols + 1, 3)
    wv[:64] *= 6.0            # rows of W_hh of very different sizes: every row has a scale of its own
    wg[5] = 0.0               # an all-zero row
    xv_d, wv_d, xg_d, wg_d = dev(xv), dev(wv), dev(xg), dev(wg)
    bg = dev(rnd(rs, 3 * H, scale=0.1))
    work_v, err_v = _seg_ws(segs)
    work_r, err_r = _seg_ws(segs)
    outs = {}
    for arith in ("f32", "f16x2", "f16x2"):
        h_p, saved_p = torch.zeros(calls, n, H, device="cuda"), torch.full((pos_v, 6, H), 7.0, device="cuda")
        g_p, gsaved_p = torch.zeros(calls, n, H, device="cuda"), torch.full((pos_r, 5, H), 7.0, device="cuda")
        hip.chain_pair_fwd_seg(xv_d, wv_d, xg_d, wg_d, bg, n, p0, p0 + 1, calls, segs, warm, h_p, saved_p, g_p, work_v,
                               work_r, err_v, err_r, warmup_r=warm + 5, saved_r=gsaved_p, arith=arith)
        torch.cuda.synchronize()
        got = (h_p, saved_p, g_p, gsaved_p, err_v[0].item(), err_r[0].item())
        if arith in outs:     # the second f16 call: the same bits as the first
            assert all(torch.equal(a, b) for a, b in zip(got[:4], outs[arith][:4])) and got[4:] == outs[arith][4:]
        outs[arith] = got
    ref, new = outs["f32"], outs["f16x2"]
    assert not (new[1] == 7.0).any() and not (new[3] == 7.0).any(), "every position's activations are written"
    for a, b in zip(new[:4], ref[:4]):
        assert (a - b).abs().max().item() <= 2e-6, (a - b).abs().max().item()
    assert abs(new[4] - ref[4]) <= 2e-6 and abs(new[5] - ref[5]) <= 2e-6
    # float64 walk of the first positions of the LSTM chain (segment 0 starts from the true zero state)
    steps = min(pos_v, 40)
    h = torch.zeros(H, dtype=torch.float64)
    c = torch.zeros(H, dtype=torch.float64)
    w64 = torch.from_numpy(wv).double()
    worst_new = worst_ref = 0.0
    for pos in range(steps):
        col, row = (pos // n), pos % n       # call 0 walks columns 0 .. p0-1 row by row
        if col >= p0:
            break
        g = torch.from_numpy(xv[col * n + row]).double() + w64 @ h
        i, f, gg, o = g.chunk(4)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        worst_new = max(worst_new, (new[1][pos, 5].cpu().double() - h).abs().max().item())
        worst_ref = max(worst_ref, (ref[1][pos, 5].cpu().double() - h).abs().max().item())
    assert worst_new <= 4e-6 and worst_new <= 4.0 * worst_ref + 1e-6, (worst_new, worst_ref)


@pytest.mark.parametrize("n,p0,calls,segs,warm,fc", [(64, 1, 19, 32, 64, 0), (48, 2, 12, 96, 16, 0), (512, 1, 19, 256, 96, 1),
                                                     (40, 1, 12, 50, 24, 2)])
def test_backward_chain_owner_computes_launch_on_the_f16_instruction(n, p0, calls, segs, warm, fc):
    """icrl_lstm_chain_bwd_seg_f16x2 (each workgroup multiplies its own gate gradients, partial sums exchanged, f16
    pieces with a scale per segment row) against the float32 gathering launch on the same kept activations: every gate
    gradient to 2e-6 of the largest one, the same seam verdict, twice in a row; with injected gradients spanning six
    orders of magnitude over the calls, so that the per-row scaling is exercised."""
    assert hip.chain_pair_supported(segs) and hip.chain_bwd_f16x2_supported(segs)
    rs = np.random.RandomState(segs + 9)
    cols = p0 + calls - 1
    positions = hip.chain_positions(n, p0, calls, fc)
    xv, wv = _chain_inputs(rs, n, cols, 4)
    xv_d, wv_d = dev(xv), dev(wv)
    work_v, err_v = _seg_ws(segs)
    h_s, saved_s = torch.zeros(calls, n, H, device="cuda"), torch.zeros(positions, 6, H, device="cuda")
    hip.lstm_chain_fwd_seg(xv_d, wv_d, n, p0, calls, segs, warm, h_s, saved_s, work_v, err_v, persistent=True, full_calls=fc)
    d_emit = rnd(rs, calls, n, H, scale=0.01)
    d_emit *= (10.0 ** rs.uniform(-5, 1, size=(calls, 1, 1))).astype(np.float32)
    d_emit = dev(d_emit)
    dg_ref = torch.zeros(positions, 4 * H, device="cuda")
    err_ref, err_new = torch.zeros(2, device="cuda"), torch.zeros(2, device="cuda")
    hip.lstm_chain_bwd_seg(wv_d, saved_s, d_emit, n, p0, calls, segs, warm, dg_ref, work_v, err_ref, persistent=True,
                           full_calls=fc)
    work_b = torch.zeros(hip.chain_bwd_f16x2_workspace_floats(segs), device="cuda")
    first = None
    for _ in range(2):
        dg_new = torch.full_like(dg_ref, float("nan"))
        hip.lstm_chain_bwd_seg(wv_d, saved_s, d_emit, n, p0, calls, segs, warm, dg_new, work_b, err_new, full_calls=fc,
                               arith="f16x2")
        torch.cuda.synchronize()
        assert torch.isfinite(dg_new).all()
        top = dg_ref.abs().max().item()
        assert (dg_new - dg_ref).abs().max().item() <= 2e-6 * top, ((dg_new - dg_ref).abs().max().item(), top)
        # row by row as well: a position whose gradients are tiny beside the largest is held to its own scale
        rows_top = dg_ref.abs().amax(dim=1)
        rel = ((dg_new - dg_ref).abs().amax(dim=1) / rows_top.clamp_min(1e-30))[rows_top > 1e-12 * top]
        assert rel.max().item() <= 2e-4, rel.max().item()
        e_ref, e_new = err_ref.tolist(), err_new.tolist()
        assert abs(e_new[1] - e_ref[1]) <= 1e-5 * e_ref[1] + 1e-30
        assert abs(e_new[0] - e_ref[0]) <= 2e-6 * e_ref[1] + 1e-30
        if first is not None:
            assert torch.equal(dg_new, first) and e_new == first_err
        first, first_err = dg_new, e_new


# --------------------------------------------------------------------------- a wait that runs out
@pytest.mark.parametrize("wide", [0, 1])
def test_a_wait_that_runs_out_raises_the_abort_word_and_says_who_waited_for_whom(wide, capfd):
    """The bounded wait of the persistent launches on flags that never arrive: after 20 ms it raises the abort word,
    records the waiting workgroup, the step and the producers that were behind, and the follow-up kernel poisons the
    error words and prints one line.  With every flag in place the same wait returns at once and nothing is raised."""
    import time
    producers = 32 if wide else 16
    flags = torch.zeros(32, dtype=torch.int32, device="cuda")
    flags[:producers] = 7
    flags[3] = 6
    flags[producers - 1] = 2
    words = torch.zeros(4, dtype=torch.int32, device="cuda")
    err = torch.zeros(2, device="cuda")
    t0 = time.perf_counter()
    hip.chain_give_up_selftest(flags, words, 7, wide, err)
    torch.cuda.synchronize()
    waited = time.perf_counter() - t0
    assert [w & 0xffffffff for w in words.cpu().tolist()] == [1, 0, 7, (1 << 3) | (1 << (producers - 1))]
    assert torch.isinf(err).all()
    assert 0.015 < waited < 0.5, waited
    out = capfd.readouterr().out
    assert "[icrl] a persistent launch" in out and "step 7" in out and ("mask %08x" % ((1 << 3) | (1 << (producers - 1)))) in out
    flags[:producers] = 7
    words.zero_(); err.zero_()
    t0 = time.perf_counter()
    hip.chain_give_up_selftest(flags, words, 7, wide, err)
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 0.01
    assert words.cpu().tolist() == [0, 0, 0, 0] and not err.any()
    assert "[icrl]" not in capfd.readouterr().out
