"""ctypes binding of libicrl_a2c.so (the C ABI declared in include/icrl_a2c.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``csrc/Makefile``.
There is no CPU fallback: if the shared object is missing, every call raises.
Tensors cross the boundary as raw device pointers (``tensor.data_ptr()``) plus the
current HIP stream; PyTorch only owns the memory.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libicrl_a2c.so")

c_int, c_i64, c_f32, c_f64, c_vp = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double, ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/icrl_a2c.h one to one
SIGNATURES = {
    "icrl_abi_version": (c_int, []),
    "icrl_last_error": (ctypes.c_char_p, []),
    "icrl_gemm_f32": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp, c_i64, c_int, c_vp, c_i64, c_int, c_int, c_int,
                              c_vp, c_vp, c_int, c_int, c_vp]),
    "icrl_gemm_f32_splitk": (c_int, [c_vp, c_i64, c_int, c_vp, c_i64, c_int, c_vp, c_i64, c_i64, c_int, c_int, c_int,
                                     c_int, c_int, c_vp]),
    "icrl_gemm_bf16": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp, c_i64, c_int, c_vp, c_i64, c_int, c_int, c_int,
                               c_vp, c_vp, c_int, c_int, c_vp]),
    "icrl_gemm_bf16_splitk": (c_int, [c_vp, c_i64, c_int, c_vp, c_i64, c_int, c_vp, c_i64, c_i64, c_int, c_int, c_int,
                                      c_int, c_int, c_vp]),
    "icrl_gemm_f32x3": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp, c_i64, c_int, c_vp, c_i64, c_int, c_int, c_int,
                                c_vp, c_vp, c_int, c_int, c_vp]),
    "icrl_gemm_f32x3_splitk": (c_int, [c_vp, c_i64, c_int, c_vp, c_i64, c_int, c_vp, c_i64, c_i64, c_int, c_int, c_int,
                                       c_int, c_int, c_vp]),
    "icrl_f16x2_piece_halves": (c_i64, [c_int, c_int]),
    "icrl_cut_f16x2": (c_int, [c_vp, c_i64, c_int, c_int, ctypes.c_float, c_vp, c_vp, c_vp]),
    "icrl_gemm_f16x2": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_vp, c_vp]),
    "icrl_gemm_f16x2s": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_i64, c_int, c_int, c_int,
                                 c_vp, c_int, c_vp]),
    "icrl_gemm_f16x2s_splitk": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_i64, c_i64, c_int, c_int,
                                        c_int, c_int, c_vp]),
    "icrl_gemm_f16x2s_splitk_tile": (c_int, [c_vp, c_i64, c_int, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_i64, c_i64, c_int, c_int,
                                        c_int, c_int, c_int, c_vp]),
    "icrl_absmax_scales": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "icrl_g0_row_scales": (c_int, [c_vp, c_int, c_vp, c_vp]),
    "icrl_sum_slabs": (c_int, [c_vp, c_i64, c_int, c_i64, c_vp, c_int, c_vp]),
    "icrl_chain_seg_workspace_floats": (c_i64, [c_int]),
    "icrl_lstm_chain_fwd_seg": (c_int, [c_vp, c_vp] + [c_int] * 6 + [c_vp] * 6),
    "icrl_lstm_chain_fwd_seg_static": (c_int, [c_vp, c_vp] + [c_int] * 6 + [c_vp] * 5),
    "icrl_masked_nll": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_vp]),
    "icrl_mse_loss": (c_int, [c_vp, c_vp, c_int, c_vp, c_vp, c_vp]),
    "icrl_topk_rows": (c_int, [c_vp, c_i64, c_int, c_int, c_int, c_vp, c_vp, c_vp]),
    "icrl_chain_pair_supported": (c_int, [c_int]),
    "icrl_lstm_chain_fwd_seg_persist": (c_int, [c_vp, c_vp] + [c_int] * 6 + [c_vp] * 6),
    "icrl_gru_chain_fwd_seg_persist": (c_int, [c_vp] * 3 + [c_int] * 6 + [c_vp] * 5),
    "icrl_lstm_chain_bwd_seg_persist": (c_int, [c_vp] * 3 + [c_int] * 6 + [c_vp] * 4),
    "icrl_chain_pair_fwd_seg": (c_int, [c_vp] * 5 + [c_int] * 8 + [c_vp] * 9),
    "icrl_chain_pair_fwd_seg_warm16": (c_int, [c_vp] * 5 + [c_int] * 11 + [c_vp] * 9),
    "icrl_chain_warm16_supported": (c_int, [c_int]),
    "icrl_chain_f16x2_supported": (c_int, [c_int]),
    "icrl_chain_pair_fwd_seg_f16x2": (c_int, [c_vp] * 5 + [c_int] * 8 + [c_vp] * 9),
    "icrl_chain_bwd_warm16_supported": (c_int, [c_int]),
    "icrl_lstm_chain_bwd_seg_warm16": (c_int, [c_vp] * 3 + [c_int] * 7 + [c_vp] * 4),
    "icrl_chain_give_up_selftest": (c_int, [c_vp, c_vp, ctypes.c_uint, c_int, c_vp, c_vp]),
    "icrl_chain_pair_replay_check": (c_int, [c_vp] * 7 + [c_int] * 6 + [c_vp] * 4),
    "icrl_chain_pair_replay_check_f16x2": (c_int, [c_vp] * 7 + [c_int] * 6 + [c_vp] * 4),
    "icrl_chain_pair_fwd_rows": (c_int, [c_vp] * 5 + [c_int] * 5 + [c_vp] * 7),
    "icrl_lstm_chain_bwd_rows": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "icrl_gru_chain_fwd_seg": (c_int, [c_vp] * 3 + [c_int] * 6 + [c_vp] * 5),
    "icrl_gru_chain_bwd_seg": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "icrl_lstm_chain_bwd_seg": (c_int, [c_vp] * 3 + [c_int] * 6 + [c_vp] * 4),
    "icrl_gather_rows": (c_int, [c_vp, c_int, c_vp, c_i64, c_int, c_vp, c_vp]),
    "icrl_scatter_add_rows": (c_int, [c_vp, c_int, c_vp, c_i64, c_int, c_vp, c_vp]),
    "icrl_colsum": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_int, c_vp, c_vp, c_vp]),
    "icrl_scale_rows": (c_int, [c_vp, c_i64, c_vp, c_vp, c_i64, c_int, c_int, c_vp]),
    "icrl_colsum_scratch_floats": (c_i64, [c_int, c_int]),
    "icrl_embedding_grad_work_ints": (c_i64, [c_int, c_int]),
    "icrl_token_order": (c_int, [c_vp, c_i64, c_int, c_int, c_vp, c_vp]),
    "icrl_embedding_grad": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_vp]),
    "icrl_lstm_cell_fwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_vp]),
    "icrl_gru_cell_fwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_vp]),
    "icrl_lstm_cell_bwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_vp]),
    "icrl_policy_lstm_bptt": (c_int, [c_vp] * 9 + [c_i64] + [c_int] * 5 + [c_vp]),
    "icrl_policy_lstm_bptt_persist": (c_int, [c_vp] * 4 + [c_int] * 2 + [c_vp] * 5),
    "icrl_policy_lstm_bptt_f16x2": (c_int, [c_vp] * 4 + [c_int] * 2 + [c_vp] * 5),
    "icrl_chain_bwd_f16x2_workspace_floats": (c_i64, [c_int]),
    "icrl_chain_bwd_f16x2_supported": (c_int, [c_int]),
    "icrl_lstm_chain_bwd_seg_f16x2": (c_int, [c_vp] * 3 + [c_int] * 6 + [c_vp] * 4),
    "icrl_policy_lstm_step": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int,
                                      c_vp]),
    "icrl_sample_tokens": (c_int, [c_vp, c_i64, c_int, c_int, c_int, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp,
                                   c_vp]),
    "icrl_sample_tokens_lstm_cell": (c_int, [c_vp, c_i64, c_int, c_int, c_int, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp,
                                             c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "icrl_policy_rollout_rounds": (c_int, [c_vp] * 6 + [c_i64] + [c_int] * 5 + [c_vp] * 10),
    "icrl_policy_dlogits": (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_i64, c_vp, c_vp]),
    "icrl_chain_sync_bytes": (c_i64, []),
    "icrl_lstm_chain_fwd": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "icrl_gru_chain_fwd": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "icrl_gru_chain_bwd": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "icrl_vse_loss": (c_int, [c_vp, c_int, ctypes.c_float, c_vp, c_vp, c_vp, c_vp]),
    "icrl_lstm_chain_bwd": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp]),
    "icrl_chain_fold_calls": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp]),
    "icrl_chain_hprev": (c_int, [c_vp, c_i64, c_int, c_i64, c_vp, c_vp]),
    "icrl_value_head_fwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp]),
    "icrl_cosine_reward": (c_int, [c_vp, c_vp, c_int, c_int, c_vp, c_vp]),
    "icrl_a2c_loss": (c_int, [c_vp, c_vp, c_vp, c_int, c_vp, c_vp, c_vp, c_vp]),
    "icrl_adam_step": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_f64, c_f64, c_f64, c_f64, c_int, c_vp]),
}

SAMPLE, GREEDY, FORCED = 0, 1, 2
EMIT_DLOGITS = 16   # or-ed into SAMPLE / FORCED: the probs output receives onehot(token) - softmax
ABI_VERSION = 4

_lib = None

# When set to a list, gemm() and the chain launchers bracket each launch with HIP events
# recorded on the stream the kernel is launched on and append (kind, work, start, end);
# `work` is the launch's algorithmic FLOPs (GEMM) or dependent steps (chains); a fifth field holds the
# launch's algorithmic bytes (operands read once + result written once) for GEMM launches.
PROFILE = None


class _Timed:
    """steps: launches / in-launch steps a segmented chain call really executes (warm-up included);
    flops: the MFMA work of those steps, when it is not gates * 2 * segments * 512 * 512 per step."""

    def __init__(self, kind, work, nbytes=0.0, steps=0, flops=0.0):
        self.kind, self.work, self.nbytes, self.steps, self.flops = kind, work, nbytes, steps, flops

    def __enter__(self):
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record(_torch_stream())

    def __exit__(self, *exc):
        if PROFILE is not None:
            self.e1.record(_torch_stream())
            PROFILE.append((self.kind, self.work, self.e0, self.e1, self.nbytes, self.steps, self.flops))


def _torch_stream():
    """The stream launches go to, as a torch object (for event records)."""
    if STREAM is None:
        return torch.cuda.current_stream()
    return torch.cuda.ExternalStream(STREAM)


def chain_positions(n, p0, n_calls, full_calls=0):
    """Positions of a chain whose first `full_calls` calls (0: all) are walked in full and whose later calls
    contribute their last column only."""
    fc = n_calls if not full_calls else full_calls
    return n * (sum(p0 + s for s in range(fc)) + n_calls - fc)


def _seg_steps(n, p0, n_calls, full_calls, segments, warmup):
    """Steps a segmented chain call walks: warm-up + positions per segment."""
    return warmup + -(-chain_positions(n, p0, n_calls, full_calls) // segments)


class HipExtensionMissing(RuntimeError):
    pass


def load():
    """Load the shared object and bind every declared symbol (no GPU needed for this)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipExtensionMissing(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the A2C kernels)" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.icrl_abi_version() != ABI_VERSION:
        raise HipExtensionMissing("libicrl_a2c.so has ABI %d, bindings expect %d; rebuild it"
                                  % (lib.icrl_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def _ptr(t):
    if t is None:
        return None
    if isinstance(t, int):
        return t
    assert t.is_cuda, "the HIP kernels take device tensors"
    return t.data_ptr()


# When set (an int: a HIP stream handle), every launch goes to that stream instead of PyTorch's current one.
# Cheaper than entering a torch.cuda.stream() context per launch group for code that alternates between
# streams hundreds of times per minibatch.
STREAM = None


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """Handle of the stream launches go to.  torch.cuda.current_stream() builds a Stream object through several
    Python layers (a few microseconds, on each of a minibatch's ~200 launches); the raw accessors behind it
    return the same handle in a fraction of that."""
    if STREAM is not None:
        return STREAM
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (what, rc, load().icrl_last_error().decode()))


def _chain_shapes(who, n, p0, n_calls, gates, xproj=None, h_emit=None, saved=None, slots=0, d_emit=None, dgates=(),
                  work=None, segments=None, w_hh=None, full_calls=0):
    """Host-side check that every buffer of a chain call is as large as the kernels' indexing assumes (a
    kernel that runs past a buffer faults the GPU, it does not raise)."""
    cols, positions = p0 + n_calls - 1, chain_positions(n, p0, n_calls, full_calls)
    assert 0 <= full_calls <= n_calls, "%s: full_calls must be in [0, n_calls]" % who
    def need(t, count, name):
        if t is not None and not isinstance(t, int):
            assert t.numel() >= count, "%s: %s has %d elements, the chain needs %d" % (who, name, t.numel(), count)
    need(xproj, cols * n * gates * 512, "xproj")
    need(w_hh, gates * 512 * 512, "w_hh")
    need(h_emit, n_calls * n * 512, "h_emit")
    need(d_emit, n_calls * n * 512, "d_emit")
    need(saved, positions * slots * 512, "saved")
    for i, t in enumerate(dgates):
        need(t, positions * gates * 512, "dgates[%d]" % i)
    if work is not None:
        need(work, chain_seg_workspace_floats(segments), "work")


def _f32c(t):
    assert t.dtype == torch.float32 and t.is_contiguous(), (t.dtype, t.is_contiguous())
    return t


def gemm(a, b, out, *, a_kmajor=False, b_kmajor=False, a_rows=None, bias=None, bias2=None, accumulate=False,
         m=None, n=None, k=None, lda=None, ldb=None, ldc=None, tile=0, dtype="f32", a_scale=None, b_scale=None):
    """out[M,N] (+)= A[M,K] @ B[K,N] (+ bias + bias2).  tile: 0 = by shape, 1 = 128x128, 2 = 128x64, 3 = 64x64.
    dtype "bf16": operands rounded to bf16 on chip, products on the bf16 matrix instruction, float32 sums
    (tiles 1 and 3 only).  dtype "f32x3": every element cut into three bf16 pieces on chip, six piece products per
    element product on the bf16 matrix instruction: f32 rounding-level accuracy at about twice the f32 kernel's rate
    on long products (128x128 tiles only; no a_rows, no bias2).  dtype "f16x2": the same with two f16 pieces of the
    elements scaled by a_scale[row of A] / b_scale[row of B] (powers of two from absmax_scales), three piece products.

    a: stored [M, K] (a_kmajor False) or [K, M] (True); b: stored [N, K] (False, an
    nn.Linear weight) or [K, N] (True).  2-D tensors whose last dimension is contiguous;
    row strides become the leading dimensions, so column slices work."""
    assert a.dtype == b.dtype == out.dtype == torch.float32
    assert a.stride(-1) == 1 and b.stride(-1) == 1 and out.stride(-1) == 1
    if m is None:
        m = a.shape[1] if a_kmajor else (a_rows.numel() if a_rows is not None else a.shape[0])
    if k is None:
        k = a.shape[0] if a_kmajor else a.shape[1]
    if n is None:
        n = b.shape[1] if b_kmajor else b.shape[0]
    assert (b.shape[0] if b_kmajor else b.shape[1]) == k, (tuple(a.shape), tuple(b.shape))
    assert out.shape[0] == m and out.shape[1] == n, (tuple(out.shape), m, n)
    lda = a.stride(0) if lda is None else lda
    ldb = b.stride(0) if ldb is None else ldb
    ldc = out.stride(0) if ldc is None else ldc
    if a_rows is not None:
        assert a_rows.dtype == torch.int64 and a_rows.is_contiguous()
    assert dtype in ("f32", "bf16", "f32x3", "f16x2")
    if dtype == "f16x2":
        # two scaled f16 pieces per element, three piece products: a_scale [m] / b_scale [n] from absmax_scales
        assert a_rows is None and bias2 is None
        for t, count in ((a_scale, m), (b_scale, n)):
            assert t is not None and t.dtype == torch.float32 and t.is_contiguous() and t.numel() >= count
        with _Timed("gemm_f16x2s", 2.0 * m * n * k, 4.0 * (m * k + k * n + m * n)):
            rc = load().icrl_gemm_f16x2s(_ptr(a), lda, int(a_kmajor), _ptr(a_scale), _ptr(b), ldb, int(b_kmajor), _ptr(b_scale),
                                         _ptr(out), ldc, m, n, k, _ptr(bias), int(accumulate), _stream())
        _check(rc, "icrl_gemm_f16x2s")
        return out
    if dtype == "bf16":
        fn, kind, tile = load().icrl_gemm_bf16, "gemm_bf16", (1 if tile == 2 else tile)
    elif dtype == "f32x3":
        fn, kind, tile = load().icrl_gemm_f32x3, "gemm_f32x3", 1
    else:
        fn, kind = load().icrl_gemm_f32, "gemm"
    with _Timed(kind, 2.0 * m * n * k, 4.0 * (m * k + k * n + m * n)):
        rc = fn(_ptr(a), lda, int(a_kmajor), _ptr(a_rows), _ptr(b), ldb, int(b_kmajor),
                _ptr(out), ldc, m, n, k, _ptr(bias), _ptr(bias2), int(accumulate), int(tile), _stream())
    _check(rc, "icrl_gemm_" + dtype)
    return out


def f16x2_pieces(rows, k, device):
    """Zeroed piece buffer and descale vector for a [rows, k] operand of gemm_f16x2 (the padding rows up to the
    next multiple of 32 must stay zero; cut_f16x2 never writes them)."""
    halves = load().icrl_f16x2_piece_halves(rows, k)
    assert halves > 0, "gemm_f16x2 operands need k % 16 == 0"
    return (torch.zeros(halves, dtype=torch.float16, device=device), torch.zeros(rows, dtype=torch.float32, device=device))


def cut_f16x2(x, pieces, descale, fixed_scale=0.0):
    """Cuts the rows of x [rows, k] (float32, last dimension contiguous) into two scaled f16 pieces in the matrix
    instruction's fragment order; descale[r] = 1 / the row's power-of-two scale (its own with fixed_scale 0)."""
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    rows, k = x.shape
    assert pieces.dtype == torch.float16 and pieces.is_contiguous() and pieces.numel() >= load().icrl_f16x2_piece_halves(rows, k) > 0
    assert descale.dtype == torch.float32 and descale.is_contiguous() and descale.numel() >= rows
    _check(load().icrl_cut_f16x2(_ptr(x), x.stride(0), rows, k, float(fixed_scale), _ptr(pieces), _ptr(descale), _stream()),
           "icrl_cut_f16x2")


def gemm_f16x2(a_pieces, a_descale, b_pieces, b_descale, out, k, bias=None):
    """out[M, N] = A . B^T (+ bias) from operands cut by cut_f16x2 (three f16 piece products per element product,
    float32 sums: float32 rounding-level accuracy)."""
    m, n = out.shape
    assert out.dtype == torch.float32 and out.stride(1) == 1
    for t, rows in ((a_pieces, m), (b_pieces, n)):
        assert t.dtype == torch.float16 and t.is_contiguous() and t.numel() >= load().icrl_f16x2_piece_halves(rows, k) > 0
    assert a_descale.numel() >= m and b_descale.numel() >= n and a_descale.dtype == b_descale.dtype == torch.float32
    assert bias is None or (bias.dtype == torch.float32 and bias.is_contiguous() and bias.numel() >= n)
    with _Timed("gemm_f16x2", 2.0 * m * n * k, 4.0 * (m * k + k * n + m * n)):
        rc = load().icrl_gemm_f16x2(_ptr(a_pieces), _ptr(a_descale), _ptr(b_pieces), _ptr(b_descale), _ptr(out),
                                    out.stride(0), m, n, k, _ptr(bias), _stream())
    _check(rc, "icrl_gemm_f16x2")
    return out


def gather_rows(table, idx, idx_stride, rows, out):
    _check(load().icrl_gather_rows(_ptr(_f32c(table)), table.shape[1], _ptr(idx), idx_stride, rows, _ptr(out),
                                   _stream()), "icrl_gather_rows")
    return out


def scatter_add_rows(grad_table, idx, idx_stride, rows, rows_in):
    _check(load().icrl_scatter_add_rows(_ptr(_f32c(grad_table)), grad_table.shape[1], _ptr(idx), idx_stride, rows,
                                        _ptr(_f32c(rows_in)), _stream()), "icrl_scatter_add_rows")


def colsum(x, out, accumulate=False, rows=None, scratch=None, row_scale=None):
    """out[c] (+)= sum_r row_scale[r] * x[r, c] (row_scale None = ones).  scratch: a float32 buffer of at least
    colsum_scratch_floats(rows, cols); with it the result is reproducible bit for bit (fixed summation order),
    without it float atomics are used."""
    rows = x.shape[0] if rows is None else rows
    if scratch is not None:
        assert scratch.dtype == torch.float32 and scratch.numel() >= colsum_scratch_floats(rows, x.shape[1])
    if row_scale is not None:
        assert row_scale.dtype == torch.float32 and row_scale.is_contiguous() and row_scale.numel() >= rows
    _check(load().icrl_colsum(_ptr(x), x.stride(0), rows, x.shape[1], _ptr(out), int(accumulate), _ptr(row_scale),
                              _ptr(scratch), _stream()), "icrl_colsum")
    return out


def scale_rows(x, coef, out):
    """out[r, :] = coef[r] * x[r, :] for 2-D x / out whose rows are contiguous."""
    rows, cols = x.shape
    assert out.shape == x.shape and x.stride(1) == 1 and out.stride(1) == 1 and coef.numel() >= rows
    _check(load().icrl_scale_rows(_ptr(x), x.stride(0), _ptr(_f32c(coef)), _ptr(out), out.stride(0), rows, cols,
                                  _stream()), "icrl_scale_rows")
    return out


def colsum_scratch_floats(rows, cols):
    return int(load().icrl_colsum_scratch_floats(int(rows), int(cols)))


def token_order(idx, idx_stride, rows, vocab, work):
    """For every token the rows that hold it, in ascending order, into work (int32,
    embedding_grad_work_ints(vocab, rows)): what embedding_grad sums by."""
    assert idx.dtype == torch.int64 and work.dtype == torch.int32 and work.is_contiguous()
    assert work.numel() >= embedding_grad_work_ints(vocab, rows) and idx.numel() >= (rows - 1) * idx_stride + 1
    _check(load().icrl_token_order(_ptr(idx), idx_stride, rows, vocab, _ptr(work), _stream()), "icrl_token_order")


def embedding_grad(grad_table, rows, rows_in, work):
    """grad_table[v] = sum of rows_in[r] over the rows token_order listed for v, in ascending r (reproducible);
    the whole table is overwritten."""
    vocab, width = grad_table.shape
    assert work.dtype == torch.int32 and work.numel() >= embedding_grad_work_ints(vocab, rows)
    assert rows_in.numel() >= rows * width
    _check(load().icrl_embedding_grad(_ptr(_f32c(grad_table)), vocab, width, _ptr(_f32c(rows_in)), _ptr(work), _stream()),
           "icrl_embedding_grad")


def embedding_grad_work_ints(vocab, rows):
    return int(load().icrl_embedding_grad_work_ints(int(vocab), int(rows)))


def gru_cell_fwd(gx, gh, h_prev, h):
    _check(load().icrl_gru_cell_fwd(_ptr(_f32c(gx)), _ptr(_f32c(gh)), _ptr(_f32c(h_prev)), _ptr(_f32c(h)),
                                    gx.shape[0], _stream()), "icrl_gru_cell_fwd")


def lstm_cell_fwd(gates, c_prev, c, h):
    _check(load().icrl_lstm_cell_fwd(_ptr(_f32c(gates)), _ptr(_f32c(c_prev)), _ptr(_f32c(c)), _ptr(_f32c(h)),
                                     gates.shape[0], _stream()), "icrl_lstm_cell_fwd")


def lstm_cell_bwd(gates_act, c_prev, c, dh, dc, dgates, dc_prev):
    _check(load().icrl_lstm_cell_bwd(_ptr(_f32c(gates_act)), _ptr(_f32c(c_prev)), _ptr(_f32c(c)), _ptr(_f32c(dh)),
                                     _ptr(dc), _ptr(_f32c(dgates)), _ptr(_f32c(dc_prev)), gates_act.shape[0],
                                     _stream()), "icrl_lstm_cell_bwd")


def sample_tokens(logits, mode, tokens_out, token_stride, *, uniforms=None, log_prob=None, log_prob_stride=1,
                  probs=None, entropy=None):
    n, v = logits.shape
    assert logits.dtype == torch.float32 and logits.stride(1) == 1
    assert tokens_out.dtype == torch.int64
    if uniforms is not None:
        assert uniforms.dtype == torch.float64 and uniforms.is_contiguous() and uniforms.numel() >= n
    _check(load().icrl_sample_tokens(_ptr(logits), logits.stride(0), n, v, mode, _ptr(uniforms), _ptr(tokens_out),
                                     token_stride, _ptr(log_prob), log_prob_stride, _ptr(probs), _ptr(entropy),
                                     _stream()), "icrl_sample_tokens")


def sample_tokens_lstm_cell(logits, mode, tokens_out, token_stride, pre, table, gates_out, c_prev, c_out, h_out, *,
                            uniforms=None, log_prob=None, log_prob_stride=1, probs=None, entropy=None, h_pieces=None):
    """sample_tokens, then the policy LSTM's next cell update in the same launch: gates = pre[i] + table[token_i].
    logits [N,V] and pre [N,2048] may be column ranges of one wider matrix (row strides are passed).  h_pieces:
    a gemm_f16x2 piece buffer for [N, 512] that receives the new h cut at scale 4096."""
    n, v = logits.shape
    assert logits.dtype == torch.float32 and logits.stride(1) == 1 and tokens_out.dtype == torch.int64
    assert pre.dtype == torch.float32 and pre.shape == (n, 2048) and pre.stride(1) == 1
    assert table.shape == (v, 2048) and table.is_contiguous() and table.dtype == torch.float32
    for t, width in ((gates_out, 2048), (c_prev, 512), (c_out, 512), (h_out, 512)):
        assert t.dtype == torch.float32 and t.shape == (n, width) and t.is_contiguous()
    if uniforms is not None:
        assert uniforms.dtype == torch.float64 and uniforms.is_contiguous() and uniforms.numel() >= n
    if h_pieces is not None:
        assert h_pieces.dtype == torch.float16 and h_pieces.is_contiguous()
        assert h_pieces.numel() >= load().icrl_f16x2_piece_halves(n, 512)
    _check(load().icrl_sample_tokens_lstm_cell(_ptr(logits), logits.stride(0), n, v, mode, _ptr(uniforms), _ptr(tokens_out),
                                               token_stride, _ptr(log_prob), log_prob_stride, _ptr(probs), _ptr(entropy),
                                               _ptr(pre), pre.stride(0), _ptr(table), _ptr(gates_out), _ptr(c_prev),
                                               _ptr(c_out), _ptr(h_out), _ptr(h_pieces), _stream()),
           "icrl_sample_tokens_lstm_cell")


def policy_rollout_rounds(w_pieces, w_descale, bias, h_pieces, h_descale, ext, n, v, pre_col, rounds, mode, tokens, log_prob,
                          table, gates, c_state, h_state, *, uniforms=None, probs=None, entropy=None):
    """`rounds` rollout rounds (gemm_f16x2 then sample_tokens_lstm_cell each) queued by one call into the library.
    tokens [rounds, n] int64, log_prob [rounds, n], probs [rounds, n, v], entropy [rounds, n], gates [rounds, n, 2048],
    c_state [rounds + 1, n, 512], h_state [rounds, n, 512]: contiguous slices starting at the first of these rounds."""
    assert ext.dtype == torch.float32 and ext.stride(1) == 1 and ext.shape[0] >= n and ext.shape[1] >= pre_col + 2048
    assert tokens.dtype == torch.int64 and tokens.is_contiguous() and tokens.numel() >= rounds * n
    for t, count in ((log_prob, rounds * n), (gates, rounds * n * 2048), (c_state, (rounds + 1) * n * 512),
                     (h_state, rounds * n * 512), (probs, rounds * n * v), (entropy, rounds * n)):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.numel() >= count)
    assert uniforms is None or (uniforms.dtype == torch.float64 and uniforms.is_contiguous() and uniforms.numel() >= rounds * n)
    assert table.shape == (v, 2048) and table.is_contiguous() and bias.numel() >= pre_col + 2048
    assert w_pieces.numel() >= load().icrl_f16x2_piece_halves(pre_col + 2048, 512) and w_descale.numel() >= pre_col + 2048
    assert h_pieces.numel() >= load().icrl_f16x2_piece_halves(n, 512) and h_descale.numel() >= n
    _check(load().icrl_policy_rollout_rounds(_ptr(w_pieces), _ptr(w_descale), _ptr(bias), _ptr(h_pieces), _ptr(h_descale),
                                             _ptr(ext), ext.stride(0), n, v, pre_col, rounds, mode, _ptr(uniforms),
                                             _ptr(tokens), _ptr(log_prob), _ptr(probs), _ptr(entropy), _ptr(table),
                                             _ptr(gates), _ptr(c_state), _ptr(h_state), _stream()),
           "icrl_policy_rollout_rounds")


def masked_nll(log_probs, valid_steps, row_weight, loss, d_logp=None):
    """log_probs [S, N] float32; valid_steps [N] int32; row_weight [N] float32; loss: 1-element view."""
    s_, n = log_probs.shape
    assert valid_steps.dtype == torch.int32 and row_weight.dtype == torch.float32
    _check(load().icrl_masked_nll(_ptr(_f32c(log_probs)), _ptr(valid_steps), _ptr(row_weight), s_, n, _ptr(loss),
                                  _ptr(d_logp), _stream()), "icrl_masked_nll")


def mse_loss(values, targets, loss, d_value=None):
    _check(load().icrl_mse_loss(_ptr(_f32c(values)), _ptr(_f32c(targets)), values.numel(), _ptr(loss), _ptr(d_value),
                                _stream()), "icrl_mse_loss")


def topk_rows(logits, k):
    """(values [N,k] float32, indices [N,k] int64) of the k largest entries per row, descending."""
    n, v = logits.shape
    assert logits.dtype == torch.float32 and logits.stride(1) == 1
    values = torch.empty(n, k, dtype=torch.float32, device=logits.device)
    indices = torch.empty(n, k, dtype=torch.int64, device=logits.device)
    _check(load().icrl_topk_rows(_ptr(logits), logits.stride(0), n, v, k, _ptr(values), _ptr(indices), _stream()),
           "icrl_topk_rows")
    return values, indices


def policy_dlogits(probs, dlogits, tokens, token_stride, coef):
    """dlogits[r, k] = coef[r] * ((k == tokens[r]) - probs[r, k]); coef None = ones."""
    rows, v = probs.shape
    _check(load().icrl_policy_dlogits(_ptr(_f32c(probs)), _ptr(_f32c(dlogits)), rows, v, _ptr(tokens), token_stride,
                                      _ptr(None if coef is None else _f32c(coef)), _stream()), "icrl_policy_dlogits")


def chain_sync_bytes():
    return int(load().icrl_chain_sync_bytes())


def lstm_chain_fwd(xproj, w_hh, n, p0, n_calls, h_emit, saved, sync_ws, status, state_in=None, state_out=None):
    """p0 = caption columns seen by the first call; call s covers p0 + s columns."""
    _chain_shapes("lstm_chain_fwd", n, p0, n_calls, 4, xproj=xproj, w_hh=w_hh, h_emit=h_emit, saved=saved, slots=6)
    with _Timed("lstm_chain_fwd", chain_positions(n, p0, n_calls), steps=chain_positions(n, p0, n_calls)):
        rc = load().icrl_lstm_chain_fwd(_ptr(_f32c(xproj)), _ptr(_f32c(w_hh)), n, p0, n_calls, _ptr(_f32c(h_emit)),
                                        _ptr(saved), _ptr(state_in), _ptr(state_out), _ptr(sync_ws), _ptr(status),
                                        _stream())
    _check(rc, "icrl_lstm_chain_fwd")


def gru_chain_fwd(xproj, w_hh, b_hh, n, p0, n_calls, h_emit, sync_ws, status, state_in=None, state_out=None,
                  saved=None):
    _chain_shapes("gru_chain_fwd", n, p0, n_calls, 3, xproj=xproj, w_hh=w_hh, h_emit=h_emit, saved=saved, slots=5)
    with _Timed("gru_chain_fwd", chain_positions(n, p0, n_calls), steps=chain_positions(n, p0, n_calls)):
        rc = load().icrl_gru_chain_fwd(_ptr(_f32c(xproj)), _ptr(_f32c(w_hh)), _ptr(_f32c(b_hh)), n, p0, n_calls,
                                       _ptr(_f32c(h_emit)), _ptr(saved), _ptr(state_in), _ptr(state_out),
                                       _ptr(sync_ws), _ptr(status), _stream())
    _check(rc, "icrl_gru_chain_fwd")


def gru_chain_bwd(w_hh, saved, d_emit, n, p0, n_calls, dgates_x, dgates_h, sync_ws, status):
    _chain_shapes("gru_chain_bwd", n, p0, n_calls, 3, w_hh=w_hh, saved=saved, slots=5, d_emit=d_emit,
                  dgates=(dgates_x, dgates_h))
    _check(load().icrl_gru_chain_bwd(_ptr(_f32c(w_hh)), _ptr(_f32c(saved)), _ptr(_f32c(d_emit)), n, p0, n_calls,
                                     _ptr(_f32c(dgates_x)), _ptr(_f32c(dgates_h)), _ptr(sync_ws), _ptr(status),
                                     _stream()), "icrl_gru_chain_bwd")


def vse_loss(s_matrix, beta, row_scratch, loss, d_s=None):
    """s_matrix [N, N] = visuals @ semantics^T; loss: 1-element view; d_s: optional [N, N]."""
    n = s_matrix.shape[0]
    assert s_matrix.shape == (n, n) and s_matrix.is_contiguous()
    _check(load().icrl_vse_loss(_ptr(_f32c(s_matrix)), n, float(beta), _ptr(row_scratch), _ptr(loss), _ptr(d_s),
                                _stream()), "icrl_vse_loss")


def lstm_chain_bwd(w_hh, saved, d_emit, n, p0, n_calls, dgates, sync_ws, status):
    _chain_shapes("lstm_chain_bwd", n, p0, n_calls, 4, w_hh=w_hh, saved=saved, slots=6, d_emit=d_emit, dgates=(dgates,))
    with _Timed("lstm_chain_bwd", chain_positions(n, p0, n_calls), steps=chain_positions(n, p0, n_calls)):
        rc = load().icrl_lstm_chain_bwd(_ptr(_f32c(w_hh)), _ptr(_f32c(saved)), _ptr(_f32c(d_emit)), n, p0, n_calls,
                                        _ptr(_f32c(dgates)), _ptr(sync_ws), _ptr(status), _stream())
    _check(rc, "icrl_lstm_chain_bwd")


def chain_fold_calls(dgates, n, p0, n_calls, width, out, full_calls=0):
    assert dgates.numel() >= chain_positions(n, p0, n_calls, full_calls) * width
    assert out.numel() >= (p0 + n_calls - 1) * n * width
    _check(load().icrl_chain_fold_calls(_ptr(_f32c(dgates)), n, p0, n_calls, full_calls, width, _ptr(_f32c(out)),
                                        _stream()), "icrl_chain_fold_calls")


def chain_hprev(saved, positions, h_prev, gru=False, stride=1):
    _check(load().icrl_chain_hprev(_ptr(_f32c(saved)), positions, int(gru), int(stride), _ptr(_f32c(h_prev)),
                                   _stream()), "icrl_chain_hprev")


def value_head_fwd(yh_inout, u, w2, b2, rows, n, values):
    _check(load().icrl_value_head_fwd(_ptr(_f32c(yh_inout)), _ptr(_f32c(u)), _ptr(_f32c(w2)), _ptr(_f32c(b2)), rows,
                                      n, _ptr(_f32c(values)), _stream()), "icrl_value_head_fwd")


def cosine_reward(ve, se, rows, n, reward):
    _check(load().icrl_cosine_reward(_ptr(_f32c(ve)), _ptr(_f32c(se)), rows, n, _ptr(_f32c(reward)), _stream()),
           "icrl_cosine_reward")


def a2c_loss(values, rewards, log_probs, scalars, d_value=None, d_logp=None):
    _check(load().icrl_a2c_loss(_ptr(_f32c(values)), _ptr(_f32c(rewards)), _ptr(_f32c(log_probs)), values.numel(),
                                _ptr(_f32c(scalars)), _ptr(d_value), _ptr(d_logp), _stream()), "icrl_a2c_loss")


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step):
    _check(load().icrl_adam_step(_ptr(_f32c(param)), _ptr(_f32c(grad)), _ptr(_f32c(exp_avg)),
                                 _ptr(_f32c(exp_avg_sq)), param.numel(), lr, beta1, beta2, eps, step, _stream()),
           "icrl_adam_step")


def gemm_splitk(a, b, slabs, splits, *, a_kmajor=False, b_kmajor=False, tile=0, dtype="f32", a_scale=None, b_scale=None):
    """slabs[z] = partial product of split z; slabs is [splits, M, N] contiguous."""
    m = a.shape[1] if a_kmajor else a.shape[0]
    k = a.shape[0] if a_kmajor else a.shape[1]
    n = b.shape[1] if b_kmajor else b.shape[0]
    assert slabs.shape == (splits, m, n) and slabs.is_contiguous()
    if dtype == "f16x2":
        for t, count in ((a_scale, m), (b_scale, n)):
            assert t is not None and t.dtype == torch.float32 and t.is_contiguous() and t.numel() >= count
        with _Timed("gemm_f16x2s", 2.0 * m * n * k, 4.0 * (m * k + k * n + splits * m * n)):
            rc = load().icrl_gemm_f16x2s_splitk_tile(_ptr(a), a.stride(0), int(a_kmajor), _ptr(a_scale), _ptr(b), b.stride(0),
                                                     int(b_kmajor), _ptr(b_scale), _ptr(slabs), n, m * n, m, n, k, splits,
                                                     int(tile), _stream())
        _check(rc, "icrl_gemm_f16x2s_splitk")
        return
    if dtype == "bf16":
        fn, kind, tile = load().icrl_gemm_bf16_splitk, "gemm_bf16", (1 if tile == 2 else tile)
    elif dtype == "f32x3":
        fn, kind, tile = load().icrl_gemm_f32x3_splitk, "gemm_f32x3", 1
    else:
        fn, kind = load().icrl_gemm_f32_splitk, "gemm"
    with _Timed(kind, 2.0 * m * n * k, 4.0 * (m * k + k * n + splits * m * n)):
        rc = fn(_ptr(a), a.stride(0), int(a_kmajor), _ptr(b), b.stride(0), int(b_kmajor),
                _ptr(slabs), n, m * n, m, n, k, splits, int(tile), _stream())
    _check(rc, "icrl_gemm_%s_splitk" % dtype)


def sum_slabs(slabs, out, accumulate=False):
    splits = slabs.shape[0]
    count = slabs[0].numel()
    assert out.is_contiguous() and out.numel() == count
    _check(load().icrl_sum_slabs(_ptr(_f32c(slabs)), count, splits, count, _ptr(out), int(accumulate), _stream()),
           "icrl_sum_slabs")


_PLANS = {}


def plan_gemm(m, n, k, out_contiguous=True, shared=False):
    key = (m, n, k, out_contiguous, shared)
    plan = _PLANS.get(key)
    if plan is None:
        plan = _PLANS[key] = _plan_gemm(m, n, k, out_contiguous, shared)
    return plan


def _plan_gemm(m, n, k, out_contiguous=True, shared=False):
    """(tile, splits) for an [m,n,k] product, from measurements on MI355X (tools/gemm_sweep.py).
    shared: the launch will run beside a persistent chain kernel that holds half of every CU's LDS and
    registers; two 64x64-tile workgroups still fit beside it on a CU, one of the larger tiles at most, so
    the 128x128 choice below, faster on an empty device, is not made (measured on the whole minibatch).
    tile codes: 1 = 128x128, 2 = 128x64, 3 = 64x64.  Short k loops: 64x64 tiles.  Small outputs: 64x64
    tiles, k split so that about a thousand workgroups exist.  Otherwise 128x128 tiles with the split
    that best fills the 256 CUs: a CU's last, partly filled turn costs as much as a full one, few
    workgroups per CU leave pipeline fill and drain exposed, and every split costs a partial slab."""
    t64 = ((m + 63) // 64) * ((n + 63) // 64)
    if k <= 1024:
        return 3, 1
    if not out_contiguous:
        return (3 if t64 <= 512 else 2), 1
    t128 = ((m + 127) // 128) * ((n + 127) // 128)
    if shared and t64 > 512:
        return 3, (2 if k >= 2048 else 1)
    if t128 < 32 or shared:
        want = (2048 if k >= 32768 else 1024) // t64
        splits = max(1, min(k // 512, want, 16))
        return (1 if (k >= 32768 and splits > 1) else 3), splits
    ideal_us = 2.0 * m * n * k / 110e6
    best = None
    for splits in (1, 2, 3, 4, 5, 6, 8, 12, 16):   # the measured set
        if splits > 1 and k // splits < 384:
            break
        per_cu = t128 * splits / 256.0
        balance = per_cu / -(-per_cu // 1)
        est = ideal_us / (balance * per_cu / (per_cu + 0.5))
        if splits > 1:
            est += splits * m * n * 8.0 / 7e6
        if best is None or est < best[0]:
            best = (est, splits)
    return 1, best[1]


def gemm_auto(a, b, out, scratch, *, a_kmajor=False, b_kmajor=False, accumulate=False, bias=None, shared=False,
              dtype="f32", a_scale=None, b_scale=None, big_tile=False):
    """gemm() with the tile / split-K choice of plan_gemm(); `scratch` is a flat float32 buffer for
    the split-K partial slabs (summed in a fixed order, so results are reproducible)."""
    m = a.shape[1] if a_kmajor else a.shape[0]
    k = a.shape[0] if a_kmajor else a.shape[1]
    n = b.shape[1] if b_kmajor else b.shape[0]
    contiguous = out.is_contiguous() and bias is None
    tile, splits = plan_gemm(m, n, k, contiguous, shared)
    if dtype == "bf16" and contiguous and k >= 2048 and m * n >= (1 << 20):
        # the bf16 kernel's plans, from its own sweep: long k over large outputs in 128x128 tiles with k in four,
        # long k over small outputs in 64x64 tiles with k in two, k ~ 2000 unsplit in 64x64 tiles
        tile, splits = ((1, 4) if m * n >= (1 << 21) else (3, 2)) if k >= 4096 else (3, 1)
    if dtype in ("f32x3", "f16x2"):
        # 128x128 tiles, 36 KB of LDS each: two workgroups fit a CU beside a persistent chain workgroup and the kernel
        # runs as fast at two or three per CU as at four, so the split is the one whose workgroups fill whole turns
        # of 512 places best (measured: k in five for 304 tiles, in eight for 316 and for 64)
        tiles = ((m + 127) // 128) * ((n + 127) // 128)
        fits = [sp for sp in range(1, 9) if sp == 1 or (contiguous and k // sp >= 512)]
        tile, splits = 1, min(fits, key=lambda sp: (-(-tiles * sp // 512) / (tiles * sp / 512.0), -sp))
    if big_tile and dtype == "f16x2" and contiguous and scratch is not None and m >= 256 and n >= 256:
        # 256x256 tiles, one 512-thread workgroup per CU: the split whose workgroups fill whole turns of 256 places
        # best, the larger one on a tie (always through the slab path: the large tile has no direct-store entry)
        tiles = ((m + 255) // 256) * ((n + 255) // 256)
        fits = [sp for sp in range(1, 9) if (sp == 1 or k // sp >= 512) and sp * m * n <= scratch.numel()]
        if fits:
            splits = min(fits, key=lambda sp: (-(-tiles * sp // 256) / (tiles * sp / 256.0), -sp))
            slabs = scratch[:splits * m * n].view(splits, m, n)
            gemm_splitk(a, b, slabs, splits, a_kmajor=a_kmajor, b_kmajor=b_kmajor, tile=2, dtype=dtype, a_scale=a_scale,
                        b_scale=b_scale)
            sum_slabs(slabs, out, accumulate=accumulate)
            return out
    if splits > 1:  # no more partial slabs than the scratch buffer holds
        splits = min(splits, (scratch.numel() // (m * n)) if scratch is not None else 1)
    if splits > 1:
        slabs = scratch[:splits * m * n].view(splits, m, n)
        gemm_splitk(a, b, slabs, splits, a_kmajor=a_kmajor, b_kmajor=b_kmajor, tile=tile, dtype=dtype, a_scale=a_scale,
                    b_scale=b_scale)
        sum_slabs(slabs, out, accumulate=accumulate)
        return out
    return gemm(a, b, out, a_kmajor=a_kmajor, b_kmajor=b_kmajor, accumulate=accumulate, bias=bias, tile=tile, dtype=dtype,
                a_scale=a_scale, b_scale=b_scale)


def absmax_scales(x, work, row_scale=None, col_scale=None):
    """Power-of-two scales for gemm(dtype="f16x2"): row_scale[r] / col_scale[c] brings the largest magnitude of row r /
    column c of x [rows, cols] (float32, last dimension contiguous) into [2^14, 2^15).  work: int32, rows + cols."""
    rows, cols = x.shape
    assert x.dtype == torch.float32 and x.stride(1) == 1 and work.dtype == torch.int32 and work.numel() >= rows + cols
    for t, count in ((row_scale, rows), (col_scale, cols)):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.numel() >= count)
    _check(load().icrl_absmax_scales(_ptr(x), x.stride(0), rows, cols, _ptr(row_scale), _ptr(col_scale), _ptr(work), _stream()),
           "icrl_absmax_scales")


def g0_row_scales(log_prob, scale):
    """Scales of the rows of G0 = onehot(token) - softmax (largest magnitude 1 - p_token) from their log-probabilities."""
    assert log_prob.dtype == scale.dtype == torch.float32 and log_prob.is_contiguous() and scale.numel() >= log_prob.numel()
    _check(load().icrl_g0_row_scales(_ptr(log_prob), log_prob.numel(), _ptr(scale), _stream()), "icrl_g0_row_scales")


def chain_seg_workspace_floats(segments):
    return int(load().icrl_chain_seg_workspace_floats(segments))


def lstm_chain_fwd_seg(xproj, w_hh, n, p0, n_calls, segments, warmup, h_emit, saved, work, seam_err, state_in=None,
                       persistent=False, full_calls=0):
    """state_in: {h, c} (1024 floats) the chain starts from; None = zeros.  persistent: all steps in one
    launch (needs chain_pair_supported(segments))."""
    _chain_shapes("lstm_chain_fwd_seg", n, p0, n_calls, 4, xproj=xproj, w_hh=w_hh, h_emit=h_emit, saved=saved, slots=6,
                  work=work, segments=segments, full_calls=full_calls)
    fn = load().icrl_lstm_chain_fwd_seg_persist if persistent else load().icrl_lstm_chain_fwd_seg
    with _Timed("lstm_chain_fwd", chain_positions(n, p0, n_calls, full_calls), steps=_seg_steps(n, p0, n_calls, full_calls, segments, warmup)):
        rc = fn(_ptr(_f32c(xproj)), _ptr(_f32c(w_hh)), n, p0, n_calls, full_calls, segments, warmup,
                                            _ptr(_f32c(h_emit)), _ptr(saved), _ptr(state_in), _ptr(_f32c(work)),
                                            _ptr(seam_err), _stream())
    _check(rc, "icrl_lstm_chain_fwd_seg")


def lstm_chain_fwd_seg_static(xproj, w_hh, n, p0, n_calls, segments, warmup, h_emit, saved, seam_err, state_in=None,
                              full_calls=0):
    """lstm_chain_fwd_seg on the library's static workspace (at most 128 segments): no allocation, no
    workspace argument.  Calls on one GPU share that workspace and must not overlap."""
    _chain_shapes("lstm_chain_fwd_seg_static", n, p0, n_calls, 4, xproj=xproj, w_hh=w_hh, h_emit=h_emit, saved=saved,
                  slots=6, full_calls=full_calls)
    rc = load().icrl_lstm_chain_fwd_seg_static(_ptr(_f32c(xproj)), _ptr(_f32c(w_hh)), n, p0, n_calls, full_calls, segments, warmup,
                                               _ptr(_f32c(h_emit)), _ptr(saved), _ptr(state_in), _ptr(seam_err),
                                               _stream())
    _check(rc, "icrl_lstm_chain_fwd_seg_static")


def gru_chain_fwd_seg(xproj, w_hh, b_hh, n, p0, n_calls, segments, warmup, h_emit, work, seam_err, saved=None,
                      persistent=False, full_calls=0):
    """persistent: all steps in one launch (needs chain_pair_supported(segments))."""
    _chain_shapes("gru_chain_fwd_seg", n, p0, n_calls, 3, xproj=xproj, w_hh=w_hh, h_emit=h_emit, saved=saved, slots=5,
                  work=work, segments=segments, full_calls=full_calls)
    fn = load().icrl_gru_chain_fwd_seg_persist if persistent else load().icrl_gru_chain_fwd_seg
    with _Timed("gru_chain_fwd", chain_positions(n, p0, n_calls, full_calls), steps=_seg_steps(n, p0, n_calls, full_calls, segments, warmup)):
        rc = fn(_ptr(_f32c(xproj)), _ptr(_f32c(w_hh)), _ptr(_f32c(b_hh)), n, p0, n_calls, full_calls,
                                           segments, warmup, _ptr(_f32c(h_emit)), _ptr(saved), _ptr(_f32c(work)),
                                           _ptr(seam_err), _stream())
    _check(rc, "icrl_gru_chain_fwd_seg")


def gru_chain_bwd_seg(w_hh, saved, d_emit, n, p0, n_calls, segments, warmup, dgates_x, dgates_h, work, seam_err):
    _chain_shapes("gru_chain_bwd_seg", n, p0, n_calls, 3, w_hh=w_hh, saved=saved, slots=5, d_emit=d_emit,
                  dgates=(dgates_x, dgates_h), work=work, segments=segments)
    _check(load().icrl_gru_chain_bwd_seg(_ptr(_f32c(w_hh)), _ptr(_f32c(saved)), _ptr(_f32c(d_emit)), n, p0, n_calls,
                                         segments, warmup, _ptr(_f32c(dgates_x)), _ptr(_f32c(dgates_h)),
                                         _ptr(_f32c(work)), _ptr(seam_err), _stream()), "icrl_gru_chain_bwd_seg")


def chain_pair_supported(segments):
    return bool(load().icrl_chain_pair_supported(int(segments)))


def chain_warm16_supported(segments):
    return bool(load().icrl_chain_warm16_supported(int(segments)))


def chain_f16x2_supported(segments):
    return bool(load().icrl_chain_f16x2_supported(int(segments)))


def chain_pair_fwd_seg(xproj_v, w_hh_v, xproj_r, w_hh_r, b_hh_r, n, p0_v, p0_r, n_calls, segments, warmup, h_emit_v,
                       saved_v, h_emit_r, work_v, work_r, seam_err_v, seam_err_r, warmup_r=None, full_calls=0,
                       saved_r=None, cheap=(0, 0), phase=0, arith="f32"):
    """Value LSTM chain and reward GRU chain, every step, in one persistent launch (warmup_r: the
    reward chain's own warm-up length; default: the value chain's).  cheap = (value, reward): the first steps of
    each warm-up that run at 16-bit split precision in a launch of their own (icrl_chain_pair_fwd_seg_warm16);
    phase 1 queues that launch alone, phase 2 (same arguments) the rest.  arith="f16x2": every step of both chains
    on the f16 matrix instruction at float32-level accuracy (icrl_chain_pair_fwd_seg_f16x2; no cheap part)."""
    _chain_shapes("chain_pair_fwd_seg (value)", n, p0_v, n_calls, 4, xproj=xproj_v, w_hh=w_hh_v, h_emit=h_emit_v,
                  saved=saved_v, slots=6, work=work_v, segments=segments, full_calls=full_calls)
    _chain_shapes("chain_pair_fwd_seg (reward)", n, p0_r, n_calls, 3, xproj=xproj_r, w_hh=w_hh_r, h_emit=h_emit_r,
                  saved=saved_r, slots=5, work=work_r, segments=segments, full_calls=full_calls)
    warmup_r = warmup if warmup_r is None else warmup_r
    steps_v = _seg_steps(n, p0_v, n_calls, full_calls, segments, warmup)
    steps_r = _seg_steps(n, p0_r, n_calls, full_calls, segments, warmup_r)
    positions = chain_positions(n, p0_v, n_calls, full_calls) + chain_positions(n, p0_r, n_calls, full_calls)
    if arith == "f16x2":
        assert phase == 0 and not (cheap[0] or cheap[1])
        timed = _Timed("chain_pair_fwd_f16x2", positions, steps=max(steps_v, steps_r),
                       flops=2.0 * segments * 512 * 512 * (4 * steps_v + 3 * steps_r))
    elif phase == 1:    # the 16-bit launch alone: no float32 matrix-instruction work
        timed = _Timed("chain_warm16", 0, steps=max(cheap))
    else:
        # float32 matrix-instruction work: the steps that do not run at 16-bit split precision; a phase-2 call
        # walks only those
        walked = max(steps_v - cheap[0], steps_r - cheap[1]) if phase == 2 else max(steps_v, steps_r)
        timed = _Timed("chain_pair_fwd", positions, steps=walked,
                       flops=2.0 * segments * 512 * 512 * (4 * (steps_v - cheap[0]) + 3 * (steps_r - cheap[1])))
    with timed:
        head = (_ptr(_f32c(xproj_v)), _ptr(_f32c(w_hh_v)), _ptr(_f32c(xproj_r)), _ptr(_f32c(w_hh_r)), _ptr(_f32c(b_hh_r)),
                n, p0_v, p0_r, n_calls, full_calls, segments, warmup, warmup_r)
        tail = (_ptr(_f32c(h_emit_v)), _ptr(saved_v), _ptr(_f32c(h_emit_r)), _ptr(saved_r), _ptr(_f32c(work_v)),
                _ptr(_f32c(work_r)), _ptr(seam_err_v), _ptr(seam_err_r), _stream())
        if arith == "f16x2":
            rc = load().icrl_chain_pair_fwd_seg_f16x2(*head, *tail)
        elif cheap[0] or cheap[1]:
            rc = load().icrl_chain_pair_fwd_seg_warm16(*head, int(cheap[0]), int(cheap[1]), int(phase), *tail)
        else:
            assert phase == 0
            rc = load().icrl_chain_pair_fwd_seg(*head, *tail)
    _check(rc, "icrl_chain_pair_fwd_seg")


def chain_pair_replay_check(xproj_v, w_hh_v, saved_v, xproj_r, w_hh_r, b_hh_r, saved_r, n, p0_v, p0_r, n_calls, full_calls,
                            steps, work_v, work_r, err, arith="f32"):
    """Runs, for every call whose replay the chains left out (full_calls < n_calls), the first `steps` positions
    of that replay from the state the chain had before the call, and writes to err[0:2] how far the value LSTM's /
    reward GRU's state then is from the state the last fully walked call had there."""
    rows = n_calls - full_calls
    assert 1 <= full_calls < n_calls and 1 <= steps <= n * (min(p0_v, p0_r) + full_calls - 1)
    assert saved_v.numel() >= chain_positions(n, p0_v, n_calls, full_calls) * 6 * 512
    assert saved_r.numel() >= chain_positions(n, p0_r, n_calls, full_calls) * 5 * 512
    assert xproj_v.numel() >= steps * 4 * 512 and xproj_r.numel() >= steps * 3 * 512
    assert work_v.numel() >= chain_seg_workspace_floats(rows) and work_r.numel() >= chain_seg_workspace_floats(rows)
    assert err.numel() >= 2
    assert arith in ("f32", "f16x2")
    fn = load().icrl_chain_pair_replay_check_f16x2 if arith == "f16x2" else load().icrl_chain_pair_replay_check
    with _Timed("replay_check", rows * steps, steps=steps + 1):
        rc = fn(_ptr(_f32c(xproj_v)), _ptr(_f32c(w_hh_v)), _ptr(_f32c(saved_v)),
                                                 _ptr(_f32c(xproj_r)), _ptr(_f32c(w_hh_r)), _ptr(_f32c(b_hh_r)),
                                                 _ptr(_f32c(saved_r)), n, p0_v, p0_r, n_calls, full_calls, steps,
                                                 _ptr(_f32c(work_v)), _ptr(_f32c(work_r)), _ptr(_f32c(err)), _stream())
    _check(rc, "icrl_chain_pair_replay_check")


def chain_give_up_selftest(flags, abort_words, epoch, wide, err):
    """One bounded wait of the persistent launches on caller-supplied flags, then the kernel that reports a wait
    that ran out (see icrl_chain_give_up_selftest)."""
    assert flags.dtype == abort_words.dtype == torch.int32 and flags.numel() >= 32 and abort_words.numel() >= 4
    assert err.dtype == torch.float32 and err.numel() >= 2
    _check(load().icrl_chain_give_up_selftest(_ptr(flags), _ptr(abort_words), int(epoch), int(wide), _ptr(err), _stream()),
           "icrl_chain_give_up_selftest")


def chain_pair_fwd_rows(xproj_v, w_hh_v, xproj_r, w_hh_r, b_hh_r, n, p0_v, p0_r, n_calls, h_emit_v, saved_v, h_emit_r,
                        work_v, work_r, status, persistent=True):
    """Value LSTM and reward GRU with every row its own chain (the reference's classes at batch size 1)."""
    _chain_shapes("chain_pair_fwd_rows (value)", n, p0_v, n_calls, 4, xproj=xproj_v, w_hh=w_hh_v, h_emit=h_emit_v,
                  saved=saved_v, slots=6, work=work_v, segments=n)
    _chain_shapes("chain_pair_fwd_rows (reward)", n, p0_r, n_calls, 3, xproj=xproj_r, w_hh=w_hh_r, h_emit=h_emit_r,
                  work=work_r, segments=n)
    assert status.dtype == torch.int32
    with _Timed("chain_pair_fwd", chain_positions(1, p0_v, n_calls) + chain_positions(1, p0_r, n_calls)):
        rc = load().icrl_chain_pair_fwd_rows(_ptr(_f32c(xproj_v)), _ptr(_f32c(w_hh_v)), _ptr(_f32c(xproj_r)),
                                             _ptr(_f32c(w_hh_r)), _ptr(_f32c(b_hh_r)), n, p0_v, p0_r, n_calls,
                                             int(persistent), _ptr(_f32c(h_emit_v)), _ptr(saved_v),
                                             _ptr(_f32c(h_emit_r)), _ptr(_f32c(work_v)), _ptr(_f32c(work_r)),
                                             _ptr(status), _stream())
    _check(rc, "icrl_chain_pair_fwd_rows")


def lstm_chain_bwd_rows(w_hh, saved, d_emit, n, p0, n_calls, dgates, work, status, persistent=True):
    _chain_shapes("lstm_chain_bwd_rows", n, p0, n_calls, 4, w_hh=w_hh, saved=saved, slots=6, d_emit=d_emit,
                  dgates=(dgates,), work=work, segments=n)
    assert status.dtype == torch.int32
    with _Timed("lstm_chain_bwd", chain_positions(1, p0, n_calls)):
        rc = load().icrl_lstm_chain_bwd_rows(_ptr(_f32c(w_hh)), _ptr(_f32c(saved)), _ptr(_f32c(d_emit)), n, p0, n_calls,
                                             int(persistent), _ptr(_f32c(dgates)), _ptr(_f32c(work)), _ptr(status),
                                             _stream())
    _check(rc, "icrl_lstm_chain_bwd_rows")


def chain_bwd_warm16_supported(segments):
    return bool(load().icrl_chain_bwd_warm16_supported(int(segments)))


def lstm_chain_bwd_seg(w_hh, saved, d_emit, n, p0, n_calls, segments, warmup, dgates, work, seam_err, persistent=False,
                       full_calls=0, cheap=0, arith="f32"):
    """cheap: the first steps of the warm-up that run at 16-bit split precision in a launch of their own
    (icrl_lstm_chain_bwd_seg_warm16; persistent launches only).  arith="f16x2": the owner-computes launch on the
    f16 matrix instruction (icrl_lstm_chain_bwd_seg_f16x2; work: chain_bwd_f16x2_workspace_floats(segments))."""
    _chain_shapes("lstm_chain_bwd_seg", n, p0, n_calls, 4, w_hh=w_hh, saved=saved, slots=6, d_emit=d_emit,
                  dgates=(dgates,), work=None if arith == "f16x2" else work, segments=segments, full_calls=full_calls)
    if arith == "f16x2":
        assert cheap == 0 and work.numel() >= chain_bwd_f16x2_workspace_floats(segments)
        steps = _seg_steps(n, p0, n_calls, full_calls, segments, warmup)
        with _Timed("lstm_chain_bwd_f16x2", chain_positions(n, p0, n_calls, full_calls), steps=steps,
                    flops=2.0 * segments * 512 * 512 * 4 * steps):
            rc = load().icrl_lstm_chain_bwd_seg_f16x2(_ptr(_f32c(w_hh)), _ptr(_f32c(saved)), _ptr(_f32c(d_emit)), n, p0,
                                                      n_calls, full_calls, segments, warmup, _ptr(_f32c(dgates)),
                                                      _ptr(_f32c(work)), _ptr(seam_err), _stream())
        _check(rc, "icrl_lstm_chain_bwd_seg_f16x2")
        return
    assert cheap == 0 or persistent, "the 16-bit warm-up belongs to the persistent launch"
    steps = _seg_steps(n, p0, n_calls, full_calls, segments, warmup)
    head = (_ptr(_f32c(w_hh)), _ptr(_f32c(saved)), _ptr(_f32c(d_emit)), n, p0, n_calls, full_calls, segments, warmup)
    tail = (_ptr(_f32c(dgates)), _ptr(_f32c(work)), _ptr(seam_err), _stream())
    # float32 matrix-instruction work: the steps that do not run at 16-bit split precision
    with _Timed("lstm_chain_bwd", chain_positions(n, p0, n_calls, full_calls), steps=steps,
                flops=2.0 * segments * 512 * 512 * 4 * (steps - cheap)):
        if cheap:
            rc = load().icrl_lstm_chain_bwd_seg_warm16(*head, int(cheap), *tail)
        elif persistent:
            rc = load().icrl_lstm_chain_bwd_seg_persist(*head, *tail)
        else:
            rc = load().icrl_lstm_chain_bwd_seg(*head, *tail)
    _check(rc, "icrl_lstm_chain_bwd_seg")


def policy_lstm_bptt(gates_act, c_all, d_h, w_hh, dgates, dc0, dc1, dh0, scratch, *, shared=False, dtype="f32"):
    """Backward-through-time of the policy LSTM over all T steps in one call (see icrl_policy_lstm_bptt);
    the products' tile / split-K plan is plan_gemm's, as for gemm_auto."""
    t_steps, n = gates_act.shape[0], gates_act.shape[1]
    assert gates_act.shape == (t_steps, n, 2048) and c_all.shape[0] >= t_steps + 1 and c_all.shape[1:] == (n, 512)
    assert d_h.shape == (t_steps, n, 512) and dgates.shape == (t_steps, n, 2048) and w_hh.shape == (2048, 512)
    for t in (dc0, dc1, dh0):
        assert t.shape == (n, 512)
    tile, splits = plan_gemm(n, 512, 2048, True, shared)
    if splits > 1:
        splits = max(1, min(splits, scratch.numel() // (n * 512)))
    if dtype == "bf16" and tile == 2:
        tile = 1
    # one timing record for the whole sequence, under a kind of its own: it holds the cell kernels and the slab
    # sums beside the T products of 2 N 512 2048 FLOPs each
    with _Timed("policy_bptt", t_steps, steps=t_steps, flops=2.0 * t_steps * n * 512 * 2048):
        rc = load().icrl_policy_lstm_bptt(_ptr(_f32c(gates_act)), _ptr(_f32c(c_all)), _ptr(_f32c(d_h)), _ptr(_f32c(w_hh)),
                                          _ptr(_f32c(dgates)), _ptr(_f32c(dc0)), _ptr(_f32c(dc1)), _ptr(_f32c(dh0)),
                                          _ptr(_f32c(scratch)), scratch.numel(), n, t_steps, splits, int(tile),
                                          int(dtype == "bf16"), _stream())
    _check(rc, "icrl_policy_lstm_bptt")


def chain_bwd_f16x2_workspace_floats(segments):
    return int(load().icrl_chain_bwd_f16x2_workspace_floats(int(segments)))


def chain_bwd_f16x2_supported(segments):
    return bool(load().icrl_chain_bwd_f16x2_supported(int(segments)))


def policy_lstm_bptt_persist(gates_act, c_all, d_h, w_hh, dgates, dh0, work, status, arith="f32"):
    """The policy LSTM's backward-through-time over all T steps as ONE persistent launch (the backward chain kernel
    on N per-row chains with a gradient injected at every step); d_h is read only.  Needs chain_pair_supported(N)."""
    t_steps, n = gates_act.shape[0], gates_act.shape[1]
    assert gates_act.shape == (t_steps, n, 2048) and c_all.shape[0] >= t_steps + 1 and c_all.shape[1:] == (n, 512)
    assert d_h.shape == (t_steps, n, 512) and dgates.shape == (t_steps, n, 2048) and w_hh.shape == (2048, 512)
    need = chain_bwd_f16x2_workspace_floats(n) if arith == "f16x2" else chain_seg_workspace_floats(n)
    assert dh0.shape == (n, 512) and work.numel() >= need
    assert status.dtype == torch.int32 and status.numel() >= 1
    fn = load().icrl_policy_lstm_bptt_f16x2 if arith == "f16x2" else load().icrl_policy_lstm_bptt_persist
    with _Timed("policy_bptt", t_steps, steps=t_steps + 1, flops=2.0 * t_steps * n * 512 * 2048):
        rc = fn(_ptr(_f32c(gates_act)), _ptr(_f32c(c_all)), _ptr(_f32c(d_h)), _ptr(_f32c(w_hh)), n, t_steps,
                _ptr(_f32c(dgates)), _ptr(_f32c(dh0)), _ptr(_f32c(work)), _ptr(status), _stream())
    _check(rc, "icrl_policy_lstm_bptt_persist")


def policy_lstm_step(emb, tokens, w_ih, w_hh, b_ih, b_hh, h_cur, c_cur, gates_act, c_next, h_next, x_out=None):
    """Embedding gather + both gate products + biases + LSTM cell for N rows, one launch."""
    n = h_cur.shape[0]
    assert tokens.dtype == torch.int64 and tokens.is_contiguous() and tokens.numel() == n
    _check(load().icrl_policy_lstm_step(_ptr(_f32c(emb)), emb.shape[1], _ptr(tokens), _ptr(_f32c(w_ih)),
                                        _ptr(_f32c(w_hh)), _ptr(_f32c(b_ih)), _ptr(_f32c(b_hh)), _ptr(_f32c(h_cur)),
                                        _ptr(_f32c(c_cur)), _ptr(_f32c(gates_act)), _ptr(_f32c(c_next)),
                                        _ptr(_f32c(h_next)), _ptr(x_out), n, _stream()), "icrl_policy_lstm_step")
