// Segment-parallel evaluation of the batch-1 recurrent chains, with a seam check.
//
// The value LSTM / reward GRU of the reference are ONE recurrence of ~10^5 dependent
// steps per minibatch (chain.hip walks it exactly, one step at a time).  A gated recurrence
// forgets its initial state: after enough steps two trajectories driven by the same inputs
// coincide to rounding.  So the chain is cut into K segments that run side by side as a
// batch: segment j first runs `warm` extra steps over the positions just before its start,
// from a zero state, then its own positions.  Every step of all segments is one
// [K,512] x W_hh^T product on the matrix cores plus one fused cell kernel, instead of K
// dependent mat-vecs.  Nothing is assumed about how fast the state is forgotten: the state
// each segment reached after its warm-up is compared with the state the previous segment
// actually ended in (its "seam"); the largest relative difference over all seams is
// returned in *seam_err and the caller falls back to the exact kernel when it is not at
// rounding level.  Segment 0 starts from the true initial state, so it is exact.
// The backward pass does the same in reverse (warm-up over the positions after the
// segment's end, seam check on (dh, dc)).
// Outputs are written in the same per-position layout as the exact kernels.
#include "icrl_common.h"

namespace {
using namespace icrl;

constexpr int H = kHidden;
constexpr int H4 = H / 4;

struct Geom {
    int n, p0, n_calls;   // rows per column, columns in the first call, calls
    int64_t total;        // positions in the chain
    int K, Ls, W;         // segments, positions per segment, warm-up steps
};

// position -> (call s, column t, row i)
__device__ __forceinline__ void locate(const Geom& g, int64_t pos, int& s, int& t, int& i) {
    int64_t off = 0;
    for (s = 0; s < g.n_calls - 1; ++s) {
        const int64_t len = (int64_t)(g.p0 + s) * g.n;
        if (pos < off + len) break;
        off += len;
    }
    const int64_t rem = pos - off;
    t = (int)(rem / g.n);
    i = (int)(rem % g.n);
}

__device__ __forceinline__ float sg(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float th(float x) { return 2.f * sg(2.f * x) - 1.f; }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

__device__ __forceinline__ void atomic_max_nonneg(float* addr, float v) {
    atomicMax(reinterpret_cast<unsigned*>(addr), __float_as_uint(v));  // order-preserving for v >= 0
}

// ------------------------------------------------------------------ LSTM forward cell
__global__ __launch_bounds__(256) void seg_lstm_fwd_cell(const float* __restrict__ hw, const float* __restrict__ xproj,
                                                         const float* __restrict__ c_cur, float* __restrict__ c_nxt,
                                                         float* __restrict__ h_nxt, float* __restrict__ saved,
                                                         float* __restrict__ h_emit, float* __restrict__ seam_a,
                                                         float* __restrict__ seam_b, Geom g, int m) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= g.K * H4) return;
    const int j = e / H4, u = 4 * (e % H4);
    const int64_t seg_beg = (int64_t)j * g.Ls;
    const int64_t seg_end = min(g.total, seg_beg + g.Ls);
    const int64_t pos = seg_beg - g.W + m;
    if (pos < 0 || pos >= seg_end) {
        st4(c_nxt + j * H + u, make_float4(0.f, 0.f, 0.f, 0.f));
        st4(h_nxt + j * H + u, make_float4(0.f, 0.f, 0.f, 0.f));
        return;
    }
    int s, t, i;
    locate(g, pos, s, t, i);
    const float* xr = xproj + ((int64_t)t * g.n + i) * (4 * H) + u;
    const float* hr = hw + (int64_t)j * (4 * H) + u;
    float4 gi = add4(ld4(hr), ld4(xr)), gf = add4(ld4(hr + H), ld4(xr + H));
    float4 gg = add4(ld4(hr + 2 * H), ld4(xr + 2 * H)), go = add4(ld4(hr + 3 * H), ld4(xr + 3 * H));
    const float4 cp = ld4(c_cur + j * H + u);
    float4 c, h;
#define CELL(k)                                                              \
    gi.k = sg(gi.k); gf.k = sg(gf.k); gg.k = th(gg.k); go.k = sg(go.k);      \
    c.k = gf.k * cp.k + gi.k * gg.k;                                         \
    h.k = go.k * th(c.k);
    CELL(x) CELL(y) CELL(z) CELL(w)
#undef CELL
    st4(c_nxt + j * H + u, c);
    st4(h_nxt + j * H + u, h);
    if (pos >= seg_beg) {
        if (saved) {
            float* sp = saved + pos * (6 * H) + u;
            st4(sp, gi); st4(sp + H, gf); st4(sp + 2 * H, gg); st4(sp + 3 * H, go); st4(sp + 4 * H, c); st4(sp + 5 * H, h);
        }
        if (t == g.p0 + s - 1) st4(h_emit + ((int64_t)s * g.n + i) * H + u, h);
    }
    if (pos == seg_beg - 1) { st4(seam_a + (int64_t)j * 2 * H + u, h); st4(seam_a + (int64_t)j * 2 * H + H + u, c); }
    if (pos == seg_beg + g.Ls - 1 && j + 1 < g.K) {
        st4(seam_b + (int64_t)(j + 1) * 2 * H + u, h);
        st4(seam_b + (int64_t)(j + 1) * 2 * H + H + u, c);
    }
}

// ------------------------------------------------------------------ GRU forward cell
__global__ __launch_bounds__(256) void seg_gru_fwd_cell(const float* __restrict__ hp, const float* __restrict__ xproj,
                                                        const float* __restrict__ b_hh, const float* __restrict__ h_cur,
                                                        float* __restrict__ h_nxt, float* __restrict__ h_emit,
                                                        float* __restrict__ seam_a, float* __restrict__ seam_b, Geom g,
                                                        int m) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= g.K * H4) return;
    const int j = e / H4, u = 4 * (e % H4);
    const int64_t seg_beg = (int64_t)j * g.Ls;
    const int64_t seg_end = min(g.total, seg_beg + g.Ls);
    const int64_t pos = seg_beg - g.W + m;
    if (pos < 0 || pos >= seg_end) {
        st4(h_nxt + j * H + u, make_float4(0.f, 0.f, 0.f, 0.f));
        return;
    }
    int s, t, i;
    locate(g, pos, s, t, i);
    const float* xr = xproj + ((int64_t)t * g.n + i) * (3 * H) + u;
    const float* hr = hp + (int64_t)j * (3 * H) + u;
    const float4 xr_ = ld4(xr), xz = ld4(xr + H), xn = ld4(xr + 2 * H);
    const float4 hr_ = add4(ld4(hr), ld4(b_hh + u)), hz = add4(ld4(hr + H), ld4(b_hh + H + u));
    const float4 hn = add4(ld4(hr + 2 * H), ld4(b_hh + 2 * H + u));
    const float4 hp_ = ld4(h_cur + j * H + u);
    float4 h;
#define CELL(k)                                              \
    {                                                        \
        const float r = sg(xr_.k + hr_.k);                   \
        const float z = sg(xz.k + hz.k);                     \
        const float nn = th(xn.k + r * hn.k);                \
        h.k = (1.f - z) * nn + z * hp_.k;                    \
    }
    CELL(x) CELL(y) CELL(z) CELL(w)
#undef CELL
    st4(h_nxt + j * H + u, h);
    if (pos >= seg_beg && t == g.p0 + s - 1) st4(h_emit + ((int64_t)s * g.n + i) * H + u, h);
    if (pos == seg_beg - 1) st4(seam_a + (int64_t)j * H + u, h);
    if (pos == seg_beg + g.Ls - 1 && j + 1 < g.K) st4(seam_b + (int64_t)(j + 1) * H + u, h);
}

// max over seams j >= 1 of |a - b| / max(1, |b|), `width` floats per seam
__global__ __launch_bounds__(256) void seam_check_fwd(const float* __restrict__ seam_a, const float* __restrict__ seam_b,
                                                      Geom g, int width, float* __restrict__ err) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    float d = 0.f;
    if (e < g.K * width) {
        const int j = e / width;
        // seam j exists when segment j has positions and was warmed up from a zero state
        if (j >= 1 && (int64_t)j * g.Ls < g.total && (int64_t)j * g.Ls - g.W > 0) {
            const float a = seam_a[e], b = seam_b[e];
            d = fabsf(a - b) / fmaxf(1.f, fabsf(b));
            if (!(d == d)) d = INFINITY;
        }
    }
    d = wave_max(d);
    if ((threadIdx.x & 63) == 0 && d > 0.f) atomic_max_nonneg(err, d);
}

// ------------------------------------------------------------------ LSTM backward cell
// State entering position p (walking down): dh_in = W_hh^T dgates(p+1) (sum of `nslab`
// split-K slabs) and dc (carried).  Step mp handles pos = seg_beg + Ls + W - 1 - mp.
__global__ __launch_bounds__(256) void seg_lstm_bwd_cell(const float* __restrict__ dhw, int nslab, int64_t slab,
                                                         const float* __restrict__ saved,
                                                         const float* __restrict__ d_emit, float* __restrict__ dc,
                                                         float* __restrict__ dg_cur, float* __restrict__ dgv,
                                                         float* __restrict__ seam_a, Geom g, int mp) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= g.K * H4) return;
    const int j = e / H4, u = 4 * (e % H4);
    const int64_t seg_beg = (int64_t)j * g.Ls;
    const int64_t pos = seg_beg + g.Ls + g.W - 1 - mp;
    float4 dh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mp > 0)
        for (int k = 0; k < nslab; ++k) dh = add4(dh, ld4(dhw + k * slab + (int64_t)j * H + u));
    float4 dcv = ld4(dc + j * H + u);
    if (pos == seg_beg + g.Ls - 1) {  // state entering the segment's last own position
        st4(seam_a + (int64_t)j * 2 * H + u, dh);
        st4(seam_a + (int64_t)j * 2 * H + H + u, dcv);
    }
    float* dgo = dg_cur + (int64_t)j * (4 * H) + u;
    if (pos >= g.total || seg_beg >= g.total) {
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        st4(dgo, z); st4(dgo + H, z); st4(dgo + 2 * H, z); st4(dgo + 3 * H, z);
        st4(dc + j * H + u, z);
        return;
    }
    int s, t, i;
    locate(g, pos, s, t, i);
    if (t == g.p0 + s - 1) dh = add4(dh, ld4(d_emit + ((int64_t)s * g.n + i) * H + u));
    const float* sp = saved + pos * (6 * H) + u;
    const float4 gi = ld4(sp), gf = ld4(sp + H), gg = ld4(sp + 2 * H), go = ld4(sp + 3 * H), cc = ld4(sp + 4 * H);
    float4 cp = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pos > 0) cp = ld4(saved + (pos - 1) * (6 * H) + 4 * H + u);
    float4 d_i, d_f, d_g, d_o, dcn;
#define CELL(k)                                                       \
    {                                                                 \
        const float tc = th(cc.k);                                    \
        const float dcc = dh.k * go.k * (1.f - tc * tc) + dcv.k;      \
        d_i.k = dcc * gg.k * gi.k * (1.f - gi.k);                     \
        d_f.k = dcc * cp.k * gf.k * (1.f - gf.k);                     \
        d_g.k = dcc * gi.k * (1.f - gg.k * gg.k);                     \
        d_o.k = dh.k * tc * go.k * (1.f - go.k);                      \
        dcn.k = dcc * gf.k;                                           \
    }
    CELL(x) CELL(y) CELL(z) CELL(w)
#undef CELL
    st4(dgo, d_i); st4(dgo + H, d_f); st4(dgo + 2 * H, d_g); st4(dgo + 3 * H, d_o);
    st4(dc + j * H + u, dcn);
    if (pos < seg_beg + g.Ls) {
        float* dp = dgv + pos * (4 * H) + u;
        st4(dp, d_i); st4(dp + H, d_f); st4(dp + 2 * H, d_g); st4(dp + 3 * H, d_o);
    }
}

// seam j (between segments j and j+1): warm-up state of segment j vs final state of j+1,
// relative to the largest entry of the true state vector's kind (dh or dc) over all seams.
__global__ __launch_bounds__(256) void seam_check_bwd(const float* __restrict__ seam_a, const float* __restrict__ dhw,
                                                      int nslab, int64_t slab, const float* __restrict__ dc, Geom g,
                                                      float* __restrict__ err_abs, float* __restrict__ ref_abs) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    float d = 0.f, r = 0.f;
    if (e < g.K * 2 * H) {
        const int j = e / (2 * H), k = e % (2 * H);
        if (j + 1 < g.K && (int64_t)(j + 1) * g.Ls + g.W < g.total) {
            float b;
            if (k < H) {
                b = 0.f;
                for (int q = 0; q < nslab; ++q) b += dhw[q * slab + (int64_t)(j + 1) * H + k];
            } else {
                b = dc[(int64_t)(j + 1) * H + (k - H)];
            }
            d = fabsf(seam_a[e] - b);
            r = fabsf(b);
            if (!(d == d)) d = INFINITY;
        }
    }
    d = wave_max(d);
    r = wave_max(r);
    if ((threadIdx.x & 63) == 0) {
        if (d > 0.f) atomic_max_nonneg(err_abs, d);
        if (r > 0.f) atomic_max_nonneg(ref_abs, r);
    }
}

int make_geom(const char* who, int N, int p0, int n_calls, int segments, int warmup, Geom& g) {
    ICRL_REQUIRE(N > 0 && p0 > 0 && n_calls > 0, "%s: N, p0 and n_calls must be positive", who);
    ICRL_REQUIRE(segments >= 2 && segments <= 4096 && warmup >= 1, "%s: need 2 <= segments <= 4096, warmup >= 1", who);
    g.n = N; g.p0 = p0; g.n_calls = n_calls;
    int64_t cols = 0;
    for (int s = 0; s < n_calls; ++s) cols += p0 + s;
    g.total = cols * N;
    g.K = segments;
    g.Ls = (int)((g.total + segments - 1) / segments);
    g.W = warmup;
    return ICRL_OK;
}

inline int cell_grid(const Geom& g) { return (g.K * H4 + 255) / 256; }

}  // namespace

extern "C" int64_t icrl_chain_seg_workspace_floats(int segments) { return (int64_t)segments * 16 * H; }

extern "C" int icrl_lstm_chain_fwd_seg(const float* xproj, const float* w_hh, int N, int p0, int n_calls,
                                       int segments, int warmup, float* h_emit, float* saved, float* work,
                                       float* seam_err, icrl_stream_t stream) {
    Geom g;
    if (int rc = make_geom("icrl_lstm_chain_fwd_seg", N, p0, n_calls, segments, warmup, g)) return rc;
    ICRL_REQUIRE(xproj && w_hh && h_emit && work && seam_err, "icrl_lstm_chain_fwd_seg: null pointer");
    hipStream_t st = icrl::as_stream(stream);
    const int64_t kh = (int64_t)g.K * H;
    float* hb[2] = {work, work + kh};
    float* cb[2] = {work + 2 * kh, work + 3 * kh};
    float* hw = work + 4 * kh;            // [K, 4H]
    float* seam_a = work + 8 * kh;        // [K, 2H]
    float* seam_b = work + 10 * kh;       // [K, 2H]
    ICRL_HIP(hipMemsetAsync(work, 0, sizeof(float) * 4 * kh, st));
    ICRL_HIP(hipMemsetAsync(seam_err, 0, sizeof(float), st));
    const int steps = g.W + g.Ls;
    for (int m = 0; m < steps; ++m) {
        const int cur = m & 1;
        if (int rc = icrl::gemm(hb[cur], H, 0, w_hh, H, 0, hw, 4 * H, g.K, 4 * H, H, 1, 0, st)) return rc;
        hipLaunchKernelGGL(seg_lstm_fwd_cell, dim3(cell_grid(g)), dim3(256), 0, st, hw, xproj, cb[cur], cb[cur ^ 1],
                           hb[cur ^ 1], saved, h_emit, seam_a, seam_b, g, m);
    }
    hipLaunchKernelGGL(seam_check_fwd, dim3((g.K * 2 * H + 255) / 256), dim3(256), 0, st, seam_a, seam_b, g, 2 * H,
                       seam_err);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_gru_chain_fwd_seg(const float* xproj, const float* w_hh, const float* b_hh, int N, int p0,
                                      int n_calls, int segments, int warmup, float* h_emit, float* work,
                                      float* seam_err, icrl_stream_t stream) {
    Geom g;
    if (int rc = make_geom("icrl_gru_chain_fwd_seg", N, p0, n_calls, segments, warmup, g)) return rc;
    ICRL_REQUIRE(xproj && w_hh && b_hh && h_emit && work && seam_err, "icrl_gru_chain_fwd_seg: null pointer");
    hipStream_t st = icrl::as_stream(stream);
    const int64_t kh = (int64_t)g.K * H;
    float* hb[2] = {work, work + kh};
    float* hp = work + 2 * kh;            // [K, 3H]
    float* seam_a = work + 5 * kh;        // [K, H]
    float* seam_b = work + 6 * kh;        // [K, H]
    ICRL_HIP(hipMemsetAsync(work, 0, sizeof(float) * 2 * kh, st));
    ICRL_HIP(hipMemsetAsync(seam_err, 0, sizeof(float), st));
    const int steps = g.W + g.Ls;
    for (int m = 0; m < steps; ++m) {
        const int cur = m & 1;
        if (int rc = icrl::gemm(hb[cur], H, 0, w_hh, H, 0, hp, 3 * H, g.K, 3 * H, H, 1, 0, st)) return rc;
        hipLaunchKernelGGL(seg_gru_fwd_cell, dim3(cell_grid(g)), dim3(256), 0, st, hp, xproj, b_hh, hb[cur],
                           hb[cur ^ 1], h_emit, seam_a, seam_b, g, m);
    }
    hipLaunchKernelGGL(seam_check_fwd, dim3((g.K * H + 255) / 256), dim3(256), 0, st, seam_a, seam_b, g, H, seam_err);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_lstm_chain_bwd_seg(const float* w_hh, const float* saved, const float* d_emit, int N, int p0,
                                       int n_calls, int segments, int warmup, float* dgates, float* work,
                                       float* seam_err, icrl_stream_t stream) {
    Geom g;
    if (int rc = make_geom("icrl_lstm_chain_bwd_seg", N, p0, n_calls, segments, warmup, g)) return rc;
    ICRL_REQUIRE(w_hh && saved && d_emit && dgates && work && seam_err, "icrl_lstm_chain_bwd_seg: null pointer");
    hipStream_t st = icrl::as_stream(stream);
    const int64_t kh = (int64_t)g.K * H;
    constexpr int SPLITS = 4;
    float* dg_cur = work;                 // [K, 4H]
    float* dhw = work + 4 * kh;           // [SPLITS][K, H]
    float* dc = work + 8 * kh;            // [K, H]
    float* seam_a = work + 9 * kh;        // [K, 2H]
    ICRL_HIP(hipMemsetAsync(dc, 0, sizeof(float) * 3 * kh, st));
    ICRL_HIP(hipMemsetAsync(seam_err, 0, 2 * sizeof(float), st));
    const int steps = g.W + g.Ls;
    for (int mp = 0; mp < steps; ++mp) {
        hipLaunchKernelGGL(seg_lstm_bwd_cell, dim3(cell_grid(g)), dim3(256), 0, st, dhw, SPLITS, kh, saved, d_emit, dc,
                           dg_cur, dgates, seam_a, g, mp);
        // dh entering the next-lower position: dgates . W_hh, as SPLITS partial sums over the 2048 gate rows
        if (int rc = icrl::gemm(dg_cur, 4 * H, 0, w_hh, H, 1, dhw, H, g.K, H, 4 * H, SPLITS, kh, st)) return rc;
    }
    hipLaunchKernelGGL(seam_check_bwd, dim3((g.K * 2 * H + 255) / 256), dim3(256), 0, st, seam_a, dhw, SPLITS, kh, dc, g,
                       seam_err, seam_err + 1);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}
