// Segment-parallel evaluation of the batch-1 recurrent chains, with a seam check.
//
// The value LSTM / reward GRU of the reference are ONE recurrence of ~10^5 dependent
// steps per minibatch (chain.hip walks it exactly, one step at a time).  A gated recurrence
// forgets its initial state: after enough steps two trajectories driven by the same inputs
// coincide to rounding.  So the chain is cut into K segments that run side by side as a
// batch: segment j first runs `warm` extra steps over the positions just before its start,
// from a zero state, then its own positions.  Nothing is assumed about how fast the state
// is forgotten: the state each segment reached after its warm-up is compared with the state
// the previous segment actually ended in (its "seam"); the largest mismatch over all seams
// is returned in *seam_err and the caller falls back to a longer warm-up or to the exact
// kernel when it is not at rounding level.  Segment 0 starts from the true initial state.
// The backward pass does the same in reverse (warm-up over the positions after the
// segment's end, seam check on (dh, dc)).  Outputs use the exact kernels' per-position
// layouts.
//
// One launch per step, product and cell fused:
//   forward  : [32 segments] x [G gates x 32 units] tile per workgroup; wave q owns gate q's
//              32x32 accumulator (v_mfma_f32_32x32x2_f32, k = 512 in stages of 64 through
//              double-buffered LDS); the G gate tiles meet in LDS and the cell runs in the
//              epilogue (input projection added, c/h/saved/emit written).
//   backward : [32 segments] x [32 units] tile of dh = dgates . W_hh per workgroup; the four
//              waves split the 2048 gate rows (k), their partial tiles are summed in LDS,
//              the cell's backward runs in the epilogue and writes the next step's dgates.
#include "icrl_common.h"

namespace {
using namespace icrl;

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int H = kHidden;
constexpr int BK = 64;        // k per LDS stage
constexpr int LDT = BK + 4;   // padded row of a k-contiguous tile (ds_read_b128 conflict-free)
constexpr int PLD = 33;       // padded row of the epilogue exchange tiles

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
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 ldp4(const float* p) { return make_float4(p[0], p[1], p[2], p[3]); }

__device__ __forceinline__ void atomic_max_nonneg(float* addr, float v) {
    atomicMax(reinterpret_cast<unsigned*>(addr), __float_as_uint(v));  // order-preserving for v >= 0
}

// 8 MFMA steps of one 16-k group: lane half hh supplies k = 8*hh + s at step s for both operands
__device__ __forceinline__ void mfma_group(f32x16& acc, const float* a_row, const float* b_row, int grp, int hh) {
    const float4 a0 = ld4(a_row + 16 * grp + 8 * hh), a1 = ld4(a_row + 16 * grp + 8 * hh + 4);
    const float4 b0 = ld4(b_row + 16 * grp + 8 * hh), b1 = ld4(b_row + 16 * grp + 8 * hh + 4);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, b0.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, b0.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, b0.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, b0.w, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, acc, 0, 0, 0);
}

// accumulator register r of lane (l31, hh) is element [row = (r&3) + 8*(r>>2) + 4*hh][col = l31]
__device__ __forceinline__ void acc_to_lds(const f32x16& acc, float* tile, int l31, int hh) {
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * hh) * PLD + l31] = acc[r];
}

// ------------------------------------------------------------------ forward step (LSTM: G = 4, GRU: G = 3)
template <int G>
__global__ __launch_bounds__(G * 64) void seg_fwd_step(const float* __restrict__ h_cur, const float* __restrict__ w_hh,
                                                       const float* __restrict__ xproj, const float* __restrict__ b_hh,
                                                       const float* __restrict__ c_cur, float* __restrict__ c_nxt,
                                                       float* __restrict__ h_nxt, float* __restrict__ saved,
                                                       float* __restrict__ h_emit, float* __restrict__ seam_a,
                                                       float* __restrict__ seam_b, Geom g, int m) {
    constexpr int NT = G * 64;                 // threads
    constexpr int ROWS = 32 + G * 32;          // stage rows: 32 of H, then G x 32 of W_hh
    constexpr int F4 = ROWS * (BK / 4);        // float4 per stage
    constexpr int PER = (F4 + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) float lds[2 * ROWS * LDT];
    const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    // blocks x, x+8, ... share an XCD (round-robin dispatch): giving them the same unit tiles keeps that
    // XCD's slice of W_hh (1/8 of it) resident in its L2 across the segment-row tiles and across steps
    const int row0 = blockIdx.y * 32, u0 = blockIdx.x * 32;

    // source row pointer of each float4 this thread stages (fixed over the k loop)
    const float* src[PER];
    int dst[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int f = tid + NT * i;
        const int r = f / (BK / 4), kq = f % (BK / 4);
        dst[i] = (f < F4) ? r * LDT + 4 * kq : -1;
        if (f >= F4) { src[i] = h_cur; continue; }
        if (r < 32) {
            // rows past the last segment read a valid row instead of branching; their results are never used
            src[i] = h_cur + (int64_t)min(row0 + r, g.K - 1) * H + 4 * kq;
        } else {
            const int rb = r - 32;
            src[i] = w_hh + (int64_t)((rb >> 5) * H + u0 + (rb & 31)) * H + 4 * kq;
        }
    }
    // The epilogue's operands (input projection row, previous c / h) do not depend on the product:
    // fetch them now so that their HBM latency hides under the k loop.
    constexpr int NI = (256 + NT - 1) / NT;   // epilogue items (segment row, 4 units) per thread
    int it_state[NI];                          // 0 = none, 1 = outside the segment's range (zero state), 2 = live
    int64_t it_pos[NI];
    int it_s[NI], it_t[NI], it_i[NI];
    float4 px[NI][G], pprev[NI];
#pragma unroll
    for (int ii = 0; ii < NI; ++ii) {
        const int item = tid + NT * ii;
        const int j = row0 + (item >> 3), u = u0 + 4 * (item & 7);
        it_state[ii] = 0; it_pos[ii] = 0; it_s[ii] = it_t[ii] = it_i[ii] = 0;
        if (item >= 256 || j >= g.K) continue;
        const int64_t seg_beg = (int64_t)j * g.Ls;
        const int64_t pos = seg_beg - g.W + m;
        it_pos[ii] = pos;
        if (pos < 0 || pos >= min(g.total, seg_beg + g.Ls)) { it_state[ii] = 1; continue; }
        it_state[ii] = 2;
        locate(g, pos, it_s[ii], it_t[ii], it_i[ii]);
        const float* xr = xproj + ((int64_t)it_t[ii] * g.n + it_i[ii]) * (G * H) + u;
#pragma unroll
        for (int gq = 0; gq < G; ++gq) px[ii][gq] = ld4(xr + gq * H);
        pprev[ii] = ld4((G == 4 ? c_cur : h_cur) + (int64_t)j * H + u);
    }

    float4 stage[PER];
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    if (m > 0) {  // the first step starts from a zero state: nothing to multiply
#pragma unroll
        for (int i = 0; i < PER; ++i) stage[i] = ld4(src[i]);
#pragma unroll
        for (int i = 0; i < PER; ++i) if (dst[i] >= 0) st4(lds + dst[i], stage[i]);
        __syncthreads();
        constexpr int NK = H / BK;
        for (int kt = 0; kt < NK; ++kt) {
            const float* cur = lds + (kt & 1) * ROWS * LDT;
            if (kt + 1 < NK) {
#pragma unroll
                for (int i = 0; i < PER; ++i) stage[i] = ld4(src[i] + (kt + 1) * BK);
            }
            const float* a_row = cur + l31 * LDT;
            const float* b_row = cur + (32 + q * 32 + l31) * LDT;
#pragma unroll
            for (int grp = 0; grp < BK / 16; ++grp) mfma_group(acc, a_row, b_row, grp, hh);
            if (kt + 1 < NK) {
                float* nxt = lds + ((kt + 1) & 1) * ROWS * LDT;
#pragma unroll
                for (int i = 0; i < PER; ++i) if (dst[i] >= 0) st4(nxt + dst[i], stage[i]);
            }
            __syncthreads();
        }
    }
    // gate tiles meet in LDS: P[gate][segment row][unit]
    float* P = lds;
    acc_to_lds(acc, P + q * 32 * PLD, l31, hh);
    __syncthreads();

#pragma unroll
    for (int ii = 0; ii < NI; ++ii) {
        if (it_state[ii] == 0) continue;
        const int item = tid + NT * ii;
        const int rr = item >> 3, uq = 4 * (item & 7);
        const int j = row0 + rr, u = u0 + uq;
        const int64_t seg_beg = (int64_t)j * g.Ls;
        const int64_t pos = it_pos[ii];
        if (it_state[ii] == 1) {
            st4(h_nxt + (int64_t)j * H + u, zero4());
            if (G == 4) st4(c_nxt + (int64_t)j * H + u, zero4());
            continue;
        }
        const int s = it_s[ii], t = it_t[ii], i = it_i[ii];
        const float* pr = P + rr * PLD + uq;
        float4 h;
        if (G == 4) {
            float4 gi = add4(ldp4(pr), px[ii][0]), gf = add4(ldp4(pr + 32 * PLD), px[ii][1]);
            float4 gg = add4(ldp4(pr + 64 * PLD), px[ii][2]), go = add4(ldp4(pr + 96 * PLD), px[ii][G - 1]);
            const float4 cp = pprev[ii];
            float4 c;
#define CELL(k)                                                              \
    gi.k = sg(gi.k); gf.k = sg(gf.k); gg.k = th(gg.k); go.k = sg(go.k);      \
    c.k = gf.k * cp.k + gi.k * gg.k;                                         \
    h.k = go.k * th(c.k);
            CELL(x) CELL(y) CELL(z) CELL(w)
#undef CELL
            st4(c_nxt + (int64_t)j * H + u, c);
            if (pos >= seg_beg && saved) {
                float* sp = saved + pos * (6 * H) + u;
                st4(sp, gi); st4(sp + H, gf); st4(sp + 2 * H, gg); st4(sp + 3 * H, go); st4(sp + 4 * H, c); st4(sp + 5 * H, h);
            }
            if (pos == seg_beg - 1) st4(seam_a + (int64_t)j * 2 * H + H + u, c);
            if (pos == seg_beg + g.Ls - 1 && j + 1 < g.K) st4(seam_b + (int64_t)(j + 1) * 2 * H + H + u, c);
        } else {
            const float4 xr_ = px[ii][0], xz = px[ii][1], xn = px[ii][2];
            const float4 hr_ = add4(ldp4(pr), ld4(b_hh + u)), hz = add4(ldp4(pr + 32 * PLD), ld4(b_hh + H + u));
            const float4 hn = add4(ldp4(pr + 64 * PLD), ld4(b_hh + 2 * H + u));
            const float4 hp = pprev[ii];
#define CELL(k)                                              \
    {                                                        \
        const float r = sg(xr_.k + hr_.k);                   \
        const float z = sg(xz.k + hz.k);                     \
        const float nn = th(xn.k + r * hn.k);                \
        h.k = (1.f - z) * nn + z * hp.k;                     \
    }
            CELL(x) CELL(y) CELL(z) CELL(w)
#undef CELL
        }
        st4(h_nxt + (int64_t)j * H + u, h);
        if (pos >= seg_beg && t == g.p0 + s - 1) st4(h_emit + ((int64_t)s * g.n + i) * H + u, h);
        constexpr int SW = (G == 4) ? 2 * H : H;  // seam record: {h, c} or {h}
        if (pos == seg_beg - 1) st4(seam_a + (int64_t)j * SW + u, h);
        if (pos == seg_beg + g.Ls - 1 && j + 1 < g.K) st4(seam_b + (int64_t)(j + 1) * SW + u, h);
    }
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

// ------------------------------------------------------------------ LSTM backward step
// State entering position p (walking down): dh = W_hh^T dgates(p+1) and dc (carried).
// Step mp handles pos = seg_beg + Ls + W - 1 - mp.  final = 1: only the product, written
// to dh_out (the state below each segment's first position, for the seam check).
__global__ __launch_bounds__(256) void seg_bwd_step(const float* __restrict__ dg_prev, const float* __restrict__ w_hh,
                                                    const float* __restrict__ saved, const float* __restrict__ d_emit,
                                                    float* __restrict__ dc, float* __restrict__ dg_next,
                                                    float* __restrict__ dgv, float* __restrict__ seam_a,
                                                    float* __restrict__ dh_out, Geom g, int mp, int final) {
    constexpr int A_T = 32 * LDT;              // per wave: 32 segment rows x 64 k (k contiguous)
    constexpr int B_T = BK * 32;               // per wave: 64 k x 32 units (unit contiguous)
    constexpr int STAGE = 4 * (A_T + B_T);
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    // blocks x, x+8, ... share an XCD (round-robin dispatch): giving them the same unit tiles keeps that
    // XCD's slice of W_hh (1/8 of it) resident in its L2 across the segment-row tiles and across steps
    const int row0 = blockIdx.y * 32, u0 = blockIdx.x * 32;

    // this thread's epilogue item and its operands (saved activations, carried dc, injected gradient),
    // fetched before the k loop so that their HBM latency hides under it
    const int rr = tid >> 3, uq = 4 * (tid & 7);
    const int j = row0 + rr, u = u0 + uq;
    const int64_t seg_beg = (int64_t)j * g.Ls;
    const int64_t pos = seg_beg + g.Ls + g.W - 1 - mp;
    const bool have_row = j < g.K;
    const bool live = have_row && !final && pos < g.total && seg_beg < g.total;
    float4 gi = zero4(), gf = zero4(), gg = zero4(), go = zero4(), cc = zero4(), cp = zero4(), de = zero4();
    float4 dcv = zero4();
    if (have_row && !final) dcv = ld4(dc + (int64_t)j * H + u);
    if (live) {
        int s, t, i;
        locate(g, pos, s, t, i);
        const float* sp = saved + pos * (6 * H) + u;
        gi = ld4(sp); gf = ld4(sp + H); gg = ld4(sp + 2 * H); go = ld4(sp + 3 * H); cc = ld4(sp + 4 * H);
        if (pos > 0) cp = ld4(saved + (pos - 1) * (6 * H) + 4 * H + u);
        if (t == g.p0 + s - 1) de = ld4(d_emit + ((int64_t)s * g.n + i) * H + u);
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    if (mp > 0) {
        // staging: A (dgates rows) 4 waves x 32 rows x 16 float4; B (W_hh rows k) 4 waves x 64 k x 8 float4
        const float* asrc[8];
        int adst[8];
        const float* bsrc[8];
        int bdst[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = tid + 256 * i;                 // 0..2047
            const int ww = f >> 9, r = (f >> 4) & 31, kq = f & 15;
            asrc[i] = dg_prev + (int64_t)min(row0 + r, g.K - 1) * (4 * H) + ww * H + 4 * kq;
            adst[i] = ww * (A_T + B_T) + r * LDT + 4 * kq;
            const int kk = (f >> 3) & 63, uq = f & 7;
            bsrc[i] = w_hh + (int64_t)(ww * H + kk) * H + u0 + 4 * uq;
            bdst[i] = ww * (A_T + B_T) + A_T + kk * 32 + 4 * uq;
        }
        float4 sa[8], sb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { sa[i] = ld4(asrc[i]); sb[i] = ld4(bsrc[i]); }
#pragma unroll
        for (int i = 0; i < 8; ++i) { st4(lds + adst[i], sa[i]); st4(lds + bdst[i], sb[i]); }
        __syncthreads();
        constexpr int NK = H / BK;  // each wave walks its own 512 gate rows
        for (int kt = 0; kt < NK; ++kt) {
            const float* cur = lds + (kt & 1) * STAGE + w * (A_T + B_T);
            if (kt + 1 < NK) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    sa[i] = ld4(asrc[i] + (kt + 1) * BK);
                    sb[i] = ld4(bsrc[i] + (int64_t)(kt + 1) * BK * H);
                }
            }
            const float* a_row = cur + l31 * LDT;
            const float* b_t = cur + A_T;
#pragma unroll
            for (int grp = 0; grp < BK / 16; ++grp) {
                const float4 a0 = ld4(a_row + 16 * grp + 8 * hh), a1 = ld4(a_row + 16 * grp + 8 * hh + 4);
                const float* bk = b_t + (16 * grp + 8 * hh) * 32 + l31;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, bk[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, bk[32], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, bk[64], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, bk[96], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, bk[128], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, bk[160], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, bk[192], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, bk[224], acc, 0, 0, 0);
            }
            if (kt + 1 < NK) {
                float* nxt = lds + ((kt + 1) & 1) * STAGE;
#pragma unroll
                for (int i = 0; i < 8; ++i) { st4(nxt + adst[i], sa[i]); st4(nxt + bdst[i], sb[i]); }
            }
            __syncthreads();
        }
    }
    // the four waves' partial [segment row][unit] tiles meet in LDS
    float* P = lds;
    acc_to_lds(acc, P + w * 32 * PLD, l31, hh);
    __syncthreads();

    if (!have_row) return;
    const float* pr = P + rr * PLD + uq;
    float4 dh = add4(add4(ldp4(pr), ldp4(pr + 32 * PLD)), add4(ldp4(pr + 64 * PLD), ldp4(pr + 96 * PLD)));
    if (final) {
        st4(dh_out + (int64_t)j * H + u, dh);
        return;
    }
    if (pos == seg_beg + g.Ls - 1) {  // state entering the segment's last own position
        st4(seam_a + (int64_t)j * 2 * H + u, dh);
        st4(seam_a + (int64_t)j * 2 * H + H + u, dcv);
    }
    float* dgo = dg_next + (int64_t)j * (4 * H) + u;
    if (!live) {
        st4(dgo, zero4()); st4(dgo + H, zero4()); st4(dgo + 2 * H, zero4()); st4(dgo + 3 * H, zero4());
        st4(dc + (int64_t)j * H + u, zero4());
        return;
    }
    dh = add4(dh, de);
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
    st4(dc + (int64_t)j * H + u, dcn);
    if (pos < seg_beg + g.Ls) {
        float* dp = dgv + pos * (4 * H) + u;
        st4(dp, d_i); st4(dp + H, d_f); st4(dp + 2 * H, d_g); st4(dp + 3 * H, d_o);
    }
}

// seam j (between segments j and j+1): warm-up state of segment j vs final state of j+1;
// err_abs = max |a - b|, ref_abs = max |b| over dh and dc of all seams.
__global__ __launch_bounds__(256) void seam_check_bwd(const float* __restrict__ seam_a, const float* __restrict__ dh_fin,
                                                      const float* __restrict__ dc, Geom g,
                                                      float* __restrict__ err_abs, float* __restrict__ ref_abs) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    float d = 0.f, r = 0.f;
    if (e < g.K * 2 * H) {
        const int j = e / (2 * H), k = e % (2 * H);
        if (j + 1 < g.K && (int64_t)(j + 1) * g.Ls + g.W < g.total) {
            const float b = (k < H) ? dh_fin[(int64_t)(j + 1) * H + k] : dc[(int64_t)(j + 1) * H + (k - H)];
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

bool aligned16(const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; }

}  // namespace

extern "C" int64_t icrl_chain_seg_workspace_floats(int segments) { return (int64_t)segments * 16 * H; }

extern "C" int icrl_lstm_chain_fwd_seg(const float* xproj, const float* w_hh, int N, int p0, int n_calls,
                                       int segments, int warmup, float* h_emit, float* saved, float* work,
                                       float* seam_err, icrl_stream_t stream) {
    Geom g;
    if (int rc = make_geom("icrl_lstm_chain_fwd_seg", N, p0, n_calls, segments, warmup, g)) return rc;
    ICRL_REQUIRE(xproj && w_hh && h_emit && work && seam_err, "icrl_lstm_chain_fwd_seg: null pointer");
    ICRL_REQUIRE(aligned16(xproj) && aligned16(w_hh) && aligned16(h_emit) && aligned16(saved) && aligned16(work),
                 "icrl_lstm_chain_fwd_seg: buffers must be 16-byte aligned");
    hipStream_t st = icrl::as_stream(stream);
    const int64_t kh = (int64_t)g.K * H;
    float* hb[2] = {work, work + kh};
    float* cb[2] = {work + 2 * kh, work + 3 * kh};
    float* seam_a = work + 4 * kh;        // [K, 2H]
    float* seam_b = work + 6 * kh;        // [K, 2H]
    ICRL_HIP(hipMemsetAsync(work, 0, sizeof(float) * 4 * kh, st));
    ICRL_HIP(hipMemsetAsync(seam_err, 0, sizeof(float), st));
    const dim3 grid(H / 32, (g.K + 31) / 32);
    const int steps = g.W + g.Ls;
    for (int m = 0; m < steps; ++m) {
        const int cur = m & 1;
        hipLaunchKernelGGL(seg_fwd_step<4>, grid, dim3(256), 0, st, hb[cur], w_hh, xproj, (const float*)nullptr,
                           cb[cur], cb[cur ^ 1], hb[cur ^ 1], saved, h_emit, seam_a, seam_b, g, m);
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
    ICRL_REQUIRE(aligned16(xproj) && aligned16(w_hh) && aligned16(b_hh) && aligned16(h_emit) && aligned16(work),
                 "icrl_gru_chain_fwd_seg: buffers must be 16-byte aligned");
    hipStream_t st = icrl::as_stream(stream);
    const int64_t kh = (int64_t)g.K * H;
    float* hb[2] = {work, work + kh};
    float* seam_a = work + 2 * kh;        // [K, H]
    float* seam_b = work + 3 * kh;        // [K, H]
    ICRL_HIP(hipMemsetAsync(work, 0, sizeof(float) * 2 * kh, st));
    ICRL_HIP(hipMemsetAsync(seam_err, 0, sizeof(float), st));
    const dim3 grid(H / 32, (g.K + 31) / 32);
    const int steps = g.W + g.Ls;
    for (int m = 0; m < steps; ++m) {
        const int cur = m & 1;
        hipLaunchKernelGGL(seg_fwd_step<3>, grid, dim3(192), 0, st, hb[cur], w_hh, xproj, b_hh, (const float*)nullptr,
                           (float*)nullptr, hb[cur ^ 1], (float*)nullptr, h_emit, seam_a, seam_b, g, m);
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
    ICRL_REQUIRE(aligned16(w_hh) && aligned16(saved) && aligned16(d_emit) && aligned16(dgates) && aligned16(work),
                 "icrl_lstm_chain_bwd_seg: buffers must be 16-byte aligned");
    hipStream_t st = icrl::as_stream(stream);
    const int64_t kh = (int64_t)g.K * H;
    float* dg[2] = {work, work + 4 * kh}; // [K, 4H] each
    float* dc = work + 8 * kh;            // [K, H]
    float* seam_a = work + 9 * kh;        // [K, 2H]
    float* dh_fin = work + 11 * kh;       // [K, H]
    ICRL_HIP(hipMemsetAsync(dc, 0, sizeof(float) * 3 * kh, st));
    ICRL_HIP(hipMemsetAsync(seam_err, 0, 2 * sizeof(float), st));
    const dim3 grid(H / 32, (g.K + 31) / 32);
    const int steps = g.W + g.Ls;
    for (int mp = 0; mp < steps; ++mp)
        hipLaunchKernelGGL(seg_bwd_step, grid, dim3(256), 0, st, dg[(mp & 1) ^ 1], w_hh, saved, d_emit, dc, dg[mp & 1],
                           dgates, seam_a, dh_fin, g, mp, 0);
    hipLaunchKernelGGL(seg_bwd_step, grid, dim3(256), 0, st, dg[(steps & 1) ^ 1], w_hh, saved, d_emit, dc,
                       dg[steps & 1], dgates, seam_a, dh_fin, g, steps, 1);
    hipLaunchKernelGGL(seam_check_bwd, dim3((g.K * 2 * H + 255) / 256), dim3(256), 0, st, seam_a, dh_fin, dc, g,
                       seam_err, seam_err + 1);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}
