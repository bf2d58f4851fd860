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
#include <algorithm>
#include <atomic>

#include "chain_seg_common.h"

namespace {
using namespace icrl;
using namespace icrl::seg;

// The cells, with the fused multiply-adds written out so that every kernel in this file rounds alike.
#define LSTM_CELL1(k, gi, gf, gg, go, cp, c, h)                              \
    gi.k = sg(gi.k); gf.k = sg(gf.k); gg.k = th(gg.k); go.k = sg(go.k);      \
    c.k = __fmaf_rn(gf.k, cp.k, gi.k * gg.k);                                \
    h.k = go.k * th(c.k);
#define LSTM_CELL4(gi, gf, gg, go, cp, c, h) \
    LSTM_CELL1(x, gi, gf, gg, go, cp, c, h) LSTM_CELL1(y, gi, gf, gg, go, cp, c, h) \
    LSTM_CELL1(z, gi, gf, gg, go, cp, c, h) LSTM_CELL1(w, gi, gf, gg, go, cp, c, h)
#define GRU_CELL1(k, xr, xz, xn, hr, hz, hn, hp, h)                          \
    {                                                                        \
        const float r_ = sg(xr.k + hr.k);                                    \
        const float z_ = sg(xz.k + hz.k);                                    \
        const float n_ = th(__fmaf_rn(r_, hn.k, xn.k));                      \
        h.k = __fmaf_rn(z_, hp.k - n_, n_);                                  \
    }
#define GRU_CELL4(xr, xz, xn, hr, hz, hn, hp, h) \
    GRU_CELL1(x, xr, xz, xn, hr, hz, hn, hp, h) GRU_CELL1(y, xr, xz, xn, hr, hz, hn, hp, h) \
    GRU_CELL1(z, xr, xz, xn, hr, hz, hn, hp, h) GRU_CELL1(w, xr, xz, xn, hr, hz, hn, hp, h)

#define GRU_GATES1(k, xr, xz, xn, hr, hz, hn, r4, z4, n4)                     \
    r4.k = sg(xr.k + hr.k); z4.k = sg(xz.k + hz.k); n4.k = th(__fmaf_rn(r4.k, hn.k, xn.k));
#define GRU_GATES4(xr, xz, xn, hr, hz, hn, r4, z4, n4) \
    GRU_GATES1(x, xr, xz, xn, hr, hz, hn, r4, z4, n4) GRU_GATES1(y, xr, xz, xn, hr, hz, hn, r4, z4, n4) \
    GRU_GATES1(z, xr, xz, xn, hr, hz, hn, r4, z4, n4) GRU_GATES1(w, xr, xz, xn, hr, hz, hn, r4, z4, n4)
// GRU cell backward for one component: DH = total gradient of the position's output h
#define GRU_BWD1(k, DH, GR, GZ, GN, GHN, HP, DR, DZ, DN, DHN, CY)            \
    {                                                                        \
        DN.k = DH.k * (1.f - GZ.k) * __fmaf_rn(-GN.k, GN.k, 1.f);            \
        DZ.k = DH.k * (HP.k - GN.k) * GZ.k * (1.f - GZ.k);                   \
        DR.k = DN.k * GHN.k * GR.k * (1.f - GR.k);                           \
        DHN.k = DN.k * GR.k;                                                 \
        CY.k = DH.k * GZ.k;                                                  \
    }
#define GRU_BWD4(DH, GR, GZ, GN, GHN, HP, DR, DZ, DN, DHN, CY)                \
    GRU_BWD1(x, DH, GR, GZ, GN, GHN, HP, DR, DZ, DN, DHN, CY)                 \
    GRU_BWD1(y, DH, GR, GZ, GN, GHN, HP, DR, DZ, DN, DHN, CY)                 \
    GRU_BWD1(z, DH, GR, GZ, GN, GHN, HP, DR, DZ, DN, DHN, CY)                 \
    GRU_BWD1(w, DH, GR, GZ, GN, GHN, HP, DR, DZ, DN, DHN, CY)

#define LSTM_BWD1(k, dh, dcv, gi, gf, gg, go, cc, cp, d_i, d_f, d_g, d_o, dcn)  \
    {                                                                           \
        const float tc = th(cc.k);                                              \
        const float dcc = __fmaf_rn(dh.k * go.k, __fmaf_rn(-tc, tc, 1.f), dcv.k); \
        d_i.k = dcc * gg.k * gi.k * (1.f - gi.k);                               \
        d_f.k = dcc * cp.k * gf.k * (1.f - gf.k);                               \
        d_g.k = dcc * gi.k * __fmaf_rn(-gg.k, gg.k, 1.f);                       \
        d_o.k = dh.k * tc * go.k * (1.f - go.k);                                \
        dcn.k = dcc * gf.k;                                                     \
    }
#define LSTM_BWD4(dh, dcv, gi, gf, gg, go, cc, cp, d_i, d_f, d_g, d_o, dcn)                    \
    LSTM_BWD1(x, dh, dcv, gi, gf, gg, go, cc, cp, d_i, d_f, d_g, d_o, dcn)                     \
    LSTM_BWD1(y, dh, dcv, gi, gf, gg, go, cc, cp, d_i, d_f, d_g, d_o, dcn)                     \
    LSTM_BWD1(z, dh, dcv, gi, gf, gg, go, cc, cp, d_i, d_f, d_g, d_o, dcn)                     \
    LSTM_BWD1(w, dh, dcv, gi, gf, gg, go, cc, cp, d_i, d_f, d_g, d_o, dcn)

// ------------------------------------------------------------------ forward step (LSTM: G = 4, GRU: G = 3)
// GRU (three gate tiles): a wave per gate would leave a SIMD idle, so all four waves work on every
// 64-k stage instead, wave q taking the 16-k group q of each of the three gates (three accumulators,
// 24 MFMAs per stage and wave instead of 32); the waves' partial tiles are summed in LDS.
struct FragK { float4 a0, a1, b[3][2]; };
__device__ __forceinline__ void mfma8g(f32x16& acc, const FragK& f, int gq) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0.x, f.b[gq][0].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0.y, f.b[gq][0].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0.z, f.b[gq][0].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0.w, f.b[gq][0].w, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1.x, f.b[gq][1].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1.y, f.b[gq][1].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1.z, f.b[gq][1].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1.w, f.b[gq][1].w, acc, 0, 0, 0);
}
// a_row / b_row: this lane's rows of the stage (h row l31; W_hh row l31 of gate 0), `gate` floats between
// gates, o0 / o1: float offsets of the lane's two 16-byte chunks within a row
__device__ __forceinline__ FragK load_frag_k(const float* a_row, const float* b_row, int gate, int o0, int o1) {
    FragK f;
    f.a0 = ld4(a_row + o0); f.a1 = ld4(a_row + o1);
#pragma unroll
    for (int gq = 0; gq < 3; ++gq) { f.b[gq][0] = ld4(b_row + gq * gate + o0); f.b[gq][1] = ld4(b_row + gq * gate + o1); }
    return f;
}
// sum of the four waves' partial tiles of one gate: P[wave][gate][32][PLD]
__device__ __forceinline__ float4 sum_partials(const float* p) {
    constexpr int W = 3 * 32 * PLD;
    return add4(add4(ldp4(p), ldp4(p + W)), add4(ldp4(p + 2 * W), ldp4(p + 3 * W)));
}

template <int G>
__global__ __launch_bounds__(256) void seg_fwd_step(const float* __restrict__ h_cur, const float* __restrict__ w_hh,
                                                       const float* __restrict__ xproj, const float* __restrict__ b_hh,
                                                       const float* __restrict__ c_cur, float* __restrict__ c_nxt,
                                                       float* __restrict__ h_nxt, float* __restrict__ saved,
                                                       float* __restrict__ h_emit, float* __restrict__ seam_a,
                                                       float* __restrict__ seam_b, const float* __restrict__ state_in,
                                                       Geom g, int m) {
    constexpr int NT = 256;                    // threads
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
        const int64_t pos = seg_pos(g, j, m - g.W);
        it_pos[ii] = pos;
        if (pos < 0 || m - g.W >= g.Ls || pos >= g.total) { it_state[ii] = 1; continue; }
        it_state[ii] = 2;
        locate_fast(g, (int)pos, it_s[ii], it_t[ii], it_i[ii]);
        const float* xr = xproj + ((int64_t)it_t[ii] * g.n + it_i[ii]) * (G * H) + u;
#pragma unroll
        for (int gq = 0; gq < G; ++gq) px[ii][gq] = ld4(xr + gq * H);
        pprev[ii] = ld4((G == 4 ? c_cur : h_cur) + (int64_t)j * H + u);
    }

    // k loop, software-pipelined so that the MFMA pipe never drains: global loads run DEPTH stages
    // ahead in a register ring (fully unrolled, every ring slot static); operand fragments for the
    // next 16-k group are read from LDS before the current group's MFMAs issue; the one barrier per
    // stage sits between the last two groups, after every read of the current buffer has been issued,
    // so the wait for the other waves hides under eight MFMAs already in flight.
    constexpr int NK = H / BK;
    constexpr int DEPTH = 3;
    float4 ring[DEPTH][PER];
    constexpr int NACC = (G == 3) ? 3 : 1;
    f32x16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    if (m > 0) {  // the first step starts from a zero state: nothing to multiply
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int i = 0; i < PER; ++i) ring[d][i] = ld4(src[i] + d * BK);
#pragma unroll
        for (int i = 0; i < PER; ++i) if (dst[i] >= 0) st4(lds + dst[i], ring[0][i]);
        if (DEPTH < NK) {
#pragma unroll
            for (int i = 0; i < PER; ++i) ring[0][i] = ld4(src[i] + DEPTH * BK);
        }
        __syncthreads();
        constexpr int BUF = ROWS * LDT;
        const float* a_row = lds + l31 * LDT;
        if constexpr (G == 3) {
            const float* b_row = lds + (32 + l31) * LDT;
            const int o0 = 16 * q + 8 * hh, o1 = o0 + 4;
            FragK f = load_frag_k(a_row, b_row, 32 * LDT, o0, o1), fn;
#pragma unroll
            for (int kt = 0; kt < NK; ++kt) {
                const int cn = ((kt + 1) & 1) * BUF;
                if (kt + 1 < NK) {   // the previous stage's barrier came after the last read of buffer `cn`
#pragma unroll
                    for (int i = 0; i < PER; ++i) if (dst[i] >= 0) st4(lds + cn + dst[i], ring[(kt + 1) % DEPTH][i]);
                    if (kt + 1 + DEPTH < NK) {
#pragma unroll
                        for (int i = 0; i < PER; ++i) ring[(kt + 1) % DEPTH][i] = ld4(src[i] + (kt + 1 + DEPTH) * BK);
                    }
                }
                mfma8g(acc[0], f, 0);
                mfma8g(acc[1], f, 1);
                if (kt + 1 < NK) {
                    __syncthreads();
                    fn = load_frag_k(a_row + cn, b_row + cn, 32 * LDT, o0, o1);
                }
                mfma8g(acc[2], f, 2);
                f = fn;
            }
        } else {
            const float* b_row = lds + (32 + q * 32 + l31) * LDT;
            Frag f0 = load_frag(a_row, b_row, 0, hh), f1;
#pragma unroll
            for (int kt = 0; kt < NK; ++kt) {
                const int co = (kt & 1) * BUF, cn = ((kt + 1) & 1) * BUF;
                f1 = load_frag(a_row + co, b_row + co, 1, hh);
                mfma8(acc[0], f0);
                if (kt + 1 < NK) {   // nobody reads buffer `cn` any more: the previous stage's barrier came after its last read
#pragma unroll
                    for (int i = 0; i < PER; ++i) if (dst[i] >= 0) st4(lds + cn + dst[i], ring[(kt + 1) % DEPTH][i]);
                    if (kt + 1 + DEPTH < NK) {
#pragma unroll
                        for (int i = 0; i < PER; ++i) ring[(kt + 1) % DEPTH][i] = ld4(src[i] + (kt + 1 + DEPTH) * BK);
                    }
                }
                f0 = load_frag(a_row + co, b_row + co, 2, hh);
                mfma8(acc[0], f1);
                f1 = load_frag(a_row + co, b_row + co, 3, hh);
                mfma8(acc[0], f0);
                if (kt + 1 < NK) {
                    __syncthreads();
                    f0 = load_frag(a_row + cn, b_row + cn, 0, hh);
                }
                mfma8(acc[0], f1);
            }
        }
        __syncthreads();  // every wave is done reading before the gate tiles overlay the stage buffers
    }
    // gate tiles meet in LDS: LSTM P[gate][segment row][unit]; GRU P[wave][gate][segment row][unit] partials
    float* P = lds;
    if constexpr (G == 3) {
#pragma unroll
        for (int gq = 0; gq < 3; ++gq) acc_to_lds(acc[gq], P + (q * 3 + gq) * 32 * PLD, l31, hh);
    } else {
        acc_to_lds(acc[0], P + q * 32 * PLD, l31, hh);
    }
    __syncthreads();

#pragma unroll
    for (int ii = 0; ii < NI; ++ii) {
        if (it_state[ii] == 0) continue;
        const int item = tid + NT * ii;
        const int rr = item >> 3, uq = 4 * (item & 7);
        const int j = row0 + rr, u = u0 + uq;
        const int qs = m - g.W;     // step within the segment: negative during the warm-up
        const int64_t pos = it_pos[ii];
        if (it_state[ii] == 1) {
            // outside the segment's range the state is zero - except that the chain itself may start from a
            // given state {h, c}: every segment whose warm-up reaches back to the chain's start picks it up on
            // the step before position 0, and is exact from there
            float4 h0 = zero4(), c0 = zero4();
            if (state_in && pos == -1) {
                h0 = ld4(state_in + u);
                if (G == 4) c0 = ld4(state_in + H + u);
            }
            st4(h_nxt + (int64_t)j * H + u, h0);
            if (G == 4) st4(c_nxt + (int64_t)j * H + u, c0);
            continue;
        }
        const int s = it_s[ii], t = it_t[ii], i = it_i[ii];
        const float* pr = P + rr * PLD + uq;
        float4 h;
        if (G == 4) {
            float4 gi = add4(ldp4(pr), px[ii][0]), gf = add4(ldp4(pr + 32 * PLD), px[ii][1]);
            float4 gg = add4(ldp4(pr + 64 * PLD), px[ii][2]), go = add4(ldp4(pr + 96 * PLD), px[ii][G - 1]);
            float4 c;
            LSTM_CELL4(gi, gf, gg, go, pprev[ii], c, h)
            st4(c_nxt + (int64_t)j * H + u, c);
            if (qs >= 0 && saved) {
                float* sp = saved + pos * (6 * H) + u;
                st4(sp, gi); st4(sp + H, gf); st4(sp + 2 * H, gg); st4(sp + 3 * H, go); st4(sp + 4 * H, c); st4(sp + 5 * H, h);
            }
            if (qs == -1) st4(seam_a + (int64_t)j * 2 * H + H + u, c);
            if (qs == g.Ls - 1 && j + 1 < g.K) st4(seam_b + (int64_t)(j + 1) * 2 * H + H + u, c);
        } else {
            const float4 xr_ = px[ii][0], xz = px[ii][1], xn = px[ii][2];
            const float4 hr_ = add4(sum_partials(pr), ld4(b_hh + u));
            const float4 hz = add4(sum_partials(pr + 32 * PLD), ld4(b_hh + H + u));
            const float4 hn = add4(sum_partials(pr + 64 * PLD), ld4(b_hh + 2 * H + u));
            GRU_CELL4(xr_, xz, xn, hr_, hz, hn, pprev[ii], h)
            if (qs >= 0 && saved) {   // {r, z, n, hn, h}: the same expressions as in the cell, so the same bits
                float4 r4, z4, n4;
                GRU_GATES4(xr_, xz, xn, hr_, hz, hn, r4, z4, n4)
                float* sp = saved + pos * (5 * H) + u;
                st4(sp, r4); st4(sp + H, z4); st4(sp + 2 * H, n4); st4(sp + 3 * H, hn); st4(sp + 4 * H, h);
            }
        }
        st4(h_nxt + (int64_t)j * H + u, h);
        if (qs >= 0 && t == g.p0 + s - 1) st4(h_emit + ((int64_t)s * g.n + i) * H + u, h);
        constexpr int SW = (G == 4) ? 2 * H : H;  // seam record: {h, c} or {h}
        if (qs == -1) st4(seam_a + (int64_t)j * SW + u, h);
        if (qs == g.Ls - 1 && j + 1 < g.K) st4(seam_b + (int64_t)(j + 1) * SW + u, h);
    }
}

// max over seams j >= 1 of |a - b| / max(1, |b|), `width` floats per seam.  Grid-stride, one atomic
// per workgroup (thousands of atomics on one word cost ~100 us).
__global__ __launch_bounds__(256) void seam_check_fwd(const float* __restrict__ seam_a, const float* __restrict__ seam_b,
                                                      Geom g, int width, float* __restrict__ err) {
    __shared__ float red[4];
    float d = 0.f;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < g.K * width; e += gridDim.x * 256) {
        const int j = e / width;
        // seam j exists when segment j has positions and was warmed up from a zero state
        if (j >= 1 && (int64_t)j * g.Ls < g.total && (int64_t)j * g.Ls - g.W > 0) {
            const float a = seam_a[e], b = seam_b[e];
            float v = fabsf(a - b) / fmaxf(1.f, fabsf(b));
            if (!(v == v)) v = INFINITY;
            d = fmaxf(d, v);
        }
    }
    d = wave_max(d);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        d = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (d > 0.f) atomic_max_nonneg(err, d);
    }
}

// What a persistent launch leaves behind when one of its waits runs out (give_up, below): abort word, then the
// workgroup that gave up, the step it was at and a bit mask of the producers of its row group that had not reached
// that step.  One line on stdout per such launch (device printf from the one-thread / first-thread kernels that run
// behind every persistent launch, not from the chain kernels, whose register budgets it would cost).
__device__ void report_abort(const unsigned* abort_w, int kind) {
    if (*abort_w == 0) return;
    static const char* const names[] = {"forward chain pair", "value chain BPTT", "replay check", "forward chain",
                                        "per-row chains", "policy LSTM BPTT"};
    (void)names;
    printf("[icrl] a persistent launch (kind %d: 0 forward pair, 1 value BPTT, 2 replay check, 3 one forward chain, "
           "4 per-row chains, 5 policy BPTT) gave up: workgroup %u had waited 20 ms at step %u for workgroups of its "
           "row group, mask %08x (bit i: producer i behind; all bits set at step 1 = they never started, i.e. other "
           "work held the CUs the launch needs)\n", kind, abort_w[1], abort_w[2], abort_w[3]);
}

// Both forward chains' seam checks and the abort poison in ONE launch (three tiny launches in a row sit on the
// step's critical path): blocks [0, nb) check the LSTM chain's seams, blocks [nb, 2 nb) the GRU chain's; a launch
// that gave up waiting (abort word set) reports +inf for both.
__global__ __launch_bounds__(256) void seam_check_fwd_pair(const float* __restrict__ a_v, const float* __restrict__ b_v,
                                                           Geom gv, float* __restrict__ err_v,
                                                           const float* __restrict__ a_r, const float* __restrict__ b_r,
                                                           Geom gr, float* __restrict__ err_r,
                                                           const unsigned* __restrict__ abort_w, int nb) {
    __shared__ float red[4];
    const bool second = (int)blockIdx.x >= nb;
    const float* seam_a = second ? a_r : a_v;
    const float* seam_b = second ? b_r : b_v;
    const Geom& g = second ? gr : gv;
    const int width = second ? H : 2 * H;
    float* err = second ? err_r : err_v;
    const int b0 = second ? blockIdx.x - nb : blockIdx.x;
    float d = 0.f;
    for (int e = b0 * 256 + threadIdx.x; e < g.K * width; e += nb * 256) {
        const int j = e / width;
        if (j >= 1 && (int64_t)j * g.Ls < g.total && (int64_t)j * g.Ls - g.W > 0) {
            const float a = seam_a[e], b = seam_b[e];
            float v = fabsf(a - b) / fmaxf(1.f, fabsf(b));
            if (!(v == v)) v = INFINITY;
            d = fmaxf(d, v);
        }
    }
    d = wave_max(d);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        d = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (*abort_w != 0) d = INFINITY;
        if (blockIdx.x == 0) report_abort(abort_w, 0);
        if (d > 0.f) atomic_max_nonneg(err, d);
    }
}

// ------------------------------------------------------------------ LSTM backward step
// State entering position p (walking down): dh = W_hh^T dgates(p+1) and dc (carried).
// Step mp handles step Ls + W - 1 - mp of every segment.  final = 1: only the product, written
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
    const int qs = g.Ls + g.W - 1 - mp;   // step within the segment: >= Ls during the warm-up
    const int64_t pos = seg_pos(g, j, qs);
    const bool have_row = j < g.K;
    const bool live = have_row && !final && pos < g.total && seg_pos(g, j, 0) < g.total;
    float4 gi = zero4(), gf = zero4(), gg = zero4(), go = zero4(), cc = zero4(), cp = zero4(), de = zero4();
    float4 dcv = zero4();
    if (have_row && !final) dcv = ld4(dc + (int64_t)j * H + u);
    if (live) {
        int s, t, i;
        locate_fast(g, (int)pos, s, t, i);
        const float* sp = saved + pos * (6 * H) + u;
        gi = ld4(sp); gf = ld4(sp + H); gg = ld4(sp + 2 * H); go = ld4(sp + 3 * H); cc = ld4(sp + 4 * H);
        if (pos >= g.stride) cp = ld4(saved + (pos - g.stride) * (6 * H) + 4 * H + u);
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
    if (qs == g.Ls - 1) {  // state entering the segment's last own position
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
    LSTM_BWD4(dh, dcv, gi, gf, gg, go, cc, cp, d_i, d_f, d_g, d_o, dcn)
    st4(dgo, d_i); st4(dgo + H, d_f); st4(dgo + 2 * H, d_g); st4(dgo + 3 * H, d_o);
    st4(dc + (int64_t)j * H + u, dcn);
    if (qs < g.Ls) {
        float* dp = dgv + pos * (4 * H) + u;
        st4(dp, d_i); st4(dp + H, d_f); st4(dp + 2 * H, d_g); st4(dp + 3 * H, d_o);
    }
}

// ------------------------------------------------------------------ GRU backward step
// The same scheme for the reward GRU (the BPTT the reward pre-training loop needs, trainers.py:288-300).
// State entering a position (walking down): P = W_hh^T dgates_h(pos+1), and the carry dh(pos+1) * z(pos+1);
// dh(pos) = P + carry + injected gradient.  The four waves split the 1536 rows of dgates_h (384 each).
// Outputs per position: dgates_x = {dr, dz, dn} (gradient of x W_ih^T + b_ih) and dgates_h = {dr, dz, r dn}
// (gradient of h W_hh^T + b_hh).  saved = {r, z, n, hn, h} per position.
__global__ __launch_bounds__(256) void seg_gru_bwd_step(const float* __restrict__ dg_prev, const float* __restrict__ w_hh,
                                                        const float* __restrict__ saved, const float* __restrict__ d_emit,
                                                        float* __restrict__ carry, float* __restrict__ dg_next,
                                                        float* __restrict__ dgx, float* __restrict__ dgh,
                                                        float* __restrict__ seam_a, float* __restrict__ dh_out, Geom g,
                                                        int mp, int final) {
    constexpr int A_T = 32 * LDT, B_T = BK * 32, STAGE = 4 * (A_T + B_T);
    constexpr int KW = 3 * H / 4;              // gate rows per wave
    __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int row0 = blockIdx.y * 32, u0 = blockIdx.x * 32;

    const int rr = tid >> 3, uq = 4 * (tid & 7);
    const int j = row0 + rr, u = u0 + uq;
    const int qs = g.Ls + g.W - 1 - mp;
    const int64_t pos = seg_pos(g, j, qs);
    const bool have_row = j < g.K;
    const bool live = have_row && !final && pos < g.total && seg_pos(g, j, 0) < g.total;
    float4 gr = zero4(), gz = zero4(), gn = zero4(), ghn = zero4(), hp = zero4(), de = zero4(), cin = zero4();
    if (have_row && !final) cin = ld4(carry + (int64_t)j * H + u);
    if (live) {
        int s, t, i;
        locate_fast(g, (int)pos, s, t, i);
        const float* sp = saved + pos * (5 * H) + u;
        gr = ld4(sp); gz = ld4(sp + H); gn = ld4(sp + 2 * H); ghn = ld4(sp + 3 * H);
        if (pos >= g.stride) hp = ld4(saved + (pos - g.stride) * (5 * H) + 4 * H + u);
        if (t == g.p0 + s - 1) de = ld4(d_emit + ((int64_t)s * g.n + i) * H + u);
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (mp > 0) {
        const float* asrc[8];
        int adst[8];
        const float* bsrc[8];
        int bdst[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int f = tid + 256 * i;
            const int ww = f >> 9, r = (f >> 4) & 31, kq = f & 15;
            asrc[i] = dg_prev + (int64_t)min(row0 + r, g.K - 1) * (3 * H) + ww * KW + 4 * kq;
            adst[i] = ww * (A_T + B_T) + r * LDT + 4 * kq;
            const int kk = (f >> 3) & 63, uq2 = f & 7;
            bsrc[i] = w_hh + (int64_t)(ww * KW + kk) * H + u0 + 4 * uq2;
            bdst[i] = ww * (A_T + B_T) + A_T + kk * 32 + 4 * uq2;
        }
        float4 sa[8], sb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { sa[i] = ld4(asrc[i]); sb[i] = ld4(bsrc[i]); }
#pragma unroll
        for (int i = 0; i < 8; ++i) { st4(lds + adst[i], sa[i]); st4(lds + bdst[i], sb[i]); }
        __syncthreads();
        constexpr int NK = KW / BK;
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
            // issue order of the stage: the next stage's global loads first, LDS reads between the matrix
            // instructions, the LDS stores of the loaded stage between the last ones
            __builtin_amdgcn_sched_group_barrier(0x020, 16, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
            __syncthreads();
        }
    }
    float* P = lds;
    acc_to_lds(acc, P + w * 32 * PLD, l31, hh);
    __syncthreads();

    if (!have_row) return;
    const float* pr = P + rr * PLD + uq;
    const float4 prod = add4(add4(ldp4(pr), ldp4(pr + 32 * PLD)), add4(ldp4(pr + 64 * PLD), ldp4(pr + 96 * PLD)));
    if (final) {
        st4(dh_out + (int64_t)j * H + u, prod);
        return;
    }
    if (qs == g.Ls - 1) {  // state entering the segment's last own position
        st4(seam_a + (int64_t)j * 2 * H + u, prod);
        st4(seam_a + (int64_t)j * 2 * H + H + u, cin);
    }
    float* dgo = dg_next + (int64_t)j * (3 * H) + u;
    if (!live) {
        st4(dgo, zero4()); st4(dgo + H, zero4()); st4(dgo + 2 * H, zero4());
        st4(carry + (int64_t)j * H + u, zero4());
        return;
    }
    const float4 dh = add4(add4(prod, cin), de);
    float4 dr, dz, dn, dhn, cy;
    GRU_BWD4(dh, gr, gz, gn, ghn, hp, dr, dz, dn, dhn, cy)
    st4(dgo, dr); st4(dgo + H, dz); st4(dgo + 2 * H, dhn);
    st4(carry + (int64_t)j * H + u, cy);
    if (qs < g.Ls) {
        float* dx = dgx + pos * (3 * H) + u;
        float* dg = dgh + pos * (3 * H) + u;
        st4(dx, dr); st4(dx + H, dz); st4(dx + 2 * H, dn);
        st4(dg, dr); st4(dg + H, dz); st4(dg + 2 * H, dhn);
    }
}

// seam j (between segments j and j+1): warm-up state of segment j vs final state of j+1;
// err_abs = max |a - b|, ref_abs = max |b| over dh and dc of all seams.
// abort_w (may be null): the abort word of a persistent launch; a launch that gave up waiting reports +inf.
__global__ __launch_bounds__(256) void seam_check_bwd(const float* __restrict__ seam_a, const float* __restrict__ dh_fin,
                                                      const float* __restrict__ dc, Geom g,
                                                      float* __restrict__ err_abs, float* __restrict__ ref_abs,
                                                      const unsigned* __restrict__ abort_w = nullptr) {
    __shared__ float red[8];
    float d = 0.f, r = 0.f;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < g.K * 2 * H; e += gridDim.x * 256) {
        const int j = e / (2 * H), k = e % (2 * H);
        if (j + 1 < g.K && (int64_t)(j + 1) * g.Ls + g.W < g.total) {
            const float b = (k < H) ? dh_fin[(int64_t)(j + 1) * H + k] : dc[(int64_t)(j + 1) * H + (k - H)];
            float v = fabsf(seam_a[e] - b);
            if (!(v == v)) v = INFINITY;
            d = fmaxf(d, v);
            r = fmaxf(r, fabsf(b));
        }
    }
    d = wave_max(d);
    r = wave_max(r);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = d; red[4 + (threadIdx.x >> 6)] = r; }
    __syncthreads();
    if (threadIdx.x == 0) {
        d = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        r = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
        if (abort_w != nullptr && *abort_w != 0) { d = INFINITY; r = fmaxf(r, 1.f); }
        if (abort_w != nullptr && blockIdx.x == 0) report_abort(abort_w, 1);
        if (d > 0.f) atomic_max_nonneg(err_abs, d);
        if (r > 0.f) atomic_max_nonneg(ref_abs, r);
    }
}

// ------------------------------------------------------------------ persistent forward pair
// Both forward chains (value LSTM, reward GRU) in ONE launch that walks every step.  Workgroups
// [0, 16*ny) run the LSTM chain, the next 16*ny the GRU chain; a workgroup owns unit tile x of
// segment-row tile y of its chain for all steps and keeps its cell state in registers.  At 80 KB of
// LDS two workgroups share a CU, so one chain's waits, loads and cell arithmetic run under the other
// chain's MFMAs.  Per step a workgroup needs the h rows of its row tile from the 16 workgroups (all
// x) of its row group; they are handed over inside the launch, correct on any workgroup placement:
//   producer: every h store is a 16-byte write-through (sc1) buffer store; every wave drains its
//             stores (s_waitcnt vmcnt(0)); workgroup barrier; ONE lane stores flag[y][x] = step + 1
//             (sc1);
//   consumer: ONE wave polls the 16 flags of its row group with sc1 loads, workgroup barrier, then
//             EVERY load of exchanged h bytes is a 16-byte sc1 buffer load to registers.
// h is double-buffered by step parity: a workgroup overwrites buffer p only after it has seen all
// 16 flags of the step in between, i.e. after every reader of p's old contents has finished.
// W_hh is read with plain loads (nobody writes it) and stays in the XCD's L2 across steps: there is
// no kernel boundary to invalidate it, and unit tile x runs on the XCD of block id x % 8.
// Spins are bounded by wall clock; a timeout raises an abort word that releases every waiter, and
// the host then poisons the seam errors so that the caller rejects the result.
// All 2 * 16 * ny workgroups must be resident together: the host checks the CU count.
using gu32 = __attribute__((address_space(1))) unsigned;
typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr int SC1 = 16;                       // buffer cache-policy bit: write-through store / L1-bypassing load
constexpr int PDEPTH = 2;                     // stages of global loads in flight
constexpr int PBUF = (32 + 4 * 32) * BK;      // floats per LDS stage buffer: 160 rows x 64 k, swizzled, no padding
constexpr unsigned long long SPIN_LIMIT = 2000000ull;  // 20 ms of the 100 MHz wall clock

struct ChainArgs {
    const float* xproj; const float* w_hh; const float* b_hh;
    const float* state_in;    // {h, c} (GRU: {h}) the chain starts from, or null
    int state_stride;         // floats between the start states of consecutive segments; 0: one state for the chain
    const float* warm_state;  // [K][H]: every segment's c (GRU: h) after step m_begin - 1, left by seg_warm_pair
    int m_begin;              // first step walked here (0, or the steps seg_warm_pair walked; hb then holds their h)
    float* hb;                // [2][K][H] exchanged state
    float* saved; float* h_emit; float* seam_a; float* seam_b;
    unsigned* flags;          // [row tiles][16]
    Geom g;
};

__device__ __forceinline__ float4 as_f4(v4u v) { return __builtin_bit_cast(float4, v); }
__device__ __forceinline__ v4u as_u4(float4 v) { return __builtin_bit_cast(v4u, v); }

// A flag word: step count in the low 24 bits, the producer's XCD in the high 8
constexpr unsigned EPOCH_MASK = 0x00ffffffu;
__device__ __forceinline__ unsigned my_xcd() { return __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u; }  // XCC_ID[3:0]

// A wait that has run out: the first wave to get here raises the abort word (which releases every other waiter) and
// leaves who it is, the step it waited at and which producers were behind (lag: ballot over the polling lanes).
__device__ __forceinline__ void give_up(unsigned* abort_w, unsigned step, unsigned long long lag, int lane) {
    if (lane == 0 && __hip_atomic_exchange((gu32*)abort_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        abort_w[1] = blockIdx.x;
        abort_w[2] = step;
        abort_w[3] = (unsigned)lag;
    }
}

// wave 0: lanes 0..15 each poll one producer's flag, lane 16 polls the abort word.  Returns whether all
// 16 producers of the row group run on XCD `xcd` (then their plain stores sit in this XCD's L2, where
// an sc1 load finds them).
__device__ __forceinline__ bool wait_row_group(unsigned* flags16, unsigned epoch, unsigned* abort_w, int lane,
                                               unsigned xcd = 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        unsigned v = epoch | (xcd << 24), ab = 0;
        if (lane < 16) v = __hip_atomic_load((gu32*)(flags16 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (lane == 16) ab = __hip_atomic_load((gu32*)abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__any(ab != 0)) return false;
        if (__all((v & EPOCH_MASK) >= epoch)) return __all((v >> 24) == xcd);
        if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_LIMIT) {
            give_up(abort_w, epoch, __ballot(lane < 16 && (v & EPOCH_MASK) < epoch), lane);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// Stage buffers hold rows of 64 k-contiguous floats without padding; the 16-byte chunk c of row R
// sits at chunk slot c ^ (R & 15), which keeps both the staging stores and the fragment reads
// (32 lanes, 32 rows, one chunk) free of bank conflicts.
struct FragOff { int o[4][2]; };
__device__ __forceinline__ Frag load_frag_swz(const float* a_row, const float* b_row, const FragOff& fo, int grp) {
    Frag f;
    f.a0 = ld4(a_row + fo.o[grp][0]); f.a1 = ld4(a_row + fo.o[grp][1]);
    f.b0 = ld4(b_row + fo.o[grp][0]); f.b1 = ld4(b_row + fo.o[grp][1]);
    return f;
}

// One chain, all steps, for this workgroup's tile (bx = unit tile, by = segment-row tile).
template <int G>
__device__ __forceinline__ void persist_chain(const ChainArgs& a, int bx, int by, unsigned* abort_w, float* lds) {
    const Geom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int row0 = by * 32, u0 = bx * 32;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(a.hb, 0, (int)(2 * (int64_t)g.K * H * sizeof(float)), 0x00020000);

    // staging map: float4 (tid + 256*i) of a stage; 16 float4 per row; rows 0..31 = h, then G*32 rows of W_hh
    const int sr = tid >> 4, kq = tid & 15;
    const int lds_off = sr * BK + 4 * (kq ^ sr);          // every row this thread stages has R & 15 == sr
    int a_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) a_off[i] = (min(row0 + sr + 16 * i, g.K - 1) * H + 4 * kq) * 4;
    // W_hh through a buffer descriptor too: one per-thread byte offset, slot and stage offsets are uniform
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.w_hh), 0, (int)((int64_t)G * H * H * sizeof(float)), 0x00020000);
    const int w_off = ((u0 + sr) * H + 4 * kq) * 4;   // slot i adds ((i>>1)*H + 16*(i&1)) * H floats
#define W_LOAD(i, stage) \
    as_f4(__builtin_amdgcn_raw_buffer_load_b128(rw, w_off, (((i) >> 1) * H + 16 * ((i) & 1)) * H * 4 + (stage) * BK * 4, 0))
    FragOff fo;
#pragma unroll
    for (int grp = 0; grp < 4; ++grp)
#pragma unroll
        for (int e = 0; e < 2; ++e) fo.o[grp][e] = 4 * ((4 * grp + 2 * hh + e) ^ (l31 & 15));
    const float* a_row = lds + l31 * BK;
    const float* b_row = lds + (32 + (q < G ? q : 0) * 32 + l31) * BK;   // LSTM: wave q owns gate q

    // this thread's epilogue item: segment row j, units u..u+3; its cell state lives in registers
    const int rr = tid >> 3, uq = 4 * (tid & 7);
    const int j = row0 + rr, u = u0 + uq;
    float4 state = zero4();            // LSTM: c, GRU: h
    const int m0 = a.m_begin;
    if (m0 > 0 && j < g.K) state = ld4(a.warm_state + (int64_t)j * H + u);
    int ps = 0, pt = 0, pi = 0;        // (call, column, row) of the current position
    float4 bias[G == 3 ? 3 : 1];
    if (G == 3) {
#pragma unroll
        for (int gq = 0; gq < 3; ++gq) bias[gq] = ld4(a.b_hh + gq * H + u);
    }

    const int steps = g.W + g.Ls;
    for (int m = m0; m < steps; ++m) {   // flags count the steps walked here: m - m0
        const int cur = m & 1, nxt = cur ^ 1;
        const int qs = m - g.W;     // step within the segment: negative during the warm-up
        const int64_t pos = seg_pos(g, j, qs);
        int st = 0;   // 0 = no row, 1 = outside the segment's range (zero state), 2 = live
        float4 px[G];
        if (j < g.K) {
            st = (pos < 0 || qs >= g.Ls || pos >= g.total) ? 1 : 2;
            if (st == 2) {
                locate_fast(g, (int)pos, ps, pt, pi);
                // input projection row: does not depend on the product, its latency hides under the k loop
                const float* xr = a.xproj + ((int64_t)pt * g.n + pi) * (G * H) + u;
#pragma unroll
                for (int gq = 0; gq < G; ++gq) px[gq] = ld4(xr + gq * H);
            }
        }

        constexpr int NACC = (G == 3) ? 3 : 1;
        f32x16 acc[NACC];
#pragma unroll
        for (int a_ = 0; a_ < NACC; ++a_)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a_][r] = 0.f;

        if (m > 0) {
            constexpr int NK = H / BK;
            constexpr int NW = 2 * G;
            float4 ring_w[PDEPTH][NW];
            v4u ring_a[PDEPTH][2];
            // W_hh does not depend on the hand-off: start its loads before waiting
#pragma unroll
            for (int d = 0; d < PDEPTH; ++d)
#pragma unroll
                for (int i = 0; i < NW; ++i)
                    ring_w[d][i] = W_LOAD(i, d);
            if (q == 0) wait_row_group(a.flags + by * 16, (unsigned)(m - m0), abort_w, lane);
            __syncthreads();
            const int a_base = cur * g.K * H * 4;
#pragma unroll
            for (int d = 0; d < PDEPTH; ++d)
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    ring_a[d][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, a_off[i], a_base + d * BK * 4, SC1);

#define ISSUE_INTERLEAVED(n)                                                                                \
    _Pragma("unroll") for (int i_ = 0; i_ < (n); ++i_) {                                                    \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                  \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                  \
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                                  \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                  \
    }
#define STORE_STAGE(buf, slot)                                                                              \
    {                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                       \
            st4(lds + (buf) + lds_off + 16 * i * BK, as_f4(ring_a[slot][i]));                               \
        _Pragma("unroll") for (int i = 0; i < NW; ++i)                                                      \
            st4(lds + (buf) + lds_off + (32 + 16 * i) * BK, ring_w[slot][i]);                               \
    }
#define LOAD_STAGE(slot, stage)                                                                             \
    {                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                       \
            ring_a[slot][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, a_off[i], a_base + (stage) * BK * 4, SC1); \
        _Pragma("unroll") for (int i = 0; i < NW; ++i)                                                      \
            ring_w[slot][i] = W_LOAD(i, stage);                                                             \
    }
            STORE_STAGE(0, 0)
            if (PDEPTH < NK) LOAD_STAGE(0, PDEPTH)
            __syncthreads();
            if constexpr (G == 3) {
                const float* b_row0 = lds + (32 + l31) * BK;
                const int o0 = fo.o[q][0], o1 = fo.o[q][1];
                FragK f = load_frag_k(a_row, b_row0, 32 * BK, o0, o1), fn;
#pragma unroll
                for (int kt = 0; kt < NK; ++kt) {
                    const int cn = ((kt + 1) & 1) * PBUF;
                    if (kt + 1 < NK) {   // the previous stage's barrier came after the last read of buffer `cn`
                        STORE_STAGE(cn, (kt + 1) % PDEPTH)
                        if (kt + 1 + PDEPTH < NK) LOAD_STAGE((kt + 1) % PDEPTH, kt + 1 + PDEPTH)
                    }
                    mfma8g(acc[0], f, 0);
                    mfma8g(acc[1], f, 1);
                    ISSUE_INTERLEAVED(24)   // between two barriers: behind each matrix instruction one LDS read, one LDS write, one global load
                    if (kt + 1 < NK) {
                        __syncthreads();
                        fn = load_frag_k(a_row + cn, b_row0 + cn, 32 * BK, o0, o1);
                    }
                    mfma8g(acc[2], f, 2);
                    f = fn;
                }
            } else {
                Frag f0 = load_frag_swz(a_row, b_row, fo, 0), f1;
#pragma unroll
                for (int kt = 0; kt < NK; ++kt) {
                    const int co = (kt & 1) * PBUF, cn = ((kt + 1) & 1) * PBUF;
                    f1 = load_frag_swz(a_row + co, b_row + co, fo, 1);
                    mfma8(acc[0], f0);
                    if (kt + 1 < NK) {   // nobody reads buffer `cn` any more: the previous stage's barrier came after its last read
                        STORE_STAGE(cn, (kt + 1) % PDEPTH)
                        if (kt + 1 + PDEPTH < NK) LOAD_STAGE((kt + 1) % PDEPTH, kt + 1 + PDEPTH)
                    }
                    f0 = load_frag_swz(a_row + co, b_row + co, fo, 2);
                    mfma8(acc[0], f1);
                    f1 = load_frag_swz(a_row + co, b_row + co, fo, 3);
                    mfma8(acc[0], f0);
                    ISSUE_INTERLEAVED(32)
                    if (kt + 1 < NK) {
                        __syncthreads();
                        f0 = load_frag_swz(a_row + cn, b_row + cn, fo, 0);
                    }
                    mfma8(acc[0], f1);
                }
            }
#undef ISSUE_INTERLEAVED
#undef STORE_STAGE
#undef LOAD_STAGE
            __syncthreads();  // every wave is done reading before the gate tiles overlay the stage buffers
        }
        float* P = lds;
        if constexpr (G == 3) {
#pragma unroll
            for (int gq = 0; gq < 3; ++gq) acc_to_lds(acc[gq], P + (q * 3 + gq) * 32 * PLD, l31, hh);
        } else {
            acc_to_lds(acc[0], P + q * 32 * PLD, l31, hh);
        }
        __syncthreads();

        if (st != 0) {
            float4 h = zero4();
            // where a segment's final state is recorded: with the next segment's seam, or (segments that all
            // replay the same positions) under its own index
            const int rec = g.stride == 0 ? j : j + 1;
            if (st == 1) {
                state = zero4();
                if (a.state_in != nullptr && pos == -1) {   // the step before the chain's first position
                    const float* si = a.state_in + (int64_t)j * a.state_stride;
                    h = ld4(si + u);
                    state = (G == 4) ? ld4(si + H + u) : h;
                }
            } else {
                const float* pr = P + rr * PLD + uq;
                if (G == 4) {
                    float4 gi = add4(ldp4(pr), px[0]), gf = add4(ldp4(pr + 32 * PLD), px[1]);
                    float4 gg = add4(ldp4(pr + 64 * PLD), px[2]), go = add4(ldp4(pr + 96 * PLD), px[G - 1]);
                    float4 c;
                    LSTM_CELL4(gi, gf, gg, go, state, c, h)
                    state = c;
                    if (qs >= 0 && a.saved) {
                        float* sp = a.saved + pos * (6 * H) + u;
                        st4(sp, gi); st4(sp + H, gf); st4(sp + 2 * H, gg); st4(sp + 3 * H, go); st4(sp + 4 * H, c); st4(sp + 5 * H, h);
                    }
                    if (qs == -1) st4(a.seam_a + (int64_t)j * 2 * H + H + u, c);
                    if (qs == g.Ls - 1 && rec < g.K) st4(a.seam_b + (int64_t)rec * 2 * H + H + u, c);
                } else {
                    const float4 hr_ = add4(sum_partials(pr), bias[0]), hz = add4(sum_partials(pr + 32 * PLD), bias[1]);
                    const float4 hn = add4(sum_partials(pr + 64 * PLD), bias[G == 3 ? 2 : 0]);
                    GRU_CELL4(px[0], px[1], px[2], hr_, hz, hn, state, h)
                    state = h;
                    if (qs >= 0 && a.saved) {   // {r, z, n, hn, h}: the same expressions as in the cell, so the same bits
                        float4 r4, z4, n4;
                        GRU_GATES4(px[0], px[1], px[2], hr_, hz, hn, r4, z4, n4)
                        float* sp = a.saved + pos * (5 * H) + u;
                        st4(sp, r4); st4(sp + H, z4); st4(sp + 2 * H, n4); st4(sp + 3 * H, hn); st4(sp + 4 * H, h);
                    }
                }
                if (qs >= 0 && a.h_emit != nullptr && pt == g.p0 + ps - 1) st4(a.h_emit + ((int64_t)ps * g.n + pi) * H + u, h);
                constexpr int SW = (G == 4) ? 2 * H : H;  // seam record: {h, c} or {h}
                if (qs == -1) st4(a.seam_a + (int64_t)j * SW + u, h);
                if (qs == g.Ls - 1 && rec < g.K) st4(a.seam_b + (int64_t)rec * SW + u, h);
            }
            // the exchanged copy of h: write-through, 16 bytes per lane
            __builtin_amdgcn_raw_buffer_store_b128(as_u4(h), rs, (j * H + u) * 4, nxt * g.K * H * 4, SC1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its stores ...
        __syncthreads();                                    // ... before ONE lane signals for all of them
        if (tid == 0)
            __hip_atomic_store((gu32*)(a.flags + by * 16 + bx), (unsigned)(m + 1 - m0), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
}

#undef W_LOAD

// ------------------------------------------------------------------ the first steps of a warm-up, at 16-bit precision
// A warm-up's results are thrown away: all that matters is the state it leaves, and the seam check measures
// that.  Its first steps therefore run on the bf16 matrix instruction, at 1/16 of the f32 instruction's time,
// with both operands cut in two bf16 pieces (x = hi + mid exactly to 16 significant bits; the three products
// hi.hi + hi.mid + mid.hi carry 2^-16 relative error).  The f32 launch that follows continues from the state
// left here (ChainArgs::warm_state, m_begin) with a few more warm-up steps, which wash the 2^-16 out.
// Plain bf16 (8 bits) is not enough: a long-memory LSTM keeps its 4e-3 error for hundreds of steps.
//
// Workgroup (unit tile bx of 16 units, row group by of 32 segments) of one chain; wave q owns gate q and keeps
// its 16 x 512 slice of W_hh, both pieces, in registers for the whole launch (128 VGPRs), so a step moves only
// the row group's h: 32 x 512 floats gathered from the 32 workgroups of the group (sc1 loads after the flags),
// cut into the two planes in LDS, 96 matrix instructions of 16x16x32 per wave, the gate tiles through LDS, the
// cell for two units per lane with c (GRU: h) in registers, h handed on as in persist_chain.
typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wbf16x4 __attribute__((ext_vector_type(4)));
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef unsigned v2u __attribute__((ext_vector_type(2)));
constexpr int WLD = H + 8;     // bf16 per LDS row of an h plane: 1040 bytes, so 16 rows at one k fall in 16 x 4 different banks
constexpr int WGT = 17;        // floats per row of a gate tile
constexpr int WARM_LDS = 2 * 32 * WLD * 2 + 4 * 32 * WGT * 4;

struct WarmArgs {
    const float* xproj; const float* w_hh; const float* b_hh;
    float* hb;              // [2][K][H] exchanged h; the last step leaves its h in buffer steps & 1
    float* state_out;       // [K][H]: c (GRU: h) after the last step
    unsigned* flags;        // [row groups][32]
    Geom g;
    int steps;              // steps m = 0 .. steps - 1 of the chain's W + Ls are walked here (steps < W)
};

__device__ __forceinline__ void split2(float x, __bf16& hi, __bf16& mid) {
    hi = (__bf16)x;
    mid = (__bf16)(x - (float)hi);
}

// wave 0: lanes 0..31 each poll one producer's flag, lane 32 polls the abort word
__device__ __forceinline__ bool wait_row_group32(unsigned* flags32, unsigned epoch, unsigned* abort_w, int lane) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        unsigned v = epoch, ab = 0;
        if (lane < 32) v = __hip_atomic_load((gu32*)(flags32 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (lane == 32) ab = __hip_atomic_load((gu32*)abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__any(ab != 0)) return false;
        if (__all(v >= epoch)) return true;
        if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_LIMIT) {
            give_up(abort_w, epoch, __ballot(lane < 32 && v < epoch), lane);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

template <int G>
__device__ __forceinline__ void warm_chain(const WarmArgs& a, int bx, int by, unsigned* abort_w, char* lds_raw) {
    const Geom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
    const int l15 = lane & 15, kg = lane >> 4;
    const int row0 = by * 32, u0 = bx * 16;
    __bf16* plane = reinterpret_cast<__bf16*>(lds_raw);                                      // [2][32][WLD]
    float* gt = reinterpret_cast<float*>(lds_raw + 2 * 32 * WLD * sizeof(__bf16));           // [G][32][WGT]
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(a.hb, 0, (int)(2 * (int64_t)g.K * H * sizeof(float)), 0x00020000);

    // B operand of lane (column l15, k group kg) at k step ks: W_hh[gate q, unit u0 + l15][32 ks + 8 kg + 0..7]
    wbf16x8 bw[2][16];
    if (q < G) {
        const float* wrow = a.w_hh + ((int64_t)q * H + u0 + l15) * H + 8 * kg;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const float4 w0 = ld4(wrow + 32 * ks), w1 = ld4(wrow + 32 * ks + 4);
            const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                __bf16 hi, mid;
                split2(w[e], hi, mid);
                bw[0][ks][e] = hi;
                bw[1][ks][e] = mid;
            }
        }
    }

    // this lane's epilogue item: segment row j, units u, u + 1; its cell state lives in registers
    const int rr = tid >> 3, uq = 2 * (tid & 7);
    const int j = row0 + rr, u = u0 + uq;
    float2 state = make_float2(0.f, 0.f);
    int ps = 0, pt = 0, pi = 0;
    float2 bias[3];
    if (G == 3) {
#pragma unroll
        for (int gq = 0; gq < 3; ++gq) bias[gq] = *reinterpret_cast<const float2*>(a.b_hh + gq * H + u);
    }

    for (int m = 0; m < a.steps; ++m) {
        const int cur = m & 1, nxt = cur ^ 1;
        const int qs = m - g.W;
        const int64_t pos = seg_pos(g, j, qs);
        int st = 0;   // 0 = no row, 1 = before the chain's first position (zero state), 2 = live
        float2 px[G];
        if (j < g.K) {
            st = (pos < 0 || pos >= g.total) ? 1 : 2;
            if (st == 2) {
                locate_fast(g, (int)pos, ps, pt, pi);
                const float* xr = a.xproj + ((int64_t)pt * g.n + pi) * (G * H) + u;
#pragma unroll
                for (int gq = 0; gq < G; ++gq) px[gq] = *reinterpret_cast<const float2*>(xr + gq * H);
            }
        }
        f32x4 acc[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mt][r] = 0.f;

        if (m > 0) {
            if (q == 0) wait_row_group32(a.flags + by * 32, (unsigned)m, abort_w, lane);
            __syncthreads();
            // the row group's h: float4 (tid + 256 i) of 32 rows x 128, cut into the hi and mid planes
#pragma unroll
            for (int half = 0; half < 2; ++half) {   // eight loads in flight at a time
                v4u raw[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int f = tid + 256 * (8 * half + i), row = f >> 7, c4 = f & 127;
                    raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (min(row0 + row, g.K - 1) * H + 4 * c4) * 4,
                                                                   cur * g.K * H * 4, SC1);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int f = tid + 256 * (8 * half + i), row = f >> 7, c4 = f & 127;
                    const float4 v = as_f4(raw[i]);
                    wbf16x4 hi, mid;
                    __bf16 h_, m_;
                    split2(v.x, h_, m_); hi[0] = h_; mid[0] = m_;
                    split2(v.y, h_, m_); hi[1] = h_; mid[1] = m_;
                    split2(v.z, h_, m_); hi[2] = h_; mid[2] = m_;
                    split2(v.w, h_, m_); hi[3] = h_; mid[3] = m_;
                    *reinterpret_cast<wbf16x4*>(plane + row * WLD + 4 * c4) = hi;
                    *reinterpret_cast<wbf16x4*>(plane + (32 + row) * WLD + 4 * c4) = mid;
                }
            }
            __syncthreads();
            if (q < G) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const __bf16* ar = plane + (mt * 16 + l15) * WLD + 8 * kg;
#pragma unroll
                    for (int ks = 0; ks < 16; ++ks) {
                        const wbf16x8 a_hi = *reinterpret_cast<const wbf16x8*>(ar + 32 * ks);
                        const wbf16x8 a_mid = *reinterpret_cast<const wbf16x8*>(ar + 32 * WLD + 32 * ks);
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, bw[0][ks], acc[mt], 0, 0, 0);
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, bw[1][ks], acc[mt], 0, 0, 0);
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_mid, bw[0][ks], acc[mt], 0, 0, 0);
                    }
                }
            }
        }
        // accumulator register r of lane (l15, kg) is element [row 4 kg + r][column l15] of its 16 x 16 tile
        if (q < G) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) gt[(q * 32 + mt * 16 + 4 * kg + r) * WGT + l15] = acc[mt][r];
        }
        __syncthreads();

        if (st != 0) {
            float2 h = make_float2(0.f, 0.f);
            if (st == 1) {
                state = make_float2(0.f, 0.f);
            } else {
                const float* pr = gt + rr * WGT + uq;
                if (G == 4) {
                    float2 gi = make_float2(pr[0] + px[0].x, pr[1] + px[0].y);
                    float2 gf = make_float2(pr[32 * WGT] + px[1].x, pr[32 * WGT + 1] + px[1].y);
                    float2 gg = make_float2(pr[64 * WGT] + px[2].x, pr[64 * WGT + 1] + px[2].y);
                    float2 go = make_float2(pr[96 * WGT] + px[G - 1].x, pr[96 * WGT + 1] + px[G - 1].y);
                    float2 c;
                    LSTM_CELL1(x, gi, gf, gg, go, state, c, h)
                    LSTM_CELL1(y, gi, gf, gg, go, state, c, h)
                    state = c;
                } else {
                    const float2 hr_ = make_float2(pr[0] + bias[0].x, pr[1] + bias[0].y);
                    const float2 hz = make_float2(pr[32 * WGT] + bias[1].x, pr[32 * WGT + 1] + bias[1].y);
                    const float2 hn = make_float2(pr[64 * WGT] + bias[2].x, pr[64 * WGT + 1] + bias[2].y);
                    GRU_CELL1(x, px[0], px[1], px[2], hr_, hz, hn, state, h)
                    GRU_CELL1(y, px[0], px[1], px[2], hr_, hz, hn, state, h)
                    state = h;
                }
            }
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, h), rs, (j * H + u) * 4, nxt * g.K * H * 4, SC1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0)
            __hip_atomic_store((gu32*)(a.flags + by * 32 + bx), (unsigned)(m + 1), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
    if (j < g.K) *reinterpret_cast<float2*>(a.state_out + (int64_t)j * H + u) = state;
}

// Both chains' first warm-up steps in one launch.  Block b runs on XCD b % 8; a row group's 32 workgroups are
// given block ids of one XCD (speed only: every hand-off store is write-through).
__global__ __launch_bounds__(256, 2) void seg_warm_pair(WarmArgs v, WarmArgs r, unsigned* abort_w, int per_chain, int ny) {
    extern __shared__ __attribute__((aligned(16))) char warm_lds[];
    __builtin_amdgcn_s_setprio(3);
    const int chain = blockIdx.x / per_chain, w = blockIdx.x % per_chain;
    const int xcd = w & 7, t = w >> 3;
    const int by = (t >> 5) * 8 + xcd, bx = t & 31;
    if (by >= ny) return;
    if (chain == 0) { if (v.steps > 0) warm_chain<4>(v, bx, by, abort_w, warm_lds); }
    else if (r.steps > 0) warm_chain<3>(r, bx, by, abort_w, warm_lds);
}


// ------------------------------------------------------------------ both forward chains, every step, on the f16 matrix instruction
// The same workgroup shape as seg_warm_pair (unit tile of 16 units, row group of 32 segments, wave q = gate q with
// its slice of W_hh in registers for the whole launch), but at float32-level accuracy, so that it walks the WHOLE
// chain - warm-up and own positions - and writes what seg_fwd_pair_persist writes (kept activations, emitted h,
// seam records).  Every operand is cut in two f16 pieces after a power-of-two scaling that keeps both pieces in
// f16's normal range: h (|h| < 1) by 2^12, a row of W_hh by the power of two that brings its largest entry into
// [2^10, 2^11).  x s = hi + mid + r with |r| <= 2^-22 |x s|, and the product is hi.hi + hi.mid + mid.hi on
// v_mfma_f32_16x16x32_f16 (the piece products are exact, the sums float32); what is dropped - mid.mid and the two
// r terms - is below 2^-21 of each element product, the size of float32's own rounding of the 512-term sum.  Three
// matrix instructions of 16 x 16 x 32 do the work of sixteen 32 x 32 x 2 float32 ones, and a step moves only the row
// group's h (W_hh never leaves the registers).  The scalings are powers of two, so undoing them is exact.
typedef _Float16 hf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 hf16x4 __attribute__((ext_vector_type(4)));
constexpr float HP_H_SCALE = 4096.f;           // 2^12

struct HpArgs {
    const float* xproj; const float* w_hh; const float* b_hh;
    float* hb;                // [2][K][H] exchanged h
    float* saved; float* h_emit; float* seam_a; float* seam_b;
    unsigned* flags;          // [row groups][32]
    Geom g;
    const float* state_in;    // {h, c} (GRU: {h}) the chains start from (loaded at position -1), or null
    int state_stride;         // floats between the start states of consecutive segments
};

__device__ __forceinline__ void split2h(float x, _Float16& hi, _Float16& mid) {
    hi = (_Float16)x;
    mid = (_Float16)(x - (float)hi);
}

template <int G>
__device__ __forceinline__ void hp_chain(const HpArgs& a, int bx, int by, unsigned* abort_w, char* lds_raw) {
    const Geom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
    const int l15 = lane & 15, kg = lane >> 4;
    const int row0 = by * 32, u0 = bx * 16;
    _Float16* plane = reinterpret_cast<_Float16*>(lds_raw);                                   // [2][32][WLD]
    float* gt = reinterpret_cast<float*>(lds_raw + 2 * 32 * WLD * sizeof(_Float16));          // [G][32][WGT]
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(a.hb, 0, (int)(2 * (int64_t)g.K * H * sizeof(float)), 0x00020000);

    // B operand of lane (column l15, k group kg) at k step ks: W_hh[gate q, unit u0 + l15][32 ks + 8 kg + 0..7], the
    // row scaled by a power of two of its own; `descale` undoes that and h's scaling on the column's sums
    hf16x8 bw[2][16];
    float descale = 1.f;
    if (q < G) {
        const float* wrow = a.w_hh + ((int64_t)q * H + u0 + l15) * H + 8 * kg;
        float top = 0.f;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const float4 w0 = ld4(wrow + 32 * ks), w1 = ld4(wrow + 32 * ks + 4);
            top = fmaxf(top, fmaxf(fmaxf(fabsf(w0.x), fabsf(w0.y)), fmaxf(fabsf(w0.z), fabsf(w0.w))));
            top = fmaxf(top, fmaxf(fmaxf(fabsf(w1.x), fabsf(w1.y)), fmaxf(fabsf(w1.z), fabsf(w1.w))));
        }
        top = fmaxf(top, __shfl_xor(top, 16));   // the four lanes of a column hold a quarter of its row each
        top = fmaxf(top, __shfl_xor(top, 32));
        int ex = 0;
        if (top > 0.f && top < INFINITY) frexpf(top, &ex);    // top = f 2^ex, f in [0.5, 1)
        ex = max(-100, min(100, ex));
        const float scale = ldexpf(1.f, 11 - ex);             // top * scale in [2^10, 2^11)
        descale = ldexpf(1.f, ex - 11 - 12);
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const float4 w0 = ld4(wrow + 32 * ks), w1 = ld4(wrow + 32 * ks + 4);
            const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                _Float16 hi, mid;
                split2h(w[e] * scale, hi, mid);
                bw[0][ks][e] = hi;
                bw[1][ks][e] = mid;
            }
        }
    }

    // this lane's epilogue item: segment row j, units u, u + 1; its cell state lives in registers
    const int rr = tid >> 3, uq = 2 * (tid & 7);
    const int j = row0 + rr, u = u0 + uq;
    float2 state = make_float2(0.f, 0.f);      // LSTM: c, GRU: h
    int ps = 0, pt = 0, pi = 0;
    float2 bias[3];
    if (G == 3) {
#pragma unroll
        for (int gq = 0; gq < 3; ++gq) bias[gq] = *reinterpret_cast<const float2*>(a.b_hh + gq * H + u);
    }
#define ST2(p, v) (*reinterpret_cast<float2*>(p) = (v))

    const int steps = g.W + g.Ls;
    for (int m = 0; m < steps; ++m) {
        const int cur = m & 1, nxt = cur ^ 1;
        const int qs = m - g.W;     // step within the segment: negative during the warm-up
        const int64_t pos = seg_pos(g, j, qs);
        int st = 0;   // 0 = no row, 1 = outside the segment's range (zero state), 2 = live
        float2 px[G];
        if (j < g.K) {
            st = (pos < 0 || qs >= g.Ls || pos >= g.total) ? 1 : 2;
            if (st == 2) {
                locate_fast(g, (int)pos, ps, pt, pi);
                const float* xr = a.xproj + ((int64_t)pt * g.n + pi) * (G * H) + u;
#pragma unroll
                for (int gq = 0; gq < G; ++gq) px[gq] = *reinterpret_cast<const float2*>(xr + gq * H);
            }
        }
        f32x4 acc[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mt][r] = 0.f;

        if (m > 0) {
            if (q == 0) wait_row_group32(a.flags + by * 32, (unsigned)m, abort_w, lane);
            __syncthreads();
            // the row group's h: float4 (tid + 256 i) of 32 rows x 128, scaled and cut into the hi and mid planes
#pragma unroll
            for (int half = 0; half < 2; ++half) {   // eight loads in flight at a time
                v4u raw[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int f = tid + 256 * (8 * half + i), row = f >> 7, c4 = f & 127;
                    raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, (min(row0 + row, g.K - 1) * H + 4 * c4) * 4,
                                                                   cur * g.K * H * 4, SC1);
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int f = tid + 256 * (8 * half + i), row = f >> 7, c4 = f & 127;
                    const float4 v = as_f4(raw[i]);
                    hf16x4 hi, mid;
                    _Float16 h_, m_;
                    split2h(v.x * HP_H_SCALE, h_, m_); hi[0] = h_; mid[0] = m_;
                    split2h(v.y * HP_H_SCALE, h_, m_); hi[1] = h_; mid[1] = m_;
                    split2h(v.z * HP_H_SCALE, h_, m_); hi[2] = h_; mid[2] = m_;
                    split2h(v.w * HP_H_SCALE, h_, m_); hi[3] = h_; mid[3] = m_;
                    *reinterpret_cast<hf16x4*>(plane + row * WLD + 4 * c4) = hi;
                    *reinterpret_cast<hf16x4*>(plane + (32 + row) * WLD + 4 * c4) = mid;
                }
            }
            __syncthreads();
            if (q < G) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const _Float16* ar = plane + (mt * 16 + l15) * WLD + 8 * kg;
#pragma unroll
                    for (int ks = 0; ks < 16; ++ks) {
                        const hf16x8 a_hi = *reinterpret_cast<const hf16x8*>(ar + 32 * ks);
                        const hf16x8 a_mid = *reinterpret_cast<const hf16x8*>(ar + 32 * WLD + 32 * ks);
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, bw[0][ks], acc[mt], 0, 0, 0);
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, bw[1][ks], acc[mt], 0, 0, 0);
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_mid, bw[0][ks], acc[mt], 0, 0, 0);
                    }
                }
            }
        }
        // accumulator register r of lane (l15, kg) is element [row 4 kg + r][column l15] of its 16 x 16 tile
        if (q < G) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) gt[(q * 32 + mt * 16 + 4 * kg + r) * WGT + l15] = acc[mt][r] * descale;
        }
        __syncthreads();

        if (st != 0) {
            float2 h = make_float2(0.f, 0.f);
            // where a segment's final state is recorded: with the next segment's seam, or (segments that all
            // replay the same positions) under its own index
            const int rec = g.stride == 0 ? j : j + 1;
            if (st == 1) {
                state = make_float2(0.f, 0.f);
                if (a.state_in != nullptr && pos == -1) {   // the step before the chain's first position
                    const float* si = a.state_in + (int64_t)j * a.state_stride;
                    h = *reinterpret_cast<const float2*>(si + u);
                    state = (G == 4) ? *reinterpret_cast<const float2*>(si + H + u) : h;
                }
            } else {
                const float* pr = gt + rr * WGT + uq;
                if (G == 4) {
                    float2 gi = make_float2(pr[0] + px[0].x, pr[1] + px[0].y);
                    float2 gf = make_float2(pr[32 * WGT] + px[1].x, pr[32 * WGT + 1] + px[1].y);
                    float2 gg = make_float2(pr[64 * WGT] + px[2].x, pr[64 * WGT + 1] + px[2].y);
                    float2 go = make_float2(pr[96 * WGT] + px[G - 1].x, pr[96 * WGT + 1] + px[G - 1].y);
                    float2 c;
                    LSTM_CELL1(x, gi, gf, gg, go, state, c, h)
                    LSTM_CELL1(y, gi, gf, gg, go, state, c, h)
                    state = c;
                    if (qs >= 0 && a.saved) {
                        float* sp = a.saved + pos * (6 * H) + u;
                        ST2(sp, gi); ST2(sp + H, gf); ST2(sp + 2 * H, gg); ST2(sp + 3 * H, go); ST2(sp + 4 * H, c); ST2(sp + 5 * H, h);
                    }
                    if (qs == -1) ST2(a.seam_a + (int64_t)j * 2 * H + H + u, c);
                    if (qs == g.Ls - 1 && rec < g.K) ST2(a.seam_b + (int64_t)rec * 2 * H + H + u, c);
                } else {
                    const float2 hr_ = make_float2(pr[0] + bias[0].x, pr[1] + bias[0].y);
                    const float2 hz = make_float2(pr[32 * WGT] + bias[1].x, pr[32 * WGT + 1] + bias[1].y);
                    const float2 hn = make_float2(pr[64 * WGT] + bias[2].x, pr[64 * WGT + 1] + bias[2].y);
                    GRU_CELL1(x, px[0], px[1], px[2], hr_, hz, hn, state, h)
                    GRU_CELL1(y, px[0], px[1], px[2], hr_, hz, hn, state, h)
                    if (qs >= 0 && a.saved) {   // {r, z, n, hn, h}: the same expressions as in the cell
                        float2 r2, z2, n2;
                        GRU_GATES1(x, px[0], px[1], px[2], hr_, hz, hn, r2, z2, n2)
                        GRU_GATES1(y, px[0], px[1], px[2], hr_, hz, hn, r2, z2, n2)
                        float* sp = a.saved + pos * (5 * H) + u;
                        ST2(sp, r2); ST2(sp + H, z2); ST2(sp + 2 * H, n2); ST2(sp + 3 * H, hn); ST2(sp + 4 * H, h);
                    }
                    state = h;
                }
                if (qs >= 0 && a.h_emit != nullptr && pt == g.p0 + ps - 1) ST2(a.h_emit + ((int64_t)ps * g.n + pi) * H + u, h);
                constexpr int SW = (G == 4) ? 2 * H : H;  // seam record: {h, c} or {h}
                if (qs == -1) ST2(a.seam_a + (int64_t)j * SW + u, h);
                if (qs == g.Ls - 1 && rec < g.K) ST2(a.seam_b + (int64_t)rec * SW + u, h);
            }
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, h), rs, (j * H + u) * 4, nxt * g.K * H * 4, SC1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its stores ...
        __syncthreads();                                    // ... before ONE lane signals for all of them
        if (tid == 0)
            __hip_atomic_store((gu32*)(a.flags + by * 32 + bx), (unsigned)(m + 1), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
#undef ST2
}

// Block b runs on XCD b % 8; a row group's 32 workgroups are given block ids of one XCD (speed only: every hand-off
// store is write-through).
__global__ __launch_bounds__(256, 2) void seg_fwd_pair_hp(HpArgs v, HpArgs r, unsigned* abort_w, int per_chain, int ny) {
    extern __shared__ __attribute__((aligned(16))) char hp_lds[];
    __builtin_amdgcn_s_setprio(3);
    const int chain = blockIdx.x / per_chain, w = blockIdx.x % per_chain;
    int by, bx;
    if (ny < 8) {
        // fewer than eight row groups (the replay check's single one): consecutive block ids, which the dispatcher
        // deals out over all XCDs.  Packed into one XCD they fill its CUs' registers, and a persistent launch of
        // another stream that needs a CU there for one of its workgroups (the value chain's BPTT) cannot start
        by = w >> 5; bx = w & 31;
    } else {
        const int xcd = w & 7, t = w >> 3;
        by = (t >> 5) * 8 + xcd; bx = t & 31;
    }
    if (by >= ny) return;
    if (chain == 0) hp_chain<4>(v, bx, by, abort_w, hp_lds);
    else hp_chain<3>(r, bx, by, abort_w, hp_lds);
}

// only: 0 = both chains (2 * per_chain workgroups), 1 = the LSTM chain alone, 2 = the GRU chain alone
// (per_chain workgroups)
__global__ __launch_bounds__(256, 2) void seg_fwd_pair_persist(ChainArgs v, ChainArgs r, unsigned* abort_w, int per_chain,
                                                               int only) {
    __shared__ __attribute__((aligned(16))) float lds[2 * PBUF];
    // every step of a chain waits for the step before it: its waves go first wherever another kernel's waves
    // share their SIMD (products of another stream, whose work has no such chain of waits)
    __builtin_amdgcn_s_setprio(3);
    const int id = blockIdx.x % per_chain;
    if (only == 1 || (only == 0 && (int)blockIdx.x < per_chain)) persist_chain<4>(v, id % 16, id / 16, abort_w, lds);
    else persist_chain<3>(r, id % 16, id / 16, abort_w, lds);
}

// ------------------------------------------------------------------ persistent LSTM backward
// The backward chain in ONE launch, same arithmetic as seg_bwd_step step by step.  Workgroup (x, y)
// owns dh tile [32 segment rows of tile y] x [units 32x..32x+31] for every step and carries dc in
// registers; the dgates rows it needs each step come from the 16 workgroups of its row group through
// the same in-launch hand-off as the forward pair (16-byte sc1 stores, drained, one flag per
// workgroup and step; sc1 loads after the poll).  One workgroup per CU at K <= 512 segments, 70 KB of
// LDS and at most 256 registers per lane, so that another stream's GEMM workgroups can share the CU.
constexpr int BK2 = 32;                        // k per wave and stage (four waves: 128 gate rows per stage)
constexpr int LDA2 = BK2 + 4;
constexpr int TILE2 = 32 * LDA2 + BK2 * 32;    // per wave: A [32 rows][32 k] padded + B [32 k][32 units]
constexpr int PBUF2 = 4 * TILE2;

struct BwdArgs {
    const float* w_hh; const float* d_emit;
    // the forward pass's activations of position p: gates {i, f, g, o} at saved + p * gate_stride (+ 0, H, 2H, 3H) and
    // the cell state at cc + p * c_stride.  The chains keep them as one record {i, f, g, o, c, h} per position
    // (gate_stride = c_stride = 6 H, cc = saved + 4 H); a caller with gates and cell states in arrays of their
    // own (the policy LSTM: [T, N, 4H] and [T + 1, N, H]) points at those
    const float* saved; const float* cc;
    int gate_stride, c_stride;
    float* dgx;               // [2][K][4H] exchanged dgates
    float* dc; float* dgv; float* seam_a; float* dh_fin;
    unsigned* flags;          // [row tiles][32], 16 used
    Geom g;
    int m_begin;              // first step walked here (0, or the steps seg_warm_bwd walked: dgx[(m_begin - 1) & 1] then
                              // holds their last dgates and dc their carried dc)
};

struct FragB2 { float4 a0, a1; float b[8]; };
__device__ __forceinline__ FragB2 load_frag2(const float* tile, int grp, int l31, int hh) {
    FragB2 f;
    const float* a_row = tile + l31 * LDA2 + 16 * grp + 8 * hh;
    f.a0 = ld4(a_row); f.a1 = ld4(a_row + 4);
    const float* bk = tile + 32 * LDA2 + (16 * grp + 8 * hh) * 32 + l31;
#pragma unroll
    for (int s = 0; s < 8; ++s) f.b[s] = bk[32 * s];
    return f;
}
__device__ __forceinline__ void mfma8(f32x16& acc, const FragB2& f) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0.x, f.b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0.y, f.b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0.z, f.b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0.w, f.b[3], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1.x, f.b[4], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1.y, f.b[5], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1.z, f.b[6], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1.w, f.b[7], acc, 0, 0, 0);
}

__global__ __launch_bounds__(256, 2) void seg_bwd_persist(BwdArgs a, unsigned* abort_w) {
    // 84 KB although the stages need 70: two of these workgroups must never share a CU (they would
    // halve the chain's speed for the whole launch), while a GEMM workgroup of another stream still fits
    constexpr int LDS_FLOATS = 21 * 1024;
    static_assert(LDS_FLOATS >= 2 * PBUF2 && TILE2 >= 32 * PLD, "stage buffers, partial tile inside a wave's stage tile");
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    __builtin_amdgcn_s_setprio(3);   // as in the forward launch: the chain's waves before another stream's
    const Geom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    // Workgroup b is dispatched to XCD b % 8.  With a multiple of 8 row groups, the 16 workgroups of a
    // row group are given ids of one residue, so that their exchange can stay inside that XCD's L2
    // (checked at run time through the XCD ids in the flags: `local`).
    const int groups = gridDim.x / 16;
    int bx = blockIdx.x % 16, by = blockIdx.x / 16;
    if (groups % 8 == 0) {
        const int s = blockIdx.x / 8;
        bx = s % 16;
        by = (blockIdx.x % 8) * (groups / 8) + s / 16;
    }
    const unsigned xcd = my_xcd();
    int* local_w = reinterpret_cast<int*>(lds + LDS_FLOATS - 4);
    bool local = false;
    const int row0 = by * 32, u0 = bx * 32;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(a.dgx, 0, (int)(2 * (int64_t)g.K * 4 * H * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.w_hh), 0, (int)((int64_t)4 * H * H * sizeof(float)), 0x00020000);

    // staging map: every wave stages its own tiles (its 512 gate rows are its k range), so the k loop
    // needs no workgroup barrier: a wave's LDS stores and loads execute in order.  Piece i of a stage:
    // A row lane/8 + 8i, float4 lane%8 of the 32 k; B k row lane/8 + 8i, float4 lane%8 of the 32 units
    const int sr = lane >> 3, sq = lane & 7;
    int a_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a_off[i] = (min(row0 + sr + 8 * i, g.K - 1) * 4 * H + w * H + 4 * sq) * 4;  // + stage*BK2 floats
    const int b_off = ((w * H + sr) * H + u0 + 4 * sq) * 4;             // + (8i + stage*BK2) * H floats
    const int a_dst = sr * LDA2 + 4 * sq, b_dst = 32 * LDA2 + sr * 32 + 4 * sq;   // + 8i rows
#define A_LOAD(i, stage, base) __builtin_amdgcn_raw_buffer_load_b128(rs, a_off[i], (base) + (stage) * BK2 * 4, SC1)
#define B_LOAD(i, stage) as_f4(__builtin_amdgcn_raw_buffer_load_b128(rw, b_off, (8 * (i) + (stage) * BK2) * H * 4, 0))
#define WAVE_SYNC() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

    // this thread's epilogue item and its carried dc
    const int rr = tid >> 3, uq = 4 * (tid & 7);
    const int j = row0 + rr, u = u0 + uq;
    const bool have_row = j < g.K;
    const int m0 = a.m_begin;
    float4 dcv = zero4();
    if (m0 > 0 && have_row) dcv = ld4(a.dc + (int64_t)j * H + u);
    int ps = 0, pt = 0, pi = 0;
    float* tile = lds + w * TILE2;

    const int steps = g.W + g.Ls;
    for (int mp = m0; mp <= steps; ++mp) {   // flags count the steps walked here: mp - m0
        const bool final = mp == steps;
        const int cur = mp & 1;   // this step writes dgx[cur]; it reads dgx[cur ^ 1]
        const int qs = g.Ls + g.W - 1 - mp;   // step within the segment: >= Ls during the warm-up
        const int64_t pos = seg_pos(g, j, qs);
        const bool live = have_row && !final && pos < g.total && seg_pos(g, j, 0) < g.total;
        float4 gi = zero4(), gf = zero4(), gg = zero4(), go = zero4(), cc = zero4(), cp = zero4(), de = zero4();
        if (live) {
            locate_fast(g, (int)pos, ps, pt, pi);
            const float* sp = a.saved + pos * a.gate_stride + u;
            gi = ld4(sp); gf = ld4(sp + H); gg = ld4(sp + 2 * H); go = ld4(sp + 3 * H);
            cc = ld4(a.cc + pos * a.c_stride + u);
            if (pos >= g.stride) cp = ld4(a.cc + (pos - g.stride) * a.c_stride + u);
            if (pt == g.p0 + ps - 1) de = ld4(a.d_emit + ((int64_t)ps * g.n + pi) * H + u);
        }

        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;

        if (mp > 0) {
            constexpr int NK = H / BK2;   // each wave walks its own 512 gate rows
            float4 ring_b[PDEPTH][4];
            v4u ring_a[PDEPTH][4];
#pragma unroll
            for (int d = 0; d < PDEPTH; ++d)
#pragma unroll
                for (int i = 0; i < 4; ++i) ring_b[d][i] = B_LOAD(i, d);
            // the first step walked here reads what an earlier launch left: nothing to wait for, and no flag
            // has told yet where the row group's workgroups run
            if (w == 0) {
                const bool l = mp > m0 && wait_row_group(a.flags + by * 32, (unsigned)(mp - m0), abort_w, lane, xcd);
                if (lane == 0) *local_w = l;
            }
            __syncthreads();
            local = *local_w != 0;
            const int a_base = (cur ^ 1) * g.K * 4 * H * 4;
#pragma unroll
            for (int d = 0; d < PDEPTH; ++d)
#pragma unroll
                for (int i = 0; i < 4; ++i) ring_a[d][i] = A_LOAD(i, d, a_base);
#define STORE_STAGE2(buf, slot)                                                                     \
    {                                                                                               \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                             \
            st4(tile + (buf) + a_dst + 8 * i * LDA2, as_f4(ring_a[slot][i]));                       \
            st4(tile + (buf) + b_dst + 8 * i * 32, ring_b[slot][i]);                                \
        }                                                                                           \
    }
#define LOAD_STAGE2(slot, stage)                                                                    \
    {                                                                                               \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                             \
            ring_a[slot][i] = A_LOAD(i, stage, a_base);                                             \
            ring_b[slot][i] = B_LOAD(i, stage);                                                     \
        }                                                                                           \
    }
            STORE_STAGE2(0, 0)
            if (PDEPTH < NK) LOAD_STAGE2(0, PDEPTH)
            WAVE_SYNC()
            FragB2 f0 = load_frag2(tile, 0, l31, hh), f1;
#pragma unroll
            for (int kt = 0; kt < NK; ++kt) {
                const int co = (kt & 1) * PBUF2, cn = ((kt + 1) & 1) * PBUF2;
                f1 = load_frag2(tile + co, 1, l31, hh);
                mfma8(acc, f0);
                if (kt + 1 < NK) {   // this wave's reads of buffer `cn` were issued before these stores
                    STORE_STAGE2(cn, (kt + 1) % PDEPTH)
                    if (kt + 1 + PDEPTH < NK) LOAD_STAGE2((kt + 1) % PDEPTH, kt + 1 + PDEPTH)
                }
                // issue order: behind each matrix instruction one LDS read, one LDS write and one global
                // load, so that the matrix unit never waits for a block of staging instructions
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                if (kt + 1 < NK) {
                    WAVE_SYNC()
                    f0 = load_frag2(tile + cn, 0, l31, hh);
                }
                mfma8(acc, f1);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
#undef STORE_STAGE2
#undef LOAD_STAGE2
            WAVE_SYNC()   // the partial tile below overlays this wave's own stage buffer only
        }
        float* P = lds;
        acc_to_lds(acc, tile, l31, hh);
        __syncthreads();

        if (have_row) {
            const float* pr = P + rr * PLD + uq;
            float4 dh = add4(add4(ldp4(pr), ldp4(pr + TILE2)), add4(ldp4(pr + 2 * TILE2), ldp4(pr + 3 * TILE2)));
            if (final) {
                st4(a.dh_fin + (int64_t)j * H + u, dh);
                st4(a.dc + (int64_t)j * H + u, dcv);
            } else {
                if (qs == g.Ls - 1) {  // state entering the segment's last own position
                    st4(a.seam_a + (int64_t)j * 2 * H + u, dh);
                    st4(a.seam_a + (int64_t)j * 2 * H + H + u, dcv);
                }
                float4 d_i = zero4(), d_f = zero4(), d_g = zero4(), d_o = zero4(), dcn = zero4();
                if (live) {
                    dh = add4(dh, de);
                    LSTM_BWD4(dh, dcv, gi, gf, gg, go, cc, cp, d_i, d_f, d_g, d_o, dcn)
                    if (qs < g.Ls) {
                        float* dp = a.dgv + pos * (4 * H) + u;
                        st4(dp, d_i); st4(dp + H, d_f); st4(dp + 2 * H, d_g); st4(dp + 3 * H, d_o);
                    }
                }
                dcv = dcn;
                // the exchanged copy of this step's dgates: write-through, 16 bytes per lane
                const int o = (j * 4 * H + u) * 4, sb = cur * g.K * 4 * H * 4;
                if (local) {   // every reader shares this XCD's L2: no write-through needed
                    __builtin_amdgcn_raw_buffer_store_b128(as_u4(d_i), rs, o, sb, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(as_u4(d_f), rs, o + H * 4, sb, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(as_u4(d_g), rs, o + 2 * H * 4, sb, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(as_u4(d_o), rs, o + 3 * H * 4, sb, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(as_u4(d_i), rs, o, sb, SC1);
                    __builtin_amdgcn_raw_buffer_store_b128(as_u4(d_f), rs, o + H * 4, sb, SC1);
                    __builtin_amdgcn_raw_buffer_store_b128(as_u4(d_g), rs, o + 2 * H * 4, sb, SC1);
                    __builtin_amdgcn_raw_buffer_store_b128(as_u4(d_o), rs, o + 3 * H * 4, sb, SC1);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its stores ...
        __syncthreads();                                    // ... before ONE lane signals for all of them
        if (tid == 0 && !final) {
            const unsigned word = (unsigned)(mp + 1 - m0) | (xcd << 24);
            gu32* fp = (gu32*)(a.flags + by * 32 + bx);
            if (local) __hip_atomic_store(fp, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_store(fp, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#undef WAVE_SYNC
#undef A_LOAD
#undef B_LOAD
}

// ------------------------------------------------------------------ the first steps of the backward warm-up, at 16-bit precision
// The backward warm-up (the steps after a segment's last own position, walked down from a zero (dh, dc)) is
// thrown away like the forward one: what matters is the (dh, dc) it leaves, and the backward seam check measures
// that.  Its first steps run here with both operands of dh = dgates . W_hh cut in two bf16 pieces (the three
// products hi.hi + hi.mid + mid.hi on the bf16 matrix instruction, 2^-16 relative error); seg_bwd_persist
// continues from the dgates and dc left here (BwdArgs::m_begin).
//
// Workgroup (unit tile bx of 16 units, row group by of 32 segments); wave w takes the 512 gate rows of gate w as
// its k range and keeps its 512 x 16 slice of W_hh for the whole launch: the hi pieces in registers (64 VGPRs),
// the mid pieces in LDS (64 KB per workgroup, one 16-byte read per k step) - with both in registers the compiler
// has no room left to keep more than four of a step's 64 operand loads in flight.
// The dgates are exchanged ALREADY cut in two: every producer stores the hi and mid pieces of its 4 x 2 values
// into two bf16 planes (the same bytes as float32 dgates), so a consumer loads matrix-instruction operands
// straight from memory into registers - 16 bytes = 8 consecutive gate rows of one segment - with no LDS staging
// and no barrier in the k loop; the four waves' partial tiles meet in LDS.  Hand-off as in seg_bwd_persist:
// write-through stores, drained, one flag per workgroup and step, sc1 loads after the poll - and plain stores
// once the XCD ids in the flags have shown that the row group's 32 workgroups share an XCD (the sc1 loads then
// find the lines in that XCD's L2 instead of fetching 256 KB per workgroup and step through the fabric).
// Planes of parity p, row group by: dgx + p * K * 4H floats + by * 32 * 4H floats, {hi[32][4H], mid[32][4H]} bf16 -
// exactly the bytes of the row group's float32 rows dgx[p][32 by ..][4H], which the LAST step writes instead
// (only the row group's own workgroups ever read those bytes, and the flags say they are done).  K % 32 == 0.
struct WarmBwdArgs {
    const float* w_hh; const float* saved; const float* d_emit;
    float* dgx;               // [2][K][4H] floats: bf16 planes while the steps run, float32 dgates of the last step
    float* dc;                // [K][H]: dc after the last step
    unsigned* flags;          // [row groups][32]
    Geom g;
    int steps;                // steps mp = 0 .. steps - 1 of the chain's W + Ls + 1 are walked here (steps <= W)
};

// wave 0: lanes 0..31 each poll one producer's flag (step count in the low 24 bits, the producer's XCD above), lane 32
// polls the abort word.  Returns whether all 32 producers run on XCD `xcd`.
__device__ __forceinline__ bool wait_row_group32_xcd(unsigned* flags32, unsigned epoch, unsigned* abort_w, int lane,
                                                     unsigned xcd) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (;;) {
        unsigned v = epoch | (xcd << 24), ab = 0;
        if (lane < 32) v = __hip_atomic_load((gu32*)(flags32 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (lane == 32) ab = __hip_atomic_load((gu32*)abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__any(ab != 0)) return false;
        if (__all((v & EPOCH_MASK) >= epoch)) return __all((v >> 24) == xcd);
        if (__builtin_amdgcn_s_memrealtime() - t0 > SPIN_LIMIT) {
            give_up(abort_w, epoch, __ballot(lane < 32 && (v & EPOCH_MASK) < epoch), lane);
            return false;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

__device__ __forceinline__ unsigned pack_bf16x2(__bf16 a, __bf16 b) {
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}

__global__ __launch_bounds__(256, 1) void seg_warm_bwd(WarmBwdArgs a, unsigned* abort_w, int ny) {
    __shared__ float gt[4 * 32 * WGT + 4];   // the four waves' partial [segment row][unit] tiles; one word for `local`
    __shared__ __attribute__((aligned(16))) v4u wmid[4 * 16 * 64];   // [wave][k step][lane]: mid pieces of W_hh
    __builtin_amdgcn_s_setprio(3);
    // block b runs on XCD b % 8; a row group's 32 workgroups are given block ids of one XCD (speed only: whether
    // they really share one is read from the flags)
    const int wb = blockIdx.x, t = wb >> 3;
    const int by = (t >> 5) * 8 + (wb & 7), bx = t & 31;
    if (by >= ny) return;
    const unsigned xcd = my_xcd();
    int* local_w = reinterpret_cast<int*>(gt + 4 * 32 * WGT);
    bool local = false;
    const Geom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l15 = lane & 15, kg = lane >> 4;
    const int row0 = by * 32, u0 = bx * 16;
    const int parity_bytes = g.K * 4 * H * 4, group_bytes = by * 32 * 4 * H * 4;
    constexpr int PLANE_BYTES = 32 * 4 * H * 2;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.dgx, 0, 2 * parity_bytes, 0x00020000);

    // B operand of lane (column l15, k group kg) at k step ks: W_hh[gate row w H + 32 ks + 8 kg + 0..7][unit u0 + l15]
    wbf16x8 bw[16];
    v4u* wm = wmid + w * 16 * 64 + lane;
    {
        const float* wcol = a.w_hh + ((int64_t)w * H + 8 * kg) * H + u0 + l15;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            wbf16x8 bm;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                __bf16 hi, mid;
                split2(wcol[(int64_t)(32 * ks + e) * H], hi, mid);
                bw[ks][e] = hi;
                bm[e] = mid;
            }
            wm[ks * 64] = __builtin_bit_cast(v4u, bm);   // read back by this lane only
        }
    }
    // A operand of lane (row l15, k group kg): 8 consecutive gate rows of segment row0 + 16 mt + l15, from a plane
    const int a_voff = (l15 * 4 * H + w * H + 8 * kg) * 2;
#define WB_LOAD(mt, plane, ks, base) \
    __builtin_amdgcn_raw_buffer_load_b128(rs, a_voff + ((mt) * 16 * 4 * H + 32 * (ks)) * 2 + (plane) * PLANE_BYTES, (base), SC1)

    // this lane's epilogue item: segment row j, units u, u + 1; its dc lives in registers
    const int rr = tid >> 3, uq = 2 * (tid & 7);
    const int j = row0 + rr, u = u0 + uq;
    const bool have_row = j < g.K;
    float2 dcv = make_float2(0.f, 0.f);
    int ps = 0, pt = 0, pi = 0;
    const int st_voff = (rr * 4 * H + u) * 2;   // this item's place in a plane (+ gate * H)

    for (int mp = 0; mp < a.steps; ++mp) {
        const bool last = mp == a.steps - 1;
        const int cur = mp & 1;   // this step writes parity cur; it reads parity cur ^ 1
        const int qs = g.Ls + g.W - 1 - mp;   // >= Ls: every step here is a warm-up step
        const int64_t pos = seg_pos(g, j, qs);
        const bool live = have_row && pos < g.total && seg_pos(g, j, 0) < g.total;
        const float2 z2 = make_float2(0.f, 0.f);
        float2 gi = z2, gf = z2, gg = z2, go = z2, cc = z2, cp = z2, de = z2;
        if (live) {
            locate_fast(g, (int)pos, ps, pt, pi);
            const float* sp = a.saved + pos * (6 * H) + u;
#define LD2(p) (*reinterpret_cast<const float2*>(p))
            gi = LD2(sp); gf = LD2(sp + H); gg = LD2(sp + 2 * H); go = LD2(sp + 3 * H); cc = LD2(sp + 4 * H);
            if (pos >= g.stride) cp = LD2(a.saved + (pos - g.stride) * (6 * H) + 4 * H + u);
            if (pt == g.p0 + ps - 1) de = LD2(a.d_emit + ((int64_t)ps * g.n + pi) * H + u);
#undef LD2
        }
        f32x4 acc[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[mt][r] = 0.f;

        if (mp > 0) {
            if (w == 0) {
                const bool l = wait_row_group32_xcd(a.flags + by * 32, (unsigned)mp, abort_w, lane, xcd);
                if (lane == 0) *local_w = l;
            }
            __syncthreads();
            local = *local_w != 0;
            const int base = (cur ^ 1) * parity_bytes + group_bytes;
            // batches of two k steps (8 loads of 16 bytes: 2 row tiles x 2 planes x 2 k steps), three in flight
            constexpr int NB = 8, DEPTH = 3;
            v4u ring[DEPTH][2][2][2];
#pragma unroll
            for (int d = 0; d < DEPTH; ++d)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int pl = 0; pl < 2; ++pl) ring[d][kk][mt][pl] = WB_LOAD(mt, pl, 2 * d + kk, base);
            // the scheduler would sink every load to its first use (four in flight instead of 24): nothing may
            // cross the lines between a batch's matrix instructions and the loads that refill its slot
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int slot = b % DEPTH;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int ks = 2 * b + kk;
                    const wbf16x8 bm = __builtin_bit_cast(wbf16x8, wm[ks * 64]);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const wbf16x8 a_hi = __builtin_bit_cast(wbf16x8, ring[slot][kk][mt][0]);
                        const wbf16x8 a_mid = __builtin_bit_cast(wbf16x8, ring[slot][kk][mt][1]);
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, bw[ks], acc[mt], 0, 0, 0);
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi, bm, acc[mt], 0, 0, 0);
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_mid, bw[ks], acc[mt], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (b + DEPTH < NB) {   // the slot is free again
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                            for (int pl = 0; pl < 2; ++pl)
                                ring[slot][kk][mt][pl] = WB_LOAD(mt, pl, 2 * (b + DEPTH) + kk, base);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // accumulator register r of lane (l15, kg) is element [row 4 kg + r][column l15] of its 16 x 16 tile
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) gt[(w * 32 + mt * 16 + 4 * kg + r) * WGT + l15] = acc[mt][r];
        __syncthreads();

        if (have_row) {
            const float* pr = gt + rr * WGT + uq;
            float2 dh = make_float2((pr[0] + pr[32 * WGT]) + (pr[64 * WGT] + pr[96 * WGT]),
                                    (pr[1] + pr[32 * WGT + 1]) + (pr[64 * WGT + 1] + pr[96 * WGT + 1]));
            float2 d_i = z2, d_f = z2, d_g = z2, d_o = z2, dcn = z2;
            if (live) {
                dh.x += de.x; dh.y += de.y;
                LSTM_BWD1(x, dh, dcv, gi, gf, gg, go, cc, cp, d_i, d_f, d_g, d_o, dcn)
                LSTM_BWD1(y, dh, dcv, gi, gf, gg, go, cc, cp, d_i, d_f, d_g, d_o, dcn)
            }
            dcv = dcn;
            if (last) {   // what the float32 launch continues from
                float* dp = a.dgx + (int64_t)cur * g.K * 4 * H + (int64_t)j * 4 * H + u;
                *reinterpret_cast<float2*>(dp) = d_i; *reinterpret_cast<float2*>(dp + H) = d_f;
                *reinterpret_cast<float2*>(dp + 2 * H) = d_g; *reinterpret_cast<float2*>(dp + 3 * H) = d_o;
                *reinterpret_cast<float2*>(a.dc + (int64_t)j * H + u) = dcv;
            } else {
                const int sb = cur * parity_bytes + group_bytes;
                const float2 dv[4] = {d_i, d_f, d_g, d_o};
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    __bf16 hx, mx, hy, my;
                    split2(dv[gq].x, hx, mx);
                    split2(dv[gq].y, hy, my);
                    if (local) {   // every reader shares this XCD's L2: no write-through needed
                        __builtin_amdgcn_raw_buffer_store_b32(pack_bf16x2(hx, hy), rs, st_voff + gq * H * 2, sb, 0);
                        __builtin_amdgcn_raw_buffer_store_b32(pack_bf16x2(mx, my), rs, st_voff + gq * H * 2 + PLANE_BYTES, sb, 0);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b32(pack_bf16x2(hx, hy), rs, st_voff + gq * H * 2, sb, SC1);
                        __builtin_amdgcn_raw_buffer_store_b32(pack_bf16x2(mx, my), rs, st_voff + gq * H * 2 + PLANE_BYTES, sb, SC1);
                    }
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its stores ...
        __syncthreads();                                    // ... before ONE lane signals for all of them
        if (tid == 0 && !last) {
            const unsigned word = (unsigned)(mp + 1) | (xcd << 24);
            gu32* fp = (gu32*)(a.flags + by * 32 + bx);
            if (local) __hip_atomic_store(fp, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_store(fp, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#undef WB_LOAD
}


// ------------------------------------------------------------------ LSTM backward chain: owner computes, partial sums exchanged, f16 matrix instruction
// dh(p) = dgates(p+1) . W_hh sums over all 2048 gate rows, and the gate gradients of a position are produced 128 at
// a time by the workgroup that owns their 32 units.  seg_bwd_persist gathers: every workgroup of a row group pulls
// the group's whole 32 x 2048 dgates panel (256 KB) each step.  Here a workgroup multiplies only what it produced
// itself: partial[32 segments][512 units] = own dgates [32][128] . W_hh[its 128 gate rows][512], and hands each of the
// 16 workgroups of its row group the 32 x 32 tile that belongs to that one's units; a workgroup then adds up the 16
// tiles it was sent.  64 KB out and 64 KB in per workgroup and step, and the dgates never leave the workgroup - so
// they can be cut into f16 pieces with a scale of their own per segment row (the largest of the row's 128 values
// brought into [2^13, 2^14); gradients span too many binades for one scale, and the exchanged partial sums are
// float32, descaled exactly).  W_hh's slice is cut into two f16 pieces once (one power-of-two scale per wave) and
// stays in registers; products as in hp_chain: hi.hi + hi.mid + mid.hi, float32 sums, 2^-21 per element product.
// 512 threads: the eight waves split the 512 output units; threads 0..255 each own one (segment row, 4 units) item
// of the cell's backward pass.  Hand-off: write-through stores, drained, one flag per workgroup and step, sc1 loads.
constexpr int RS_LDB = 128 + 8;     // f16 per LDS row of a dgates plane: the workgroup's 128 gate rows, padded

struct RsArgs {
    const float* w_hh; const float* d_emit;
    const float* saved; const float* cc;   // activations of the forward pass, as in BwdArgs
    int gate_stride, c_stride;
    float* part;              // [2][row groups][16 receivers][16 senders][32 rows][32 units] partial products
    float* dc; float* dgv; float* seam_a; float* dh_fin;
    unsigned* flags;          // [row groups][16]
    Geom g;
};

__global__ __launch_bounds__(512, 1) void seg_bwd_rs(RsArgs a, unsigned* abort_w) {
    __shared__ __attribute__((aligned(16))) _Float16 planes[2 * 32 * RS_LDB];   // [hi | mid][segment row][own gate row]
    __shared__ float row_descale[32];
    __builtin_amdgcn_s_setprio(3);
    const Geom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l15 = lane & 15, kg = lane >> 4;
    const int bx = blockIdx.x % 16, by = blockIdx.x / 16;
    const int groups = gridDim.x / 16;
    const int row0 = by * 32, u0 = bx * 32;
    constexpr int TILE_B = 32 * 32 * 4;                                 // bytes of one 32 x 32 partial tile
    const int parity_b = groups * 16 * 16 * TILE_B;                     // bytes of one parity of the exchange buffer
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(a.part, 0, 2 * parity_b, 0x00020000);

    // A operand of lane (output unit l15 of tile mt, k group kg) at k step ks: W_hh[gate ks, unit u0 + 8 kg + 0..7][64 w + 16 mt + l15]
    hf16x8 wa[2][4][4];
    float descale_w;
    {
        const float* wcol = a.w_hh + (int64_t)(u0 + 8 * kg) * H + 64 * w + l15;
        float top = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int e = 0; e < 8; ++e) top = fmaxf(top, fabsf(wcol[((int64_t)ks * H + e) * H + 16 * mt]));
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) top = fmaxf(top, __shfl_xor(top, sh));
        int ex = 0;
        if (top > 0.f && top < INFINITY) frexpf(top, &ex);
        ex = max(-100, min(100, ex));
        const float scale = ldexpf(1.f, 11 - ex);       // the wave's largest entry in [2^10, 2^11)
        descale_w = ldexpf(1.f, ex - 11);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    _Float16 hi, mid;
                    split2h(wcol[((int64_t)ks * H + e) * H + 16 * mt] * scale, hi, mid);
                    wa[0][mt][ks][e] = hi;
                    wa[1][mt][ks][e] = mid;
                }
    }

    // threads 0..255: the epilogue item (segment row j, units u..u+3) and its carried dc
    const bool item = tid < 256;
    const int rr = (tid >> 3) & 31, quad = tid & 7;
    const int j = row0 + rr, u = u0 + 4 * quad;
    const bool have_row = item && j < g.K;
    float4 dcv = zero4();
    int ps = 0, pt = 0, pi = 0;
    const int in_voff = (rr * 32 + 4 * quad) * 4;                       // this item's place in a received tile
    const int in_soff = ((by * 16 + bx) * 16) * TILE_B;                 // + sender * TILE_B + parity
    // where this lane's products go: receiver 2 w + mt / 2, rows 16 nt + l15, units 16 (mt % 2) + 4 kg .. + 3
    const int out_voff = (l15 * 32 + 4 * kg) * 4;

    const int steps = g.W + g.Ls;
    for (int mp = 0; mp <= steps; ++mp) {
        const bool final = mp == steps;
        const int cur = mp & 1;                  // this step writes parity cur and reads parity cur ^ 1
        const int qs = g.Ls + g.W - 1 - mp;      // step within the segment: >= Ls during the warm-up
        const int64_t pos = seg_pos(g, j, qs);
        const bool live = have_row && !final && pos < g.total && seg_pos(g, j, 0) < g.total;
        float4 gi = zero4(), gf = zero4(), gg = zero4(), go = zero4(), cc = zero4(), cp = zero4(), de = zero4();
        if (live) {
            locate_fast(g, (int)pos, ps, pt, pi);
            const float* sp = a.saved + pos * a.gate_stride + u;
            gi = ld4(sp); gf = ld4(sp + H); gg = ld4(sp + 2 * H); go = ld4(sp + 3 * H);
            cc = ld4(a.cc + pos * a.c_stride + u);
            if (pos >= g.stride) cp = ld4(a.cc + (pos - g.stride) * a.c_stride + u);
            if (pt == g.p0 + ps - 1) de = ld4(a.d_emit + ((int64_t)ps * g.n + pi) * H + u);
        }

        float4 dh = zero4();
        if (mp > 0) {
            if (w == 0) wait_row_group(a.flags + by * 16, (unsigned)mp, abort_w, lane);
            __syncthreads();
            if (item) {   // the 16 tiles this workgroup was sent, added up in a fixed order
                v4u raw[16];
                const int so = (cur ^ 1) * parity_b + in_soff;
#pragma unroll
                for (int sd = 0; sd < 16; ++sd) raw[sd] = __builtin_amdgcn_raw_buffer_load_b128(rs, in_voff, so + sd * TILE_B, SC1);
                float4 s0 = add4(add4(as_f4(raw[0]), as_f4(raw[1])), add4(as_f4(raw[2]), as_f4(raw[3])));
                float4 s1 = add4(add4(as_f4(raw[4]), as_f4(raw[5])), add4(as_f4(raw[6]), as_f4(raw[7])));
                float4 s2 = add4(add4(as_f4(raw[8]), as_f4(raw[9])), add4(as_f4(raw[10]), as_f4(raw[11])));
                float4 s3 = add4(add4(as_f4(raw[12]), as_f4(raw[13])), add4(as_f4(raw[14]), as_f4(raw[15])));
                dh = add4(add4(s0, s1), add4(s2, s3));
            }
        }
        float4 d_i = zero4(), d_f = zero4(), d_g = zero4(), d_o = zero4(), dcn = zero4();
        if (have_row) {
            if (final) {
                st4(a.dh_fin + (int64_t)j * H + u, dh);
                st4(a.dc + (int64_t)j * H + u, dcv);
            } else {
                if (qs == g.Ls - 1) {  // state entering the segment's last own position
                    st4(a.seam_a + (int64_t)j * 2 * H + u, dh);
                    st4(a.seam_a + (int64_t)j * 2 * H + H + u, dcv);
                }
                if (live) {
                    dh = add4(dh, de);
                    LSTM_BWD4(dh, dcv, gi, gf, gg, go, cc, cp, d_i, d_f, d_g, d_o, dcn)
                    if (qs < g.Ls) {
                        float* dp = a.dgv + pos * (4 * H) + u;
                        st4(dp, d_i); st4(dp + H, d_f); st4(dp + 2 * H, d_g); st4(dp + 3 * H, d_o);
                    }
                }
                dcv = dcn;
            }
        }
        if (final) break;

        // this step's own dgates, scaled per segment row and cut in two, into LDS as the B operand
        if (item) {
            float top = fmaxf(fmaxf(fmaxf(fabsf(d_i.x), fabsf(d_i.y)), fmaxf(fabsf(d_i.z), fabsf(d_i.w))),
                              fmaxf(fmaxf(fabsf(d_f.x), fabsf(d_f.y)), fmaxf(fabsf(d_f.z), fabsf(d_f.w))));
            top = fmaxf(top, fmaxf(fmaxf(fmaxf(fabsf(d_g.x), fabsf(d_g.y)), fmaxf(fabsf(d_g.z), fabsf(d_g.w))),
                                   fmaxf(fmaxf(fabsf(d_o.x), fabsf(d_o.y)), fmaxf(fabsf(d_o.z), fabsf(d_o.w)))));
            top = fmaxf(top, __shfl_xor(top, 1));     // the eight lanes of a segment row are neighbours
            top = fmaxf(top, __shfl_xor(top, 2));
            top = fmaxf(top, __shfl_xor(top, 4));
            int ex = 0;
            if (top > 0.f && top < INFINITY) frexpf(top, &ex);
            ex = max(-110, min(110, ex));
            const float scale = ldexpf(1.f, 14 - ex);   // the row's largest value in [2^13, 2^14)
            if (quad == 0) row_descale[rr] = ldexpf(1.f, ex - 14);
            const float4 dv[4] = {d_i, d_f, d_g, d_o};
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                hf16x4 hi, mid;
                _Float16 h_, m_;
                split2h(dv[gq].x * scale, h_, m_); hi[0] = h_; mid[0] = m_;
                split2h(dv[gq].y * scale, h_, m_); hi[1] = h_; mid[1] = m_;
                split2h(dv[gq].z * scale, h_, m_); hi[2] = h_; mid[2] = m_;
                split2h(dv[gq].w * scale, h_, m_); hi[3] = h_; mid[3] = m_;
                *reinterpret_cast<hf16x4*>(planes + rr * RS_LDB + 32 * gq + 4 * quad) = hi;
                *reinterpret_cast<hf16x4*>(planes + (32 + rr) * RS_LDB + 32 * gq + 4 * quad) = mid;
            }
        }
        __syncthreads();

        // partial[unit 64 w + 16 mt + ..][segment row 16 nt + ..] over the workgroup's 128 gate rows
        f32x4 acc[4][2];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[mt][nt][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            hf16x8 b_hi[2], b_mid[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const _Float16* br = planes + (16 * nt + l15) * RS_LDB + 32 * ks + 8 * kg;
                b_hi[nt] = *reinterpret_cast<const hf16x8*>(br);
                b_mid[nt] = *reinterpret_cast<const hf16x8*>(br + 32 * RS_LDB);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[0][mt][ks], b_hi[nt], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[0][mt][ks], b_mid[nt], acc[mt][nt], 0, 0, 0);
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[1][mt][ks], b_hi[nt], acc[mt][nt], 0, 0, 0);
                }
        }
        // accumulator register r of lane (l15, kg) is [unit 4 kg + r of tile mt][segment row l15 of tile nt]:
        // four consecutive units of one row, 16 bytes of the tile the receiver 2 w + mt / 2 reads
        {
            const float ds[2] = {descale_w * row_descale[l15], descale_w * row_descale[16 + l15]};
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int so = cur * parity_b + ((by * 16 + 2 * w + (mt >> 1)) * 16 + bx) * TILE_B + (mt & 1) * 64;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const float4 v = make_float4(acc[mt][nt][0] * ds[nt], acc[mt][nt][1] * ds[nt], acc[mt][nt][2] * ds[nt],
                                                 acc[mt][nt][3] * ds[nt]);
                    __builtin_amdgcn_raw_buffer_store_b128(as_u4(v), rs, out_voff + nt * 16 * 32 * 4, so, SC1);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its stores ...
        __syncthreads();                                    // ... before ONE lane signals for all of them
        if (tid == 0)
            __hip_atomic_store((gu32*)(a.flags + by * 16 + bx), (unsigned)(mp + 1), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
}

// after the seam checks: a launch that gave up waiting must not pass them
__global__ void poison_on_abort(const unsigned* abort_w, float* seam_v, float* seam_r, int kind) {
    if (*abort_w != 0) { *seam_v = INFINITY; *seam_r = INFINITY; }
    report_abort(abort_w, kind);
}
int seam_blocks(int64_t elems) { return (int)std::min<int64_t>(128, (elems + 255) / 256); }

// Up to five regions cleared by ONE launch (a persistent chain launch is preceded by that many small clears -
// polled words, exchanged state, seam errors - and each hipMemsetAsync is a launch of its own on the step's
// critical path).  Counts in floats; a null pointer or a zero count skips the region.
struct ZeroArgs { float* p[5]; int64_t n[5]; };
__global__ __launch_bounds__(256) void zero_regions(ZeroArgs z) {
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        if (z.p[r] == nullptr) continue;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < z.n[r]; i += (int64_t)gridDim.x * 256) z.p[r][i] = 0.f;
    }
}

// a launch that gave up waiting, reported through a status word (the per-row chains have no seam error)
__global__ void flag_on_abort(const unsigned* abort_w, int* status, int kind) {
    if (*abort_w != 0) *status = 1;
    report_abort(abort_w, kind);
}

// Workspace layouts (floats, kh = K * 512), K segments:
//   forward LSTM : h[2][kh], c[2][kh], seam_a[2 kh], seam_b[2 kh]
//   forward GRU  : h[2][kh], seam_a[kh], seam_b[kh]
//   backward     : dgates[2][4 kh], dc[kh], seam_a[2 kh], dh_fin[kh]
//   all          : polled words of the persistent launches from 12 kh
int fwd_steps_lstm(const Geom& g, const float* xproj, const float* w_hh, float* h_emit, float* saved, float* work,
                    const float* state_in, hipStream_t st) {
    const int64_t kh = (int64_t)g.K * H;
    float* hb[2] = {work, work + kh};
    float* cb[2] = {work + 2 * kh, work + 3 * kh};
    ICRL_HIP(hipMemsetAsync(work, 0, sizeof(float) * 4 * kh, st));
    const dim3 grid(H / 32, (g.K + 31) / 32);
    for (int m = 0; m < g.W + g.Ls; ++m) {
        const int cur = m & 1;
        hipLaunchKernelGGL(seg_fwd_step<4>, grid, dim3(256), 0, st, hb[cur], w_hh, xproj, (const float*)nullptr,
                           cb[cur], cb[cur ^ 1], hb[cur ^ 1], saved, h_emit, work + 4 * kh, work + 6 * kh, state_in, g, m);
    }
    return ICRL_OK;
}

int fwd_steps_gru(const Geom& g, const float* xproj, const float* w_hh, const float* b_hh, float* h_emit,
                   float* saved, float* work, hipStream_t st) {
    const int64_t kh = (int64_t)g.K * H;
    float* hb[2] = {work, work + kh};
    ICRL_HIP(hipMemsetAsync(work, 0, sizeof(float) * 2 * kh, st));
    const dim3 grid(H / 32, (g.K + 31) / 32);
    for (int m = 0; m < g.W + g.Ls; ++m) {
        const int cur = m & 1;
        hipLaunchKernelGGL(seg_fwd_step<3>, grid, dim3(256), 0, st, hb[cur], w_hh, xproj, b_hh, (const float*)nullptr,
                           (float*)nullptr, hb[cur ^ 1], saved, h_emit, work + 2 * kh, work + 3 * kh,
                           (const float*)nullptr, g, m);
    }
    return ICRL_OK;
}

int bwd_steps(const Geom& g, const float* w_hh, const float* saved, const float* d_emit, float* dgates, float* work,
               hipStream_t st) {
    const int64_t kh = (int64_t)g.K * H;
    float* dg[2] = {work, work + 4 * kh};
    float* dc = work + 8 * kh;
    float* seam_a = work + 9 * kh;
    float* dh_fin = work + 11 * kh;
    ICRL_HIP(hipMemsetAsync(dc, 0, sizeof(float) * 3 * kh, st));
    const dim3 grid(H / 32, (g.K + 31) / 32);
    const int steps = g.W + g.Ls;
    for (int mp = 0; mp < steps; ++mp)
        hipLaunchKernelGGL(seg_bwd_step, grid, dim3(256), 0, st, dg[(mp & 1) ^ 1], w_hh, saved, d_emit, dc, dg[mp & 1],
                           dgates, seam_a, dh_fin, g, mp, 0);
    hipLaunchKernelGGL(seg_bwd_step, grid, dim3(256), 0, st, dg[(steps & 1) ^ 1], w_hh, saved, d_emit, dc,
                       dg[steps & 1], dgates, seam_a, dh_fin, g, steps, 1);
    return ICRL_OK;
}

// both forward chains in one persistent launch; returns the abort word through *abort_out.  cheap_v / cheap_r:
// the first steps of each chain's warm-up that run on seg_warm_pair (16-bit split products) before it.
// phase: 0 = everything, 1 = up to and including the cheap warm-up launch, 2 = the float32 launch alone (after a
// phase-1 call on the same workspaces).
int pair_persist(ChainArgs& v, ChainArgs& r, float* work_v, float* work_r, hipStream_t st, unsigned** abort_out,
                 int cheap_v = 0, int cheap_r = 0, int phase = 0, float* err_a = nullptr, float* err_b = nullptr,
                 int err_floats = 0) {
    const int K = v.g.K;
    const int64_t kh = (int64_t)K * H;
    const int ny = (K + 31) / 32;
    // polled words (flags of both chains, abort word, flags of the cheap warm-up): a block of their own, zeroed on every call
    unsigned* words = reinterpret_cast<unsigned*>(work_v + 12 * kh);
    const size_t word_bytes = ((size_t)(2 * ny * 16 + 4 + 2 * ny * 32) * sizeof(unsigned) + 255) / 256 * 256;
    ICRL_REQUIRE((int64_t)(word_bytes / sizeof(float)) <= 4 * kh, "persistent chain launch: workspace too small");
    if (phase != 2) {   // polled words, both chains' exchanged state and (err_a, err_b: the caller's error words)
        ZeroArgs z{{reinterpret_cast<float*>(words), work_v, work_r, err_a, err_b},
                   {(int64_t)(word_bytes / sizeof(float)), 2 * kh, 2 * kh, err_floats, err_floats}};
        hipLaunchKernelGGL(zero_regions, dim3(128), dim3(256), 0, st, z);
    }
    v.hb = work_v; v.seam_a = work_v + 4 * kh; v.seam_b = work_v + 6 * kh; v.flags = words;
    r.hb = work_r; r.seam_a = work_r + 2 * kh; r.seam_b = work_r + 3 * kh; r.flags = words + ny * 16;
    *abort_out = words + 2 * ny * 16;
    if (cheap_v > 0 || cheap_r > 0) {
        WarmArgs wv{}, wr{};
        unsigned* wflags = words + 2 * ny * 16 + 4;
        wv.xproj = v.xproj; wv.w_hh = v.w_hh; wv.b_hh = nullptr; wv.hb = v.hb; wv.state_out = work_v + 2 * kh;
        wv.flags = wflags; wv.g = v.g; wv.steps = cheap_v;
        wr.xproj = r.xproj; wr.w_hh = r.w_hh; wr.b_hh = r.b_hh; wr.hb = r.hb; wr.state_out = work_r + 4 * kh;
        wr.flags = wflags + ny * 32; wr.g = r.g; wr.steps = cheap_r;
        const int warm_per_chain = (ny + 7) / 8 * 8 * 32;
        if (phase != 2)
            hipLaunchKernelGGL(seg_warm_pair, dim3(2 * warm_per_chain), dim3(256), WARM_LDS, st, wv, wr, *abort_out,
                               warm_per_chain, ny);
        v.warm_state = wv.state_out; v.m_begin = cheap_v;
        r.warm_state = wr.state_out; r.m_begin = cheap_r;
    }
    if (phase == 1) return ICRL_OK;
    const int per_chain = (H / 32) * ny;
    hipLaunchKernelGGL(seg_fwd_pair_persist, dim3(2 * per_chain), dim3(256), 0, st, v, r, *abort_out, per_chain, 0);
    return ICRL_OK;
}


// both forward chains, every step, in the f16 launch (seg_fwd_pair_hp); workspace layout as pair_persist's
int pair_hp(HpArgs& v, HpArgs& r, float* work_v, float* work_r, hipStream_t st, unsigned** abort_out, float* err_a,
            float* err_b) {
    const int K = v.g.K;
    const int64_t kh = (int64_t)K * H;
    const int ny = (K + 31) / 32;
    unsigned* words = reinterpret_cast<unsigned*>(work_v + 12 * kh);
    const size_t word_bytes = ((size_t)(2 * ny * 32 + 4) * sizeof(unsigned) + 255) / 256 * 256;
    ICRL_REQUIRE((int64_t)(word_bytes / sizeof(float)) <= 4 * kh, "persistent chain launch: workspace too small");
    ZeroArgs z{{reinterpret_cast<float*>(words), work_v, work_r, err_a, err_b},
               {(int64_t)(word_bytes / sizeof(float)), 2 * kh, 2 * kh, 1, 1}};
    hipLaunchKernelGGL(zero_regions, dim3(128), dim3(256), 0, st, z);
    v.hb = work_v; v.seam_a = work_v + 4 * kh; v.seam_b = work_v + 6 * kh; v.flags = words;
    r.hb = work_r; r.seam_a = work_r + 2 * kh; r.seam_b = work_r + 3 * kh; r.flags = words + ny * 32;
    *abort_out = words + 2 * ny * 32;
    const int per_chain = ny < 8 ? ny * 32 : (ny + 7) / 8 * 8 * 32;
    hipLaunchKernelGGL(seg_fwd_pair_hp, dim3(2 * per_chain), dim3(256), WARM_LDS, st, v, r, *abort_out, per_chain, ny);
    return ICRL_OK;
}

// one forward chain alone in the persistent launch (gates = 4: LSTM, 3: GRU); same workspace layout as
// the per-step schedules above
int solo_persist(ChainArgs& a, int gates, float* work, hipStream_t st, unsigned** abort_out) {
    const int K = a.g.K;
    const int64_t kh = (int64_t)K * H;
    const int ny = (K + 31) / 32;
    unsigned* words = reinterpret_cast<unsigned*>(work + 12 * kh);
    const size_t word_bytes = ((size_t)(ny * 16 + 4) * sizeof(unsigned) + 255) / 256 * 256;
    ICRL_REQUIRE((int64_t)(word_bytes / sizeof(float)) <= 4 * kh, "persistent chain launch: workspace too small");
    ICRL_HIP(hipMemsetAsync(words, 0, word_bytes, st));
    ICRL_HIP(hipMemsetAsync(work, 0, sizeof(float) * 2 * kh, st));
    a.hb = work; a.flags = words;
    a.seam_a = work + (gates == 4 ? 4 : 2) * kh;
    a.seam_b = work + (gates == 4 ? 6 : 3) * kh;
    *abort_out = words + ny * 16;
    const int per_chain = (H / 32) * ny;
    hipLaunchKernelGGL(seg_fwd_pair_persist, dim3(per_chain), dim3(256), 0, st, a, a, *abort_out, per_chain,
                       gates == 4 ? 1 : 2);
    return ICRL_OK;
}

// cheap: the first steps of the warm-up that run on seg_warm_bwd (16-bit split products) before the float32 launch
int bwd_persist(BwdArgs& a, float* work, hipStream_t st, unsigned** abort_out, int cheap = 0, float* seam_err = nullptr,
                float* dh_out = nullptr) {
    const int K = a.g.K;
    if (a.cc == nullptr) {   // the chains' own records {i, f, g, o, c, h}
        a.cc = a.saved + 4 * H;
        a.gate_stride = a.c_stride = 6 * H;
    }
    const int64_t kh = (int64_t)K * H;
    const int ny = (K + 31) / 32;
    unsigned* words = reinterpret_cast<unsigned*>(work + 12 * kh);
    // a 128-byte line of flags per row group: a line is then written from one XCD only when the group is XCD-local;
    // then the abort word, then the flags of the 16-bit launch
    const size_t word_bytes = ((size_t)(2 * ny * 32 + 4) * sizeof(unsigned) + 255) / 256 * 256;
    ICRL_REQUIRE((int64_t)(word_bytes / sizeof(float)) <= 4 * kh, "persistent chain launch: workspace too small");
    {
        ZeroArgs z{{reinterpret_cast<float*>(words), seam_err, nullptr, nullptr, nullptr},
                   {(int64_t)(word_bytes / sizeof(float)), seam_err ? 2 : 0, 0, 0, 0}};
        hipLaunchKernelGGL(zero_regions, dim3(8), dim3(256), 0, st, z);
    }
    a.dgx = work; a.dc = work + 8 * kh; a.seam_a = work + 9 * kh; a.dh_fin = dh_out ? dh_out : work + 11 * kh; a.flags = words;
    *abort_out = words + ny * 32;
    if (cheap > 0) {
        WarmBwdArgs wa{};
        wa.w_hh = a.w_hh; wa.saved = a.saved; wa.d_emit = a.d_emit; wa.dgx = a.dgx; wa.dc = a.dc;
        wa.flags = words + ny * 32 + 4; wa.g = a.g; wa.steps = cheap;
        hipLaunchKernelGGL(seg_warm_bwd, dim3((ny + 7) / 8 * 8 * 32), dim3(256), 0, st, wa, *abort_out, ny);
        a.m_begin = cheap;
    }
    hipLaunchKernelGGL(seg_bwd_persist, dim3(16 * ny), dim3(256), 0, st, a, *abort_out);
    return ICRL_OK;
}


// floats of the owner-computes backward launch's workspace: the exchanged partial tiles (two parities), then dc,
// seam_a, dh_fin per segment, then the polled words
int64_t bwd_rs_floats(int K) {
    const int64_t groups = (K + 31) / 32;
    return groups * (2 * 16 * 16 * 1024) + groups * 32 * (4 * H) + 1024;
}

int bwd_rs(RsArgs& a, float* work, hipStream_t st, unsigned** abort_out, float* seam_err, float* dh_out = nullptr) {
    const int K = a.g.K;
    const int64_t groups = (K + 31) / 32, rows = groups * 32;
    if (a.cc == nullptr) {   // the chains' own records {i, f, g, o, c, h}
        a.cc = a.saved + 4 * H;
        a.gate_stride = a.c_stride = 6 * H;
    }
    a.part = work;
    float* tail = work + groups * (2 * 16 * 16 * 1024);
    a.dc = tail; a.seam_a = tail + rows * H; a.dh_fin = dh_out ? dh_out : tail + 3 * rows * H;
    unsigned* words = reinterpret_cast<unsigned*>(tail + 4 * rows * H);
    a.flags = words;
    *abort_out = words + groups * 16;
    ICRL_REQUIRE(groups * 16 + 4 <= 1024, "owner-computes backward launch: too many row groups");
    ZeroArgs z{{reinterpret_cast<float*>(words), seam_err, nullptr, nullptr, nullptr}, {1024, seam_err ? 2 : 0, 0, 0, 0}};
    hipLaunchKernelGGL(zero_regions, dim3(8), dim3(256), 0, st, z);
    hipLaunchKernelGGL(seg_bwd_rs, dim3(16 * (unsigned)groups), dim3(512), 0, st, a, *abort_out);
    return ICRL_OK;
}
}  // namespace

extern "C" int64_t icrl_chain_seg_workspace_floats(int segments) { return (int64_t)segments * 16 * H; }

namespace {
// icrl_lstm_chain_fwd_seg on the workspace `work` (the caller's, or this library's static scratch array)
int lstm_fwd_seg_on(const char* who, const float* xproj, const float* w_hh, int N, int p0, int n_calls, int full_calls,
                    int segments,
                    int warmup, float* h_emit, float* saved, const float* state_in, float* work, float* seam_err,
                    hipStream_t st) {
    Geom g;
    if (int rc = make_geom(who, N, p0, n_calls, full_calls, segments, warmup, g)) return rc;
    ICRL_REQUIRE(xproj && w_hh && h_emit && work && seam_err, "%s: null pointer", who);
    ICRL_REQUIRE(aligned16(xproj) && aligned16(w_hh) && aligned16(h_emit) && aligned16(saved) && aligned16(work) &&
                 aligned16(state_in), "%s: buffers must be 16-byte aligned", who);
    const int64_t kh = (int64_t)g.K * H;
    ICRL_HIP(hipMemsetAsync(seam_err, 0, sizeof(float), st));
    if (int rc = fwd_steps_lstm(g, xproj, w_hh, h_emit, saved, work, state_in, st)) return rc;
    hipLaunchKernelGGL(seam_check_fwd, dim3(seam_blocks(g.K * 2 * H)), dim3(256), 0, st, work + 4 * kh, work + 6 * kh,
                       g, 2 * H, seam_err);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

// Scratch of the entry point that allocates nothing: the segment workspace (state double buffers, seam
// records) for up to STATIC_MAX_SEGMENTS segments, 16 * 512 floats each = 4 MiB, resident on every GPU the
// library runs on until the process ends.  In host code the array's name is only a handle for
// hipGetSymbolAddress; every use of its memory goes through the per-device address kept below.
constexpr int STATIC_MAX_SEGMENTS = 128;
constexpr int STATIC_MAX_DEVICES = 64;
static __device__ __align__(256) float scratch[(size_t)STATIC_MAX_SEGMENTS * 16 * H];
// zero-initialised; set on the first call on a device, that device current.  Two host threads that make their
// first calls together both look the same address up and both store it: atomic, so neither reads a torn pointer
std::atomic<float*> scratch_of_device[STATIC_MAX_DEVICES];
}  // namespace

extern "C" int icrl_lstm_chain_fwd_seg(const float* xproj, const float* w_hh, int N, int p0, int n_calls, int full_calls,
                                       int segments, int warmup, float* h_emit, float* saved,
                                       const float* state_in, float* work, float* seam_err, icrl_stream_t stream) {
    return lstm_fwd_seg_on("icrl_lstm_chain_fwd_seg", xproj, w_hh, N, p0, n_calls, full_calls, segments, warmup, h_emit,
                           saved, state_in, work, seam_err, icrl::as_stream(stream));
}

extern "C" int icrl_lstm_chain_fwd_seg_static(const float* xproj, const float* w_hh, int N, int p0, int n_calls, int full_calls,
                                              int segments, int warmup, float* h_emit, float* saved,
                                              const float* state_in, float* seam_err, icrl_stream_t stream) {
    ICRL_REQUIRE(segments <= STATIC_MAX_SEGMENTS,
                 "icrl_lstm_chain_fwd_seg_static: %d segments exceed the static workspace (at most %d); use "
                 "icrl_lstm_chain_fwd_seg with a workspace of your own", segments, STATIC_MAX_SEGMENTS);
    int dev = -1;
    ICRL_HIP(hipGetDevice(&dev));
    ICRL_REQUIRE(dev >= 0 && dev < STATIC_MAX_DEVICES, "icrl_lstm_chain_fwd_seg_static: device %d not supported", dev);
    // every GPU holds its own copy of the array: the one used is the current device's, so the stream must
    // belong to it (a stream of another GPU would be handed an address of the wrong GPU's memory)
    hipStream_t st = icrl::as_stream(stream);
    if (st != nullptr) {
        hipDevice_t sdev = -1;
        ICRL_HIP(hipStreamGetDevice(st, &sdev));
        ICRL_REQUIRE((int)sdev == dev, "icrl_lstm_chain_fwd_seg_static: the stream belongs to device %d, the current "
                     "device is %d; make the stream's device current before the call", (int)sdev, dev);
    }
    float* ptr = scratch_of_device[dev].load(std::memory_order_acquire);
    if (ptr == nullptr) {   // first call on this device (it is the current one)
        ICRL_HIP(hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(scratch)));
        ICRL_REQUIRE(ptr != nullptr, "icrl_lstm_chain_fwd_seg_static: no address for the static workspace");
        scratch_of_device[dev].store(ptr, std::memory_order_release);
    }
    return lstm_fwd_seg_on("icrl_lstm_chain_fwd_seg_static", xproj, w_hh, N, p0, n_calls, full_calls, segments, warmup,
                           h_emit, saved, state_in, ptr, seam_err, st);
}

extern "C" int icrl_gru_chain_fwd_seg(const float* xproj, const float* w_hh, const float* b_hh, int N, int p0,
                                      int n_calls, int full_calls, int segments, int warmup, float* h_emit, float* saved,
                                      float* work, float* seam_err, icrl_stream_t stream) {
    Geom g;
    if (int rc = make_geom("icrl_gru_chain_fwd_seg", N, p0, n_calls, full_calls, segments, warmup, g)) return rc;
    ICRL_REQUIRE(xproj && w_hh && b_hh && h_emit && work && seam_err, "icrl_gru_chain_fwd_seg: null pointer");
    ICRL_REQUIRE(aligned16(xproj) && aligned16(w_hh) && aligned16(b_hh) && aligned16(h_emit) && aligned16(work),
                 "icrl_gru_chain_fwd_seg: buffers must be 16-byte aligned");
    hipStream_t st = icrl::as_stream(stream);
    const int64_t kh = (int64_t)g.K * H;
    ICRL_HIP(hipMemsetAsync(seam_err, 0, sizeof(float), st));
    ICRL_REQUIRE(aligned16(saved), "icrl_gru_chain_fwd_seg: buffers must be 16-byte aligned");
    if (int rc = fwd_steps_gru(g, xproj, w_hh, b_hh, h_emit, saved, work, st)) return rc;
    hipLaunchKernelGGL(seam_check_fwd, dim3(seam_blocks(g.K * H)), dim3(256), 0, st, work + 2 * kh, work + 3 * kh, g, H,
                       seam_err);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_lstm_chain_bwd_seg(const float* w_hh, const float* saved, const float* d_emit, int N, int p0,
                                       int n_calls, int full_calls, int segments, int warmup, float* dgates, float* work,
                                       float* seam_err, icrl_stream_t stream) {
    Geom g;
    if (int rc = make_geom("icrl_lstm_chain_bwd_seg", N, p0, n_calls, full_calls, segments, warmup, g)) return rc;
    ICRL_REQUIRE(w_hh && saved && d_emit && dgates && work && seam_err, "icrl_lstm_chain_bwd_seg: null pointer");
    ICRL_REQUIRE(aligned16(w_hh) && aligned16(saved) && aligned16(d_emit) && aligned16(dgates) && aligned16(work),
                 "icrl_lstm_chain_bwd_seg: buffers must be 16-byte aligned");
    hipStream_t st = icrl::as_stream(stream);
    const int64_t kh = (int64_t)g.K * H;
    ICRL_HIP(hipMemsetAsync(seam_err, 0, 2 * sizeof(float), st));
    if (int rc = bwd_steps(g, w_hh, saved, d_emit, dgates, work, st)) return rc;
    hipLaunchKernelGGL(seam_check_bwd, dim3(seam_blocks(g.K * 2 * H)), dim3(256), 0, st, work + 9 * kh, work + 11 * kh,
                       work + 8 * kh, g, seam_err, seam_err + 1);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_gru_chain_bwd_seg(const float* w_hh, const float* saved, const float* d_emit, int N, int p0,
                                      int n_calls, int segments, int warmup, float* dgates_x, float* dgates_h,
                                      float* work, float* seam_err, icrl_stream_t stream) {
    Geom g;
    if (int rc = make_geom("icrl_gru_chain_bwd_seg", N, p0, n_calls, 0, segments, warmup, g)) return rc;
    ICRL_REQUIRE(w_hh && saved && d_emit && dgates_x && dgates_h && work && seam_err,
                 "icrl_gru_chain_bwd_seg: null pointer");
    ICRL_REQUIRE(aligned16(w_hh) && aligned16(saved) && aligned16(d_emit) && aligned16(dgates_x) &&
                 aligned16(dgates_h) && aligned16(work), "icrl_gru_chain_bwd_seg: buffers must be 16-byte aligned");
    hipStream_t st = icrl::as_stream(stream);
    const int64_t kh = (int64_t)g.K * H;
    float* dg[2] = {work, work + 4 * kh};  // [K, 3H] each
    float* carry = work + 8 * kh;          // [K, H]
    float* seam_a = work + 9 * kh;         // [K, 2H]
    float* dh_fin = work + 11 * kh;        // [K, H]
    ICRL_HIP(hipMemsetAsync(carry, 0, sizeof(float) * 3 * kh, st));
    ICRL_HIP(hipMemsetAsync(seam_err, 0, 2 * sizeof(float), st));
    const dim3 grid(H / 32, (g.K + 31) / 32);
    const int steps = g.W + g.Ls;
    for (int mp = 0; mp <= steps; ++mp)
        hipLaunchKernelGGL(seg_gru_bwd_step, grid, dim3(256), 0, st, dg[(mp & 1) ^ 1], w_hh, saved, d_emit, carry,
                           dg[mp & 1], dgates_x, dgates_h, seam_a, dh_fin, g, mp, mp == steps ? 1 : 0);
    hipLaunchKernelGGL(seam_check_bwd, dim3(seam_blocks(g.K * 2 * H)), dim3(256), 0, st, seam_a, dh_fin, carry, g,
                       seam_err, seam_err + 1);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_chain_pair_supported(int segments) {
    // every workgroup of a persistent launch must be resident at once: 2 x 16 x ceil(segments / 32) of the forward
    // pair kernel, 16 x ceil(segments / 32) of the backward kernel, against CUs x what the runtime says fits one CU
    static int cus = -1, per_cu_fwd = 0, per_cu_bwd = 0;
    if (cus < 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_fwd, seg_fwd_pair_persist, 256, 0) != hipSuccess) return 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_bwd, seg_bwd_persist, 256, 0) != hipSuccess) return 0;
        cus = n;
    }
    if (segments < 1) return 0;
    const int per_chain = (H / 32) * ((segments + 31) / 32);
    return 2 * per_chain <= cus * per_cu_fwd && per_chain <= cus * per_cu_bwd ? 1 : 0;
}

namespace {
int pair_fwd_impl(const char* who, const float* xproj_v, const float* w_hh_v, const float* xproj_r, const float* w_hh_r,
                  const float* b_hh_r, int N, int p0_v, int p0_r, int n_calls, int full_calls, int segments, int warmup_v,
                  int warmup_r, int cheap_v, int cheap_r, int phase, float* h_emit_v, float* saved_v, float* h_emit_r,
                  float* saved_r, float* work_v, float* work_r, float* seam_err_v, float* seam_err_r, icrl_stream_t stream) {
    ChainArgs v{}, r{};
    if (int rc = make_geom(who, N, p0_v, n_calls, full_calls, segments, warmup_v, v.g)) return rc;
    if (int rc = make_geom(who, N, p0_r, n_calls, full_calls, segments, warmup_r, r.g)) return rc;
    ICRL_REQUIRE(xproj_v && w_hh_v && xproj_r && w_hh_r && b_hh_r && h_emit_v && h_emit_r && work_v && work_r &&
                 seam_err_v && seam_err_r, "%s: null pointer", who);
    ICRL_REQUIRE(aligned16(xproj_v) && aligned16(w_hh_v) && aligned16(xproj_r) && aligned16(w_hh_r) &&
                 aligned16(b_hh_r) && aligned16(h_emit_v) && aligned16(saved_v) && aligned16(h_emit_r) &&
                 aligned16(saved_r) && aligned16(work_v) && aligned16(work_r),
                 "%s: buffers must be 16-byte aligned", who);
    ICRL_REQUIRE(icrl_chain_pair_supported(segments),
                 "%s: %d segments need more workgroups than the device has CUs; its workgroups "
                 "wait for each other and must all be resident", who, segments);
    ICRL_REQUIRE(cheap_v >= 0 && cheap_r >= 0 && cheap_v < warmup_v && cheap_r < warmup_r,
                 "%s: the 16-bit part of a warm-up must be shorter than the warm-up (at least one float32 step follows)", who);
    ICRL_REQUIRE((cheap_v == 0 && cheap_r == 0) || icrl_chain_warm16_supported(segments),
                 "%s: the 16-bit warm-up launch does not fit the device at %d segments", who, segments);
    ICRL_REQUIRE(phase >= 0 && phase <= 2 && (phase == 0 || cheap_v > 0 || cheap_r > 0),
                 "%s: phase must be 0, or 1 / 2 with a 16-bit part to run", who);
    hipStream_t st = icrl::as_stream(stream);
    v.xproj = xproj_v; v.w_hh = w_hh_v; v.b_hh = nullptr; v.saved = saved_v; v.h_emit = h_emit_v; v.state_in = nullptr;
    r.xproj = xproj_r; r.w_hh = w_hh_r; r.b_hh = b_hh_r; r.saved = saved_r; r.h_emit = h_emit_r; r.state_in = nullptr;
    unsigned* abort_w = nullptr;
    if (int rc = pair_persist(v, r, work_v, work_r, st, &abort_w, cheap_v, cheap_r, phase, seam_err_v, seam_err_r, 1)) return rc;
    if (phase == 1) {
        ICRL_LAUNCH_CHECK();
        return ICRL_OK;
    }
    const int nb = seam_blocks(v.g.K * 2 * H);
    hipLaunchKernelGGL(seam_check_fwd_pair, dim3(2 * nb), dim3(256), 0, st, v.seam_a, v.seam_b, v.g, seam_err_v, r.seam_a,
                       r.seam_b, r.g, seam_err_r, abort_w, nb);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}
}  // namespace

extern "C" int icrl_chain_warm16_supported(int segments) {
    // all 2 x 32 x ceil(segments / 32) workgroups of the 16-bit warm-up launch (rounded up to whole groups of 8 row
    // groups) must be resident at once
    static int cus = -1, per_cu = 0;
    if (cus < 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(seg_warm_pair), hipFuncAttributeMaxDynamicSharedMemorySize,
                                WARM_LDS) != hipSuccess) return 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, seg_warm_pair, 256, WARM_LDS) != hipSuccess) return 0;
        cus = n;
    }
    if (segments < 1) return 0;
    const int ny = (segments + 31) / 32;
    return 2 * ((ny + 7) / 8 * 8 * 32) <= cus * per_cu ? 1 : 0;
}

extern "C" int icrl_chain_pair_fwd_seg(const float* xproj_v, const float* w_hh_v, const float* xproj_r,
                                       const float* w_hh_r, const float* b_hh_r, int N, int p0_v, int p0_r,
                                       int n_calls, int full_calls, int segments, int warmup_v, int warmup_r, float* h_emit_v,
                                       float* saved_v, float* h_emit_r, float* saved_r, float* work_v, float* work_r,
                                       float* seam_err_v, float* seam_err_r, icrl_stream_t stream) {
    return pair_fwd_impl("icrl_chain_pair_fwd_seg", xproj_v, w_hh_v, xproj_r, w_hh_r, b_hh_r, N, p0_v, p0_r, n_calls,
                         full_calls, segments, warmup_v, warmup_r, 0, 0, 0, h_emit_v, saved_v, h_emit_r, saved_r, work_v,
                         work_r, seam_err_v, seam_err_r, stream);
}

extern "C" int icrl_chain_pair_fwd_seg_warm16(const float* xproj_v, const float* w_hh_v, const float* xproj_r,
                                              const float* w_hh_r, const float* b_hh_r, int N, int p0_v, int p0_r,
                                              int n_calls, int full_calls, int segments, int warmup_v, int warmup_r,
                                              int cheap_v, int cheap_r, int phase, float* h_emit_v, float* saved_v,
                                              float* h_emit_r, float* saved_r, float* work_v, float* work_r,
                                              float* seam_err_v, float* seam_err_r, icrl_stream_t stream) {
    return pair_fwd_impl("icrl_chain_pair_fwd_seg_warm16", xproj_v, w_hh_v, xproj_r, w_hh_r, b_hh_r, N, p0_v, p0_r,
                         n_calls, full_calls, segments, warmup_v, warmup_r, cheap_v, cheap_r, phase, h_emit_v, saved_v,
                         h_emit_r, saved_r, work_v, work_r, seam_err_v, seam_err_r, stream);
}


extern "C" int icrl_chain_f16x2_supported(int segments) {
    // all 2 x 32 x ceil(segments / 32) workgroups of the f16 forward launch (rounded up to whole groups of 8 row
    // groups) must be resident at once
    static int cus = -1, per_cu = 0;
    if (cus < 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(seg_fwd_pair_hp), hipFuncAttributeMaxDynamicSharedMemorySize,
                                WARM_LDS) != hipSuccess) return 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, seg_fwd_pair_hp, 256, WARM_LDS) != hipSuccess) return 0;
        cus = n;
    }
    if (segments < 1) return 0;
    const int ny = (segments + 31) / 32;
    return 2 * ((ny + 7) / 8 * 8 * 32) <= cus * per_cu ? 1 : 0;
}

extern "C" int icrl_chain_pair_fwd_seg_f16x2(const float* xproj_v, const float* w_hh_v, const float* xproj_r,
                                             const float* w_hh_r, const float* b_hh_r, int N, int p0_v, int p0_r,
                                             int n_calls, int full_calls, int segments, int warmup_v, int warmup_r,
                                             float* h_emit_v, float* saved_v, float* h_emit_r, float* saved_r,
                                             float* work_v, float* work_r, float* seam_err_v, float* seam_err_r,
                                             icrl_stream_t stream) {
    const char* who = "icrl_chain_pair_fwd_seg_f16x2";
    HpArgs v{}, r{};
    if (int rc = make_geom(who, N, p0_v, n_calls, full_calls, segments, warmup_v, v.g)) return rc;
    if (int rc = make_geom(who, N, p0_r, n_calls, full_calls, segments, warmup_r, r.g)) return rc;
    ICRL_REQUIRE(xproj_v && w_hh_v && xproj_r && w_hh_r && b_hh_r && h_emit_v && h_emit_r && work_v && work_r &&
                 seam_err_v && seam_err_r, "%s: null pointer", who);
    ICRL_REQUIRE(aligned16(xproj_v) && aligned16(w_hh_v) && aligned16(xproj_r) && aligned16(w_hh_r) &&
                 aligned16(b_hh_r) && aligned16(h_emit_v) && aligned16(saved_v) && aligned16(h_emit_r) &&
                 aligned16(saved_r) && aligned16(work_v) && aligned16(work_r),
                 "%s: buffers must be 16-byte aligned", who);
    ICRL_REQUIRE(icrl_chain_f16x2_supported(segments),
                 "%s: %d segments need more workgroups than the device can hold at once; its workgroups wait for each "
                 "other and must all be resident", who, segments);
    hipStream_t st = icrl::as_stream(stream);
    v.xproj = xproj_v; v.w_hh = w_hh_v; v.b_hh = nullptr; v.saved = saved_v; v.h_emit = h_emit_v;
    r.xproj = xproj_r; r.w_hh = w_hh_r; r.b_hh = b_hh_r; r.saved = saved_r; r.h_emit = h_emit_r;
    unsigned* abort_w = nullptr;
    if (int rc = pair_hp(v, r, work_v, work_r, st, &abort_w, seam_err_v, seam_err_r)) return rc;
    const int nb = seam_blocks(v.g.K * 2 * H);
    hipLaunchKernelGGL(seam_check_fwd_pair, dim3(2 * nb), dim3(256), 0, st, v.seam_a, v.seam_b, v.g, seam_err_v, r.seam_a,
                       r.seam_b, r.g, seam_err_r, abort_w, nb);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_lstm_chain_fwd_seg_persist(const float* xproj, const float* w_hh, int N, int p0, int n_calls, int full_calls,
                                               int segments, int warmup, float* h_emit, float* saved,
                                               const float* state_in, float* work, float* seam_err,
                                               icrl_stream_t stream) {
    ChainArgs a{};
    if (int rc = make_geom("icrl_lstm_chain_fwd_seg_persist", N, p0, n_calls, full_calls, segments, warmup, a.g)) return rc;
    ICRL_REQUIRE(xproj && w_hh && h_emit && work && seam_err, "icrl_lstm_chain_fwd_seg_persist: null pointer");
    ICRL_REQUIRE(aligned16(xproj) && aligned16(w_hh) && aligned16(h_emit) && aligned16(saved) && aligned16(work) &&
                 aligned16(state_in), "icrl_lstm_chain_fwd_seg_persist: buffers must be 16-byte aligned");
    ICRL_REQUIRE(warmup >= 1 || !state_in, "icrl_lstm_chain_fwd_seg_persist: a start state needs a warm-up of at least 1");
    ICRL_REQUIRE(icrl_chain_pair_supported(segments),
                 "icrl_lstm_chain_fwd_seg_persist: %d segments need more workgroups than the device has CUs; its "
                 "workgroups wait for each other and must all be resident", segments);
    hipStream_t st = icrl::as_stream(stream);
    ICRL_HIP(hipMemsetAsync(seam_err, 0, sizeof(float), st));
    a.xproj = xproj; a.w_hh = w_hh; a.b_hh = nullptr; a.saved = saved; a.h_emit = h_emit; a.state_in = state_in;
    unsigned* abort_w = nullptr;
    if (int rc = solo_persist(a, 4, work, st, &abort_w)) return rc;
    hipLaunchKernelGGL(seam_check_fwd, dim3(seam_blocks(a.g.K * 2 * H)), dim3(256), 0, st, a.seam_a, a.seam_b, a.g,
                       2 * H, seam_err);
    hipLaunchKernelGGL(poison_on_abort, dim3(1), dim3(1), 0, st, abort_w, seam_err, seam_err, 3);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_gru_chain_fwd_seg_persist(const float* xproj, const float* w_hh, const float* b_hh, int N, int p0,
                                              int n_calls, int full_calls, int segments, int warmup, float* h_emit, float* saved,
                                              float* work, float* seam_err, icrl_stream_t stream) {
    ChainArgs a{};
    if (int rc = make_geom("icrl_gru_chain_fwd_seg_persist", N, p0, n_calls, full_calls, segments, warmup, a.g)) return rc;
    ICRL_REQUIRE(xproj && w_hh && b_hh && h_emit && work && seam_err, "icrl_gru_chain_fwd_seg_persist: null pointer");
    ICRL_REQUIRE(aligned16(xproj) && aligned16(w_hh) && aligned16(b_hh) && aligned16(h_emit) && aligned16(saved) &&
                 aligned16(work), "icrl_gru_chain_fwd_seg_persist: buffers must be 16-byte aligned");
    ICRL_REQUIRE(icrl_chain_pair_supported(segments),
                 "icrl_gru_chain_fwd_seg_persist: %d segments need more workgroups than the device has CUs; its "
                 "workgroups wait for each other and must all be resident", segments);
    hipStream_t st = icrl::as_stream(stream);
    ICRL_HIP(hipMemsetAsync(seam_err, 0, sizeof(float), st));
    a.xproj = xproj; a.w_hh = w_hh; a.b_hh = b_hh; a.saved = saved; a.h_emit = h_emit; a.state_in = nullptr;
    unsigned* abort_w = nullptr;
    if (int rc = solo_persist(a, 3, work, st, &abort_w)) return rc;
    hipLaunchKernelGGL(seam_check_fwd, dim3(seam_blocks(a.g.K * H)), dim3(256), 0, st, a.seam_a, a.seam_b, a.g, H, seam_err);
    hipLaunchKernelGGL(poison_on_abort, dim3(1), dim3(1), 0, st, abort_w, seam_err, seam_err, 3);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

namespace {
int lstm_bwd_persist_impl(const char* who, const float* w_hh, const float* saved, const float* d_emit, int N, int p0,
                          int n_calls, int full_calls, int segments, int warmup, int cheap, float* dgates, float* work,
                          float* seam_err, icrl_stream_t stream) {
    BwdArgs a{};
    if (int rc = make_geom(who, N, p0, n_calls, full_calls, segments, warmup, a.g)) return rc;
    ICRL_REQUIRE(w_hh && saved && d_emit && dgates && work && seam_err, "%s: null pointer", who);
    ICRL_REQUIRE(aligned16(w_hh) && aligned16(saved) && aligned16(d_emit) && aligned16(dgates) && aligned16(work),
                 "%s: buffers must be 16-byte aligned", who);
    ICRL_REQUIRE(icrl_chain_pair_supported(segments),
                 "%s: %d segments need more workgroups than the device has CUs; its "
                 "workgroups wait for each other and must all be resident", who, segments);
    ICRL_REQUIRE(cheap >= 0 && cheap <= warmup, "%s: the 16-bit part of a warm-up cannot be longer than the warm-up", who);
    ICRL_REQUIRE(cheap == 0 || icrl_chain_bwd_warm16_supported(segments),
                 "%s: the 16-bit warm-up launch needs a multiple of 32 segments that fits the device (%d given)", who,
                 segments);
    hipStream_t st = icrl::as_stream(stream);
    a.w_hh = w_hh; a.saved = saved; a.d_emit = d_emit; a.dgv = dgates;
    unsigned* abort_w = nullptr;
    if (int rc = bwd_persist(a, work, st, &abort_w, cheap, seam_err)) return rc;
    hipLaunchKernelGGL(seam_check_bwd, dim3(seam_blocks(a.g.K * 2 * H)), dim3(256), 0, st, a.seam_a, a.dh_fin, a.dc,
                       a.g, seam_err, seam_err + 1, (const unsigned*)abort_w);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}
}  // namespace

extern "C" int icrl_chain_bwd_warm16_supported(int segments) {
    // all 32 x ceil(segments / 32) workgroups of the 16-bit backward warm-up launch (rounded up to whole groups of
    // 8 row groups) must be resident at once; its exchange layout needs whole row groups
    static int cus = -1, per_cu = 0;
    if (cus < 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, seg_warm_bwd, 256, 0) != hipSuccess) return 0;
        cus = n;
    }
    if (segments < 32 || segments % 32 != 0) return 0;
    const int ny = segments / 32;
    return (ny + 7) / 8 * 8 * 32 <= cus * per_cu ? 1 : 0;
}

extern "C" int icrl_lstm_chain_bwd_seg_persist(const float* w_hh, const float* saved, const float* d_emit, int N,
                                               int p0, int n_calls, int full_calls, int segments, int warmup, float* dgates,
                                               float* work, float* seam_err, icrl_stream_t stream) {
    return lstm_bwd_persist_impl("icrl_lstm_chain_bwd_seg_persist", w_hh, saved, d_emit, N, p0, n_calls, full_calls,
                                 segments, warmup, 0, dgates, work, seam_err, stream);
}

extern "C" int icrl_lstm_chain_bwd_seg_warm16(const float* w_hh, const float* saved, const float* d_emit, int N,
                                              int p0, int n_calls, int full_calls, int segments, int warmup, int cheap,
                                              float* dgates, float* work, float* seam_err, icrl_stream_t stream) {
    return lstm_bwd_persist_impl("icrl_lstm_chain_bwd_seg_warm16", w_hh, saved, d_emit, N, p0, n_calls, full_calls,
                                 segments, warmup, cheap, dgates, work, seam_err, stream);
}

// ------------------------------------------------------------------ may the replays be left out?
// A chain evaluated with full_calls < n_calls leaves out, for every call s >= full_calls, the replay of the
// inputs of call s-1 that the reference runs before the call's new column.  That is right exactly when such
// a replay, started from the state call s-1 ended in, comes back to that same state.  This entry point runs
// the replays' first `steps` positions for all those calls side by side (one row per call, each from its own
// start state, all over the same inputs: the persistent pair launch with stride 0) and reports how far each
// row's state then is from the state the last fully walked call had at that position: once the two
// trajectories meet they stay together, since from there on they see the same inputs.
__global__ __launch_bounds__(256) void replay_states(const float* __restrict__ saved, int slots, int h_slot, int c_slot,
                                                     int64_t first_pos, int64_t pos_step, int rows,
                                                     float* __restrict__ out) {
    const int sw = c_slot >= 0 ? 2 * H : H;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < rows * sw; e += gridDim.x * 256) {
        const int j = e / sw, k = e % sw;
        const int64_t pos = first_pos + (int64_t)j * pos_step;
        out[e] = saved[(pos * slots + (k < H ? h_slot : c_slot)) * H + (k & (H - 1))];
    }
}

__global__ __launch_bounds__(256) void replay_compare(const float* __restrict__ fin, const float* __restrict__ saved,
                                                      int slots, int h_slot, int c_slot, int64_t target_pos, int rows,
                                                      float* __restrict__ err) {
    __shared__ float red[4];
    const int sw = c_slot >= 0 ? 2 * H : H;
    float d = 0.f;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < rows * sw; e += gridDim.x * 256) {
        const int k = e % sw;
        const float b = saved[(target_pos * slots + (k < H ? h_slot : c_slot)) * H + (k & (H - 1))];
        float v = fabsf(fin[e] - b) / fmaxf(1.f, fabsf(b));
        if (!(v == v)) v = INFINITY;
        d = fmaxf(d, v);
    }
    d = wave_max(d);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
    __syncthreads();
    if (threadIdx.x == 0) {
        d = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (d > 0.f) atomic_max_nonneg(err, d);
    }
}

namespace {
// geometry of the replay rows: `rows` chains that all walk positions 0 .. steps-1 (one warm-up step that only
// loads the row's start state)
void replay_geom(Geom& g, int N, int rows, int steps) {
    const int cols = (steps + N - 1) / N + 1;   // one column more than the rows reach: nothing is ever emitted
    g.n = N; g.p0 = cols; g.n_calls = 1; g.fc = 1; g.c0 = cols;
    g.total = steps; g.K = rows; g.Ls = steps; g.W = 1; g.stride = 0;
}
}  // namespace

namespace {
int replay_check(const char* who, bool f16x2, const float* xproj_v, const float* w_hh_v, const float* saved_v,
                 const float* xproj_r, const float* w_hh_r, const float* b_hh_r, const float* saved_r, int N, int p0_v,
                 int p0_r, int n_calls, int full_calls, int steps, float* work_v, float* work_r, float* err,
                 icrl_stream_t stream) {
    ICRL_REQUIRE(N > 0 && p0_v > 0 && p0_r > 0 && n_calls > 1, "%s: bad shape", who);
    ICRL_REQUIRE(full_calls >= 1 && full_calls < n_calls,
                 "%s: full_calls must be in [1, n_calls): nothing is left out otherwise", who);
    const int rows = n_calls - full_calls;
    const int64_t len_v = (int64_t)(p0_v + full_calls - 1) * N, len_r = (int64_t)(p0_r + full_calls - 1) * N;
    ICRL_REQUIRE(steps >= 1 && steps <= len_v && steps <= len_r,
                 "%s: steps must be in [1, %lld], the length of the shortest replay left out", who,
                 (long long)std::min(len_v, len_r));
    ICRL_REQUIRE(xproj_v && w_hh_v && saved_v && xproj_r && w_hh_r && b_hh_r && saved_r && work_v && work_r && err,
                 "%s: null pointer", who);
    ICRL_REQUIRE(aligned16(xproj_v) && aligned16(w_hh_v) && aligned16(saved_v) && aligned16(xproj_r) &&
                 aligned16(w_hh_r) && aligned16(b_hh_r) && aligned16(saved_r) && aligned16(work_v) && aligned16(work_r),
                 "%s: buffers must be 16-byte aligned", who);
    ICRL_REQUIRE(icrl_chain_pair_supported(rows), "%s: too many rows for this device", who);
    hipStream_t st = icrl::as_stream(stream);
    ChainArgs v{}, r{};
    replay_geom(v.g, N, rows, steps);
    replay_geom(r.g, N, rows, steps);
    const int64_t kh = (int64_t)rows * H;
    const int c0_v = full_calls * p0_v + full_calls * (full_calls - 1) / 2;
    const int c0_r = full_calls * p0_r + full_calls * (full_calls - 1) / 2;
    float* states_v = work_v + 8 * kh;   // [rows][2H]
    float* states_r = work_r + 8 * kh;   // [rows][H]
    ICRL_HIP(hipMemsetAsync(err, 0, 2 * sizeof(float), st));
    // row j stands for call full_calls + j: it starts from the state at the end of the call before it
    hipLaunchKernelGGL(replay_states, dim3(seam_blocks(rows * 2 * H)), dim3(256), 0, st, saved_v, 6, 5, 4,
                       (int64_t)c0_v * N - 1, (int64_t)N, rows, states_v);
    hipLaunchKernelGGL(replay_states, dim3(seam_blocks(rows * H)), dim3(256), 0, st, saved_r, 5, 4, -1,
                       (int64_t)c0_r * N - 1, (int64_t)N, rows, states_r);
    v.xproj = xproj_v; v.w_hh = w_hh_v; v.state_in = states_v; v.state_stride = 2 * H;
    r.xproj = xproj_r; r.w_hh = w_hh_r; r.b_hh = b_hh_r; r.state_in = states_r; r.state_stride = H;
    unsigned* abort_w = nullptr;
    if (f16x2) {   // the same replays in the f16 launch (what the chains themselves were walked with)
        ICRL_REQUIRE(icrl_chain_f16x2_supported(rows), "%s: too many rows for this device", who);
        HpArgs hv{}, hr{};
        hv.xproj = xproj_v; hv.w_hh = w_hh_v; hv.b_hh = nullptr; hv.g = v.g; hv.state_in = states_v; hv.state_stride = 2 * H;
        hr.xproj = xproj_r; hr.w_hh = w_hh_r; hr.b_hh = b_hh_r; hr.g = r.g; hr.state_in = states_r; hr.state_stride = H;
        if (int rc = pair_hp(hv, hr, work_v, work_r, st, &abort_w, nullptr, nullptr)) return rc;
        v.seam_b = hv.seam_b; r.seam_b = hr.seam_b;
    } else if (int rc = pair_persist(v, r, work_v, work_r, st, &abort_w)) return rc;
    // the state the last fully walked call had after the same `steps` inputs
    const int64_t target_v = ((int64_t)c0_v - (p0_v + full_calls - 1)) * N + steps - 1;
    const int64_t target_r = ((int64_t)c0_r - (p0_r + full_calls - 1)) * N + steps - 1;
    hipLaunchKernelGGL(replay_compare, dim3(seam_blocks(rows * 2 * H)), dim3(256), 0, st, v.seam_b, saved_v, 6, 5, 4,
                       target_v, rows, err);
    hipLaunchKernelGGL(replay_compare, dim3(seam_blocks(rows * H)), dim3(256), 0, st, r.seam_b, saved_r, 5, 4, -1,
                       target_r, rows, err + 1);
    hipLaunchKernelGGL(poison_on_abort, dim3(1), dim3(1), 0, st, abort_w, err, err + 1, 2);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}
}  // namespace

extern "C" int icrl_chain_pair_replay_check(const float* xproj_v, const float* w_hh_v, const float* saved_v,
                                            const float* xproj_r, const float* w_hh_r, const float* b_hh_r,
                                            const float* saved_r, int N, int p0_v, int p0_r, int n_calls,
                                            int full_calls, int steps, float* work_v, float* work_r, float* err,
                                            icrl_stream_t stream) {
    return replay_check("icrl_chain_pair_replay_check", false, xproj_v, w_hh_v, saved_v, xproj_r, w_hh_r, b_hh_r, saved_r, N,
                        p0_v, p0_r, n_calls, full_calls, steps, work_v, work_r, err, stream);
}

extern "C" int icrl_chain_pair_replay_check_f16x2(const float* xproj_v, const float* w_hh_v, const float* saved_v,
                                                  const float* xproj_r, const float* w_hh_r, const float* b_hh_r,
                                                  const float* saved_r, int N, int p0_v, int p0_r, int n_calls,
                                                  int full_calls, int steps, float* work_v, float* work_r, float* err,
                                                  icrl_stream_t stream) {
    return replay_check("icrl_chain_pair_replay_check_f16x2", true, xproj_v, w_hh_v, saved_v, xproj_r, w_hh_r, b_hh_r, saved_r,
                        N, p0_v, p0_r, n_calls, full_calls, steps, work_v, work_r, err, stream);
}

// ------------------------------------------------------------------ every row its own chain
// The recurrences as a per-image model would run them (the reference's own classes at batch size 1): row i
// walks its own columns, call after call, from a zero state, independently of the other rows.  The N
// chains advance together through the same kernels as the segments above (a "segment" is a row, no
// warm-up, no seams), with the same per-position layouts, so everything downstream is shared.
extern "C" int icrl_chain_pair_fwd_rows(const float* xproj_v, const float* w_hh_v, const float* xproj_r,
                                        const float* w_hh_r, const float* b_hh_r, int N, int p0_v, int p0_r,
                                        int n_calls, int persistent, float* h_emit_v, float* saved_v,
                                        float* h_emit_r, float* work_v, float* work_r, int* status,
                                        icrl_stream_t stream) {
    ChainArgs v{}, r{};
    if (int rc = make_geom_rows("icrl_chain_pair_fwd_rows", N, p0_v, n_calls, v.g)) return rc;
    if (int rc = make_geom_rows("icrl_chain_pair_fwd_rows", N, p0_r, n_calls, r.g)) return rc;
    ICRL_REQUIRE(xproj_v && w_hh_v && xproj_r && w_hh_r && b_hh_r && h_emit_v && h_emit_r && work_v && work_r && status,
                 "icrl_chain_pair_fwd_rows: null pointer");
    ICRL_REQUIRE(aligned16(xproj_v) && aligned16(w_hh_v) && aligned16(xproj_r) && aligned16(w_hh_r) &&
                 aligned16(b_hh_r) && aligned16(h_emit_v) && aligned16(saved_v) && aligned16(h_emit_r) &&
                 aligned16(work_v) && aligned16(work_r), "icrl_chain_pair_fwd_rows: buffers must be 16-byte aligned");
    hipStream_t st = icrl::as_stream(stream);
    ICRL_HIP(hipMemsetAsync(status, 0, sizeof(int), st));
    if (persistent) {
        ICRL_REQUIRE(icrl_chain_pair_supported(N),
                     "icrl_chain_pair_fwd_rows: %d rows need more workgroups than the device has CUs for a persistent "
                     "launch; pass persistent = 0", N);
        v.xproj = xproj_v; v.w_hh = w_hh_v; v.b_hh = nullptr; v.saved = saved_v; v.h_emit = h_emit_v;
        r.xproj = xproj_r; r.w_hh = w_hh_r; r.b_hh = b_hh_r; r.saved = nullptr; r.h_emit = h_emit_r;
        unsigned* abort_w = nullptr;
        if (int rc = pair_persist(v, r, work_v, work_r, st, &abort_w)) return rc;
        hipLaunchKernelGGL(flag_on_abort, dim3(1), dim3(1), 0, st, abort_w, status, 4);
    } else {
        if (int rc = fwd_steps_lstm(v.g, xproj_v, w_hh_v, h_emit_v, saved_v, work_v, nullptr, st)) return rc;
        if (int rc = fwd_steps_gru(r.g, xproj_r, w_hh_r, b_hh_r, h_emit_r, nullptr, work_r, st)) return rc;
    }
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_lstm_chain_bwd_rows(const float* w_hh, const float* saved, const float* d_emit, int N, int p0,
                                        int n_calls, int persistent, float* dgates, float* work, int* status,
                                        icrl_stream_t stream) {
    BwdArgs a{};
    if (int rc = make_geom_rows("icrl_lstm_chain_bwd_rows", N, p0, n_calls, a.g)) return rc;
    ICRL_REQUIRE(w_hh && saved && d_emit && dgates && work && status, "icrl_lstm_chain_bwd_rows: null pointer");
    ICRL_REQUIRE(aligned16(w_hh) && aligned16(saved) && aligned16(d_emit) && aligned16(dgates) && aligned16(work),
                 "icrl_lstm_chain_bwd_rows: buffers must be 16-byte aligned");
    hipStream_t st = icrl::as_stream(stream);
    ICRL_HIP(hipMemsetAsync(status, 0, sizeof(int), st));
    if (persistent) {
        ICRL_REQUIRE(icrl_chain_pair_supported(N),
                     "icrl_lstm_chain_bwd_rows: %d rows need more workgroups than the device has CUs for a persistent "
                     "launch; pass persistent = 0", N);
        a.w_hh = w_hh; a.saved = saved; a.d_emit = d_emit; a.dgv = dgates;
        unsigned* abort_w = nullptr;
        if (int rc = bwd_persist(a, work, st, &abort_w)) return rc;
        hipLaunchKernelGGL(flag_on_abort, dim3(1), dim3(1), 0, st, abort_w, status, 4);
    } else {
        if (int rc = bwd_steps(a.g, w_hh, saved, d_emit, dgates, work, st)) return rc;
    }
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

// ------------------------------------------------------------------ the policy LSTM's BPTT as one launch
// The backward-through-time of the policy network's LSTM over a rollout (autograd of models.py:80): N independent
// rows, T steps each, a gradient injected at every step (what flows back from that step's vocabulary projection).
// That is the per-row chain geometry with every column the last column of a call of its own (full_calls = 1), so
// the persistent backward kernel walks it as it stands: T + 1 steps inside one launch instead of 3 T dependent
// launches.  gates_act [T, N, 4H] and c_all [T + 1, N, H] are the rollout's activations (c_all[0] = the initial
// cell state, zeros), d_h [T, N, H] the injected gradients; dgates [T, N, 4H] receives the gate gradients and
// dh0 [N, H] what flows into the initial hidden state.  work: icrl_chain_seg_workspace_floats(N) floats.
extern "C" int icrl_policy_lstm_bptt_persist(const float* gates_act, const float* c_all, const float* d_h,
                                             const float* w_hh, int N, int T, float* dgates, float* dh0, float* work,
                                             int* status, icrl_stream_t stream) {
    ICRL_REQUIRE(gates_act && c_all && d_h && w_hh && dgates && dh0 && work && status,
                 "icrl_policy_lstm_bptt_persist: null pointer");
    ICRL_REQUIRE(N > 0 && N <= 4096 && T > 0 && (int64_t)N * T < ((int64_t)1 << 30),
                 "icrl_policy_lstm_bptt_persist: need 0 < N <= 4096, T > 0, N T < 2^30");
    ICRL_REQUIRE(aligned16(gates_act) && aligned16(c_all) && aligned16(d_h) && aligned16(w_hh) && aligned16(dgates) &&
                 aligned16(dh0) && aligned16(work), "icrl_policy_lstm_bptt_persist: buffers must be 16-byte aligned");
    ICRL_REQUIRE(icrl_chain_pair_supported(N),
                 "icrl_policy_lstm_bptt_persist: %d rows need more workgroups than the device can hold at once", N);
    BwdArgs a{};
    Geom& g = a.g;
    g.n = N; g.p0 = 1; g.n_calls = T; g.fc = 1; g.c0 = 1;
    g.total = (int64_t)N * T; g.K = N; g.Ls = T; g.W = 0; g.stride = N;
    a.w_hh = w_hh; a.d_emit = d_h; a.dgv = dgates;
    a.saved = gates_act; a.gate_stride = 4 * H;
    a.cc = c_all + (int64_t)N * H; a.c_stride = H;   // the cell state AFTER step t; the one before it is a row stride back
    hipStream_t st = icrl::as_stream(stream);
    ICRL_HIP(hipMemsetAsync(status, 0, sizeof(int), st));
    unsigned* abort_w = nullptr;
    if (int rc = bwd_persist(a, work, st, &abort_w, 0, nullptr, dh0)) return rc;
    hipLaunchKernelGGL(flag_on_abort, dim3(1), dim3(1), 0, st, abort_w, status, 5);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

// ------------------------------------------------------------------ backward chain, owner-computes launch (seg_bwd_rs)
extern "C" int64_t icrl_chain_bwd_f16x2_workspace_floats(int segments) { return segments > 0 ? bwd_rs_floats(segments) : 0; }

extern "C" int icrl_chain_bwd_f16x2_supported(int segments) {
    // all 16 x ceil(segments / 32) workgroups (512 threads each) must be resident at once
    static int cus = -1, per_cu = 0;
    if (cus < 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, seg_bwd_rs, 512, 0) != hipSuccess) return 0;
        cus = n;
    }
    if (segments < 1) return 0;
    return 16 * ((segments + 31) / 32) <= cus * per_cu ? 1 : 0;
}

extern "C" int icrl_lstm_chain_bwd_seg_f16x2(const float* w_hh, const float* saved, const float* d_emit, int N, int p0,
                                             int n_calls, int full_calls, int segments, int warmup, float* dgates,
                                             float* work, float* seam_err, icrl_stream_t stream) {
    const char* who = "icrl_lstm_chain_bwd_seg_f16x2";
    RsArgs a{};
    if (int rc = make_geom(who, N, p0, n_calls, full_calls, segments, warmup, a.g)) return rc;
    ICRL_REQUIRE(w_hh && saved && d_emit && dgates && work && seam_err, "%s: null pointer", who);
    ICRL_REQUIRE(aligned16(w_hh) && aligned16(saved) && aligned16(d_emit) && aligned16(dgates) && aligned16(work),
                 "%s: buffers must be 16-byte aligned", who);
    ICRL_REQUIRE(icrl_chain_bwd_f16x2_supported(segments),
                 "%s: %d segments need more workgroups than the device can hold at once; its workgroups wait for each "
                 "other and must all be resident", who, segments);
    hipStream_t st = icrl::as_stream(stream);
    a.w_hh = w_hh; a.saved = saved; a.d_emit = d_emit; a.dgv = dgates;
    unsigned* abort_w = nullptr;
    if (int rc = bwd_rs(a, work, st, &abort_w, seam_err)) return rc;
    hipLaunchKernelGGL(seam_check_bwd, dim3(seam_blocks(a.g.K * 2 * H)), dim3(256), 0, st, a.seam_a, a.dh_fin, a.dc,
                       a.g, seam_err, seam_err + 1, (const unsigned*)abort_w);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_policy_lstm_bptt_f16x2(const float* gates_act, const float* c_all, const float* d_h,
                                           const float* w_hh, int N, int T, float* dgates, float* dh0, float* work,
                                           int* status, icrl_stream_t stream) {
    const char* who = "icrl_policy_lstm_bptt_f16x2";
    ICRL_REQUIRE(gates_act && c_all && d_h && w_hh && dgates && dh0 && work && status, "%s: null pointer", who);
    ICRL_REQUIRE(N > 0 && N <= 4096 && T > 0 && (int64_t)N * T < ((int64_t)1 << 30),
                 "%s: need 0 < N <= 4096, T > 0, N T < 2^30", who);
    ICRL_REQUIRE(aligned16(gates_act) && aligned16(c_all) && aligned16(d_h) && aligned16(w_hh) && aligned16(dgates) &&
                 aligned16(dh0) && aligned16(work), "%s: buffers must be 16-byte aligned", who);
    ICRL_REQUIRE(icrl_chain_bwd_f16x2_supported(N), "%s: %d rows need more workgroups than the device can hold at once",
                 who, N);
    RsArgs a{};
    Geom& g = a.g;
    g.n = N; g.p0 = 1; g.n_calls = T; g.fc = 1; g.c0 = 1;
    g.total = (int64_t)N * T; g.K = N; g.Ls = T; g.W = 0; g.stride = N;
    a.w_hh = w_hh; a.d_emit = d_h; a.dgv = dgates;
    a.saved = gates_act; a.gate_stride = 4 * H;
    a.cc = c_all + (int64_t)N * H; a.c_stride = H;
    hipStream_t st = icrl::as_stream(stream);
    ICRL_HIP(hipMemsetAsync(status, 0, sizeof(int), st));
    unsigned* abort_w = nullptr;
    if (int rc = bwd_rs(a, work, st, &abort_w, nullptr, dh0)) return rc;
    hipLaunchKernelGGL(flag_on_abort, dim3(1), dim3(1), 0, st, abort_w, status, 5);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}


// ------------------------------------------------------------------ the give-up path, on its own (test hook)
namespace {
__global__ void wait_that_runs_out(unsigned* flags, unsigned* abort_w, unsigned epoch, int wide) {
    const int lane = threadIdx.x & 63;
    if (wide) wait_row_group32(flags, epoch, abort_w, lane);
    else wait_row_group(flags, epoch, abort_w, lane);
}
}  // namespace

extern "C" int icrl_chain_give_up_selftest(unsigned* flags, unsigned* abort_words, unsigned epoch, int wide, float* err,
                                           icrl_stream_t stream) {
    ICRL_REQUIRE(flags && abort_words && err, "icrl_chain_give_up_selftest: null pointer");
    hipStream_t st = icrl::as_stream(stream);
    hipLaunchKernelGGL(wait_that_runs_out, dim3(1), dim3(64), 0, st, flags, abort_words, epoch, wide);
    hipLaunchKernelGGL(poison_on_abort, dim3(1), dim3(1), 0, st, abort_words, err, err + 1, wide ? 0 : 1);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}
