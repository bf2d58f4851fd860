// One step of the policy network's LSTM (models.py:73, 80) for a whole minibatch in ONE launch:
// embedding gather, x.W_ih^T, h.W_hh^T, both biases and the cell.
//
// Workgroup tile = 32 minibatch rows x (4 gates x 32 hidden units); wave q owns gate q's 32x32
// accumulator (v_mfma_f32_32x32x2_f32).  The k loop first walks the word-vector width with the A
// rows gathered from the embedding table by token id, then the hidden width with the A rows taken
// from the previous h; k advances in stages of 64 through double-buffered LDS.  The four gate tiles
// meet in LDS and the cell runs in the epilogue, which writes the activated gates (kept for the
// backward pass), c and h.  Workgroups of the first unit tile also copy their gathered embedding rows
// out, which is the x matrix the weight gradient needs later.
#include "icrl_common.h"

namespace {
using namespace icrl;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int H = kHidden;
constexpr int BK = 64;
constexpr int LDT = BK + 4;
constexpr int PLD = 33;
constexpr int NT = 256;
constexpr int ROWS = 32 + 4 * 32;
constexpr int F4 = ROWS * (BK / 4);
constexpr int PER = F4 / NT;  // 10

__device__ __forceinline__ float sg(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float th(float x) { return 2.f * sg(2.f * x) - 1.f; }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 ldp4(const float* p) { return make_float4(p[0], p[1], p[2], p[3]); }

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

__global__ __launch_bounds__(NT) void policy_lstm_step_kernel(const float* __restrict__ emb, int E,
                                                              const int64_t* __restrict__ tok,
                                                              const float* __restrict__ w_ih,
                                                              const float* __restrict__ w_hh,
                                                              const float* __restrict__ b_ih,
                                                              const float* __restrict__ b_hh,
                                                              const float* __restrict__ h_cur,
                                                              const float* __restrict__ c_cur,
                                                              float* __restrict__ gates_act, float* __restrict__ c_nxt,
                                                              float* __restrict__ h_nxt, float* __restrict__ x_out,
                                                              int N) {
    __shared__ __attribute__((aligned(16))) float lds[2 * ROWS * LDT];
    const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int row0 = blockIdx.y * 32, u0 = blockIdx.x * 32;

    // per staged float4: its source row in the x phase and in the h phase (null = zero fill)
    const float* src_x[PER];
    const float* src_h[PER];
    float* copy_x[PER];
    int dst[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int f = tid + NT * i;
        const int r = f / (BK / 4), kq = f % (BK / 4);
        dst[i] = r * LDT + 4 * kq;
        copy_x[i] = nullptr;
        if (r < 32) {
            // rows past N read row N-1 instead of branching in the k loop; their results are never stored
            const int row = min(row0 + r, N - 1);
            src_x[i] = emb + tok[row] * (int64_t)E + 4 * kq;
            src_h[i] = h_cur + (int64_t)row * H + 4 * kq;
            if (x_out && blockIdx.x == 0 && row0 + r < N) copy_x[i] = x_out + (int64_t)row * E + 4 * kq;
        } else {
            const int rb = r - 32;
            const int64_t wrow = (rb >> 5) * H + u0 + (rb & 31);
            src_x[i] = w_ih + wrow * E + 4 * kq;
            src_h[i] = w_hh + wrow * H + 4 * kq;
        }
    }
    // epilogue operands, fetched early
    const int rr = tid >> 3, uq = 4 * (tid & 7);
    const int row = row0 + rr, u = u0 + uq;
    float4 bias[4], cp = zero4();
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) bias[gq] = add4(ld4(b_ih + gq * H + u), ld4(b_hh + gq * H + u));
    if (row < N) cp = ld4(c_cur + (int64_t)row * H + u);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nk1 = E / BK, nk = nk1 + H / BK;
    float4 stage[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        stage[i] = ld4(src_x[i]);
        if (copy_x[i]) st4(copy_x[i], stage[i]);
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) st4(lds + dst[i], stage[i]);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const float* cur = lds + (kt & 1) * ROWS * LDT;
        const int kn = kt + 1;
        if (kn < nk) {
            if (kn < nk1) {
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    stage[i] = ld4(src_x[i] + kn * BK);
                    if (copy_x[i]) st4(copy_x[i] + kn * BK, stage[i]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < PER; ++i) stage[i] = ld4(src_h[i] + (kn - nk1) * BK);
            }
        }
        const float* a_row = cur + l31 * LDT;
        const float* b_row = cur + (32 + q * 32 + l31) * LDT;
#pragma unroll
        for (int grp = 0; grp < BK / 16; ++grp) mfma_group(acc, a_row, b_row, grp, hh);
        if (kn < nk) {
            float* nxt = lds + (kn & 1) * ROWS * LDT;
#pragma unroll
            for (int i = 0; i < PER; ++i) st4(nxt + dst[i], stage[i]);
        }
        __syncthreads();
    }
    float* P = lds + q * 32 * PLD;
#pragma unroll
    for (int r = 0; r < 16; ++r) P[((r & 3) + 8 * (r >> 2) + 4 * hh) * PLD + l31] = acc[r];
    __syncthreads();
    if (row >= N) return;
    const float* pr = lds + rr * PLD + uq;
    float4 gi = add4(ldp4(pr), bias[0]), gf = add4(ldp4(pr + 32 * PLD), bias[1]);
    float4 gg = add4(ldp4(pr + 64 * PLD), bias[2]), go = add4(ldp4(pr + 96 * PLD), bias[3]);
    float4 c, h;
#define CELL(k)                                                              \
    gi.k = sg(gi.k); gf.k = sg(gf.k); gg.k = th(gg.k); go.k = sg(go.k);      \
    c.k = gf.k * cp.k + gi.k * gg.k;                                         \
    h.k = go.k * th(c.k);
    CELL(x) CELL(y) CELL(z) CELL(w)
#undef CELL
    float* ga = gates_act + (int64_t)row * (4 * H) + u;
    st4(ga, gi); st4(ga + H, gf); st4(ga + 2 * H, gg); st4(ga + 3 * H, go);
    st4(c_nxt + (int64_t)row * H + u, c);
    st4(h_nxt + (int64_t)row * H + u, h);
}

// The same step with the k loop software-pipelined (word-vector width a compile-time NK1 * 64): global
// loads run three stages ahead in a register ring, operand fragments of the next 16-k group are read
// from LDS before the current group's MFMAs issue, and the one barrier per stage sits between the last
// two groups, so the matrix pipe never drains.  Same arithmetic in the same order: same bits.
struct Frag { float4 a0, a1, b0, b1; };
__device__ __forceinline__ Frag load_frag(const float* a_row, const float* b_row, int grp, int hh) {
    Frag f;
    f.a0 = ld4(a_row + 16 * grp + 8 * hh); f.a1 = ld4(a_row + 16 * grp + 8 * hh + 4);
    f.b0 = ld4(b_row + 16 * grp + 8 * hh); f.b1 = ld4(b_row + 16 * grp + 8 * hh + 4);
    return f;
}
__device__ __forceinline__ void mfma8(f32x16& acc, const Frag& f) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0.x, f.b0.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0.y, f.b0.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0.z, f.b0.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0.w, f.b0.w, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1.x, f.b1.x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1.y, f.b1.y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1.z, f.b1.z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1.w, f.b1.w, acc, 0, 0, 0);
}

template <int NK1>
__global__ __launch_bounds__(NT) void policy_lstm_step_pipelined(const float* __restrict__ emb,
                                                                 const int64_t* __restrict__ tok,
                                                                 const float* __restrict__ w_ih,
                                                                 const float* __restrict__ w_hh,
                                                                 const float* __restrict__ b_ih,
                                                                 const float* __restrict__ b_hh,
                                                                 const float* __restrict__ h_cur,
                                                                 const float* __restrict__ c_cur,
                                                                 float* __restrict__ gates_act, float* __restrict__ c_nxt,
                                                                 float* __restrict__ h_nxt, float* __restrict__ x_out,
                                                                 int N) {
    constexpr int E = NK1 * BK, NK = NK1 + H / BK, DEPTH = 3, BUF = ROWS * LDT;
    static_assert(NK > DEPTH, "ring depth");
    __shared__ __attribute__((aligned(16))) float lds[2 * BUF];
    const int tid = threadIdx.x, lane = tid & 63, q = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int row0 = blockIdx.y * 32, u0 = blockIdx.x * 32;

    const float* src_x[PER];
    const float* src_h[PER];
    float* copy_x[PER];
    int dst[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int f = tid + NT * i;
        const int r = f / (BK / 4), kq = f % (BK / 4);
        dst[i] = r * LDT + 4 * kq;
        copy_x[i] = nullptr;
        if (r < 32) {
            const int row = min(row0 + r, N - 1);   // rows past N read row N-1; their results are never stored
            src_x[i] = emb + tok[row] * (int64_t)E + 4 * kq;
            src_h[i] = h_cur + (int64_t)row * H + 4 * kq;
            if (x_out && blockIdx.x == 0 && row0 + r < N) copy_x[i] = x_out + (int64_t)row * E + 4 * kq;
        } else {
            const int rb = r - 32;
            const int64_t wrow = (rb >> 5) * H + u0 + (rb & 31);
            src_x[i] = w_ih + wrow * E + 4 * kq;
            src_h[i] = w_hh + wrow * H + 4 * kq;
        }
    }
    const int rr = tid >> 3, uq = 4 * (tid & 7);
    const int row = row0 + rr, u = u0 + uq;
    float4 bias[4], cp = zero4();
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) bias[gq] = add4(ld4(b_ih + gq * H + u), ld4(b_hh + gq * H + u));
    if (row < N) cp = ld4(c_cur + (int64_t)row * H + u);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 ring[DEPTH][PER];
    // stage kt of the k loop: word-vector columns first, then h.  The gathered rows are copied out when
    // their stage moves from the ring to LDS (by then the load has landed: a copy right behind the load
    // would wait for it and drain the ring).
#define STAGE_TO_LDS(buf, slot, kt)                                                                 \
    {                                                                                               \
        _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                           \
            st4(lds + (buf) + dst[i], ring[slot][i]);                                               \
            if ((kt) < NK1 && copy_x[i]) st4(copy_x[i] + (kt) * BK, ring[slot][i]);                 \
        }                                                                                           \
    }
#define STAGE_LOAD(slot, kt)                                                                        \
    {                                                                                               \
        _Pragma("unroll") for (int i = 0; i < PER; ++i) {                                           \
            if ((kt) < NK1) {                                                                       \
                ring[slot][i] = ld4(src_x[i] + (kt) * BK);                                          \
            } else {                                                                                \
                ring[slot][i] = ld4(src_h[i] + ((kt) - NK1) * BK);                                  \
            }                                                                                       \
        }                                                                                           \
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) STAGE_LOAD(d, d)
    STAGE_TO_LDS(0, 0, 0)
    STAGE_LOAD(0, DEPTH)
    __syncthreads();
    const float* a_row = lds + l31 * LDT;
    const float* b_row = lds + (32 + q * 32 + l31) * LDT;
    Frag f0 = load_frag(a_row, b_row, 0, hh), f1;
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) {
        const int co = (kt & 1) * BUF, cn = ((kt + 1) & 1) * BUF;
        f1 = load_frag(a_row + co, b_row + co, 1, hh);
        mfma8(acc, f0);
        if (kt + 1 < NK) {   // nobody reads buffer `cn` any more: the previous stage's barrier came after its last read
            STAGE_TO_LDS(cn, (kt + 1) % DEPTH, kt + 1)
            if (kt + 1 + DEPTH < NK) STAGE_LOAD((kt + 1) % DEPTH, kt + 1 + DEPTH)
        }
        f0 = load_frag(a_row + co, b_row + co, 2, hh);
        mfma8(acc, f1);
        f1 = load_frag(a_row + co, b_row + co, 3, hh);
        mfma8(acc, f0);
        if (kt + 1 < NK) {
            __syncthreads();
            f0 = load_frag(a_row + cn, b_row + cn, 0, hh);
        }
        mfma8(acc, f1);
    }
#undef STAGE_LOAD
#undef STAGE_TO_LDS
    __syncthreads();   // every wave is done reading before the gate tiles overlay the stage buffers
    float* P = lds + q * 32 * PLD;
#pragma unroll
    for (int r = 0; r < 16; ++r) P[((r & 3) + 8 * (r >> 2) + 4 * hh) * PLD + l31] = acc[r];
    __syncthreads();
    if (row >= N) return;
    const float* pr = lds + rr * PLD + uq;
    float4 gi = add4(ldp4(pr), bias[0]), gf = add4(ldp4(pr + 32 * PLD), bias[1]);
    float4 gg = add4(ldp4(pr + 64 * PLD), bias[2]), go = add4(ldp4(pr + 96 * PLD), bias[3]);
    float4 c, h;
#define CELL(k)                                                              \
    gi.k = sg(gi.k); gf.k = sg(gf.k); gg.k = th(gg.k); go.k = sg(go.k);      \
    c.k = gf.k * cp.k + gi.k * gg.k;                                         \
    h.k = go.k * th(c.k);
    CELL(x) CELL(y) CELL(z) CELL(w)
#undef CELL
    float* ga = gates_act + (int64_t)row * (4 * H) + u;
    st4(ga, gi); st4(ga + H, gf); st4(ga + 2 * H, gg); st4(ga + 3 * H, go);
    st4(c_nxt + (int64_t)row * H + u, c);
    st4(h_nxt + (int64_t)row * H + u, h);
}

bool aligned16(const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; }

}  // namespace

extern "C" int icrl_policy_lstm_step(const float* emb, int wordvec_dim, const int64_t* tokens, const float* w_ih,
                                     const float* w_hh, const float* b_ih, const float* b_hh, const float* h_cur,
                                     const float* c_cur, float* gates_act, float* c_next, float* h_next, float* x_out,
                                     int N, icrl_stream_t stream) {
    ICRL_REQUIRE(emb && tokens && w_ih && w_hh && b_ih && b_hh && h_cur && c_cur && gates_act && c_next && h_next,
                 "icrl_policy_lstm_step: null pointer");
    ICRL_REQUIRE(N > 0 && wordvec_dim > 0 && wordvec_dim % 64 == 0,
                 "icrl_policy_lstm_step: N > 0 and a word-vector width that is a multiple of 64 are required (got %d, %d)",
                 N, wordvec_dim);
    ICRL_REQUIRE(aligned16(emb) && aligned16(w_ih) && aligned16(w_hh) && aligned16(b_ih) && aligned16(b_hh) &&
                 aligned16(h_cur) && aligned16(c_cur) && aligned16(gates_act) && aligned16(c_next) &&
                 aligned16(h_next) && aligned16(x_out), "icrl_policy_lstm_step: buffers must be 16-byte aligned");
    const dim3 grid(H / 32, (N + 31) / 32);
    if (wordvec_dim == 512)
        hipLaunchKernelGGL(policy_lstm_step_pipelined<8>, grid, dim3(NT), 0, icrl::as_stream(stream), emb, tokens, w_ih,
                           w_hh, b_ih, b_hh, h_cur, c_cur, gates_act, c_next, h_next, x_out, N);
    else
        hipLaunchKernelGGL(policy_lstm_step_kernel, grid, dim3(NT), 0, icrl::as_stream(stream), emb, wordvec_dim, tokens,
                           w_ih, w_hh, b_ih, b_hh, h_cur, c_cur, gates_act, c_next, h_next, x_out, N);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}
