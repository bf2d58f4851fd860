// Persistent batch-1 recurrent chains (value-network LSTM forward/backward, reward-network
// GRU forward/backward).
//
// The reference runs nn.LSTM / nn.GRU with (seq = N, batch = 1), so a minibatch is ONE
// recurrence of ~10^5 dependent 512-wide steps.  Each step is a 512 x (G*512) mat-vec plus
// a pointwise cell: far too little work for a launch per step, and W_hh (4 MB f32) does
// not fit one CU.  So ICRL_CHAIN_WGS = 64 workgroups stay resident for the whole chain;
// workgroup b owns hidden units [8b, 8b+8), keeps the 4 (3) gate rows of W_hh for those
// units in registers (64 floats per lane), and once per step
//   1. gathers the full previous state (512 floats; 2048 gate gradients in the backward
//      pass) from the other workgroups into LDS,
//   2. does its slice of the mat-vec (each row's 512 columns are spread over 8 lanes in
//      interleaved float4 chunks so that the ds_read_b128 of a row group is contiguous),
//   3. runs the cell for its 8 units and publishes their new state.
// The exchange uses 8-byte {tag = step+1, value} granules written by ONE write-through
// (agent-scope relaxed atomic) store and polled with agent-scope relaxed loads, so
// the data is its own flag: no fence, no dependence on dispatch order or on which XCD a
// workgroup landed on.  Two granule sets alternate by step parity; a producer can only be
// one step ahead of the slowest consumer, because publishing step t+1 needs every
// workgroup's step-t output.  Every spin is bounded by a wall-clock limit; on time-out the
// workgroup raises *status and every workgroup drains.
#include "icrl_common.h"

namespace {
using namespace icrl;
typedef unsigned long long u64;

constexpr int H = kHidden;
constexpr int WGS = ICRL_CHAIN_WGS;
constexpr int UPW = H / WGS;           // hidden units per workgroup (8)
constexpr int LPR = 8;                 // lanes per weight row in the forward kernels
constexpr int GRAN_PER_SET = 2048;     // granules per parity set (forward uses 512 of them)
constexpr int64_t SYNC_BYTES = 2 * GRAN_PER_SET * 8;
constexpr u64 TIMEOUT_TICKS = 400000000ull;  // 4 s of the 100 MHz wall clock

#define RLX __ATOMIC_RELAXED
#define AGENT __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ u64 pack(unsigned epoch, float v) {
    return ((u64)epoch << 32) | (u64)__float_as_uint(v);
}

// Poll K consecutive granules until every tag equals `epoch`.  false = timed out.
template <int K>
__device__ __forceinline__ bool poll_granules(const u64* g, unsigned epoch, float (&v)[K]) {
    unsigned spins = 0;
    u64 t0 = 0;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const u64 x = __hip_atomic_load(g + k, RLX, AGENT);
            v[k] = __uint_as_float((unsigned)x);
            ok &= ((unsigned)(x >> 32) == epoch);
        }
        if (ok) return true;
        if ((++spins & 255u) == 0u) {
            const u64 now = wall_clock64();
            if (t0 == 0) t0 = now;
            else if (now - t0 > TIMEOUT_TICKS) return false;
        }
    }
}

// Walks the chain's positions: for call s: for column t < p0x + s: for row i < n.
struct PosIter {
    int n, p0x, s, t, i;
    __device__ void first(int n_, int p0x_) { n = n_; p0x = p0x_; s = 0; t = 0; i = 0; }
    __device__ void last(int n_, int p0x_, int n_calls) {
        n = n_; p0x = p0x_; s = n_calls - 1; t = p0x + s - 1; i = n - 1;
    }
    __device__ int64_t row() const { return (int64_t)t * n + i; }       // row of xproj
    __device__ bool emits() const { return t == p0x + s - 1; }           // last column of call s
    __device__ int64_t emit_row() const { return (int64_t)s * n + i; }
    __device__ void next() { if (++i == n) { i = 0; if (++t == p0x + s) { t = 0; ++s; } } }
    __device__ void prev() { if (--i < 0) { i = n - 1; if (--t < 0) { --s; t = p0x + s - 1; } } }
};

__device__ __forceinline__ float sigm(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// ------------------------------------------------------------------ forward (LSTM / GRU)
template <bool GRU>
__global__ __launch_bounds__(256) void chain_fwd_kernel(const float* __restrict__ xproj,
                                                        const float* __restrict__ w_hh,
                                                        const float* __restrict__ b_hh, int n, int p0x,
                                                        int n_calls, int64_t total,
                                                        float* __restrict__ h_emit, float* __restrict__ saved,
                                                        const float* __restrict__ state_in,
                                                        float* __restrict__ state_out, u64* gran, int* status) {
    constexpr int G = GRU ? 3 : 4;
    __shared__ __attribute__((aligned(16))) float hbuf[2][H];
    __shared__ int abort_flag;
    const int tid = threadIdx.x;
    const int cq = tid & (LPR - 1);        // column chunk
    const int q = (tid >> 3) & 3;          // gate slot (GRU leaves slot 3 idle)
    const int lu = tid >> 5;               // local unit
    const int u = blockIdx.x * UPW + lu;   // hidden unit
    const bool live = q < G;
    if (tid == 0) abort_flag = 0;

    // this lane's 64 weights: row q*H+u, float4 chunks k*LPR + cq
    float4 w[16];
    {
        const float4* wr = reinterpret_cast<const float4*>(w_hh + (int64_t)(live ? q * H + u : 0) * H);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            w[k] = wr[k * LPR + cq];
            if (!live) w[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const float bh = (GRU && live) ? b_hh[q * H + u] : 0.f;

    PosIter it;
    it.first(n, p0x);
    // carried state (h then c, 2*H floats); the reference keeps it across forward() calls
    float c_state = (state_in && !GRU) ? state_in[H + u] : 0.f;
    float h_state = state_in ? state_in[u] : 0.f;
    const int64_t xoff = (int64_t)(live ? q : 0) * H + u;
    float xq = xproj[it.row() * (G * H) + xoff];
    __syncthreads();

    for (int64_t pos = 0; pos < total; ++pos) {
        const unsigned par = (unsigned)pos & 1u;
        // prefetch the next position's input projection (a full step ahead of its use)
        PosIter nx = it;
        nx.next();
        float xq_next = 0.f;
        if (pos + 1 < total) xq_next = xproj[nx.row() * (G * H) + xoff];

        // 1. previous state -> LDS
        if (pos == 0) {
            hbuf[par][2 * tid] = state_in ? state_in[2 * tid] : 0.f;
            hbuf[par][2 * tid + 1] = state_in ? state_in[2 * tid + 1] : 0.f;
        } else {
            float hv[2];
            const bool ok = poll_granules<2>(gran + (par ^ 1u) * GRAN_PER_SET + 2 * tid, (unsigned)pos, hv);
            if (!ok) abort_flag = 1;
            *reinterpret_cast<float2*>(&hbuf[par][2 * tid]) = make_float2(hv[0], hv[1]);
        }
        __syncthreads();
        if (abort_flag) {
            if (tid == 0) atomicExch(status, 1);
            return;
        }

        // 2. this lane's 64 columns of its row
        const float4* h4 = reinterpret_cast<const float4*>(hbuf[par]);
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float4 hv = h4[k * LPR + cq];
            a0 = fmaf(w[k].x, hv.x, a0);
            a1 = fmaf(w[k].y, hv.y, a1);
            a2 = fmaf(w[k].z, hv.z, a2);
            a3 = fmaf(w[k].w, hv.w, a3);
        }
        float acc = (a0 + a1) + (a2 + a3);
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);

        // 3. cell; lanes of gate slot 0 end up holding the unit's new state
        float h_new;
        if (!GRU) {
            const float pre = acc + xq;
            const float sg = sigm(q == 2 ? 2.f * pre : pre);
            const float act = (q == 2) ? 2.f * sg - 1.f : sg;      // tanh(x) = 2*sigmoid(2x) - 1
            const float gf = __shfl_xor(act, 8, 64);               // slot q^1
            const float gg = __shfl_xor(act, 16, 64);              // slot q^2
            const float go = __shfl_xor(act, 24, 64);              // slot q^3
            c_state = gf * c_state + act * gg;                     // meaningful where q == 0
            const float tc = 2.f * sigm(2.f * c_state) - 1.f;
            h_new = go * tc;
            if (saved && q == 0 && cq < 6) {
                const float sv = cq == 0 ? act : cq == 1 ? gf : cq == 2 ? gg : cq == 3 ? go : cq == 4 ? c_state : h_new;
                saved[pos * (6 * H) + cq * H + u] = sv;
            }
        } else {
            const float hb = acc + bh;
            const float a = sigm(xq + hb);                         // r (slot 0), z (slot 1)
            const float r_at_n = __shfl_xor(a, 16, 64);            // slot 2 receives r
            const float nval = 2.f * sigm(2.f * (xq + r_at_n * hb)) - 1.f;
            const float z = __shfl_xor(a, 8, 64);                  // slot 0 receives z
            const float nn = __shfl_xor(nval, 16, 64);             // slot 0 receives n
            const float hn = __shfl_xor(hb, 16, 64);               // slot 0 receives W_hn.h + b_hn
            h_new = (1.f - z) * nn + z * h_state;
            if (saved && q == 0 && cq < 5) {
                const float sv = cq == 0 ? a : cq == 1 ? z : cq == 2 ? nn : cq == 3 ? hn : h_new;
                saved[pos * (5 * H) + cq * H + u] = sv;
            }
        }
        h_state = h_new;
        if (q == 0 && cq == 0) {
            __hip_atomic_store(gran + par * GRAN_PER_SET + u, pack((unsigned)pos + 1u, h_new), RLX, AGENT);
            if (it.emits()) h_emit[it.emit_row() * H + u] = h_new;
            if (state_out && pos + 1 == total) {
                state_out[u] = h_new;
                if (!GRU) state_out[H + u] = c_state;
            }
        }
        xq = xq_next;
        it = nx;
    }
}

// ------------------------------------------------------------------ LSTM backward
// Walks positions in reverse.  Workgroup b owns units [8b, 8b+8): 32 lanes per unit, lane j
// holds W_hh[rho, u] for the 64 gate rows rho = 4*(k*32 + j) + m.  Per step it computes
// dh_prev[u] = sum_rho W_hh[rho, u] * dgates[rho] from the gathered 2048 gate gradients,
// adds the value head's gradient where the position fed it, runs the cell backward for its
// units and publishes their 4 gate gradients.
__global__ __launch_bounds__(256) void lstm_chain_bwd_kernel(const float* __restrict__ w_hh,
                                                             const float* __restrict__ saved,
                                                             const float* __restrict__ d_emit, int n, int p0x,
                                                             int n_calls, int64_t total,
                                                             float* __restrict__ dgates,
                                                             u64* gran, int* status) {
    __shared__ __attribute__((aligned(16))) float dgb[2][4 * H];
    __shared__ int abort_flag;
    const int tid = threadIdx.x;
    const int j = tid & 31;
    const int lu = tid >> 5;
    const int u = blockIdx.x * UPW + lu;
    if (tid == 0) abort_flag = 0;

    float4 w[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int rho = 4 * (k * 32 + j);
        w[k].x = w_hh[(int64_t)(rho + 0) * H + u];
        w[k].y = w_hh[(int64_t)(rho + 1) * H + u];
        w[k].z = w_hh[(int64_t)(rho + 2) * H + u];
        w[k].w = w_hh[(int64_t)(rho + 3) * H + u];
    }

    PosIter it;
    it.last(n, p0x, n_calls);
    float dc_carry = 0.f;
    // saved activations of the position being processed (all 32 lanes of a unit load the same words)
    const float* sp = saved + (total - 1) * (6 * H) + u;
    float gi = sp[0], gf = sp[H], gg = sp[2 * H], go = sp[3 * H], cc = sp[4 * H];
    float de = it.emits() ? d_emit[it.emit_row() * H + u] : 0.f;
    __syncthreads();

    for (int64_t k = 0; k < total; ++k) {
        const int64_t pos = total - 1 - k;
        const unsigned par = (unsigned)k & 1u;
        // prefetch position pos-1 (its cell state is also this position's c_prev)
        PosIter pv = it;
        float ni = 0.f, nf = 0.f, ng = 0.f, no = 0.f, nc = 0.f, nde = 0.f;
        if (pos > 0) {
            pv.prev();
            const float* sn = saved + (pos - 1) * (6 * H) + u;
            ni = sn[0]; nf = sn[H]; ng = sn[2 * H]; no = sn[3 * H]; nc = sn[4 * H];
            if (pv.emits()) nde = d_emit[pv.emit_row() * H + u];
        }

        float dh = 0.f;
        if (k > 0) {
            float dv[8];
            const bool ok = poll_granules<8>(gran + (par ^ 1u) * GRAN_PER_SET + 8 * tid, (unsigned)k, dv);
            if (!ok) abort_flag = 1;
            float4* dst = reinterpret_cast<float4*>(&dgb[par][8 * tid]);
            dst[0] = make_float4(dv[0], dv[1], dv[2], dv[3]);
            dst[1] = make_float4(dv[4], dv[5], dv[6], dv[7]);
            __syncthreads();
            if (abort_flag) {
                if (tid == 0) atomicExch(status, 2);
                return;
            }
            const float4* d4 = reinterpret_cast<const float4*>(dgb[par]);
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const float4 d = d4[kk * 32 + j];
                a0 = fmaf(w[kk].x, d.x, a0);
                a1 = fmaf(w[kk].y, d.y, a1);
                a2 = fmaf(w[kk].z, d.z, a2);
                a3 = fmaf(w[kk].w, d.w, a3);
            }
            dh = (a0 + a1) + (a2 + a3);
            dh += __shfl_xor(dh, 1, 64);
            dh += __shfl_xor(dh, 2, 64);
            dh += __shfl_xor(dh, 4, 64);
            dh += __shfl_xor(dh, 8, 64);
            dh += __shfl_xor(dh, 16, 64);
        }
        dh += de;

        // cell backward for unit u (identical in the unit's 32 lanes)
        const float tc = 2.f * sigm(2.f * cc) - 1.f;
        const float dc = dh * go * (1.f - tc * tc) + dc_carry;
        const float c_prev = nc;  // zero at pos 0
        const float d_i = dc * gg * gi * (1.f - gi);
        const float d_f = dc * c_prev * gf * (1.f - gf);
        const float d_g = dc * gi * (1.f - gg * gg);
        const float d_o = dh * tc * go * (1.f - go);
        dc_carry = dc * gf;
        if (j < 4) {
            const float dq = j == 0 ? d_i : j == 1 ? d_f : j == 2 ? d_g : d_o;
            dgates[pos * (4 * H) + j * H + u] = dq;
            if (pos > 0)
                __hip_atomic_store(gran + par * GRAN_PER_SET + j * H + u, pack((unsigned)k + 1u, dq), RLX, AGENT);
        }
        gi = ni; gf = nf; gg = ng; go = no; cc = nc; de = nde;
        it = pv;
    }
}

// ------------------------------------------------------------------ GRU backward
// Walks positions in reverse (BPTT of models.py:226 for the reward pre-training loop).  Workgroup b
// owns units [8b, 8b+8): 32 lanes per unit, lane j holds W_hh[rho, u] for the 48 gate rows
// rho = 4*(k*32 + j) + m of the 1536.  Per step: dh = z-path carry + sum_rho W_hh[rho, u] * dgh[rho]
// (the gathered hidden-side gate gradients of the position above) + the injected gradient; then the
// cell backward for its units, which publishes the unit's three hidden-side gate gradients.
// saved[pos] = {r, z, n, hn = W_hn.h_prev + b_hn, h}; the forward pass started from a zero state.
// dgx[pos] = gradient w.r.t. the input-side pre-activations (r, z, n), dgh[pos] = w.r.t. the
// hidden-side ones (r, z, hn): they differ in the n slot because n = tanh(x_n + r * hn).
__global__ __launch_bounds__(256) void gru_chain_bwd_kernel(const float* __restrict__ w_hh,
                                                            const float* __restrict__ saved,
                                                            const float* __restrict__ d_emit, int n, int p0x,
                                                            int n_calls, int64_t total, float* __restrict__ dgx,
                                                            float* __restrict__ dgh, u64* gran, int* status) {
    __shared__ __attribute__((aligned(16))) float dgb[2][3 * H];
    __shared__ int abort_flag;
    const int tid = threadIdx.x;
    const int j = tid & 31;
    const int lu = tid >> 5;
    const int u = blockIdx.x * UPW + lu;
    if (tid == 0) abort_flag = 0;

    float4 w[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const int rho = 4 * (k * 32 + j);
        w[k].x = w_hh[(int64_t)(rho + 0) * H + u];
        w[k].y = w_hh[(int64_t)(rho + 1) * H + u];
        w[k].z = w_hh[(int64_t)(rho + 2) * H + u];
        w[k].w = w_hh[(int64_t)(rho + 3) * H + u];
    }

    PosIter it;
    it.last(n, p0x, n_calls);
    float carry = 0.f;   // dh * z of the position above
    const float* sp = saved + (total - 1) * (5 * H) + u;
    float gr = sp[0], gz = sp[H], gn = sp[2 * H], ghn = sp[3 * H];
    float de = it.emits() ? d_emit[it.emit_row() * H + u] : 0.f;
    __syncthreads();

    for (int64_t k = 0; k < total; ++k) {
        const int64_t pos = total - 1 - k;
        const unsigned par = (unsigned)k & 1u;
        // prefetch position pos-1: its h is this position's h_prev
        PosIter pv = it;
        float nr = 0.f, nz = 0.f, nnv = 0.f, nhn = 0.f, hprev = 0.f, nde = 0.f;
        if (pos > 0) {
            pv.prev();
            const float* sn = saved + (pos - 1) * (5 * H) + u;
            nr = sn[0]; nz = sn[H]; nnv = sn[2 * H]; nhn = sn[3 * H]; hprev = sn[4 * H];
            if (pv.emits()) nde = d_emit[pv.emit_row() * H + u];
        }

        float dh = 0.f;
        if (k > 0) {
            float dv[6];
            const bool ok = poll_granules<6>(gran + (par ^ 1u) * GRAN_PER_SET + 6 * tid, (unsigned)k, dv);
            if (!ok) abort_flag = 1;
            float2* dst = reinterpret_cast<float2*>(&dgb[par][6 * tid]);
            dst[0] = make_float2(dv[0], dv[1]);
            dst[1] = make_float2(dv[2], dv[3]);
            dst[2] = make_float2(dv[4], dv[5]);
            __syncthreads();
            if (abort_flag) {
                if (tid == 0) atomicExch(status, 3);
                return;
            }
            const float4* d4 = reinterpret_cast<const float4*>(dgb[par]);
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
            for (int kk = 0; kk < 12; ++kk) {
                const float4 d = d4[kk * 32 + j];
                a0 = fmaf(w[kk].x, d.x, a0);
                a1 = fmaf(w[kk].y, d.y, a1);
                a2 = fmaf(w[kk].z, d.z, a2);
                a3 = fmaf(w[kk].w, d.w, a3);
            }
            dh = (a0 + a1) + (a2 + a3);
            dh += __shfl_xor(dh, 1, 64);
            dh += __shfl_xor(dh, 2, 64);
            dh += __shfl_xor(dh, 4, 64);
            dh += __shfl_xor(dh, 8, 64);
            dh += __shfl_xor(dh, 16, 64);
        }
        dh += carry + de;

        // cell backward for unit u (identical in the unit's 32 lanes)
        const float dn_pre = dh * (1.f - gz) * (1.f - gn * gn);
        const float dz_pre = dh * (hprev - gn) * gz * (1.f - gz);
        const float dr_pre = dn_pre * ghn * gr * (1.f - gr);
        const float d_hn = dn_pre * gr;
        carry = dh * gz;
        if (j < 3) {
            const float qx = j == 0 ? dr_pre : j == 1 ? dz_pre : dn_pre;
            const float qh = j == 0 ? dr_pre : j == 1 ? dz_pre : d_hn;
            dgx[pos * (3 * H) + j * H + u] = qx;
            dgh[pos * (3 * H) + j * H + u] = qh;
            if (pos > 0)
                __hip_atomic_store(gran + par * GRAN_PER_SET + j * H + u, pack((unsigned)k + 1u, qh), RLX, AGENT);
        }
        gr = nr; gz = nz; gn = nnv; ghn = nhn; de = nde;
        it = pv;
    }
}

int chain_args_ok(const char* who, int N, int p0, int n_calls) {
    if (N <= 0 || p0 <= 0 || n_calls <= 0) {
        set_error("%s: N, p0 and n_calls must be positive (got %d, %d, %d)", who, N, p0, n_calls);
        return 0;
    }
    return 1;
}

int64_t chain_len(int N, int p0x, int n_calls) {
    int64_t cols = 0;
    for (int s = 0; s < n_calls; ++s) cols += p0x + s;
    return cols * N;
}

}  // namespace

extern "C" int64_t icrl_chain_sync_bytes(void) { return SYNC_BYTES; }

extern "C" int icrl_lstm_chain_fwd(const float* xproj, const float* w_hh, int N, int p0, int n_calls,
                                   float* h_emit, float* saved, const float* state_in, float* state_out,
                                   void* sync_ws, int32_t* status, icrl_stream_t stream) {
    if (!chain_args_ok("icrl_lstm_chain_fwd", N, p0, n_calls)) return ICRL_ERR_ARG;
    ICRL_REQUIRE(xproj && w_hh && h_emit && sync_ws && status, "icrl_lstm_chain_fwd: null pointer");
    ICRL_REQUIRE(reinterpret_cast<uintptr_t>(w_hh) % 16 == 0 && reinterpret_cast<uintptr_t>(sync_ws) % 16 == 0,
                 "icrl_lstm_chain_fwd: w_hh and sync_ws must be 16-byte aligned");
    hipStream_t st = icrl::as_stream(stream);
    ICRL_HIP(hipMemsetAsync(sync_ws, 0, SYNC_BYTES, st));
    ICRL_HIP(hipMemsetAsync(status, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(chain_fwd_kernel<false>, dim3(WGS), dim3(256), 0, st, xproj, w_hh, (const float*)nullptr, N,
                       p0, n_calls, chain_len(N, p0, n_calls), h_emit, saved, state_in, state_out,
                       reinterpret_cast<u64*>(sync_ws), reinterpret_cast<int*>(status));
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_gru_chain_fwd(const float* xproj, const float* w_hh, const float* b_hh, int N, int p0,
                                  int n_calls, float* h_emit, float* saved, const float* state_in,
                                  float* state_out, void* sync_ws, int32_t* status, icrl_stream_t stream) {
    if (!chain_args_ok("icrl_gru_chain_fwd", N, p0, n_calls)) return ICRL_ERR_ARG;
    ICRL_REQUIRE(xproj && w_hh && b_hh && h_emit && sync_ws && status, "icrl_gru_chain_fwd: null pointer");
    ICRL_REQUIRE(reinterpret_cast<uintptr_t>(w_hh) % 16 == 0 && reinterpret_cast<uintptr_t>(sync_ws) % 16 == 0,
                 "icrl_gru_chain_fwd: w_hh and sync_ws must be 16-byte aligned");
    hipStream_t st = icrl::as_stream(stream);
    ICRL_HIP(hipMemsetAsync(sync_ws, 0, SYNC_BYTES, st));
    ICRL_HIP(hipMemsetAsync(status, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(chain_fwd_kernel<true>, dim3(WGS), dim3(256), 0, st, xproj, w_hh, b_hh, N, p0, n_calls,
                       chain_len(N, p0, n_calls), h_emit, saved, state_in, state_out,
                       reinterpret_cast<u64*>(sync_ws), reinterpret_cast<int*>(status));
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_lstm_chain_bwd(const float* w_hh, const float* saved, const float* d_emit, int N, int p0,
                                   int n_calls, float* dgates, void* sync_ws, int32_t* status,
                                   icrl_stream_t stream) {
    if (!chain_args_ok("icrl_lstm_chain_bwd", N, p0, n_calls)) return ICRL_ERR_ARG;
    ICRL_REQUIRE(w_hh && saved && d_emit && dgates && sync_ws && status, "icrl_lstm_chain_bwd: null pointer");
    ICRL_REQUIRE(reinterpret_cast<uintptr_t>(sync_ws) % 16 == 0, "icrl_lstm_chain_bwd: sync_ws must be 16-byte aligned");
    hipStream_t st = icrl::as_stream(stream);
    ICRL_HIP(hipMemsetAsync(sync_ws, 0, SYNC_BYTES, st));
    ICRL_HIP(hipMemsetAsync(status, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(lstm_chain_bwd_kernel, dim3(WGS), dim3(256), 0, st, w_hh, saved, d_emit, N, p0, n_calls,
                       chain_len(N, p0, n_calls), dgates, reinterpret_cast<u64*>(sync_ws),
                       reinterpret_cast<int*>(status));
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_gru_chain_bwd(const float* w_hh, const float* saved, const float* d_emit, int N, int p0,
                                  int n_calls, float* dgates_x, float* dgates_h, void* sync_ws, int32_t* status,
                                  icrl_stream_t stream) {
    if (!chain_args_ok("icrl_gru_chain_bwd", N, p0, n_calls)) return ICRL_ERR_ARG;
    ICRL_REQUIRE(w_hh && saved && d_emit && dgates_x && dgates_h && sync_ws && status,
                 "icrl_gru_chain_bwd: null pointer");
    ICRL_REQUIRE(reinterpret_cast<uintptr_t>(sync_ws) % 16 == 0, "icrl_gru_chain_bwd: sync_ws must be 16-byte aligned");
    hipStream_t st = icrl::as_stream(stream);
    ICRL_HIP(hipMemsetAsync(sync_ws, 0, SYNC_BYTES, st));
    ICRL_HIP(hipMemsetAsync(status, 0, sizeof(int32_t), st));
    hipLaunchKernelGGL(gru_chain_bwd_kernel, dim3(WGS), dim3(256), 0, st, w_hh, saved, d_emit, N, p0, n_calls,
                       chain_len(N, p0, n_calls), dgates_x, dgates_h, reinterpret_cast<u64*>(sync_ws),
                       reinterpret_cast<int*>(status));
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}
