// Memory-bound kernels of the A2C step: embedding gather / scatter-add, bias-gradient
// column sums, the batched LSTM cell, value head finish, cosine reward, the
// advantage / policy-gradient loss, d-logits and Adam.  All are HBM-bound: 16-byte
// accesses per lane, grid capped at ~2048 workgroups with grid-stride loops.
#include <algorithm>

#include "icrl_common.h"

namespace {
using namespace icrl;

constexpr int H = kHidden;

inline int grid_for(int64_t work_items, int block) {
    int64_t b = (work_items + block - 1) / block;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (int)b;
}

// ---------------------------------------------------------------- gather / scatter
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table, int width,
                                                          const int64_t* __restrict__ idx,
                                                          int64_t idx_stride, int rows,
                                                          float* __restrict__ out) {
    const int w4 = width >> 2;
    const int64_t total = (int64_t)rows * w4;
    for (int64_t e = blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int r = (int)(e / w4), c = (int)(e % w4);
        const int64_t src = idx[r * idx_stride];
        reinterpret_cast<float4*>(out)[e] = reinterpret_cast<const float4*>(table + src * width)[c];
    }
}

__global__ __launch_bounds__(256) void scatter_add_rows_kernel(float* __restrict__ grad, int width,
                                                               const int64_t* __restrict__ idx,
                                                               int64_t idx_stride, int rows,
                                                               const float* __restrict__ in) {
    // one wave-instruction adds 256 contiguous bytes of one destination row
    const int64_t total = (int64_t)rows * width;
    for (int64_t e = blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int r = (int)(e / width), c = (int)(e % width);
        const int64_t dst = idx[r * idx_stride];
        atomicAdd(grad + dst * width + c, in[e]);
    }
}

// out[n] (+)= sum_r x[r, n]: workgroup (bx, by) sums 64 columns over its slice of the rows (4 waves
// stride it) and adds its partial to out[] with one float atomic per column.
// partials != null: the workgroup's partial goes to partials[blockIdx.y][col] instead, and colsum_finish adds
// the row blocks' partials in a fixed order (same bits on every run, which float atomics do not give).
// row_scale != null: out[n] = sum_r row_scale[r] * x[r, n].
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int64_t ldx, int rows,
                                                     int cols, int rows_per_block, float* __restrict__ out,
                                                     float* __restrict__ partials, const float* __restrict__ row_scale) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float s0 = 0.f, s1 = 0.f;
    if (col < cols) {
        int r = r0 + wave;
        if (row_scale) {
            for (; r + 4 < r1; r += 8) {
                s0 = fmaf(row_scale[r], x[(int64_t)r * ldx + col], s0);
                s1 = fmaf(row_scale[r + 4], x[(int64_t)(r + 4) * ldx + col], s1);
            }
            if (r < r1) s0 = fmaf(row_scale[r], x[(int64_t)r * ldx + col], s0);
        } else {
            for (; r + 4 < r1; r += 8) { s0 += x[(int64_t)r * ldx + col]; s1 += x[(int64_t)(r + 4) * ldx + col]; }
            if (r < r1) s0 += x[(int64_t)r * ldx + col];
        }
    }
    part[wave][lane] = s0 + s1;
    __syncthreads();
    if (wave == 0 && col < cols) {
        const float v = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        if (partials) partials[(int64_t)blockIdx.y * cols + col] = v;
        else atomicAdd(out + col, v);
    }
}

__global__ __launch_bounds__(256) void colsum_finish(const float* __restrict__ partials, int row_blocks, int cols,
                                                     float* __restrict__ out, int accumulate) {
    __shared__ float part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f;
    if (col < cols) {
        int b = wave;
        for (; b + 4 < row_blocks; b += 8) { s0 += partials[(int64_t)b * cols + col]; s1 += partials[(int64_t)(b + 4) * cols + col]; }
        if (b < row_blocks) s0 += partials[(int64_t)b * cols + col];
    }
    part[wave][lane] = s0 + s1;
    __syncthreads();
    if (wave == 0 && col < cols) {
        const float v = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        out[col] = accumulate ? out[col] + v : v;
    }
}

// ---------------------------------------------------------------- embedding gradient, reproducible
// grad_table[v, :] = sum over the rows r with idx[r] == v of rows_in[r, :], added in ascending r: the rows are
// first ordered by token (a counting sort whose every step is exact integer work: per-token counts, their
// exclusive scan, then each token's rows listed in ascending order), then one workgroup per token sums its
// rows.  The same bits on every run; tokens nobody used get zeros (the table is overwritten, not added to).
constexpr int EMB_CHUNK = 4096;   // tokens staged in LDS per pass of the scan over the rows
constexpr int EMB_SLICES = 8;     // the rows are cut into this many contiguous slices, scanned by different workgroups

// workgroup (x, y): tokens [256x, 256x + 256) against the rows of slice y.  FILL = false: slice_count[y][v] = how
// many of the slice's rows hold v.  FILL = true: slice_count[y][v] holds where v's rows of this slice start in
// `order` (emb_offsets put it there); they are listed from there on in ascending order.
template <bool FILL>
__global__ __launch_bounds__(256) void emb_rows_of_token(const int64_t* __restrict__ idx, int64_t idx_stride, int rows,
                                                         int vocab, int* __restrict__ slice_count,
                                                         int* __restrict__ order) {
    __shared__ __attribute__((aligned(16))) int tok[EMB_CHUNK];
    const int v = blockIdx.x * 256 + threadIdx.x;
    const int per_slice = ((rows + EMB_SLICES - 1) / EMB_SLICES + 15) / 16 * 16;
    const int row0 = blockIdx.y * per_slice, row1 = min(rows, row0 + per_slice);
    int* mine = slice_count + (int64_t)blockIdx.y * vocab;
    int found = 0;
    int next = (FILL && v < vocab) ? mine[v] : 0;
    for (int base = row0; base < row1; base += EMB_CHUNK) {
        const int n = min(EMB_CHUNK, row1 - base);
        for (int k = threadIdx.x; k < EMB_CHUNK; k += 256)
            tok[k] = k < n ? (int)idx[(int64_t)(base + k) * idx_stride] : -1;
        __syncthreads();
        if (v < vocab) {
            const int4* t4 = reinterpret_cast<const int4*>(tok);
            for (int k = 0; k < (n + 15) / 16; ++k) {   // the chunk is padded with -1 up to EMB_CHUNK
                int4 t[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) t[q] = t4[4 * k + q];   // every lane reads the same address: LDS broadcasts
                int hits = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) hits += (t[q].x == v) + (t[q].y == v) + (t[q].z == v) + (t[q].w == v);
                if (!FILL) {
                    found += hits;
                } else if (hits) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int r = base + 16 * k + 4 * q;
                        if (t[q].x == v) order[next++] = r;
                        if (t[q].y == v) order[next++] = r + 1;
                        if (t[q].z == v) order[next++] = r + 2;
                        if (t[q].w == v) order[next++] = r + 3;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (!FILL && v < vocab) mine[v] = found;
}

// count[v] = rows that hold v (all slices), offset[v] = sum of count[0 .. v), and slice_count[y][v] becomes where
// v's rows of slice y start: offset[v] + the counts of the slices before y  (one workgroup; chunks of 1024 tokens
// with a running carry)
__global__ __launch_bounds__(1024) void emb_offsets(int* __restrict__ slice_count, int vocab, int* __restrict__ count,
                                                    int* __restrict__ offset) {
    __shared__ int part[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < vocab; base += 1024) {
        const int v = base + threadIdx.x;
        int c = 0;
        if (v < vocab)
            for (int y = 0; y < EMB_SLICES; ++y) c += slice_count[(int64_t)y * vocab + v];
        part[threadIdx.x] = c;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {   // inclusive scan
            const int add = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
            __syncthreads();
            part[threadIdx.x] += add;
            __syncthreads();
        }
        if (v < vocab) {
            int at = carry + part[threadIdx.x] - c;
            count[v] = c;
            offset[v] = at;
            for (int y = 0; y < EMB_SLICES; ++y) {
                const int here = slice_count[(int64_t)y * vocab + v];
                slice_count[(int64_t)y * vocab + v] = at;
                at += here;
            }
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry += part[1023];
        __syncthreads();
    }
}

// Sum of rows_in[order[k]] over k in [lo, hi), added in that order; eight row loads in flight.
__device__ __forceinline__ float4 emb_sum_range(const float* __restrict__ rows_in, int width, const int* __restrict__ order,
                                                int lo, int hi, int c) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = lo;
    for (; k + 8 <= hi; k += 8) {
        int r[8];
        float4 a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = order[k + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = reinterpret_cast<const float4*>(rows_in + (int64_t)r[u] * width)[c];
#pragma unroll
        for (int u = 0; u < 8; ++u) { s.x += a[u].x; s.y += a[u].y; s.z += a[u].z; s.w += a[u].w; }
    }
    for (; k < hi; ++k) {
        const float4 a = reinterpret_cast<const float4*>(rows_in + (int64_t)order[k] * width)[c];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    return s;
}

// One workgroup per token: grad[v] = sum of the token's rows.  Short lists (the usual case) are summed in
// ascending row order by the first 128 lanes; a list of EMB_SPLIT_MIN rows or more (a caption's start or
// padding token holds a row of every image) is cut into four consecutive quarters, one per group of 128
// lanes, each summed in ascending order, and the quarters are added in order: a fixed order either way.
constexpr int EMB_SPLIT_MIN = 32;
__global__ __launch_bounds__(512) void emb_sum_rows(const float* __restrict__ rows_in, int width, const int* __restrict__ count,
                                                    const int* __restrict__ offset, const int* __restrict__ order,
                                                    float* __restrict__ grad) {
    __shared__ float4 part[3][128];
    const int v = blockIdx.x;
    const int n = count[v], first = offset[v];
    const int w4 = width >> 2;
    const int col = threadIdx.x & 127, g = threadIdx.x >> 7;
    if (n < EMB_SPLIT_MIN) {      // uniform over the workgroup
        if (g != 0) return;
        for (int c = col; c < w4; c += 128)
            reinterpret_cast<float4*>(grad + (int64_t)v * width)[c] = emb_sum_range(rows_in, width, order, first, first + n, c);
        return;
    }
    const int q = (n + 3) >> 2;
    const int lo = first + min(n, g * q), hi = first + min(n, (g + 1) * q);
    for (int base = 0; base < w4; base += 128) {
        const int c = base + col;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < w4) s = emb_sum_range(rows_in, width, order, lo, hi, c);
        if (g != 0) part[g - 1][col] = s;
        __syncthreads();
        if (g == 0 && c < w4) {
#pragma unroll
            for (int p = 0; p < 3; ++p) { const float4 a = part[p][col]; s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w; }
            reinterpret_cast<float4*>(grad + (int64_t)v * width)[c] = s;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- batched LSTM cell
__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(float* __restrict__ gates,
                                                            const float* __restrict__ c_prev,
                                                            float* __restrict__ c_out,
                                                            float* __restrict__ h_out, int n) {
    const int64_t total = (int64_t)n * (H / 4);
    for (int64_t e = blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int row = (int)(e / (H / 4)), j4 = (int)(e % (H / 4));
        float4* g = reinterpret_cast<float4*>(gates + (int64_t)row * 4 * H);
        float4 gi = g[j4], gf = g[H / 4 + j4], gg = g[2 * (H / 4) + j4], go = g[3 * (H / 4) + j4];
        const float4 cp = reinterpret_cast<const float4*>(c_prev)[e];
        float4 c, h;
#define ICRL_CELL(k)                                            \
    gi.k = sigmoidf_(gi.k); gf.k = sigmoidf_(gf.k);             \
    gg.k = tanhf_(gg.k);   go.k = sigmoidf_(go.k);              \
    c.k = gf.k * cp.k + gi.k * gg.k;                            \
    h.k = go.k * tanhf_(c.k);
        ICRL_CELL(x) ICRL_CELL(y) ICRL_CELL(z) ICRL_CELL(w)
#undef ICRL_CELL
        g[j4] = gi; g[H / 4 + j4] = gf; g[2 * (H / 4) + j4] = gg; g[3 * (H / 4) + j4] = go;
        reinterpret_cast<float4*>(c_out)[e] = c;
        reinterpret_cast<float4*>(h_out)[e] = h;
    }
}

// batched GRU cell (PyTorch gate order r, z, n): gx = x W_ih^T + b_ih, gh = h W_hh^T + b_hh, both [n, 3H]
__global__ __launch_bounds__(256) void gru_cell_fwd_kernel(const float* __restrict__ gx, const float* __restrict__ gh,
                                                           const float* __restrict__ h_prev,
                                                           float* __restrict__ h_out, int n) {
    const int64_t total = (int64_t)n * (H / 4);
    for (int64_t e = blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int row = (int)(e / (H / 4)), j4 = (int)(e % (H / 4));
        const float4* x = reinterpret_cast<const float4*>(gx + (int64_t)row * 3 * H);
        const float4* g = reinterpret_cast<const float4*>(gh + (int64_t)row * 3 * H);
        const float4 xr = x[j4], xz = x[H / 4 + j4], xn = x[2 * (H / 4) + j4];
        const float4 hr = g[j4], hz = g[H / 4 + j4], hn = g[2 * (H / 4) + j4];
        const float4 hp = reinterpret_cast<const float4*>(h_prev)[e];
        float4 h;
#define ICRL_GRU(k)                                             \
    {                                                           \
        const float r_ = sigmoidf_(xr.k + hr.k);                \
        const float z_ = sigmoidf_(xz.k + hz.k);                \
        const float n_ = tanhf_(xn.k + r_ * hn.k);              \
        h.k = n_ + z_ * (hp.k - n_);                            \
    }
        ICRL_GRU(x) ICRL_GRU(y) ICRL_GRU(z) ICRL_GRU(w)
#undef ICRL_GRU
        reinterpret_cast<float4*>(h_out)[e] = h;
    }
}

__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const float* __restrict__ gates,
                                                            const float* __restrict__ c_prev,
                                                            const float* __restrict__ c,
                                                            const float* __restrict__ dh,
                                                            const float* __restrict__ dc_in,
                                                            float* __restrict__ dgates,
                                                            float* __restrict__ dc_prev, int n) {
    const int64_t total = (int64_t)n * H;
    for (int64_t e = blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int row = (int)(e / H), j = (int)(e % H);
        const float* g = gates + (int64_t)row * 4 * H;
        const float i = g[j], f = g[H + j], gg = g[2 * H + j], o = g[3 * H + j];
        const float tc = tanhf_(c[e]);
        const float dhv = dh[e];
        float dcv = dhv * o * (1.f - tc * tc);
        if (dc_in) dcv += dc_in[e];
        float* d = dgates + (int64_t)row * 4 * H;
        d[j] = dcv * gg * i * (1.f - i);
        d[H + j] = dcv * c_prev[e] * f * (1.f - f);
        d[2 * H + j] = dcv * i * (1.f - gg * gg);
        d[3 * H + j] = dhv * tc * o * (1.f - o);
        dc_prev[e] = dcv * f;
    }
}

// ---------------------------------------------------------------- value head finish
// one wave per row: y = yh + u[row % N]; value = y.w2 + b2
__global__ __launch_bounds__(256) void value_head_kernel(float* __restrict__ yh, const float* __restrict__ u,
                                                         const float* __restrict__ w2,
                                                         const float* __restrict__ b2, int rows, int n,
                                                         float* __restrict__ values) {
    const int lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        float4* y = reinterpret_cast<float4*>(yh + (int64_t)row * 512);
        const float4* uu = reinterpret_cast<const float4*>(u + (int64_t)(row % n) * 512);
        const float4* w = reinterpret_cast<const float4*>(w2);
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float4 a = y[lane + 64 * k];
            const float4 b = uu[lane + 64 * k], ww = w[lane + 64 * k];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            y[lane + 64 * k] = a;
            s += a.x * ww.x + a.y * ww.y + a.z * ww.z + a.w * ww.w;
        }
        s = wave_sum(s);
        if (lane == 0) values[row] = s + b2[0];
    }
}

// ---------------------------------------------------------------- cosine reward
__global__ __launch_bounds__(256) void cosine_reward_kernel(const float* __restrict__ ve,
                                                            const float* __restrict__ se, int rows, int n,
                                                            float* __restrict__ reward) {
    const int lane = threadIdx.x & 63;
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        const float4* a = reinterpret_cast<const float4*>(ve + (int64_t)(row % n) * 512);
        const float4* b = reinterpret_cast<const float4*>(se + (int64_t)row * 512);
        float aa = 0.f, bb = 0.f, ab = 0.f;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float4 x = a[lane + 64 * k], y = b[lane + 64 * k];
            aa += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
            bb += y.x * y.x + y.y * y.y + y.z * y.z + y.w * y.w;
            ab += x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
        }
        aa = wave_sum(aa); bb = wave_sum(bb); ab = wave_sum(ab);
        if (lane == 0) {
            const float na = fmaxf(sqrtf(aa), 1e-12f), nb = fmaxf(sqrtf(bb), 1e-12f);
            reward[row] = ab / (na * nb);
        }
    }
}

// ---------------------------------------------------------------- A2C loss
// Single workgroup (count is N*(L-1) <= a few 10^4): double accumulation of the sums.
__global__ __launch_bounds__(1024) void a2c_loss_kernel(const float* __restrict__ values,
                                                        const float* __restrict__ rewards,
                                                        const float* __restrict__ logp, int count,
                                                        float* __restrict__ scalars,
                                                        float* __restrict__ d_value,
                                                        float* __restrict__ d_logp) {
    __shared__ double red[16][4];
    const float inv = 1.0f / (float)count;
    double s_act = 0., s_cri = 0., s_rew = 0., s_adv = 0.;
    for (int e = threadIdx.x; e < count; e += 1024) {
        const float v = values[e], r = rewards[e], lp = logp[e];
        const float adv = v - r;
        s_act += (double)(-lp * adv);
        s_cri += (double)(adv * adv);
        s_rew += (double)r;
        s_adv += (double)adv;
        if (d_value) d_value[e] = (adv - lp) * inv;
        if (d_logp) d_logp[e] = -adv * inv;
    }
    s_act = wave_sum_d(s_act); s_cri = wave_sum_d(s_cri);
    s_rew = wave_sum_d(s_rew); s_adv = wave_sum_d(s_adv);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave][0] = s_act; red[wave][1] = s_cri; red[wave][2] = s_rew; red[wave][3] = s_adv; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0., c = 0., r = 0., d = 0.;
        for (int w = 0; w < 16; ++w) { a += red[w][0]; c += red[w][1]; r += red[w][2]; d += red[w][3]; }
        const double actor = a / count, critic = 0.5 * c / count;
        scalars[0] = (float)(actor + critic);
        scalars[1] = (float)actor;
        scalars[2] = (float)critic;
        scalars[3] = (float)(r / count);
        scalars[4] = (float)(d / count);
    }
}

// ---------------------------------------------------------------- pre-training losses
// Length-weighted teacher-forced cross entropy (trainers.py:238-242): loss = sum_i weight[i] *
// sum_{t < valid[i]} -logp[t, i]; d loss / d logp[t, i] = -weight[i] inside the mask, 0 outside.
__global__ __launch_bounds__(1024) void masked_nll_kernel(const float* __restrict__ logp, const int* __restrict__ valid,
                                                          const float* __restrict__ weight, int steps, int n,
                                                          float* __restrict__ loss, float* __restrict__ d_logp) {
    __shared__ double red[16];
    double acc = 0.;
    for (int e = threadIdx.x; e < steps * n; e += 1024) {
        const int t = e / n, i = e % n;
        const bool in = t < valid[i];
        if (in) acc += (double)(-logp[e] * weight[i]);
        if (d_logp) d_logp[e] = in ? -weight[i] : 0.f;
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.;
        for (int w = 0; w < 16; ++w) a += red[w];
        loss[0] = (float)a;
    }
}

// nn.MSELoss (mean) of values against targets (trainers.py:155, 178): loss and d loss / d values
__global__ __launch_bounds__(1024) void mse_loss_kernel(const float* __restrict__ values, const float* __restrict__ targets,
                                                        int count, float* __restrict__ loss, float* __restrict__ d_value) {
    __shared__ double red[16];
    double acc = 0.;
    const float inv = 1.0f / (float)count;
    for (int e = threadIdx.x; e < count; e += 1024) {
        const float d = values[e] - targets[e];
        acc += (double)(d * d);
        if (d_value) d_value[e] = 2.f * d * inv;
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.;
        for (int w = 0; w < 16; ++w) a += red[w];
        loss[0] = (float)(a / count);
    }
}

// ---------------------------------------------------------------- visual-semantic embedding loss
// VisualSemanticEmbeddingLoss (trainers.py:21-54) from S = visuals . semantics^T [N, N], margin = beta / N:
//   loss = (sum_ij relu(S_ij - S_ii + m_ij) + sum_ij relu(S_ji - S_ii + m_ij)) / N,  m_ij = margin * (i != j).
// One workgroup per row i: the row's share of the loss and row i of dLoss/dS,
//   dS_ij = ([S_ij - S_ii + m > 0] + [S_ij - S_jj + m > 0]) / N                  (i != j)
//   dS_ii = -(sum_j [S_ij - S_ii + m > 0] + sum_j [S_ji - S_ii + m > 0]) / N     (j != i)
__global__ __launch_bounds__(256) void vse_rows_kernel(const float* __restrict__ S, int n, float margin,
                                                       float* __restrict__ row_loss, float* __restrict__ dS) {
    __shared__ double redl[4];
    __shared__ float redc[4];
    const int i = blockIdx.x;
    const float sii = S[(int64_t)i * n + i];
    const float inv = 1.0f / (float)n;
    double loss = 0.;
    float cnt = 0.f;
    for (int j = threadIdx.x; j < n; j += 256) {
        if (j == i) continue;
        const float sij = S[(int64_t)i * n + j], sji = S[(int64_t)j * n + i], sjj = S[(int64_t)j * n + j];
        const float a = sij - sii + margin;     // image i against caption j
        const float b = sji - sii + margin;     // caption i against image j
        const float a2 = sij - sjj + margin;    // caption j against image i (its term lives in row j's sum)
        loss += (double)fmaxf(a, 0.f) + (double)fmaxf(b, 0.f);
        cnt += (a > 0.f ? 1.f : 0.f) + (b > 0.f ? 1.f : 0.f);
        if (dS) dS[(int64_t)i * n + j] = ((a > 0.f ? 1.f : 0.f) + (a2 > 0.f ? 1.f : 0.f)) * inv;
    }
    loss = wave_sum_d(loss);
    cnt = wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) { redl[threadIdx.x >> 6] = loss; redc[threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        row_loss[i] = (float)((redl[0] + redl[1]) + (redl[2] + redl[3]));
        if (dS) dS[(int64_t)i * n + i] = -((redc[0] + redc[1]) + (redc[2] + redc[3])) * inv;
    }
}

__global__ __launch_bounds__(1024) void vse_sum_kernel(const float* __restrict__ row_loss, int n, float* __restrict__ loss) {
    __shared__ double red[16];
    double acc = 0.;
    for (int e = threadIdx.x; e < n; e += 1024) acc += (double)row_loss[e];
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.;
        for (int w = 0; w < 16; ++w) a += red[w];
        loss[0] = (float)(a / n);
    }
}

// g[r,k] = coef[r] * (onehot(token[r])[k] - p[r,k]); g may alias p
__global__ __launch_bounds__(256) void policy_dlogits_kernel(const float* p, float* g, int rows, int V,
                                                             const int64_t* __restrict__ tokens,
                                                             int64_t token_stride,
                                                             const float* __restrict__ coef) {
    const int64_t total = (int64_t)rows * V;
    for (int64_t e = blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int r = (int)(e / V), k = (int)(e % V);
        const float c = coef ? coef[r] : 1.f;
        const float hot = (tokens[r * token_stride] == (int64_t)k) ? 1.f : 0.f;
        g[e] = c * (hot - p[e]);
    }
}

// out[r, :] = coef[r] * x[r, :]  (cols % 4 == 0, 16-byte aligned rows)
__global__ __launch_bounds__(256) void scale_rows_kernel(const float* __restrict__ x, int64_t ldx,
                                                         const float* __restrict__ coef, float* __restrict__ out,
                                                         int64_t ldo, int rows, int c4) {
    const int64_t total = (int64_t)rows * c4;
    for (int64_t e = blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int r = (int)(e / c4), c = (int)(e - (int64_t)r * c4);
        const float k = coef[r];
        float4 v = reinterpret_cast<const float4*>(x + (int64_t)r * ldx)[c];
        v.x *= k; v.y *= k; v.z *= k; v.w *= k;
        reinterpret_cast<float4*>(out + (int64_t)r * ldo)[c] = v;
    }
}

// ---------------------------------------------------------------- Adam
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                   float lr, float b1, float b2, float omb1, float omb2,
                                                   float eps, float bc1, float bc2_sqrt) {
    // torch.optim.Adam: denom = sqrt(v)/sqrt(1-b2^t) + eps; p -= (lr/(1-b1^t)) * m/denom
    const float step_size = lr / bc1;
    for (int64_t e = blockIdx.x * 256 + threadIdx.x; e < n; e += (int64_t)gridDim.x * 256) {
        const float gv = g[e];
        // omb = 1 - beta rounded from double, as torch passes it (1.f - 0.999f is off by 1.3e-5 relative)
        const float mv = b1 * m[e] + omb1 * gv;
        const float vv = b2 * v[e] + omb2 * gv * gv;
        m[e] = mv; v[e] = vv;
        p[e] -= step_size * mv / (sqrtf(vv) / bc2_sqrt + eps);
    }
}

// out[i] = sum_z slabs[z*stride + i]
__global__ __launch_bounds__(256) void sum_slabs_kernel(const float* __restrict__ slabs, int64_t stride, int splits,
                                                        int64_t count4, float4* __restrict__ out, int accumulate) {
    for (int64_t e = blockIdx.x * 256 + threadIdx.x; e < count4; e += (int64_t)gridDim.x * 256) {
        float4 s = reinterpret_cast<const float4*>(slabs)[e];
        for (int z = 1; z < splits; ++z) {
            const float4 v = reinterpret_cast<const float4*>(slabs + z * stride)[e];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        if (accumulate) {
            const float4 o = out[e];
            s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
        }
        out[e] = s;
    }
}

// ---------------------------------------------------------------- chain helpers
// out[(t*N+i), :] = sum over the calls s that hold column t of dgates[position of (s, t, i), :].  The chain is
// calls 0 .. fc-1 in full (call s: columns 0 .. p0+s-1, its positions starting at column offset
// s*p0 + s(s-1)/2), then column p0+s-1 alone for each later call s.  One workgroup row per output row
// (so call and column are uniform and cost scalar arithmetic only), four loads in flight per lane.
__global__ __launch_bounds__(256) void chain_fold_kernel(const float* __restrict__ dg, int n, int p0, int n_calls, int fc,
                                                         int w4, float4* __restrict__ out) {
    const int c = blockIdx.y * 256 + threadIdx.x;
    if (c >= w4) return;
    const int64_t rowi = blockIdx.x;   // t*N + i
    const int t = (int)(rowi / n);
    const int64_t in_col = rowi - (int64_t)t * n;
    const int c0 = fc * p0 + fc * (fc - 1) / 2;
    const float4* dg4 = reinterpret_cast<const float4*>(dg) + c;
    auto col_of = [&](int call) { return (int64_t)call * p0 + (int64_t)call * (call - 1) / 2 + t; };
    auto ld = [&](int64_t col) { return dg4[(col * n + in_col) * w4]; };
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    const int first = max(0, t - p0 + 1);   // the first call that holds column t
    float4 s = zero;
    if (first >= fc) {                      // a call past the full ones: its single column
        if (first < n_calls) s = ld((int64_t)c0 + first - fc);
    } else {
        for (int call = first; call < fc; call += 4) {
            const float4 a = ld(col_of(call));
            const float4 b = call + 1 < fc ? ld(col_of(call + 1)) : zero;
            const float4 d = call + 2 < fc ? ld(col_of(call + 2)) : zero;
            const float4 e = call + 3 < fc ? ld(col_of(call + 3)) : zero;
            s.x += (a.x + b.x) + (d.x + e.x); s.y += (a.y + b.y) + (d.y + e.y);
            s.z += (a.z + b.z) + (d.z + e.z); s.w += (a.w + b.w) + (d.w + e.w);
        }
    }
    out[rowi * w4 + c] = s;
}

// h_prev[pos,:] = saved[pos-stride, h_slot] (zeros for pos < stride); `slots` x 512 floats per position
__global__ __launch_bounds__(256) void chain_hprev_kernel(const float* __restrict__ saved, int64_t positions,
                                                          int slots, int h_slot, int64_t stride,
                                                          float4* __restrict__ h_prev) {
    const int64_t total = positions * (H / 4);
    for (int64_t e = blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t pos = e / (H / 4);
        const int c = (int)(e % (H / 4));
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pos >= stride) v = reinterpret_cast<const float4*>(saved + ((pos - stride) * slots + h_slot) * H)[c];
        h_prev[e] = v;
    }
}

}  // namespace

using icrl::as_stream;

extern "C" int icrl_sum_slabs(const float* slabs, int64_t slab_stride, int splits, int64_t count, float* out,
                              int accumulate, icrl_stream_t stream) {
    ICRL_REQUIRE(slabs && out && splits >= 1 && count > 0 && count % 4 == 0 && slab_stride % 4 == 0,
                 "icrl_sum_slabs: bad argument");
    hipLaunchKernelGGL(sum_slabs_kernel, dim3(grid_for(count / 4, 256)), dim3(256), 0, as_stream(stream), slabs,
                       slab_stride, splits, count / 4, reinterpret_cast<float4*>(out), accumulate);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_gather_rows(const float* table, int width, const int64_t* idx, int64_t idx_stride,
                                int rows, float* out, icrl_stream_t stream) {
    ICRL_REQUIRE(table && idx && out && rows > 0, "icrl_gather_rows: null pointer or no rows");
    ICRL_REQUIRE(width % 4 == 0 && reinterpret_cast<uintptr_t>(table) % 16 == 0 &&
                 reinterpret_cast<uintptr_t>(out) % 16 == 0, "icrl_gather_rows: width %% 4 and 16-byte alignment required");
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((int64_t)rows * width / 4, 256)), dim3(256), 0,
                       as_stream(stream), table, width, idx, idx_stride, rows, out);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_scatter_add_rows(float* grad_table, int width, const int64_t* idx, int64_t idx_stride,
                                     int rows, const float* rows_in, icrl_stream_t stream) {
    ICRL_REQUIRE(grad_table && idx && rows_in && rows > 0, "icrl_scatter_add_rows: null pointer or no rows");
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(grid_for((int64_t)rows * width, 256)), dim3(256), 0,
                       as_stream(stream), grad_table, width, idx, idx_stride, rows, rows_in);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int64_t icrl_colsum_scratch_floats(int rows, int cols) {
    if (rows <= 0 || cols <= 0) return 0;
    const int col_blocks = (cols + 63) / 64;
    int row_blocks = std::max(1, std::min(2048 / col_blocks, (rows + 63) / 64));
    return (int64_t)row_blocks * cols;
}

extern "C" int icrl_colsum(const float* x, int64_t ldx, int rows, int cols, float* out, int accumulate,
                           const float* row_scale, float* scratch, icrl_stream_t stream) {
    ICRL_REQUIRE(x && out && rows > 0 && cols > 0, "icrl_colsum: null pointer or empty shape");
    hipStream_t st = as_stream(stream);
    const int col_blocks = (cols + 63) / 64;
    const int row_blocks = std::max(1, std::min(2048 / col_blocks, (rows + 63) / 64));   // aim at ~2048 workgroups
    const int rows_per_block = (rows + row_blocks - 1) / row_blocks;
    if (!scratch && !accumulate) ICRL_HIP(hipMemsetAsync(out, 0, sizeof(float) * cols, st));
    hipLaunchKernelGGL(colsum_kernel, dim3(col_blocks, row_blocks), dim3(256), 0, st, x, ldx, rows, cols,
                       rows_per_block, out, scratch, row_scale);
    if (scratch)
        hipLaunchKernelGGL(colsum_finish, dim3((cols + 63) / 64), dim3(256), 0, st, scratch, row_blocks, cols, out,
                           accumulate);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int64_t icrl_embedding_grad_work_ints(int vocab, int rows) {
    return vocab > 0 && rows > 0 ? (2 + EMB_SLICES) * (int64_t)vocab + rows : 0;
}

extern "C" int icrl_token_order(const int64_t* idx, int64_t idx_stride, int rows, int vocab, int32_t* work,
                                icrl_stream_t stream) {
    ICRL_REQUIRE(idx && work && rows > 0 && vocab > 0, "icrl_token_order: null pointer or empty shape");
    hipStream_t st = as_stream(stream);
    // work: count[vocab], offset[vocab], order[rows], then the per-slice counts / starts [EMB_SLICES][vocab]
    int* count = work;
    int* offset = work + vocab;
    int* order = work + 2 * (int64_t)vocab;
    int* slice_count = order + rows;
    const dim3 grid((vocab + 255) / 256, EMB_SLICES);
    hipLaunchKernelGGL(emb_rows_of_token<false>, grid, dim3(256), 0, st, idx, idx_stride, rows, vocab, slice_count,
                       (int*)nullptr);
    hipLaunchKernelGGL(emb_offsets, dim3(1), dim3(1024), 0, st, slice_count, vocab, count, offset);
    hipLaunchKernelGGL(emb_rows_of_token<true>, grid, dim3(256), 0, st, idx, idx_stride, rows, vocab, slice_count, order);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_embedding_grad(float* grad_table, int vocab, int width, const float* rows_in, const int32_t* work,
                                   icrl_stream_t stream) {
    ICRL_REQUIRE(grad_table && rows_in && work && vocab > 0, "icrl_embedding_grad: null pointer or empty shape");
    ICRL_REQUIRE(width % 4 == 0 && reinterpret_cast<uintptr_t>(grad_table) % 16 == 0 &&
                 reinterpret_cast<uintptr_t>(rows_in) % 16 == 0, "icrl_embedding_grad: width %% 4 and 16-byte alignment required");
    hipLaunchKernelGGL(emb_sum_rows, dim3(vocab), dim3(512), 0, as_stream(stream), rows_in, width, work, work + vocab,
                       work + 2 * (int64_t)vocab, grad_table);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_lstm_cell_fwd(float* gates, const float* c_prev, float* c, float* h, int N,
                                  icrl_stream_t stream) {
    ICRL_REQUIRE(gates && c_prev && c && h && N > 0, "icrl_lstm_cell_fwd: null pointer or N <= 0");
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, dim3(grid_for((int64_t)N * H / 4, 256)), dim3(256), 0,
                       as_stream(stream), gates, c_prev, c, h, N);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_gru_cell_fwd(const float* gx, const float* gh, const float* h_prev, float* h, int N,
                                 icrl_stream_t stream) {
    ICRL_REQUIRE(gx && gh && h_prev && h && N > 0, "icrl_gru_cell_fwd: null pointer or N <= 0");
    hipLaunchKernelGGL(gru_cell_fwd_kernel, dim3(grid_for((int64_t)N * H / 4, 256)), dim3(256), 0,
                       as_stream(stream), gx, gh, h_prev, h, N);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_lstm_cell_bwd(const float* gates_act, const float* c_prev, const float* c,
                                  const float* dh, const float* dc, float* dgates, float* dc_prev, int N,
                                  icrl_stream_t stream) {
    ICRL_REQUIRE(gates_act && c_prev && c && dh && dgates && dc_prev && N > 0,
                 "icrl_lstm_cell_bwd: null pointer or N <= 0");
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, dim3(grid_for((int64_t)N * H, 256)), dim3(256), 0,
                       as_stream(stream), gates_act, c_prev, c, dh, dc, dgates, dc_prev, N);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_value_head_fwd(float* yh_inout, const float* u, const float* w2, const float* b2,
                                   int rows, int N, float* values, icrl_stream_t stream) {
    ICRL_REQUIRE(yh_inout && u && w2 && b2 && values && rows > 0 && N > 0,
                 "icrl_value_head_fwd: null pointer or empty shape");
    hipLaunchKernelGGL(value_head_kernel, dim3(grid_for(rows, 4)), dim3(256), 0, as_stream(stream), yh_inout,
                       u, w2, b2, rows, N, values);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_cosine_reward(const float* ve, const float* se, int rows, int N, float* reward,
                                  icrl_stream_t stream) {
    ICRL_REQUIRE(ve && se && reward && rows > 0 && N > 0, "icrl_cosine_reward: null pointer or empty shape");
    hipLaunchKernelGGL(cosine_reward_kernel, dim3(grid_for(rows, 4)), dim3(256), 0, as_stream(stream), ve, se,
                       rows, N, reward);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_a2c_loss(const float* values, const float* rewards, const float* log_probs, int count,
                             float* scalars, float* d_value, float* d_logp, icrl_stream_t stream) {
    ICRL_REQUIRE(values && rewards && log_probs && scalars && count > 0,
                 "icrl_a2c_loss: null pointer or count <= 0");
    hipLaunchKernelGGL(a2c_loss_kernel, dim3(1), dim3(1024), 0, as_stream(stream), values, rewards, log_probs,
                       count, scalars, d_value, d_logp);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_masked_nll(const float* log_probs, const int* valid_steps, const float* row_weight, int steps, int N,
                               float* loss, float* d_logp, icrl_stream_t stream) {
    ICRL_REQUIRE(log_probs && valid_steps && row_weight && loss && steps > 0 && N > 0,
                 "icrl_masked_nll: null pointer or empty shape");
    hipLaunchKernelGGL(masked_nll_kernel, dim3(1), dim3(1024), 0, as_stream(stream), log_probs, valid_steps, row_weight,
                       steps, N, loss, d_logp);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_mse_loss(const float* values, const float* targets, int count, float* loss, float* d_value,
                             icrl_stream_t stream) {
    ICRL_REQUIRE(values && targets && loss && count > 0, "icrl_mse_loss: null pointer or count <= 0");
    hipLaunchKernelGGL(mse_loss_kernel, dim3(1), dim3(1024), 0, as_stream(stream), values, targets, count, loss,
                       d_value);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_vse_loss(const float* S, int N, float beta, float* row_scratch, float* loss, float* dS,
                             icrl_stream_t stream) {
    ICRL_REQUIRE(S && row_scratch && loss && N > 0, "icrl_vse_loss: null pointer or N <= 0");
    hipLaunchKernelGGL(vse_rows_kernel, dim3(N), dim3(256), 0, as_stream(stream), S, N, beta / (float)N, row_scratch, dS);
    hipLaunchKernelGGL(vse_sum_kernel, dim3(1), dim3(1024), 0, as_stream(stream), row_scratch, N, loss);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_policy_dlogits(const float* probs, float* dlogits, int rows, int V, const int64_t* tokens,
                                   int64_t token_stride, const float* coef, icrl_stream_t stream) {
    ICRL_REQUIRE(probs && dlogits && tokens && rows > 0 && V > 0,
                 "icrl_policy_dlogits: null pointer or empty shape");
    hipLaunchKernelGGL(policy_dlogits_kernel, dim3(grid_for((int64_t)rows * V, 256)), dim3(256), 0,
                       as_stream(stream), probs, dlogits, rows, V, tokens, token_stride, coef);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_scale_rows(const float* x, int64_t ldx, const float* coef, float* out, int64_t ldo, int rows,
                               int cols, icrl_stream_t stream) {
    ICRL_REQUIRE(x && coef && out && rows > 0 && cols > 0, "icrl_scale_rows: null pointer or empty shape");
    ICRL_REQUIRE(cols % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 &&
                 reinterpret_cast<uintptr_t>(out) % 16 == 0, "icrl_scale_rows: rows must be 16-byte aligned, cols %% 4 == 0");
    hipLaunchKernelGGL(scale_rows_kernel, dim3(grid_for((int64_t)rows * cols / 4, 256)), dim3(256), 0, as_stream(stream),
                       x, ldx, coef, out, ldo, rows, cols / 4);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                              double lr, double beta1, double beta2, double eps, int step,
                              icrl_stream_t stream) {
    ICRL_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step > 0,
                 "icrl_adam_step: null pointer, n <= 0 or step <= 0");
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n, 256)), dim3(256), 0, as_stream(stream), param, grad,
                       exp_avg, exp_avg_sq, n, (float)lr, (float)beta1, (float)beta2, (float)(1.0 - beta1),
                       (float)(1.0 - beta2), (float)eps, (float)bc1, (float)sqrt(bc2));
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_chain_fold_calls(const float* dgates, int N, int p0, int n_calls, int full_calls, int width,
                                     float* out, icrl_stream_t stream) {
    ICRL_REQUIRE(dgates && out && N > 0 && p0 > 0 && n_calls > 0 && width % 4 == 0 && full_calls >= 0 &&
                 full_calls <= n_calls, "icrl_chain_fold_calls: bad argument");
    const int64_t rows = (int64_t)(p0 + n_calls - 1) * N;
    ICRL_REQUIRE(rows < ((int64_t)1 << 31), "icrl_chain_fold_calls: too many rows");
    const int w4 = width / 4;
    hipLaunchKernelGGL(chain_fold_kernel, dim3((unsigned)rows, (w4 + 255) / 256), dim3(256), 0, as_stream(stream), dgates,
                       N, p0, n_calls, full_calls == 0 ? n_calls : full_calls, w4, reinterpret_cast<float4*>(out));
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_chain_hprev(const float* saved, int64_t positions, int gru, int64_t stride, float* h_prev,
                                icrl_stream_t stream) {
    ICRL_REQUIRE(saved && h_prev && positions > 0 && stride >= 1, "icrl_chain_hprev: bad argument");
    hipLaunchKernelGGL(chain_hprev_kernel, dim3(grid_for(positions * (H / 4), 256)), dim3(256), 0,
                       as_stream(stream), saved, positions, gru ? 5 : 6, gru ? 4 : 5, stride,
                       reinterpret_cast<float4*>(h_prev));
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// Power-of-two scales for icrl_gemm_f16x2s: per row and / or per column of x, the scale that brings the largest
// magnitude of that row / column into [2^14, 2^15).
namespace {

// bits: float bits of non-negative maxima (monotonic as unsigned integers), zeroed by the caller
__global__ __launch_bounds__(256) void absmax2d_kernel(const float* __restrict__ x, int64_t ld, int rows, int cols,
                                                       unsigned* __restrict__ rowbits, unsigned* __restrict__ colbits) {
    // workgroup: 64 rows x 256 columns; wave w takes rows w, w + 4, ...; lane l columns 4 l .. 4 l + 3 (one 16-byte load)
    __shared__ float4 red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 256 + 4 * lane;
    const int r_end = min(rows, (int)(blockIdx.y + 1) * 64);
    const bool vec = cols % 4 == 0 && ld % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0;
    float4 cm = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = blockIdx.y * 64 + wave; r < r_end; r += 4) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* p = x + (int64_t)r * ld + c;
        if (vec) {
            if (c < cols) v = *reinterpret_cast<const float4*>(p);
        } else {
            if (c + 0 < cols) v.x = p[0];
            if (c + 1 < cols) v.y = p[1];
            if (c + 2 < cols) v.z = p[2];
            if (c + 3 < cols) v.w = p[3];
        }
        v.x = fabsf(v.x); v.y = fabsf(v.y); v.z = fabsf(v.z); v.w = fabsf(v.w);
        cm.x = fmaxf(cm.x, v.x); cm.y = fmaxf(cm.y, v.y); cm.z = fmaxf(cm.z, v.z); cm.w = fmaxf(cm.w, v.w);
        if (rowbits) {
            const float rm = icrl::wave_max(fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)));
            if (lane == 0) atomicMax(rowbits + r, __float_as_uint(rm));
        }
    }
    if (!colbits) return;
    red[wave][lane] = cm;
    __syncthreads();
    if (wave == 0 && c < cols) {
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float4 o = red[w][lane];
            cm.x = fmaxf(cm.x, o.x); cm.y = fmaxf(cm.y, o.y); cm.z = fmaxf(cm.z, o.z); cm.w = fmaxf(cm.w, o.w);
        }
        atomicMax(colbits + c + 0, __float_as_uint(cm.x));
        if (c + 1 < cols) atomicMax(colbits + c + 1, __float_as_uint(cm.y));
        if (c + 2 < cols) atomicMax(colbits + c + 2, __float_as_uint(cm.z));
        if (c + 3 < cols) atomicMax(colbits + c + 3, __float_as_uint(cm.w));
    }
}

__device__ __forceinline__ float scale_for(float top) {
    int ex = 0;
    if (top > 0.f && top < INFINITY) frexpf(top, &ex);   // top = f 2^ex, f in [0.5, 1)
    ex = max(-60, min(60, ex));
    return ldexpf(1.f, 15 - ex);                          // top * scale in [2^14, 2^15)
}

__global__ void scales_from_bits_kernel(const unsigned* __restrict__ bits, int count, float* __restrict__ scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) scale[i] = scale_for(__uint_as_float(bits[i]));
}

// rows of G0 = onehot(token) - softmax: the largest magnitude of a row is its token entry, 1 - p_token (every other
// entry is a probability, and together they sum to 1 - p_token), known from the row's log-probability alone
__global__ void g0_row_scales_kernel(const float* __restrict__ log_prob, int count, float* __restrict__ scale) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) scale[i] = scale_for(fmaxf(-expm1f(log_prob[i]), 0x1p-22f) * 1.0001f);
}

}  // namespace

extern "C" int icrl_absmax_scales(const float* x, int64_t ld, int rows, int cols, float* row_scale, float* col_scale,
                                  unsigned* work, icrl_stream_t stream) {
    ICRL_REQUIRE(x && work && (row_scale || col_scale), "icrl_absmax_scales: null pointer");
    ICRL_REQUIRE(rows > 0 && cols > 0 && ld >= cols, "icrl_absmax_scales: %d x %d (ld %lld)", rows, cols, (long long)ld);
    hipStream_t st = icrl::as_stream(stream);
    unsigned* rowbits = row_scale ? work : nullptr;
    unsigned* colbits = col_scale ? work + (row_scale ? rows : 0) : nullptr;
    ICRL_HIP(hipMemsetAsync(work, 0, sizeof(unsigned) * ((row_scale ? rows : 0) + (col_scale ? cols : 0)), st));
    hipLaunchKernelGGL(absmax2d_kernel, dim3((cols + 255) / 256, (rows + 63) / 64), dim3(256), 0, st, x, ld, rows, cols, rowbits,
                       colbits);
    if (row_scale) hipLaunchKernelGGL(scales_from_bits_kernel, dim3((rows + 255) / 256), dim3(256), 0, st, rowbits, rows, row_scale);
    if (col_scale) hipLaunchKernelGGL(scales_from_bits_kernel, dim3((cols + 255) / 256), dim3(256), 0, st, colbits, cols, col_scale);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_g0_row_scales(const float* log_prob, int count, float* scale, icrl_stream_t stream) {
    ICRL_REQUIRE(log_prob && scale && count > 0, "icrl_g0_row_scales: null pointer or empty");
    hipLaunchKernelGGL(g0_row_scales_kernel, dim3((count + 255) / 256), dim3(256), 0, icrl::as_stream(stream), log_prob, count, scale);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}
