// Batched autoregressive sampler: one workgroup (256 threads, 512 for large vocabularies) per row of logits[N,V].
//
// Sampling reproduces numpy.random.choice(V, p=softmax(logits)) for an externally
// supplied uniform u: the probabilities are float32 (the reference copies a float32
// softmax to the host), the CDF is accumulated in float64 and divided by its last entry,
// and the token is the number of CDF entries <= u (searchsorted side="right").
// The row lives in LDS (V*4 bytes) between the passes; the float64 prefix sums are
// built from per-thread contiguous chunks + a wavefront scan (64-lane __shfl_up) +
// a 4-entry cross-wave scan.
#include "icrl_common.h"

namespace {
using namespace icrl;

constexpr int H = kHidden;

// Reductions over a workgroup of NW waves (red: NW floats of LDS)
template <int NW>
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float m = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = fmaxf(m, red[w]);
    return m;
}

// fixed order: pairs of neighbouring waves first, then the pair sums in ascending order
template <int NW>
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = red[0] + red[1];
#pragma unroll
    for (int w = 2; w < NW; w += 2) s += red[w] + red[w + 1];
    return s;
}

// The policy LSTM's next cell update, run by the workgroup that has just fixed its row's token (pre == nullptr:
// none).  pre[r] holds h . W_hh^T for the row's new state input (no bias), table[w] the input side of word w
// (W_ih emb[w] + b_ih + b_hh); gates_out receives the activated i, f, g, o.
struct NextCell {
    const float* pre;
    int64_t ld_pre;
    const float* table;
    float* gates_out;
    const float* c_prev;
    float* c_out;
    float* h_out;
    _Float16* h_pieces;   // optional: the new h once more, as the two scaled f16 pieces icrl_gemm_f16x2 takes (scale 2^12)
};

// DL: the `probs` output receives onehot(token) - softmax instead of the softmax (what the policy's backward pass
// needs, up to the per-row coefficient dLoss / dlog_prob): -p from every thread in pass 3, and the thread that
// wrote the token's entry corrects it to 1 - p once the token is fixed (same thread, same address: ordered).
template <int MODE, bool DL, int NT>
__global__ __launch_bounds__(NT) void sample_kernel(const float* __restrict__ logits, int64_t ld, int V,
                                                     const double* __restrict__ uniforms,
                                                     int64_t* __restrict__ tokens, int64_t token_stride,
                                                     float* __restrict__ log_prob, int64_t lp_stride,
                                                     float* __restrict__ probs, float* __restrict__ entropy,
                                                     NextCell cell) {
    extern __shared__ __attribute__((aligned(16))) float row[];  // V floats
    constexpr int NW = NT / 64;   // waves
    __shared__ float redf[NW];
    __shared__ double redd[NW];
    __shared__ int redi[NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = blockIdx.x;
    const float* src = logits + (int64_t)r * ld;

    if (MODE == ICRL_GREEDY) {
        // argmax, first index wins on ties (torch.argmax on a CPU/GPU tensor row)
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int k = tid; k < V; k += NT) {
            const float v = src[k];
            if (v > best) { best = v; bi = k; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) { redf[wave] = best; redi[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < NW; ++w)
                if (redf[w] > best || (redf[w] == best && redi[w] < bi)) { best = redf[w]; bi = redi[w]; }
            tokens[r * token_stride] = (int64_t)bi;
            if (log_prob) log_prob[r * lp_stride] = 0.f;
        }
        return;
    }

    // pass 1: row -> LDS, maximum (16-byte loads when the row allows, all of a thread's loads in flight)
    float m = -INFINITY;
    const bool vec = (V % 4 == 0) && (ld % 4 == 0) && (reinterpret_cast<uintptr_t>(logits) % 16 == 0) &&
                     (!probs || reinterpret_cast<uintptr_t>(probs) % 16 == 0);
    if (vec) {
        const int v4 = V / 4;
        const float4* src4 = reinterpret_cast<const float4*>(src);
        float4* row4 = reinterpret_cast<float4*>(row);
        for (int k0 = 0; k0 < v4; k0 += NT * 4) {
            float4 t[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + tid + NT * j;
                t[j] = (k < v4) ? src4[k] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + tid + NT * j;
                if (k < v4) row4[k] = t[j];
                m = fmaxf(fmaxf(m, fmaxf(t[j].x, t[j].y)), fmaxf(t[j].z, t[j].w));
            }
        }
    } else {
        for (int k = tid; k < V; k += NT) {
            const float v = src[k];
            row[k] = v;
            m = fmaxf(m, v);
        }
    }
    m = block_max<NW>(m, redf);
    // pass 2: exponentials, their float32 sum (element order of the sum: strided by the workgroup size, as below)
    float s = 0.f;
    for (int k = tid; k < V; k += NT) {
        const float e = expf(row[k] - m);
        row[k] = e;
        s += e;
    }
    s = block_sum<NW>(s, redf);
    // pass 3: probabilities (float32 division, as torch's softmax does), optional outputs
    float ent = 0.f;
    if (vec) {
        const int v4 = V / 4;
        float4* row4 = reinterpret_cast<float4*>(row);
        float4* out4 = probs ? reinterpret_cast<float4*>(probs + (int64_t)r * V) : nullptr;
        for (int k = tid; k < v4; k += NT) {
            float4 p = row4[k];
            p.x /= s; p.y /= s; p.z /= s; p.w /= s;
            row4[k] = p;
            if (out4) out4[k] = DL ? make_float4(0.f - p.x, 0.f - p.y, 0.f - p.z, 0.f - p.w) : p;
            if (entropy) {
                if (p.x > 0.f) ent -= p.x * logf(p.x);
                if (p.y > 0.f) ent -= p.y * logf(p.y);
                if (p.z > 0.f) ent -= p.z * logf(p.z);
                if (p.w > 0.f) ent -= p.w * logf(p.w);
            }
        }
    } else {
        for (int k = tid; k < V; k += NT) {
            const float p = row[k] / s;
            row[k] = p;
            if (probs) probs[(int64_t)r * V + k] = DL ? 0.f - p : p;
            if (entropy && p > 0.f) ent -= p * logf(p);
        }
    }
    if (entropy) {
        ent = block_sum<NW>(ent, redf);
        if (tid == 0) entropy[r] = ent;
    }
    __syncthreads();

    int token;
    if (MODE == ICRL_SAMPLE) {
        const int chunk = (V + NT - 1) / NT;
        const int k0 = tid * chunk, k1 = min(k0 + chunk, V);
        double local = 0.;
        for (int k = k0; k < k1; ++k) local += (double)row[k];
        // inclusive scan over the NT chunk sums
        double inc = local;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double up = __shfl_up(inc, o, 64);
            if (lane >= o) inc += up;
        }
        if (lane == 63) redd[wave] = inc;
        __syncthreads();
        double wave_off = 0., total = 0.;
        for (int w = 0; w < NW; ++w) {
            if (w < wave) wave_off += redd[w];
            total += redd[w];
        }
        double running = wave_off + inc - local;  // exclusive prefix of this chunk
        const double u = uniforms[r];
        // count CDF entries running/total <= u.  The float64 division is only needed when the entry is
        // within rounding distance of u: elsewhere the comparison against u*total gives the same answer
        // (a quotient's relative error is 2^-53, the guard band is 2^-40).
        const double ut = u * total, band = total * 0x1p-40;
        int cnt = 0;
        for (int k = k0; k < k1; ++k) {
            running += (double)row[k];
            const double d = running - ut;
            if (fabs(d) > band) cnt += (d < 0.) ? 1 : 0;
            else cnt += (running / total <= u) ? 1 : 0;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        if (lane == 0) redi[wave] = cnt;
        __syncthreads();
        token = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) token += redi[w];
        if (token > V - 1) token = V - 1;
        if (tid == 0) tokens[r * token_stride] = (int64_t)token;
    } else {
        token = (int)tokens[r * token_stride];
    }
    if (tid == 0 && log_prob) {
        const bool ok = token >= 0 && token < V;
        log_prob[r * lp_stride] = ok ? logf(row[token]) : NAN;
    }
    if (DL && probs && token >= 0 && token < V && tid == (vec ? (token >> 2) & (NT - 1) : token & (NT - 1)))
        probs[(int64_t)r * V + token] = 1.f - row[token];
    if (cell.pre != nullptr && token >= 0 && token < V) {
        // gates = h . W_hh^T + (W_ih x + b) for the token just fixed, then the cell (models.py:80): two units a lane
        const float* pr = cell.pre + (int64_t)r * cell.ld_pre;
        const float* tb = cell.table + (int64_t)token * (4 * H);
        float* go_ = cell.gates_out + (int64_t)r * (4 * H);
        for (int u = 2 * tid; u < H; u += 2 * NT) {
            float2 g4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float2 a = *reinterpret_cast<const float2*>(pr + q * H + u);
                const float2 b = *reinterpret_cast<const float2*>(tb + q * H + u);
                g4[q] = make_float2(a.x + b.x, a.y + b.y);
            }
            const float2 cp = *reinterpret_cast<const float2*>(cell.c_prev + (int64_t)r * H + u);
            float2 c, h;
#define ICRL_NEXT_CELL(k)                                                   \
    g4[0].k = sigmoidf_(g4[0].k); g4[1].k = sigmoidf_(g4[1].k);             \
    g4[2].k = tanhf_(g4[2].k);    g4[3].k = sigmoidf_(g4[3].k);             \
    c.k = g4[1].k * cp.k + g4[0].k * g4[2].k;                               \
    h.k = g4[3].k * tanhf_(c.k);
            ICRL_NEXT_CELL(x) ICRL_NEXT_CELL(y)
#undef ICRL_NEXT_CELL
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<float2*>(go_ + q * H + u) = g4[q];
            *reinterpret_cast<float2*>(cell.c_out + (int64_t)r * H + u) = c;
            *reinterpret_cast<float2*>(cell.h_out + (int64_t)r * H + u) = h;
            if (cell.h_pieces) {
                // fragment order of icrl_cut_f16x2: row block r / 32, k block u / 16, lane r % 32 + 32 (u % 16 / 8)
                typedef _Float16 hf16x2 __attribute__((ext_vector_type(2)));
                hf16x2 hi, mid;
                const float sx = h.x * 4096.f, sy = h.y * 4096.f;
                hi[0] = (_Float16)sx; hi[1] = (_Float16)sy;
                mid[0] = (_Float16)(sx - (float)hi[0]); mid[1] = (_Float16)(sy - (float)hi[1]);
                _Float16* dst = cell.h_pieces + ((((int64_t)(r >> 5) * (H / 16) + (u >> 4)) * 2) * 64 + (r & 31) + 32 * ((u >> 3) & 1)) * 8 + (u & 7);
                *reinterpret_cast<hf16x2*>(dst) = hi;
                *reinterpret_cast<hf16x2*>(dst + 64 * 8) = mid;
            }
        }
    }
}


// Top-k of every row, descending (torch.topk(logits, k) of trainers.py:96): the row is staged in LDS,
// then k rounds of a workgroup-wide argmax (ties: lowest index) that masks its winner.
__global__ __launch_bounds__(256) void topk_kernel(const float* __restrict__ logits, int64_t ld, int V, int k,
                                                   float* __restrict__ values, int64_t* __restrict__ indices) {
    extern __shared__ __attribute__((aligned(16))) float row[];  // V floats
    __shared__ float redf[4];
    __shared__ int redi[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = blockIdx.x;
    const float* src = logits + (int64_t)r * ld;
    for (int j = tid; j < V; j += 256) row[j] = src[j];
    __syncthreads();
    for (int round = 0; round < k; ++round) {
        float best = -INFINITY;
        int bi = 0x7fffffff;
        for (int j = tid; j < V; j += 256) {
            const float v = row[j];
            if (v > best) { best = v; bi = j; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) { redf[wave] = best; redi[wave] = bi; }
        __syncthreads();
        for (int w = 0; w < 4; ++w)
            if (redf[w] > best || (redf[w] == best && redi[w] < bi)) { best = redf[w]; bi = redi[w]; }
        if (tid == 0) {
            values[(int64_t)r * k + round] = best;
            indices[(int64_t)r * k + round] = (int64_t)(bi == 0x7fffffff ? 0 : bi);
            if (bi != 0x7fffffff) row[bi] = -INFINITY;
        }
        __syncthreads();
    }
}

}  // namespace

namespace {
int launch_sampler(const float* logits, int64_t ld_logits, int N, int V, int mode, const double* uniforms,
                   int64_t* tokens_out, int64_t token_stride, float* log_prob, int64_t log_prob_stride, float* probs,
                   float* entropy, const NextCell& cell, icrl_stream_t stream) {
    hipStream_t st = icrl::as_stream(stream);
    const size_t lds = (size_t)V * sizeof(float);
    // 512 threads a row from 4096 words on: the row's passes are chains of dependent reductions, and a CU holds
    // only as many rows as fit its LDS; twice the waves per row halve every thread's share of them (the A2C
    // step's rollout 2.28 -> 2.17 ms; 1024 threads gain nothing more).  Smaller vocabularies keep 256 threads -
    // and with them the summation order, so their results do not change.
    const bool wide = V >= 4096;
    dim3 grid(N), block(wide ? 512 : 256);
    const bool dl = (mode & ICRL_EMIT_DLOGITS) != 0;
    mode &= ~ICRL_EMIT_DLOGITS;
#define ICRL_SAMPLER_LAUNCH2(M, D, T, LDS)                                                                          \
    hipLaunchKernelGGL((sample_kernel<M, D, T>), grid, block, LDS, st, logits, ld_logits, V, uniforms, tokens_out,  \
                       token_stride, log_prob, log_prob_stride, probs, entropy, cell)
#define ICRL_SAMPLER_LAUNCH(M, D, LDS)                                                                              \
    if (wide) ICRL_SAMPLER_LAUNCH2(M, D, 512, LDS); else ICRL_SAMPLER_LAUNCH2(M, D, 256, LDS)
    switch (mode) {
        case ICRL_SAMPLE:
            if (dl) { ICRL_SAMPLER_LAUNCH(ICRL_SAMPLE, true, lds); } else { ICRL_SAMPLER_LAUNCH(ICRL_SAMPLE, false, lds); }
            break;
        case ICRL_GREEDY:
            if (dl) { icrl::set_error("icrl_sample_tokens: ICRL_EMIT_DLOGITS needs ICRL_SAMPLE or ICRL_FORCED"); return ICRL_ERR_ARG; }
            ICRL_SAMPLER_LAUNCH(ICRL_GREEDY, false, 0);
            break;
        case ICRL_FORCED:
            if (dl) { ICRL_SAMPLER_LAUNCH(ICRL_FORCED, true, lds); } else { ICRL_SAMPLER_LAUNCH(ICRL_FORCED, false, lds); }
            break;
        default:
            icrl::set_error("icrl_sample_tokens: unknown mode %d", mode);
            return ICRL_ERR_ARG;
    }
#undef ICRL_SAMPLER_LAUNCH2
#undef ICRL_SAMPLER_LAUNCH
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}
}  // namespace

extern "C" int icrl_sample_tokens(const float* logits, int64_t ld_logits, int N, int V, int mode,
                                  const double* uniforms, int64_t* tokens_out, int64_t token_stride,
                                  float* log_prob, int64_t log_prob_stride, float* probs, float* entropy,
                                  icrl_stream_t stream) {
    ICRL_REQUIRE(logits && tokens_out && N > 0 && V > 0, "icrl_sample_tokens: null pointer or empty shape");
    ICRL_REQUIRE(ld_logits >= V, "icrl_sample_tokens: ld_logits < V");
    ICRL_REQUIRE((int64_t)V * 4 <= 64 * 1024, "icrl_sample_tokens: vocabulary %d exceeds the 16384-entry LDS row", V);
    ICRL_REQUIRE((mode & ~ICRL_EMIT_DLOGITS) != ICRL_SAMPLE || uniforms, "icrl_sample_tokens: ICRL_SAMPLE needs uniforms");
    ICRL_REQUIRE(!(mode & ICRL_EMIT_DLOGITS) || probs, "icrl_sample_tokens: ICRL_EMIT_DLOGITS needs the probs output");
    return launch_sampler(logits, ld_logits, N, V, mode, uniforms, tokens_out, token_stride, log_prob, log_prob_stride,
                          probs, entropy, NextCell{}, stream);
}

extern "C" int icrl_sample_tokens_lstm_cell(const float* logits, int64_t ld_logits, int N, int V, int mode,
                                            const double* uniforms, int64_t* tokens_out, int64_t token_stride,
                                            float* log_prob, int64_t log_prob_stride, float* probs, float* entropy,
                                            const float* pre, int64_t ld_pre, const float* table, float* gates_out,
                                            const float* c_prev, float* c_out, float* h_out, void* h_pieces,
                                            icrl_stream_t stream) {
    ICRL_REQUIRE(logits && tokens_out && N > 0 && V > 0, "icrl_sample_tokens_lstm_cell: null pointer or empty shape");
    ICRL_REQUIRE(ld_logits >= V, "icrl_sample_tokens_lstm_cell: ld_logits < V");
    ICRL_REQUIRE((int64_t)V * 4 <= 64 * 1024, "icrl_sample_tokens_lstm_cell: vocabulary %d exceeds the 16384-entry LDS row", V);
    const int base_mode = mode & ~ICRL_EMIT_DLOGITS;
    ICRL_REQUIRE(base_mode == ICRL_SAMPLE || base_mode == ICRL_FORCED, "icrl_sample_tokens_lstm_cell: mode must be ICRL_SAMPLE or ICRL_FORCED");
    ICRL_REQUIRE(base_mode != ICRL_SAMPLE || uniforms, "icrl_sample_tokens_lstm_cell: ICRL_SAMPLE needs uniforms");
    ICRL_REQUIRE(!(mode & ICRL_EMIT_DLOGITS) || probs, "icrl_sample_tokens_lstm_cell: ICRL_EMIT_DLOGITS needs the probs output");
    ICRL_REQUIRE(pre && table && gates_out && c_prev && c_out && h_out, "icrl_sample_tokens_lstm_cell: null cell operand");
    ICRL_REQUIRE(ld_pre >= 2048 && ld_pre % 2 == 0 && reinterpret_cast<uintptr_t>(pre) % 8 == 0,
                 "icrl_sample_tokens_lstm_cell: pre must be 8-byte aligned with an even row stride >= 2048");
    ICRL_REQUIRE(reinterpret_cast<uintptr_t>(h_pieces) % 16 == 0, "icrl_sample_tokens_lstm_cell: h_pieces must be 16-byte aligned");
    NextCell cell{pre, ld_pre, table, gates_out, c_prev, c_out, h_out, static_cast<_Float16*>(h_pieces)};
    return launch_sampler(logits, ld_logits, N, V, mode, uniforms, tokens_out, token_stride, log_prob, log_prob_stride,
                          probs, entropy, cell, stream);
}

extern "C" int icrl_topk_rows(const float* logits, int64_t ld_logits, int N, int V, int k, float* values,
                              int64_t* indices, icrl_stream_t stream) {
    ICRL_REQUIRE(logits && values && indices && N > 0 && V > 0, "icrl_topk_rows: null pointer or empty shape");
    ICRL_REQUIRE(ld_logits >= V, "icrl_topk_rows: ld_logits < V");
    ICRL_REQUIRE(k >= 1 && k <= 64 && k <= V, "icrl_topk_rows: k must be in [1, min(64, V)] (got %d)", k);
    ICRL_REQUIRE((int64_t)V * 4 <= 64 * 1024, "icrl_topk_rows: vocabulary %d exceeds the 16384-entry LDS row", V);
    hipLaunchKernelGGL(topk_kernel, dim3(N), dim3(256), (size_t)V * sizeof(float), icrl::as_stream(stream), logits,
                       ld_logits, V, k, values, indices);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}
