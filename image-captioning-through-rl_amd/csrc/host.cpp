// Host-side plumbing of libicrl_a2c.so: error text, ABI version, and a launch sequence that is cheaper to queue
// from here than launch by launch from a Python host.
#include <stdarg.h>
#include <stdio.h>

#include "icrl_common.h"

namespace icrl {
static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}
}  // namespace icrl

extern "C" int icrl_abi_version(void) { return ICRL_ABI_VERSION; }
extern "C" const char* icrl_last_error(void) { return icrl::g_error; }

// The policy LSTM's backward-through-time (autograd of models.py:80 over the rollout's T steps) as ONE call:
// per step k = T-1 .. 0 the cell's backward (icrl_lstm_cell_bwd), then what flows into the previous step's
// output, dH[k-1] += dgates[k] . W_hh (split over k into `splits` slabs, summed in a fixed order), and for
// k = 0 the gradient of the initial state, dh0 = dgates[0] . W_hh.  The same launches, in the same order and
// with the same plans, as a caller issuing them one by one - 3 T launches that cost a Python host several
// microseconds each to queue.
extern "C" int icrl_policy_lstm_bptt(const float* gates_act, const float* c_all, float* dH, const float* w_hh,
                                     float* dgates, float* dc0, float* dc1, float* dh0, float* scratch,
                                     int64_t scratch_floats, int N, int T, int splits, int tile, int bf16,
                                     icrl_stream_t stream) {
    constexpr int H = icrl::kHidden;
    ICRL_REQUIRE(gates_act && c_all && dH && w_hh && dgates && dc0 && dc1 && dh0, "icrl_policy_lstm_bptt: null pointer");
    ICRL_REQUIRE(N > 0 && T > 0 && splits >= 1 && splits <= 64, "icrl_policy_lstm_bptt: bad shape or splits");
    const int64_t nh = (int64_t)N * H;
    ICRL_REQUIRE(splits == 1 || (scratch && scratch_floats >= splits * nh),
                 "icrl_policy_lstm_bptt: scratch holds fewer than splits x N x 512 floats");
    float* dc[2] = {dc0, dc1};
    for (int k = T - 1; k >= 0; --k) {
        if (int rc = icrl_lstm_cell_bwd(gates_act + k * 4 * nh, c_all + k * nh, c_all + (k + 1) * nh, dH + k * nh,
                                        k == T - 1 ? nullptr : dc[(k + 1) & 1], dgates + k * 4 * nh, dc[k & 1], N, stream))
            return rc;
        float* out = k > 0 ? dH + (k - 1) * nh : dh0;
        const int accumulate = k > 0 ? 1 : 0;
        if (splits > 1) {
            const int rc = bf16 ? icrl_gemm_bf16_splitk(dgates + k * 4 * nh, 4 * H, 0, w_hh, H, 1, scratch, H, nh, N, H, 4 * H,
                                                        splits, tile, stream)
                                : icrl_gemm_f32_splitk(dgates + k * 4 * nh, 4 * H, 0, w_hh, H, 1, scratch, H, nh, N, H, 4 * H,
                                                       splits, tile, stream);
            if (rc) return rc;
            if (int rc2 = icrl_sum_slabs(scratch, nh, splits, nh, out, accumulate, stream)) return rc2;
        } else {
            const int rc = bf16 ? icrl_gemm_bf16(dgates + k * 4 * nh, 4 * H, 0, nullptr, w_hh, H, 1, out, H, N, H, 4 * H, nullptr,
                                                 nullptr, accumulate, tile, stream)
                                : icrl_gemm_f32(dgates + k * 4 * nh, 4 * H, 0, nullptr, w_hh, H, 1, out, H, N, H, 4 * H, nullptr,
                                                nullptr, accumulate, tile, stream);
            if (rc) return rc;
        }
    }
    return ICRL_OK;
}

extern "C" int icrl_policy_rollout_rounds(const void* w_pieces, const float* w_descale, const float* bias, void* h_pieces,
                                          const float* h_descale, float* ext, int64_t ld_ext, int N, int V, int pre_col,
                                          int rounds, int mode, const double* uniforms, int64_t* tokens, float* log_prob,
                                          float* probs, float* entropy, const float* table, float* gates,
                                          const float* c_state, float* h_state, icrl_stream_t stream) {
    constexpr int H = icrl::kHidden;
    ICRL_REQUIRE(rounds >= 0 && N > 0 && V > 0 && pre_col >= V && ld_ext >= pre_col + 4 * H,
                 "icrl_policy_rollout_rounds: bad shape (rounds %d, N %d, V %d, pre_col %d, ld_ext %lld)", rounds, N, V, pre_col,
                 (long long)ld_ext);
    const int64_t nh = (int64_t)N * H;
    for (int r = 0; r < rounds; ++r) {
        if (int rc = icrl_gemm_f16x2(h_pieces, h_descale, w_pieces, w_descale, ext, ld_ext, N, pre_col + 4 * H, H, bias, stream))
            return rc;
        if (int rc = icrl_sample_tokens_lstm_cell(ext, ld_ext, N, V, mode, uniforms ? uniforms + (int64_t)r * N : nullptr,
                                                  tokens + (int64_t)r * N, 1, log_prob + (int64_t)r * N, 1,
                                                  probs ? probs + (int64_t)r * N * V : nullptr,
                                                  entropy ? entropy + (int64_t)r * N : nullptr, ext + pre_col, ld_ext, table,
                                                  gates + 4 * r * nh, c_state + r * nh, const_cast<float*>(c_state) + (r + 1) * nh,
                                                  h_state + r * nh, h_pieces, stream))
            return rc;
    }
    return ICRL_OK;
}
