// C = A . B^T (+ bias) at close to f32 accuracy on the f16 matrix instruction, for operands that are cut once and
// multiplied many times (a weight matrix over the 19 rounds of a rollout) or cut by the kernel that produces them.
//
// Arithmetic.  A row x of an operand is scaled by a power of two s (its own, so that its largest entry lands in
// [2^10, 2^11), or a fixed one for rows known to lie in [-1, 1]) and cut into two f16 pieces, x s = hi + mid + r with
// |r| <= 2^-22 |x s| wherever hi and mid are normal numbers (entries within 2^-13 of the row's largest; smaller ones
// keep an absolute error of 2^-25 / s).  An element product is hi.hi + hi.mid + mid.hi on v_mfma_f32_32x32x16_f16:
// the piece products are exact, the sums float32, and what is dropped (mid.mid and the r terms) is below 2^-21 of
// the product - the size of float32's own rounding of a few-hundred-term sum.  The scales are powers of two, so
// undoing them (a row factor times a column factor in the epilogue) is exact.  Three f16 instructions do the work of
// sixteen v_mfma_f32_32x32x2_f32.
//
// Layout.  The pieces are stored in the order the matrix instruction takes them: rows in blocks of 32, k in blocks
// of 16; block (rb, kb) is 2 x 64 x 8 halves, piece p of lane l = r % 32 + 32 * (k % 16 / 8) at
//     (((rb * K/16 + kb) * 2 + p) * 64 + l) * 8 + k % 8.
// A wave's fragment load is then 1 KB of consecutive memory, straight into the registers the instruction reads:
// no LDS, no conversion, no barrier in the product kernel.  Rows past the matrix' last one (up to the next multiple
// of 32) must hold zeros: the cutter never writes them, so a buffer zeroed once stays valid.
//
// Tiling.  256 threads as 2 x 2 waves, wave tile 64 x 96 (2 x 3 blocks, 96 accumulator registers), workgroup tile
// 128 x 192: the rollout's [512 x 512] . [512 x 12096] is 4 x 63 workgroups, one per CU.  Four k blocks of fragment
// loads are in flight per wave.  Workgroup b runs on XCD b % 8; the workgroups that share an operand tile run on one
// XCD, so that the tile crosses the fabric once.
#include "icrl_common.h"

namespace {

typedef _Float16 hf16x8 __attribute__((ext_vector_type(8)));
using f32x16 = __attribute__((ext_vector_type(16))) float;

struct CutArgs {
    const float* x; int64_t ld;
    int rows, K;
    float fixed_scale;        // > 0: every row scaled by this; 0: by its own power of two
    _Float16* pieces;
    float* descale;           // [rows] 1 / scale
};

__device__ __forceinline__ void cut2(float x, _Float16& hi, _Float16& mid) {
    hi = (_Float16)x;
    mid = (_Float16)(x - (float)hi);
}

// one wave per row
__global__ __launch_bounds__(256) void cut_f16x2_kernel(CutArgs a) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= a.rows) return;
    const float* row = a.x + (int64_t)r * a.ld;
    float scale = a.fixed_scale, descale = 1.f / a.fixed_scale;
    if (!(a.fixed_scale > 0.f)) {
        float top = 0.f;
        for (int k = 4 * lane; k < a.K; k += 256) {
            const float4 v = *reinterpret_cast<const float4*>(row + k);
            top = fmaxf(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))), top);
        }
        top = icrl::wave_max(top);
        int ex = 0;
        if (top > 0.f && top < INFINITY) frexpf(top, &ex);   // top = f 2^ex, f in [0.5, 1)
        ex = max(-100, min(100, ex));
        scale = ldexpf(1.f, 11 - ex);                          // top * scale in [2^10, 2^11)
        descale = ldexpf(1.f, ex - 11);
    }
    if (lane == 0) a.descale[r] = descale;
    const int KB = a.K / 16;
    _Float16* base = a.pieces + ((int64_t)(r / 32) * KB * 2 * 64 + (r % 32)) * 8;
    for (int o = lane; o < a.K / 8; o += 64) {   // octet o: k = 8 o .. 8 o + 7 = k block o / 2, half o % 2
        const float4 v0 = *reinterpret_cast<const float4*>(row + 8 * o);
        const float4 v1 = *reinterpret_cast<const float4*>(row + 8 * o + 4);
        const float x[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        hf16x8 hi, mid;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            _Float16 h_, m_;
            cut2(x[e] * scale, h_, m_);
            hi[e] = h_; mid[e] = m_;
        }
        _Float16* dst = base + ((int64_t)(o >> 1) * 2 * 64 + 32 * (o & 1)) * 8;
        *reinterpret_cast<hf16x8*>(dst) = hi;
        *reinterpret_cast<hf16x8*>(dst + 64 * 8) = mid;
    }
}

struct HpGemmArgs {
    const _Float16* A; const float* da;
    const _Float16* B; const float* db;
    const float* bias;
    float* C; int64_t ldc;
    int M, N, K;
};

template <int MI, int NJ>
__global__ __launch_bounds__(256, 1) void gemm_f16x2_kernel(HpGemmArgs g) {
    constexpr int TM = 64 * MI, TN = 64 * NJ, DEPTH = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int tiles_m = (g.M + TM - 1) / TM, tiles_n = (g.N + TN - 1) / TN;
    // Workgroup b runs on XCD b % 8.  Every XCD takes one contiguous run of tile ids, and ids advance fastest along
    // the dimension with fewer tiles: the few workgroups that share the other operand's tile (for the rollout, the
    // four row tiles that share 192 rows of the weights) run side by side on one XCD and fetch it once.
    const int per = (tiles_m * tiles_n + 7) / 8;
    const int id = (int)(blockIdx.x % 8) * per + (int)(blockIdx.x / 8);
    if ((int)(blockIdx.x / 8) >= per || id >= tiles_m * tiles_n) return;
    const bool m_fast = tiles_m <= tiles_n;
    const int mt = m_fast ? id % tiles_m : id / tiles_n, nt = m_fast ? id / tiles_m : id % tiles_n;
    const int KB = g.K / 16;
    const int MB = (g.M + 31) / 32, NB = (g.N + 31) / 32;
    const int mb0 = (mt * TM) / 32 + wr * MI, nb0 = (nt * TN) / 32 + wc * NJ;
    if (mb0 >= MB || nb0 >= NB) return;   // (no barrier in this kernel: a wave without a block of its own leaves)

    // blocks past the last one of a ragged tile repeat the last one; their products are not stored
    const _Float16* ap[MI];
    const _Float16* bp[NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i) ap[i] = g.A + ((int64_t)min(mb0 + i, MB - 1) * KB * 2 * 64 + lane) * 8;
#pragma unroll
    for (int j = 0; j < NJ; ++j) bp[j] = g.B + ((int64_t)min(nb0 + j, NB - 1) * KB * 2 * 64 + lane) * 8;

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    hf16x8 fa[DEPTH][MI][2], fb[DEPTH][NJ][2];
    auto load = [&](int d, int kb) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                fa[d][i][p] = *reinterpret_cast<const hf16x8*>(ap[i] + ((int64_t)kb * 2 + p) * 64 * 8);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int p = 0; p < 2; ++p)
                fb[d][j][p] = *reinterpret_cast<const hf16x8*>(bp[j] + ((int64_t)kb * 2 + p) * 64 * 8);
    };
    // No branch around a load: the waits before a stage's products then count exactly the loads issued after that
    // stage's (a load behind a condition makes every later wait a wait for all of them).  Stages past the last k
    // block load the last one again and are not multiplied.
    auto multiply = [&](int d) {
        // smallest terms first: hi.mid, mid.hi, hi.hi
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[d][j][1], fa[d][i][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[d][j][0], fa[d][i][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[d][j][0], fa[d][i][0], acc[i][j], 0, 0, 0);
            }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load(d, min(d, KB - 1));
    int kb0 = 0;
    for (; kb0 + DEPTH <= KB; kb0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            multiply(d);
            load(d, min(kb0 + d + DEPTH, KB - 1));
            __builtin_amdgcn_sched_barrier(0);   // (left alone the scheduler gathers the four stages' loads at the loop's end)
        }
    }
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d)
        if (kb0 + d < KB) multiply(d);

    // The B fragment goes in as the instruction's first operand, so the result block is transposed: lane (l31, h)
    // holds row m = l31 of the block and columns n = 8 q + 4 h + (0..3), q = r >> 2: four consecutive floats of C per
    // q, one 16-byte store (single-float stores, 96 per wave, were 5 us of a 20 us workgroup).  The scale factors are
    // read through clamped indices, all at once: behind a bounds test each of them is a branch and a wait of its own.
    const bool vec = g.ldc % 4 == 0 && g.N % 4 == 0 && reinterpret_cast<uintptr_t>(g.C) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(g.db) % 16 == 0 && (!g.bias || reinterpret_cast<uintptr_t>(g.bias) % 16 == 0);
    float da[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) da[i] = g.da[min((mb0 + i) * 32 + l31, g.M - 1)];
    if (vec) {
        float4 dbv[NJ][4], badd[NJ][4];
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = min((nb0 + j) * 32 + 8 * q + 4 * h, g.N - 4);
                dbv[j][q] = *reinterpret_cast<const float4*>(g.db + col);
                badd[j][q] = g.bias ? *reinterpret_cast<const float4*>(g.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = (mb0 + i) * 32 + l31;
            if (row >= g.M) continue;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = (nb0 + j) * 32 + 8 * q + 4 * h;
                    if (col >= g.N) continue;
                    float4 v;
                    v.x = acc[i][j][4 * q + 0] * (da[i] * dbv[j][q].x) + badd[j][q].x;
                    v.y = acc[i][j][4 * q + 1] * (da[i] * dbv[j][q].y) + badd[j][q].y;
                    v.z = acc[i][j][4 * q + 2] * (da[i] * dbv[j][q].z) + badd[j][q].z;
                    v.w = acc[i][j][4 * q + 3] * (da[i] * dbv[j][q].w) + badd[j][q].w;
                    *reinterpret_cast<float4*>(g.C + (int64_t)row * g.ldc + col) = v;
                }
        }
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int row = (mb0 + i) * 32 + l31;
            if (row >= g.M) continue;
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                for (int r = 0; r < 16; ++r) {
                    const int col = (nb0 + j) * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                    if (col < g.N)
                        g.C[(int64_t)row * g.ldc + col] = acc[i][j][r] * (da[i] * g.db[col]) + (g.bias ? g.bias[col] : 0.f);
                }
        }
    }
}

}  // namespace

extern "C" int64_t icrl_f16x2_piece_halves(int rows, int K) {
    if (rows <= 0 || K <= 0 || K % 16) return -1;
    return (int64_t)((rows + 31) / 32) * 32 * K * 2;
}

extern "C" int icrl_cut_f16x2(const float* x, int64_t ld, int rows, int K, float fixed_scale, void* pieces,
                              float* descale, icrl_stream_t stream) {
    ICRL_REQUIRE(x && pieces && descale, "icrl_cut_f16x2: null operand");
    ICRL_REQUIRE(rows > 0 && K > 0 && K % 16 == 0, "icrl_cut_f16x2: rows %d, K %d (K must be a multiple of 16)", rows, K);
    ICRL_REQUIRE(ld >= K && ld % 4 == 0 && reinterpret_cast<uintptr_t>(x) % 16 == 0 &&
                 reinterpret_cast<uintptr_t>(pieces) % 16 == 0,
                 "icrl_cut_f16x2: rows must be 16-byte aligned (ld %lld)", (long long)ld);
    ICRL_REQUIRE(fixed_scale >= 0.f && fixed_scale < 65536.f, "icrl_cut_f16x2: fixed_scale %g", (double)fixed_scale);
    CutArgs a;
    a.x = x; a.ld = ld; a.rows = rows; a.K = K; a.fixed_scale = fixed_scale;
    a.pieces = static_cast<_Float16*>(pieces); a.descale = descale;
    hipLaunchKernelGGL(cut_f16x2_kernel, dim3((rows + 3) / 4), dim3(256), 0, icrl::as_stream(stream), a);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

extern "C" int icrl_gemm_f16x2(const void* a_pieces, const float* a_descale, const void* b_pieces, const float* b_descale,
                               float* C, int64_t ldc, int M, int N, int K, const float* bias, icrl_stream_t stream) {
    ICRL_REQUIRE(a_pieces && a_descale && b_pieces && b_descale && C, "icrl_gemm_f16x2: null operand");
    ICRL_REQUIRE(M > 0 && N > 0 && K > 0 && K % 16 == 0, "icrl_gemm_f16x2: shape %dx%dx%d (K must be a multiple of 16)", M, N, K);
    ICRL_REQUIRE(ldc >= N, "icrl_gemm_f16x2: ldc %lld smaller than N %d", (long long)ldc, N);
    ICRL_REQUIRE(reinterpret_cast<uintptr_t>(a_pieces) % 16 == 0 && reinterpret_cast<uintptr_t>(b_pieces) % 16 == 0,
                 "icrl_gemm_f16x2: the piece buffers must be 16-byte aligned");
    HpGemmArgs g;
    g.A = static_cast<const _Float16*>(a_pieces); g.da = a_descale;
    g.B = static_cast<const _Float16*>(b_pieces); g.db = b_descale;
    g.bias = bias; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K;
    constexpr int MI = 2, NJ = 3;
    const int tiles_m = (M + 64 * MI - 1) / (64 * MI), tiles_n = (N + 64 * NJ - 1) / (64 * NJ);
    const dim3 grid(8 * ((tiles_m * tiles_n + 7) / 8));
    hipLaunchKernelGGL((gemm_f16x2_kernel<MI, NJ>), grid, dim3(256), 0, icrl::as_stream(stream), g);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}
