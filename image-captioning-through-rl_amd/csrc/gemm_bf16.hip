// GEMM with f32 operands in memory, bf16 products on v_mfma_f32_32x32x16_bf16 and f32 accumulation (gfx950).
//
// C[M,N] = (acc ? C : 0) + A[M,K]*B[K,N] + bias[n] + bias2[n], same calling convention, operand layouts, split-K
// slabs and XCD-aware tile order as the exact-f32 kernel of gemm_f32.hip.  The operands stay f32 in HBM (they
// are the networks' parameters and activations, which every other kernel reads as f32); a tile is rounded to
// bf16 (round to nearest even) on its way from registers to LDS, so LDS holds half the bytes and the matrix
// instruction runs at 16x the f32 rate.  Used for the policy network's products when the caller asks for
// bf16 arithmetic; the recurrent chains never use it.
//
// LDS image, both operands: [row][k] bf16 with k contiguous, rows padded to 72 elements (36 dwords: the
// ds_read_b128 of 16 rows at one k offset is conflict-free).  A k-contiguous source ("MK": [rows][ld]) is
// staged float4 -> 4 bf16 -> one 8-byte store; a row-contiguous source ("KM": [K][ld]) is transposed on the
// way: a thread owns one row and eight consecutive k (eight dword loads, each of them 64 consecutive rows per
// wave) and stores one 16-byte piece, so that the eight lanes of a store hit eight different bank groups
// (float4 loads along the rows would put the lanes of a store four rows apart: two banks for all of them).
// Fragment of lane (r = lane & 31, h = lane >> 5) for k-step s of 16: elements k = 16s + 8h + j, j = 0..7 of
// row r: one ds_read_b128.
#include "icrl_common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 64;
constexpr int LDK = BK + 8;   // bf16 elements per LDS row

struct GemmArgs {
    const float* A; int64_t lda; const int64_t* a_rows;
    const float* B; int64_t ldb;
    float* C; int64_t ldc;
    int M, N, K;
    const float* bias; const float* bias2;
    int accumulate;
    int kchunk;        // k-range per split (split-K); a multiple of BK
    int64_t slab;      // C offset (floats) between split-K partial results
};

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 p;
    p[0] = (__bf16)lo;
    p[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, p);
}

// Staging of one operand tile of ROWS x BK.  KM = false: piece i of thread tid is row (f >> 4), k quad (f & 15),
// f = tid + 256 i: one float4.  KM = true: piece i is row f % ROWS, k octet f / ROWS: eight floats.
template <int ROWS, bool KM>
struct Stage {
    static constexpr int PIECES = KM ? ROWS * BK / 8 / 256 : ROWS * BK / 4 / 256;
    float x[PIECES][KM ? 8 : 4];

    // EDGE = false: whole k stages, rows not gathered, every float4 aligned: plain loads, no branches (so that the
    // compiler can keep several stages' loads outstanding).  Rows past the matrix' last one read the last row again
    // (the epilogue does not store their products): a ragged last tile row or column stays on this path.
    template <bool EDGE>
    __device__ __forceinline__ void load(const float* base, int64_t ld, const int64_t* gather, int row0, int nrows,
                                         int k0, int kend, int tid) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int f = tid + 256 * i;
            if (!KM) {
                const int gr = EDGE ? row0 + (f >> 4) : min(row0 + (f >> 4), nrows - 1), gk = k0 + 4 * (f & 15);
                if (!EDGE) {
                    const float4 v = *reinterpret_cast<const float4*>(base + (int64_t)gr * ld + gk);
                    x[i][0] = v.x; x[i][1] = v.y; x[i][2] = v.z; x[i][3] = v.w;
                } else {
                    const int64_t src = (gr < nrows && gather) ? gather[gr] : (int64_t)gr;
                    const float* p = base + src * ld + gk;
#pragma unroll
                    for (int e = 0; e < 4; ++e) x[i][e] = (gr < nrows && gk + e < kend) ? p[e] : 0.f;
                }
            } else {
                const int gr = EDGE ? row0 + f % ROWS : min(row0 + f % ROWS, nrows - 1), gk = k0 + 8 * (f / ROWS);
                const float* p = base + (int64_t)gk * ld + gr;
#pragma unroll
                for (int e = 0; e < 8; ++e) x[i][e] = (!EDGE || (gr < nrows && gk + e < kend)) ? p[e * ld] : 0.f;
            }
        }
    }

    __device__ __forceinline__ void store(__bf16* lds, int tid) const {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int f = tid + 256 * i;
            if (!KM) {
                uint2 w;
                w.x = pack2(x[i][0], x[i][1]);
                w.y = pack2(x[i][2], x[i][3]);
                *reinterpret_cast<uint2*>(lds + (f >> 4) * LDK + 4 * (f & 15)) = w;
            } else {
                u32x4 w;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = pack2(x[i][2 * e], x[i][2 * e + 1]);
                *reinterpret_cast<u32x4*>(lds + (f % ROWS) * LDK + 8 * (f / ROWS)) = w;
            }
        }
    }
};

__device__ __forceinline__ bf16x8 frag(const __bf16* lds, int row, int kstep, int h) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(lds + row * LDK + 8 * (2 * kstep + h)));
}

// The k loop of one workgroup: acc += A[m0.., kbeg..kend) * B[kbeg..kend), n0..).
template <int BM, int BN, bool A_KM, bool B_KM, bool EDGE>
__device__ __forceinline__ void k_loop(const GemmArgs& g, f32x16 (&acc)[BM / 64][BN / 64], __bf16* lds, int m0, int n0,
                                       int kbeg, int kend, int nk, int tid) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int A_TILE = BM * LDK, B_TILE = BN * LDK;
    constexpr int DEPTH = (BM * BN <= 64 * 64) ? 4 : 2;   // stages in flight in registers (32 / 64 VGPRs each)
    // Operand tiles travel global -> registers -> LDS.  A workgroup's matrix instructions of one stage take a
    // fraction of a global load's latency, so DEPTH stages are kept in flight in registers: stage kt + DEPTH is
    // requested as soon as stage kt has left its registers for LDS.  Loads return in order, so waiting for the
    // oldest stage leaves the younger ones outstanding.  LDS holds two stages: the one being multiplied and
    // the one being written; one barrier per stage separates them.
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    Stage<BM, A_KM> sa[DEPTH];
    Stage<BN, B_KM> sb[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (d < nk) {
            sa[d].template load<EDGE>(g.A, g.lda, g.a_rows, m0, g.M, kbeg + d * BK, kend, tid);
            sb[d].template load<EDGE>(g.B, g.ldb, nullptr, n0, g.N, kbeg + d * BK, kend, tid);
        }

    for (int kt0 = 0; kt0 < nk; kt0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int kt = kt0 + d;
            if (kt < nk) {
                __bf16* la = lds + (kt & 1) * (A_TILE + B_TILE);
                __bf16* lb = la + A_TILE;
                sa[d].store(la, tid);
                sb[d].store(lb, tid);
                if (kt + DEPTH < nk) {
                    sa[d].template load<EDGE>(g.A, g.lda, g.a_rows, m0, g.M, kbeg + (kt + DEPTH) * BK, kend, tid);
                    sb[d].template load<EDGE>(g.B, g.ldb, nullptr, n0, g.N, kbeg + (kt + DEPTH) * BK, kend, tid);
                }
                __syncthreads();
#pragma unroll
                for (int ks = 0; ks < BK / 16; ++ks) {
                    bf16x8 av[TM], bv[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) av[i] = frag(la, wr * (BM / 2) + i * 32 + l31, ks, h);
#pragma unroll
                    for (int j = 0; j < TN; ++j) bv[j] = frag(lb, wc * (BN / 2) + j * 32 + l31, ks, h);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
                }
            }
        }
    }

}

template <int BM, int BN, bool A_KM, bool B_KM>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs g) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int A_TILE = BM * LDK, B_TILE = BN * LDK;
    __shared__ __attribute__((aligned(16))) __bf16 lds[2 * (A_TILE + B_TILE)];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;

    // XCD-aware tile order (as in gemm_f32.hip): workgroup b runs on XCD b % 8; every XCD gets one contiguous run of
    // (split, tile); tiles are walked in bands of 8 along the dimension of the larger operand, in 8 x 8 blocks.
    constexpr int BAND = 8;
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (g.N + BN - 1) / BN;
    const int tiles = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg / 8, r = nwg % 8, xcd = bid % 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
    }
    const int split = bid / tiles;
    const int id = bid - split * tiles;
    int mt, nt;
    {
        const bool m_fast = g.M < g.N;
        const int tf = m_fast ? tiles_m : tiles_n, ts = m_fast ? tiles_n : tiles_m;
        const int b = id / (tf * BAND), r = id - b * tf * BAND;
        const int w = min(BAND, ts - b * BAND);
        const int c = r / (BAND * w), r2 = r - c * BAND * w;
        const int hc = min(BAND, tf - c * BAND);
        const int s_in = r2 / hc, f_in = r2 - s_in * hc;
        const int f = c * BAND + f_in, sl = b * BAND + s_in;
        mt = m_fast ? f : sl;
        nt = m_fast ? sl : f;
    }
    const int m0 = mt * BM, n0 = nt * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int kbeg = split * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int nk = (kend - kbeg + BK - 1) / BK;
    float* const Cout = g.C + (int64_t)split * g.slab;

    const bool aligned = (reinterpret_cast<uintptr_t>(g.A) % 16 == 0) && (reinterpret_cast<uintptr_t>(g.B) % 16 == 0) &&
                         g.lda % 4 == 0 && g.ldb % 4 == 0 && kbeg % 4 == 0;
    // a tile inside the matrix walks its full stages on the branch-free path; a last, partial stage (k not a
    // multiple of BK) and the tiles on the matrix' edges take the bounds-checked one
    const bool interior = aligned && !g.a_rows;   // (ragged tile rows and columns included: Stage::load)
    const int full = (kend - kbeg) / BK;
    if (interior && full > 0) {
        k_loop<BM, BN, A_KM, B_KM, false>(g, acc, lds, m0, n0, kbeg, kbeg + full * BK, full, tid);
        if (full < nk) {
            __syncthreads();
            k_loop<BM, BN, A_KM, B_KM, true>(g, acc, lds, m0, n0, kbeg + full * BK, kend, nk - full, tid);
        }
    } else {
        k_loop<BM, BN, A_KM, B_KM, true>(g, acc, lds, m0, n0, kbeg, kend, nk, tid);
    }

    // C/D layout of the 32x32 MFMA: column = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wc * (BN / 2) + j * 32 + l31;
            if (col >= g.N) continue;
            float badd = 0.f;
            if (g.bias && split == 0) badd += g.bias[col];
            if (g.bias2 && split == 0) badd += g.bias2[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < g.M) {
                    float* dst = Cout + (int64_t)row * g.ldc + col;
                    float v = acc[i][j][r] + badd;
                    if (g.accumulate) v += *dst;
                    *dst = v;
                }
            }
        }
}

template <int BM, int BN>
int launch(const GemmArgs& g, int a_km, int b_km, int splits, hipStream_t st) {
    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    dim3 grid(tiles * splits), block(256);
    if (!a_km && !b_km) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, false, false>), grid, block, 0, st, g);
    else if (!a_km && b_km) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, false, true>), grid, block, 0, st, g);
    else if (a_km && !b_km) hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, true, false>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, true, true>), grid, block, 0, st, g);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

int dispatch(const float* A, int64_t lda, int a_kmajor, const int64_t* a_rows, const float* B, int64_t ldb,
             int b_kmajor, float* C, int64_t ldc, int M, int N, int K, const float* bias, const float* bias2,
             int accumulate, int splits, int64_t slab, int tile, hipStream_t st, const char* who) {
    ICRL_REQUIRE(A && B && C, "%s: null operand", who);
    ICRL_REQUIRE(M > 0 && N > 0 && K > 0, "%s: empty shape %dx%dx%d", who, M, N, K);
    ICRL_REQUIRE(!(a_rows && a_kmajor), "%s: row gather needs a_kmajor = 0", who);
    ICRL_REQUIRE(lda >= (a_kmajor ? M : K) && ldb >= (b_kmajor ? N : K) && ldc >= N,
                 "%s: leading dimension smaller than the row length", who);
    ICRL_REQUIRE(splits >= 1 && splits <= 64, "%s: splits must be in [1, 64]", who);
    ICRL_REQUIRE(tile == 0 || tile == 1 || tile == 3, "%s: tile must be 0 (by shape), 1 (128x128) or 3 (64x64)", who);
    GemmArgs g;
    g.A = A; g.lda = lda; g.a_rows = a_rows; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.bias = bias; g.bias2 = bias2; g.accumulate = accumulate;
    const int ksteps = (K + BK - 1) / BK;
    g.kchunk = ((ksteps + splits - 1) / splits) * BK;
    g.slab = slab;
    if (tile == 0) tile = ((int64_t)((M + 127) / 128) * ((N + 127) / 128) * splits >= 192) ? 1 : 3;
    if (tile == 1) return launch<128, 128>(g, a_kmajor, b_kmajor, splits, st);
    return launch<64, 64>(g, a_kmajor, b_kmajor, splits, st);
}

}  // namespace

extern "C" int icrl_gemm_bf16(const float* A, int64_t lda, int a_kmajor, const int64_t* a_rows,
                              const float* B, int64_t ldb, int b_kmajor, float* C, int64_t ldc,
                              int M, int N, int K, const float* bias, const float* bias2,
                              int accumulate, int tile, icrl_stream_t stream) {
    return dispatch(A, lda, a_kmajor, a_rows, B, ldb, b_kmajor, C, ldc, M, N, K, bias, bias2, accumulate, 1, 0, tile,
                    icrl::as_stream(stream), "icrl_gemm_bf16");
}

extern "C" int icrl_gemm_bf16_splitk(const float* A, int64_t lda, int a_kmajor, const float* B, int64_t ldb,
                                     int b_kmajor, float* C_slabs, int64_t ldc, int64_t slab_stride, int M, int N,
                                     int K, int splits, int tile, icrl_stream_t stream) {
    ICRL_REQUIRE(slab_stride >= (int64_t)(M - 1) * ldc + N, "icrl_gemm_bf16_splitk: slabs overlap");
    return dispatch(A, lda, a_kmajor, nullptr, B, ldb, b_kmajor, C_slabs, ldc, M, N, K, nullptr, nullptr, 0, splits,
                    slab_stride, tile, icrl::as_stream(stream), "icrl_gemm_bf16_splitk");
}
