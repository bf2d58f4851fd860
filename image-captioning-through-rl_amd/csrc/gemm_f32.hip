// f32 GEMM on the exact-f32 matrix instruction v_mfma_f32_32x32x2_f32 (gfx950).
//
// C[M,N] = (acc ? C : 0) + A[M,K]*B[K,N] + bias[n] + bias2[n].
// 256 threads = 4 waves in a 2x2 arrangement; each wave owns a (BM/2)x(BN/2) block of
// the BMxBN output tile as TMxTN accumulator tiles of 32x32.  K is walked in steps of
// 32 through double-buffered LDS; the next K-step's global loads are issued before the
// current step's MFMAs and written to the other LDS buffer after them (one barrier per
// K-step).
//
// Either operand may be stored with k contiguous ("MK": [rows][ld], an nn.Linear
// weight or an activation matrix) or with the row index contiguous ("KM": [K][ld], a
// transposed view).  MK tiles sit in LDS as [row][36] (pad 4: ds_read_b128 of 16
// different rows at one k offset is conflict-free), KM tiles as [k][rows].
// The MFMA sums over k in any order, so within a 16-k group lane-half h supplies
// k = 8h + j at MFMA step j for BOTH operands: an MK operand then reads its 8 values
// with two ds_read_b128, a KM operand with eight ds_read_b32 of 32 consecutive floats.
#include "icrl_common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BK = 32;
constexpr int MK_LD = 36;

struct GemmArgs {
    const float* A; int64_t lda; const int64_t* a_rows;
    const float* B; int64_t ldb;
    float* C; int64_t ldc;
    int M, N, K;
    const float* bias; const float* bias2;
    int accumulate;
    int a_vec, b_vec;  // 1 = 16-byte loads are aligned for this operand
    int kchunk;        // k-range per split (split-K); a multiple of BK
    int64_t slab;      // C offset (floats) between split-K partial results
};

// Load 4 consecutive floats p[0..3] of which the first `valid` exist.
__device__ __forceinline__ float4 load4(const float* p, int valid, bool vec) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid >= 4 && vec) {
        v = *reinterpret_cast<const float4*>(p);
    } else {
        if (valid > 0) v.x = p[0];
        if (valid > 1) v.y = p[1];
        if (valid > 2) v.z = p[2];
        if (valid > 3) v.w = p[3];
    }
    return v;
}

// Global -> registers for one operand tile of ROWS x 32 (R = ROWS/32 float4 per thread).
template <int ROWS, bool KM>
__device__ __forceinline__ void tile_load(float4 (&r)[ROWS / 32], const float* base, int64_t ld,
                                          const int64_t* gather, int row0, int nrows, int k0, int K,
                                          bool vec, int tid) {
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int f = tid + 256 * i;
        if (!KM) {
            const int row = f >> 3, kq = f & 7;
            const int gr = row0 + row, gk = k0 + 4 * kq;
            if (gr < nrows && gk < K) {
                const int64_t src = gather ? gather[gr] : (int64_t)gr;
                r[i] = load4(base + src * ld + gk, K - gk, vec);
            } else {
                r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            const int k = f / (ROWS / 4), m4 = f % (ROWS / 4);
            const int gk = k0 + k, gr = row0 + 4 * m4;
            if (gk < K && gr < nrows) {
                r[i] = load4(base + (int64_t)gk * ld + gr, nrows - gr, vec);
            } else {
                r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }
}

// Interior tiles (16-byte aligned, whole k stage in range): one pointer per staged float4, set up once, no
// bounds logic in the k loop.  A k-contiguous operand may run past its last row: the rows beyond it read the
// last row again (their products land in output rows or columns that the epilogue does not store), so that a
// ragged last tile row or column takes this path too instead of the loop with bounds checks, whose workgroups
// would finish long after the others.  A row-contiguous operand must lie inside the matrix (its float4 run along
// the rows).
template <int ROWS, bool KM>
__device__ __forceinline__ void tile_ptrs(const float* (&p)[ROWS / 32], const float* base, int64_t ld,
                                          const int64_t* gather, int row0, int nrows, int kbeg, int tid) {
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int f = tid + 256 * i;
        if (!KM) {
            const int row = min(row0 + (f >> 3), nrows - 1), kq = f & 7;
            const int64_t src = gather ? gather[row] : (int64_t)row;
            p[i] = base + src * ld + kbeg + 4 * kq;
        } else {
            const int k = f / (ROWS / 4), m4 = f % (ROWS / 4);
            p[i] = base + (int64_t)(kbeg + k) * ld + row0 + 4 * m4;
        }
    }
}

template <int ROWS>
__device__ __forceinline__ void tile_load_fast(float4 (&r)[ROWS / 32], const float* const (&p)[ROWS / 32], int64_t off) {
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) r[i] = *reinterpret_cast<const float4*>(p[i] + off);
}

template <int ROWS, bool KM>
__device__ __forceinline__ void tile_store(float* lds, const float4 (&r)[ROWS / 32], int tid) {
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int f = tid + 256 * i;
        if (!KM) {
            const int row = f >> 3, kq = f & 7;
            *reinterpret_cast<float4*>(lds + row * MK_LD + 4 * kq) = r[i];
        } else {
            const int k = f / (ROWS / 4), m4 = f % (ROWS / 4);
            *reinterpret_cast<float4*>(lds + k * ROWS + 4 * m4) = r[i];
        }
    }
}

// Fragment values for one 16-k group g: v[j] = tile[row][16g + 8h + j], j = 0..7.
template <int ROWS, bool KM>
__device__ __forceinline__ void frag_read(float (&v)[8], const float* lds, int row, int g, int h) {
    if (!KM) {
        const float4 a = *reinterpret_cast<const float4*>(lds + row * MK_LD + 16 * g + 8 * h);
        const float4 b = *reinterpret_cast<const float4*>(lds + row * MK_LD + 16 * g + 8 * h + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = lds[(16 * g + 8 * h + j) * ROWS + row];
    }
}

template <int BM, int BN, bool A_KM, bool B_KM>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int A_TILE = A_KM ? BK * BM : BM * MK_LD;
    constexpr int B_TILE = B_KM ? BK * BN : BN * MK_LD;
    __shared__ __attribute__((aligned(16))) float lds[2 * (A_TILE + B_TILE)];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;

    // XCD-aware tile order.  Workgroup b runs on XCD b % 8 (round-robin dispatch of a 1-D grid), and each
    // XCD has its own L2.  Every XCD gets one contiguous run of the sequence (split, tile).  The tiles
    // of a split are sequenced so that a run covers ALL tiles along the dimension of the smaller operand
    // and a slice of the other (M < N: all tile-rows, a range of tile-columns: the XCD reads all of A
    // and 1/8 of B), in bands 8 tiles wide that are walked in 8 x 8 blocks; inside a block the fast
    // index is the one whose tiles share a panel of the LARGER operand, so that the workgroups
    // launched back to back - which then run in step - fetch that panel once.
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
        const int b = id / (tf * BAND), r = id - b * tf * BAND;      // band along the slow dimension
        const int w = min(BAND, ts - b * BAND);
        const int c = r / (BAND * w), r2 = r - c * BAND * w;         // block along the fast dimension
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

    float4 ra[BM / 32], rb[BN / 32];
    // split-K: split s owns k in [kbeg, kend) and writes its own partial slab
    const int kbeg = split * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int nk = (kend - kbeg + BK - 1) / BK;
    float* const Cout = g.C + (int64_t)split * g.slab;
    const int nk_full = (kend - kbeg) / BK;  // stages that lie wholly inside the k range
    const bool fast_a = g.a_vec && (!A_KM || m0 + BM <= g.M), fast_b = g.b_vec && (!B_KM || n0 + BN <= g.N);
    const float* pa[BM / 32];
    const float* pb[BN / 32];
    if (fast_a) tile_ptrs<BM, A_KM>(pa, g.A, g.lda, g.a_rows, m0, g.M, kbeg, tid);
    if (fast_b) tile_ptrs<BN, B_KM>(pb, g.B, g.ldb, nullptr, n0, g.N, kbeg, tid);
    const int64_t step_a = A_KM ? (int64_t)BK * g.lda : BK, step_b = B_KM ? (int64_t)BK * g.ldb : BK;
    if (fast_a && nk_full > 0) tile_load_fast<BM>(ra, pa, 0);
    else tile_load<BM, A_KM>(ra, g.A, g.lda, g.a_rows, m0, g.M, kbeg, kend, g.a_vec, tid);
    if (fast_b && nk_full > 0) tile_load_fast<BN>(rb, pb, 0);
    else tile_load<BN, B_KM>(rb, g.B, g.ldb, nullptr, n0, g.N, kbeg, kend, g.b_vec, tid);
    tile_store<BM, A_KM>(lds, ra, tid);
    tile_store<BN, B_KM>(lds + A_TILE, rb, tid);
    __syncthreads();

    // one k stage of matrix instructions from the LDS buffers la / lb
    auto stage_mfma = [&](const float* la, const float* lb) {
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) {
            float av[TM][8], bv[TN][8];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                frag_read<BM, A_KM>(av[i], la, wr * (BM / 2) + i * 32 + l31, grp, h);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                frag_read<BN, B_KM>(bv[j], lb, wc * (BN / 2) + j * 32 + l31, grp, h);
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][s], bv[j][s], acc[i][j], 0, 0, 0);
        }
    };

    int kt = 0;
    // Interior tiles: every stage but the last few is whole and aligned; their loop carries no bounds
    // logic at all (the general loop below, even when its branches are never taken, costs several
    // hundred predicated instructions per stage).
    if (fast_a && fast_b) {
        // one interior stage: compute from buffer `cur` (a compile-time 0 / 1 in the unrolled pair below, so
        // that every LDS address is a constant offset from a per-thread base), stage kt + 1 into the other
        auto interior = [&](const float* la, float* wa) {
            tile_load_fast<BM>(ra, pa, (kt + 1) * step_a);
            tile_load_fast<BN>(rb, pb, (kt + 1) * step_b);
            stage_mfma(la, la + A_TILE);
            tile_store<BM, A_KM>(wa, ra, tid);
            tile_store<BN, B_KM>(wa + A_TILE, rb, tid);
            // issue order of the stage: the next stage's global loads first (their latency runs under the
            // whole stage), LDS fragment reads spread between the matrix instructions, the LDS stores of the
            // loaded tiles between the last ones
            constexpr int NM = TM * TN * 16, NL = (BM + BN) / 32;
            __builtin_amdgcn_sched_group_barrier(0x020, NL, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
            for (int i = 0; i < NM - NL; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int i = 0; i < NL; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
            __syncthreads();
        };
        float* const buf1 = lds + A_TILE + B_TILE;
        // (the pair costs a little set-up: only worth it for long k loops)
        for (; nk_full >= 64 && kt + 2 < nk_full; kt += 2) {   // kt is even here
            interior(lds, buf1);
            ++kt;
            interior(buf1, lds);
            --kt;
        }
        for (; kt + 1 < nk_full; ++kt)
            interior(lds + (kt & 1) * (A_TILE + B_TILE), lds + ((kt + 1) & 1) * (A_TILE + B_TILE));
    }
    for (; kt < nk; ++kt) {
        const float* la = lds + (kt & 1) * (A_TILE + B_TILE);
        const float* lb = la + A_TILE;
        if (kt + 1 < nk) {
            if (fast_a && kt + 1 < nk_full) tile_load_fast<BM>(ra, pa, (kt + 1) * step_a);
            else tile_load<BM, A_KM>(ra, g.A, g.lda, g.a_rows, m0, g.M, kbeg + (kt + 1) * BK, kend, g.a_vec, tid);
            if (fast_b && kt + 1 < nk_full) tile_load_fast<BN>(rb, pb, (kt + 1) * step_b);
            else tile_load<BN, B_KM>(rb, g.B, g.ldb, nullptr, n0, g.N, kbeg + (kt + 1) * BK, kend, g.b_vec, tid);
        }
        stage_mfma(la, lb);
        if (kt + 1 < nk) {
            float* wa = lds + ((kt + 1) & 1) * (A_TILE + B_TILE);
            tile_store<BM, A_KM>(wa, ra, tid);
            tile_store<BN, B_KM>(wa + A_TILE, rb, tid);
        }
        __syncthreads();
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
    if (!a_km && !b_km) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, false, false>), grid, block, 0, st, g);
    else if (!a_km && b_km) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, false, true>), grid, block, 0, st, g);
    else if (a_km && !b_km) hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, true, false>), grid, block, 0, st, g);
    else hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, true, true>), grid, block, 0, st, g);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

}  // namespace

namespace {
int gemm_dispatch(const float* A, int64_t lda, int a_kmajor, const int64_t* a_rows, const float* B, int64_t ldb,
                  int b_kmajor, float* C, int64_t ldc, int M, int N, int K, const float* bias, const float* bias2,
                  int accumulate, int splits, int64_t slab, int tile, hipStream_t st, const char* who) {
    ICRL_REQUIRE(A && B && C, "%s: null operand", who);
    ICRL_REQUIRE(M > 0 && N > 0 && K > 0, "%s: empty shape %dx%dx%d", who, M, N, K);
    ICRL_REQUIRE(!(a_rows && a_kmajor), "%s: row gather needs a_kmajor = 0", who);
    ICRL_REQUIRE(lda >= (a_kmajor ? M : K) && ldb >= (b_kmajor ? N : K) && ldc >= N,
                 "%s: leading dimension smaller than the row length", who);
    ICRL_REQUIRE(splits >= 1 && splits <= 64, "%s: splits must be in [1, 64]", who);
    ICRL_REQUIRE(tile >= 0 && tile <= 3, "%s: tile must be 0..3", who);
    GemmArgs g;
    g.A = A; g.lda = lda; g.a_rows = a_rows; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.bias = bias; g.bias2 = bias2; g.accumulate = accumulate;
    g.a_vec = (reinterpret_cast<uintptr_t>(A) % 16 == 0) && (lda % 4 == 0);
    g.b_vec = (reinterpret_cast<uintptr_t>(B) % 16 == 0) && (ldb % 4 == 0);
    const int ksteps = (K + BK - 1) / BK;
    g.kchunk = ((ksteps + splits - 1) / splits) * BK;
    g.slab = slab;
    const int64_t big_tiles = (int64_t)((M + 127) / 128) * ((N + 127) / 128) * splits;
    if (tile == 0) tile = (big_tiles >= 256 && K > 1024) ? 2 : 3;  // measured: 64x64 wins for short k loops
    if (tile == 1) return launch<128, 128>(g, a_kmajor, b_kmajor, splits, st);
    if (tile == 2) return launch<128, 64>(g, a_kmajor, b_kmajor, splits, st);
    return launch<64, 64>(g, a_kmajor, b_kmajor, splits, st);
}
}  // namespace

namespace icrl {
// Used by the host-side chain schedules in this library.
int gemm(const float* A, int64_t lda, int a_kmajor, const float* B, int64_t ldb, int b_kmajor, float* C,
         int64_t ldc, int M, int N, int K, int splits, int64_t slab, hipStream_t st) {
    return gemm_dispatch(A, lda, a_kmajor, nullptr, B, ldb, b_kmajor, C, ldc, M, N, K, nullptr, nullptr, 0,
                         splits, slab, 0, st, "icrl::gemm");
}
}  // namespace icrl

extern "C" int icrl_gemm_f32(const float* A, int64_t lda, int a_kmajor, const int64_t* a_rows,
                             const float* B, int64_t ldb, int b_kmajor, float* C, int64_t ldc,
                             int M, int N, int K, const float* bias, const float* bias2,
                             int accumulate, int tile, icrl_stream_t stream) {
    return gemm_dispatch(A, lda, a_kmajor, a_rows, B, ldb, b_kmajor, C, ldc, M, N, K, bias, bias2, accumulate, 1,
                         0, tile, icrl::as_stream(stream), "icrl_gemm_f32");
}

extern "C" int icrl_gemm_f32_splitk(const float* A, int64_t lda, int a_kmajor, const float* B, int64_t ldb,
                                    int b_kmajor, float* C_slabs, int64_t ldc, int64_t slab_stride, int M, int N,
                                    int K, int splits, int tile, icrl_stream_t stream) {
    ICRL_REQUIRE(slab_stride >= (int64_t)(M - 1) * ldc + N, "icrl_gemm_f32_splitk: slabs overlap");
    return gemm_dispatch(A, lda, a_kmajor, nullptr, B, ldb, b_kmajor, C_slabs, ldc, M, N, K, nullptr, nullptr, 0,
                         splits, slab_stride, tile, icrl::as_stream(stream), "icrl_gemm_f32_splitk");
}
