// f32 GEMM at close to f32 accuracy on the bf16 matrix instruction: every operand element is cut into three
// bf16 pieces x = hi + mid + lo (exactly: 3 x 8 significant bits) on its way into LDS, and a product of two
// elements is the six piece products of weight 2^-16 and above,
//     a.b ~ hi.hi + (hi.mid + mid.hi) + (mid.mid + hi.lo + lo.hi),
// summed in f32 by v_mfma_f32_32x32x16_bf16 (every piece product is exact; what is dropped, mid.lo + lo.mid +
// lo.lo, is below 2^-23 of |a.b|, the size of one f32 rounding of the product).  Six bf16 instructions do the work
// of sixteen v_mfma_f32_32x32x2_f32 (3/8 of the matrix units' cycles), and a long product runs 1.3-1.5 times as
// fast as on the exact-f32 kernel: what bounds it is the delivery of its f32 operands to the CUs (two 128-wide
// tile edges per 128 x 128 x 32 stage).  Sums are accumulated in a different order and with different roundings
// than there, so results agree to f32 rounding error, not bit for bit.  Used for the backward pass's products over
// all (step, row) positions: the two vocabulary-sized ones and the recurrent layers' weight and input gradients.
//
// Same calling convention, operand layouts, split-K slabs and XCD-aware tile order as gemm_f32.hip; 128x128 tiles
// on 256 threads, one wave per 64 x 64 of the tile; k stages of 16 with four stages of global loads in flight in
// registers; LDS rows of k as in gemm_bf16.hip, three images per operand, one stage at a time: 36 KB per
// workgroup, so that four fit a CU and two fit beside a persistent chain workgroup (which leaves 76-80 KB).  With
// 32-deep stages (60 KB) the same loop ran 5-15 % slower at two per CU and 1.4 times slower where only one fits.
//
// NP = 2 is the same kernel on two f16 pieces (icrl_gemm_f16x2s): every row of A and of B (as the product sees them:
// one per output row / column) is scaled by a power of two the caller supplies - its largest entry in [2^14, 2^15),
// see icrl_absmax_scales - and cut into hi + mid, x s = hi + mid to 2^-22 of the element wherever both are normal
// f16 numbers (entries within 2^-18 of the row's largest; smaller ones keep an absolute error of 2^-40 of the
// largest).  An element product is hi.hi + hi.mid + mid.hi on v_mfma_f32_32x32x16_f16, three instructions instead of
// six, two LDS images instead of three (24 KB a stage), and the epilogue divides the scales out again (exactly).
// 1.45-1.6 times the three-piece kernel's rate on the backward pass's shapes, same accuracy class (2^-21 per element
// product against 2^-23: below the float32 rounding of the 10^4-term sums these products are).
#include "icrl_common.h"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 hf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short lds_t;   // a bf16 (NP = 3) or f16 (NP = 2) piece
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 16;
constexpr int LDK = BK + 8;   // bf16 elements per LDS row (12 dwords: the ds_read_b128 of 16 rows at one k offset is conflict-free)
constexpr int BM = 128, BN = 128, NT = 256;
constexpr int A_PLANE = BM * LDK, B_PLANE = BN * LDK;      // one piece's image of an operand tile (bf16 elements)
constexpr int BUF = 3 * (A_PLANE + B_PLANE);               // one stage: A hi, mid, lo, B hi, mid, lo (36 KB; 24 KB of it with two pieces)

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 p;
    p[0] = (__bf16)lo;
    p[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, p);
}

// x = hi + mid + lo exactly, each piece a bf16 (8 significant bits each: x - hi and (x - hi) - mid are exact in f32
// and the last remainder has at most 8 bits); two elements at a time, packed {low half: first element}.
__device__ __forceinline__ void split3(float x, float y, unsigned& hi, unsigned& mid, unsigned& lo) {
    hi = pack2(x, y);
    const float rx = x - __builtin_bit_cast(float, hi << 16), ry = y - __builtin_bit_cast(float, hi & 0xffff0000u);
    mid = pack2(rx, ry);
    const float sx = rx - __builtin_bit_cast(float, mid << 16), sy = ry - __builtin_bit_cast(float, mid & 0xffff0000u);
    lo = pack2(sx, sy);
}

// x = hi + mid + r, |r| <= 2^-22 |x|, two f16 pieces (x already scaled into f16's range)
__device__ __forceinline__ void split2(float x, float y, unsigned& hi, unsigned& mid) {
    typedef _Float16 hf16x2 __attribute__((ext_vector_type(2)));
    hf16x2 h, m;
    h[0] = (_Float16)x; h[1] = (_Float16)y;
    m[0] = (_Float16)(x - (float)h[0]); m[1] = (_Float16)(y - (float)h[1]);
    hi = __builtin_bit_cast(unsigned, h);
    mid = __builtin_bit_cast(unsigned, m);
}

// Staging of one operand tile of ROWS x BK: global -> registers (load), registers -> three LDS images [row][k]
// `plane` elements apart (store3).
// KM = false (k contiguous in memory, [rows][ld]): piece i of thread tid is row f / 4, k quad f % 4, f = tid + NT i:
// one float4, stored as three 8-byte pieces.
// KM = true (rows contiguous in memory, [K][ld]): piece i is row f % ROWS, k octet f / ROWS: eight dword loads, each
// of them 64 consecutive rows per wave, stored as three 16-byte pieces (eight lanes of a store hit eight different
// bank groups; a thread that owned four rows of one k, as float4 loads along the rows would have it, puts the
// lanes of a store four rows apart, which with rows a multiple of 16 bytes apart is two banks for all of them).
template <int ROWS, bool KM, int NTH = NT>
struct Stage {
    static constexpr int PIECES = KM ? ROWS * BK / 8 / NTH : ROWS * BK / 4 / NTH;
    static_assert(PIECES >= 1 && PIECES * NTH * (KM ? 8 : 4) == ROWS * BK, "the tile divides among the threads");
    float x[PIECES][KM ? 8 : 4];
    float sc[PIECES];   // NP = 2: the power-of-two scale of each piece's row (a piece's row is the same in every stage)

    __device__ __forceinline__ void scales(const float* s, int row0, int nrows, int tid) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int f = tid + NTH * i;
            sc[i] = s[min(row0 + (KM ? f % ROWS : f / (BK / 4)), nrows - 1)];
        }
    }

    // EDGE = false: whole k stages and (KM = false) aligned float4: plain loads, no branches.  Rows past the matrix'
    // last one read the last row again (the epilogue does not store their products), so that a ragged last tile row
    // or column does not fall to the bounds-checked loads, whose workgroups would finish long after the others.
    template <bool EDGE>
    __device__ __forceinline__ void load(const float* base, int64_t ld, int row0, int nrows, int k0, int kend, int tid) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int f = tid + NTH * i;
            if (!KM) {
                const int gr = EDGE ? row0 + f / (BK / 4) : min(row0 + f / (BK / 4), nrows - 1), gk = k0 + 4 * (f % (BK / 4));
                const float* p = base + (int64_t)gr * ld + gk;
                if (!EDGE) {
                    const float4 v = *reinterpret_cast<const float4*>(p);
                    x[i][0] = v.x; x[i][1] = v.y; x[i][2] = v.z; x[i][3] = v.w;
                } else {
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

    template <int NP>
    __device__ __forceinline__ void store(lds_t* lds, int plane, int tid) const {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int f = tid + NTH * i;
            if (!KM) {
                uint2 w[3];
                if (NP == 3) {
                    split3(x[i][0], x[i][1], w[0].x, w[1].x, w[2].x);
                    split3(x[i][2], x[i][3], w[0].y, w[1].y, w[2].y);
                } else {
                    split2(x[i][0] * sc[i], x[i][1] * sc[i], w[0].x, w[1].x);
                    split2(x[i][2] * sc[i], x[i][3] * sc[i], w[0].y, w[1].y);
                }
                lds_t* dst = lds + (f / (BK / 4)) * LDK + 4 * (f % (BK / 4));
#pragma unroll
                for (int p = 0; p < NP; ++p) *reinterpret_cast<uint2*>(dst + p * plane) = w[p];
            } else {
                u32x4 w[3];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned hi, mid, lo = 0;
                    if (NP == 3) split3(x[i][2 * e], x[i][2 * e + 1], hi, mid, lo);
                    else split2(x[i][2 * e] * sc[i], x[i][2 * e + 1] * sc[i], hi, mid);
                    w[0][e] = hi; w[1][e] = mid; w[2][e] = lo;
                }
                lds_t* dst = lds + (f % ROWS) * LDK + 8 * (f / ROWS);
#pragma unroll
                for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(dst + p * plane) = w[p];
            }
        }
    }
};

// Fragment of lane (r = lane & 31, h = lane >> 5) for k-step s of 16: elements k = 16 s + 8 h + j, j = 0..7 of row r
__device__ __forceinline__ u32x4 frag(const lds_t* lds, int row, int kstep, int h) {
    return *reinterpret_cast<const u32x4*>(lds + row * LDK + 8 * (2 * kstep + h));
}

struct GemmArgs {
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    float* C; int64_t ldc;
    int M, N, K;
    const float* bias;
    int accumulate;
    int kchunk;        // k-range per split (split-K); a multiple of BK
    int64_t slab;      // C offset (floats) between split-K partial results
    const float* sa;   // NP = 2: power-of-two scale of every row of A (M entries) and of B (N entries)
    const float* sb;
};

template <int NP, bool A_KM, bool B_KM, bool EDGE, int BM_ = BM, int BN_ = BN, int NTH = NT, int WI = 2, int WJ = 2>
__device__ __forceinline__ void k_loop(const GemmArgs& g, f32x16 (&acc)[WI][WJ], lds_t* lds, int m0, int n0, int kbeg,
                                       int kend, int nk, int tid) {
    constexpr int DEPTH = BM_ * BN_ > BM * BN ? 2 : 4;   // stages in flight in registers (the large tile has the accumulators of four)
    constexpr int A_PL = BM_ * LDK, B_PL = BN_ * LDK;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave / (BN_ / (32 * WJ)), wc = wave % (BN_ / (32 * WJ));
    const int l31 = lane & 31, h = lane >> 5;
    Stage<BM_, A_KM, NTH> sa[DEPTH];
    Stage<BN_, B_KM, NTH> sb[DEPTH];
    if (NP == 2) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            sa[d].scales(g.sa, m0, g.M, tid);
            sb[d].scales(g.sb, n0, g.N, tid);
        }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (d < nk) {
            sa[d].template load<EDGE>(g.A, g.lda, m0, g.M, kbeg + d * BK, kend, tid);
            sb[d].template load<EDGE>(g.B, g.ldb, n0, g.N, kbeg + d * BK, kend, tid);
        }
    // One LDS image of a stage, several workgroups per CU: while one multiplies, another cuts its next stage into
    // pieces and writes it.  Two barriers per stage: after the last reads of the stage before, after the writes
    // of this one.
    for (int kt0 = 0; kt0 < nk; kt0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int kt = kt0 + d;
            if (kt < nk) {
                const lds_t* la = lds;
                const lds_t* lb = la + NP * A_PL;
                if (kt > 0) __syncthreads();
                sa[d].template store<NP>(lds, A_PL, tid);
                sb[d].template store<NP>(lds + NP * A_PL, B_PL, tid);
                if (kt + DEPTH < nk) {
                    sa[d].template load<EDGE>(g.A, g.lda, m0, g.M, kbeg + (kt + DEPTH) * BK, kend, tid);
                    sb[d].template load<EDGE>(g.B, g.ldb, n0, g.N, kbeg + (kt + DEPTH) * BK, kend, tid);
                }
                __syncthreads();
#pragma unroll
                for (int ks = 0; ks < BK / 16; ++ks) {
                    u32x4 av[WI][NP], bv[WJ][NP];
#pragma unroll
                    for (int i = 0; i < WI; ++i)
#pragma unroll
                        for (int p = 0; p < NP; ++p)
                            av[i][p] = frag(la + p * A_PL, wr * (32 * WI) + i * 32 + l31, ks, h);
#pragma unroll
                    for (int j = 0; j < WJ; ++j)
#pragma unroll
                        for (int p = 0; p < NP; ++p)
                            bv[j][p] = frag(lb + p * B_PL, wc * (32 * WJ) + j * 32 + l31, ks, h);
                    // piece products, smallest first: (a piece, b piece) with piece 0 = hi, 1 = mid, 2 = lo
                    constexpr int NQ = NP == 3 ? 6 : 3;
                    constexpr int PA[6] = {NP == 3 ? 2 : 1, 0, NP == 3 ? 1 : 0, 1, 0, 0};
                    constexpr int PB[6] = {0, NP == 3 ? 2 : 1, NP == 3 ? 1 : 0, 0, 1, 0};
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
#pragma unroll
                        for (int i = 0; i < WI; ++i)
#pragma unroll
                            for (int j = 0; j < WJ; ++j) {
                                if (NP == 3)
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                                        __builtin_bit_cast(bf16x8, av[i][PA[q]]), __builtin_bit_cast(bf16x8, bv[j][PB[q]]), acc[i][j], 0, 0, 0);
                                else
                                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                        __builtin_bit_cast(hf16x8, av[i][PA[q]]), __builtin_bit_cast(hf16x8, bv[j][PB[q]]), acc[i][j], 0, 0, 0);
                            }
                }
            }
        }
    }
}

template <int NP, bool A_KM, bool B_KM, int BM_ = BM, int BN_ = BN, int NTH = NT, int WI = 2, int WJ = 2>
__device__ __forceinline__ void gemm_pieces(const GemmArgs& g, lds_t* lds) {

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave / (BN_ / (32 * WJ)), wc = wave % (BN_ / (32 * WJ));
    const int l31 = lane & 31, h = lane >> 5;

    // XCD-aware tile order (as in gemm_f32.hip): workgroup b runs on XCD b % 8; every XCD gets one contiguous run of
    // (split, tile); tiles are walked in bands of 8 along the dimension of the larger operand, in 8 x 8 blocks.
    constexpr int BAND = 8;
    const int tiles_m = (g.M + BM_ - 1) / BM_, tiles_n = (g.N + BN_ - 1) / BN_;
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
    const int m0 = mt * BM_, n0 = nt * BN_;

    f32x16 acc[WI][WJ];
#pragma unroll
    for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int j = 0; j < WJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int kbeg = split * g.kchunk;
    const int kend = min(g.K, kbeg + g.kchunk);
    const int nk = (kend - kbeg + BK - 1) / BK;
    float* const Cout = g.C + (int64_t)split * g.slab;

    const bool aligned = (reinterpret_cast<uintptr_t>(g.A) % 16 == 0) && (reinterpret_cast<uintptr_t>(g.B) % 16 == 0) &&
                         g.lda % 4 == 0 && g.ldb % 4 == 0 && kbeg % 4 == 0;
    const bool interior = aligned;   // (ragged tile rows and columns included: Stage::load)
    const int full = (kend - kbeg) / BK;
    if (interior && full > 0) {
        k_loop<NP, A_KM, B_KM, false, BM_, BN_, NTH, WI, WJ>(g, acc, lds, m0, n0, kbeg, kbeg + full * BK, full, tid);
        if (full < nk) {
            __syncthreads();
            k_loop<NP, A_KM, B_KM, true, BM_, BN_, NTH, WI, WJ>(g, acc, lds, m0, n0, kbeg + full * BK, kend, nk - full, tid);
        }
    } else if (nk > 0) {
        k_loop<NP, A_KM, B_KM, true, BM_, BN_, NTH, WI, WJ>(g, acc, lds, m0, n0, kbeg, kend, nk, tid);
    }

    // C/D layout of the 32x32 MFMA: column = lane & 31, row = (r & 3) + 8*(r >> 2) + 4*(lane >> 5)
#pragma unroll
    for (int i = 0; i < WI; ++i)
#pragma unroll
        for (int j = 0; j < WJ; ++j) {
            const int col = n0 + wc * (32 * WJ) + j * 32 + l31;
            if (col >= g.N) continue;
            const float badd = (g.bias && split == 0) ? g.bias[col] : 0.f;
            const float db = NP == 2 ? 1.f / g.sb[col] : 1.f;   // (powers of two: exact)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wr * (32 * WI) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < g.M) {
                    float* dst = Cout + (int64_t)row * g.ldc + col;
                    float v = (NP == 2 ? acc[i][j][r] * (1.f / g.sa[row]) * db : acc[i][j][r]) + badd;
                    if (g.accumulate) v += *dst;
                    *dst = v;
                }
            }
        }
}

template <bool A_KM, bool B_KM>
__global__ __launch_bounds__(NT) void gemm_f32x3_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) lds_t lds[BUF];
    gemm_pieces<3, A_KM, B_KM>(g, lds);
}

template <bool A_KM, bool B_KM>
__global__ __launch_bounds__(NT) void gemm_f16x2s_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) lds_t lds[2 * (A_PLANE + B_PLANE)];
    gemm_pieces<2, A_KM, B_KM>(g, lds);
}

// The same on 256 x 256 tiles: 512 threads, a wave per 128 x 64 (four by two 32 x 32 blocks), one workgroup per CU.
// Per element product half the operand bytes from L2, half the cutting and three quarters of the fragment reads of the
// 128 x 128 tile.
constexpr int BMB = 256, BNB = 256, NTB = 512;
template <bool A_KM, bool B_KM>
__global__ __launch_bounds__(NTB) void gemm_f16x2s_big_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) lds_t lds[2 * (BMB + BNB) * LDK];
    gemm_pieces<2, A_KM, B_KM, BMB, BNB, NTB, 4, 2>(g, lds);
}

template <bool A_KM, bool B_KM>
int launch_one(const GemmArgs& g, int splits, hipStream_t st, bool big = false) {
    if (big) {
        const dim3 grid(((g.M + BMB - 1) / BMB) * ((g.N + BNB - 1) / BNB) * splits);
        hipLaunchKernelGGL((gemm_f16x2s_big_kernel<A_KM, B_KM>), grid, dim3(NTB), 0, st, g);
        ICRL_LAUNCH_CHECK();
        return ICRL_OK;
    }
    const dim3 grid(((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN) * splits);
    if (g.sa) hipLaunchKernelGGL((gemm_f16x2s_kernel<A_KM, B_KM>), grid, dim3(NT), 0, st, g);
    else hipLaunchKernelGGL((gemm_f32x3_kernel<A_KM, B_KM>), grid, dim3(NT), 0, st, g);
    ICRL_LAUNCH_CHECK();
    return ICRL_OK;
}

int dispatch(const float* A, int64_t lda, int a_kmajor, const float* B, int64_t ldb, int b_kmajor, float* C,
             int64_t ldc, int M, int N, int K, const float* bias, int accumulate, int splits, int64_t slab, int tile,
             hipStream_t st, const char* who, const float* sa = nullptr, const float* sb = nullptr) {
    ICRL_REQUIRE(tile == 0 || tile == 1 || (tile == 2 && sa), "%s: tile must be 0 or 1 (128x128), or 2 (256x256, two f16 pieces only)", who);
    ICRL_REQUIRE(A && B && C, "%s: null operand", who);
    ICRL_REQUIRE(M > 0 && N > 0 && K > 0, "%s: empty shape %dx%dx%d", who, M, N, K);
    ICRL_REQUIRE(lda >= (a_kmajor ? M : K) && ldb >= (b_kmajor ? N : K) && ldc >= N,
                 "%s: leading dimension smaller than the row length", who);
    ICRL_REQUIRE(splits >= 1 && splits <= 64, "%s: splits must be in [1, 64]", who);
    GemmArgs g;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.M = M; g.N = N; g.K = K; g.bias = bias; g.accumulate = accumulate;
    const int ksteps = (K + BK - 1) / BK;
    g.kchunk = ((ksteps + splits - 1) / splits) * BK;
    g.slab = slab;
    g.sa = sa; g.sb = sb;
    const bool big = sa && tile == 2;
    if (!a_kmajor && !b_kmajor) return launch_one<false, false>(g, splits, st, big);
    if (!a_kmajor && b_kmajor) return launch_one<false, true>(g, splits, st, big);
    if (a_kmajor && !b_kmajor) return launch_one<true, false>(g, splits, st, big);
    return launch_one<true, true>(g, splits, st, big);
}

}  // namespace

extern "C" int icrl_gemm_f32x3(const float* A, int64_t lda, int a_kmajor, const int64_t* a_rows,
                               const float* B, int64_t ldb, int b_kmajor, float* C, int64_t ldc,
                               int M, int N, int K, const float* bias, const float* bias2,
                               int accumulate, int tile, icrl_stream_t stream) {
    ICRL_REQUIRE(!a_rows && !bias2, "icrl_gemm_f32x3: no row gather and no second bias in this kernel");
    return dispatch(A, lda, a_kmajor, B, ldb, b_kmajor, C, ldc, M, N, K, bias, accumulate, 1, 0, tile,
                    icrl::as_stream(stream), "icrl_gemm_f32x3");
}

extern "C" int icrl_gemm_f32x3_splitk(const float* A, int64_t lda, int a_kmajor, const float* B, int64_t ldb,
                                      int b_kmajor, float* C_slabs, int64_t ldc, int64_t slab_stride, int M, int N,
                                      int K, int splits, int tile, icrl_stream_t stream) {
    ICRL_REQUIRE(slab_stride >= (int64_t)(M - 1) * ldc + N, "icrl_gemm_f32x3_splitk: slabs overlap");
    return dispatch(A, lda, a_kmajor, B, ldb, b_kmajor, C_slabs, ldc, M, N, K, nullptr, 0, splits, slab_stride, tile,
                    icrl::as_stream(stream), "icrl_gemm_f32x3_splitk");
}

extern "C" int icrl_gemm_f16x2s(const float* A, int64_t lda, int a_kmajor, const float* a_scale, const float* B, int64_t ldb,
                                int b_kmajor, const float* b_scale, float* C, int64_t ldc, int M, int N, int K,
                                const float* bias, int accumulate, icrl_stream_t stream) {
    ICRL_REQUIRE(a_scale && b_scale, "icrl_gemm_f16x2s: null scale vector");
    return dispatch(A, lda, a_kmajor, B, ldb, b_kmajor, C, ldc, M, N, K, bias, accumulate, 1, 0, 0,
                    icrl::as_stream(stream), "icrl_gemm_f16x2s", a_scale, b_scale);
}

extern "C" int icrl_gemm_f16x2s_splitk(const float* A, int64_t lda, int a_kmajor, const float* a_scale, const float* B,
                                       int64_t ldb, int b_kmajor, const float* b_scale, float* C_slabs, int64_t ldc,
                                       int64_t slab_stride, int M, int N, int K, int splits, icrl_stream_t stream) {
    ICRL_REQUIRE(a_scale && b_scale, "icrl_gemm_f16x2s_splitk: null scale vector");
    ICRL_REQUIRE(slab_stride >= (int64_t)(M - 1) * ldc + N, "icrl_gemm_f16x2s_splitk: slabs overlap");
    return dispatch(A, lda, a_kmajor, B, ldb, b_kmajor, C_slabs, ldc, M, N, K, nullptr, 0, splits, slab_stride, 0,
                    icrl::as_stream(stream), "icrl_gemm_f16x2s_splitk", a_scale, b_scale);
}

extern "C" int icrl_gemm_f16x2s_splitk_tile(const float* A, int64_t lda, int a_kmajor, const float* a_scale, const float* B,
                                            int64_t ldb, int b_kmajor, const float* b_scale, float* C_slabs, int64_t ldc,
                                            int64_t slab_stride, int M, int N, int K, int splits, int tile,
                                            icrl_stream_t stream) {
    ICRL_REQUIRE(a_scale && b_scale, "icrl_gemm_f16x2s_splitk_tile: null scale vector");
    ICRL_REQUIRE(slab_stride >= (int64_t)(M - 1) * ldc + N, "icrl_gemm_f16x2s_splitk_tile: slabs overlap");
    return dispatch(A, lda, a_kmajor, B, ldb, b_kmajor, C_slabs, ldc, M, N, K, nullptr, 0, splits, slab_stride, tile,
                    icrl::as_stream(stream), "icrl_gemm_f16x2s_splitk_tile", a_scale, b_scale);
}
