// Helpers shared by the segment-parallel chain kernels (per-step launches in chain_seg.hip, the
// persistent forward pair in chain_persist.hip): chain geometry, MFMA fragment helpers, cell math.
#pragma once
#include "icrl_common.h"

// Per-step and persistent kernels must round alike (their results are compared bit for bit), so the
// compiler may not choose where to fuse: only the fused multiply-adds written out below are fused.
#pragma clang fp contract(off)

namespace icrl {
namespace seg {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int H = kHidden;
constexpr int BK = 64;        // k per LDS stage
constexpr int LDT = BK + 4;   // padded row of a k-contiguous tile (ds_read_b128 conflict-free)
constexpr int PLD = 33;       // padded row of the epilogue exchange tiles

struct Geom {
    int n, p0, n_calls;   // rows per column, columns in the first call, calls
    int fc, c0;           // calls walked in full, and their columns fc*p0 + fc(fc-1)/2; a later call contributes its last column only
    int64_t total;        // positions in the chain
    int K, Ls, W;         // segments, positions per segment, warm-up steps
    int stride;           // positions between consecutive steps of a segment: 1; n when every row is its own chain;
                          // 0 when every "segment" walks the SAME positions 0, 1, 2, ... from a state of its own (replay check)
};

// Which positions a chain has.  A call s of the reference walks columns 0 .. p0+s-1 of the caption matrix (all
// n rows of a column, then the next column), so call s+1 first REPLAYS every input of call s from the state
// call s ended in, and then reads one new column.  A recurrence that forgets ends such a replay in the very
// state it started from (the one call s ended in), so the replay can be left out: calls fc .. n_calls-1 then
// contribute only their last column.  fc = n_calls walks every call in full.  Whether the replay may be left
// out is checked by the caller (icrl_chain_pair_replay_check), never assumed.
//
// Position of step q of segment j (0 <= q < Ls: its own positions, q < 0: warm-up before them, q >= Ls:
// warm-up after them).  stride 1: the batch-1 chain cut into K runs of Ls consecutive positions.
// stride n: K = n independent chains, one per row, each walking its row's positions column by column.
__device__ __forceinline__ int64_t seg_pos(const Geom& g, int j, int q) {
    return g.stride == 1 ? (int64_t)j * g.Ls + q : g.stride == 0 ? (int64_t)q : (int64_t)q * g.n + j;
}

// position -> (call s, column t, row i)
__device__ __forceinline__ void locate(const Geom& g, int64_t pos, int& s, int& t, int& i) {
    const int64_t c = pos / g.n;
    i = (int)(pos - c * g.n);
    if (c >= g.c0) { s = g.fc + (int)(c - g.c0); t = g.p0 + s - 1; return; }
    int64_t off = 0;
    for (s = 0; s < g.fc - 1; ++s) {
        if (c < off + g.p0 + s) break;
        off += g.p0 + s;
    }
    t = (int)(c - off);
}

// The same without a search loop (make_geom bounds the chain below 2^30 positions): the column index
// c = pos / n, then the call s from the triangular column count C(s) = s*p0 + s(s-1)/2 <= c by a
// float square root, corrected by at most a step either way.
__device__ __forceinline__ void locate_fast(const Geom& g, int pos, int& s, int& t, int& i) {
    const unsigned c = (unsigned)pos / (unsigned)g.n;
    i = pos - (int)(c * (unsigned)g.n);
    if ((int)c >= g.c0) { s = g.fc + ((int)c - g.c0); t = g.p0 + s - 1; return; }
    const float b = (float)(2 * g.p0 - 1);
    s = (int)((__fsqrt_rn(__fmaf_rn(8.f, (float)c, b * b)) - b) * 0.5f);
    s = min(s, g.fc - 1);
    while (s * g.p0 + s * (s - 1) / 2 > (int)c) --s;
    while (s + 1 <= g.fc - 1 && (s + 1) * g.p0 + (s + 1) * s / 2 <= (int)c) ++s;
    t = (int)c - (s * g.p0 + s * (s - 1) / 2);
}

__device__ __forceinline__ float sg(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float th(float x) { return __fmaf_rn(2.f, sg(2.f * x), -1.f); }
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 ldp4(const float* p) { return make_float4(p[0], p[1], p[2], p[3]); }

__device__ __forceinline__ void atomic_max_nonneg(float* addr, float v) {
    atomicMax(reinterpret_cast<unsigned*>(addr), __float_as_uint(v));  // order-preserving for v >= 0
}

// 8 MFMA steps of one 16-k group: lane half hh supplies k = 8*hh + s at step s for both operands
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

// accumulator register r of lane (l31, hh) is element [row = (r&3) + 8*(r>>2) + 4*hh][col = l31]
__device__ __forceinline__ void acc_to_lds(const f32x16& acc, float* tile, int l31, int hh) {
#pragma unroll
    for (int r = 0; r < 16; ++r) tile[((r & 3) + 8 * (r >> 2) + 4 * hh) * PLD + l31] = acc[r];
}


// positions of a chain whose first `fc` calls are walked in full (see Geom)
inline int64_t chain_columns(int p0, int n_calls, int fc) {
    return (int64_t)fc * p0 + (int64_t)fc * (fc - 1) / 2 + (n_calls - fc);
}

// full_calls: 0 or n_calls = every call in full
inline int make_geom(const char* who, int N, int p0, int n_calls, int full_calls, int segments, int warmup, Geom& g) {
    ICRL_REQUIRE(N > 0 && p0 > 0 && n_calls > 0, "%s: N, p0 and n_calls must be positive", who);
    ICRL_REQUIRE(full_calls >= 0 && full_calls <= n_calls, "%s: full_calls must be in [0, n_calls]", who);
    ICRL_REQUIRE(segments >= 2 && segments <= 4096 && warmup >= 1, "%s: need 2 <= segments <= 4096, warmup >= 1", who);
    g.n = N; g.p0 = p0; g.n_calls = n_calls;
    g.fc = full_calls == 0 ? n_calls : full_calls;
    g.c0 = g.fc * p0 + g.fc * (g.fc - 1) / 2;
    g.total = chain_columns(p0, n_calls, g.fc) * N;
    ICRL_REQUIRE(g.total < (int64_t)1 << 30, "%s: chains of 2^30 or more positions are not supported", who);
    g.K = segments;
    g.Ls = (int)((g.total + segments - 1) / segments);
    g.W = warmup;
    g.stride = 1;
    return ICRL_OK;
}

// every row its own chain over its columns (call after call), from a zero state: N "segments" of
// cols positions each, no warm-up, no seams
inline int make_geom_rows(const char* who, int N, int p0, int n_calls, Geom& g) {
    ICRL_REQUIRE(N > 0 && N <= 4096 && p0 > 0 && n_calls > 0, "%s: need 0 < N <= 4096, p0 > 0, n_calls > 0", who);
    g.n = N; g.p0 = p0; g.n_calls = n_calls;
    g.fc = n_calls;
    g.c0 = g.fc * p0 + g.fc * (g.fc - 1) / 2;
    const int64_t cols = chain_columns(p0, n_calls, n_calls);
    g.total = cols * N;
    ICRL_REQUIRE(g.total < (int64_t)1 << 30, "%s: chains of 2^30 or more positions are not supported", who);
    g.K = N; g.Ls = (int)cols; g.W = 0; g.stride = N;
    return ICRL_OK;
}

inline bool aligned16(const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; }


}  // namespace seg
}  // namespace icrl
