// Shared host/device helpers for the A2C caption-training kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/icrl_a2c.h"

namespace icrl {

// Records the message returned by icrl_last_error() (per host thread).
void set_error(const char* fmt, ...);

inline int hip_fail(hipError_t e, const char* what, const char* file, int line) {
    set_error("%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    return ICRL_ERR_HIP;
}

#define ICRL_HIP(call)                                                          \
    do {                                                                        \
        hipError_t e_ = (call);                                                 \
        if (e_ != hipSuccess) return icrl::hip_fail(e_, #call, __FILE__, __LINE__); \
    } while (0)

#define ICRL_REQUIRE(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            icrl::set_error(__VA_ARGS__);       \
            return ICRL_ERR_ARG;                \
        }                                       \
    } while (0)

#define ICRL_LAUNCH_CHECK() ICRL_HIP(hipGetLastError())

static inline hipStream_t as_stream(icrl_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// gemm_f32.hip: C (or `splits` partial slabs, `slab` floats apart) = A * B on the f32 MFMA.
int gemm(const float* A, int64_t lda, int a_kmajor, const float* B, int64_t ldb, int b_kmajor, float* C,
         int64_t ldc, int M, int N, int K, int splits, int64_t slab, hipStream_t st);

constexpr int kHidden = 512;  // hidden width of every recurrent layer on this path

#if defined(__HIPCC__)
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
    // tanh(x) = 1 - 2/(exp(2x)+1); exact limits at +-inf, ~1 ulp-level error in between
    const float e = __expf(2.0f * x);
    return 1.0f - 2.0f / (e + 1.0f);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
#endif

}  // namespace icrl
