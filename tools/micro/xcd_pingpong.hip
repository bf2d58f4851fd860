// Flag ping-pong between two workgroups, same XCD vs different XCDs, for several cache-policy bits on
// the polling load and the store.  Prints the round trip per policy and whether the poll ever saw the
// partner's store (a policy that lets the load hit the CU's own L1 never does).
//   hipcc --offload-arch=gfx950 -O2 tools/micro/xcd_pingpong.hip -o gpurun_out/xcd_pingpong && gpurun_out/xcd_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <int LD, int ST>
__global__ void pingpong(unsigned* buf, int partner, int iters, unsigned long long* ticks, unsigned* xcc, unsigned* fail) {
    const int b = blockIdx.x;
    if (threadIdx.x == 0) xcc[b] = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15;
    if (b != 0 && b != partner) return;
    if (threadIdx.x != 0) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 4096, 0x00020000);
    const int mine = (b == 0) ? 0 : 256, theirs = (b == 0) ? 256 : 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 1; i <= iters; ++i) {
        if (b == 0) {
            __builtin_amdgcn_raw_buffer_store_b32((unsigned)i, rs, mine, 0, ST);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        int spins = 0;
        while (true) {
            unsigned v = __builtin_amdgcn_raw_buffer_load_b32(rs, theirs, 0, LD);
            if (v >= (unsigned)i) break;
            if (++spins > 200000) { *fail = 1; return; }
        }
        if (b != 0) {
            __builtin_amdgcn_raw_buffer_store_b32((unsigned)i, rs, mine, 0, ST);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    if (b == 0) *ticks = __builtin_readcyclecounter() - t0;
}

template <int LD, int ST>
void run(const char* name, int partner) {
    unsigned *buf, *xcc, *fail;
    unsigned long long* ticks;
    hipMalloc(&buf, 4096); hipMalloc(&xcc, 64 * 4); hipMalloc(&fail, 4); hipMalloc(&ticks, 8);
    hipMemset(buf, 0, 4096); hipMemset(fail, 0, 4); hipMemset(ticks, 0, 8);
    const int iters = 2000;
    pingpong<LD, ST><<<16, 64>>>(buf, partner, iters, ticks, xcc, fail);
    hipDeviceSynchronize();
    unsigned hx[16], hf; unsigned long long ht;
    hipMemcpy(hx, xcc, 64, hipMemcpyDeviceToHost); hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
    hipMemcpy(&ht, ticks, 8, hipMemcpyDeviceToHost);
    printf("%-28s partner %2d (xcc %u vs %u): %s  %.0f shader cycles per round trip\n", name, partner, hx[0], hx[partner],
           hf ? "NEVER SEEN" : "ok", hf ? 0.0 : (double)ht / iters);
    hipFree(buf); hipFree(xcc); hipFree(fail); hipFree(ticks);
}

int main() {
    for (int partner : {8, 1}) {
        run<16, 16>("load sc1, store sc1", partner);
        run<17, 17>("load sc0 sc1, store sc0 sc1", partner);
        run<1, 0>("load sc0, store plain", partner);
        run<1, 1>("load sc0, store sc0", partner);
        run<2, 0>("load nt, store plain", partner);
        run<0, 0>("load plain, store plain", partner);
        run<16, 0>("load sc1, store plain", partner);
        run<1, 16>("load sc0, store sc1", partner);
    }
    return 0;
}
