// Where do the workgroups of a launch on a CU-masked stream run?  Prints, for a few masks, how many workgroups ran on
// each XCD (XCC_ID) - the bit -> CU mapping of hipExtStreamCreateWithCUMask is not documented for multi-XCD parts.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/cu_mask_probe.hip -o gpurun_out/cu_mask_probe && gpurun_out/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

__global__ void where(unsigned* out, int spin) {
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u;   // XCC_ID[3:0]
    const unsigned hwid = __builtin_amdgcn_s_getreg(4 | (31 << 11));       // HW_ID
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
}

static void run(const char* name, const std::vector<uint32_t>& mask) {
    hipStream_t st;
    printf("%s: creating stream\n", name);
    if (mask.empty()) { if (hipStreamCreate(&st) != hipSuccess) { printf("create failed\n"); return; } }
    else if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: create with mask failed\n", name); return; }
    const int blocks = 2048;
    unsigned* d; hipMalloc(&d, blocks * 2 * sizeof(unsigned));
    hipMemsetAsync(d, 0xff, blocks * 2 * sizeof(unsigned), st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st);
    hipLaunchKernelGGL(where, dim3(blocks), dim3(256), 0, st, d, 200000);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned> h(blocks * 2);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    int per[16] = {0};
    for (int b = 0; b < blocks; ++b) per[h[2 * b] & 15]++;
    printf("%-28s %.3f ms  workgroups per XCD:", name, ms);
    for (int x = 0; x < 8; ++x) printf(" %d", per[x]);
    printf("\n");
    hipFree(d); hipStreamDestroy(st);
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("CUs %d\n", p.multiProcessorCount);
    run("no mask", {});
    std::vector<uint32_t> all(8, 0xffffffffu);
    run("all 256 bits", all);
    std::vector<uint32_t> low(8, 0u); for (int i = 0; i < 128; ++i) low[i / 32] |= 1u << (i % 32);
    run("bits 0..127", low);
    std::vector<uint32_t> mod(8, 0u); for (int i = 0; i < 256; ++i) if (i % 8 < 4) mod[i / 32] |= 1u << (i % 32);
    run("bits with i % 8 < 4", mod);
    std::vector<uint32_t> hi(8, 0u); for (int i = 0; i < 256; ++i) if (i % 8 >= 4) hi[i / 32] |= 1u << (i % 32);
    run("bits with i % 8 >= 4", hi);
    std::vector<uint32_t> even(8, 0u); for (int i = 0; i < 256; ++i) if (i % 2 == 0) even[i / 32] |= 1u << (i % 32);
    run("even bits", even);
    return 0;
}
