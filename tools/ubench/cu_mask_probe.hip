// Does hipExtStreamCreateWithCUMask partition an MI355X (256 CUs in 8 XCDs, SPX mode), which bit is which CU, and do two streams with
// disjoint masks run their kernels at the same time?  (tools/README.md; DESIGN 9 "a side stream for the latency-bound chain".)
//   build: hipcc --offload-arch=gfx950 -O2 -o cu_mask_probe cu_mask_probe.hip      run: ./cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <set>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ void where_kernel(unsigned* out, int spin) {
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);       // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20);      // HW_REG_XCC_ID
        out[blockIdx.x] = (xcc << 16) | ((hw >> 13) & 7) << 8 | ((hw >> 12) & 1) << 7 | ((hw >> 8) & 15);   // xcc | se | sh | cu
    }
    long long t0 = clock64();
    while (clock64() - t0 < spin) { }
}
__global__ void spin_kernel(long long cycles, unsigned* sink) {
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (threadIdx.x == 0 && blockIdx.x == 0) sink[0] = 1;
}
__global__ void tiny_kernel(unsigned* sink) { if (threadIdx.x == 0) atomicAdd(sink, 1u); }

static int census(hipStream_t st, unsigned* d_out, int n_wg, const char* name) {
    std::vector<unsigned> h(n_wg);
    CHK(hipMemsetAsync(d_out, 0xFF, n_wg * 4, st));
    hipLaunchKernelGGL(where_kernel, dim3(n_wg), dim3(256), 0, st, d_out, 20000);
    CHK(hipStreamSynchronize(st));
    CHK(hipMemcpy(h.data(), d_out, n_wg * 4, hipMemcpyDeviceToHost));
    std::set<unsigned> cus, xccs;
    int per_xcc[16] = {0};
    for (unsigned v : h) { if (cus.insert(v).second) per_xcc[(v >> 16) & 15]++; xccs.insert(v >> 16); }
    printf("%-44s distinct CUs %3zu over %zu XCCs, per XCC:", name, cus.size(), xccs.size());
    for (int i = 0; i < 8; ++i) printf(" %d", per_xcc[i]);
    printf("\n");
    return 0;
}

int main() {
    CHK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, wall clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
    unsigned* d_out;
    CHK(hipMalloc(&d_out, 1 << 20));
    hipStream_t plain;
    CHK(hipStreamCreate(&plain));
    if (census(plain, d_out, 4096, "no mask")) return 1;
    const int words = 8;                                     // 256 bits
    struct { const char* name; uint32_t m[8]; } masks[] = {
        {"bits 0..7", {0xFFu, 0, 0, 0, 0, 0, 0, 0}},
        {"bits 0..31", {0xFFFFFFFFu, 0, 0, 0, 0, 0, 0, 0}},
        {"bits 8..255", {0xFFFFFF00u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}},
        {"bits 0..247", {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, 0x00FFFFFFu}},
        {"bits 248..255", {0, 0, 0, 0, 0, 0, 0, 0xFF000000u}},
        {"every 32nd bit (8 bits)", {1u, 1u, 1u, 1u, 1u, 1u, 1u, 1u}},
        {"bits 0..15", {0xFFFFu, 0, 0, 0, 0, 0, 0, 0}},
    };
    for (auto& mk : masks) {
        hipStream_t st;
        hipError_t e = hipExtStreamCreateWithCUMask(&st, words, mk.m);
        if (e != hipSuccess) { printf("%-44s hipExtStreamCreateWithCUMask: %s\n", mk.name, hipGetErrorString(e)); continue; }
        if (census(st, d_out, 4096, mk.name)) return 1;
        CHK(hipStreamDestroy(st));
    }
    // concurrency: a long kernel on the big partition, a tiny one on the small partition meanwhile
    uint32_t big[8] = {0xFFFFFF00u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}, small[8] = {0xFFu, 0, 0, 0, 0, 0, 0, 0};
    hipStream_t sb, ss;
    CHK(hipExtStreamCreateWithCUMask(&sb, words, big));
    CHK(hipExtStreamCreateWithCUMask(&ss, words, small));
    unsigned* sink;
    CHK(hipMalloc(&sink, 64));
    CHK(hipMemset(sink, 0, 64));
    hipEvent_t a, b, c, d;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b)); CHK(hipEventCreate(&c)); CHK(hipEventCreate(&d));
    for (int trial = 0; trial < 3; ++trial) {
        const bool masked = trial != 2;                     // trial 2: the same on two plain streams (256 workgroups of 1024 threads fill wave slots, not CUs)
        hipStream_t s1 = masked ? sb : plain, s2 = masked ? ss : nullptr;
        hipStream_t plain2 = nullptr;
        if (!masked) { CHK(hipStreamCreate(&plain2)); s2 = plain2; }
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(a, s1));
        hipLaunchKernelGGL(spin_kernel, dim3(248), dim3(256), 65536, s1, (long long)(0.010 * 1e8), sink + 1);   // ~10 ms at 100 MHz wall clock, 64 KiB LDS
        CHK(hipEventRecord(b, s1));
        CHK(hipEventRecord(c, s2));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(8), dim3(256), 0, s2, sink);
        CHK(hipEventRecord(d, s2));
        CHK(hipDeviceSynchronize());
        float t_big, t_small, gap;
        CHK(hipEventElapsedTime(&t_big, a, b)); CHK(hipEventElapsedTime(&t_small, c, d)); CHK(hipEventElapsedTime(&gap, a, d));
        printf("%s: long kernel %.3f ms; 20 tiny kernels on the other stream %.3f ms, done %.3f ms after the long one started -> %s\n",
               masked ? "disjoint CU masks" : "two plain streams", t_big, t_small, gap, gap < t_big * 0.5f ? "CONCURRENT" : "serialised");
        if (plain2) CHK(hipStreamDestroy(plain2));
    }
    return 0;
}
