// Microbenchmark: HBM read bandwidth vs bytes in flight per CU, with the scan kernel's launch geometry
// (256 workgroups x 4 waves, one workgroup per CU, 1-KiB wave loads, round-robin tiles).  hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

template <int DEPTH>   // loads (1 KiB each per wave) kept in flight per wave
__global__ __launch_bounds__(256) void stream_k(const uint4* __restrict__ src, long n_tiles, unsigned* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long nt = (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    uint4 buf[DEPTH];
    unsigned acc = 0;
    // piece p of tile t: 24 pieces per tile, wave w takes pieces 4i + w
    auto addr = [&](long j) {   // j-th piece of this wave
        const long t = (j / 6) * gridDim.x + blockIdx.x;
        const long p = (j % 6) * 4 + wave;
        return src + (t * 24576 + p * 1024) / 16 + lane;
    };
    const long total = nt * 6;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) buf[d] = (d < total) ? *addr(d) : make_uint4(0, 0, 0, 0);
    for (long j = 0; j < total; j += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const uint4 v = buf[d];
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
            const long nj = j + d + DEPTH;
            buf[d] = (nj < total) ? *addr(nj) : make_uint4(0, 0, 0, 0);
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int DEPTH>
static void run(const uint4* src, long n_tiles, unsigned* out, int grid, const char* tag) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(stream_k<DEPTH>, dim3(grid), dim3(256), 0, 0, src, n_tiles, out);
    hipEventRecord(a);
    const int reps = 5;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(stream_k<DEPTH>, dim3(grid), dim3(256), 0, 0, src, n_tiles, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double gb = (double)n_tiles * 24576 * reps / 1e9;
    printf("%s grid=%d depth=%d (%.0f KiB in flight per workgroup): %.1f GB/s\n", tag, grid, DEPTH, DEPTH * 4.0, gb / (ms / 1e3));
}

int main() {
    const long n_tiles = 1250000;   // 30.7 GB
    uint4* src; unsigned* out;
    if (hipMalloc(&src, n_tiles * 24576) != hipSuccess) return 1;
    hipMalloc(&out, 64);
    hipMemset(src, 1, n_tiles * 24576);
    for (int grid : {256, 512, 1024}) {
        run<6>(src, n_tiles, out, grid, "stream");
        run<12>(src, n_tiles, out, grid, "stream");
        run<24>(src, n_tiles, out, grid, "stream");
        run<48>(src, n_tiles, out, grid, "stream");
    }
    return 0;
}
