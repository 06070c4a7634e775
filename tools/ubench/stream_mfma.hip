// Microbenchmark: does matrix / LDS work on the same CU cost HBM read bandwidth?  The stream kernel of
// stream_inflight.hip (scan-kernel launch geometry: 256 workgroups x 4 waves, one per CU, 1-KiB wave loads, round-robin
// 24 KiB tiles, DEPTH loads in flight per wave) plus, per 1-KiB piece a wave loads, MF int8 MFMAs (32x32x32; the lazy
// scan does 4 per piece, the eager scan 8) and optionally the scan's LDS traffic (1 ds_write_b128 + 4 ds_read_b128 per
// piece and lane).  hipcc --offload-arch=gfx950 -O3 -o stream_mfma stream_mfma.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int DEPTH, int MF, bool LDS, bool EARLY = false, int NREAD = 4, bool PIPE = false, bool BAR = false>
__global__ __launch_bounds__(256) void stream_k(const uint4* __restrict__ src, long n_tiles, unsigned* __restrict__ out) {
    __shared__ uint4 lds[256 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long nt = (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
    uint4 buf[DEPTH];
    unsigned acc = 0;
    v16i c = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    v4i a = {lane, lane * 3, lane * 5, lane * 7}, b = {1, 2, 3, 4};
    v4i an = a;                                      // PIPE: operand read one piece ahead of its MFMAs
    auto addr = [&](long j) {
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
            const long nj = j + d + DEPTH;
            if constexpr (EARLY) {                      // re-issue the load before this piece's LDS / MFMA work
                asm volatile("" : "+v"(buf[d].x), "+v"(buf[d].y), "+v"(buf[d].z), "+v"(buf[d].w));
                buf[d] = (nj < total) ? *addr(nj) : make_uint4(0, 0, 0, 0);
            }
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
            if constexpr (LDS) {
                lds[threadIdx.x + 256 * (d & 3)] = v;
                v4i& dst = PIPE ? an : a;
#pragma unroll
                for (int i = 0; i < NREAD; ++i) {
                    const uint4 r = lds[(threadIdx.x * 5 + 64 * i + 256 * (d & 3)) & 1023];
                    dst.x ^= (int)r.x; dst.y ^= (int)r.y; dst.z ^= (int)r.z; dst.w ^= (int)r.w;
                }
            }
#pragma unroll
            for (int m = 0; m < MF; ++m) c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
            if constexpr (PIPE) { a = an; }
            if constexpr (BAR) { if ((d % 6) == 5) __builtin_amdgcn_s_barrier(); }   // one workgroup barrier per 24 KiB tile, like the scan's hand-over
            if constexpr (!EARLY) buf[d] = (nj < total) ? *addr(nj) : make_uint4(0, 0, 0, 0);
        }
    }
    acc ^= (unsigned)(c[0] ^ c[5] ^ c[15]);
    if (acc == 0x12345678u) out[0] = acc;
}

template <int DEPTH, int MF, bool LDS, bool EARLY = false, int NREAD = 4, bool PIPE = false, bool BAR = false>
static void run(const uint4* src, long n_tiles, unsigned* out) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((stream_k<DEPTH, MF, LDS, EARLY, NREAD, PIPE, BAR>), dim3(256), dim3(256), 0, 0, src, n_tiles, out);
    hipEventRecord(a);
    const int reps = 5;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((stream_k<DEPTH, MF, LDS, EARLY, NREAD, PIPE, BAR>), dim3(256), dim3(256), 0, 0, src, n_tiles, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double gb = (double)n_tiles * 24576 * reps / 1e9;
    printf("depth=%d (%d KiB in flight per CU) mfma_per_KiB=%d lds=%d early_reissue=%d lds_reads=%d pipelined=%d tile_barrier=%d: %.1f GB/s\n", DEPTH, DEPTH * 4, MF,
           (int)LDS, (int)EARLY, LDS ? NREAD : 0, (int)PIPE, (int)BAR, gb / (ms / 1e3));
}

int main() {
    const long n_tiles = 1250000;   // 30.7 GB
    uint4* src; unsigned* out;
    if (hipMalloc(&src, n_tiles * 24576) != hipSuccess) return 1;
    hipMalloc(&out, 64);
    hipMemset(src, 1, n_tiles * 24576);
    run<24, 0, false>(src, n_tiles, out);
    run<24, 4, false>(src, n_tiles, out);
    run<24, 8, false>(src, n_tiles, out);
    run<24, 0, true>(src, n_tiles, out);
    run<24, 4, true>(src, n_tiles, out);
    run<24, 8, true>(src, n_tiles, out);
    run<12, 4, true>(src, n_tiles, out);
    // second series: what recovers the loss of <24,4,lds>?
    run<24, 4, true, true>(src, n_tiles, out);        // earlier re-issue
    run<48, 4, true>(src, n_tiles, out);              // twice the bytes in flight
    run<48, 4, true, true>(src, n_tiles, out);
    run<24, 2, true>(src, n_tiles, out);              // half the matrix work
    run<24, 4, true, false, 2>(src, n_tiles, out);    // half the LDS reads
    run<24, 4, true, false, 1>(src, n_tiles, out);
    // third series: operand reads one piece ahead of the MFMAs that use them
    run<24, 4, true, false, 4, true>(src, n_tiles, out);
    run<48, 4, true, false, 4, true>(src, n_tiles, out);
    run<24, 8, true, false, 4, true>(src, n_tiles, out);
    run<12, 4, true, false, 4, true>(src, n_tiles, out);
    // fourth series: the scan's one workgroup barrier per tile
    run<24, 4, true, false, 4, true, true>(src, n_tiles, out);
    run<24, 0, false, false, 4, false, true>(src, n_tiles, out);
    run<48, 4, true, false, 4, true, true>(src, n_tiles, out);
    run<24, 0, false>(src, n_tiles, out);
    return 0;
}
