// dph_ivf.hip -- the coarse quantizer of the IVF path (BASELINE.json configs[3]: IVF-4096 + in-list exact IP).
// Replaces the IndexFlatIP coarse search inside faiss IndexIVF that the reference offloads to the GPU
// (/root/reference/densephrases/index.py:52-56, nprobe 256 at :53/:62): scores = q . centroid^T, the nprobe best
// lists per query row in (score desc, list id asc) order, emitted as a list-major bit mask for the scan.
//   dph_coarse_kernel   one workgroup per query row: fp64 dot products with every centroid (the oracle's coarse
//                       scores are float64 too, so the probed set is identical, not just similar), exact k-th
//                       largest by bitwise binary search, ties by list id, atomicOr into listmask[nlist][8]
//   dph_tilemask_kernel tilemask[tile] = listmask[list of tile] (32 B per 24 KiB tile: what the scan reads)
#include "dph_internal.h"

__device__ __forceinline__ unsigned long long f64_key(double v) {      // order-preserving map to unsigned
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}

__global__ __launch_bounds__(512) void dph_coarse_kernel(const float* __restrict__ x, int q0, int n_q_host,
                                                         const int* __restrict__ gate, int gate_base,
                                                         const float* __restrict__ centroids, int nlist, int nprobe,
                                                         unsigned* __restrict__ listmask) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long* key = (unsigned long long*)smem;        // [nlist]
    float* q_lds = (float*)(key + nlist);                       // [768]
    unsigned* red = (unsigned*)(q_lds + DPH_DIM);               // [16]
    const int n_q = dph_gated_rows(gate, gate_base, n_q_host);
    const int qi = blockIdx.x;
    if (qi >= n_q) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int j = tid; j < DPH_DIM; j += 512) q_lds[j] = x[(int64_t)(q0 + qi) * DPH_DIM + j];
    __syncthreads();
    double qv[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) qv[j] = (double)q_lds[lane * 12 + j];
    for (int c = wv; c < nlist; c += 8) {
        const float* cp = centroids + (int64_t)c * DPH_DIM + lane * 12;
        const float4 a = *(const float4*)cp, b = *(const float4*)(cp + 4), d = *(const float4*)(cp + 8);
        double acc = qv[0] * a.x + qv[1] * a.y + qv[2] * a.z + qv[3] * a.w;
        acc += qv[4] * b.x + qv[5] * b.y + qv[6] * b.z + qv[7] * b.w;
        acc += qv[8] * d.x + qv[9] * d.y + qv[10] * d.z + qv[11] * d.w;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) key[c] = f64_key(acc);
    }
    __syncthreads();
    const int np = nprobe < nlist ? nprobe : nlist;
    // k-th largest key, bit by bit
    unsigned long long ans = 0;
    for (int bit = 63; bit >= 0; --bit) {
        const unsigned long long cand = ans | (1ull << bit);
        unsigned c = 0;
        for (int i = tid; i < nlist; i += 512) c += key[i] >= cand ? 1u : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if (lane == 0) red[wv] = c;
        __syncthreads();
        unsigned total = 0;
        for (int w = 0; w < 8; ++w) total += red[w];
        __syncthreads();
        if (total >= (unsigned)np) ans = cand;
    }
    // lists strictly above the k-th value are probed; of the ones equal to it, the lowest ids fill the remainder
    unsigned above = 0;
    for (int i = tid; i < nlist; i += 512) above += key[i] > ans ? 1u : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) above += __shfl_xor(above, o);
    if (lane == 0) red[wv] = above;
    __syncthreads();
    unsigned n_above = 0;
    for (int w = 0; w < 8; ++w) n_above += red[w];
    const unsigned word = (unsigned)qi >> 5, bitv = 1u << (qi & 31);
    for (int i = tid; i < nlist; i += 512)
        if (key[i] > ans) atomicOr(&listmask[(int64_t)i * 8 + word], bitv);
    if (tid == 0) {
        int left = np - (int)n_above;
        for (int i = 0; i < nlist && left > 0; ++i)
            if (key[i] == ans) { atomicOr(&listmask[(int64_t)i * 8 + word], bitv); --left; }
    }
}

__global__ __launch_bounds__(256) void dph_tilemask_kernel(const int32_t* __restrict__ tile_list, int64_t n_tiles,
                                                           const uint4* __restrict__ listmask, uint4* __restrict__ tilemask) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < n_tiles) {
        const int64_t l = tile_list[t];
        tilemask[2 * t] = listmask[2 * l];
        tilemask[2 * t + 1] = listmask[2 * l + 1];
    }
}

void dph_launch_coarse(const float* x_dev, int q0, int n_q, const int* gate, int gate_base, const float* centroids, int nlist,
                       int nprobe, unsigned* listmask, const int32_t* tile_list, int64_t n_tiles, unsigned* tilemask,
                       hipStream_t st) {
    (void)hipMemsetAsync(listmask, 0, (size_t)nlist * 32, st);
    const size_t lds = (size_t)nlist * 8 + DPH_DIM * 4 + 64;
    static int attr_lds[64] = {};         // per device: the largest size the attribute was raised to
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || attr_lds[dev] < (int)lds) {
        (void)hipFuncSetAttribute((const void*)dph_coarse_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (dev >= 0 && dev < 64) attr_lds[dev] = (int)lds;
    }
    hipLaunchKernelGGL(dph_coarse_kernel, dim3(n_q), dim3(512), lds, st, x_dev, q0, n_q, gate, gate_base, centroids, nlist,
                       nprobe, listmask);
    hipLaunchKernelGGL(dph_tilemask_kernel, dim3((unsigned)((n_tiles + 255) / 256)), dim3(256), 0, st, tile_list, n_tiles,
                       (const uint4*)listmask, (uint4*)tilemask);
}
