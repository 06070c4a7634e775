// dph_kmeans.hip -- the centroid UPDATE of the coarse quantizer's k-means, over int8 rows where they lie in HBM.
//
// Replaces what FAISS does inside IndexIVF::train for the reference's IVF indexes (/root/reference/build_phrase_index.py
// :96-142 trains `IndexIVFPQ(IndexFlatIP quantizer, ..., METRIC_INNER_PRODUCT)` on the sample of :60-93): Lloyd
// iterations whose assignment step is a search of the quantizer (max inner product -- libdph's fused MFMA GEMM + arg-max,
// dph_ivf.hip) and whose update step is the mean of the assigned points, L2-normalised for inner-product indexes (FAISS
// sets ClusteringParameters::spherical for METRIC_INNER_PRODUCT).  The update is a segmented reduction:
//   dph_kmeans_accumulate_kernel   exact INTEGER sums of the int8 codes per list (64-bit atomics in the L2: order-free,
//                                  deterministic) + member counts
//   dph_kmeans_finish_kernel       centroid = de-quantised mean (sum/count/scale + offset), normalised if spherical;
//                                  empty lists keep their old centroid (the caller splits a big list into them, as
//                                  FAISS' Clustering does)
// HBM-bound and tiny next to the assignment GEMM: m rows x 768 B read once per iteration.
#include "dph_internal.h"

#include <algorithm>

// one workgroup of 192 threads walks rows blockIdx.x, blockIdx.x + gridDim.x, ...; thread t owns codes 4t .. 4t+3
__global__ __launch_bounds__(192) void dph_kmeans_accumulate_kernel(const int8_t* __restrict__ rows, const int32_t* __restrict__ assign,
                                                                    int64_t m, int nlist, long long* __restrict__ sums,
                                                                    unsigned* __restrict__ counts) {
    const int t = threadIdx.x;
    for (int64_t r = blockIdx.x; r < m; r += gridDim.x) {
        const int l = assign[r];
        if (l < 0 || l >= nlist) continue;
        const unsigned w = *(const unsigned*)(rows + r * DPH_DIM + 4 * t);
        unsigned long long* dst = (unsigned long long*)(sums + (int64_t)l * DPH_DIM + 4 * t);
#pragma unroll
        for (int b = 0; b < 4; ++b) atomicAdd(dst + b, (unsigned long long)(long long)(int8_t)(w >> (8 * b)));
        if (t == 0) atomicAdd(counts + l, 1u);
    }
}

// one workgroup per list
__global__ __launch_bounds__(256) void dph_kmeans_finish_kernel(const long long* __restrict__ sums, const unsigned* __restrict__ counts,
                                                                int nlist, float offset, float scale, int spherical,
                                                                float* __restrict__ centroids) {
    __shared__ double red[4];
    const int l = blockIdx.x, t = threadIdx.x;
    const unsigned c = counts[l];
    if (c == 0u) return;                                   // keep the old centroid
    double v[3], n2 = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        v[i] = (double)sums[(int64_t)l * DPH_DIM + t + 256 * i] / (double)c / (double)scale + (double)offset;
        n2 += v[i] * v[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n2 += __shfl_xor(n2, o);
    if ((t & 63) == 0) red[t >> 6] = n2;
    __syncthreads();
    const double norm = sqrt(red[0] + red[1] + red[2] + red[3]);
    const double s = (spherical && norm > 0.0) ? 1.0 / norm : 1.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) centroids[(int64_t)l * DPH_DIM + t + 256 * i] = (float)(v[i] * s);
}

// sample gather: out[i] = row idx[i] of the resident shard (one wave per row, 12 bytes per lane; waves stride over the sample so
// that the launch stays far below the 2^32 work-items a HIP launch can address)
__global__ __launch_bounds__(256) void dph_gather_sample_kernel(const int8_t* __restrict__ db, int64_t n_rows, const int64_t* __restrict__ idx,
                                                                int64_t m, int8_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * 4;
    for (int64_t w = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6; w < m; w += stride) {
        const int64_t r = idx[w];
        const bool ok = r >= 0 && r < n_rows;
        const unsigned* src = (const unsigned*)(db + (ok ? r : 0) * DPH_DIM) + 3 * lane;
        unsigned* dst = (unsigned*)(out + w * DPH_DIM) + 3 * lane;
#pragma unroll
        for (int i = 0; i < 3; ++i) dst[i] = ok ? src[i] : 0u;
    }
}
void dph_launch_gather_sample(const int8_t* db, int64_t n_rows, const int64_t* idx, int64_t m, int8_t* out, hipStream_t st) {
    if (m > 0)
        hipLaunchKernelGGL(dph_gather_sample_kernel, dim3((unsigned)std::min<int64_t>((m + 3) / 4, 1 << 20)), dim3(256), 0, st, db, n_rows, idx, m,
                           out);
}

void dph_launch_kmeans_update(const int8_t* rows, const int32_t* assign, int64_t m, int nlist, float offset, float scale,
                              int spherical, long long* sums, unsigned* counts, float* centroids, hipStream_t st) {
    (void)hipMemsetAsync(sums, 0, (size_t)nlist * DPH_DIM * sizeof(long long), st);
    (void)hipMemsetAsync(counts, 0, (size_t)nlist * sizeof(unsigned), st);
    if (m > 0) {
        const unsigned grid = (unsigned)(m < 65536 ? m : 65536);
        hipLaunchKernelGGL(dph_kmeans_accumulate_kernel, dim3(grid), dim3(192), 0, st, rows, assign, m, nlist, sums, counts);
    }
    hipLaunchKernelGGL(dph_kmeans_finish_kernel, dim3((unsigned)nlist), dim3(256), 0, st, sums, counts, nlist, offset, scale, spherical,
                       centroids);
}
