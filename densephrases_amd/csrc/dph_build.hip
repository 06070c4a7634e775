// dph_build.hip -- device-side list builder: turns a flat shard that is resident in HBM into a LIST-MAJOR shard (rows of
// one inverted list contiguous, every list padded to whole 32-row tiles) without the rows ever leaving the GPU.
// Replaces the add-to-index step of /root/reference/build_phrase_index.py:145-153 (faiss add_with_ids into inverted
// lists) for the exact in-list variant.  The list of every row comes from dph_index_assign_dev (dph_ivf.hip).
//   keys     (list << 32 | row), radix-sorted (rocPRIM): list order, id order inside a list -- ties still resolve to
//            the lowest id first
//   starts   first sorted position of every list (binary search per list)
//   gather   sorted position p of list l -> stored row dst_start[l] + (p - src_start[l]); one wave copies one 768-byte
//            row; row_ids / inv_row are written on the way
#include <string.h>
#include <algorithm>
#include <cstring>
#include "dph_internal.h"
#include <rocprim/rocprim.hpp>

__global__ __launch_bounds__(256) void dph_lm_keys_kernel(const int32_t* __restrict__ assign, int64_t n, int nlist,
                                                          uint64_t* __restrict__ keys, int* __restrict__ bad) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const int l = assign[r];
    if (l < 0 || l >= nlist) { *bad = 1; keys[r] = ((uint64_t)(nlist - 1) << 32) | (uint64_t)r; return; }
    keys[r] = ((uint64_t)l << 32) | (uint64_t)r;
}

__global__ __launch_bounds__(256) void dph_lm_starts_kernel(const uint64_t* __restrict__ keys, int64_t n, int nlist,
                                                            int64_t* __restrict__ starts) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l > nlist) return;
    const uint64_t want = (uint64_t)l << 32;
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (keys[mid] < want) lo = mid + 1; else hi = mid; }
    starts[l] = lo;
}

__global__ __launch_bounds__(256) void dph_lm_gather_kernel(const int8_t* __restrict__ src, const uint64_t* __restrict__ keys,
                                                            int64_t n, const int64_t* __restrict__ src_start,
                                                            const int64_t* __restrict__ dst_start, int64_t id_base,
                                                            int8_t* __restrict__ dst, int64_t* __restrict__ row_ids,
                                                            int32_t* __restrict__ inv_row) {
    // grid-stride: one wave per row would be n * 64 work-items, and a HIP launch silently wraps a global size beyond
    // 2^32 -- the first version gathered 36 M of 170 M rows and left the rest of the shard zero (round 3: every full-size
    // list-major timing before this fix searched a fifth of the dump)
    const int lane = threadIdx.x & 63;
    for (int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); p < n; p += (int64_t)gridDim.x * 4) {
        const uint64_t key = keys[p];
        const int l = (int)(key >> 32);
        const int64_t r = (int64_t)(key & 0xFFFFFFFFull);
        const int64_t d = dst_start[l] + (p - src_start[l]);
        if (lane < DPH_DIM / 16)
            ((uint4*)(dst + d * DPH_DIM))[lane] = ((const uint4*)(src + r * DPH_DIM))[lane];
        if (lane == 0) { row_ids[d] = id_base + r; inv_row[r] = (int32_t)d; }
    }
}

// sorted keys (device, caller frees with hipFree) + first sorted position of every list (host, nlist + 1 entries)
int dph_list_major_sort(const int32_t* assign_dev, int64_t n, int nlist, uint64_t** keys_out, int64_t* starts_host, hipStream_t st) {
    uint64_t *k0 = nullptr, *k1 = nullptr;
    int64_t* starts = nullptr;
    int* bad = nullptr;
    void* temp = nullptr;
    int rc = 1;
    do {
        if (hipMalloc((void**)&k0, (size_t)(n > 0 ? n : 1) * 8) != hipSuccess) break;
        if (hipMalloc((void**)&k1, (size_t)(n > 0 ? n : 1) * 8) != hipSuccess) break;
        if (hipMalloc((void**)&starts, (size_t)(nlist + 1) * 8) != hipSuccess) break;
        if (hipMalloc((void**)&bad, 4) != hipSuccess) break;
        (void)hipMemsetAsync(bad, 0, 4, st);
        if (n > 0)
            hipLaunchKernelGGL(dph_lm_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, assign_dev, n, nlist, k0, bad);
        int end_bit = 33;
        while (end_bit < 64 && (1ll << (end_bit - 32)) < nlist) ++end_bit;
        size_t tb = 0;
        if (n > 0) {
            (void)hipGetLastError();        // (rocPRIM reports the thread's last error, whoever left it)
            if (rocprim::radix_sort_keys(nullptr, tb, k0, k1, (size_t)n, 0, (unsigned)end_bit, st) != hipSuccess) break;
            if (hipMalloc(&temp, tb ? tb : 1) != hipSuccess) break;
            if (rocprim::radix_sort_keys(temp, tb, k0, k1, (size_t)n, 0, (unsigned)end_bit, st) != hipSuccess) break;
        }
        hipLaunchKernelGGL(dph_lm_starts_kernel, dim3((nlist + 1 + 255) / 256), dim3(256), 0, st, k1, n, nlist, starts);
        int bad_h = 0;
        if (hipMemcpyAsync(starts_host, starts, (size_t)(nlist + 1) * 8, hipMemcpyDeviceToHost, st) != hipSuccess) break;
        if (hipMemcpyAsync(&bad_h, bad, 4, hipMemcpyDeviceToHost, st) != hipSuccess) break;
        if (hipStreamSynchronize(st) != hipSuccess) break;
        rc = bad_h ? 2 : 0;
    } while (0);
    if (k0) (void)hipFree(k0);
    if (temp) (void)hipFree(temp);
    if (starts) (void)hipFree(starts);
    if (bad) (void)hipFree(bad);
    if (rc != 0 && k1) { (void)hipFree(k1); k1 = nullptr; }
    *keys_out = k1;
    return rc;
}

void dph_launch_list_major_gather(const int8_t* src, const uint64_t* keys, int64_t n, const int64_t* src_start_dev,
                                  const int64_t* dst_start_dev, int64_t id_base, int8_t* dst, int64_t* row_ids,
                                  int32_t* inv_row, hipStream_t st) {
    if (n > 0)
        hipLaunchKernelGGL(dph_lm_gather_kernel, dim3((unsigned)std::min<int64_t>((n + 3) / 4, 1 << 20)), dim3(256), 0, st, src, keys, n, src_start_dev,
                           dst_start_dev, id_base, dst, row_ids, inv_row);
}

// (id, position) pairs sorted by id: the direct map of a PQ index (dph_pq.hip; FAISS DirectMap, build_phrase_index.py:138-142)
int dph_sort_pairs_i64_u32(const int64_t* keys_in, const unsigned* vals_in, int64_t* keys_out, unsigned* vals_out, int64_t n, hipStream_t st) {
    if (n <= 0) return 0;
    size_t tb = 0;
    void* temp = nullptr;
    (void)hipGetLastError();        // rocPRIM reports whatever the thread's last error is: only its own launches count
    if (rocprim::radix_sort_pairs(nullptr, tb, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0, 64, st) != hipSuccess) return 1;
    if (hipMalloc(&temp, tb ? tb : 1) != hipSuccess) return 1;
    const hipError_t e = rocprim::radix_sort_pairs(temp, tb, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0, 64, st);
    const hipError_t e2 = hipStreamSynchronize(st);
    (void)hipFree(temp);
    return (e == hipSuccess && e2 == hipSuccess) ? 0 : 1;
}
