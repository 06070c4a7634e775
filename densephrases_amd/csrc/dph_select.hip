// dph_select.hip -- turns the candidate lists of the int8 scan into the exact FAISS answer, and proves it.
//
//   dph_select_kernel : one workgroup per query row.  Gathers that row's 2*grid lists, sorts the pool
//                       (bitonic, LDS), re-scores the best C candidates with the EXACT score
//                       S = sum_j q_j * x32(n_j) in fp64 (x32 = the reference's fp32 de-quantisation,
//                       embed_utils.py:148-149), orders them (S desc, id asc) and certifies:
//                       every row that is NOT a candidate has integer score <= a_rest, hence
//                         S(row) <= (sc*a_rest + ||e||_2*rmax + c*sum(e)) / scale + offset*sum(q) + ||q||_1*dmax
//                       and if the k-th candidate beats that bound the result is the exact top-k.
//   dph_exact_*       : fp64 full scan for rows the certificate could not cover (threshold collect + sort).
//   dph_merge_kernel  : (score desc, id asc) merge of per-shard top-k lists (multi-GPU).
#include "dph_internal.h"

#define SEL_THREADS 512
#define FLT_MAX_F 3.4028234663852886e38f

// exact fp64 score of one database row against a query held in LDS; all 64 lanes call, all get the sum.
// Fixed evaluation order (lane-strided chunks of 12, then a butterfly) so equal rows give equal bits.
__device__ __forceinline__ double exact_dot_row(const int8_t* __restrict__ row, const float* q_lds,
                                                const float* lut_lds, int lane) {
    const unsigned* p = (const unsigned*)(row + lane * 12);
    const unsigned w0 = p[0], w1 = p[1], w2 = p[2];
    const unsigned w[3] = {w0, w1, w2};
    double acc = 0.0;
#pragma unroll
    for (int d = 0; d < 3; ++d)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int n = (int)(int8_t)(w[d] >> (8 * b));
            acc += (double)q_lds[lane * 12 + d * 4 + b] * (double)lut_lds[n + 128];
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    return acc;
}

template <int KP>
__global__ __launch_bounds__(SEL_THREADS) void dph_select_kernel(
    const uint64_t* __restrict__ lists, int grid, int pool_pow2, const int8_t* __restrict__ db, int64_t n_rows,
    int64_t id_base, const float* __restrict__ x, const dph_qinfo* __restrict__ qinfo,
    const float* __restrict__ lut, int q0, int n_q, int k, int C, double rmax, double delta_max, float offset,
    float scale, const int* __restrict__ tau_init, const int64_t* __restrict__ row_ids, float* __restrict__ D,
    int64_t* __restrict__ I, int32_t* __restrict__ status, double* __restrict__ bound_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* pool = (uint64_t*)smem;                          // [pool_pow2]
    float* q_lds = (float*)(pool + pool_pow2);                 // [768]
    float* lut_lds = q_lds + DPH_DIM;                          // [256]
    double* cS = (double*)(lut_lds + 256);                     // [C]
    int64_t* cId = (int64_t*)(cS + C);                         // [C] global ids (the tie order of the answer)
    int* red = (int*)(cId + C);                                // [16]

    const int qi = blockIdx.x;                 // row inside this pass
    if (qi >= n_q) return;
    const int qrow = q0 + qi;                  // row of the whole search call
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int qw = qi >> 5, qc = qi & 31;      // owning wave / column inside the scan workgroup

    // ---- gather + compact: lanes (qc) and (qc+32) of wave qw of every scan workgroup; empty slots are skipped so
    //      the sort below only pays for what the lists actually hold (a few hundred keys after the pre-pass)
    const int n_lists = grid * 2;
    int worst_full = (int)0x80000000;          // M: best score any *full* list may have dropped below
    if (tid == 0) red[9] = 0;
    __syncthreads();
    for (int e = tid; e < n_lists * KP; e += SEL_THREADS) {
        const int l = e / KP, i = e % KP;
        const int blk = l >> 1, half = l & 1;
        const uint64_t key = lists[((int64_t)blk * DPH_SCAN_THREADS + qw * 64 + half * 32 + qc) * KP + i];
        if (key != 0) {
            if (i == KP - 1) worst_full = max(worst_full, dph_key_score(key));
            pool[atomicAdd(&red[9], 1)] = key;
        }
    }
    for (int j = tid; j < DPH_DIM; j += SEL_THREADS) q_lds[j] = x[(int64_t)qrow * DPH_DIM + j];
    for (int j = tid; j < 256; j += SEL_THREADS) lut_lds[j] = lut[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) worst_full = max(worst_full, __shfl_xor(worst_full, o));
    if (lane == 0) red[wv] = worst_full;
    __syncthreads();
    worst_full = red[0];
    for (int w = 1; w < SEL_THREADS / 64; ++w) worst_full = max(worst_full, red[w]);
    // rows below the pre-pass bound never entered any list: they are "dropped" rows too
    if (tau_init) worst_full = max(worst_full, tau_init[qi]);
    const int nvalid = red[9];
    int sort_n = 64;
    while (sort_n < nvalid) sort_n <<= 1;       // <= pool_pow2
    for (int e = nvalid + tid; e < sort_n; e += SEL_THREADS) pool[e] = 0;
    __syncthreads();

    // ---- bitonic sort of the compacted pool, descending (keys are distinct except the 0 padding)
    for (int len = 2; len <= sort_n; len <<= 1) {
        for (int j = len >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < sort_n / 2; t += SEL_THREADS) {
                const int i0 = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int i1 = i0 | j;
                const bool desc = ((i0 & len) == 0);
                const uint64_t a = pool[i0], b = pool[i1];
                if (desc ? (a < b) : (a > b)) { pool[i0] = b; pool[i1] = a; }
            }
            __syncthreads();
        }
    }

    const int nc = nvalid < C ? nvalid : C;

    // ---- exact re-score of the best nc candidates (one wave per candidate)
    for (int c = wv; c < nc; c += SEL_THREADS / 64) {
        const unsigned row = dph_key_row(pool[c]);
        const double s = exact_dot_row(db + (int64_t)row * DPH_DIM, q_lds, lut_lds, lane);
        if (lane == 0) { cS[c] = s; cId[c] = row_ids ? row_ids[row] : id_base + (int64_t)row; }
    }
    __syncthreads();

    // ---- rank by (S desc, row asc) and emit the top k
    double kth = -1.0e300;
    for (int c = tid; c < nc; c += SEL_THREADS) {
        const double s = cS[c];
        const int64_t r = cId[c];
        int rank = 0;
        for (int u = 0; u < nc; ++u) {
            const double su = cS[u];
            rank += (su > s || (su == s && cId[u] < r)) ? 1 : 0;
        }
        if (rank < k) {
            D[(int64_t)qrow * k + rank] = (float)s;
            I[(int64_t)qrow * k + rank] = r;
        }
        if (rank == k - 1) red[8] = c;          // exactly one candidate has this rank
    }
    for (int c = nc + tid; c < k; c += SEL_THREADS) {   // FAISS padding
        D[(int64_t)qrow * k + c] = -FLT_MAX_F;
        I[(int64_t)qrow * k + c] = -1;
    }
    __syncthreads();

    // ---- certificate
    if (tid == 0) {
        const dph_qinfo qi_ = qinfo[qrow];
        const bool have_rest_pool = nvalid > nc;
        const bool have_rest_lists = worst_full != (int)0x80000000;
        int st = 0;
        double bound = -1.0e300;                // upper bound of the reference score of any row not returned
        if (have_rest_pool || have_rest_lists) {
            int a_rest = (int)0x80000000;
            if (have_rest_pool) a_rest = max(a_rest, dph_key_score(pool[nc]));
            if (have_rest_lists) a_rest = max(a_rest, worst_full);
            const double g_bound = qi_.sc * (double)a_rest + qi_.e_norm2 * rmax + (double)DPH_CENTER * qi_.e_sum;
            bound = g_bound / (double)scale + (double)offset * qi_.q_sum + qi_.q_l1 * delta_max;
            bound += 1e-9 * (fabs(bound) + 1.0);
            if (nc < k) {
                st = 1;                         // rows were dropped before k candidates were collected
            } else {
                kth = cS[red[8]];
                st = (kth > bound) ? 0 : 1;
            }
            // sharded search under a bound taken over all shards: this shard may hold fewer than k rows above it, and
            // its k-th may be beaten elsewhere -- the merge decides (certified iff the merged k-th beats `bound`)
            if (bound_out && st == 1) st = 2;
        }
        if (bound_out) bound_out[qrow] = bound;
        status[qrow] = st;
    }
}

void dph_launch_select(int kp, int grid, const uint64_t* lists, const int8_t* db, int64_t n_rows, int64_t id_base,
                       const float* x_dev, const dph_qinfo* qinfo, const float* lut_dev, int q0, int n_q, int k,
                       double rmax, double delta_max, float offset, float scale, const int* tau_init,
                       const int64_t* row_ids, float* D, int64_t* I, int32_t* status, double* bound_out, hipStream_t st) {
    int pool = 1;
    while (pool < grid * 2 * kp) pool <<= 1;
    int C = k + 32;
    if (C < 2 * k) C = 2 * k;
    if (C > pool) C = pool;
    const size_t lds = (size_t)pool * 8 + (DPH_DIM + 256) * 4 + (size_t)C * 16 + 64 + 16;
    if (kp == 16) {
        (void)hipFuncSetAttribute((const void*)dph_select_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        hipLaunchKernelGGL((dph_select_kernel<16>), dim3(n_q), dim3(SEL_THREADS), lds, st, lists, grid, pool, db,
                           n_rows, id_base, x_dev, qinfo, lut_dev, q0, n_q, k, C, rmax, delta_max, offset, scale,
                           tau_init, row_ids, D, I, status, bound_out);
    } else {
        (void)hipFuncSetAttribute((const void*)dph_select_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        hipLaunchKernelGGL((dph_select_kernel<32>), dim3(n_q), dim3(SEL_THREADS), lds, st, lists, grid, pool, db,
                           n_rows, id_base, x_dev, qinfo, lut_dev, q0, n_q, k, C, rmax, delta_max, offset, scale,
                           tau_init, row_ids, D, I, status, bound_out);
    }
}

// ------------------------------------------------------------------------------------------ exact fallback
// Phase 1: for every failing query row f, collect every database row whose exact score is >= thr[f]
//          (thr = the k-th best exact score already known: a lower bound of the true k-th best, or -inf).
// Phase 2: sort what was collected by (S desc, id asc) and write the top k.
struct dph_exact_hit { double s; int64_t id; };

__global__ __launch_bounds__(256) void dph_exact_collect_kernel(
    const int8_t* __restrict__ db, int64_t n_rows, const float* __restrict__ x, const float* __restrict__ lut,
    const int32_t* __restrict__ fail_rows, int n_fail, int k, const float* __restrict__ D_in, int64_t id_base,
    const int64_t* __restrict__ row_ids, const unsigned* __restrict__ tilemask, dph_exact_hit* __restrict__ hits,
    unsigned* __restrict__ counts, unsigned cap) {
    __shared__ float q_lds[DPH_DIM];
    __shared__ float lut_lds[256];
    const int f = blockIdx.y;
    const int qrow = fail_rows[f];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int j = tid; j < DPH_DIM; j += 256) q_lds[j] = x[(int64_t)qrow * DPH_DIM + j];
    for (int j = tid; j < 256; j += 256) lut_lds[j] = lut[j];
    __syncthreads();
    // threshold: the k-th score of the uncertified answer, lowered by one fp32 ulp-ish margin (D is fp32)
    const float dk = D_in[(int64_t)qrow * k + (k - 1)];
    const double thr = (dk <= -FLT_MAX_F) ? -1.0e300 : (double)dk - 1e-6 * (fabs((double)dk) + 1.0);
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (tid >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    // IVF: only rows of lists this query row probes (the mask is the one of the pass the row belongs to: qrow < 128)
    const int mword = (qrow & 127) >> 5;
    const unsigned mbit = 1u << (qrow & 31);
    for (int64_t row = wave0; row < n_rows; row += nwaves) {
        const int64_t id = row_ids ? row_ids[row] : id_base + row;
        if (id < 0) continue;
        if (tilemask && !(tilemask[(row >> 5) * 4 + mword] & mbit)) continue;
        const double s = exact_dot_row(db + row * DPH_DIM, q_lds, lut_lds, lane);
        if (lane == 0 && s >= thr) {
            const unsigned pos = atomicAdd(&counts[f], 1u);
            if (pos < cap) { hits[(int64_t)f * cap + pos].s = s; hits[(int64_t)f * cap + pos].id = id; }
        }
    }
}

__global__ __launch_bounds__(256) void dph_exact_finish_kernel(
    const dph_exact_hit* __restrict__ hits, const unsigned* __restrict__ counts, unsigned cap,
    const int32_t* __restrict__ fail_rows, int k, float* __restrict__ D, int64_t* __restrict__ I,
    int32_t* __restrict__ status) {
    const int f = blockIdx.x;
    const int qrow = fail_rows[f];
    const unsigned n = counts[f];
    if (n > cap) { if (threadIdx.x == 0) status[qrow] = 2; return; }     // too many boundary ties to certify
    const dph_exact_hit* h = hits + (int64_t)f * cap;
    for (unsigned c = threadIdx.x; c < n; c += 256) {
        const double s = h[c].s;
        const int64_t r = h[c].id;
        unsigned rank = 0;
        for (unsigned u = 0; u < n; ++u) rank += (h[u].s > s || (h[u].s == s && h[u].id < r)) ? 1u : 0u;
        if (rank < (unsigned)k) {
            D[(int64_t)qrow * k + rank] = (float)s;
            I[(int64_t)qrow * k + rank] = r;
        }
    }
    for (unsigned c = n + threadIdx.x; c < (unsigned)k; c += 256) {
        D[(int64_t)qrow * k + c] = -FLT_MAX_F;
        I[(int64_t)qrow * k + c] = -1;
    }
    if (threadIdx.x == 0) status[qrow] = 0;
}

void dph_launch_exact(const int8_t* db, int64_t n_rows, int64_t id_base, const float* x_dev, const float* lut_dev,
                      const int32_t* rows_dev, int n_fail, int k, const int64_t* row_ids, const unsigned* tilemask,
                      float* D, int64_t* I, int32_t* status, void* scratch, size_t scratch_bytes, hipStream_t st) {
    // scratch: [n_fail] counters (256-byte aligned block) followed by [n_fail][cap] hits
    unsigned* counts = (unsigned*)scratch;
    const size_t head = ((size_t)n_fail * 4 + 255) / 256 * 256;
    const size_t cap64 = (scratch_bytes - head) / ((size_t)n_fail * sizeof(dph_exact_hit));
    const unsigned cap = (unsigned)(cap64 > 0xFFFFFFu ? 0xFFFFFFu : cap64);
    dph_exact_hit* hits = (dph_exact_hit*)((char*)scratch + head);
    (void)hipMemsetAsync(counts, 0, (size_t)n_fail * 4, st);
    hipLaunchKernelGGL(dph_exact_collect_kernel, dim3(1024, n_fail), dim3(256), 0, st, db, n_rows, x_dev, lut_dev,
                       rows_dev, n_fail, k, D, id_base, row_ids, tilemask, hits, counts, cap);
    hipLaunchKernelGGL(dph_exact_finish_kernel, dim3(n_fail), dim3(256), 0, st, hits, counts, cap, rows_dev, k, D, I,
                       status);
}

// ------------------------------------------------------------------------------------------ multi-GPU merge
// part p's [n,k] block starts p*sD bytes after Dp and p*sI bytes after Ip (dense: sD = n*k*4, sI = n*k*8; views
// into one packed all-gather buffer: sD = sI = the per-rank record size).
__global__ __launch_bounds__(256) void dph_merge_kernel(const char* __restrict__ Dp, const char* __restrict__ Ip,
                                                        const char* __restrict__ Bp, const char* __restrict__ Pp,
                                                        const char* __restrict__ Sp, const char* __restrict__ Gp,
                                                        int n_parts, int64_t sD,
                                                        int64_t sI, int64_t sB, int64_t sP, int64_t sS, int64_t sG, int64_t n,
                                                        int k, float* __restrict__ Do, int64_t* __restrict__ Io,
                                                        int32_t* __restrict__ src, double* __restrict__ Bo,
                                                        int32_t* __restrict__ Po, int32_t* __restrict__ So) {
    const int64_t row = blockIdx.x;
    const int m = n_parts * k;
    auto ld_i = [&](int c) { return ((const int64_t*)(Ip + (int64_t)(c / k) * sI))[row * k + (c % k)]; };
    auto ld_d = [&](int c) { return ((const float*)(Dp + (int64_t)(c / k) * sD))[row * k + (c % k)]; };
    __shared__ int nvalid;
    __shared__ float kth_sh;
    if (threadIdx.x == 0) { nvalid = 0; kth_sh = -FLT_MAX_F; }
    __syncthreads();
    int local = 0;
    for (int c = threadIdx.x; c < m; c += 256) {
        const int64_t id = ld_i(c);
        if (id < 0) continue;
        ++local;
        const float s = ld_d(c);
        int rank = 0;
        for (int u = 0; u < m; ++u) {
            const int64_t iu = ld_i(u);
            if (iu < 0) continue;
            const float su = ld_d(u);
            rank += (su > s || (su == s && iu < id)) ? 1 : 0;
        }
        if (rank < k) {
            Do[row * k + rank] = s;
            Io[row * k + rank] = id;
            if (src) src[row * k + rank] = c;
            if (rank == k - 1) kth_sh = s;
            // follow the winner into its part's window results (the candidate keeps the re-score of its home shard)
            if (Bo) Bo[row * k + rank] = ((const double*)(Bp + (int64_t)(c / k) * sB))[row * k + (c % k)];
            if (Po) Po[row * k + rank] = ((const int32_t*)(Pp + (int64_t)(c / k) * sP))[row * k + (c % k)];
        }
    }
    atomicAdd(&nvalid, local);
    __syncthreads();
    for (int c = nvalid + threadIdx.x; c < k; c += 256) {      // FAISS padding
        Do[row * k + c] = -FLT_MAX_F;
        Io[row * k + c] = -1;
        if (src) src[row * k + c] = -1;
        if (Bo) Bo[row * k + c] = -1e9;
        if (Po) Po[row * k + c] = -1;
    }
    if (So && threadIdx.x == 0) {
        // certified iff every part is: a part that closed its own top-k (status 0) is (the merged k-th is at least
        // its k-th); a part that deferred (status 2, search under a union bound) is iff the merged k-th score beats
        // the bound of everything that part did not return (D is the fp32 rounding of the score: one ulp of margin)
        const double kth = nvalid >= k ? (double)kth_sh - fabs((double)kth_sh) * 1.2e-7 : -1.0e300;
        int32_t worst = 0;
        for (int p = 0; p < n_parts; ++p) {
            const int32_t v = ((const int32_t*)(Sp + (int64_t)p * sS))[row];
            if (v == 0) continue;
            if (v == 2 && Gp && kth > ((const double*)(Gp + (int64_t)p * sG))[row]) continue;
            worst = 1;
        }
        So[row] = worst;
    }
}

void dph_launch_merge(const float* D_parts, const int64_t* I_parts, const double* best_parts, const int32_t* pred_parts,
                      const int32_t* status_parts, const double* bound_parts, int n_parts, int64_t stride_bytes, int64_t n,
                      int k, float* D_out, int64_t* I_out, int32_t* src_out, double* best_out, int32_t* pred_out,
                      int32_t* status_out, hipStream_t st) {
    const int64_t sD = stride_bytes ? stride_bytes : n * k * 4;
    const int64_t sI = stride_bytes ? stride_bytes : n * k * 8;
    const int64_t sB = stride_bytes ? stride_bytes : n * k * 8;
    const int64_t sP = stride_bytes ? stride_bytes : n * k * 4;
    const int64_t sS = stride_bytes ? stride_bytes : n * 4;
    const int64_t sG = stride_bytes ? stride_bytes : n * 8;
    hipLaunchKernelGGL(dph_merge_kernel, dim3((unsigned)n), dim3(256), 0, st, (const char*)D_parts, (const char*)I_parts,
                       (const char*)best_parts, (const char*)pred_parts, (const char*)status_parts,
                       (const char*)bound_parts, n_parts, sD, sI, sB, sP, sS, sG, n, k, D_out, I_out, src_out,
                       best_parts ? best_out : nullptr, pred_parts ? pred_out : nullptr,
                       status_parts ? status_out : nullptr);
}
