// dph_select.hip -- turns the candidate buckets of the filter scan into the exact FAISS answer, and proves it.
//
//   dph_select_kernel : one workgroup per query row.  Sorts the row's bucket (exact integer scores, bitonic in LDS),
//                       re-scores the best C candidates with the EXACT score S = sum_j q_j * x32(n_j) in fp64
//                       (x32 = the reference's fp32 de-quantisation, embed_utils.py:148-149), orders them
//                       (S desc, id asc) and certifies: every row that is NOT re-scored has integer score <= a_rest
//                       (the bound the scan ran under, or the best key left in the bucket), hence
//                         S(row) <= (sc*a_rest + ||e||_2*rmax + <e,mu>) / scale + offset*sum(q) + ||q||_1*dmax
//                       and if the k-th candidate beats that bound the result is the exact top-k.
//   retry plumbing    : rows that could not be certified (lost pairs, ties at the boundary) are compacted on the
//                       device, re-scanned under a bound derived from their own k-th best integer score and selected
//                       again with a larger C -- all enqueued without a host round trip (dph_api.hip).
//   dph_exact_*       : fp64 full scan for rows even that could not cover (threshold collect + sort).
//   dph_merge_kernel  : (score desc, id asc) merge of per-shard top-k lists (multi-GPU).
#include <stdio.h>
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

__global__ __launch_bounds__(SEL_THREADS) void dph_select_kernel(
    const uint64_t* __restrict__ buckets, const unsigned* __restrict__ bucket_counts, const unsigned* __restrict__ overflow,
    const int8_t* __restrict__ db, dph_idmap idmap, const float* __restrict__ x, const dph_qinfo* __restrict__ qinfo,
    const float* __restrict__ lut, const int64_t* __restrict__ row_ids, const unsigned* __restrict__ outliers, int n_out,
    double rmax_all, int q0, const int* __restrict__ gate, int gate_base, int n_q_host, int k, int C, double rmax,
    double delta_max, float offset, float scale, const int* __restrict__ tau, const int* __restrict__ rowmap,
    float* __restrict__ D, int64_t* __restrict__ I, int32_t* __restrict__ status, double* __restrict__ bound_out,
    int32_t* __restrict__ ik_out, int32_t* __restrict__ fail_out, const int32_t* __restrict__ only_failed,
    int* __restrict__ reselect_count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint64_t* pool = (uint64_t*)smem;                          // [DPH_POOL_MAX]
    float* q_lds = (float*)(pool + DPH_POOL_MAX);              // [768]
    float* lut_lds = q_lds + DPH_DIM;                          // [256]
    double* cS = (double*)(lut_lds + 256);                     // [C]
    int64_t* cId = (int64_t*)(cS + C);                         // [C] global ids (the tie order of the answer)
    int* red = (int*)(cId + C);                                // [16]

    const int n_q = dph_gated_rows(gate, gate_base, n_q_host);
    const int qi = blockIdx.x;                 // row inside this pass
    if (qi >= n_q) return;
    const int qrow = q0 + qi;                  // row of the query arrays of this call (x, qinfo)
    const int64_t orow = rowmap ? (int64_t)rowmap[qrow] : (int64_t)qrow;      // row of the caller's output arrays
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // second look at the SAME bucket with a larger C (run_pass): only the rows the first look flagged
    if (only_failed && only_failed[orow] == 0) return;

    const unsigned raw = bucket_counts[qi];
    // pairs were lost on the way (the pair pool ran dry, or more keys than the bucket holds): what is here is a subset of the
    // candidates -- still real rows with exact scores, but nothing can be certified
    const bool lost = raw > (unsigned)DPH_BUCKET_CAP || overflow[qi] != 0u;
    if (only_failed && lost) return;            // a wider re-score cannot repair lost pairs: the row stays flagged for the re-scan
    const uint64_t* keys = buckets + (int64_t)qi * DPH_BUCKET_CAP;
    if (tid == 0) { red[8] = 0; red[10] = (int)0x80000000; red[12] = 0; red[13] = 0; }
    // More keys than the sort holds (8192), all of them in the bucket (round 4; rounds 1-3 called that "lost" and re-scanned the whole
    // shard for the row -- one bench batch in four on the document-ordered dump, whose buckets run 2.4 k keys in the median and up to
    // 8.4 k): take the DPH_POOL_MAX best by integer score.  Radix select of the threshold score sT (4 x 8 bits over the bucket in
    // global memory), every key above it, then keys AT it while there is room; everything left out has I <= sT, which the
    // certificate accounts for (a_cut) exactly like the best key left in the pool.
    int a_cut = (int)0x80000000;
    int nvalid = (int)(raw < (unsigned)DPH_POOL_MAX ? raw : (unsigned)DPH_POOL_MAX);
    const bool cut = !lost && raw > (unsigned)DPH_POOL_MAX;
    if (cut) {
        unsigned* const hist = (unsigned*)q_lds;                  // 256 + 2 words; the query is loaded afterwards
        unsigned prefix = 0;
        int left = DPH_POOL_MAX;
        for (int shift = 24; shift >= 0; shift -= 8) {
            for (int i = tid; i < 258; i += SEL_THREADS) hist[i] = 0;
            __syncthreads();
            const unsigned hi_mask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
            for (unsigned e = tid; e < raw; e += SEL_THREADS) {
                const unsigned sc = (unsigned)(keys[e] >> 32);
                if ((sc & hi_mask) == prefix) atomicAdd(&hist[(sc >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid == 0) {
                unsigned cum = 0;
                int bin = 255;
                for (; bin > 0; --bin) { if (cum + hist[bin] >= (unsigned)left) break; cum += hist[bin]; }
                hist[256] = (unsigned)bin;
                hist[257] = (unsigned)left - cum;
            }
            __syncthreads();
            prefix |= hist[256] << shift;
            left = (int)hist[257];
            __syncthreads();
        }
        const unsigned sT = prefix;                               // the DPH_POOL_MAX-th largest biased score
        for (unsigned e = tid; e < raw; e += SEL_THREADS) {
            const uint64_t key = keys[e];
            if ((unsigned)(key >> 32) > sT) pool[atomicAdd((unsigned*)&red[12], 1u)] = key;       // fewer than DPH_POOL_MAX of them
        }
        __syncthreads();
        const unsigned n_above = (unsigned)red[12];
        for (unsigned e = tid; e < raw; e += SEL_THREADS) {
            const uint64_t key = keys[e];
            if ((unsigned)(key >> 32) == sT) {
                const unsigned slot = n_above + atomicAdd((unsigned*)&red[13], 1u);
                if (slot < (unsigned)DPH_POOL_MAX) pool[slot] = key;
            }
        }
        __syncthreads();
        nvalid = DPH_POOL_MAX;                                    // n_above < DPH_POOL_MAX <= n_above + keys at sT
        a_cut = (int)(sT ^ 0x80000000u);
    }
    int sort_n = 64;
    while (sort_n < nvalid) sort_n <<= 1;       // <= DPH_POOL_MAX
    if (!cut) for (int e = tid; e < sort_n; e += SEL_THREADS) pool[e] = e < nvalid ? keys[e] : 0ull;
    for (int j = tid; j < DPH_DIM; j += SEL_THREADS) q_lds[j] = x[(int64_t)qrow * DPH_DIM + j];
    for (int j = tid; j < 256; j += SEL_THREADS) lut_lds[j] = lut[j];
    __syncthreads();

    // ---- bitonic sort of the pool, descending (keys are distinct except the 0 padding).  Compare-exchange distances
    //      j < 128 stay inside a 256-key chunk: those passes run wave-locally (wave w owns chunks w, w+8, ...; LDS
    //      operations of one wave execute in order, so no workgroup barrier is needed between them) -- only the
    //      ~log^2(n/256)/2 passes with j >= 128 synchronise the whole workgroup
    auto cmpx = [&](int t, int j, int len) {
        const int i0 = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int i1 = i0 | j;
        const bool desc = ((i0 & len) == 0);
        const uint64_t a = pool[i0], b = pool[i1];
        if (desc ? (a < b) : (a > b)) { pool[i0] = b; pool[i1] = a; }
    };
    for (int len = 2; len <= sort_n; len <<= 1) {
        int j = len >> 1;
        for (; j >= 128; j >>= 1) {
            for (int t = tid; t < sort_n / 2; t += SEL_THREADS) cmpx(t, j, len);
            __syncthreads();
        }
        for (int chunk = wv; chunk < (sort_n + 255) / 256; chunk += SEL_THREADS / 64) {
            for (int jj = j; jj > 0; jj >>= 1) {
                for (int t = chunk * 128 + lane; t < chunk * 128 + 128 && t < sort_n / 2; t += 64) cmpx(t, jj, len);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
    }

    const int nc = nvalid < C ? nvalid : C;

    // ---- exact re-score of the best nc candidates (one wave per candidate)
    for (int c = wv; c < nc; c += SEL_THREADS / 64) {
        const unsigned row = dph_key_row(pool[c]);
        const double s = exact_dot_row(db + (int64_t)row * DPH_DIM, q_lds, lut_lds, lane);
        if (lane == 0) { cS[c] = s; cId[c] = row_ids ? row_ids[row] : dph_id_of_row(idmap, (int64_t)row); }
    }
    // ---- outlier rows left in the rest of the pool are not covered by the row-norm cut of the bound: take the best
    //      integer score among them (bounded below with the shard's true maximum norm)
    if (n_out > 0) {
        int best_o = (int)0x80000000;
        for (int e = nc + tid; e < nvalid; e += SEL_THREADS) {
            const unsigned row = dph_key_row(pool[e]);
            int lo = 0, hi = n_out;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (outliers[mid] < row) lo = mid + 1; else hi = mid; }
            if (lo < n_out && outliers[lo] == row) best_o = max(best_o, dph_key_score(pool[e]));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) best_o = max(best_o, __shfl_xor(best_o, o));
        if (lane == 0 && best_o != (int)0x80000000) atomicMax(&red[10], best_o);
    }
    __syncthreads();

    // ---- rank by (S desc, id asc) and emit the top k
    for (int c = tid; c < nc; c += SEL_THREADS) {
        const double s = cS[c];
        const int64_t r = cId[c];
        int rank = 0;
        for (int u = 0; u < nc; ++u) {
            const double su = cS[u];
            rank += (su > s || (su == s && cId[u] < r)) ? 1 : 0;
        }
        if (rank < k) {
            D[orow * k + rank] = (float)s;
            I[orow * k + rank] = r;
        }
        if (rank == k - 1) red[8] = c;          // exactly one candidate has this rank
    }
    for (int c = nc + tid; c < k; c += SEL_THREADS) {   // FAISS padding
        D[orow * k + c] = -FLT_MAX_F;
        I[orow * k + c] = -1;
    }
    __syncthreads();

    // ---- certificate
    if (tid == 0) {
        const dph_qinfo qi_ = qinfo[qrow];
        const bool have_rest_pool = nvalid > nc;
        const bool have_bound = tau != nullptr && tau[qi] != (int)0x80000000;
        int st = 0;
        double bound = -1.0e300;                // upper bound of the reference score of any row not returned
        auto score_bound = [&](int a, double rm) {
            // <q, n> = sc * I + <e, n>,  <e, n> = <e, n - mu> + <e, mu> <= ||e|| * ||n - mu|| + <e, mu>
            const double g = qi_.sc * (double)a + qi_.e_norm2 * rm + qi_.e_mu;
            double b = g / (double)scale + (double)offset * qi_.q_sum + qi_.q_l1 * delta_max;
            return b + 1e-9 * (fabs(b) + 1.0);
        };
        if (lost) {
            st = 1;
            bound = 1.0e300;
        } else if (have_rest_pool || have_bound || a_cut != (int)0x80000000) {
            int a_rest = a_cut;                                   // keys of the bucket the pool had no room for have I <= a_cut
            if (have_rest_pool) a_rest = max(a_rest, dph_key_score(pool[nc]));
            if (have_bound) a_rest = max(a_rest, tau[qi]);        // rows the scan did not emit have I <= tau
            bound = score_bound(a_rest, rmax);
            if (red[10] != (int)0x80000000) bound = fmax(bound, score_bound(red[10], rmax_all));
            // ... and an outlier row among them is not covered by the row-norm cut of the bound
            if (a_cut != (int)0x80000000 && n_out > 0) bound = fmax(bound, score_bound(a_cut, rmax_all));
            if (nc < k) st = 1;                 // rows were dropped before k candidates were collected
            else st = (cS[red[8]] > bound) ? 0 : 1;
        }
        // k-th best exact integer score seen: a lower bound of the true k-th best, what a retry scans under
        if (ik_out) ik_out[orow] = nvalid >= k ? dph_key_score(pool[k - 1]) : (int)0x80000000;
        if (fail_out) fail_out[orow] = (st == 1 && (lost || !bound_out)) ? 1 : 0;
        // sharded search under a bound taken over all shards: this shard may hold fewer than k rows above it, and
        // its k-th may be beaten elsewhere -- the merge decides (certified iff the merged k-th beats `bound`)
        if (bound_out && st == 1 && !lost) st = 2;
        if (bound_out) bound_out[orow] = bound;
        status[orow] = st;
        if (only_failed && st == 0 && reselect_count) atomicAdd(reselect_count, 1);
    }
}

void dph_launch_select(const dph_pass& p, const dph_select_args& a, hipStream_t st) {
    const size_t lds = (size_t)DPH_POOL_MAX * 8 + (DPH_DIM + 256) * 4 + (size_t)a.C * 16 + 64 + 16;
    static std::atomic<bool> attr_set[64];       // the attribute is per device; sized for the largest C (retry passes)
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        const hipError_t e = hipFuncSetAttribute((const void*)dph_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 (int)((size_t)DPH_POOL_MAX * 8 + (DPH_DIM + 256) * 4 + (size_t)DPH_SELECT_C_MAX * 16 + 80));
        if (e != hipSuccess) fprintf(stderr, "libdph: hipFuncSetAttribute(dph_select_kernel): %s\n", hipGetErrorString(e));
        if (dev >= 0 && dev < 64) attr_set[dev] = e == hipSuccess;
    }
    hipLaunchKernelGGL(dph_select_kernel, dim3(p.unit_recs ? p.n_q : DPH_QROWS * p.qb), dim3(SEL_THREADS), lds, st, p.buckets, p.bucket_counts,
                       p.overflow, p.db, p.idmap, p.x, p.qinfo, a.lut, p.row_ids, p.outliers, p.n_out, a.rmax_all, p.q0,
                       p.gate, p.gate_base, p.n_q, a.k, a.C, a.rmax, a.delta_max, a.offset, a.scale, a.tau, a.rowmap, a.D, a.I,
                       a.status, a.bound_out, a.ik_out, a.fail_out, a.only_failed, a.reselect_count);
}

// ------------------------------------------------------------------------------------------ retry plumbing
// ordered compaction of the rows whose flag is set: rows_out[0..*count), *count (one workgroup; n is a few thousand)
__global__ __launch_bounds__(1024) void dph_compact_kernel(const int32_t* __restrict__ fail, int64_t n, int match,
                                                           int32_t* __restrict__ rows_out, int* __restrict__ count_out) {
    __shared__ int wsum[16];
    __shared__ int base_sh;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) base_sh = 0;
    __syncthreads();
    for (int64_t r0 = 0; r0 < n; r0 += 1024) {
        const int64_t r = r0 + tid;
        const bool f = r < n && (match ? fail[r] == match : fail[r] != 0);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(f);
        if (lane == 0) wsum[wv] = (int)__builtin_popcountll(m);
        __syncthreads();
        int off = base_sh;
        for (int w = 0; w < wv; ++w) off += wsum[w];
        if (f) rows_out[off + (int)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = (int32_t)r;
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wsum[w]; base_sh += t; }
        __syncthreads();
    }
    if (tid == 0) *count_out = base_sh;
}
// slot s of the compacted batch <- query vector of row rows[s]
__global__ __launch_bounds__(256) void dph_gather_rows_kernel(const float* __restrict__ x, const int32_t* __restrict__ rows,
                                                              const int* __restrict__ count, float* __restrict__ x_out) {
    const int s = blockIdx.x;
    if (s >= *count) return;
    const float* src = x + (int64_t)rows[s] * DPH_DIM;
    for (int j = threadIdx.x; j < DPH_DIM; j += 256) x_out[(int64_t)s * DPH_DIM + j] = src[j];
}
void dph_launch_gather_rows(const float* x, const int32_t* rows, const int* count, float* x_out, int max_rows, hipStream_t st) {
    if (max_rows > 0) hipLaunchKernelGGL(dph_gather_rows_kernel, dim3(max_rows), dim3(256), 0, st, x, rows, count, x_out);
}
// match = 0: rows whose flag is non-zero; otherwise rows whose flag equals `match`
void dph_launch_compact_failing(const int32_t* fail, int64_t n, int match, const float* x, int32_t* rows_out, int* count_out,
                                float* x_out, int max_rows, hipStream_t st) {
    hipLaunchKernelGGL(dph_compact_kernel, dim3(1), dim3(1024), 0, st, fail, n, match, rows_out, count_out);
    if (x_out) dph_launch_gather_rows(x, rows_out, count_out, x_out, max_rows, st);
}

// bound of a retry: the true k-th best row has integer score >= ik (the k-th best already seen), so every row of the
// true top-k passes 128*H + lmax > ik - margin, and with margin >= 2*(||e||*rmax + |<e,mu>| + scale*||q||_1*dmax)/sc
// the certificate (k-th exact score > score bound of tau) holds by construction once all such rows are re-scored.
__global__ __launch_bounds__(256) void dph_retry_tau_kernel(const int* __restrict__ gate, const int32_t* __restrict__ rows,
                                                            const int32_t* __restrict__ ik, const dph_qinfo* __restrict__ qinfo,
                                                            double rmax, double delta_max, float scale, int* __restrict__ tau_out) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= *gate) return;
    const int v = ik[rows[s]];
    int t = (int)0x80000000;
    if (v != (int)0x80000000) {
        const dph_qinfo q = qinfo[s];
        const double E = q.e_norm2 * rmax + fabs(q.e_mu);
        const double m = ceil((2.0 * E + 2.0 * (double)scale * q.q_l1 * delta_max) / q.sc * (1.0 + 1e-6)) + 4.0;
        const double tv = (double)v - m;
        t = tv < -2147483000.0 ? (int)0x80000000 : (int)tv;
    }
    tau_out[s] = t;
}
void dph_launch_retry_tau(const int* gate, int64_t n_max, const int32_t* rows, const int32_t* ik, const dph_qinfo* qinfo,
                          double rmax, double delta_max, float scale, int* tau_out, hipStream_t st) {
    if (n_max <= 0) return;
    hipLaunchKernelGGL(dph_retry_tau_kernel, dim3((unsigned)((n_max + 255) / 256)), dim3(256), 0, st, gate, rows, ik, qinfo,
                       rmax, delta_max, scale, tau_out);
}

// ------------------------------------------------------------------------------------------ exact fallback
// Phase 1: for failing slot f (query vector x[f] of the compacted batch, output row rows[f]), collect every database
//          row whose exact score is >= thr (thr = the k-th best exact score already known for that row: a lower bound
//          of the true k-th best, or -inf).
// Phase 2: sort what was collected by (S desc, id asc) and write the top k.
// Both are gated by the device-side count: slots >= *n_fail exit at once.
struct dph_exact_hit { double s; int64_t id; };

__global__ __launch_bounds__(256) void dph_exact_collect_kernel(
    const int8_t* __restrict__ db, int64_t n_rows, const float* __restrict__ x, const float* __restrict__ lut,
    const int32_t* __restrict__ rows_out, const int* __restrict__ n_fail, int n_fail_max, int k,
    const float* __restrict__ D_in, dph_idmap idmap, const int64_t* __restrict__ row_ids,
    const unsigned* __restrict__ tilemask, dph_exact_hit* __restrict__ hits, unsigned* __restrict__ counts, unsigned cap,
    const int* __restrict__ redo, const double* __restrict__ thr2) {
    __shared__ float q_lds[DPH_DIM];
    __shared__ float lut_lds[256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int nf = *n_fail < n_fail_max ? *n_fail : n_fail_max;
    if (nf <= 0) return;
    for (int j = tid; j < 256; j += 256) lut_lds[j] = lut[j];
    for (int f = 0; f < nf; ++f) {
        if (redo && !redo[f]) continue;                    // (block-uniform) a later pass: only the slots whose buffer overflowed
        const int64_t orow = rows_out[f];
        __syncthreads();
        for (int j = tid; j < DPH_DIM; j += 256) q_lds[j] = x[(int64_t)f * DPH_DIM + j];
        __syncthreads();
        // threshold: the k-th score of the uncertified answer, lowered by one fp32 ulp-ish margin (D is fp32) -- or, in a later pass,
        // the k-th best of the hits the previous pass kept (dph_exact_tighten_kernel)
        const float dk = D_in[orow * k + (k - 1)];
        const double thr = redo ? thr2[f] : ((dk <= -FLT_MAX_F) ? -1.0e300 : (double)dk - 1e-6 * (fabs((double)dk) + 1.0));
        const int64_t wave0 = (int64_t)blockIdx.x * 4 + (tid >> 6);
        const int64_t nwaves = (int64_t)gridDim.x * 4;
        // IVF: only rows of lists this query row probes (the mask was computed for the compacted batch: slot f)
        const int mword = f >> 5;
        const unsigned mbit = 1u << (f & 31);
        for (int64_t row = wave0; row < n_rows; row += nwaves) {
            const int64_t id = row_ids ? row_ids[row] : dph_id_of_row(idmap, row);
            if (id < 0) continue;
            if (tilemask && !(tilemask[(row >> 5) * 8 + mword] & mbit)) continue;
            const double s = exact_dot_row(db + row * DPH_DIM, q_lds, lut_lds, lane);
            if (lane == 0 && s >= thr) {
                const unsigned pos = atomicAdd(&counts[f], 1u);
                if (pos < cap) { hits[(int64_t)f * cap + pos].s = s; hits[(int64_t)f * cap + pos].id = id; }
            }
        }
    }
}

// A slot whose hits overflowed the buffer (the uncertified answer's k-th score was far below the true one -- or -inf: fewer than k
// candidates survived the filter chain, e.g. a top-k made of saturated rows whose integer scores are too coarse to rank): the k-th best
// of the `cap` hits that WERE kept is the exact score of a real row with k - 1 better ones, i.e. a valid and much tighter threshold.
// It goes to thr2[f], the count is reset and redo[f] asks the next collect pass for this slot again.  (The hits kept are those of the
// rows the waves reached first -- a prefix of every wave's stride, not a random sample -- so a second round may still overflow on
// ordered dumps: dph_launch_exact runs up to three.)
__global__ __launch_bounds__(256) void dph_exact_tighten_kernel(const dph_exact_hit* __restrict__ hits, unsigned* __restrict__ counts,
                                                                unsigned cap, const int* __restrict__ n_fail, int k,
                                                                int* __restrict__ redo, double* __restrict__ thr2) {
    __shared__ double rs[256];
    __shared__ long long ri[256];
    const int f = blockIdx.x;
    if (f >= *n_fail) { if (threadIdx.x == 0) redo[f] = 0; return; }
    if (counts[f] <= cap || cap < (unsigned)k) { if (threadIdx.x == 0) redo[f] = 0; return; }
    const dph_exact_hit* h = hits + (int64_t)f * cap;
    double ps = 0.0;
    int64_t pi = -1;
    for (int r = 0; r < k; ++r) {                          // k rounds of "the best hit behind the previous one", (score desc, id asc)
        double bs = 0.0;
        int64_t bi = -1;
        for (unsigned c = threadIdx.x; c < cap; c += 256) {
            const double s = h[c].s;
            const int64_t id = h[c].id;
            const bool after = r == 0 || s < ps || (s == ps && id > pi);
            if (after && (bi < 0 || s > bs || (s == bs && id < bi))) { bs = s; bi = id; }
        }
        rs[threadIdx.x] = bs; ri[threadIdx.x] = bi;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) {
                const double s2 = rs[threadIdx.x + o];
                const long long i2 = ri[threadIdx.x + o];
                if (i2 >= 0 && (ri[threadIdx.x] < 0 || s2 > rs[threadIdx.x] || (s2 == rs[threadIdx.x] && i2 < ri[threadIdx.x]))) {
                    rs[threadIdx.x] = s2; ri[threadIdx.x] = i2;
                }
            }
            __syncthreads();
        }
        ps = rs[0]; pi = ri[0];
        __syncthreads();
    }
    if (threadIdx.x == 0) { thr2[f] = ps; counts[f] = 0u; redo[f] = 1; }
}

__global__ __launch_bounds__(256) void dph_exact_finish_kernel(
    const dph_exact_hit* __restrict__ hits, const unsigned* __restrict__ counts, unsigned cap,
    const int32_t* __restrict__ rows_out, const int* __restrict__ n_fail, int k, float* __restrict__ D,
    int64_t* __restrict__ I, int32_t* __restrict__ status) {
    const int f = blockIdx.x;
    if (f >= *n_fail) return;
    const int64_t orow = rows_out[f];
    const unsigned n = counts[f];
    if (n > cap) { if (threadIdx.x == 0) status[orow] = 1; return; }     // too many boundary ties to certify
    const dph_exact_hit* h = hits + (int64_t)f * cap;
    if ((uint64_t)n * n <= (uint64_t)4096 * 4096 + (uint64_t)k * n) {
        // rank by counting: n^2 / 256 comparisons per thread
        for (unsigned c = threadIdx.x; c < n; c += 256) {
            const double s = h[c].s;
            const int64_t r = h[c].id;
            unsigned rank = 0;
            for (unsigned u = 0; u < n; ++u) rank += (h[u].s > s || (h[u].s == s && h[u].id < r)) ? 1u : 0u;
            if (rank < (unsigned)k) {
                D[orow * k + rank] = (float)s;
                I[orow * k + rank] = r;
            }
        }
    } else {
        // a long hit list (an all-zero query ties with every row of the shard: n = the shard): k rounds of "the best hit behind the
        // previous one" in (score desc, id asc) order -- k n / 256 comparisons per thread instead of n^2 / 256 (300 000 ties: a minute)
        __shared__ double rs[256];
        __shared__ long long ri[256];
        double ps = 0.0;
        int64_t pi = -1;
        const unsigned kk = n < (unsigned)k ? n : (unsigned)k;
        for (unsigned r = 0; r < kk; ++r) {
            double bs = 0.0;
            int64_t bi = -1;                               // -1: none yet
            for (unsigned c = threadIdx.x; c < n; c += 256) {
                const double s = h[c].s;
                const int64_t id = h[c].id;
                const bool after = r == 0 || s < ps || (s == ps && id > pi);
                if (after && (bi < 0 || s > bs || (s == bs && id < bi))) { bs = s; bi = id; }
            }
            rs[threadIdx.x] = bs; ri[threadIdx.x] = bi;
            __syncthreads();
            for (int o = 128; o > 0; o >>= 1) {
                if ((int)threadIdx.x < o) {
                    const double s2 = rs[threadIdx.x + o];
                    const long long i2 = ri[threadIdx.x + o];
                    if (i2 >= 0 && (ri[threadIdx.x] < 0 || s2 > rs[threadIdx.x] || (s2 == rs[threadIdx.x] && i2 < ri[threadIdx.x]))) {
                        rs[threadIdx.x] = s2; ri[threadIdx.x] = i2;
                    }
                }
                __syncthreads();
            }
            ps = rs[0]; pi = ri[0];
            __syncthreads();
            if (threadIdx.x == 0) { D[orow * k + r] = (float)ps; I[orow * k + r] = pi; }
        }
    }
    for (unsigned c = n + threadIdx.x; c < (unsigned)k; c += 256) {
        D[orow * k + c] = -FLT_MAX_F;
        I[orow * k + c] = -1;
    }
    if (threadIdx.x == 0) status[orow] = 0;
}

void dph_launch_exact(const int8_t* db, int64_t n_rows, dph_idmap idmap, const float* x_dev, const float* lut_dev,
                      const int32_t* rows_dev, const int* n_fail_dev, int n_fail_max, int k, const int64_t* row_ids,
                      const unsigned* tilemask, float* D, int64_t* I, int32_t* status, void* scratch, size_t scratch_bytes,
                      hipStream_t st, int tighten_rounds) {
    // scratch: a head of DPH_EXACT_HEAD bytes -- [n_fail_max] hit counters | [n_fail_max] redo flags | [n_fail_max] tightened
    // thresholds (fp64) -- followed by [n_fail_max][cap] hits
    static_assert(DPH_EXACT_ROWS_DEV * 4 <= 256 && DPH_EXACT_ROWS_DEV * 8 <= DPH_EXACT_HEAD - 512, "the head holds every slot's words");
    unsigned* counts = (unsigned*)scratch;
    int* redo = (int*)((char*)scratch + 256);
    double* thr2 = (double*)((char*)scratch + 512);
    const size_t head = DPH_EXACT_HEAD;
    const size_t cap64 = (scratch_bytes - head) / ((size_t)n_fail_max * sizeof(dph_exact_hit));
    const unsigned cap = (unsigned)(cap64 > 0xFFFFFFu ? 0xFFFFFFu : cap64);
    dph_exact_hit* hits = (dph_exact_hit*)((char*)scratch + head);
    (void)hipMemsetAsync(scratch, 0, head, st);
    hipLaunchKernelGGL(dph_exact_collect_kernel, dim3(1024), dim3(256), 0, st, db, n_rows, x_dev, lut_dev,
                       rows_dev, n_fail_dev, n_fail_max, k, D, idmap, row_ids, tilemask, hits, counts, cap, (const int*)nullptr,
                       (const double*)nullptr);
    // buffers that overflowed: tighten the threshold from what was kept and scan those slots again (nothing to do -- two empty
    // launches per round -- when no buffer overflowed: the slots exit at their redo flag).  The device chain of every search call runs
    // ONE round (it is enqueued whether or not a row needs it); the host form's loop over the rows still uncertified runs three.
    for (int round = 0; round < tighten_rounds; ++round) {
        hipLaunchKernelGGL(dph_exact_tighten_kernel, dim3(n_fail_max), dim3(256), 0, st, hits, counts, cap, n_fail_dev, k, redo, thr2);
        hipLaunchKernelGGL(dph_exact_collect_kernel, dim3(1024), dim3(256), 0, st, db, n_rows, x_dev, lut_dev,
                           rows_dev, n_fail_dev, n_fail_max, k, D, idmap, row_ids, tilemask, hits, counts, cap, (const int*)redo,
                           (const double*)thr2);
    }
    hipLaunchKernelGGL(dph_exact_finish_kernel, dim3(n_fail_max), dim3(256), 0, st, hits, counts, cap, rows_dev,
                       n_fail_dev, k, D, I, status);
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
        bool nonfinite = false;                  // (every part sees the same query row: they all flag it)
        for (int p = 0; p < n_parts; ++p) {
            const int32_t v = ((const int32_t*)(Sp + (int64_t)p * sS))[row];
            if (v == 0) continue;
            if (v == DPH_ROW_NONFINITE) { nonfinite = true; continue; }
            if (v == 2 && Gp && kth > ((const double*)(Gp + (int64_t)p * sG))[row]) continue;
            worst = 1;
        }
        So[row] = nonfinite ? DPH_ROW_NONFINITE : worst;
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
