// dph_refine.hip -- between the filter scan and the select/certify step:
//   dph_refine_kernel    exact integer score I = 128*(<q1,n> + replica digits x rogue codes) + <q2,n> of every (row, query row) pair the scan emitted,
//                        bucketed per query row as 64-bit keys (score desc, row asc)
//   dph_outlier_kernel   the shard's outlier rows (norm above the certificate's row-norm cut) against EVERY query row:
//                        they are candidates by construction, never bounded
//   dph_threshold_kernel KP-th best integer score of a query row's bucket -> the bound tau the next (finer) scan level
//                        runs under (a sampled lower bound of the KP-th best score of the whole shard)
//   dph_union_bounds_kernel  the same bound over the union of several shards' samples (range-sharded dumps)
// No reference counterpart: FAISS keeps a heap per query inside its scan (IndexFlat / IVF scanners behind
// /root/reference/densephrases/index.py:200); here the scan only filters and these kernels do the bookkeeping.
#include "dph_internal.h"

__device__ __forceinline__ int dot48(const uint4 (&a)[3], const uint4 (&b)[3]) {
    int acc = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        acc = __builtin_amdgcn_sdot4((int)a[i].x, (int)b[i].x, acc, false);
        acc = __builtin_amdgcn_sdot4((int)a[i].y, (int)b[i].y, acc, false);
        acc = __builtin_amdgcn_sdot4((int)a[i].z, (int)b[i].z, acc, false);
        acc = __builtin_amdgcn_sdot4((int)a[i].w, (int)b[i].w, acc, false);
    }
    return acc;
}
__device__ __forceinline__ int sum16(int v) {        // sum over the 16 lanes of a group (all get it)
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}
// The replica digits of a query row (dph_quantize_kernel: further high digits of the shard's rogue dimensions) against a database
// row: sum over the replica slots of digit x raw code of the slot's dimension -- part of the HIGH-digit score.  Lane l16 of the
// 16 that score a pair takes slots l16 and l16 + 16; the caller's sum16 adds them up.
__device__ __forceinline__ int replica_part(const int8_t* __restrict__ row, const int8_t* __restrict__ qaux_row,
                                            const dph_aux_layout& lay, int l16) {
    int acc = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int s = l16 + 16 * h;
        if (s < lay.n_rep) acc += (int)qaux_row[lay.n_norm + s] * (int)row[lay.rep_dim[s]];
    }
    return acc;
}
// is `row` one of the shard's (sorted) outlier rows?
__device__ __forceinline__ bool is_outlier(const unsigned* __restrict__ outl, int n_out, unsigned row) {
    int lo = 0, hi = n_out;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (outl[mid] < row) lo = mid + 1; else hi = mid;
    }
    return lo < n_out && outl[lo] == row;
}

// One workgroup per claimed chunk of the pair pool (dph_internal.h: DPH_CHUNK_PAIRS pairs each; the launch is a fixed
// grid walking the chunks the scan claimed -- a device-side count).  Sixteen lanes score one pair: 48 bytes of the
// database row and of both query digits per lane, 24 v_dot4_i32_i8, a 4-step butterfly.  Bucket slots are reserved per
// (chunk, query row) -- one global atomic per query row of a chunk instead of one per pair.  Pairs carry the query row
// of the pass, so a chunk may hold any of them (a flat scan wave emits for its own 32*qb rows, a unit-scan wave for
// whatever rows probe the lists it walked).
#define REFINE_GRID 4096
__global__ __launch_bounds__(256) void dph_refine_kernel(
    const int8_t* __restrict__ db, const int64_t* __restrict__ row_ids, uint2* __restrict__ pairs,
    const unsigned* __restrict__ pool_head, const unsigned* __restrict__ chunk_fill, const int8_t* __restrict__ q1,
    const int8_t* __restrict__ q2, int q0, const int* __restrict__ gate, int gate_base, int n_q_host,
    const unsigned* __restrict__ outliers, int n_out, uint64_t* __restrict__ buckets, unsigned* __restrict__ bucket_counts,
    const int8_t* __restrict__ qaux, dph_aux_layout lay) {
    __shared__ unsigned lcount[DPH_PASS_MAX], lbase[DPH_PASS_MAX];
    const int n_q = dph_gated_rows(gate, gate_base, n_q_host);
    if (n_q <= 0) return;
    const int tid = threadIdx.x;
    const unsigned claimed = *pool_head;
    const unsigned used = claimed < (unsigned)DPH_POOL_CHUNKS ? claimed : (unsigned)DPH_POOL_CHUNKS;
    for (unsigned ch = blockIdx.x; ch < used; ch += gridDim.x) {
        const unsigned raw = chunk_fill[ch];
        const unsigned cnt = raw < (unsigned)DPH_CHUNK_PAIRS ? raw : (unsigned)DPH_CHUNK_PAIRS;
        uint2* const reg = pairs + (size_t)ch * DPH_CHUNK_PAIRS;
        __syncthreads();                                    // the previous chunk's phase B is done with lbase / lcount
        for (int i = tid; i < n_q; i += 256) lcount[i] = 0;
        __syncthreads();
        // phase A: drop pairs that are not candidates of their own (list padding, outlier rows: dph_outlier_kernel adds
        // those for every query row), count the rest per query row
        uint2 mine = make_uint2(0u, 0xFFFFFFFFu);
        if ((unsigned)tid < cnt) {
            mine = reg[tid];
            const bool dead = (row_ids && row_ids[mine.x] < 0) || (n_out > 0 && is_outlier(outliers, n_out, mine.x));
            if (dead) reg[tid].y = 0xFFFFFFFFu;
            else atomicAdd(&lcount[mine.y], 1u);
        }
        __syncthreads();
        for (int i = tid; i < n_q; i += 256) {
            const unsigned c = lcount[i];
            lbase[i] = c ? atomicAdd(&bucket_counts[i], c) : 0u;
            lcount[i] = 0;
        }
        __syncthreads();
        // phase B
        const int grp = tid >> 4, l16 = tid & 15;
        for (unsigned e = grp; e < cnt; e += 16) {
            const uint2 pr = reg[e];
            if (pr.y == 0xFFFFFFFFu) continue;              // group-uniform
            const uint4* dp = (const uint4*)(db + (int64_t)pr.x * DPH_DIM + l16 * 48);
            const uint4* ap = (const uint4*)(q1 + (int64_t)(q0 + pr.y) * DPH_DIM + l16 * 48);
            const uint4* bp = (const uint4*)(q2 + (int64_t)(q0 + pr.y) * DPH_DIM + l16 * 48);
            const uint4 d[3] = {dp[0], dp[1], dp[2]};
            const uint4 a[3] = {ap[0], ap[1], ap[2]};
            const uint4 c[3] = {bp[0], bp[1], bp[2]};
            int hp = dot48(d, a);
            if (lay.n_rep > 0) hp += replica_part(db + (int64_t)pr.x * DPH_DIM, qaux + (int64_t)(q0 + pr.y) * DPH_AUX_SLOTS, lay, l16);
            const int H = sum16(hp), L = sum16(dot48(d, c));
            if (l16 == 0) {
                const unsigned idx = lbase[pr.y] + atomicAdd(&lcount[pr.y], 1u);
                if (idx < (unsigned)DPH_BUCKET_CAP)
                    buckets[(int64_t)pr.y * DPH_BUCKET_CAP + idx] = dph_make_key(128 * H + L, pr.x);
            }
        }
    }
}

// grid (ceil(n_out/16), 32): group g of workgroup (bx, by) scores outlier row 16*bx + g against query rows by, by+32, ...
// Runs BEFORE dph_refine_kernel on freshly zeroed counters.  Flat shards: outlier o takes slot o of every bucket and
// the counter starts at n_out (no atomics: n_out increments of ONE address per query row would serialise in the L2);
// list-major shards skip the rows of unprobed lists, so there the slots are handed out by atomics.
__global__ __launch_bounds__(256) void dph_outlier_kernel(
    const int8_t* __restrict__ db, const unsigned* __restrict__ outliers, int n_out, const int8_t* __restrict__ q1,
    const int8_t* __restrict__ q2, int q0, const int* __restrict__ gate, int gate_base, int n_q_host,
    const unsigned* __restrict__ tilemask, const int32_t* __restrict__ tile_list, int mask_words,
    uint64_t* __restrict__ buckets, unsigned* __restrict__ bucket_counts, const int8_t* __restrict__ qaux, dph_aux_layout lay) {
    // probe mask of the row's list: tilemask[tile][8] (masked scan) or listmask[tile_list[tile]][mask_words] (unit scan)
    const int n_q = dph_gated_rows(gate, gate_base, n_q_host);
    if (n_q <= 0) return;
    const int grp = threadIdx.x >> 4, l16 = threadIdx.x & 15;
    const int o = blockIdx.x * 16 + grp;
    if (o >= n_out) return;
    const unsigned row = outliers[o];
    const uint4* dp = (const uint4*)(db + (int64_t)row * DPH_DIM + l16 * 48);
    const uint4 d[3] = {dp[0], dp[1], dp[2]};
    const int64_t mrow = tilemask ? (int64_t)(tile_list ? tile_list[row >> 5] : (int)(row >> 5)) * mask_words : 0;
    for (int q = blockIdx.y; q < n_q; q += gridDim.y) {
        if (tilemask && !((tilemask[mrow + (q >> 5)] >> (q & 31)) & 1u)) continue;   // list not probed
        const uint4* ap = (const uint4*)(q1 + (int64_t)(q0 + q) * DPH_DIM + l16 * 48);
        const uint4* bp = (const uint4*)(q2 + (int64_t)(q0 + q) * DPH_DIM + l16 * 48);
        const uint4 a[3] = {ap[0], ap[1], ap[2]};
        const uint4 c[3] = {bp[0], bp[1], bp[2]};
        int hp = dot48(d, a);
        if (lay.n_rep > 0) hp += replica_part(db + (int64_t)row * DPH_DIM, qaux + (int64_t)(q0 + q) * DPH_AUX_SLOTS, lay, l16);
        const int H = sum16(hp), L = sum16(dot48(d, c));
        if (l16 == 0) {
            unsigned idx;
            if (tilemask) {
                idx = atomicAdd(&bucket_counts[q], 1u);
            } else {
                idx = (unsigned)o;
                if (o == 0) bucket_counts[q] = (unsigned)n_out;
            }
            if (idx < (unsigned)DPH_BUCKET_CAP) buckets[(int64_t)q * DPH_BUCKET_CAP + idx] = dph_make_key(128 * H + L, row);
        }
    }
}

void dph_launch_refine(const dph_pass& p, hipStream_t st) {
    const unsigned* outliers = p.outliers;
    const int n_out = p.n_out;
    // the bucket counts are zero here: every scan launch clears them (dph_clear_pass_counters) -- except behind an accumulating
    // scan, where the buckets already hold the fused level's keys AND the outlier rows
    if (n_out > 0 && !p.accumulate) {
        if (p.unit_recs)
            hipLaunchKernelGGL(dph_outlier_kernel, dim3((n_out + 15) / 16, 32), dim3(256), 0, st, p.db, outliers, n_out, p.q1, p.q2,
                               p.q0, p.gate, p.gate_base, p.n_q, p.listmask, p.tile_list, p.mask_words, p.buckets, p.bucket_counts, p.qaux, p.aux_lay);
        else
            hipLaunchKernelGGL(dph_outlier_kernel, dim3((n_out + 15) / 16, 32), dim3(256), 0, st, p.db, outliers, n_out, p.q1, p.q2,
                               p.q0, p.gate, p.gate_base, p.n_q, p.tilemask, (const int32_t*)nullptr, 8, p.buckets, p.bucket_counts, p.qaux, p.aux_lay);
    }
    hipLaunchKernelGGL(dph_refine_kernel, dim3(REFINE_GRID), dim3(256), 0, st, p.db, p.row_ids, p.pairs,
                       (const unsigned*)p.queue_head + 1, p.chunk_fill, p.q1, p.q2, p.q0, p.gate, p.gate_base, p.n_q, outliers, n_out,
                       p.buckets, p.bucket_counts, p.qaux, p.aux_lay);
}

// ------------------------------------------------------------------------------------------ sampled bound
// One workgroup per query row of the pass: the KP-th largest integer score in that row's bucket, minus one (rows
// scoring exactly the KP-th value must still pass), or INT_MIN when the bucket holds fewer than KP keys.  1024 threads,
// up to 32 biased scores per thread in registers; the 32 bit-steps are a latency chain, so each step is kept short:
// wave counts come from ballots + scalar pop-counts and the cross-wave sum is double-buffered (one barrier per step).
#define THR_THREADS 1024
#define THR_PER (DPH_BUCKET_CAP / THR_THREADS)
__global__ __launch_bounds__(THR_THREADS) void dph_threshold_kernel(
    const uint64_t* __restrict__ buckets, const unsigned* __restrict__ bucket_counts, int kp, const int* __restrict__ gate,
    int gate_base, int n_q_host, const int* __restrict__ floor_tau, int* __restrict__ tau_out, int* __restrict__ top_out,
    const unsigned* __restrict__ outliers, int n_out) {
    __shared__ unsigned cnt_sh[2][THR_THREADS / 64];
    __shared__ unsigned top_cnt;
    const int n_q = dph_gated_rows(gate, gate_base, n_q_host);
    const int qi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (qi >= n_q) return;
    const unsigned raw = bucket_counts[qi];
    const int n = (int)(raw < (unsigned)DPH_BUCKET_CAP ? raw : (unsigned)DPH_BUCKET_CAP);
    const int nj = (n + THR_THREADS - 1) / THR_THREADS;          // registers in use (block-uniform)
    const uint64_t* keys = buckets + (int64_t)qi * DPH_BUCKET_CAP;
    unsigned u[THR_PER];                             // statically indexed: stays in registers
#pragma unroll
    for (int j = 0; j < THR_PER; ++j) {
        const int e = tid + THR_THREADS * j;
        u[j] = (e < n) ? (unsigned)(keys[e] >> 32) : 0u;         // 0 for empty slots, >= 1 for real scores
        // OUTLIER rows do not count (round 6).  They sit in every level's bucket IN FULL (dph_outlier_kernel), not as a 1-in-stride
        // sample: where they outscore the ordinary rows -- a few hundred saturated rows -- the kp best keys were outliers and the bound
        // was the kp-th OUTLIER's score: kp rows above it in the whole shard instead of ~kp x stride.  With k <= kp that still
        // certified; beyond (k = 50: kp stays 16) no first attempt could, the retry scanned under the k-th key -- an ordinary
        // row's score level -- millions of pairs, buckets overflowed, and every row ended in the fp64 scan (seconds per batch at
        // 170 M rows; tests/test_fuzz_gpu.py's full-size round).  The bound is the kp-th best SAMPLED row, as the ladder assumes.
        if (u[j] != 0u && n_out > 0) {
            const unsigned row = dph_key_row(keys[e]);
            int lo = 0, hi = n_out;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (outliers[mid] < row) lo = mid + 1; else hi = mid; }
            if (lo < n_out && outliers[lo] == row) u[j] = 0u;
        }
    }
    unsigned ans = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned cand = ans | (1u << bit);
        unsigned c = 0;                              // wave-uniform: ballots + scalar pop-counts, no cross-lane shuffles
#pragma unroll
        for (int j = 0; j < THR_PER; ++j)
            if (j < nj) c += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(u[j] >= cand));
        // double-buffered by bit parity: one barrier per step (a wave can be at most one step ahead)
        if (lane == 0) cnt_sh[bit & 1][wv] = c;
        __syncthreads();
        unsigned total = 0;
#pragma unroll
        for (int w = 0; w < THR_THREADS / 64; ++w) total += cnt_sh[bit & 1][w];
        if (total >= (unsigned)kp) ans = cand;
    }
    if (tid == 0) {
        const int kth = (int)(ans ^ 0x80000000u);
        int t = (ans == 0u || kth == (int)0x80000000) ? (int)0x80000000 : kth - 1;
        if (floor_tau) t = max(t, floor_tau[qi]);      // a finer level only saw rows above the coarser level's bound
        if (tau_out) tau_out[qi] = t;
    }
    if (top_out) {
        // the KEEP best sampled scores themselves (any order; INT_MIN where the sample holds fewer): what a rank shares
        // so that the bound can be taken over the union of all ranks' samples (dph_union_bounds_kernel); kp == KEEP here
        if (tid == 0) top_cnt = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < THR_PER; ++j) {
            if (u[j] != 0u && u[j] >= ans) {
                const unsigned slot = atomicAdd(&top_cnt, 1u);
                if (slot < (unsigned)DPH_SAMPLE_KEEP) top_out[(int64_t)qi * DPH_SAMPLE_KEEP + slot] = (int)(u[j] ^ 0x80000000u);
            }
        }
        __syncthreads();
        for (unsigned t = top_cnt + tid; t < (unsigned)DPH_SAMPLE_KEEP; t += THR_THREADS)
            top_out[(int64_t)qi * DPH_SAMPLE_KEEP + t] = (int)0x80000000;
    }
}

void dph_launch_threshold(const dph_pass& p, int kp, const int* floor_tau, int* tau_out, int* top_out, hipStream_t st) {
    hipLaunchKernelGGL(dph_threshold_kernel, dim3(p.unit_recs ? p.n_q : DPH_QROWS * p.qb), dim3(THR_THREADS), 0, st, p.buckets, p.bucket_counts,
                       top_out ? DPH_SAMPLE_KEEP : kp, p.gate, p.gate_base, p.n_q, floor_tau, tau_out, top_out, p.outliers, p.n_out);
}

// bound over the union of n_parts samples: the KEEP-th largest of the n_parts*KEEP shared scores of a row, minus one
// (INT_MIN when the union holds fewer).  One wave per row, rank by counting.
__global__ __launch_bounds__(64) void dph_union_bounds_kernel(const int* __restrict__ top_parts, int n_parts,
                                                              int64_t n, int* __restrict__ tau_out) {
    const int64_t row = blockIdx.x;
    const int lane = threadIdx.x, m = n_parts * DPH_SAMPLE_KEEP;
    int best = (int)0x80000000;
    for (int c = lane; c < m; c += 64) {
        const int p = c / DPH_SAMPLE_KEEP, i = c % DPH_SAMPLE_KEEP;
        const int v = top_parts[((int64_t)p * n + row) * DPH_SAMPLE_KEEP + i];
        if (v == (int)0x80000000) continue;
        int rank = 0;                                  // entries strictly better, ties broken by position
        for (int u = 0; u < m; ++u) {
            const int w = top_parts[((int64_t)(u / DPH_SAMPLE_KEEP) * n + row) * DPH_SAMPLE_KEEP + (u % DPH_SAMPLE_KEEP)];
            rank += (w != (int)0x80000000 && (w > v || (w == v && u < c))) ? 1 : 0;
        }
        if (rank == DPH_SAMPLE_KEEP - 1) best = v - 1; // exactly one entry has this rank
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o));
    if (lane == 0) tau_out[row] = best;
}

void dph_launch_union_bounds(const int* top_parts, int n_parts, int64_t n, int* tau_out, hipStream_t st) {
    hipLaunchKernelGGL(dph_union_bounds_kernel, dim3((unsigned)n), dim3(64), 0, st, top_parts, n_parts, n, tau_out);
}
