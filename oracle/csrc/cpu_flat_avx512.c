/* cpu_flat_avx512.c -- TEST / MEASUREMENT INFRASTRUCTURE (oracle/): the CPU comparator of bench.py's `cpu_baseline` as plain C.
 * FAISS-CPU IndexFlatIP.search (densephrases/index.py:200 on the fp32 de-quantised dump, embed_utils.py:148) restated the way FAISS
 * executes it -- the database walked in blocks by one OpenMP thread per core, an sgemm micro-kernel per block, a running top-k per
 * thread, merged at the end -- with the micro-kernel written for AVX-512 (FAISS links MKL / OpenBLAS for it; neither is usable here:
 * numpy's OpenBLAS has no AVX-512 path in this image and torch's MKL crawls on the box's 256 threads).
 * The product never links or loads this; it is compiled by __graft_entry__.build() into oracle/_cbuild/ for bench.py and the tests.
 *
 *   scores[q] (q < nq <= 128 .. any multiple of 16) of THREE database rows at a time: acc[3][nq / 16] zmm accumulators,
 *   for every dimension j: one 64-byte load of the query panel's row j per 16 queries (the panel is k-major, [768][nq], L2-resident),
 *   three broadcasts of the rows' x_j, three FMAs per panel load.  A row's scores are compared with the queries' current k-th best
 *   (one vector compare per 16 queries); only the rare score above it takes the scalar insertion path.
 * Ties: (score desc, id asc), like the oracle (oracle/mips_oracle.py flat_ip_search).  fp32 accumulation in dimension order.      */
#include <immintrin.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define DIM 768
#define MAXQ 256

typedef struct { float s; int64_t id; } hit;

/* insert (s, id) into the descending list best[0..k), (score desc, id asc); returns the new k-th score */
static inline float insert_hit(hit* best, int k, float s, int64_t id) {
    int p = k - 1;
    if (!(s > best[p].s || (s == best[p].s && id < best[p].id))) return best[k - 1].s;
    while (p > 0 && (s > best[p - 1].s || (s == best[p - 1].s && id < best[p - 1].id))) { best[p] = best[p - 1]; --p; }
    best[p].s = s; best[p].id = id;
    return best[k - 1].s;
}

/* db [n][768] fp32 row-major, qt [768][nq] fp32 (the queries, k-major), nq a multiple of 16 and <= MAXQ, k <= 64.
 * out_s / out_i [nq][k]: the k best per query over rows [0, n), ids = id_base + row.  Returns 0, or -1 on bad arguments.      */
int cpu_flat_ip_topk(const float* db, int64_t n, int64_t id_base, const float* qt, int nq, int k, float* out_s, int64_t* out_i, int threads) {
    if (!db || !qt || !out_s || !out_i || n < 0 || nq <= 0 || nq > MAXQ || (nq & 15) || k <= 0 || k > 64) return -1;
    const int nv = nq / 16;
    if (threads <= 0) threads = omp_get_max_threads();
    hit* all = (hit*)malloc((size_t)threads * nq * k * sizeof(hit));
    if (!all) return -1;
#pragma omp parallel num_threads(threads)
    {
        const int t = omp_get_thread_num(), T = omp_get_num_threads();
        hit* best = all + (size_t)t * nq * k;
        float thr[MAXQ] __attribute__((aligned(64)));
        for (int i = 0; i < nq * k; ++i) { best[i].s = -3.4028234663852886e38f; best[i].id = INT64_MAX; }
        for (int q = 0; q < nq; ++q) thr[q] = -3.4028234663852886e38f;
        /* contiguous share of the rows, in units of 3 */
        const int64_t per = ((n + T - 1) / T + 2) / 3 * 3;
        const int64_t r_lo = (int64_t)t * per < n ? (int64_t)t * per : n, r_hi = r_lo + per < n ? r_lo + per : n;
        for (int64_t r = r_lo; r < r_hi; r += 3) {
            const int rows = r_hi - r >= 3 ? 3 : (int)(r_hi - r);
            const float* x0 = db + r * DIM;
            const float* x1 = rows > 1 ? x0 + DIM : x0;
            const float* x2 = rows > 2 ? x0 + 2 * DIM : x0;
            __m512 a0[MAXQ / 16], a1[MAXQ / 16], a2[MAXQ / 16];
            for (int v = 0; v < nv; ++v) { a0[v] = _mm512_setzero_ps(); a1[v] = _mm512_setzero_ps(); a2[v] = _mm512_setzero_ps(); }
            if (nv == 8) {           /* the batch of 64 (128 query rows): everything in the 32 registers */
                __m512 b0 = a0[0], b1 = a0[1], b2 = a0[2], b3 = a0[3], b4 = a0[4], b5 = a0[5], b6 = a0[6], b7 = a0[7];
                __m512 c0 = b0, c1 = b0, c2 = b0, c3 = b0, c4 = b0, c5 = b0, c6 = b0, c7 = b0;
                __m512 d0 = b0, d1 = b0, d2 = b0, d3 = b0, d4 = b0, d5 = b0, d6 = b0, d7 = b0;
                for (int j = 0; j < DIM; ++j) {
                    const float* qj = qt + (size_t)j * 128;
                    const __m512 u0 = _mm512_set1_ps(x0[j]), u1 = _mm512_set1_ps(x1[j]), u2 = _mm512_set1_ps(x2[j]);
                    __m512 p;
                    p = _mm512_loadu_ps(qj);       b0 = _mm512_fmadd_ps(u0, p, b0); c0 = _mm512_fmadd_ps(u1, p, c0); d0 = _mm512_fmadd_ps(u2, p, d0);
                    p = _mm512_loadu_ps(qj + 16);  b1 = _mm512_fmadd_ps(u0, p, b1); c1 = _mm512_fmadd_ps(u1, p, c1); d1 = _mm512_fmadd_ps(u2, p, d1);
                    p = _mm512_loadu_ps(qj + 32);  b2 = _mm512_fmadd_ps(u0, p, b2); c2 = _mm512_fmadd_ps(u1, p, c2); d2 = _mm512_fmadd_ps(u2, p, d2);
                    p = _mm512_loadu_ps(qj + 48);  b3 = _mm512_fmadd_ps(u0, p, b3); c3 = _mm512_fmadd_ps(u1, p, c3); d3 = _mm512_fmadd_ps(u2, p, d3);
                    p = _mm512_loadu_ps(qj + 64);  b4 = _mm512_fmadd_ps(u0, p, b4); c4 = _mm512_fmadd_ps(u1, p, c4); d4 = _mm512_fmadd_ps(u2, p, d4);
                    p = _mm512_loadu_ps(qj + 80);  b5 = _mm512_fmadd_ps(u0, p, b5); c5 = _mm512_fmadd_ps(u1, p, c5); d5 = _mm512_fmadd_ps(u2, p, d5);
                    p = _mm512_loadu_ps(qj + 96);  b6 = _mm512_fmadd_ps(u0, p, b6); c6 = _mm512_fmadd_ps(u1, p, c6); d6 = _mm512_fmadd_ps(u2, p, d6);
                    p = _mm512_loadu_ps(qj + 112); b7 = _mm512_fmadd_ps(u0, p, b7); c7 = _mm512_fmadd_ps(u1, p, c7); d7 = _mm512_fmadd_ps(u2, p, d7);
                }
                a0[0] = b0; a0[1] = b1; a0[2] = b2; a0[3] = b3; a0[4] = b4; a0[5] = b5; a0[6] = b6; a0[7] = b7;
                a1[0] = c0; a1[1] = c1; a1[2] = c2; a1[3] = c3; a1[4] = c4; a1[5] = c5; a1[6] = c6; a1[7] = c7;
                a2[0] = d0; a2[1] = d1; a2[2] = d2; a2[3] = d3; a2[4] = d4; a2[5] = d5; a2[6] = d6; a2[7] = d7;
            } else {
                for (int j = 0; j < DIM; ++j) {
                    const float* qj = qt + (size_t)j * nq;
                    const __m512 u0 = _mm512_set1_ps(x0[j]), u1 = _mm512_set1_ps(x1[j]), u2 = _mm512_set1_ps(x2[j]);
                    for (int v = 0; v < nv; ++v) {
                        const __m512 p = _mm512_loadu_ps(qj + 16 * v);
                        a0[v] = _mm512_fmadd_ps(u0, p, a0[v]); a1[v] = _mm512_fmadd_ps(u1, p, a1[v]); a2[v] = _mm512_fmadd_ps(u2, p, a2[v]);
                    }
                }
            }
            for (int rr = 0; rr < rows; ++rr) {
                const __m512* a = rr == 0 ? a0 : (rr == 1 ? a1 : a2);
                for (int v = 0; v < nv; ++v) {
                    /* >= : an equal score with a lower id would still enter (ids ascend inside a thread, so only across threads: the merge decides) */
                    __mmask16 m = _mm512_cmp_ps_mask(a[v], _mm512_load_ps(thr + 16 * v), _CMP_GT_OQ);
                    if (m) {
                        float s[16] __attribute__((aligned(64)));
                        _mm512_store_ps(s, a[v]);
                        while (m) {
                            const int b = __builtin_ctz(m);
                            m &= (__mmask16)(m - 1);
                            const int q = 16 * v + b;
                            thr[q] = insert_hit(best + (size_t)q * k, k, s[b], id_base + r + rr);
                        }
                    }
                }
            }
        }
    }
    /* merge the threads' lists (score desc, id asc) */
    for (int q = 0; q < nq; ++q) {
        hit m[64];
        for (int i = 0; i < k; ++i) { m[i].s = -3.4028234663852886e38f; m[i].id = INT64_MAX; }
        for (int t = 0; t < threads; ++t) {
            const hit* b = all + ((size_t)t * nq + q) * k;
            for (int i = 0; i < k; ++i) if (b[i].id != INT64_MAX) insert_hit(m, k, b[i].s, b[i].id);
        }
        for (int i = 0; i < k; ++i) { out_s[(size_t)q * k + i] = m[i].s; out_i[(size_t)q * k + i] = m[i].id == INT64_MAX ? -1 : m[i].id; }
    }
    free(all);
    return 0;
}

int cpu_flat_has_avx512(void) { return __builtin_cpu_supports("avx512f") ? 1 : 0; }
int cpu_flat_max_threads(void) { return omp_get_max_threads(); }
