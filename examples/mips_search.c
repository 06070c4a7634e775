/* mips_search.c -- the hot path of densephrases/index.py (MIPS.search: the FAISS search of :200, MIPS.get_idxs of :124-141 and the
 * start/end window re-scoring of :323-370) driven from PLAIN C through include/dph.h: what a host in another language binds
 * (INTEGRATION.md shows the ctypes / cgo / JNI stubs for the same calls).  No torch, no python.
 *
 *   gcc -std=c11 -O2 -I include examples/mips_search.c -L densephrases_amd/csrc -ldph -Wl,-rpath,$PWD/densephrases_amd/csrc -lm -o mips_search
 *   ./mips_search [rows = 2000000] [queries = 4]          (needs an MI355X; tests/test_abi.py compiles and links it on CPU)
 *
 * The shard is the deterministic synthetic dump of BASELINE.md (documents of 100 rows, every token kept); a query is a stored row
 * de-quantised again, so the nearest neighbour of query i is known: row 12345 + 1000 i comes back first. */
#include <stdio.h>
#include <stdlib.h>

#include "dph.h"

#define CHECK(call)                                                                              \
    do {                                                                                         \
        const int rc_ = (call);                                                                  \
        if (rc_ != DPH_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, dph_last_error()); return 1; } \
    } while (0)

int main(int argc, char** argv) {
    const int64_t n_rows = argc > 1 ? atoll(argv[1]) : 2000000;
    const int n_q = argc > 2 ? atoi(argv[2]) : 4, k = 10, L = 10;
    if (dph_abi_version() != DPH_ABI_VERSION) { fprintf(stderr, "libdph ABI %d, header %d\n", dph_abi_version(), DPH_ABI_VERSION); return 1; }
    if (dph_device_count() < 1) { fprintf(stderr, "no GPU: the product path has no CPU fallback\n"); return 1; }

    /* ---- MIPS.__init__ (index.py:24-88): the shard, idx2id, f2o */
    dph_index* h = NULL;
    CHECK(dph_index_create(0, n_rows, 0, &h));
    CHECK(dph_index_fill_synthetic(h, 1234, NULL));                  /* a real host: dph_index_upload_rows of the int8 dump */
    const int64_t n_docs = (n_rows + 99) / 100;
    int32_t* doc = malloc(sizeof(int32_t) * n_rows), *word = malloc(sizeof(int32_t) * n_rows);
    int32_t* doc_ids = malloc(sizeof(int32_t) * n_docs), *f2o = malloc(sizeof(int32_t) * n_docs * 100);
    int64_t* f2o_off = malloc(sizeof(int64_t) * (n_docs + 1));
    if (!doc || !word || !doc_ids || !f2o || !f2o_off) return 1;
    for (int64_t r = 0; r < n_rows; ++r) { doc[r] = (int32_t)(r / 100); word[r] = (int32_t)(r % 100); }
    for (int64_t d = 0; d < n_docs; ++d) { doc_ids[d] = (int32_t)d; f2o_off[d] = 100 * d; for (int t = 0; t < 100; ++t) f2o[100 * d + t] = t; }
    f2o_off[n_docs] = 100 * n_docs;
    CHECK(dph_index_set_idx2id(h, doc, word));
    CHECK(dph_index_set_f2o(h, n_docs, doc_ids, f2o_off, f2o));
    CHECK(dph_index_finalize(h, NULL));

    /* ---- queries: [start | end] halves, both the de-quantised row 12345 + 1000 i */
    float* q_start = malloc(sizeof(float) * n_q * DPH_DIM), *q_end = malloc(sizeof(float) * n_q * DPH_DIM);
    float* x = malloc(sizeof(float) * 2 * n_q * DPH_DIM);            /* the 2 B rows MIPS.search_dense searches (index.py:193-199) */
    if (!q_start || !q_end || !x) return 1;
    for (int i = 0; i < n_q; ++i) {
        CHECK(dph_reconstruct(h, (12345 + 1000 * (int64_t)i) % n_rows, q_start + (size_t)i * DPH_DIM));
        for (int j = 0; j < DPH_DIM; ++j) {
            q_end[(size_t)i * DPH_DIM + j] = q_start[(size_t)i * DPH_DIM + j];
            x[(size_t)i * DPH_DIM + j] = x[(size_t)(n_q + i) * DPH_DIM + j] = q_start[(size_t)i * DPH_DIM + j];
        }
    }

    /* ---- faiss Index.search (index.py:200): exact, certified, (score desc, id asc) */
    float* D = malloc(sizeof(float) * 2 * n_q * k);
    int64_t* I = malloc(sizeof(int64_t) * 2 * n_q * k);
    CHECK(dph_search(h, x, 2 * n_q, k, D, I));
    dph_search_stats st;
    CHECK(dph_search_get_stats(h, &st));
    printf("%d query rows: %d certified exact by the first int8 scan, %d by the on-device retry, %d by the fp64 scan, %d uncertified\n", st.rows,
           st.certified_fast, st.certified_wide, st.exact_fallback, st.uncertified);

    /* ---- MIPS.get_idxs (index.py:124-141) and "find end for start" (:323-346) over the start candidates */
    int32_t* cdoc = malloc(sizeof(int32_t) * n_q * k), *cword = malloc(sizeof(int32_t) * n_q * k);
    int32_t* pred_end = malloc(sizeof(int32_t) * n_q * k), *slot = malloc(sizeof(int32_t) * n_q * k);
    double* best = malloc(sizeof(double) * n_q * k);
    CHECK(dph_id2docword(h, I, (int64_t)n_q * k, cdoc, cword));
    CHECK(dph_rescore(h, 0, q_end, n_q, k, L, I, cdoc, cword, D, pred_end, best, slot, NULL));
    for (int i = 0; i < n_q; ++i)
        printf("query %d: first id %lld (doc %d, words %d..%d), first-stage score %.3f, start+end score %.3f\n", i, (long long)I[(size_t)i * k],
               cdoc[(size_t)i * k], cword[(size_t)i * k], pred_end[(size_t)i * k], D[(size_t)i * k], best[(size_t)i * k]);
    int ok = 1;
    for (int i = 0; i < n_q; ++i) ok &= I[(size_t)i * k] == (12345 + 1000 * (int64_t)i) % n_rows;
    printf("%s\n", ok ? "every query found the row it was made from" : "MISMATCH");
    CHECK(dph_index_destroy(h));
    return ok ? 0 : 2;
}
