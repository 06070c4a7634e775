// dph_api.hip -- the C ABI of libdph (include/dph.h): handle management, host<->HBM plumbing, and the
// search driver (quantise -> sampled bounds -> int8 filter scan -> refine -> select/certify -> on-device retry ->
// fp64 fallback).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <utility>
#include <vector>

#include "dph_internal.h"

static thread_local std::string g_err;
// (a HIP call that failed leaves its code as the thread's "last error" until somebody reads it: rocPRIM checks hipGetLastError() after its
// launches and would report the stale code of a REFUSED call -- dph_index_create on a device that does not exist -- as the failure of the
// next index's finalize.  An error that has been turned into a return code is consumed here.)
static int fail(int code, const std::string& msg) {
    g_err = msg;
    if (code == DPH_E_HIP || code == DPH_E_NOMEM) (void)hipGetLastError();
    return code;
}
#define HIPCHK(expr)                                                                                 \
    do {                                                                                             \
        hipError_t e_ = (expr);                                                                      \
        if (e_ != hipSuccess)                                                                        \
            return fail(e_ == hipErrorOutOfMemory ? DPH_E_NOMEM : DPH_E_HIP,                         \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                          \
    } while (0)

struct dph_index {
    int device = 0;
    int64_t n_rows = 0, n_tiles = 0, id_base = 0;   // n_rows = stored rows (incl. list padding on IVF shards)
    int64_t n_ids = 0;                   // ids the shard answers for: [id_base, id_base + n_ids)
    int8_t* db = nullptr;                // [n_tiles*32, 768] int8, padding rows zero
    // list-major (IVF) shards: row <-> id maps, per-tile list ids, centroids, probe masks
    int64_t* row_ids = nullptr;          // [n_rows] global id of a stored row, -1 = padding
    int32_t* inv_row = nullptr;          // [n_ids]  stored row of local id
    // shards merged from several sub-indexes: ids = id_offsets[g] + (row - row_starts[g])
    std::vector<int64_t> h_id_offsets, h_row_starts;
    int64_t* id_offsets = nullptr; int64_t* row_starts = nullptr;
    std::vector<int32_t> h_inv;
    int nlist = 0;
    float* centroids = nullptr;          // [nlist, 768] fp32
    float* coarse_scores = nullptr;      // [coarse_rows, nlist] fp32: query x centroid scores of the current pass
    void* coarse_cs = nullptr;           // candidate scratch of the one-pass probe selection (dph_launch_coarse allocates it on first use)
    int coarse_rows = 0;
    double cnorm_max = 0.0;              // max_l || c_l ||_2 (error band of the fp32 coarse scores)
    int default_nprobe = 0;              // > 0: entry points without an nprobe argument search IVF (tuning key "nprobe")
    int32_t* tile_list = nullptr;        // [n_tiles]
    unsigned* listmask = nullptr;        // [nlist][8]   bit j of word w: query row 32w+j of the pass probes the list
    unsigned* tilemask = nullptr;        // [n_tiles][8] the same per tile (what the scan reads)
    unsigned* onesmask = nullptr;        // [n_tiles][8] all ones: exact (flat) search over a list-major shard
    // unit scan (lists stored as contiguous runs of tiles): the work queue of a pass of up to DPH_PASS_MAX query rows
    bool lists_contiguous = false;
    int* list_tile0 = nullptr;           // [nlist+1] first tile of every list
    int2* unit_offsets = nullptr;        // [nlist] first chunk / first unit of every list in the current pass
    int max_list_tiles = 0; int64_t total_segs = 0;
    int ivf_spread = 1;                  // tuning key "ivf_spread": deal a chunk's query rows over the four scan waves first
    int ivf_units = -1;                  // tuning key "ivf_units": -1 = when the lists are long enough, 0 = never, 1 = always
    unsigned* listmask_u = nullptr;      // [nlist][DPH_UNIT_WORDS]
    int* unit_counts = nullptr;          // [4] chunks, units, error, spare + [DPH_UNIT_LAUNCHES] work-queue heads
    int* slot_q = nullptr; int4* unit_recs = nullptr; int4* unit_list_recs = nullptr; int8_t* unit_frags = nullptr;
    int chunk_cap = 0, unit_cap = 0;
    float offset = -2.f, scale = 20.f;
    float lut_host[256];
    float* lut_dev = nullptr;
    double delta_max = 0.0;              // max_n | x32(n) - (n/scale + offset) |
    double rmax = 0.0;                   // max over NON-outlier rows of || n - mu ||_2: the certificate's shard constant
    double rmax_all = 0.0;               // the same over every row
    double rmed = 0.0;                   // median of the same
    // per-dimension mean codes of the stored rows (integers): the centre of every Cauchy-Schwarz bound of the search
    int mu_host[DPH_DIM] = {0};
    double sd_host[DPH_DIM] = {0};       // per-dimension spread (what the rogue dimensions are told from)
    int* mu_dev = nullptr; long long* colsum_dev = nullptr;
    // aux rows (dph_scan.hip "the aux k-step"): per-row norm codes + raw codes of the rogue dimensions
    int8_t* aux = nullptr; size_t aux_bytes = 0;
    dph_aux_layout aux_lay{};            // stride 0: the shard has none (its rows are alike: one norm bound serves them all)
    int norm_unit = 1;                   // a norm code counts this many units of || n - mu ||
    int aux_mode = -1;                   // tuning key "aux": -1 = decided by dph_index_finalize, 0 = never, 4 = norm codes, 16 / 32 = norm codes + 12 / 24 replica slots
    bool aux_lay_forced = false;         // the layout was set by dph_index_set_aux_layout (a sharded job: every rank the same digits)
    unsigned* outliers = nullptr;        // sorted stored-row indices of the rows above the cut (always candidates)
    int n_out = 0;
    int n_out_found = 0; double rmax_cut = 0.0;   // what finalize found; in force only on a shard WITHOUT aux rows (derive_aux_units)
    bool finalized = false;
    // idx2id + f2o CSR
    int32_t *row2doc = nullptr, *row2word = nullptr;
    std::vector<int32_t> h_row2doc, h_row2word;
    int32_t* doc_ids = nullptr; int64_t* f2o_off = nullptr; int32_t* f2o = nullptr; int64_t n_docs = 0;
    // tuning (dph_index_set_tuning)
    std::vector<int> ladder;             // explicit pre-pass strides, coarse -> fine; empty = derived from the shard size
    int fine_stride = 0;                 // 0 = default (32 on shards >= 100 M rows, 16 below)
    int sample_kp = 16;                  // the bound of a level = its kp-th best sampled score
    int max_qb = DPH_MAX_QB;
    // search scratch (grown on demand)
    int grid = 256;
    int64_t cap_rows = 0;                // query rows the per-call scratch is sized for
    struct qimg { float* x = nullptr; int8_t* frag = nullptr; int8_t* q1 = nullptr; int8_t* q2 = nullptr;
                  dph_qinfo* qinfo = nullptr; int* lmax = nullptr; int8_t* qaux = nullptr; int32_t* nf = nullptr; };
    qimg q_main, q_retry;                // q_main.x: the query rows as every stage after the quantiser reads them (non-finite rows replaced by
                                         // their stand-in, q_main.nf flags them: dph_quantize_kernel); also the staging copy of the host entry points
    float* D_dev = nullptr; int64_t* I_dev = nullptr; int32_t* status_dev = nullptr;   // host-pointer entry points
    int cap_k = 0;
    int32_t *ik_dev = nullptr, *fail_dev = nullptr, *fail2_dev = nullptr, *retry_rows = nullptr, *exact_rows = nullptr;
    int* retry_tau = nullptr;
    int* counters = nullptr;             // [0] rows failing the first attempt, [1] failing the retry, [2..3] spare
    float* exact_x = nullptr;
    void* exact_scratch = nullptr; size_t exact_bytes = 0;
    // per-pass scratch (fixed size)
    uint2* pairs = nullptr; unsigned* chunk_fill = nullptr; unsigned* wave_counts = nullptr; uint64_t* buckets = nullptr;
    unsigned* bucket_counts = nullptr;
    unsigned* counts_raw = nullptr;      // the allocation bucket_counts lives in
    int seg_tiles = 64;                  // tuning key "scan_seg": shortest work-queue segment of the flat scan, in tiles
    int ladder_fuse = 1;                 // tuning key "ladder_fuse": the full scan skips the tiles the finest ladder level scanned
    int retry_chain = 1;                 // tuning key "retry_chain": 0 = the first attempt only (rows it cannot certify come back with status 1)
    int scan_sched[2] = {1, 1};          // tuning key "scan_sched" (qb 1, qb 2): hand-over schedule of the flat full scan (dph_scan.hip); 1 = wave after
                                         // wave: 20.25 vs 20.47 ms (128 rows) and 29.4 vs 30.1 ms (256 rows) per 170 M-row launch (profiles/r04_scan_scheds_170M.json)
    int* tau_dev = nullptr;              // [2][256] per-row bounds of the current pass (ladder ping-pong)
    unsigned long long* norm_dev = nullptr; unsigned* hist_dev = nullptr;
    long long* kmeans_sums = nullptr; int kmeans_nlist = 0;      // [nlist,768] integer sums of a k-means update
    dph_search_stats stats{};
    bool stats_pending = false;          // device counters of the last device-pointer call not read back yet
    int64_t stats_rows = 0;
    // measurement hook: event pairs around scan launches
    bool profile = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;          // around the full-scan launches
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events_ladder;   // around the scan launches of the ladder levels
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_free;
    // the reference's own index type (IndexPreTransform + IndexIVFPQ) instead of raw int8 rows: dph_pq.hip
    dph_pq* pq = nullptr;
    // a TWIN (dph_index_create_twin): a second handle over the rows, metadata and shard constants of `twin_of` with search scratch of
    // its own, so that two batches can be in flight on the same shard (the pipelined search below); owns nothing but that scratch
    dph_index* twin_of = nullptr;
    int n_twins = 0;                     // twins alive over this handle (it must outlive them and stay as it is)
    // the two-stage form of a search (dph_search_prepare_dev / dph_search_finish_dev): what the prepare stage left behind
    int side_grid = 0;                   // tuning key "side_grid": scan workgroups of the prepare stage's sampled levels (0 = grid)
    struct prepared { bool valid = false; int64_t n = 0; int k = 0; const int* tau = nullptr; int fuse_stride = 0; int qb = 1; } prep;
};

static void build_lut(dph_index* h) {
    // the reference's fp32 de-quantisation, two roundings: fl(fl(n)/scale) + offset  (embed_utils.py:148-149)
    double dm = 0.0;
    for (int n = -128; n < 128; ++n) {
        volatile float a = (float)n / h->scale;
        volatile float b = a + h->offset;
        h->lut_host[n + 128] = b;
        const double exact = (double)n / (double)h->scale + (double)h->offset;
        dm = fmax(dm, fabs((double)b - exact));
    }
    h->delta_max = dm;
}

// the rows, metadata and shard constants of an index that has twins (or is one) are shared and stay as they are
#define DPH_NOT_TWINNED(h, who) \
    do { if ((h) && ((h)->twin_of || (h)->n_twins > 0)) return fail(DPH_E_STATE, std::string(who) + ": not on an index that has twins or is one"); } while (0)

extern "C" {

int dph_abi_version(void) { return DPH_ABI_VERSION; }
const char* dph_last_error(void) { return g_err.c_str(); }
int dph_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int dph_index_create(int device, int64_t n_rows, int64_t id_base, dph_index** out) {
    if (!out || n_rows < 0) return fail(DPH_E_ARG, "dph_index_create: bad arguments");
    if (n_rows >= (int64_t)0xFFFFFFF0ll) return fail(DPH_E_ARG, "dph_index_create: shard limited to 2^32-16 rows");
    HIPCHK(hipSetDevice(device));
    dph_index* h = new dph_index();
    h->device = device;
    h->n_rows = n_rows;
    h->n_ids = n_rows;
    h->id_base = id_base;
    h->n_tiles = (n_rows + DPH_TILE_ROWS - 1) / DPH_TILE_ROWS;
    h->grid = dph_scan_grid(device);
    const size_t bytes = (size_t)(h->n_tiles > 0 ? h->n_tiles : 1) * DPH_TILE_BYTES;
    hipError_t e = hipMalloc((void**)&h->db, bytes);
    if (e != hipSuccess) { delete h; return fail(DPH_E_NOMEM, std::string("hipMalloc shard: ") + hipGetErrorString(e)); }
    // zero the padding rows of the last tile (they are excluded from the candidates by row index as well)
    if (h->n_tiles > 0) {
        const size_t used = (size_t)n_rows * DPH_DIM;
        if (bytes > used) (void)hipMemset(h->db + used, 0, bytes - used);
    }
    build_lut(h);
    if (hipMalloc((void**)&h->lut_dev, 256 * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&h->norm_dev, 2 * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc((void**)&h->hist_dev, DPH_NORM_BINS * sizeof(unsigned)) != hipSuccess ||
        hipMalloc((void**)&h->outliers, DPH_OUTLIER_MAX * sizeof(unsigned)) != hipSuccess ||
        hipMalloc((void**)&h->mu_dev, DPH_DIM * sizeof(int)) != hipSuccess ||
        hipMalloc((void**)&h->colsum_dev, 2 * DPH_DIM * sizeof(long long)) != hipSuccess) {
        dph_index_destroy(h);
        return fail(DPH_E_NOMEM, "hipMalloc lut");
    }
    (void)hipMemset(h->mu_dev, 0, DPH_DIM * sizeof(int));
    h->aux_lay.q2max = 64;
    (void)hipMemcpy(h->lut_dev, h->lut_host, sizeof(h->lut_host), hipMemcpyHostToDevice);
    *out = h;
    return DPH_OK;
}

static void free_qimg(dph_index::qimg& q) {
    void* p[] = {q.x, q.frag, q.q1, q.q2, q.qinfo, q.lmax, q.qaux, q.nf};
    for (void* v : p) if (v) (void)hipFree(v);
    q = dph_index::qimg();
}

// what a twin owns: the search scratch (ensure_scratch) and nothing else
static void free_search_scratch(dph_index* h) {
    free_qimg(h->q_main);
    free_qimg(h->q_retry);
    void* ptrs[] = {h->D_dev, h->I_dev, h->status_dev, h->ik_dev, h->fail_dev, h->fail2_dev, h->retry_rows, h->exact_rows, h->retry_tau,
                    h->counters, h->exact_x, h->exact_scratch, h->pairs, h->chunk_fill, h->wave_counts, h->buckets, h->counts_raw, h->tau_dev};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (auto& ev : h->prof_events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (auto& ev : h->prof_events_ladder) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (auto& ev : h->prof_free) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
}

int dph_index_destroy(dph_index* h) {
    if (!h) return DPH_OK;
    (void)hipSetDevice(h->device);
    if (h->twin_of) {
        free_search_scratch(h);
        h->twin_of->n_twins--;
        delete h;
        return DPH_OK;
    }
    if (h->n_twins > 0) return fail(DPH_E_STATE, "dph_index_destroy: destroy the twins of this index first");
    free_qimg(h->q_main);
    free_qimg(h->q_retry);
    void* ptrs[] = {h->db, h->lut_dev, h->row2doc, h->row2word, h->doc_ids, h->f2o_off, h->f2o, h->D_dev, h->I_dev,
                    h->status_dev, h->ik_dev, h->fail_dev, h->fail2_dev, h->retry_rows, h->exact_rows, h->retry_tau,
                    h->counters, h->exact_x, h->exact_scratch, h->kmeans_sums, h->pairs, h->chunk_fill, h->wave_counts, h->buckets, h->counts_raw,
                    h->tau_dev, h->norm_dev, h->hist_dev, h->outliers, h->mu_dev, h->colsum_dev, h->aux, h->row_ids, h->inv_row, h->id_offsets, h->row_starts, h->centroids, h->tile_list,
                    h->listmask, h->tilemask, h->onesmask, h->coarse_scores, h->list_tile0, h->listmask_u, h->unit_counts, h->unit_offsets,
                    h->slot_q, h->unit_recs, h->unit_list_recs, h->unit_frags, h->coarse_cs};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (auto& ev : h->prof_events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (auto& ev : h->prof_events_ladder) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (auto& ev : h->prof_free) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    if (h->pq) dph_pq_free(h->pq);
    delete h;
    return DPH_OK;
}

int dph_index_create_twin(dph_index* src, dph_index** out) {
    if (!src || !out) return fail(DPH_E_ARG, "dph_index_create_twin: null");
    if (src->twin_of) return fail(DPH_E_ARG, "dph_index_create_twin: make twins of the index itself, not of a twin");
    if (!src->finalized) return fail(DPH_E_STATE, "dph_index_create_twin: call dph_index_finalize first");
    if (src->pq || src->row_ids) return fail(DPH_E_STATE, "dph_index_create_twin: flat shards only (no PQ index, no list-major shard)");
    // (the host copies of idx2id stay with the index: they are large and only its host entry points read them)
    std::vector<int32_t> keep_doc, keep_word, keep_inv;
    keep_doc.swap(src->h_row2doc); keep_word.swap(src->h_row2word); keep_inv.swap(src->h_inv);
    dph_index* t = new dph_index(*src);              // every pointer to rows / metadata / constants is shared ...
    keep_doc.swap(src->h_row2doc); keep_word.swap(src->h_row2word); keep_inv.swap(src->h_inv);
    t->twin_of = src;
    t->n_twins = 0;
    // ... and the search scratch is the twin's own (allocated by its first search)
    t->q_main = dph_index::qimg(); t->q_retry = dph_index::qimg();
    t->D_dev = nullptr; t->I_dev = nullptr; t->status_dev = nullptr; t->cap_rows = 0; t->cap_k = 0;
    t->ik_dev = nullptr; t->fail_dev = nullptr; t->fail2_dev = nullptr; t->retry_rows = nullptr; t->exact_rows = nullptr;
    t->retry_tau = nullptr; t->counters = nullptr; t->exact_x = nullptr; t->exact_scratch = nullptr; t->exact_bytes = 0;
    t->pairs = nullptr; t->chunk_fill = nullptr; t->wave_counts = nullptr; t->buckets = nullptr; t->bucket_counts = nullptr;
    t->counts_raw = nullptr; t->tau_dev = nullptr;
    t->coarse_scores = nullptr; t->coarse_cs = nullptr; t->coarse_rows = 0; t->kmeans_sums = nullptr; t->kmeans_nlist = 0;
    t->stats = dph_search_stats{}; t->stats_pending = false; t->profile = false;
    t->prof_events.clear(); t->prof_events_ladder.clear(); t->prof_free.clear();
    t->prep = dph_index::prepared();
    src->n_twins++;
    *out = t;
    return DPH_OK;
}

int dph_index_set_codec(dph_index* h, float offset, float scale) {
    DPH_NOT_TWINNED(h, "dph_index_set_codec");
    if (!h || !(scale > 0.f)) return fail(DPH_E_ARG, "dph_index_set_codec: scale must be > 0");
    HIPCHK(hipSetDevice(h->device));
    h->offset = offset; h->scale = scale;
    build_lut(h);
    HIPCHK(hipMemcpy(h->lut_dev, h->lut_host, sizeof(h->lut_host), hipMemcpyHostToDevice));
    return DPH_OK;
}

int dph_index_upload_rows(dph_index* h, int64_t row0, int64_t n, const int8_t* host_rows) {
    DPH_NOT_TWINNED(h, "dph_index_upload_rows");
    if (!h || !host_rows || row0 < 0 || n < 0 || row0 + n > h->n_rows) return fail(DPH_E_ARG, "dph_index_upload_rows: range");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpy(h->db + row0 * DPH_DIM, host_rows, (size_t)n * DPH_DIM, hipMemcpyHostToDevice));
    h->finalized = false;
    return DPH_OK;
}

int dph_index_upload_rows_async(dph_index* h, int64_t row0, int64_t n, const int8_t* pinned_rows, void* stream) {
    DPH_NOT_TWINNED(h, "dph_index_upload_rows_async");
    if (!h || !pinned_rows || row0 < 0 || n < 0 || row0 + n > h->n_rows) return fail(DPH_E_ARG, "dph_index_upload_rows_async: range");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(h->db + row0 * DPH_DIM, pinned_rows, (size_t)n * DPH_DIM, hipMemcpyHostToDevice, (hipStream_t)stream));
    h->finalized = false;
    return DPH_OK;
}

int dph_host_alloc_pinned(size_t bytes, void** out) {
    if (!out) return fail(DPH_E_ARG, "dph_host_alloc_pinned: null");
    HIPCHK(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return DPH_OK;
}
int dph_host_free_pinned(void* p) {
    if (p) HIPCHK(hipHostFree(p));
    return DPH_OK;
}
int dph_stream_synchronize(int device, void* stream) {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return DPH_OK;
}

int dph_index_fill_synthetic(dph_index* h, uint64_t seed, void* stream) {
    DPH_NOT_TWINNED(h, "dph_index_fill_synthetic");
    if (!h) return fail(DPH_E_ARG, "null handle");
    HIPCHK(hipSetDevice(h->device));
    if (h->n_rows > 0) dph_launch_fill(h->db, h->n_rows, h->id_base, seed, 0, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    h->finalized = false;
    return DPH_OK;
}

int dph_index_fill_synthetic_kind(dph_index* h, uint64_t seed, int kind, void* stream) {
    DPH_NOT_TWINNED(h, "dph_index_fill_synthetic_kind");
    if (!h || kind < 0 || kind > 4)
        return fail(DPH_E_ARG, "dph_index_fill_synthetic_kind: kind is 0 (i.i.d.), 1 (mixture + outliers), 2 (document-ordered runs), 3 (mixture) or 4 (anisotropic)");
    HIPCHK(hipSetDevice(h->device));
    if (h->n_rows > 0) dph_launch_fill(h->db, h->n_rows, h->id_base, seed, kind, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    h->finalized = false;
    return DPH_OK;
}

int dph_index_set_idx2id(dph_index* h, const int32_t* doc, const int32_t* word) {
    DPH_NOT_TWINNED(h, "dph_index_set_idx2id");
    if (!h || !doc || !word) return fail(DPH_E_ARG, "dph_index_set_idx2id: null");
    HIPCHK(hipSetDevice(h->device));
    // indexed by local id (= stored row on a flat shard)
    const size_t bytes = (size_t)(h->n_ids > 0 ? h->n_ids : 1) * sizeof(int32_t);
    if (h->row2doc) { (void)hipFree(h->row2doc); h->row2doc = nullptr; }
    if (h->row2word) { (void)hipFree(h->row2word); h->row2word = nullptr; }
    HIPCHK(hipMalloc((void**)&h->row2doc, bytes));
    HIPCHK(hipMalloc((void**)&h->row2word, bytes));
    HIPCHK(hipMemcpy(h->row2doc, doc, (size_t)h->n_ids * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->row2word, word, (size_t)h->n_ids * 4, hipMemcpyHostToDevice));
    h->h_row2doc.assign(doc, doc + h->n_ids);
    h->h_row2word.assign(word, word + h->n_ids);
    return DPH_OK;
}

int dph_index_set_f2o(dph_index* h, int64_t n_docs, const int32_t* doc_ids, const int64_t* f2o_off, const int32_t* f2o) {
    DPH_NOT_TWINNED(h, "dph_index_set_f2o");
    if (!h || n_docs < 0 || (n_docs > 0 && (!doc_ids || !f2o_off || !f2o))) return fail(DPH_E_ARG, "dph_index_set_f2o: null");
    for (int64_t i = 1; i < n_docs; ++i)
        if (doc_ids[i] <= doc_ids[i - 1]) return fail(DPH_E_ARG, "dph_index_set_f2o: doc_ids must be strictly ascending");
    HIPCHK(hipSetDevice(h->device));
    {
        void* old[] = {h->doc_ids, h->f2o_off, h->f2o};
        for (void* p : old) if (p) (void)hipFree(p);
        h->doc_ids = nullptr; h->f2o_off = nullptr; h->f2o = nullptr; h->n_docs = 0;
    }
    const int64_t total = n_docs > 0 ? f2o_off[n_docs] : 0;
    HIPCHK(hipMalloc((void**)&h->doc_ids, (size_t)(n_docs > 0 ? n_docs : 1) * 4));
    HIPCHK(hipMalloc((void**)&h->f2o_off, (size_t)(n_docs + 1) * 8));
    HIPCHK(hipMalloc((void**)&h->f2o, (size_t)(total > 0 ? total : 1) * 4));
    if (n_docs > 0) {
        HIPCHK(hipMemcpy(h->doc_ids, doc_ids, (size_t)n_docs * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->f2o_off, f2o_off, (size_t)(n_docs + 1) * 8, hipMemcpyHostToDevice));
        if (total > 0) HIPCHK(hipMemcpy(h->f2o, f2o, (size_t)total * 4, hipMemcpyHostToDevice));
    } else {
        const int64_t zero = 0;
        HIPCHK(hipMemcpy(h->f2o_off, &zero, 8, hipMemcpyHostToDevice));
    }
    h->n_docs = n_docs;
    return DPH_OK;
}

// ---- shard statistics: what the bounds of the search are made of ------------------------------------------------------
// Every bound (the low-digit bound of the filter, the score bound of the rows that were not re-scored) is a Cauchy-Schwarz
// bound around the per-dimension mean code mu of the stored rows.  dph_index_finalize measures, in three passes over the rows:
//   1. per-dimension sums -> mu_j (rounded mean), spread sd_j;
//   2. the squared centred norms || n - mu ||^2: maximum, histogram -> median, and the outlier cut: ONE extreme row (a saturated
//      code vector) would loosen a shard-wide bound for every query, so the DPH_OUTLIER_MAX largest-norm rows are taken out of it when
//      that tightens it by more than 10 %: "outlier rows", scored exactly against every query (dph_outlier_kernel), never bounded;
//   3. whether the rows are ALIKE.  They are not when (a) some dimensions sit far from the others for every row -- "rogue"
//      dimensions: | mu_j - median mu | beyond 3.5 spreads of a typical dimension; a query of the same encoder carries them too and they
//      would eat the range of its high digit -- or (b) the norms are heavy-tailed (largest non-outlier norm > 1.3 x the median): then
//      the shard gets AUX ROWS (dph_scan.hip "the aux k-step") and the filter bounds every row by its own norm.
static void choose_aux_layout(dph_index* h) {
    dph_aux_layout lay{};
    lay.q2max = 64;
    // (a shard the select step sorts whole is scanned cold -- no bound, nothing for aux rows to tighten -- and its handful of rows says
    // little about dimensions)
    if (h->n_rows <= 0 || h->aux_mode == 0 || (h->n_rows <= DPH_POOL_MAX && h->aux_mode < 0)) { h->aux_lay = lay; return; }
    // rogue dimensions: mean far outside the spread the typical dimension has around the typical mean
    std::vector<double> sds(h->sd_host, h->sd_host + DPH_DIM), mus(DPH_DIM);
    for (int j = 0; j < DPH_DIM; ++j) mus[j] = (double)h->mu_host[j];
    std::nth_element(sds.begin(), sds.begin() + DPH_DIM / 2, sds.end());
    std::nth_element(mus.begin(), mus.begin() + DPH_DIM / 2, mus.end());
    const double sd_med = std::max(1.0, sds[DPH_DIM / 2]), mu_med = mus[DPH_DIM / 2];
    std::vector<std::pair<double, int>> rogue;          // (ratio, dimension): how many times the bulk's range a query needs there
    for (int j = 0; j < DPH_DIM; ++j) {
        const double ratio = fabs((double)h->mu_host[j] - mu_med) / (3.5 * sd_med);
        if (ratio >= 1.2) rogue.push_back({ratio, j});
    }
    std::sort(rogue.begin(), rogue.end(), [](const std::pair<double, int>& x, const std::pair<double, int>& y) {
        return x.first > y.first || (x.first == y.first && x.second < y.second); });
    if (rogue.size() > 12) rogue.resize(12);            // at least two replica slots each
    const bool heavy = h->rmed > 0.0 && h->rmax_cut > 1.3 * h->rmed;
    // replica slots a query that looks like a row needs: its digit in a rogue dimension is `ratio` times the bulk's range.  Twelve fit the
    // 16-byte aux row (4 norm slots), twenty-four the 32-byte one (8 norm slots); with fewer than it needs a query only loses a little
    // resolution in its bulk (the scale is max_j |q_j| / (1 + R_j)), so the narrow row -- half the extra HBM bytes -- is taken when it suffices
    int need = 0;
    for (auto& r : rogue) need += std::max(1, (int)ceil(r.first) - 1);
    int stride = h->aux_mode > 0 ? h->aux_mode : (!rogue.empty() ? (need <= 12 ? 16 : 32) : (heavy ? 4 : 0));
    if (stride >= 16 && rogue.empty()) stride = 4;
    if (stride == 0) { h->aux_lay = lay; return; }
    lay.stride = stride;
    lay.n_norm = stride == 32 ? 8 : 4;
    if (stride >= 16) {
        // the replica slots are dealt out in proportion to the ratios (largest remainder), every rogue dimension >= 1
        const int n_slots = stride == 32 ? DPH_AUX_REP_MAX : 12;
        double total = 0;
        for (auto& r : rogue) total += r.first;
        std::vector<int> cnt(rogue.size(), 1);
        int left = n_slots - (int)rogue.size();
        std::vector<double> want(rogue.size());
        for (size_t i = 0; i < rogue.size(); ++i) want[i] = rogue[i].first / total * n_slots;
        while (left > 0) {
            size_t best = 0; double gap = -1e300;
            for (size_t i = 0; i < rogue.size(); ++i) if (want[i] - cnt[i] > gap) { gap = want[i] - cnt[i]; best = i; }
            cnt[best]++; left--;
        }
        int sl = 0;
        for (size_t i = 0; i < rogue.size(); ++i)
            for (int c = 0; c < cnt[i]; ++c) lay.rep_dim[sl++] = (short)rogue[i].second;
        lay.n_rep = sl;
    }
    h->aux_lay = lay;
}

// norm unit and low-digit clamp of a layout on THIS shard: the norm codes of a row sum to ceil(norm / unit) <= 127 n_norm, and the
// digit they meet, ceil(unit ||q2|| / 128), must fit int8: q2max <= 126 * 128 / (unit * sqrt(768))
static void derive_aux_units(dph_index* h) {
    // outlier rows serve the shard-wide norm bound; with aux rows every row is bounded by its own norm and none is set aside
    if (h->aux_lay.stride <= 0) { h->norm_unit = 1; h->n_out = h->n_out_found; h->rmax = h->n_out_found > 0 ? h->rmax_cut : h->rmax_all; return; }
    h->n_out = 0;
    h->rmax = h->rmax_all;
    const double cap = 127.0 * h->aux_lay.n_norm;
    int unit = (int)ceil(h->rmax_all / cap);
    h->norm_unit = unit < 1 ? 1 : unit;
    if (!h->aux_lay_forced) {
        const int q2 = (int)floor(126.0 * 128.0 / ((double)h->norm_unit * 27.7129));
        h->aux_lay.q2max = q2 > 64 ? 64 : (q2 < 1 ? 1 : q2);
    }
}

static int build_aux_rows(dph_index* h, hipStream_t st) {
    if (h->aux_lay.stride <= 0) return DPH_OK;
    const int64_t padded = h->n_tiles * DPH_TILE_ROWS;
    const size_t bytes = (size_t)padded * h->aux_lay.stride + 256;      // + the over-read of the last lanes (stride 4: 16-byte loads)
    if (bytes > h->aux_bytes) {
        if (h->aux) (void)hipFree(h->aux);
        h->aux = nullptr; h->aux_bytes = 0;
        HIPCHK(hipMalloc((void**)&h->aux, bytes));
        h->aux_bytes = bytes;
    }
    HIPCHK(hipMemsetAsync(h->aux, 0, h->aux_bytes, st));
    dph_launch_aux_build(h->db, h->n_rows, padded, h->row_ids, h->mu_dev, h->aux_lay, h->norm_unit, h->aux, st);
    HIPCHK(hipGetLastError());
    return DPH_OK;
}

int dph_index_finalize(dph_index* h, void* stream) {
    DPH_NOT_TWINNED(h, "dph_index_finalize");
    if (!h) return fail(DPH_E_ARG, "null handle");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    if (h->pq) {                         // a PQ index has no row statistics: its finalize builds the direct map
        const int rc = dph_pq_finalize(h->pq, st);
        if (rc) return fail(rc, dph_pq_error());
        h->finalized = true;
        return DPH_OK;
    }
    // ---- 1. per-dimension mean codes
    int64_t n_real = h->row_ids ? h->n_ids : h->n_rows;
    std::vector<long long> sums(2 * DPH_DIM, 0);
    HIPCHK(hipMemsetAsync(h->colsum_dev, 0, 2 * DPH_DIM * sizeof(long long), st));
    if (h->n_rows > 0) dph_launch_colstats(h->db, h->n_rows, h->row_ids, h->colsum_dev, st);
    HIPCHK(hipMemcpyAsync(sums.data(), h->colsum_dev, 2 * DPH_DIM * sizeof(long long), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int j = 0; j < DPH_DIM; ++j) {
        const double mean = n_real > 0 ? (double)sums[j] / (double)n_real : 0.0;
        const double var = n_real > 0 ? (double)sums[DPH_DIM + j] / (double)n_real - mean * mean : 0.0;
        int m = (int)lrint(mean);
        h->mu_host[j] = m < -128 ? -128 : (m > 127 ? 127 : m);
        h->sd_host[j] = var > 0.0 ? sqrt(var) : 0.0;
    }
    HIPCHK(hipMemcpyAsync(h->mu_dev, h->mu_host, DPH_DIM * sizeof(int), hipMemcpyHostToDevice, st));
    // ---- 2. centred row norms
    HIPCHK(hipMemsetAsync(h->norm_dev, 0, 2 * sizeof(unsigned long long), st));
    HIPCHK(hipMemsetAsync(h->hist_dev, 0, DPH_NORM_BINS * sizeof(unsigned), st));
    if (h->n_rows > 0) dph_launch_rownorm(h->db, h->n_rows, h->row_ids, h->mu_dev, h->norm_dev, h->hist_dev, 0, nullptr, nullptr, 0, st);
    unsigned long long m = 0;
    std::vector<unsigned> hist(DPH_NORM_BINS);
    HIPCHK(hipMemcpyAsync(&m, h->norm_dev, sizeof(m), hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(hist.data(), h->hist_dev, DPH_NORM_BINS * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    h->rmax_all = sqrt((double)m);
    h->rmax = h->rmax_all;
    h->rmax_cut = h->rmax_all;
    h->n_out = 0;
    h->n_out_found = 0;
    {
        uint64_t total = 0, run = 0;
        for (int b = 0; b < DPH_NORM_BINS; ++b) total += hist[b];
        h->rmed = 0.0;
        for (int b = 0; b < DPH_NORM_BINS && total > 0; ++b) {
            run += hist[b];
            if (2 * run >= total) { h->rmed = sqrt(((double)b + 0.5) * DPH_NORM_BIN_W); break; }
        }
    }
    // the lowest bin edge with at most DPH_OUTLIER_MAX rows above it
    uint64_t above = 0;
    int cut_bin = DPH_NORM_BINS;         // rows in bins >= cut_bin are outliers
    for (int b = DPH_NORM_BINS - 1; b >= 0; --b) {
        if (above + hist[b] > (uint64_t)DPH_OUTLIER_MAX) break;
        above += hist[b];
        cut_bin = b;
    }
    const unsigned long long cut2 = (unsigned long long)cut_bin * DPH_NORM_BIN_W;       // norm^2 < cut2 for the rest
    if (above > 0 && cut_bin > 0 && (double)cut2 * 1.21 < (double)m) {
        unsigned* cnt = (unsigned*)(h->norm_dev + 1);
        dph_launch_rownorm(h->db, h->n_rows, h->row_ids, h->mu_dev, nullptr, nullptr, cut2 - 1, h->outliers, cnt, DPH_OUTLIER_MAX, st);
        unsigned n_out = 0;
        HIPCHK(hipMemcpyAsync(&n_out, cnt, sizeof(n_out), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (n_out > 0 && n_out <= (unsigned)DPH_OUTLIER_MAX) {
            std::vector<unsigned> rows(n_out);
            HIPCHK(hipMemcpy(rows.data(), h->outliers, n_out * sizeof(unsigned), hipMemcpyDeviceToHost));
            std::sort(rows.begin(), rows.end());
            HIPCHK(hipMemcpy(h->outliers, rows.data(), n_out * sizeof(unsigned), hipMemcpyHostToDevice));
            h->n_out_found = (int)n_out;
            h->rmax_cut = sqrt((double)cut2);
        }
    }
    // ---- 3. aux rows
    if (!h->aux_lay_forced) choose_aux_layout(h);
    derive_aux_units(h);
    if (h->aux_lay_forced && h->aux_lay.stride > 0) {
        // a layout forced by dph_index_set_aux_layout survives a re-finalize only while its low-digit clamp still serves THESE rows
        // (new rows with larger norms -> a larger norm unit -> the query's norm digit would not fit int8 and the per-row bound would be
        // too small: wrong rows filtered, result still "certified").  Otherwise the shard chooses afresh and the ranks of a sharded job
        // agree again (densephrases_amd: Shard.finalize resets the sync mark).
        const int q2 = (int)floor(126.0 * 128.0 / ((double)h->norm_unit * 27.7129));
        if (h->aux_lay.q2max > q2) {
            h->aux_lay_forced = false;
            choose_aux_layout(h);
            derive_aux_units(h);
        }
    }
    const int rc = build_aux_rows(h, st);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(st));
    h->finalized = true;
    return DPH_OK;
}

// The digits of a query row depend on the aux layout (which dimensions have replica digits, the clamp of the low digit), and the
// ranks of a range-sharded job compare INTEGER scores (dph_search_sample_dev / dph_union_bounds_dev / dph_search_bounded_dev): every
// rank must cut its queries into the same digits.  A sharded caller therefore reads the layout of every finalized shard, agrees on
// one (densephrases_amd/dist.py: the widest stride, the replica table of the lowest rank that has one, the smallest clamp) and
// sets it everywhere; the aux rows are rebuilt to it (one pass over the rows).  out / in: [0] stride, [1] n_norm, [2] n_rep,
// [3] q2max, [4 .. 4 + DPH_AUX_REP_MAX) rep_dim.
int dph_index_get_aux_layout(dph_index* h, int32_t* out) {
    if (!h || !out) return fail(DPH_E_ARG, "dph_index_get_aux_layout: null");
    if (!h->finalized) return fail(DPH_E_STATE, "dph_index_get_aux_layout: call dph_index_finalize first");
    const dph_index* src = h->twin_of ? h->twin_of : h;
    out[0] = src->aux_lay.stride; out[1] = src->aux_lay.n_norm; out[2] = src->aux_lay.n_rep; out[3] = src->aux_lay.q2max;
    if (src->aux_lay.stride <= 0) {
        // no aux rows here, but another rank of the job may force some on this shard: report the clamp THIS shard's row norms would allow
        // with the fewest norm slots a layout has (4), so that the minimum over the ranks is one every rank can set
        const int unit = std::max(1, (int)ceil(src->rmax_all / (127.0 * 4.0)));
        const int q2 = (int)floor(126.0 * 128.0 / ((double)unit * 27.7129));
        out[3] = q2 > 64 ? 64 : (q2 < 1 ? 1 : q2);
    }
    for (int i = 0; i < DPH_AUX_REP_MAX; ++i) out[4 + i] = i < src->aux_lay.n_rep ? src->aux_lay.rep_dim[i] : -1;
    return DPH_OK;
}
int dph_index_set_aux_layout(dph_index* h, const int32_t* in) {
    DPH_NOT_TWINNED(h, "dph_index_set_aux_layout");
    if (!h || !in) return fail(DPH_E_ARG, "dph_index_set_aux_layout: null");
    if (h->pq) return DPH_OK;            // a PQ index has no digits
    if (!h->finalized) return fail(DPH_E_STATE, "dph_index_set_aux_layout: call dph_index_finalize first");
    dph_aux_layout lay{};
    lay.stride = in[0]; lay.n_norm = in[1]; lay.n_rep = in[2]; lay.q2max = in[3];
    const bool ok_shape = (lay.stride == 0 && lay.n_norm == 0 && lay.n_rep == 0) || (lay.stride == 4 && lay.n_norm == 4 && lay.n_rep == 0) ||
                          (lay.stride == 16 && lay.n_norm == 4 && lay.n_rep >= 0 && lay.n_rep <= 12) ||
                          (lay.stride == 32 && lay.n_norm == 8 && lay.n_rep >= 0 && lay.n_rep <= DPH_AUX_REP_MAX);
    if (!ok_shape || lay.q2max < 1 || lay.q2max > 64) return fail(DPH_E_ARG, "dph_index_set_aux_layout: not a layout dph_index_get_aux_layout returns");
    for (int i = 0; i < lay.n_rep; ++i) {
        if (in[4 + i] < 0 || in[4 + i] >= DPH_DIM) return fail(DPH_E_ARG, "dph_index_set_aux_layout: replica dimension out of range");
        lay.rep_dim[i] = (short)in[4 + i];
    }
    HIPCHK(hipSetDevice(h->device));
    h->aux_lay = lay;
    h->aux_lay_forced = true;
    derive_aux_units(h);
    if (lay.stride > 0) {
        // the forced clamp must serve this shard's norm unit (see derive_aux_units)
        const int q2 = (int)floor(126.0 * 128.0 / ((double)h->norm_unit * 27.7129));
        if (lay.q2max > q2) return fail(DPH_E_ARG, "dph_index_set_aux_layout: q2max too large for this shard's row norms (take the minimum over the ranks)");
    }
    const int rc = build_aux_rows(h, nullptr);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(nullptr));
    return DPH_OK;
}

int64_t dph_index_ntotal(const dph_index* h) { return h ? h->n_ids : 0; }

int dph_index_set_row_ids(dph_index* h, const int64_t* row_ids, int64_t n_ids) {
    DPH_NOT_TWINNED(h, "dph_index_set_row_ids");
    if (!h || !row_ids || n_ids < 0 || n_ids > h->n_rows) return fail(DPH_E_ARG, "dph_index_set_row_ids: bad arguments");
    std::vector<int32_t> inv((size_t)n_ids, -1);
    for (int64_t r = 0; r < h->n_rows; ++r) {
        const int64_t id = row_ids[r];
        if (id < 0) continue;
        const int64_t l = id - h->id_base;
        if (l < 0 || l >= n_ids || inv[(size_t)l] != -1) return fail(DPH_E_ARG, "dph_index_set_row_ids: ids must be a permutation of [id_base, id_base+n_ids)");
        inv[(size_t)l] = (int32_t)r;
    }
    for (int64_t l = 0; l < n_ids; ++l) if (inv[(size_t)l] < 0) return fail(DPH_E_ARG, "dph_index_set_row_ids: an id has no row");
    HIPCHK(hipSetDevice(h->device));
    void* old[] = {h->row_ids, h->inv_row, h->onesmask};
    for (void* p : old) if (p) (void)hipFree(p);
    h->row_ids = nullptr; h->inv_row = nullptr; h->onesmask = nullptr;
    HIPCHK(hipMalloc((void**)&h->row_ids, (size_t)(h->n_tiles > 0 ? h->n_tiles : 1) * DPH_TILE_ROWS * 8));
    HIPCHK(hipMemset(h->row_ids, 0xFF, (size_t)(h->n_tiles > 0 ? h->n_tiles : 1) * DPH_TILE_ROWS * 8));   // padding = -1
    HIPCHK(hipMemcpy(h->row_ids, row_ids, (size_t)h->n_rows * 8, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&h->inv_row, (size_t)(n_ids > 0 ? n_ids : 1) * 4));
    HIPCHK(hipMemcpy(h->inv_row, inv.data(), (size_t)n_ids * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&h->onesmask, (size_t)(h->n_tiles > 0 ? h->n_tiles : 1) * 32));
    HIPCHK(hipMemset(h->onesmask, 0xFF, (size_t)(h->n_tiles > 0 ? h->n_tiles : 1) * 32));
    h->h_inv.swap(inv);
    h->n_ids = n_ids;
    h->finalized = false;
    return DPH_OK;
}

int dph_index_set_id_groups(dph_index* h, int n_groups, const int64_t* id_offsets, const int64_t* row_starts) {
    DPH_NOT_TWINNED(h, "dph_index_set_id_groups");
    if (!h || n_groups < 0 || (n_groups > 0 && (!id_offsets || !row_starts))) return fail(DPH_E_ARG, "dph_index_set_id_groups: bad arguments");
    if (h->row_ids) return fail(DPH_E_STATE, "dph_index_set_id_groups: not on a list-major shard (its row_ids already carry the ids)");
    if (n_groups > 0) {
        if (row_starts[0] != 0 || row_starts[n_groups] != (h->pq ? h->n_ids : h->n_rows)) return fail(DPH_E_ARG, "dph_index_set_id_groups: row_starts must run from 0 to n_rows");
        for (int g = 0; g < n_groups; ++g) {
            const int64_t len = row_starts[g + 1] - row_starts[g];
            if (len < 0 || id_offsets[g] < 0) return fail(DPH_E_ARG, "dph_index_set_id_groups: negative group");
            if (g + 1 < n_groups && id_offsets[g] + len > id_offsets[g + 1]) return fail(DPH_E_ARG, "dph_index_set_id_groups: id ranges must ascend without overlap");
        }
    }
    HIPCHK(hipSetDevice(h->device));
    if (h->id_offsets) { (void)hipFree(h->id_offsets); h->id_offsets = nullptr; }
    if (h->row_starts) { (void)hipFree(h->row_starts); h->row_starts = nullptr; }
    h->h_id_offsets.assign(id_offsets, id_offsets + n_groups);
    h->h_row_starts.assign(row_starts, row_starts + (n_groups > 0 ? n_groups + 1 : 0));
    if (n_groups > 0) {
        HIPCHK(hipMalloc((void**)&h->id_offsets, (size_t)n_groups * 8));
        HIPCHK(hipMalloc((void**)&h->row_starts, (size_t)(n_groups + 1) * 8));
        HIPCHK(hipMemcpy(h->id_offsets, id_offsets, (size_t)n_groups * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->row_starts, row_starts, (size_t)(n_groups + 1) * 8, hipMemcpyHostToDevice));
    }
    return DPH_OK;
}

// host twin of dph_local_of_id (flat / grouped shards: the stored row; list-major shards: the local id)
static int64_t host_local_of_id(const dph_index* h, int64_t id) {
    const int ng = (int)h->h_id_offsets.size();
    if (ng == 0) { const int64_t l = id - h->id_base; return (l < 0 || l >= h->n_ids) ? -1 : l; }
    if (id < h->h_id_offsets[0]) return -1;
    int g = (int)(std::upper_bound(h->h_id_offsets.begin(), h->h_id_offsets.end(), id) - h->h_id_offsets.begin()) - 1;
    const int64_t r = id - h->h_id_offsets[(size_t)g];
    return r < h->h_row_starts[(size_t)g + 1] - h->h_row_starts[(size_t)g] ? h->h_row_starts[(size_t)g] + r : -1;
}

int dph_index_set_ivf(dph_index* h, int nlist, const float* centroids, const int32_t* tile_list) {
    DPH_NOT_TWINNED(h, "dph_index_set_ivf");
    if (!h || nlist <= 0 || nlist > (1 << 20) || !centroids || !tile_list) return fail(DPH_E_ARG, "dph_index_set_ivf: bad arguments");
    if (!h->row_ids) return fail(DPH_E_STATE, "dph_index_set_ivf: call dph_index_set_row_ids first (list-major shard)");
    for (int64_t t = 0; t < h->n_tiles; ++t)
        if (tile_list[t] < 0 || tile_list[t] >= nlist) return fail(DPH_E_ARG, "dph_index_set_ivf: tile_list out of range");
    HIPCHK(hipSetDevice(h->device));
    void* old[] = {h->centroids, h->tile_list, h->listmask, h->tilemask, h->coarse_scores, h->list_tile0, h->listmask_u, h->unit_offsets,
                   h->unit_counts, h->slot_q, h->unit_recs, h->unit_list_recs, h->unit_frags};
    for (void* p : old) if (p) (void)hipFree(p);
    h->centroids = nullptr; h->tile_list = nullptr; h->listmask = nullptr; h->tilemask = nullptr; h->coarse_scores = nullptr;
    h->list_tile0 = nullptr; h->listmask_u = nullptr; h->unit_offsets = nullptr; h->unit_counts = nullptr; h->slot_q = nullptr; h->unit_recs = nullptr;
    h->unit_frags = nullptr; h->unit_list_recs = nullptr; h->chunk_cap = 0; h->unit_cap = 0;
    // the unit scan needs every list to be one contiguous run of tiles (what ivf.build_list_major writes)
    std::vector<int> tile0((size_t)nlist + 1, 0);
    h->lists_contiguous = nlist <= 65536 && h->n_tiles < (int64_t)0x7fffffff;
    for (int64_t t = 1; t < h->n_tiles && h->lists_contiguous; ++t) if (tile_list[t] < tile_list[t - 1]) h->lists_contiguous = false;
    h->max_list_tiles = 0; h->total_segs = 0;
    if (h->lists_contiguous) {
        std::vector<int> cnt((size_t)nlist, 0);
        for (int64_t t = 0; t < h->n_tiles; ++t) cnt[(size_t)tile_list[t]]++;
        for (int l = 0; l < nlist; ++l) {
            tile0[(size_t)l + 1] = tile0[(size_t)l] + cnt[(size_t)l];
            h->max_list_tiles = std::max(h->max_list_tiles, cnt[(size_t)l]);
            h->total_segs += (cnt[(size_t)l] + DPH_UNIT_TILES - 1) / DPH_UNIT_TILES;
        }
    }
    double cmax = 0.0;
    for (int l = 0; l < nlist; ++l) {
        double a = 0.0;
        for (int j = 0; j < DPH_DIM; ++j) a += (double)centroids[(size_t)l * DPH_DIM + j] * centroids[(size_t)l * DPH_DIM + j];
        cmax = fmax(cmax, a);
    }
    h->cnorm_max = sqrt(cmax);
    h->coarse_rows = h->lists_contiguous ? DPH_PASS_MAX : DPH_QROWS * DPH_MAX_QB;
    HIPCHK(hipMalloc((void**)&h->coarse_scores, (size_t)h->coarse_rows * nlist * 4));
    if (h->lists_contiguous) {
        HIPCHK(hipMalloc((void**)&h->list_tile0, ((size_t)nlist + 1) * 4));
        HIPCHK(hipMemcpy(h->list_tile0, tile0.data(), ((size_t)nlist + 1) * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMalloc((void**)&h->listmask_u, (size_t)nlist * DPH_UNIT_WORDS * 4));
        HIPCHK(hipMalloc((void**)&h->unit_offsets, (size_t)nlist * sizeof(int2)));
        HIPCHK(hipMalloc((void**)&h->unit_counts, (size_t)(4 + DPH_UNIT_LAUNCHES) * 4));
    }
    HIPCHK(hipMalloc((void**)&h->centroids, (size_t)nlist * DPH_DIM * 4));
    HIPCHK(hipMemcpy(h->centroids, centroids, (size_t)nlist * DPH_DIM * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&h->tile_list, (size_t)(h->n_tiles > 0 ? h->n_tiles : 1) * 4));
    HIPCHK(hipMemcpy(h->tile_list, tile_list, (size_t)h->n_tiles * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&h->listmask, (size_t)nlist * 32));
    HIPCHK(hipMalloc((void**)&h->tilemask, (size_t)(h->n_tiles > 0 ? h->n_tiles : 1) * 32));
    h->nlist = nlist;
    return DPH_OK;
}
// A flat shard resident in HBM -> list-major shard + IVF data, on the device (dph_build.hip): the rows are permuted into a
// second buffer (the dump is in HBM twice for the duration: 2 x 131 GB of the 288), ids stay what they were.
int dph_index_make_list_major(dph_index* h, const int32_t* assign_dev, int nlist, const float* centroids, void* stream) {
    DPH_NOT_TWINNED(h, "dph_index_make_list_major");
    if (!h || !assign_dev || !centroids || nlist <= 0 || nlist > (1 << 20)) return fail(DPH_E_ARG, "dph_index_make_list_major: bad arguments");
    if (h->row_ids) return fail(DPH_E_STATE, "dph_index_make_list_major: the shard is list-major already");
    if (!h->h_id_offsets.empty()) return fail(DPH_E_STATE, "dph_index_make_list_major: not on a shard merged from several sub-indexes");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = h->n_rows;
    std::vector<int64_t> src_start((size_t)nlist + 1, 0);
    uint64_t* keys = nullptr;
    const int src_rc = dph_list_major_sort(assign_dev, n, nlist, &keys, src_start.data(), st);
    if (src_rc == 2) return fail(DPH_E_ARG, "dph_index_make_list_major: a list number is out of range");
    if (src_rc != 0) return fail(DPH_E_NOMEM, "dph_index_make_list_major: sort scratch");
    // every list padded to whole tiles; the tile -> list table the scans read
    std::vector<int64_t> dst_start((size_t)nlist, 0);
    int64_t n_store = 0;
    for (int l = 0; l < nlist; ++l) {
        dst_start[(size_t)l] = n_store;
        const int64_t cnt = src_start[(size_t)l + 1] - src_start[(size_t)l];
        n_store += (cnt + DPH_TILE_ROWS - 1) / DPH_TILE_ROWS * DPH_TILE_ROWS;
    }
    if (n_store >= (int64_t)0xFFFFFFF0ll) { (void)hipFree(keys); return fail(DPH_E_ARG, "dph_index_make_list_major: padded shard exceeds 2^32-16 rows"); }
    const int64_t n_tiles = n_store / DPH_TILE_ROWS;
    std::vector<int32_t> tile_list((size_t)n_tiles);
    for (int l = 0; l < nlist; ++l) {
        const int64_t t0 = dst_start[(size_t)l] / DPH_TILE_ROWS;
        const int64_t t1 = (l + 1 < nlist ? dst_start[(size_t)l + 1] : n_store) / DPH_TILE_ROWS;
        for (int64_t t = t0; t < t1; ++t) tile_list[(size_t)t] = l;
    }
    int8_t* db_new = nullptr; int64_t* row_ids = nullptr; int32_t* inv_row = nullptr; unsigned* ones = nullptr;
    int64_t *src_dev = nullptr, *dst_dev = nullptr;
    auto cleanup = [&]() {
        void* p[] = {keys, db_new, row_ids, inv_row, ones, src_dev, dst_dev};
        for (void* v : p) if (v) (void)hipFree(v);
    };
    const size_t tiles_alloc = (size_t)(n_tiles > 0 ? n_tiles : 1);
    if (hipMalloc((void**)&db_new, tiles_alloc * DPH_TILE_BYTES) != hipSuccess || hipMalloc((void**)&row_ids, tiles_alloc * DPH_TILE_ROWS * 8) != hipSuccess ||
        hipMalloc((void**)&inv_row, (size_t)(n > 0 ? n : 1) * 4) != hipSuccess || hipMalloc((void**)&ones, tiles_alloc * 32) != hipSuccess ||
        hipMalloc((void**)&src_dev, ((size_t)nlist + 1) * 8) != hipSuccess || hipMalloc((void**)&dst_dev, (size_t)nlist * 8) != hipSuccess) {
        cleanup();
        return fail(DPH_E_NOMEM, "dph_index_make_list_major: the permuted copy does not fit next to the shard");
    }
    (void)hipMemsetAsync(db_new, 0, tiles_alloc * DPH_TILE_BYTES, st);                    // padding rows are zero
    (void)hipMemsetAsync(row_ids, 0xFF, tiles_alloc * DPH_TILE_ROWS * 8, st);             // ... and carry id -1
    (void)hipMemsetAsync(ones, 0xFF, tiles_alloc * 32, st);
    (void)hipMemsetAsync(inv_row, 0xFF, (size_t)(n > 0 ? n : 1) * 4, st);                 // every id must get a row: checked below
    (void)hipMemcpyAsync(src_dev, src_start.data(), ((size_t)nlist + 1) * 8, hipMemcpyHostToDevice, st);
    (void)hipMemcpyAsync(dst_dev, dst_start.data(), (size_t)nlist * 8, hipMemcpyHostToDevice, st);
    dph_launch_list_major_gather(h->db, keys, n, src_dev, dst_dev, h->id_base, db_new, row_ids, inv_row, st);
    std::vector<int32_t> inv((size_t)n);
    hipError_t e = hipMemcpyAsync(inv.data(), inv_row, (size_t)n * 4, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) { cleanup(); return fail(DPH_E_HIP, std::string("dph_index_make_list_major: ") + hipGetErrorString(e)); }
    for (int64_t r = 0; r < n; ++r)
        if (inv[(size_t)r] < 0) { cleanup(); return fail(DPH_E_HIP, "dph_index_make_list_major: the gather left row " + std::to_string(r) + " without a place"); }
    (void)hipFree(keys); (void)hipFree(src_dev); (void)hipFree(dst_dev);
    (void)hipFree(h->db);
    if (h->onesmask) (void)hipFree(h->onesmask);
    h->db = db_new; h->row_ids = row_ids; h->inv_row = inv_row; h->onesmask = ones;
    h->h_inv.swap(inv);
    h->n_ids = n;
    h->n_rows = n_store;
    h->n_tiles = n_tiles;
    h->finalized = false;
    return dph_index_set_ivf(h, nlist, centroids, tile_list.data());
}

int dph_index_rehome_rows(dph_index* h, void* stream) {
    DPH_NOT_TWINNED(h, "dph_index_rehome_rows");
    if (!h) return fail(DPH_E_ARG, "null handle");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t bytes = (size_t)(h->n_tiles > 0 ? h->n_tiles : 1) * DPH_TILE_BYTES;
    int8_t* fresh = nullptr;
    if (hipMalloc((void**)&fresh, bytes) != hipSuccess) return fail(DPH_E_NOMEM, "dph_index_rehome_rows: no room for a second copy of the rows");
    hipError_t e = hipMemcpyAsync(fresh, h->db, bytes, hipMemcpyDeviceToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { (void)hipFree(fresh); return fail(DPH_E_HIP, std::string("dph_index_rehome_rows: ") + hipGetErrorString(e)); }
    (void)hipFree(h->db);
    h->db = fresh;
    return DPH_OK;
}

// ---- the reference's own index: IndexPreTransform(OPQMatrix) + IndexIVFPQ (index.py:30-33; densephrases_amd/faiss_io.py reads the file)
int dph_index_create_pq(int device, int64_t ntotal, int nlist, int M, dph_index** out) {
    if (!out || ntotal < 0) return fail(DPH_E_ARG, "dph_index_create_pq: bad arguments");
    dph_index* h = nullptr;
    int rc = dph_index_create(device, 0, 0, &h);
    if (rc) return rc;
    rc = dph_pq_alloc(&h->pq, device, ntotal, nlist, M);
    if (rc) { dph_index_destroy(h); return fail(rc, std::string("dph_index_create_pq: ") + dph_pq_error()); }
    h->n_ids = ntotal;                   // idx2id / id groups are indexed by local id
    h->default_nprobe = 256;             // index.py:53,62
    *out = h;
    return DPH_OK;
}
int dph_index_set_pq(dph_index* h, const float* A, const float* b, const float* centroids, const float* pq_centroids, int by_residual) {
    DPH_NOT_TWINNED(h, "dph_index_set_pq");
    if (!h || !h->pq) return fail(DPH_E_STATE, "dph_index_set_pq: not a PQ index (dph_index_create_pq)");
    const int rc = dph_pq_set_params(h->pq, A, b, centroids, pq_centroids, by_residual);
    h->finalized = false;
    return rc ? fail(rc, dph_pq_error()) : DPH_OK;
}
int dph_index_set_pq_list_sizes(dph_index* h, const int64_t* sizes) {
    DPH_NOT_TWINNED(h, "dph_index_set_pq_list_sizes");
    if (!h || !h->pq) return fail(DPH_E_STATE, "dph_index_set_pq_list_sizes: not a PQ index");
    const int rc = dph_pq_set_list_sizes(h->pq, sizes);
    h->finalized = false;
    return rc ? fail(rc, dph_pq_error()) : DPH_OK;
}
int dph_index_upload_pq_codes(dph_index* h, int64_t pos0, int64_t n, const uint8_t* codes, const int64_t* ids) {
    DPH_NOT_TWINNED(h, "dph_index_upload_pq_codes");
    if (!h || !h->pq) return fail(DPH_E_STATE, "dph_index_upload_pq_codes: not a PQ index");
    const int rc = dph_pq_upload(h->pq, pos0, n, codes, ids);
    h->finalized = false;
    return rc ? fail(rc, dph_pq_error()) : DPH_OK;
}
int dph_index_get_transform(dph_index* h, float* A_out) {
    if (!h || !A_out) return fail(DPH_E_ARG, "dph_index_get_transform: null");
    if (h->pq) { memcpy(A_out, dph_pq_A_host(h->pq), (size_t)DPH_DIM * DPH_DIM * 4); return DPH_OK; }
    memset(A_out, 0, (size_t)DPH_DIM * DPH_DIM * 4);
    for (int i = 0; i < DPH_DIM; ++i) A_out[(size_t)i * DPH_DIM + i] = 1.f;
    return DPH_OK;
}

int dph_index_dim(const dph_index* h) { (void)h; return DPH_DIM; }
int dph_index_device(const dph_index* h) { return h ? h->device : -1; }
void* dph_index_rows_dev(dph_index* h) { if (h) h->finalized = false; return h ? h->db : nullptr; }

int dph_index_set_tuning(dph_index* h, const char* key, const int32_t* values, int n_values) {
    if (!h || !key || n_values < 0 || (n_values > 0 && !values)) return fail(DPH_E_ARG, "dph_index_set_tuning: bad arguments");
    const std::string k(key);
    if (h->twin_of && (k == "nprobe" || k == "ivf_units" || k == "ivf_spread" || k == "coarse_filter" || k == "coarse_teams"))
        return fail(DPH_E_STATE, "dph_index_set_tuning: " + k + ": not on a twin (flat shards only; set it on the index itself)");
    auto one = [&](int lo, int hi, int* dst) {
        if (n_values != 1 || values[0] < lo || values[0] > hi) return fail(DPH_E_ARG, "dph_index_set_tuning: " + k + ": value out of range");
        *dst = values[0];
        return DPH_OK;
    };
    if (k == "ladder") {                 // explicit pre-pass strides, coarse -> fine; {0} = no pre-pass; empty = default
        for (int i = 0; i < n_values; ++i) if (values[i] < 0) return fail(DPH_E_ARG, "dph_index_set_tuning: ladder strides must be >= 0");
        h->ladder.assign(values, values + n_values);
        return DPH_OK;
    }
    if (k == "fine_stride") return one(0, 1 << 20, &h->fine_stride);
    if (k == "sample_kp") return one(1, 1024, &h->sample_kp);
    if (k == "max_qb") return one(1, DPH_MAX_QB, &h->max_qb);
    if (k == "nprobe") return one(0, 1 << 20, &h->default_nprobe);
    if (k == "ivf_units") return one(-1, 1, &h->ivf_units);
    if (k == "ivf_spread") return one(0, 1, &h->ivf_spread);
    if (k == "scan_seg") return one(1, 1 << 16, &h->seg_tiles);
    if (k == "ladder_fuse") return one(0, 1, &h->ladder_fuse);
    if (k == "coarse_teams") {           // PQ index, filter scan: passes of more than 128 rows as ONE launch of workgroup teams (default 1)
        if (!h->pq) return fail(DPH_E_STATE, "coarse_teams: not a PQ index");
        if (n_values != 1 || values[0] < 0 || values[0] > 1) return fail(DPH_E_ARG, "coarse_teams: 0 or 1");
        dph_pq_set_coarse_teams(h->pq, values[0]);
        return DPH_OK;
    }
    if (k == "coarse_filter") {          // PQ index: 1 = the one-product filter GEMM in front of the coarse quantizer (default), 0 = the bf16x3 chain alone
        if (!h->pq) return fail(DPH_E_STATE, "coarse_filter: not a PQ index");
        if (n_values != 1 || values[0] < 0 || values[0] > 5) return fail(DPH_E_ARG, "coarse_filter: 0 .. 5");
        dph_pq_set_coarse_filter(h->pq, values[0]);
        return DPH_OK;
    }
    if (k == "pq_split_lut") {           // PQ index, row-major ADC scan: 1 = the last sixteen tables are gathered from global memory (vector L1), the rest from LDS
        if (!h->pq) return fail(DPH_E_STATE, "pq_split_lut: not a PQ index");
        if (n_values != 1 || values[0] < 0 || values[0] > 1) return fail(DPH_E_ARG, "pq_split_lut: 0 or 1");
        dph_pq_set_split_lut(h->pq, values[0]);
        return DPH_OK;
    }
    if (k == "retry_chain") return one(0, 1, &h->retry_chain);
    if (k == "aux") {                    // aux rows: -1 = dph_index_finalize decides (default), 0 = never, 4 = norm codes, 32 = norm codes + rogue replicas
        DPH_NOT_TWINNED(h, "dph_index_set_tuning(aux)");
        if (h->pq) return fail(DPH_E_STATE, "aux: not on a PQ index");
        if (n_values != 1 || (values[0] != -1 && values[0] != 0 && values[0] != 4 && values[0] != 16 && values[0] != 32)) return fail(DPH_E_ARG, "aux: -1, 0, 4, 16 or 32");
        h->aux_mode = values[0];
        h->aux_lay_forced = false;
        if (h->finalized) {              // the statistics stand: only the layout and the aux rows follow
            HIPCHK(hipSetDevice(h->device));
            choose_aux_layout(h);
            derive_aux_units(h);
            const int rc = build_aux_rows(h, nullptr);
            if (rc) return rc;
            HIPCHK(hipStreamSynchronize(nullptr));
        }
        return DPH_OK;
    }
    if (k == "side_grid") return one(0, 256, &h->side_grid);
    if (k == "scan_grid") {              // persistent scan workgroups (one per CU): fewer than the CU count leaves CUs to other streams
        const int cus = dph_scan_grid(h->device);
        if (n_values != 1 || values[0] < 0 || values[0] > cus) return fail(DPH_E_ARG, "scan_grid: 0 (= the CU count) .. CU count");
        if (h->wave_counts) return fail(DPH_E_STATE, "scan_grid: set it before the first search (scratch is sized by it)");
        h->grid = values[0] ? values[0] : cus;
        return DPH_OK;
    }
    if (k == "scan_sched") {             // one value: both kernels; two: the 128-row and the 256-row kernel
        if (n_values < 1 || n_values > 2) return fail(DPH_E_ARG, "scan_sched: one or two values");
        for (int i = 0; i < n_values; ++i) if (values[i] < 0 || values[i] > 2) return fail(DPH_E_ARG, "scan_sched: 0, 1 or 2");
        h->scan_sched[0] = values[0];
        h->scan_sched[1] = values[n_values - 1];
        return DPH_OK;
    }
    return fail(DPH_E_ARG, "dph_index_set_tuning: unknown key " + k);
}

// ------------------------------------------------------------------------------------------ search driver
static int alloc_qimg(dph_index::qimg& q, int64_t padded, bool with_x) {
    free_qimg(q);
    if (with_x) HIPCHK(hipMalloc((void**)&q.x, (size_t)padded * DPH_DIM * 4));
    HIPCHK(hipMalloc((void**)&q.frag, (size_t)(padded / DPH_QGROUP) * DPH_QGROUP_FRAG_BYTES));
    HIPCHK(hipMalloc((void**)&q.q1, (size_t)padded * DPH_DIM));
    HIPCHK(hipMalloc((void**)&q.q2, (size_t)padded * DPH_DIM));
    HIPCHK(hipMalloc((void**)&q.qinfo, (size_t)padded * sizeof(dph_qinfo)));
    HIPCHK(hipMalloc((void**)&q.lmax, (size_t)padded * sizeof(int)));
    HIPCHK(hipMalloc((void**)&q.qaux, (size_t)padded * DPH_AUX_SLOTS));
    HIPCHK(hipMemset(q.qaux, 0, (size_t)padded * DPH_AUX_SLOTS));
    HIPCHK(hipMalloc((void**)&q.nf, (size_t)padded * 4));
    HIPCHK(hipMemset(q.nf, 0, (size_t)padded * 4));
    return DPH_OK;
}

static int ensure_scratch(dph_index* h, int64_t n, int k_host) {
    // padded to whole passes of the widest kernel, plus one pass of slack for the quantiser's padding groups
    const int64_t pass = DPH_QROWS * DPH_MAX_QB;
    const int64_t padded = (n + pass - 1) / pass * pass;
    if (padded > h->cap_rows) {
        int rc = alloc_qimg(h->q_main, padded, true);
        if (rc) return rc;
        rc = alloc_qimg(h->q_retry, padded, true);
        if (rc) return rc;
        void* old[] = {h->status_dev, h->ik_dev, h->fail_dev, h->fail2_dev, h->retry_rows, h->exact_rows, h->retry_tau, h->D_dev, h->I_dev};
        for (void* p : old) if (p) (void)hipFree(p);
        h->status_dev = nullptr; h->ik_dev = nullptr; h->fail_dev = nullptr; h->fail2_dev = nullptr; h->retry_rows = nullptr;
        h->exact_rows = nullptr;
        h->retry_tau = nullptr; h->D_dev = nullptr; h->I_dev = nullptr;
        HIPCHK(hipMalloc((void**)&h->status_dev, (size_t)padded * 4));
        HIPCHK(hipMalloc((void**)&h->ik_dev, (size_t)padded * 4));
        HIPCHK(hipMalloc((void**)&h->fail_dev, (size_t)padded * 4));
        HIPCHK(hipMalloc((void**)&h->fail2_dev, (size_t)padded * 4));
        HIPCHK(hipMalloc((void**)&h->retry_rows, (size_t)padded * 4));
        HIPCHK(hipMalloc((void**)&h->exact_rows, (size_t)padded * 4));
        HIPCHK(hipMalloc((void**)&h->retry_tau, (size_t)padded * 4));
        h->cap_rows = padded;
        h->cap_k = 0;
    }
    if (k_host > h->cap_k) {
        if (h->D_dev) (void)hipFree(h->D_dev);
        if (h->I_dev) (void)hipFree(h->I_dev);
        h->D_dev = nullptr; h->I_dev = nullptr;
        HIPCHK(hipMalloc((void**)&h->D_dev, (size_t)h->cap_rows * k_host * 4));
        HIPCHK(hipMalloc((void**)&h->I_dev, (size_t)h->cap_rows * k_host * 8));
        h->cap_k = k_host;
    }
    if (!h->pairs) {
        HIPCHK(hipMalloc((void**)&h->pairs, (size_t)DPH_POOL_CHUNKS * DPH_CHUNK_PAIRS * sizeof(uint2)));
        HIPCHK(hipMalloc((void**)&h->chunk_fill, (size_t)DPH_POOL_CHUNKS * sizeof(unsigned)));
        // two images: [0] the passes of the first attempt (what dph_scan_counters reports), [1] the retry passes
        HIPCHK(hipMalloc((void**)&h->wave_counts, (size_t)2 * h->grid * 4 * 2 * sizeof(unsigned)));
        HIPCHK(hipMalloc((void**)&h->buckets, (size_t)DPH_PASS_MAX * DPH_BUCKET_CAP * sizeof(uint64_t)));
        // [4] work-queue head of the scan, chunks claimed from the pair pool (+ padding) | [DPH_PASS_MAX] bucket counts |
        // [DPH_PASS_MAX] overflow flags: one allocation, cleared by one memset in front of every scan (dph_clear_pass_counters)
        HIPCHK(hipMalloc((void**)&h->counts_raw, (size_t)(4 + 2 * DPH_PASS_MAX) * sizeof(unsigned)));
        HIPCHK(hipMemset(h->counts_raw, 0, (size_t)(4 + 2 * DPH_PASS_MAX) * sizeof(unsigned)));
        h->bucket_counts = h->counts_raw + 4;
        HIPCHK(hipMalloc((void**)&h->tau_dev, (size_t)2 * DPH_PASS_MAX * sizeof(int)));
        HIPCHK(hipMalloc((void**)&h->counters, 8 * sizeof(int)));          // [0..3] see the struct, [4] non-finite query rows of the call
        HIPCHK(hipMemset(h->counters, 0, 8 * sizeof(int)));
        HIPCHK(hipMalloc((void**)&h->exact_x, (size_t)DPH_EXACT_ROWS_DEV * DPH_DIM * 4));
    }
    if (!h->exact_scratch) {
        // DPH_EXACT_HITS boundary hits per row the on-device fp64 fallback serves -- never more than the shard has rows (a twin, or a
        // small shard, does not need the 512 MiB the full size takes)
        const size_t hits = (size_t)std::max<int64_t>(1024, std::min<int64_t>((int64_t)DPH_EXACT_HITS, h->n_rows + 64));
        const size_t want = (size_t)DPH_EXACT_HEAD + (size_t)DPH_EXACT_ROWS_DEV * hits * 16;
        HIPCHK(hipMalloc(&h->exact_scratch, want));
        h->exact_bytes = want;
    }
    return DPH_OK;
}

// The pre-pass ladder: strides of the sampled levels, coarse -> fine.  Every level scans each `stride`-th tile under
// the bound of the previous one; the first runs cold (no bound: every sampled row is emitted), so it must be small
// enough for the pair pool (the scan waves' fair share of it) and the buckets -- and a shard small enough for ITS full
// scan to run cold needs no ladder at all.  The bound of the last level is what the full scan runs under: ~ kp * stride
// rows per query row beat it whatever the shard size (plus ~2x as many that only pass the high-digit test).
static void build_ladder(const dph_index* h, int qb, std::vector<int>& out) {
    out.clear();
    const int64_t nt = h->n_tiles;
    if (!h->ladder.empty()) {
        for (int v : h->ladder) if (v > 1 && (nt + v - 1) / v >= 1) out.push_back(v);
        return;
    }
    if (nt * DPH_TILE_ROWS <= DPH_POOL_MAX) return;             // every row fits the select pool: scan cold
    // tiles a cold level may visit: the pair regions of the scan waves and the buckets must hold every sampled row.  The
    // aim is ONE tile per scan workgroup: balanced, and the refine step of that level scores every (row, query) pair
    const int64_t wave_share = (int64_t)DPH_POOL_CHUNKS * DPH_CHUNK_PAIRS / ((int64_t)h->grid * 4);
    const int64_t per_wg = wave_share / (DPH_TILE_ROWS * DPH_QGROUP * qb);
    const int64_t cold_max = std::min<int64_t>((int64_t)h->grid * per_wg / 2, DPH_BUCKET_CAP / DPH_TILE_ROWS - 64);
    const int64_t cold_aim = std::min<int64_t>(h->grid, cold_max);
    const bool big = h->n_rows >= 100000000ll;
    const int fine = h->fine_stride > 0 ? h->fine_stride : (big ? 32 : 16);
    const int ratio = big ? 16 : 8;
    const int64_t s0 = std::max<int64_t>(2, (nt + cold_aim - 1) / cold_aim);          // stride that visits ~cold_aim tiles
    std::vector<int64_t> up;             // fine -> coarse
    int64_t s = std::max<int64_t>(2, fine);
    while ((nt + s - 1) / s > cold_max) {
        up.push_back(s);
        int64_t next = s * ratio;
        if ((nt + next - 1) / next < cold_aim) next = std::max<int64_t>(s0, s + 1);   // do not overshoot the cold level
        s = next;
    }
    // s can run cold; the top level visits ~cold_aim tiles (if that sits too close above the level below, it replaces it)
    if ((nt + s - 1) / s > cold_aim && s0 > s) { if (s0 >= s * 4) up.push_back(s); s = s0; }
    if (!up.empty() && s < up.back() * 4) up.back() = s;
    else up.push_back(s);
    for (size_t i = up.size(); i-- > 0;) out.push_back((int)std::min<int64_t>(up[i], 1 << 30));
}

// ---- unit scan of a list-major shard (dph_internal.h: DPH_PASS_MAX) --------------------------------------------------
// Worth it when the lists are long: a unit pays a few microseconds of start-up (pop the queue, load its query fragments,
// fill the pipeline) against ~1 us per 10 tiles of streaming.  Shards with many short lists (the reference's 2^20 lists)
// keep the masked scan.
static bool use_units(const dph_index* h, int nprobe) {
    if (!h->row_ids || nprobe <= 0 || !h->lists_contiguous || h->ivf_units == 0) return false;
    return h->ivf_units == 1 || h->n_tiles / std::max(1, h->nlist) >= 64;
}

static int ensure_units(dph_index* h, int nprobe) {
    const int64_t np = std::min<int64_t>(nprobe, h->nlist);
    const int64_t probes = (int64_t)DPH_PASS_MAX * np;                       // (query row, list) pairs of a pass
    const int64_t extra = probes / DPH_UNIT_SLOTS + 1;                       // chunks beyond the first of a list
    const int64_t chunk_cap = std::min<int64_t>(h->nlist, probes) + extra;
    const int64_t segs_max = (h->max_list_tiles + DPH_UNIT_TILES - 1) / DPH_UNIT_TILES;
    const int64_t unit_cap = h->total_segs + extra * std::max<int64_t>(1, segs_max);
    if (chunk_cap <= h->chunk_cap && unit_cap <= h->unit_cap) return DPH_OK;
    void* old[] = {h->slot_q, h->unit_recs, h->unit_list_recs, h->unit_frags};
    for (void* p : old) if (p) (void)hipFree(p);
    h->slot_q = nullptr; h->unit_recs = nullptr; h->unit_list_recs = nullptr; h->unit_frags = nullptr; h->chunk_cap = 0; h->unit_cap = 0;
    if (chunk_cap > (1 << 22) || unit_cap > (1 << 26)) return fail(DPH_E_ARG, "unit scan: work queue too large for this nprobe");
    HIPCHK(hipMalloc((void**)&h->slot_q, (size_t)chunk_cap * DPH_UNIT_SLOTS * 4));
    HIPCHK(hipMalloc((void**)&h->unit_recs, (size_t)unit_cap * sizeof(int4)));
    HIPCHK(hipMalloc((void**)&h->unit_list_recs, (size_t)chunk_cap * sizeof(int4)));
    HIPCHK(hipMalloc((void**)&h->unit_frags, (size_t)chunk_cap * 4 * DPH_QGROUP_FRAG_BYTES));
    h->chunk_cap = (int)chunk_cap; h->unit_cap = (int)unit_cap;
    return DPH_OK;
}

// One ladder level of a pass: flat / masked scans visit every stride-th tile of the shard; the unit scan visits the
// tiles of every probed list whose index INSIDE the list is a multiple of the stride, and its cold level (stride =
// longest list: tile 0 of every probed list) emits only the rows `rowmask` names, so that the cold sample of
// DPH_PASS_MAX rows x nprobe lists fits the pair regions.
struct ladder_level { int stride; unsigned rowmask; };

static void build_ladder_units(const dph_index* h, int n_q, int nprobe, std::vector<ladder_level>& out, double* last_ratio) {
    out.clear();
    const int64_t np = std::min<int64_t>(nprobe, h->nlist);
    const int64_t probed = np * (h->n_rows / std::max(1, h->nlist));        // rows a query row scores (average)
    *last_ratio = 1.0;
    // small enough to run cold (every row of every probed list is emitted): a scan wave emits 1024 pairs per tile of a
    // unit with full slot groups, so only shards of one- or two-tile lists qualify
    if (h->max_list_tiles <= 2 && (int64_t)n_q * probed <= (1 << 19) && probed <= DPH_POOL_MAX / 2) return;
    int r = 32;                                                               // rows of a tile the cold level samples
    while (r > 2 && (int64_t)n_q * np * r > (1 << 19)) r >>= 1;            // ~0.2 us of refine per 1000 pairs
    const int64_t s0 = np * r;
    const int fine = h->fine_stride > 0 ? h->fine_stride : 32;
    std::vector<int> up;                                                      // fine -> coarse
    for (int64_t s = std::max(2, fine); s < h->max_list_tiles && probed / s >= 4 * s0; s *= 16) up.push_back((int)s);
    out.push_back({std::max(1, h->max_list_tiles), (1u << (r / 2)) - 1u});
    for (size_t i = up.size(); i-- > 0;) out.push_back({up[i], 0xFFFFu});
    *last_ratio = up.empty() ? std::max(1.0, (double)probed / (double)s0) : (double)up[0];
}

// clean = true: the rows are also written to q.x with non-finite rows replaced by their stand-in, q.nf flags them
static void quantize(dph_index* h, dph_index::qimg& q, const float* x, int64_t n, const int* gate, hipStream_t st, bool clean = false) {
    dph_launch_quantize(x, n, gate, q.frag, q.q1, q.q2, q.qinfo, h->rmax, q.lmax, q.qaux, h->mu_dev, h->aux_lay, h->norm_unit, st,
                        clean ? q.x : nullptr, clean ? q.nf : nullptr);
}

static dph_idmap make_idmap(const dph_index* h) {
    dph_idmap m{};
    m.id_offsets = h->id_offsets; m.row_starts = h->row_starts; m.n_groups = (int)h->h_id_offsets.size();
    m.id_base = h->id_base; m.n_ids = h->n_ids;
    return m;
}

static dph_pass make_pass(dph_index* h, const dph_index::qimg& q, const float* x, int q0, int n_q, int qb) {
    dph_pass p{};
    p.db = h->db; p.n_rows = h->n_rows; p.n_tiles = h->n_tiles; p.id_base = h->id_base; p.row_ids = h->row_ids;
    p.idmap = make_idmap(h);
    p.grid = h->grid;
    p.qb = qb; p.q0 = q0; p.n_q = n_q; p.gate = nullptr; p.gate_base = 0;
    p.x = x; p.qfrag_hi = q.frag; p.q1 = q.q1; p.q2 = q.q2; p.qinfo = q.qinfo; p.lmax = q.lmax;
    p.aux = h->aux_lay.stride > 0 ? h->aux : nullptr; p.aux_lay = h->aux_lay; p.qaux = q.qaux;
    p.tilemask = nullptr;
    p.outliers = h->outliers; p.n_out = h->n_out;
    p.pairs = h->pairs; p.chunk_fill = h->chunk_fill; p.wave_counts = h->wave_counts; p.buckets = h->buckets; p.bucket_counts = h->bucket_counts;
    p.overflow = h->bucket_counts + DPH_PASS_MAX;
    p.queue_head = (int*)h->counts_raw; p.seg_tiles = h->seg_tiles;
    p.sched = h->scan_sched[qb == 2 ? 1 : 0];
    return p;
}

// Sharded two-phase form (dph_search_sample_dev / dph_search_bounded_dev): `top_out` != NULL stops after the ladder and
// returns the DPH_SAMPLE_KEEP best sampled scores per row; `tau_ext` != NULL skips the ladder and scans under the
// caller's bounds, `bound_out` then receives the upper bound of the score of any row this shard did not return.
struct search_opts {
    int32_t* top_out = nullptr;
    const int32_t* tau_ext = nullptr;
    double* bound_out = nullptr;
    int phase = 0;                       // two-stage form: 1 = quantise + sampled levels only (state kept in h->prep), 2 = the rest
};

// one pass = 128*qb query rows: [probe masks] -> ladder of sampled bounds -> full filter scan -> refine -> select.
// `rowmap` != NULL marks a retry pass (outputs scattered to rowmap[slot], larger C, failures go to fail2).
// The full scan behind a FUSED finest ladder level of stride S (run_pass): that level scanned tiles 0, S, 2S, ...; the full scan
// visits the other `visit` tiles, visited index v -> tile v + v / (S - 1) + 1, and the kernel takes the quotient as
// (v * m) >> 29 with m = ceil(2^29 / (S - 1)) (dph_scan.hip, tile_of).  With err = m (S - 1) - 2^29 that is exact as long as
// v * err < 2^29 (the fraction the multiply adds stays below 1 / (S - 1)).  Returns the number of tiles to visit and the
// multiplier, or 0 when the scan is not to be fused (no bounded level, a shard of a few tiles, a multiply that would not be
// exact or would overflow).
static int64_t fused_scan_plan(int64_t n_tiles, int stride, unsigned* m_out) {
    if (stride < 2 || n_tiles <= 4 * (int64_t)stride) return 0;
    const unsigned d = (unsigned)stride - 1u;
    const unsigned m = (unsigned)(((1ull << 29) + d - 1) / d);
    const uint64_t err = (uint64_t)m * d - (1ull << 29);
    const int64_t visit = n_tiles - (n_tiles + stride - 1) / stride;
    if ((uint64_t)visit * (err ? err : 1) >= (1ull << 29) || (uint64_t)visit * m >= (1ull << 62)) return 0;
    *m_out = m;
    return visit;
}

static int run_pass(dph_index* h, dph_pass p, int k, int nprobe, const int32_t* tau_ext, int32_t* top_out,
                    const int* rowmap, float* D, int64_t* I, int32_t* status, double* bound_out, int32_t* ik_out,
                    int32_t* fail_out, hipStream_t st, int phase = 0) {
    const bool retry = rowmap != nullptr;
    const bool units = !retry && !p.gate && use_units(h, nprobe);
    if (units) {
        int rc = ensure_units(h, nprobe);
        if (rc) return rc;
        dph_launch_coarse(p.x, p.q0, p.n_q, nullptr, 0, h->centroids, h->nlist, nprobe, h->cnorm_max, h->coarse_scores,
                          h->listmask_u, DPH_UNIT_WORDS, h->tile_list, h->n_tiles, nullptr, &h->coarse_cs, st);
        dph_launch_units_build(h->listmask_u, h->nlist, h->list_tile0, p.q1, p.q0, h->chunk_cap, h->unit_cap, h->unit_counts,
                               h->unit_counts + 4, h->slot_q, h->unit_recs, h->unit_list_recs, h->unit_frags, h->unit_offsets, h->ivf_spread, st, dph_frag_x16(h->aux_lay.stride));
        p.unit_recs = h->unit_recs; p.unit_list_recs = h->unit_list_recs; p.unit_counts = h->unit_counts; p.unit_next = h->unit_counts + 4; p.unit_launch = 0;
        p.slot_q = h->slot_q; p.unit_frags = h->unit_frags; p.listmask = h->listmask_u; p.tile_list = h->tile_list;
        p.mask_words = DPH_UNIT_WORDS;
    } else if (h->row_ids) {
        if (nprobe > 0) {
            dph_launch_coarse(p.x, p.q0, p.n_q, p.gate, p.gate_base, h->centroids, h->nlist, nprobe, h->cnorm_max, h->coarse_scores, h->listmask, 8,
                              h->tile_list, h->n_tiles, h->tilemask, &h->coarse_cs, st);
            p.tilemask = h->tilemask;
        } else {
            p.tilemask = h->onesmask;
        }
    }
    int C = k + 32;
    if (C < 2 * k) C = 2 * k;
    if (retry) C = DPH_SELECT_C_MAX;
    if (C > DPH_SELECT_C_MAX) C = DPH_SELECT_C_MAX;
    const int nset = 4;
    const int* tau = nullptr;
    int fuse_stride = 0;                     // stride of the last ladder level this pass ran itself (0 = none)
    if (phase == 2) {
        tau = h->prep.tau;                   // the prepare stage's last bound; its finest level's candidates wait in the buckets
        fuse_stride = h->prep.fuse_stride;
    } else if (tau_ext) {
        tau = tau_ext;                       // [rows of the pass]: the kernels read entries < n_q only
    } else {
        // (prepare stage: the sampled levels run on the few CUs of a side stream -- as many workgroups as those, not one per CU of the chip)
        if (phase == 1 && h->side_grid > 0) p.grid = std::min(h->side_grid, h->grid);
        std::vector<ladder_level> levels;
        double last_ratio = 1.0;             // rows of the shard (of the probed lists) per row of the last level's sample
        if (units) {
            build_ladder_units(h, p.n_q, nprobe, levels, &last_ratio);
        } else {
            std::vector<int> strides;
            build_ladder(h, p.qb, strides);
            for (int v : strides) levels.push_back({v, 0xFFFFu});
            if (!strides.empty()) last_ratio = strides.back();
        }
        // the bound of a level is its kp-th best sampled score: ~kp * last_ratio rows beat it in the full scan, and
        // that must cover the C candidates the select step wants to re-score
        int kp = h->sample_kp;
        if (!levels.empty()) {
            const int need = (int)((double)C * 1.5 / last_ratio) + 8;
            if (need > kp) kp = need;
        }
        if ((int)levels.size() + 1 > DPH_UNIT_LAUNCHES) return fail(DPH_E_STATE, "ladder deeper than the unit work-queue heads");
        for (size_t i = 0; i < levels.size(); ++i) {
            const int64_t tiles = (h->n_tiles + levels[i].stride - 1) / levels[i].stride;
            int* out = h->tau_dev + (i & 1) * DPH_PASS_MAX;
            // a level that ran under a bound leaves a small bucket the full scan can build on; a COLD level (no bound: every
            // sampled row is in its bucket, more than the select step sorts) cannot be fused
            const bool bounded_level = tau != nullptr;
            std::pair<hipEvent_t, hipEvent_t> lev{nullptr, nullptr};
            if (h->profile && !retry) {
                if (!h->prof_free.empty()) { lev = h->prof_free.back(); h->prof_free.pop_back(); }
                else { (void)hipEventCreate(&lev.first); (void)hipEventCreate(&lev.second); }
                (void)hipEventRecord(lev.first, st);
            }
            if (units) { p.unit_launch = (int)i; dph_launch_scan_units(p, true, levels[i].stride, levels[i].rowmask, tau, st); }
            else dph_launch_scan(p, true, tiles, levels[i].stride, tau, nset, st);
            if (h->profile && !retry) { (void)hipEventRecord(lev.second, st); h->prof_events_ladder.push_back(lev); }
            dph_launch_refine(p, st);
            int* top = (top_out && i + 1 == levels.size()) ? top_out : nullptr;
            dph_launch_threshold(p, kp, tau, out, top, st);
            tau = out;
            fuse_stride = bounded_level ? levels[i].stride : 0;
        }
        if (phase == 1) {
            h->prep.tau = tau;
            h->prep.fuse_stride = fuse_stride;
            return DPH_OK;
        }
        if (top_out) {
            // a shard too small for any ladder level shares nothing (INT_MIN everywhere)
            if (levels.empty())
                HIPCHK(hipMemsetD32Async((hipDeviceptr_t)top_out, (int)0x80000000, (size_t)p.n_q * DPH_SAMPLE_KEEP, st));
            return DPH_OK;
        }
    }
    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
    if (h->profile && !retry) {
        if (!h->prof_free.empty()) { ev = h->prof_free.back(); h->prof_free.pop_back(); }
        else { (void)hipEventCreate(&ev.first); (void)hipEventCreate(&ev.second); }
        (void)hipEventRecord(ev.first, st);
    }
    // Fused finest level (flat shards): the last ladder level scanned every S-th tile under the previous level's bound, and its
    // refine left those tiles' candidates -- a superset of what the final bound admits -- in the buckets.  The full scan therefore
    // visits only the OTHER tiles and accumulates: the dump is read once per batch instead of 1 + 1/S times.  Every un-emitted row
    // still has I <= tau: the fused tiles were filtered under a bound <= tau.
    int64_t full_tiles = h->n_tiles;
    if (!units && !h->row_ids && !tau_ext && !retry && h->ladder_fuse) {
        unsigned m = 0;
        const int64_t visit = fused_scan_plan(h->n_tiles, fuse_stride, &m);
        if (visit > 0) { p.skip_m = m; p.accumulate = true; full_tiles = visit; }
    }
    if (!retry) h->stats.fused_stride = p.skip_m ? fuse_stride : 0;
    if (units) { p.unit_launch = DPH_UNIT_LAUNCHES - 1; dph_launch_scan_units(p, false, 1, 0xFFFFu, tau, st); }
    else dph_launch_scan(p, false, full_tiles, 1, tau, nset, st);
    if (h->profile && !retry) { (void)hipEventRecord(ev.second, st); h->prof_events.push_back(ev); }
    dph_launch_refine(p, st);
    dph_select_args a{};
    a.lut = h->lut_dev; a.k = k; a.C = C; a.rmax = h->rmax; a.rmax_all = h->rmax_all; a.delta_max = h->delta_max;
    a.offset = h->offset; a.scale = h->scale; a.tau = tau; a.rowmap = rowmap;
    a.D = D; a.I = I; a.status = status; a.bound_out = bound_out; a.ik_out = ik_out; a.fail_out = fail_out;
    dph_launch_select(p, a, st);
    // A row whose certificate did not close although every pair is in its bucket (near-ties at the k-th place: a run of near-duplicate
    // rows) does not need another scan of the shard -- the wider re-score of the retry pass can run on the bucket it already has.
    // Second look, same buckets, C = DPH_SELECT_C_MAX, flagged rows only (the others leave at once: ~5 us when nothing failed); only
    // rows that lost pairs or still fail go on to the re-scan.  (Document-ordered dump, batch 64: one such row in every fourth batch
    // used to cost a whole extra 18.5 ms scan of the 170 M-row shard.)
    if (!retry && a.fail_out && C < DPH_SELECT_C_MAX) {
        dph_select_args w = a;
        w.C = DPH_SELECT_C_MAX;
        w.only_failed = a.fail_out;
        w.reselect_count = h->counters + 3;
        dph_launch_select(p, w, st);
    }
    if (!retry) h->stats.scan_launches++;
    return DPH_OK;
}

static int check_search_args(dph_index* h, const void* x, int64_t n, int k, int nprobe, const void* D, const void* I,
                             const char* who) {
    if (!h || !x || !D || !I || n < 0 || k <= 0 || nprobe < 0) return fail(DPH_E_ARG, std::string(who) + ": bad arguments");
    if (k > 1024) return fail(DPH_E_ARG, std::string(who) + ": k <= 1024");
    if (n > (1 << 20)) return fail(DPH_E_ARG, std::string(who) + ": at most 2^20 query rows per call");
    if (!h->finalized) return fail(DPH_E_STATE, std::string(who) + ": call dph_index_finalize first");
    if (nprobe > 0 && (!h->centroids || !h->row_ids)) return fail(DPH_E_STATE, std::string(who) + ": IVF data not set (dph_index_set_ivf)");
    return DPH_OK;
}

// The whole search of n query rows, enqueued on `st` without a host round trip:
//   1. quantise, then pass by pass (256 rows while at least 129 are left, else 128): ladder, scan, refine, select;
//   2. rows the first attempt could not certify are compacted on the device and searched again (passes of 128 rows,
//      gated by the device-side count) under a bound derived from their own k-th best integer score, with up to
//      DPH_SELECT_C_MAX re-scored candidates;
//   3. up to DPH_EXACT_ROWS_DEV rows that still fail go through the fp64 full scan.
// Afterwards status[r] = 0 for every row except the ones neither step could settle (counted by dph_search_get_stats).
static int search_core(dph_index* h, const float* x_dev, int64_t n, int k, int nprobe, float* D_dev, int64_t* I_dev,
                       int32_t* status_dev, hipStream_t st, const search_opts& opt) {
    if (opt.phase != 2) {
        quantize(h, h->q_main, x_dev, n, nullptr, st, true);
        HIPCHK(hipMemsetAsync(h->counters + 3, 0, 2 * sizeof(int), st));   // rows the in-pass wide re-select certifies (run_pass); non-finite rows
    }
    // from here on the rows are read from the quantiser's clean copy (phase 2: the one the prepare stage left), never from the caller's buffer
    x_dev = h->q_main.x;
    const bool sample_only = opt.top_out != nullptr;
    for (int64_t q0 = 0; q0 < n;) {
        const int64_t left = n - q0;
        const bool units = use_units(h, nprobe);                    // a unit pass serves up to DPH_PASS_MAX rows
        const int qb = units ? 1 : ((left > DPH_QROWS && h->max_qb >= 2) ? 2 : 1);
        const int nq = (int)std::min<int64_t>(left, units ? (int64_t)DPH_PASS_MAX : (int64_t)DPH_QROWS * qb);
        dph_pass p = make_pass(h, h->q_main, x_dev, (int)q0, nq, qb);
        int rc = run_pass(h, p, k, nprobe, opt.tau_ext ? opt.tau_ext + q0 : nullptr,
                          sample_only ? opt.top_out + q0 * DPH_SAMPLE_KEEP : nullptr, nullptr, D_dev, I_dev, status_dev,
                          opt.bound_out, h->ik_dev, h->fail_dev, st, opt.phase);
        if (rc) return rc;
        q0 += nq;
    }
    if (sample_only || opt.phase == 1) { HIPCHK(hipGetLastError()); return DPH_OK; }
    if (!h->retry_chain) {
        // measurements / diagnostics: no re-scan, no fp64 fallback -- the counters say how many rows the first attempt left open, their
        // status stays 1 (and the buckets of the last pass stay readable: dph_debug_bucket_counts)
        dph_launch_compact_failing(h->fail_dev, n, 0, nullptr, h->retry_rows, h->counters + 0, nullptr, 0, st);
        HIPCHK(hipMemcpyAsync(h->counters + 1, h->counters + 0, sizeof(int), hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(h->counters + 2, h->counters + 0, sizeof(int), hipMemcpyDeviceToDevice, st));
        dph_launch_nonfinite_fix(h->q_main.nf, n, k, D_dev, I_dev, status_dev, opt.bound_out, h->counters + 4, st);
        HIPCHK(hipGetLastError());
        return DPH_OK;
    }

    // ---- 2. on-device retry of the rows that failed
    dph_launch_compact_failing(h->fail_dev, n, 0, x_dev, h->retry_rows, h->counters + 0, h->q_retry.x, (int)n, st);
    quantize(h, h->q_retry, h->q_retry.x, n, h->counters + 0, st);
    dph_launch_retry_tau(h->counters + 0, n, h->retry_rows, h->ik_dev, h->q_retry.qinfo, h->rmax, h->delta_max, h->scale,
                         h->retry_tau, st);
    HIPCHK(hipMemsetAsync(h->fail2_dev, 0, (size_t)n * 4, st));
    for (int64_t q0 = 0; q0 < n; q0 += DPH_QROWS) {
        const int nq = (int)std::min<int64_t>(n - q0, DPH_QROWS);
        dph_pass p = make_pass(h, h->q_retry, h->q_retry.x, (int)q0, nq, 1);
        p.gate = h->counters + 0;
        p.gate_base = (int)q0;
        p.wave_counts = h->wave_counts + (size_t)h->grid * 8;
        int rc = run_pass(h, p, k, nprobe, h->retry_tau + q0, nullptr, h->retry_rows, D_dev, I_dev, status_dev, opt.bound_out,
                          nullptr, h->fail2_dev, st);
        if (rc) return rc;
    }
    // ---- 3. fp64 full scan for (up to DPH_EXACT_ROWS_DEV of) the rows the retry could not settle
    dph_launch_compact_failing(h->fail2_dev, n, 0, x_dev, h->exact_rows, h->counters + 1, h->exact_x, DPH_EXACT_ROWS_DEV, st);
    const unsigned* mask = nullptr;
    if (h->row_ids && nprobe > 0) {
        dph_launch_coarse(h->exact_x, 0, DPH_EXACT_ROWS_DEV, h->counters + 1, 0, h->centroids, h->nlist, nprobe, h->cnorm_max, h->coarse_scores, h->listmask,
                          8, h->tile_list, h->n_tiles, h->tilemask, &h->coarse_cs, st);
        mask = h->tilemask;
    }
    dph_launch_exact(h->db, h->n_rows, make_idmap(h), h->exact_x, h->lut_dev, h->exact_rows, h->counters + 1, DPH_EXACT_ROWS_DEV,
                     k, h->row_ids, mask, D_dev, I_dev, status_dev, h->exact_scratch, h->exact_bytes, st);
    // the result of a non-finite query row: ids -1, scores -FLT_MAX, status DPH_ROW_NONFINITE (its stand-in was searched like any row)
    dph_launch_nonfinite_fix(h->q_main.nf, n, k, D_dev, I_dev, status_dev, opt.bound_out, h->counters + 4, st);
    // rows still flagged 1 (more than DPH_EXACT_ROWS_DEV failures, or boundary ties beyond the fp64 scan's buffer)
    dph_launch_compact_failing(status_dev, n, 1, nullptr, h->retry_rows, h->counters + 2, nullptr, 0, st);
    HIPCHK(hipGetLastError());
    return DPH_OK;
}

static int read_counters(dph_index* h, hipStream_t st, int out[5]) {
    HIPCHK(hipMemcpyAsync(out, h->counters, 5 * sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return DPH_OK;
}

static int pq_nprobe(const dph_index* h, int nprobe) { return nprobe > 0 ? nprobe : (h->default_nprobe > 0 ? h->default_nprobe : 256); }

static int search_dev_impl(dph_index* h, const float* x_dev, int64_t n, int k, int nprobe, float* D_dev, int64_t* I_dev,
                           int32_t* status_dev, void* stream, const char* who, const search_opts& opt = search_opts()) {
    if (h && h->pq) {
        if (!x_dev || !D_dev || !I_dev || !status_dev || n < 0 || k <= 0 || k > 1024) return fail(DPH_E_ARG, std::string(who) + ": bad arguments");
        if (opt.top_out || opt.tau_ext) return fail(DPH_E_STATE, std::string(who) + ": the sharded two-phase entry points do not serve PQ indexes");
        if (!h->finalized) return fail(DPH_E_STATE, std::string(who) + ": call dph_index_finalize first");
        if (n == 0) return DPH_OK;
        h->stats = dph_search_stats{};
        h->stats.rows = (int32_t)n;
        h->stats.certified_fast = (int32_t)n;
        h->stats_pending = false;
        const int rc = dph_pq_search_dev(h->pq, x_dev, n, k, pq_nprobe(h, nprobe), D_dev, I_dev, status_dev, (hipStream_t)stream);
        return rc ? fail(rc, std::string(who) + ": " + dph_pq_error()) : DPH_OK;
    }
    int rc = check_search_args(h, x_dev, n, k, nprobe, D_dev, I_dev, who);
    if (rc) return rc;
    if (!status_dev) return fail(DPH_E_ARG, std::string(who) + ": status buffer is NULL");
    if (n == 0) return DPH_OK;
    HIPCHK(hipSetDevice(h->device));
    rc = ensure_scratch(h, n, 0);
    if (rc) return rc;
    h->stats = dph_search_stats{};
    h->stats.rows = (int32_t)n;
    h->stats_pending = true;
    return search_core(h, x_dev, n, k, nprobe, D_dev, I_dev, status_dev, (hipStream_t)stream, opt);
}

static int search_host_impl(dph_index* h, const float* x, int64_t n, int k, int nprobe, float* D, int64_t* I, const char* who) {
    if (h && h->pq) {
        if (!x || !D || !I || n < 0 || k <= 0 || k > 1024) return fail(DPH_E_ARG, std::string(who) + ": bad arguments");
        if (n == 0) return DPH_OK;
        HIPCHK(hipSetDevice(h->device));
        char* blob = nullptr;
        const size_t b_x = ((size_t)n * DPH_DIM * 4 + 255) / 256 * 256, b_d = ((size_t)n * k * 4 + 255) / 256 * 256, b_i = ((size_t)n * k * 8 + 255) / 256 * 256;
        HIPCHK(hipMalloc((void**)&blob, b_x + b_d + b_i + (size_t)n * 4));
        int rc = DPH_OK;
        std::vector<int32_t> status((size_t)n);
        if (hipMemcpy(blob, x, (size_t)n * DPH_DIM * 4, hipMemcpyHostToDevice) != hipSuccess) rc = fail(DPH_E_HIP, std::string(who) + ": copy failed");
        if (!rc) rc = search_dev_impl(h, (const float*)blob, n, k, nprobe, (float*)(blob + b_x), (int64_t*)(blob + b_x + b_d),
                                      (int32_t*)(blob + b_x + b_d + b_i), nullptr, who);
        if (!rc && (hipMemcpy(D, blob + b_x, (size_t)n * k * 4, hipMemcpyDeviceToHost) != hipSuccess ||
                    hipMemcpy(I, blob + b_x + b_d, (size_t)n * k * 8, hipMemcpyDeviceToHost) != hipSuccess ||
                    hipMemcpy(status.data(), blob + b_x + b_d + b_i, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess))
            rc = fail(DPH_E_HIP, std::string(who) + ": copy failed");
        (void)hipFree(blob);
        if (rc) return rc;
        int bad = 0, nf = 0;
        for (int32_t v : status) { bad += v == 1; nf += v == DPH_ROW_NONFINITE; }
        h->stats.uncertified = bad;
        h->stats.nonfinite = nf;
        h->stats.certified_fast = (int32_t)n - bad - nf;
        if (bad) return fail(DPH_E_UNCERTIFIED, std::string(who) + ": candidate buffers of the PQ scan overflowed (boundary ties)");
        return DPH_OK;
    }
    int rc = check_search_args(h, x, n, k, nprobe, D, I, who);
    if (rc) return rc;
    if (n == 0) return DPH_OK;
    HIPCHK(hipSetDevice(h->device));
    rc = ensure_scratch(h, n, k);
    if (rc) return rc;
    hipStream_t st = nullptr;
    h->stats = dph_search_stats{};
    h->stats.rows = (int32_t)n;
    h->stats_pending = false;
    HIPCHK(hipMemcpyAsync(h->q_main.x, x, (size_t)n * DPH_DIM * 4, hipMemcpyHostToDevice, st));
    rc = search_core(h, h->q_main.x, n, k, nprobe, h->D_dev, h->I_dev, h->status_dev, st, search_opts());
    if (rc) return rc;
    int c[5] = {0, 0, 0, 0, 0};
    rc = read_counters(h, st, c);
    if (rc) return rc;
    h->stats.nonfinite = c[4];
    h->stats.certified_fast = (int32_t)(n - c[0] - c[3]);
    h->stats.certified_wide = c[0] - c[1] + c[3];
    h->stats.certified_reselect = c[3];
    int left = c[2];
    if (left > 0) {
        // more failures than the on-device fallback serves per call: feed it the rest, DPH_EXACT_ROWS_DEV at a time
        std::vector<int32_t> status((size_t)n);
        HIPCHK(hipMemcpy(status.data(), h->status_dev, (size_t)n * 4, hipMemcpyDeviceToHost));
        std::vector<int32_t> todo;
        for (int64_t r = 0; r < n; ++r) if (status[r] == 1) todo.push_back((int32_t)r);
        // (the rows the device pass already tried -- one tightening round -- get the three rounds of this loop as well)
        std::vector<int32_t> rest(todo);
        for (size_t i0 = 0; i0 < rest.size(); i0 += DPH_EXACT_ROWS_DEV) {
            const int m = (int)std::min<size_t>(DPH_EXACT_ROWS_DEV, rest.size() - i0);
            HIPCHK(hipMemcpyAsync(h->exact_rows, rest.data() + i0, (size_t)m * 4, hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(h->counters + 1, &m, sizeof(int), hipMemcpyHostToDevice, st));
            dph_launch_gather_rows(h->q_main.x, h->exact_rows, h->counters + 1, h->exact_x, DPH_EXACT_ROWS_DEV, st);
            const unsigned* mask = nullptr;
            if (h->row_ids && nprobe > 0) {
                dph_launch_coarse(h->exact_x, 0, DPH_EXACT_ROWS_DEV, h->counters + 1, 0, h->centroids, h->nlist, nprobe,
                                  h->cnorm_max, h->coarse_scores, h->listmask, 8, h->tile_list, h->n_tiles, h->tilemask, &h->coarse_cs, st);
                mask = h->tilemask;
            }
            dph_launch_exact(h->db, h->n_rows, make_idmap(h), h->exact_x, h->lut_dev, h->exact_rows, h->counters + 1,
                             DPH_EXACT_ROWS_DEV, k, h->row_ids, mask, h->D_dev, h->I_dev, h->status_dev, h->exact_scratch,
                             h->exact_bytes, st, 3);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(st));
        }
        HIPCHK(hipMemcpy(status.data(), h->status_dev, (size_t)n * 4, hipMemcpyDeviceToHost));
        left = 0;
        for (int64_t r = 0; r < n; ++r) left += status[r] == 1;
    }
    h->stats.uncertified = left;
    h->stats.exact_fallback = c[1] - left;
    HIPCHK(hipMemcpy(D, h->D_dev, (size_t)n * k * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(I, h->I_dev, (size_t)n * k * 8, hipMemcpyDeviceToHost));
    if (left > 0) return fail(DPH_E_UNCERTIFIED, std::string(who) + ": boundary ties exceed the exact-scan buffer");
    return DPH_OK;
}

static int default_nprobe(const dph_index* h) { return (h && ((h->centroids && h->row_ids) || h->pq)) ? h->default_nprobe : 0; }

int dph_search_dev(dph_index* h, const float* x_dev, int64_t n, int k, float* D_dev, int64_t* I_dev,
                   int32_t* status_dev, void* stream) {
    return search_dev_impl(h, x_dev, n, k, default_nprobe(h), D_dev, I_dev, status_dev, stream, "dph_search_dev");
}
int dph_search(dph_index* h, const float* x, int64_t n, int k, float* D, int64_t* I) {
    return search_host_impl(h, x, n, k, default_nprobe(h), D, I, "dph_search");
}
int dph_search_ivf_dev(dph_index* h, const float* x_dev, int64_t n, int k, int nprobe, float* D_dev, int64_t* I_dev,
                       int32_t* status_dev, void* stream) {
    if (nprobe <= 0) return fail(DPH_E_ARG, "dph_search_ivf_dev: nprobe must be > 0");
    return search_dev_impl(h, x_dev, n, k, nprobe, D_dev, I_dev, status_dev, stream, "dph_search_ivf_dev");
}
int dph_search_ivf(dph_index* h, const float* x, int64_t n, int k, int nprobe, float* D, int64_t* I) {
    if (nprobe <= 0) return fail(DPH_E_ARG, "dph_search_ivf: nprobe must be > 0");
    return search_host_impl(h, x, n, k, nprobe, D, I, "dph_search_ivf");
}

// ---- two-phase search of a range-sharded dump under a bound taken over the union of all shards' samples
int dph_search_sample_dev(dph_index* h, const float* x_dev, int64_t n, int32_t* top_dev, void* stream) {
    if (!h || !x_dev || !top_dev || n < 0 || n > (1 << 20)) return fail(DPH_E_ARG, "dph_search_sample_dev: bad arguments");
    if (!h->finalized) return fail(DPH_E_STATE, "dph_search_sample_dev: call dph_index_finalize first");
    if (n == 0) return DPH_OK;
    HIPCHK(hipSetDevice(h->device));
    int rc = ensure_scratch(h, n, 0);
    if (rc) return rc;
    search_opts opt;
    opt.top_out = top_dev;
    return search_core(h, x_dev, n, 1, default_nprobe(h), nullptr, nullptr, nullptr, (hipStream_t)stream, opt);
}

int dph_union_bounds_dev(int device, const int32_t* top_parts, int n_parts, int64_t n, int32_t* tau_dev, void* stream) {
    if (!top_parts || !tau_dev || n_parts <= 0 || n < 0) return fail(DPH_E_ARG, "dph_union_bounds_dev: bad arguments");
    HIPCHK(hipSetDevice(device));
    if (n > 0) dph_launch_union_bounds(top_parts, n_parts, n, tau_dev, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return DPH_OK;
}

int dph_search_bounded_dev(dph_index* h, const float* x_dev, int64_t n, int k, const int32_t* tau_dev, float* D_dev,
                           int64_t* I_dev, int32_t* status_dev, double* bound_dev, void* stream) {
    if (!tau_dev || !bound_dev) return fail(DPH_E_ARG, "dph_search_bounded_dev: null buffer");
    search_opts opt;
    opt.tau_ext = tau_dev;
    opt.bound_out = bound_dev;
    return search_dev_impl(h, x_dev, n, k, default_nprobe(h), D_dev, I_dev, status_dev, stream, "dph_search_bounded_dev", opt);
}

// ---- the search in two stages, for two batches in flight on one shard (dph_index_create_twin; one handle per batch in flight).
// prepare: quantise + the sampled levels (every scan launch of them with `side_grid` workgroups) -- leaves the bound and the finest
// level's candidates in the handle; finish: the full scan (fused with that level), refine, select, retry chain.  Together they enqueue
// exactly the launches of dph_search_dev, with the same arguments: the same result.  One pass only (n <= 256 rows), flat shards.
static int two_stage_check(dph_index* h, const float* x_dev, int64_t n, int k, const char* who) {
    if (!h || !x_dev || n <= 0 || k <= 0 || k > 1024) return fail(DPH_E_ARG, std::string(who) + ": bad arguments");
    if (!h->finalized) return fail(DPH_E_STATE, std::string(who) + ": call dph_index_finalize first");
    if (h->pq || h->row_ids) return fail(DPH_E_STATE, std::string(who) + ": flat shards only");
    if (n > (int64_t)DPH_QROWS * std::min(h->max_qb, DPH_MAX_QB)) return fail(DPH_E_ARG, std::string(who) + ": one pass per call (at most 128 x max_qb query rows)");
    return DPH_OK;
}

int dph_search_prepare_dev(dph_index* h, const float* x_dev, int64_t n, int k, void* stream) {
    int rc = two_stage_check(h, x_dev, n, k, "dph_search_prepare_dev");
    if (rc) return rc;
    HIPCHK(hipSetDevice(h->device));
    rc = ensure_scratch(h, n, 0);
    if (rc) return rc;
    h->prep = dph_index::prepared();
    search_opts opt;
    opt.phase = 1;
    rc = search_core(h, x_dev, n, k, 0, nullptr, nullptr, nullptr, (hipStream_t)stream, opt);
    if (rc) return rc;
    h->prep.valid = true; h->prep.n = n; h->prep.k = k;
    return DPH_OK;
}

int dph_search_finish_dev(dph_index* h, const float* x_dev, int64_t n, int k, float* D_dev, int64_t* I_dev, int32_t* status_dev, void* stream) {
    int rc = two_stage_check(h, x_dev, n, k, "dph_search_finish_dev");
    if (rc) return rc;
    if (!D_dev || !I_dev || !status_dev) return fail(DPH_E_ARG, "dph_search_finish_dev: null buffer");
    if (!h->prep.valid || h->prep.n != n || h->prep.k != k)
        return fail(DPH_E_STATE, "dph_search_finish_dev: no matching dph_search_prepare_dev on this handle (same n, same k)");
    HIPCHK(hipSetDevice(h->device));
    h->prep.valid = false;
    h->stats = dph_search_stats{};
    h->stats.rows = (int32_t)n;
    h->stats_pending = true;
    search_opts opt;
    opt.phase = 2;
    return search_core(h, x_dev, n, k, 0, D_dev, I_dev, status_dev, (hipStream_t)stream, opt);
}

// A stream whose kernels run on the CUs [first_cu, first_cu + n_cus) of the device's CU-mask bit order only (MI355X: bit i is a CU of
// XCD i % 8, so any run of 8 k bits takes k CUs from every XCD -- profiles/r04_cu_mask_probe.txt).  Two such streams over disjoint
// ranges run their kernels side by side whatever either of them occupies.
int dph_stream_create_cu_range(int device, int first_cu, int n_cus, void** stream_out) {
    if (!stream_out || first_cu < 0 || n_cus <= 0) return fail(DPH_E_ARG, "dph_stream_create_cu_range: bad arguments");
    HIPCHK(hipSetDevice(device));
    int cus = 0;
    HIPCHK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
    if (first_cu + n_cus > cus) return fail(DPH_E_ARG, "dph_stream_create_cu_range: range beyond the device's CUs");
    std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0u);
    for (int i = first_cu; i < first_cu + n_cus; ++i) mask[(size_t)i / 32] |= 1u << (i % 32);
    hipStream_t st = nullptr;
    HIPCHK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    *stream_out = (void*)st;
    return DPH_OK;
}
int dph_stream_destroy(void* stream) {
    if (stream) HIPCHK(hipStreamDestroy((hipStream_t)stream));
    return DPH_OK;
}

int dph_search_get_stats(dph_index* h, dph_search_stats* out) {
    if (!h || !out) return fail(DPH_E_ARG, "null");
    if (h->stats_pending) {
        // device-pointer call: the counters live on the device until somebody asks (this synchronises the device)
        HIPCHK(hipSetDevice(h->device));
        int c[5] = {0, 0, 0, 0, 0};
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(c, h->counters, sizeof(c), hipMemcpyDeviceToHost));
        h->stats.nonfinite = c[4];
        // c[3]: rows the first look could not certify and the wider look at the same bucket could (no re-scan)
        h->stats.certified_fast = h->stats.rows - c[0] - c[3];
        h->stats.certified_wide = c[0] - c[1] + c[3];
        h->stats.certified_reselect = c[3];
        h->stats.exact_fallback = c[1] - c[2];
        h->stats.uncertified = c[2];
        h->stats_pending = false;
    }
    *out = h->stats;
    return DPH_OK;
}

// pairs the scan kernels emitted / emit-path triggers (wave level) in the LAST scan launch on this handle
int dph_scan_counters(dph_index* h, int64_t* pairs_out, int64_t* triggers_out) {
    if (!h || !pairs_out || !triggers_out) return fail(DPH_E_ARG, "null");
    if (!h->wave_counts) return fail(DPH_E_STATE, "dph_scan_counters: no search yet");
    HIPCHK(hipSetDevice(h->device));
    std::vector<unsigned> wc((size_t)h->grid * 8);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(wc.data(), h->wave_counts, wc.size() * 4, hipMemcpyDeviceToHost));
    int64_t a = 0, b = 0;
    for (size_t i = 0; i < wc.size(); i += 2) { a += wc[i]; b += wc[i + 1]; }
    *pairs_out = a; *triggers_out = b;
    return DPH_OK;
}

int dph_debug_wave_pairs(dph_index* h, int image, uint32_t* pairs_out, int out_cap, int* n_waves) {
    if (!h || !pairs_out || !n_waves || (image != 0 && image != 1)) return fail(DPH_E_ARG, "dph_debug_wave_pairs: bad arguments");
    if (!h->wave_counts) return fail(DPH_E_STATE, "dph_debug_wave_pairs: no search yet");
    const int nw = h->grid * 4;
    if (out_cap < nw) return fail(DPH_E_ARG, "dph_debug_wave_pairs: buffer too small");
    HIPCHK(hipSetDevice(h->device));
    std::vector<unsigned> wc((size_t)nw * 2);
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(wc.data(), h->wave_counts + (size_t)image * nw * 2, wc.size() * 4, hipMemcpyDeviceToHost));
    for (int w = 0; w < nw; ++w) pairs_out[w] = wc[(size_t)2 * w];
    *n_waves = nw;
    return DPH_OK;
}

int dph_reconstruct(dph_index* h, int64_t id, float* out768) {
    if (!h || !out768) return fail(DPH_E_ARG, "null");
    if (h->pq) {
        // IndexIVFPQ::reconstruct through the direct map: centroid + decoded residual, in the ROTATED space (index.py:31,286)
        HIPCHK(hipSetDevice(h->device));
        char* blob = nullptr;
        HIPCHK(hipMalloc((void**)&blob, 16 + DPH_DIM * 4));
        int32_t found = 0;
        int rc = DPH_OK;
        if (hipMemcpy(blob, &id, 8, hipMemcpyHostToDevice) != hipSuccess) rc = fail(DPH_E_HIP, "dph_reconstruct: copy failed");
        if (!rc) { rc = dph_pq_reconstruct_dev(h->pq, (const int64_t*)blob, 1, (float*)(blob + 16), (int32_t*)(blob + 8), nullptr); if (rc) fail(rc, dph_pq_error()); }
        if (!rc && (hipMemcpy(&found, blob + 8, 4, hipMemcpyDeviceToHost) != hipSuccess ||
                    hipMemcpy(out768, blob + 16, DPH_DIM * 4, hipMemcpyDeviceToHost) != hipSuccess)) rc = fail(DPH_E_HIP, "dph_reconstruct: copy failed");
        (void)hipFree(blob);
        if (rc) return rc;
        return found ? DPH_OK : fail(DPH_E_NOTFOUND, "dph_reconstruct: id not in the index");
    }
    const int64_t local = host_local_of_id(h, id);
    if (local < 0) return fail(DPH_E_NOTFOUND, "dph_reconstruct: id not in this shard");
    HIPCHK(hipSetDevice(h->device));
    const int64_t srow = h->h_inv.empty() ? local : (int64_t)h->h_inv[(size_t)local];
    int8_t row[DPH_DIM];
    HIPCHK(hipMemcpy(row, h->db + srow * DPH_DIM, DPH_DIM, hipMemcpyDeviceToHost));
    for (int j = 0; j < DPH_DIM; ++j) out768[j] = h->lut_host[(int)row[j] + 128];
    return DPH_OK;
}

int dph_id2docword(dph_index* h, const int64_t* I, int64_t n, int32_t* doc, int32_t* word) {
    if (!h || !I || !doc || !word || n < 0) return fail(DPH_E_ARG, "null");
    if (h->twin_of) h = h->twin_of;                    // the host copy of idx2id lives with the index
    if (h->h_row2doc.empty() && h->n_ids > 0) return fail(DPH_E_STATE, "dph_id2docword: idx2id not set");
    const int64_t first = h->h_id_offsets.empty() ? h->id_base : h->h_id_offsets[0];
    for (int64_t i = 0; i < n; ++i) {
        if (h->n_ids == 0) { doc[i] = -1; word[i] = -1; continue; }      // empty shard: nothing to clip to
        int64_t local = host_local_of_id(h, I[i]);
        if (local < 0) local = I[i] < first ? 0 : h->n_ids - 1;          // np.clip (index.py:133); a hole between sub-indexes clips up
        doc[i] = h->h_row2doc[(size_t)local];
        word[i] = h->h_row2word[(size_t)local];
    }
    return DPH_OK;
}

int dph_rescore_dev(dph_index* h, int direction, const float* qhalf_dev, int64_t n_q, int k, int L,
                    const int64_t* ids_dev, const int32_t* doc_dev, const int32_t* word_dev, const float* first_dev,
                    int32_t* pred_word_dev, double* best_dev, int32_t* argslot_dev, float* vecs_dev, void* stream) {
    if (!h || !qhalf_dev || !ids_dev || !first_dev || !pred_word_dev || !best_dev || !argslot_dev || n_q < 0 || k <= 0 || L <= 0)
        return fail(DPH_E_ARG, "dph_rescore_dev: bad arguments");
    if (direction != 0 && direction != 1) return fail(DPH_E_ARG, "dph_rescore_dev: direction is 0 or 1");
    if (!h->doc_ids || !h->f2o_off || !h->f2o) return fail(DPH_E_STATE, "dph_rescore_dev: f2o metadata not set");
    if ((!doc_dev || !word_dev) && !h->row2doc) return fail(DPH_E_STATE, "dph_rescore_dev: idx2id not set");
    HIPCHK(hipSetDevice(h->device));
    if (h->pq) {
        const int rc = dph_pq_window(h->pq, direction, make_idmap(h), qhalf_dev, n_q, k, L, ids_dev, doc_dev, word_dev, first_dev, h->row2doc,
                                     h->row2word, h->doc_ids, h->n_docs, h->f2o_off, h->f2o, pred_word_dev, best_dev, argslot_dev, vecs_dev,
                                     (hipStream_t)stream);
        return rc ? fail(rc, std::string("dph_rescore_dev: ") + dph_pq_error()) : DPH_OK;
    }
    dph_launch_window(direction, h->db, h->n_rows, make_idmap(h), h->lut_dev, qhalf_dev, n_q * k, k, L, ids_dev, doc_dev,
                      word_dev, first_dev, h->row2doc, h->row2word, h->doc_ids, h->n_docs, h->f2o_off, h->f2o, h->inv_row,
                      pred_word_dev, best_dev, argslot_dev, vecs_dev, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return DPH_OK;
}

int dph_rescore(dph_index* h, int direction, const float* qhalf, int64_t n_q, int k, int L, const int64_t* ids,
                const int32_t* doc, const int32_t* word, const float* first, int32_t* pred_word, double* best,
                int32_t* argslot, float* vecs) {
    if (!h || !qhalf || !ids || !doc || !word || !first || !pred_word || !best || !argslot)
        return fail(DPH_E_ARG, "dph_rescore: null argument");
    HIPCHK(hipSetDevice(h->device));
    const int64_t nc = n_q * k;
    if (nc == 0) return DPH_OK;
    char* blob = nullptr;
    const size_t b_q = (size_t)n_q * DPH_DIM * 4, b_ids = (size_t)nc * 8, b_i32 = (size_t)nc * 4, b_f64 = (size_t)nc * 8;
    const size_t b_vec = vecs ? (size_t)nc * 2 * DPH_DIM * 4 : 0;
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = off; off += (b + 255) / 256 * 256; return o; };
    const size_t o_q = take(b_q), o_ids = take(b_ids), o_doc = take(b_i32), o_word = take(b_i32), o_first = take(b_i32),
                 o_pw = take(b_i32), o_best = take(b_f64), o_arg = take(b_i32), o_vec = take(b_vec);
    HIPCHK(hipMalloc((void**)&blob, off));
    int rc = DPH_OK;
    auto up = [&](size_t o, const void* p, size_t b) { if (hipMemcpy(blob + o, p, b, hipMemcpyHostToDevice) != hipSuccess) rc = DPH_E_HIP; };
    up(o_q, qhalf, b_q); up(o_ids, ids, b_ids); up(o_doc, doc, b_i32); up(o_word, word, b_i32); up(o_first, first, b_i32);
    if (!rc)
        rc = dph_rescore_dev(h, direction, (const float*)(blob + o_q), n_q, k, L, (const int64_t*)(blob + o_ids),
                             (const int32_t*)(blob + o_doc), (const int32_t*)(blob + o_word), (const float*)(blob + o_first),
                             (int32_t*)(blob + o_pw), (double*)(blob + o_best), (int32_t*)(blob + o_arg),
                             vecs ? (float*)(blob + o_vec) : nullptr, nullptr);
    auto down = [&](void* p, size_t o, size_t b) { if (hipMemcpy(p, blob + o, b, hipMemcpyDeviceToHost) != hipSuccess) rc = DPH_E_HIP; };
    if (!rc) { down(pred_word, o_pw, b_i32); down(best, o_best, b_f64); down(argslot, o_arg, b_i32); if (vecs) down(vecs, o_vec, b_vec); }
    (void)hipFree(blob);
    if (rc == DPH_E_HIP) return fail(DPH_E_HIP, "dph_rescore: copy failed");
    return rc;
}

int dph_ivf_assign_dev(int device, const float* x_dev, int64_t n, const float* centroids_dev, int nlist, const float* bias_dev,
                       float* scores_dev, int32_t* best_dev, float* gap_dev, void* stream) {
    (void)scores_dev;                    // ABI 2 took a [n, nlist] scratch here; the fused kernel writes no scores
    if (!x_dev || !centroids_dev || !best_dev || !gap_dev || n < 0 || nlist <= 0)
        return fail(DPH_E_ARG, "dph_ivf_assign_dev: bad arguments");
    HIPCHK(hipSetDevice(device));
    if (n > 0) dph_launch_assign(x_dev, false, nullptr, n, centroids_dev, nlist, bias_dev, best_dev, gap_dev, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return DPH_OK;
}

int dph_index_assign_dev(dph_index* h, int64_t row0, int64_t n, const float* centroids_dev, int nlist, const float* bias_dev,
                         int32_t* best_dev, float* gap_dev, void* stream) {
    if (!h || !centroids_dev || !best_dev || !gap_dev || row0 < 0 || n < 0 || row0 + n > h->n_rows || nlist <= 0)
        return fail(DPH_E_ARG, "dph_index_assign_dev: bad arguments");
    HIPCHK(hipSetDevice(h->device));
    if (n > 0)
        dph_launch_assign(h->db + row0 * DPH_DIM, true, h->lut_dev, n, centroids_dev, nlist, bias_dev, best_dev, gap_dev,
                          (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return DPH_OK;
}

int dph_index_gather_rows_dev(dph_index* h, const int64_t* rows_idx_dev, int64_t m, int8_t* sample_dev, void* stream) {
    if (!h || !rows_idx_dev || !sample_dev || m < 0) return fail(DPH_E_ARG, "dph_index_gather_rows_dev: bad arguments");
    HIPCHK(hipSetDevice(h->device));
    dph_launch_gather_sample(h->db, h->n_rows, rows_idx_dev, m, sample_dev, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return DPH_OK;
}

int dph_kmeans_step_dev(dph_index* h, const int8_t* rows_dev, int64_t m, float* centroids_dev, int nlist, const float* bias_dev,
                        int spherical, int32_t* assign_dev, float* gap_dev, uint32_t* counts_dev, void* stream) {
    if (!h || !rows_dev || !centroids_dev || !assign_dev || !gap_dev || !counts_dev || m < 0 || nlist <= 0 || nlist > (1 << 20))
        return fail(DPH_E_ARG, "dph_kmeans_step_dev: bad arguments");
    DPH_NOT_TWINNED(h, "dph_kmeans_step_dev");           // (it allocates per-index buffers a twin's destroy would not free)
    HIPCHK(hipSetDevice(h->device));
    if (h->kmeans_nlist < nlist) {
        if (h->kmeans_sums) { (void)hipFree(h->kmeans_sums); h->kmeans_sums = nullptr; h->kmeans_nlist = 0; }
        HIPCHK(hipMalloc((void**)&h->kmeans_sums, (size_t)nlist * DPH_DIM * sizeof(long long)));
        h->kmeans_nlist = nlist;
    }
    hipStream_t st = (hipStream_t)stream;
    if (m > 0) dph_launch_assign(rows_dev, true, h->lut_dev, m, centroids_dev, nlist, bias_dev, assign_dev, gap_dev, st);
    dph_launch_kmeans_update(rows_dev, assign_dev, m, nlist, h->offset, h->scale, spherical, h->kmeans_sums, counts_dev, centroids_dev, st);
    HIPCHK(hipGetLastError());
    return DPH_OK;
}

int64_t dph_index_stored_rows(const dph_index* h) { return h ? h->n_rows : 0; }

int dph_score_vecs_dev(int device, const float* q_dev, const float* vecs_dev, int64_t n_b, int64_t m, float* out_dev, void* stream) {
    if (!q_dev || !vecs_dev || !out_dev || n_b < 0 || m < 0) return fail(DPH_E_ARG, "dph_score_vecs_dev: bad arguments");
    HIPCHK(hipSetDevice(device));
    dph_launch_score_vecs(q_dev, vecs_dev, n_b, m, out_dev, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return DPH_OK;
}
int dph_score_vecs_bwd_dev(int device, const float* grad_dev, const float* vecs_dev, int64_t n_b, int64_t m, float* grad_q_dev,
                           void* stream) {
    if (!grad_dev || !vecs_dev || !grad_q_dev || n_b < 0 || m < 0) return fail(DPH_E_ARG, "dph_score_vecs_bwd_dev: bad arguments");
    HIPCHK(hipSetDevice(device));
    dph_launch_score_vecs_bwd(grad_dev, vecs_dev, n_b, m, grad_q_dev, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return DPH_OK;
}
int dph_dense_logits_dev(int device, const float* start_logits_dev, const float* end_logits_dev, int64_t n_b, int64_t T,
                         float* out_dev, void* stream) {
    if (!start_logits_dev || !end_logits_dev || !out_dev || n_b < 0 || T < 0) return fail(DPH_E_ARG, "dph_dense_logits_dev: bad arguments");
    HIPCHK(hipSetDevice(device));
    dph_launch_dense_logits(start_logits_dev, end_logits_dev, n_b, T, out_dev, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return DPH_OK;
}

int dph_merge_topk_dev(int device, const float* D_parts, const int64_t* I_parts, int n_parts, int64_t part_stride_bytes,
                       int64_t n, int k, float* D_out, int64_t* I_out, int32_t* src_out, void* stream) {
    if (!D_parts || !I_parts || !D_out || !I_out || n_parts <= 0 || n < 0 || k <= 0) return fail(DPH_E_ARG, "dph_merge_topk_dev: bad arguments");
    HIPCHK(hipSetDevice(device));
    if (part_stride_bytes < 0 || (part_stride_bytes % 8) != 0) return fail(DPH_E_ARG, "dph_merge_topk_dev: stride must be a multiple of 8");
    if (n > 0) dph_launch_merge(D_parts, I_parts, nullptr, nullptr, nullptr, nullptr, n_parts, part_stride_bytes, n, k, D_out,
                                I_out, src_out, nullptr, nullptr, nullptr, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return DPH_OK;
}

int dph_merge_records_dev(int device, const float* D_parts, const int64_t* I_parts, const double* best_parts,
                          const int32_t* pred_parts, const int32_t* status_parts, const double* bound_parts, int n_parts,
                          int64_t part_stride_bytes,
                          int64_t n, int k, float* D_out, int64_t* I_out, double* best_out, int32_t* pred_out,
                          int32_t* status_out, void* stream) {
    if (!D_parts || !I_parts || !best_parts || !pred_parts || !status_parts || !D_out || !I_out || !best_out || !pred_out ||
        !status_out || n_parts <= 0 || n < 0 || k <= 0)
        return fail(DPH_E_ARG, "dph_merge_records_dev: bad arguments");
    if (part_stride_bytes < 0 || (part_stride_bytes % 8) != 0) return fail(DPH_E_ARG, "dph_merge_records_dev: stride must be a multiple of 8");
    HIPCHK(hipSetDevice(device));
    if (n > 0) dph_launch_merge(D_parts, I_parts, best_parts, pred_parts, status_parts, bound_parts, n_parts, part_stride_bytes,
                                n, k, D_out, I_out, nullptr, best_out, pred_out, status_out, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return DPH_OK;
}

int dph_profile_enable(dph_index* h, int on) {
    if (!h) return fail(DPH_E_ARG, "null");
    if (h->pq) { const int rc = dph_pq_profile(h->pq, on); return rc ? fail(rc, dph_pq_error()) : DPH_OK; }
    h->profile = on != 0;
    if (h->profile && h->prof_free.size() < 256) {
        // create the events now: the first hipEventCreate of a process can take ~100 ms, and it must not land in a
        // timed region
        HIPCHK(hipSetDevice(h->device));
        while (h->prof_free.size() < 256) {
            std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
            HIPCHK(hipEventCreate(&ev.first));
            HIPCHK(hipEventCreate(&ev.second));
            h->prof_free.push_back(ev);
        }
    }
    return DPH_OK;
}

static int drain_events(dph_index* h, std::vector<std::pair<hipEvent_t, hipEvent_t>>& evs, double* total_out, int* cnt_out) {
    double total = 0.0;
    int cnt = 0;
    for (auto& ev : evs) {
        HIPCHK(hipEventSynchronize(ev.second));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, ev.first, ev.second));
        total += ms;
        ++cnt;
        h->prof_free.push_back(ev);
    }
    evs.clear();
    if (total_out) *total_out = total;
    if (cnt_out) *cnt_out = cnt;
    return DPH_OK;
}

int dph_profile_read(dph_index* h, double* scan_ms_total, int* scan_launches) {
    if (!h || !scan_ms_total || !scan_launches) return fail(DPH_E_ARG, "null");
    if (h->pq) { const int rc = dph_pq_profile_read(h->pq, scan_ms_total, scan_launches); return rc ? fail(rc, dph_pq_error()) : DPH_OK; }
    HIPCHK(hipSetDevice(h->device));
    int rc = drain_events(h, h->prof_events_ladder, nullptr, nullptr);
    if (rc) return rc;
    return drain_events(h, h->prof_events, scan_ms_total, scan_launches);
}

// every bracketed launch since the last read on its own: which = 0 the full scans (PQ index: the coarse filter scans), 1 the ladder levels'
int dph_profile_read_each(dph_index* h, int which, double* ms_out, int cap, int* n_out) {
    if (!h || !n_out || cap < 0 || (cap > 0 && !ms_out)) return fail(DPH_E_ARG, "dph_profile_read_each: bad arguments");
    if (h->pq) { const int rc = dph_pq_profile_read_each(h->pq, ms_out, cap, n_out); return rc ? fail(rc, dph_pq_error()) : DPH_OK; }
    HIPCHK(hipSetDevice(h->device));
    auto& evs = which ? h->prof_events_ladder : h->prof_events;
    int n = 0;
    for (auto& ev : evs) {
        HIPCHK(hipEventSynchronize(ev.second));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, ev.first, ev.second));
        if (n < cap) ms_out[n] = ms;
        ++n;
        h->prof_free.push_back(ev);
    }
    evs.clear();
    *n_out = n;
    return DPH_OK;
}

int dph_profile_read_all(dph_index* h, double* scan_ms_total, int* scan_launches, double* ladder_ms_total, int* ladder_launches) {
    if (!h || !scan_ms_total || !scan_launches || !ladder_ms_total || !ladder_launches) return fail(DPH_E_ARG, "null");
    HIPCHK(hipSetDevice(h->device));
    int rc = drain_events(h, h->prof_events_ladder, ladder_ms_total, ladder_launches);
    if (rc) return rc;
    return drain_events(h, h->prof_events, scan_ms_total, scan_launches);
}

int dph_debug_scan_buckets(dph_index* h, const float* x, int64_t n, const int32_t* tau_host, int tile_stride,
                           uint64_t* keys_host, uint32_t* counts_host) {
    if (!h || !x || !keys_host || !counts_host || n <= 0 || n > DPH_QROWS * DPH_MAX_QB || tile_stride < 1)
        return fail(DPH_E_ARG, "dph_debug_scan_buckets: bad arguments");
    if (!h->finalized) return fail(DPH_E_STATE, "dph_debug_scan_buckets: call dph_index_finalize first");
    HIPCHK(hipSetDevice(h->device));
    int rc = ensure_scratch(h, n, 0);
    if (rc) return rc;
    hipStream_t st = nullptr;
    HIPCHK(hipMemcpyAsync(h->q_main.x, x, (size_t)n * DPH_DIM * 4, hipMemcpyHostToDevice, st));
    quantize(h, h->q_main, h->q_main.x, n, nullptr, st);
    const int qb = n > DPH_QROWS ? 2 : 1;
    dph_pass p = make_pass(h, h->q_main, h->q_main.x, 0, (int)n, qb);
    if (h->row_ids) p.tilemask = h->onesmask;
    const int* tau = nullptr;
    if (tau_host) {
        HIPCHK(hipMemcpyAsync(h->tau_dev, tau_host, (size_t)n * 4, hipMemcpyHostToDevice, st));
        tau = h->tau_dev;
    }
    const int64_t tiles = (h->n_tiles + tile_stride - 1) / tile_stride;
    dph_launch_scan(p, tile_stride > 1, tiles, tile_stride, tau, 4, st);
    dph_launch_refine(p, st);
    HIPCHK(hipGetLastError());
    std::vector<unsigned> cnt((size_t)2 * DPH_PASS_MAX);
    HIPCHK(hipMemcpyAsync(cnt.data(), h->bucket_counts, cnt.size() * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int64_t q = 0; q < n; ++q) {
        // bit 31 of the count reports lost pairs (the pair pool ran dry) for that row
        const unsigned c = cnt[(size_t)q] < (unsigned)DPH_BUCKET_CAP ? cnt[(size_t)q] : (unsigned)DPH_BUCKET_CAP;
        counts_host[q] = c | (cnt[(size_t)(DPH_PASS_MAX + q)] ? 0x80000000u : 0u);
        HIPCHK(hipMemcpy(keys_host + q * DPH_BUCKET_CAP, h->buckets + q * DPH_BUCKET_CAP, (size_t)c * 8, hipMemcpyDeviceToHost));
    }
    return DPH_OK;
}

int dph_debug_scan_time(dph_index* h, const float* x, int64_t n, int iters, float* ms_out) {
    if (!h || !x || !ms_out || n <= 0 || n > DPH_QROWS * DPH_MAX_QB || iters <= 0 || iters > 64)
        return fail(DPH_E_ARG, "dph_debug_scan_time: bad arguments");
    if (!h->finalized) return fail(DPH_E_STATE, "dph_debug_scan_time: call dph_index_finalize first");
    if (h->pq) return fail(DPH_E_STATE, "dph_debug_scan_time: a PQ index has no filter scan");
    HIPCHK(hipSetDevice(h->device));
    int rc = ensure_scratch(h, n, 0);
    if (rc) return rc;
    hipStream_t st = nullptr;
    HIPCHK(hipMemcpyAsync(h->q_main.x, x, (size_t)n * DPH_DIM * 4, hipMemcpyHostToDevice, st));
    quantize(h, h->q_main, h->q_main.x, n, nullptr, st);
    const int qb = n > DPH_QROWS ? 2 : 1;
    dph_pass p = make_pass(h, h->q_main, h->q_main.x, 0, (int)n, qb);
    if (h->row_ids) p.tilemask = h->onesmask;
    // a bound no high-digit score reaches (|H| <= 127 * 128 * 768 < 2^24 = (2^31 - 1 - lmax) >> 7 for any lmax < 2^24): the
    // launch streams and multiplies everything and emits nothing
    HIPCHK(hipMemsetD32Async((hipDeviceptr_t)h->tau_dev, 0x7fffffff, (size_t)n, st));
    hipEvent_t a = nullptr, b = nullptr;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    for (int i = 0; i < iters; ++i) {
        HIPCHK(hipEventRecord(a, st));
        dph_launch_scan(p, false, h->n_tiles, 1, h->tau_dev, 4, st);
        HIPCHK(hipEventRecord(b, st));
        HIPCHK(hipEventSynchronize(b));
        HIPCHK(hipEventElapsedTime(&ms_out[i], a, b));
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    HIPCHK(hipGetLastError());
    return DPH_OK;
}

int dph_debug_bucket_counts(dph_index* h, int64_t n, uint32_t* raw_out, uint32_t* overflow_out) {
    if (!h || !raw_out || !overflow_out || n <= 0 || n > DPH_PASS_MAX) return fail(DPH_E_ARG, "dph_debug_bucket_counts: bad arguments");
    if (!h->bucket_counts) return fail(DPH_E_STATE, "dph_debug_bucket_counts: no search yet");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(raw_out, h->bucket_counts, (size_t)n * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(overflow_out, h->bucket_counts + DPH_PASS_MAX, (size_t)n * 4, hipMemcpyDeviceToHost));
    return DPH_OK;
}

int dph_debug_pq_coarse(dph_index* h, uint32_t out[2]) {
    if (!h || !out) return fail(DPH_E_ARG, "null");
    if (!h->pq) return fail(DPH_E_STATE, "dph_debug_pq_coarse: not a PQ index");
    const int rc = dph_pq_coarse_debug(h->pq, out);
    return rc ? fail(rc, dph_pq_error()) : DPH_OK;
}

int dph_debug_pq_pass(dph_index* h, int32_t info[8], uint32_t* per_row, int n_rows) {
    if (!h || !info || n_rows < 0 || (n_rows > 0 && !per_row)) return fail(DPH_E_ARG, "dph_debug_pq_pass: bad arguments");
    if (!h->pq) return fail(DPH_E_STATE, "dph_debug_pq_pass: not a PQ index");
    const int rc = dph_pq_debug_pass(h->pq, info, per_row, n_rows);
    return rc ? fail(rc, dph_pq_error()) : DPH_OK;
}

int dph_debug_pq_pool(dph_index* h, uint32_t* lk_host, uint16_t* q_host, int64_t cap, int64_t* count) {
    if (!h || !lk_host || !q_host || !count || cap < 0) return fail(DPH_E_ARG, "null");
    if (!h->pq) return fail(DPH_E_STATE, "dph_debug_pq_pool: not a PQ index");
    long long c = 0;
    const int rc = dph_pq_coarse_debug_pool(h->pq, lk_host, q_host, cap, &c);
    *count = c;
    return rc ? fail(rc, dph_pq_error()) : DPH_OK;
}

int dph_debug_pq_phases(dph_index* h, int which, uint64_t* out, int cap_wgs, int* n_wgs) {
    if (!h || !n_wgs || cap_wgs < 0 || which < 0 || which > 1) return fail(DPH_E_ARG, "dph_debug_pq_phases: bad arguments");
    if (!h->pq) return fail(DPH_E_STATE, "dph_debug_pq_phases: not a PQ index");
    const int n = dph_pq_debug_phases(h->pq, which, (unsigned long long*)out, cap_wgs);
    if (n < 0) return fail(DPH_E_HIP, "dph_debug_pq_phases: allocation or copy failed");
    *n_wgs = n;
    return DPH_OK;
}

int dph_debug_lmax(dph_index* h, int64_t n, int32_t* lmax_host) {
    if (!h || !lmax_host || n <= 0 || n > h->cap_rows) return fail(DPH_E_ARG, "dph_debug_lmax: bad arguments");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpy(lmax_host, h->q_main.lmax, (size_t)n * 4, hipMemcpyDeviceToHost));
    return DPH_OK;
}

int dph_debug_aux(dph_index* h, int64_t row0, int64_t n_rows, int8_t* aux_host, int64_t n_q, int8_t* qaux_host, int32_t* info) {
    if (!h || row0 < 0 || n_rows < 0 || n_q < 0) return fail(DPH_E_ARG, "dph_debug_aux: bad arguments");
    const dph_index* src = h->twin_of ? h->twin_of : h;
    if (!src->finalized) return fail(DPH_E_STATE, "dph_debug_aux: call dph_index_finalize first");
    HIPCHK(hipSetDevice(h->device));
    if (info) { info[0] = src->aux_lay.stride; info[1] = src->norm_unit; info[2] = src->aux_lay.q2max; info[3] = src->aux_lay.n_rep; }
    if (aux_host && n_rows > 0) {
        if (src->aux_lay.stride <= 0 || row0 + n_rows > src->n_tiles * DPH_TILE_ROWS) return fail(DPH_E_ARG, "dph_debug_aux: no aux rows there");
        HIPCHK(hipMemcpy(aux_host, src->aux + row0 * src->aux_lay.stride, (size_t)n_rows * src->aux_lay.stride, hipMemcpyDeviceToHost));
    }
    if (qaux_host && n_q > 0) {
        if (n_q > h->cap_rows) return fail(DPH_E_ARG, "dph_debug_aux: more query rows than the last call quantised");
        HIPCHK(hipMemcpy(qaux_host, h->q_main.qaux, (size_t)n_q * DPH_AUX_SLOTS, hipMemcpyDeviceToHost));
    }
    return DPH_OK;
}
int dph_debug_mu(dph_index* h, int32_t* mu_out) {
    if (!h || !mu_out) return fail(DPH_E_ARG, "dph_debug_mu: null");
    const dph_index* src = h->twin_of ? h->twin_of : h;
    for (int j = 0; j < DPH_DIM; ++j) mu_out[j] = src->mu_host[j];
    return DPH_OK;
}

int64_t dph_debug_fused_tile(int64_t n_tiles, int stride, int64_t v, int64_t* visit_out) {
    unsigned m = 0;
    const int64_t visit = fused_scan_plan(n_tiles, stride, &m);
    if (visit_out) *visit_out = visit;
    if (visit <= 0 || v < 0 || v >= visit) return -1;
    return v + (int64_t)(((uint64_t)v * m) >> 29) + 1;              // dph_scan.hip: tile_of
}

int64_t dph_debug_guided_segment(int64_t u, int64_t n_tiles, int grid, int seg_min, int64_t* len) {
    int64_t l = 0;
    if (u < 0 || n_tiles < 0 || grid <= 0 || seg_min <= 0) { if (len) *len = 0; return -1; }
    const int64_t first = dph_guided_segment(u, n_tiles, grid, seg_min, &l);
    if (len) *len = l;
    return first;
}

int dph_debug_units(dph_index* h, int32_t out[4]) {
    if (!h || !out) return fail(DPH_E_ARG, "dph_debug_units: null");
    if (!h->unit_counts) return fail(DPH_E_STATE, "dph_debug_units: no unit scan on this shard");
    HIPCHK(hipSetDevice(h->device));
    int c[4 + DPH_UNIT_LAUNCHES];
    HIPCHK(hipMemcpy(c, h->unit_counts, sizeof(c), hipMemcpyDeviceToHost));
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[4 + DPH_UNIT_LAUNCHES - 1];
    return DPH_OK;
}

int dph_index_shard_stats(dph_index* h, double* rmax, double* rmax_all, int* n_outliers) {
    if (!h || !rmax || !rmax_all || !n_outliers) return fail(DPH_E_ARG, "null");
    *rmax = h->rmax; *rmax_all = h->rmax_all; *n_outliers = h->n_out;
    return DPH_OK;
}

}  // extern "C"
