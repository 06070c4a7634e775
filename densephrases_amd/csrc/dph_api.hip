// dph_api.hip -- the C ABI of libdph (include/dph.h): handle management, host<->HBM plumbing, and the
// search driver (quantise -> int8 scan -> select/certify -> wider scan -> fp64 fallback).
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <utility>
#include <vector>

#include "dph_internal.h"

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
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
    std::vector<int32_t> h_inv;
    int nlist = 0;
    float* centroids = nullptr;          // [nlist, 768] fp32
    int32_t* tile_list = nullptr;        // [n_tiles]
    unsigned* listmask = nullptr;        // [nlist][4]   bit j of word w: query row 32w+j of the pass probes the list
    unsigned* tilemask = nullptr;        // [n_tiles][4] the same per tile (what the scan reads)
    unsigned* onesmask = nullptr;        // [n_tiles][4] all ones: exact (flat) search over a list-major shard
    float offset = -2.f, scale = 20.f;
    float lut_host[256];
    float* lut_dev = nullptr;
    double delta_max = 0.0;              // max_n | x32(n) - (n/scale + offset) |
    double rmax = 0.0;                   // max_row || n - c ||_2
    bool finalized = false;
    // idx2id + f2o CSR
    int32_t *row2doc = nullptr, *row2word = nullptr;
    std::vector<int32_t> h_row2doc, h_row2word;
    int32_t* doc_ids = nullptr; int64_t* f2o_off = nullptr; int32_t* f2o = nullptr; int64_t n_docs = 0;
    // search scratch (grown on demand)
    int grid = 256;
    int64_t cap_rows = 0;                // query rows the scratch is sized for
    float* x_dev = nullptr; int8_t* qfrag = nullptr; dph_qinfo* qinfo = nullptr;
    uint64_t* lists = nullptr; float* D_dev = nullptr; int64_t* I_dev = nullptr; int32_t* status_dev = nullptr;
    int cap_k = 0;
    int32_t* fail_dev = nullptr; void* exact_scratch = nullptr; size_t exact_bytes = 0;
    unsigned long long* norm_dev = nullptr;
    int* tau_dev = nullptr;              // [2][128] per-row pre-pass bounds of the current pass (coarse, fine)
    int* lmax_dev = nullptr;             // [padded rows] upper bound of the low-digit term (lazy scan)
    int64_t lmax_cap = 0;
    dph_search_stats stats{};
    // measurement hook: event pairs around scan launches
    bool profile = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_free;
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
    // zero the padding rows of the last tile (they are excluded from the lists by row index as well)
    if (h->n_tiles > 0) {
        const size_t used = (size_t)n_rows * DPH_DIM;
        if (bytes > used) (void)hipMemset(h->db + used, 0, bytes - used);
    }
    build_lut(h);
    if (hipMalloc((void**)&h->lut_dev, 256 * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&h->norm_dev, sizeof(unsigned long long)) != hipSuccess) {
        dph_index_destroy(h);
        return fail(DPH_E_NOMEM, "hipMalloc lut");
    }
    (void)hipMemcpy(h->lut_dev, h->lut_host, sizeof(h->lut_host), hipMemcpyHostToDevice);
    *out = h;
    return DPH_OK;
}

int dph_index_destroy(dph_index* h) {
    if (!h) return DPH_OK;
    (void)hipSetDevice(h->device);
    void* ptrs[] = {h->db, h->lut_dev, h->row2doc, h->row2word, h->doc_ids, h->f2o_off, h->f2o, h->x_dev, h->qfrag,
                    h->qinfo, h->lists, h->D_dev, h->I_dev, h->status_dev, h->fail_dev, h->exact_scratch, h->norm_dev,
                    h->tau_dev, h->lmax_dev, h->row_ids, h->inv_row, h->centroids, h->tile_list, h->listmask,
                    h->tilemask, h->onesmask};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    delete h;
    return DPH_OK;
}

int dph_index_set_codec(dph_index* h, float offset, float scale) {
    if (!h || !(scale > 0.f)) return fail(DPH_E_ARG, "dph_index_set_codec: scale must be > 0");
    HIPCHK(hipSetDevice(h->device));
    h->offset = offset; h->scale = scale;
    build_lut(h);
    HIPCHK(hipMemcpy(h->lut_dev, h->lut_host, sizeof(h->lut_host), hipMemcpyHostToDevice));
    return DPH_OK;
}

int dph_index_upload_rows(dph_index* h, int64_t row0, int64_t n, const int8_t* host_rows) {
    if (!h || !host_rows || row0 < 0 || n < 0 || row0 + n > h->n_rows) return fail(DPH_E_ARG, "dph_index_upload_rows: range");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpy(h->db + row0 * DPH_DIM, host_rows, (size_t)n * DPH_DIM, hipMemcpyHostToDevice));
    h->finalized = false;
    return DPH_OK;
}

int dph_index_fill_synthetic(dph_index* h, uint64_t seed, void* stream) {
    if (!h) return fail(DPH_E_ARG, "null handle");
    HIPCHK(hipSetDevice(h->device));
    if (h->n_rows > 0) dph_launch_fill(h->db, h->n_rows, h->id_base, seed, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    h->finalized = false;
    return DPH_OK;
}

int dph_index_set_idx2id(dph_index* h, const int32_t* doc, const int32_t* word) {
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

int dph_index_finalize(dph_index* h, void* stream) {
    if (!h) return fail(DPH_E_ARG, "null handle");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    HIPCHK(hipMemsetAsync(h->norm_dev, 0, sizeof(unsigned long long), st));
    if (h->n_rows > 0) dph_launch_rownorm(h->db, h->n_rows, h->row_ids, h->norm_dev, st);
    unsigned long long m = 0;
    HIPCHK(hipMemcpyAsync(&m, h->norm_dev, sizeof(m), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    h->rmax = sqrt((double)m);
    h->finalized = true;
    return DPH_OK;
}

int64_t dph_index_ntotal(const dph_index* h) { return h ? h->n_ids : 0; }

int dph_index_set_row_ids(dph_index* h, const int64_t* row_ids, int64_t n_ids) {
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
    HIPCHK(hipMalloc((void**)&h->onesmask, (size_t)(h->n_tiles > 0 ? h->n_tiles : 1) * 16));
    HIPCHK(hipMemset(h->onesmask, 0xFF, (size_t)(h->n_tiles > 0 ? h->n_tiles : 1) * 16));
    h->h_inv.swap(inv);
    h->n_ids = n_ids;
    h->finalized = false;
    return DPH_OK;
}

int dph_index_set_ivf(dph_index* h, int nlist, const float* centroids, const int32_t* tile_list) {
    if (!h || nlist <= 0 || nlist > 65536 || !centroids || !tile_list) return fail(DPH_E_ARG, "dph_index_set_ivf: bad arguments");
    if (!h->row_ids) return fail(DPH_E_STATE, "dph_index_set_ivf: call dph_index_set_row_ids first (list-major shard)");
    for (int64_t t = 0; t < h->n_tiles; ++t)
        if (tile_list[t] < 0 || tile_list[t] >= nlist) return fail(DPH_E_ARG, "dph_index_set_ivf: tile_list out of range");
    HIPCHK(hipSetDevice(h->device));
    void* old[] = {h->centroids, h->tile_list, h->listmask, h->tilemask};
    for (void* p : old) if (p) (void)hipFree(p);
    h->centroids = nullptr; h->tile_list = nullptr; h->listmask = nullptr; h->tilemask = nullptr;
    HIPCHK(hipMalloc((void**)&h->centroids, (size_t)nlist * DPH_DIM * 4));
    HIPCHK(hipMemcpy(h->centroids, centroids, (size_t)nlist * DPH_DIM * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&h->tile_list, (size_t)(h->n_tiles > 0 ? h->n_tiles : 1) * 4));
    HIPCHK(hipMemcpy(h->tile_list, tile_list, (size_t)h->n_tiles * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&h->listmask, (size_t)nlist * 16));
    HIPCHK(hipMalloc((void**)&h->tilemask, (size_t)(h->n_tiles > 0 ? h->n_tiles : 1) * 16));
    h->nlist = nlist;
    return DPH_OK;
}
int dph_index_dim(const dph_index* h) { (void)h; return DPH_DIM; }
int dph_index_device(const dph_index* h) { return h ? h->device : -1; }
void* dph_index_rows_dev(dph_index* h) { if (h) h->finalized = false; return h ? h->db : nullptr; }

// ------------------------------------------------------------------------------------------ search driver
static int ensure_scratch(dph_index* h, int64_t n, int k) {
    const int64_t padded = (n + DPH_QROWS - 1) / DPH_QROWS * DPH_QROWS;
    if (padded > h->cap_rows) {
        void* old[] = {h->x_dev, h->qfrag, h->qinfo, h->status_dev, h->fail_dev};
        for (void* p : old) if (p) (void)hipFree(p);
        h->x_dev = nullptr; h->qfrag = nullptr; h->qinfo = nullptr; h->status_dev = nullptr; h->fail_dev = nullptr;
        HIPCHK(hipMalloc((void**)&h->x_dev, (size_t)padded * DPH_DIM * 4));
        HIPCHK(hipMalloc((void**)&h->qfrag, (size_t)(padded / DPH_QROWS) * DPH_QFRAG_BYTES));
        HIPCHK(hipMalloc((void**)&h->qinfo, (size_t)padded * sizeof(dph_qinfo)));
        HIPCHK(hipMalloc((void**)&h->status_dev, (size_t)padded * 4));
        HIPCHK(hipMalloc((void**)&h->fail_dev, (size_t)padded * 4));
        h->cap_rows = padded;
        h->cap_k = 0;
    }
    if (k > h->cap_k) {
        if (h->D_dev) (void)hipFree(h->D_dev);
        if (h->I_dev) (void)hipFree(h->I_dev);
        h->D_dev = nullptr; h->I_dev = nullptr;
        HIPCHK(hipMalloc((void**)&h->D_dev, (size_t)h->cap_rows * k * 4));
        HIPCHK(hipMalloc((void**)&h->I_dev, (size_t)h->cap_rows * k * 8));
        h->cap_k = k;
    }
    if (!h->lists) HIPCHK(hipMalloc((void**)&h->lists, (size_t)h->grid * DPH_SCAN_THREADS * 32 * 8));
    if (!h->tau_dev) HIPCHK(hipMalloc((void**)&h->tau_dev, 2 * DPH_QROWS * sizeof(int)));
    if (padded > h->lmax_cap) {
        if (h->lmax_dev) (void)hipFree(h->lmax_dev);
        h->lmax_dev = nullptr;
        HIPCHK(hipMalloc((void**)&h->lmax_dev, (size_t)padded * sizeof(int)));
        h->lmax_cap = padded;
    }
    return DPH_OK;
}

// one attempt with candidate lists of kp entries per lane: quantise, then per pass of 128 rows scan + select
// nprobe > 0: IVF search (coarse quantizer -> probe masks); nprobe = 0: exact search over every row.
// Sharded two-phase form (dph_search_sample_dev / dph_search_bounded_dev): `top_out` != NULL stops after the ladder and
// returns the DPH_SAMPLE_KEEP best sampled scores per row; `tau_ext` != NULL skips the ladder and scans under the
// caller's bounds, `bound_out` then receives the upper bound of the score of any row this shard did not return.
struct attempt_opts {
    int32_t* top_out = nullptr;
    const int32_t* tau_ext = nullptr;
    double* bound_out = nullptr;
};

static int run_attempt(dph_index* h, int kp, const float* x_dev, int64_t n, int k, int nprobe, float* D_dev,
                       int64_t* I_dev, int32_t* status_dev, int8_t* qfrag, dph_qinfo* qinfo, hipStream_t st,
                       const attempt_opts& opt = attempt_opts()) {
    if (h->row_ids) kp = 16;             // list-major shards run the masked kernels, which exist for 16-entry lists
    dph_launch_quantize(x_dev, n, qfrag, qinfo, h->rmax, h->lmax_dev, st);
    // threshold pre-pass when the shard is big enough for the tile sample to give every workgroup work.  The sample is
    // taken in levels of decreasing stride (default 256s, 16s, s with s = 32; 64s, 8s, s with s = 16 on shards under
    // 100 M rows): the coarsest runs on the eager kernel from a cold start, every finer one on the
    // lazy kernel under the previous level's bound (the eager kernel spends most of a cold start in its lists).
    // DPH_PREPASS_STRIDE=s overrides the finest stride (0 = no pre-pass), DPH_PREPASS_LEVELS="a,b,c" the whole
    // ladder -- both read per call: experiments and tests flip them in-process.
    int levels[4], n_levels = 0;
    {
        const char* lv = getenv("DPH_PREPASS_LEVELS");
        const char* se = getenv("DPH_PREPASS_STRIDE");
        if (lv && *lv) {
            for (const char* c = lv; *c && n_levels < 4;) {
                const int v = atoi(c);
                if (v > 0) levels[n_levels++] = v;
                while (*c && *c != ',') ++c;
                if (*c == ',') ++c;
            }
        } else {
            // measured (tools/sweep_prepass.py, two boxes): 170 M rows 8192,512,32 = 24.96-25.03 ms per step vs
            // 2048,256,32 = 25.17-25.22 and 512,32 = 25.17-25.42; 21 M rows 1024,128,16 = 3.72 vs 512,32 = 3.82
            const bool big = h->n_rows >= 100000000ll;
            const int fine = se ? atoi(se) : (big ? DPH_SAMPLE_STRIDE : DPH_SAMPLE_STRIDE / 2);
            const int ratio = big ? 16 : 8;
            if (fine > 0) { levels[0] = ratio * ratio * fine; levels[1] = ratio * fine; levels[2] = fine; n_levels = 3; }
        }
        // the retry attempt (wider lists) and small shards use the finest level only, from a cold start
        int kept = 0;
        for (int i = 0; i < n_levels; ++i) {
            const int64_t tiles = (h->n_tiles + levels[i] - 1) / levels[i];
            const bool last = i == n_levels - 1;
            if (tiles >= (int64_t)h->grid && (kp == 16 || last)) levels[kept++] = levels[i];
        }
        n_levels = opt.tau_ext ? 0 : kept;
    }
    for (int64_t q0 = 0; q0 < n; q0 += DPH_QROWS) {
        const int nq = (int)((n - q0) < DPH_QROWS ? (n - q0) : DPH_QROWS);
        const int8_t* qf = qfrag + (q0 / DPH_QROWS) * (int64_t)DPH_QFRAG_BYTES;
        const int* tau = nullptr;
        const unsigned* mask = nullptr;      // per-tile probe masks of this pass (list-major shards only)
        if (h->row_ids) {
            if (nprobe > 0) {
                dph_launch_coarse(x_dev, (int)q0, nq, h->centroids, h->nlist, nprobe, h->listmask, h->tile_list, h->n_tiles,
                                  h->tilemask, st);
                mask = h->tilemask;
            } else {
                mask = h->onesmask;
            }
        }
        for (int i = 0; i < n_levels; ++i) {
            const int64_t tiles = (h->n_tiles + levels[i] - 1) / levels[i];
            int* out = h->tau_dev + (i & 1) * DPH_QROWS;
            dph_launch_scan(kp, true, h->db, h->n_rows, tiles, levels[i], qf, tau, tau ? h->lmax_dev + q0 : nullptr, mask,
                            h->row_ids, h->lists, h->grid, st);
            int* top = (opt.top_out && i == n_levels - 1) ? opt.top_out + q0 * DPH_SAMPLE_KEEP : nullptr;
            if (dph_launch_threshold(kp, h->lists, h->grid, tau, out, nq, top, st)) return fail(DPH_E_STATE, "threshold image too small for this grid");
            tau = out;
        }
        if (opt.top_out) {
            // a shard too small for any ladder level shares nothing (INT_MIN everywhere)
            if (n_levels == 0)
                HIPCHK(hipMemsetD32Async((hipDeviceptr_t)(opt.top_out + q0 * DPH_SAMPLE_KEEP), (int)0x80000000,
                                         (size_t)nq * DPH_SAMPLE_KEEP, st));
            continue;
        }
        if (opt.tau_ext) {
            // stage the caller's bounds of this pass into the 128-entry buffer the kernels index
            HIPCHK(hipMemsetD32Async((hipDeviceptr_t)h->tau_dev, (int)0x80000000, DPH_QROWS, st));
            HIPCHK(hipMemcpyAsync(h->tau_dev, opt.tau_ext + q0, (size_t)nq * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
            tau = h->tau_dev;
        }
        std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
        if (h->profile) {
            if (!h->prof_free.empty()) { ev = h->prof_free.back(); h->prof_free.pop_back(); }
            else { (void)hipEventCreate(&ev.first); (void)hipEventCreate(&ev.second); }
            (void)hipEventRecord(ev.first, st);
        }
        dph_launch_scan(kp, false, h->db, h->n_rows, h->n_tiles, 1, qf, tau, h->lmax_dev + q0, mask, h->row_ids, h->lists, h->grid, st);
        if (h->profile) { (void)hipEventRecord(ev.second, st); h->prof_events.push_back(ev); }
        dph_launch_select(kp, h->grid, h->lists, h->db, h->n_rows, h->id_base, x_dev, qinfo, h->lut_dev, (int)q0, nq, k,
                          h->rmax, h->delta_max, h->offset, h->scale, tau, h->row_ids, D_dev, I_dev, status_dev, opt.bound_out, st);
        h->stats.scan_launches++;
    }
    HIPCHK(hipGetLastError());
    return DPH_OK;
}

static int check_search_args(dph_index* h, const void* x, int64_t n, int k, int nprobe, const void* D, const void* I,
                             const char* who) {
    if (!h || !x || !D || !I || n < 0 || k <= 0 || nprobe < 0) return fail(DPH_E_ARG, std::string(who) + ": bad arguments");
    if (k > 1024) return fail(DPH_E_ARG, std::string(who) + ": k <= 1024");
    if (!h->finalized) return fail(DPH_E_STATE, std::string(who) + ": call dph_index_finalize first");
    if (nprobe > 0 && (!h->centroids || !h->row_ids)) return fail(DPH_E_STATE, std::string(who) + ": IVF data not set (dph_index_set_ivf)");
    return DPH_OK;
}

static int search_dev_impl(dph_index* h, const float* x_dev, int64_t n, int k, int nprobe, float* D_dev, int64_t* I_dev,
                           int32_t* status_dev, void* stream, const char* who) {
    int rc = check_search_args(h, x_dev, n, k, nprobe, D_dev, I_dev, who);
    if (rc) return rc;
    if (!status_dev) return fail(DPH_E_ARG, std::string(who) + ": status buffer is NULL");
    if (n == 0) return DPH_OK;
    HIPCHK(hipSetDevice(h->device));
    rc = ensure_scratch(h, n, 1);
    if (rc) return rc;
    h->stats = dph_search_stats{};
    h->stats.rows = (int32_t)n;
    return run_attempt(h, 16, x_dev, n, k, nprobe, D_dev, I_dev, status_dev, h->qfrag, h->qinfo, (hipStream_t)stream);
}

static int search_host_impl(dph_index* h, const float* x, int64_t n, int k, int nprobe, float* D, int64_t* I, const char* who) {
    int rc = check_search_args(h, x, n, k, nprobe, D, I, who);
    if (rc) return rc;
    if (n == 0) return DPH_OK;
    HIPCHK(hipSetDevice(h->device));
    rc = ensure_scratch(h, n, k);
    if (rc) return rc;
    hipStream_t st = nullptr;
    h->stats = dph_search_stats{};
    h->stats.rows = (int32_t)n;
    HIPCHK(hipMemcpyAsync(h->x_dev, x, (size_t)n * DPH_DIM * 4, hipMemcpyHostToDevice, st));
    rc = run_attempt(h, 16, h->x_dev, n, k, nprobe, h->D_dev, h->I_dev, h->status_dev, h->qfrag, h->qinfo, st);
    if (rc) return rc;
    std::vector<int32_t> status((size_t)n);
    HIPCHK(hipMemcpyAsync(status.data(), h->status_dev, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    std::vector<int32_t> failing;
    for (int64_t r = 0; r < n; ++r) if (status[r] != 0) failing.push_back((int32_t)r);
    h->stats.certified_fast = (int32_t)(n - (int64_t)failing.size());

    if (!failing.empty()) {
        // ---- second attempt for the failing rows only: wider per-lane lists (32 kept; list-major shards: 16 again)
        struct DevBuf {                       // frees on every exit path
            void* p = nullptr;
            ~DevBuf() { if (p) (void)hipFree(p); }
            int alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1) == hipSuccess ? 0 : 1; }
        };
        const int64_t nf = (int64_t)failing.size();
        std::vector<float> xf((size_t)nf * DPH_DIM);
        for (int64_t i = 0; i < nf; ++i) memcpy(&xf[(size_t)i * DPH_DIM], x + (int64_t)failing[i] * DPH_DIM, DPH_DIM * 4);
        const int64_t padded = (nf + DPH_QROWS - 1) / DPH_QROWS * DPH_QROWS;
        DevBuf xf_dev, qf2, qi2, D2, I2, st2, fr;
        if (xf_dev.alloc((size_t)padded * DPH_DIM * 4) || qf2.alloc((size_t)(padded / DPH_QROWS) * DPH_QFRAG_BYTES) ||
            qi2.alloc((size_t)padded * sizeof(dph_qinfo)) || D2.alloc((size_t)padded * k * 4) ||
            I2.alloc((size_t)padded * k * 8) || st2.alloc((size_t)padded * 4) || fr.alloc((size_t)padded * 4))
            return fail(DPH_E_NOMEM, std::string(who) + ": hipMalloc for the retry buffers failed");
        HIPCHK(hipMemcpyAsync(xf_dev.p, xf.data(), (size_t)nf * DPH_DIM * 4, hipMemcpyHostToDevice, st));
        rc = run_attempt(h, 32, (const float*)xf_dev.p, nf, k, nprobe, (float*)D2.p, (int64_t*)I2.p, (int32_t*)st2.p,
                         (int8_t*)qf2.p, (dph_qinfo*)qi2.p, st);
        if (rc) return rc;
        std::vector<int32_t> status2((size_t)nf);
        HIPCHK(hipMemcpyAsync(status2.data(), st2.p, (size_t)nf * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        // ---- third attempt: fp64 full scan with threshold collect, pass by pass (the probe masks are per pass)
        int64_t n_still = 0;
        for (int64_t i = 0; i < nf; ++i) n_still += status2[i] != 0;
        h->stats.certified_wide = (int32_t)(nf - n_still);
        for (int64_t p0 = 0; p0 < nf && n_still > 0; p0 += DPH_QROWS) {
            std::vector<int32_t> still;
            for (int64_t i = p0; i < nf && i < p0 + DPH_QROWS; ++i) if (status2[i] != 0) still.push_back((int32_t)i);
            if (still.empty()) continue;
            const size_t want = (size_t)256 + still.size() * ((size_t)1 << 20) * 16;   // 1M hits per row
            if (h->exact_bytes < want) {
                if (h->exact_scratch) (void)hipFree(h->exact_scratch);
                h->exact_scratch = nullptr; h->exact_bytes = 0;
                if (hipMalloc(&h->exact_scratch, want) != hipSuccess) return fail(DPH_E_NOMEM, "hipMalloc exact-scan scratch");
                h->exact_bytes = want;
            }
            const unsigned* mask = nullptr;
            if (h->row_ids && nprobe > 0) {
                const int nq = (int)((nf - p0) < DPH_QROWS ? (nf - p0) : DPH_QROWS);
                dph_launch_coarse((const float*)xf_dev.p, (int)p0, nq, h->centroids, h->nlist, nprobe, h->listmask,
                                  h->tile_list, h->n_tiles, h->tilemask, st);
                mask = h->tilemask;
            }
            HIPCHK(hipMemcpyAsync(fr.p, still.data(), still.size() * 4, hipMemcpyHostToDevice, st));
            dph_launch_exact(h->db, h->n_rows, h->id_base, (const float*)xf_dev.p, h->lut_dev, (const int32_t*)fr.p,
                             (int)still.size(), k, h->row_ids, mask, (float*)D2.p, (int64_t*)I2.p, (int32_t*)st2.p,
                             h->exact_scratch, h->exact_bytes, st);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(st));
        }
        if (n_still > 0) {
            HIPCHK(hipMemcpy(status2.data(), st2.p, (size_t)nf * 4, hipMemcpyDeviceToHost));
            for (int64_t i = 0; i < nf; ++i) if (status2[i] != 0) h->stats.uncertified++;
            h->stats.exact_fallback = (int32_t)n_still - h->stats.uncertified;
        }
        std::vector<float> Dh((size_t)nf * k);
        std::vector<int64_t> Ih((size_t)nf * k);
        HIPCHK(hipMemcpy(Dh.data(), D2.p, (size_t)nf * k * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(Ih.data(), I2.p, (size_t)nf * k * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(D, h->D_dev, (size_t)n * k * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(I, h->I_dev, (size_t)n * k * 8, hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < nf; ++i) {
            memcpy(D + (int64_t)failing[i] * k, &Dh[(size_t)i * k], (size_t)k * 4);
            memcpy(I + (int64_t)failing[i] * k, &Ih[(size_t)i * k], (size_t)k * 8);
        }
        if (h->stats.uncertified > 0) return fail(DPH_E_UNCERTIFIED, std::string(who) + ": boundary ties exceed the exact-scan buffer");
        return DPH_OK;
    }
    HIPCHK(hipMemcpy(D, h->D_dev, (size_t)n * k * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(I, h->I_dev, (size_t)n * k * 8, hipMemcpyDeviceToHost));
    return DPH_OK;
}

int dph_search_dev(dph_index* h, const float* x_dev, int64_t n, int k, float* D_dev, int64_t* I_dev,
                   int32_t* status_dev, void* stream) {
    return search_dev_impl(h, x_dev, n, k, 0, D_dev, I_dev, status_dev, stream, "dph_search_dev");
}
int dph_search(dph_index* h, const float* x, int64_t n, int k, float* D, int64_t* I) {
    return search_host_impl(h, x, n, k, 0, D, I, "dph_search");
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
    if (!h || !x_dev || !top_dev || n < 0) return fail(DPH_E_ARG, "dph_search_sample_dev: bad arguments");
    if (!h->finalized) return fail(DPH_E_STATE, "dph_search_sample_dev: call dph_index_finalize first");
    if (h->row_ids) return fail(DPH_E_STATE, "dph_search_sample_dev: flat shards only");
    if (n == 0) return DPH_OK;
    HIPCHK(hipSetDevice(h->device));
    int rc = ensure_scratch(h, n, 1);
    if (rc) return rc;
    attempt_opts opt;
    opt.top_out = top_dev;
    return run_attempt(h, 16, x_dev, n, 1, 0, nullptr, nullptr, nullptr, h->qfrag, h->qinfo, (hipStream_t)stream, opt);
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
    int rc = check_search_args(h, x_dev, n, k, 0, D_dev, I_dev, "dph_search_bounded_dev");
    if (rc) return rc;
    if (!status_dev || !tau_dev || !bound_dev) return fail(DPH_E_ARG, "dph_search_bounded_dev: null buffer");
    if (h->row_ids) return fail(DPH_E_STATE, "dph_search_bounded_dev: flat shards only");
    if (n == 0) return DPH_OK;
    HIPCHK(hipSetDevice(h->device));
    rc = ensure_scratch(h, n, 1);
    if (rc) return rc;
    h->stats = dph_search_stats{};
    h->stats.rows = (int32_t)n;
    attempt_opts opt;
    opt.tau_ext = tau_dev;
    opt.bound_out = bound_dev;
    return run_attempt(h, 16, x_dev, n, k, 0, D_dev, I_dev, status_dev, h->qfrag, h->qinfo, (hipStream_t)stream, opt);
}

int dph_search_get_stats(const dph_index* h, dph_search_stats* out) {
    if (!h || !out) return fail(DPH_E_ARG, "null");
    *out = h->stats;
    return DPH_OK;
}

int dph_reconstruct(dph_index* h, int64_t id, float* out768) {
    if (!h || !out768) return fail(DPH_E_ARG, "null");
    const int64_t local = id - h->id_base;
    if (local < 0 || local >= h->n_ids) return fail(DPH_E_NOTFOUND, "dph_reconstruct: id not in this shard");
    HIPCHK(hipSetDevice(h->device));
    const int64_t srow = h->h_inv.empty() ? local : (int64_t)h->h_inv[(size_t)local];
    int8_t row[DPH_DIM];
    HIPCHK(hipMemcpy(row, h->db + srow * DPH_DIM, DPH_DIM, hipMemcpyDeviceToHost));
    for (int j = 0; j < DPH_DIM; ++j) out768[j] = h->lut_host[(int)row[j] + 128];
    return DPH_OK;
}

int dph_id2docword(dph_index* h, const int64_t* I, int64_t n, int32_t* doc, int32_t* word) {
    if (!h || !I || !doc || !word || n < 0) return fail(DPH_E_ARG, "null");
    if (h->h_row2doc.empty() && h->n_ids > 0) return fail(DPH_E_STATE, "dph_id2docword: idx2id not set");
    for (int64_t i = 0; i < n; ++i) {
        if (h->n_ids == 0) { doc[i] = -1; word[i] = -1; continue; }      // empty shard: nothing to clip to
        int64_t local = I[i] - h->id_base;
        if (local < 0) local = 0;                         // np.clip (index.py:133)
        if (local >= h->n_ids) local = h->n_ids - 1;
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
    dph_launch_window(direction, h->db, h->n_rows, h->id_base, h->lut_dev, qhalf_dev, n_q * k, k, L, ids_dev, doc_dev,
                      word_dev, first_dev, h->row2doc, h->row2word, h->doc_ids, h->n_docs, h->f2o_off, h->f2o, h->inv_row, h->n_ids,
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
    h->profile = on != 0;
    return DPH_OK;
}

int dph_profile_read(dph_index* h, double* scan_ms_total, int* scan_launches) {
    if (!h || !scan_ms_total || !scan_launches) return fail(DPH_E_ARG, "null");
    HIPCHK(hipSetDevice(h->device));
    double total = 0.0;
    int cnt = 0;
    for (auto& ev : h->prof_events) {
        HIPCHK(hipEventSynchronize(ev.second));
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, ev.first, ev.second));
        total += ms;
        ++cnt;
        h->prof_free.push_back(ev);
    }
    h->prof_events.clear();
    *scan_ms_total = total;
    *scan_launches = cnt;
    return DPH_OK;
}

int64_t dph_debug_scan_lists_size(const dph_index* h, int kp) {
    return h ? (int64_t)h->grid * DPH_SCAN_THREADS * kp : 0;
}

int dph_debug_scan_lists(dph_index* h, const float* x, int64_t n, int kp, uint64_t* lists_host, int* grid_out) {
    if (!h || !x || !lists_host || n <= 0 || (kp != 16 && kp != 32)) return fail(DPH_E_ARG, "dph_debug_scan_lists: bad arguments");
    HIPCHK(hipSetDevice(h->device));
    if (n > DPH_QROWS) n = DPH_QROWS;
    int rc = ensure_scratch(h, n, 1);
    if (rc) return rc;
    hipStream_t st = nullptr;
    HIPCHK(hipMemcpyAsync(h->x_dev, x, (size_t)n * DPH_DIM * 4, hipMemcpyHostToDevice, st));
    dph_launch_quantize(h->x_dev, n, h->qfrag, h->qinfo, h->rmax, h->lmax_dev, st);
    dph_launch_scan(kp, false, h->db, h->n_rows, h->n_tiles, 1, h->qfrag, nullptr, nullptr, nullptr, nullptr, h->lists, h->grid, st);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(lists_host, h->lists, (size_t)h->grid * DPH_SCAN_THREADS * kp * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (grid_out) *grid_out = h->grid;
    return DPH_OK;
}

}  // extern "C"
