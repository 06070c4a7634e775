/* dph.h -- C ABI of libdph: the MI355X-native replacement for the FAISS `Index` object protocol and the
 * start/end window re-scoring that DensePhrases' `MIPS` class drives (reference: densephrases/index.py).
 *
 * The reference has no FFI of its own: its seam is the python `MIPS` class and, one level down, the SWIG
 * surface of faiss-gpu==1.6.5 (requirements.txt:2).  Each entry point below names the reference interface it
 * replaces (file:line under /root/reference).  Plain pointers and sizes only -- no torch / numpy types.
 *
 * Conventions
 *   - every function returns 0 on success, a negative DPH_E_* code on failure; dph_last_error() gives the
 *     message of the last failure on the calling thread.  No exception crosses this boundary.
 *   - "host" pointers are ordinary process memory, "dev" pointers are HIP device pointers on the index's
 *     device.  `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *   - the caller owns every in/out buffer; the library owns the device-resident shard and its scratch.
 *   - a handle may be used from one thread at a time (the reference calls MIPS from a single thread:
 *     run_demo.py:147-149).
 *   - vector dimension is fixed at 768 (SpanBERT-base; reference index.py:196, train_query.py:222-223).
 *
 * This header is the CONTRACT a binder needs: the FAISS Index protocol (search / reconstruct / ntotal / d), MIPS.get_idxs, the
 * window re-score, the encoder's start/end scoring, the loaders, the IVF / PQ index types and the sharded protocol.  Tuning keys,
 * profiling hooks, the two-batches-in-flight plumbing and every dph_debug_* test hook live in dph_debug.h (same library, same
 * version number) -- nothing there is needed to serve a search.
 */
#ifndef DPH_H
#define DPH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DPH_DIM 768
#define DPH_ABI_VERSION 7

/* error codes */
#define DPH_OK 0
#define DPH_E_ARG (-1)        /* bad argument                                   */
#define DPH_E_HIP (-2)        /* a HIP runtime call failed                      */
#define DPH_E_NOMEM (-3)      /* device or host allocation failed               */
#define DPH_E_STATE (-4)      /* index not finalized / metadata missing          */
#define DPH_E_NOTFOUND (-5)   /* id not in the index (faiss reconstruct throws) */
#define DPH_E_UNCERTIFIED (-6)/* exactness certificate could not be established */

typedef struct dph_index dph_index;   /* one shard of the phrase dump, resident in one GPU's HBM */

/* per-call statistics of the last dph_search* on a handle (observability; no reference counterpart) */
typedef struct dph_search_stats {
    int32_t rows;              /* query rows searched                                        */
    int32_t certified_fast;    /* rows certified exact by the first int8 scan                */
    int32_t certified_wide;    /* rows that needed the on-device retry scan (own bound, wider re-score) */
    int32_t exact_fallback;    /* rows that needed the fp64 full scan                        */
    int32_t uncertified;       /* rows whose result could not be certified (boundary ties)   */
    int32_t scan_launches;     /* number of scan kernel launches                             */
    int32_t fused_stride;      /* S > 0: the finest sampled level (every S-th tile) was fused into the full scan of the last
                                  pass, which then visited only the other tiles (tuning key "ladder_fuse"); 0 = not fused   */
    int32_t certified_reselect;/* of certified_wide: rows settled by a wider re-score of the bucket they already had (near-ties at
                                  the k-th place), i.e. without a second scan of the shard                                  */
    int32_t nonfinite;         /* query rows with a NaN / Inf element: answered with ids -1, scores -FLT_MAX (see dph_search) */
} dph_search_stats;

/* per-row status of the device-pointer searches */
#define DPH_ROW_OK 0            /* the row's result is certified exact                                                       */
#define DPH_ROW_UNCERTIFIED 1   /* not certified (see dph_search_dev)                                                        */
#define DPH_ROW_DEFERRED 2      /* dph_search_bounded_dev only: decided by the merge (dph_merge_records_dev)                  */
#define DPH_ROW_NONFINITE 3     /* the query row holds a NaN / Inf element: ids -1, scores -FLT_MAX, nothing else affected   */

int         dph_abi_version(void);
const char* dph_last_error(void);
int         dph_device_count(void);

/* ---- lifecycle ---------------------------------------------------------------------------------------
 * replaces faiss.read_index + MIPS.__init__ state (index.py:24-76): the shard is the raw int8 phrase dump
 * (embed_utils.py:141-149,237-241: x = n/scale + offset, scale 20, offset -2) in idx2id row order
 * (build_phrase_index.py:192-276).  `id_base` is the global id of local row 0 (range-sharded multi-GPU). */
int dph_index_create(int device, int64_t n_rows, int64_t id_base, dph_index** out);
int dph_index_destroy(dph_index* h);
int dph_index_set_codec(dph_index* h, float offset, float scale);             /* default -2, 20 */
/* host -> HBM upload of rows [row0, row0+n) (int8, row-major [n,768]) */
int dph_index_upload_rows(dph_index* h, int64_t row0, int64_t n, const int8_t* host_rows);
/* the same from PINNED host memory, asynchronous on `stream`: the loader reads phrase/<a>-<b>.hdf5 chunk by chunk
 * into two pinned staging buffers and overlaps disk reads with the uploads, so the dump is never whole in host RAM
 * (MIPS.__init__ / load_idx_f, index.py:24-88, load a 2x4 B x N idx2id and leave the vectors on disk) */
int dph_index_upload_rows_async(dph_index* h, int64_t row0, int64_t n, const int8_t* pinned_rows, void* stream);
int dph_host_alloc_pinned(size_t bytes, void** out);
int dph_host_free_pinned(void* p);
int dph_stream_synchronize(int device, void* stream);
/* fill the whole shard on-device with the deterministic synthetic dump of BASELINE.md config 2
 * (row r, column j: integer Irwin-Hall approximation of float_to_int8(N(0,0.6^2)); reproducible on the host,
 * see densephrases_amd/synth.py) -- the global row index used for hashing is id_base + local row */
int dph_index_fill_synthetic(dph_index* h, uint64_t seed, void* stream);
/* kind 0 = the i.i.d. dump above; kind 1 = SURVEY.md 8(d) config-4 data: a mixture of 4096 Gaussians
 * (sigma_between 0.5, sigma_within 0.25) with a sprinkling of SATURATED outlier rows (every code +127 / -128) -- dense
 * score neighbourhoods and extreme row norms, the shape real phrase dumps have; kind 3 = the same mixture without the
 * saturated rows; kind 2 = a document-ordered dump (runs of 56..200 consecutive near-duplicate rows: hits come in bursts
 * for a scan that walks the ids in order); all reproducible on the host in synth.py */
int dph_index_fill_synthetic_kind(dph_index* h, uint64_t seed, int kind, void* stream);
/* idx2id (index.py:78-88): doc / word of every local row, int32 [n_rows], host pointers */
int dph_index_set_idx2id(dph_index* h, const int32_t* doc, const int32_t* word);
/* a shard merged from several sub-indexes (the reference's parallel build: one dump per process added with
 * `--offset k*max_idx`, then merge_indexes: scripts/parallel/add_to_index.py:42-51, build_phrase_index.py:282-297;
 * decoded at index.py:135-140): group g holds stored rows [row_starts[g], row_starts[g+1]) under the ids
 * id_offsets[g] + (row - row_starts[g]).  I of the searches, the ids the window re-score and reconstruct take, and
 * dph_id2docword then all speak those ids; idx2id stays indexed by stored row.  n_groups = 0 restores id_base + row. */
int dph_index_set_id_groups(dph_index* h, int n_groups, const int64_t* id_offsets, const int64_t* row_starts /*[n_groups+1]*/);
/* per-document f2o_start of the dump (embed_utils.py:130,246), CSR over documents sorted by doc id:
 * doc_ids[n_docs] ascending, f2o_off[n_docs+1], f2o[f2o_off[n_docs]] -- host pointers */
int dph_index_set_f2o(dph_index* h, int64_t n_docs, const int32_t* doc_ids, const int64_t* f2o_off,
                      const int32_t* f2o);
/* must be called after the rows are in place and before searching: computes the shard statistics the
 * exactness certificate needs (max centred row norm) */
int dph_index_finalize(dph_index* h, void* stream);
/* what finalize found: the row-norm constant of the certificate (over non-outlier rows), the true maximum, and how
 * many rows were set aside as outliers (scored exactly against every query instead of being bounded) */
int dph_index_shard_stats(dph_index* h, double* rmax, double* rmax_all, int* n_outliers);
/* The aux layout of a finalized shard (no reference counterpart: internal to the exact search that replaces faiss Index.search,
 * index.py:200).  dph_index_finalize measures the per-dimension mean codes of the stored rows and decides whether the shard needs
 * AUX ROWS: per-row norm codes when the row norms are heavy-tailed, plus raw codes of "rogue" dimensions (mean far from the other
 * dimensions' for every row -- BERT-family vectors clipped by embed_utils.py:141-149) whose query digits get further high digits.
 * The layout fixes how a query row is cut into integer digits, and the ranks of a range-sharded job exchange INTEGER scores
 * (dph_search_sample_dev / dph_union_bounds_dev / dph_search_bounded_dev): read the layout of every rank's shard, agree on one
 * (the widest stride; the replica table of the lowest rank that has one; the smallest q2max) and set it on every rank before the
 * first search -- densephrases_amd/dist.py sync_aux_layout does.  layout[0] stride (0 none, 4 norm codes, 16 / 32 norm codes +
 * up to 12 / 24 replica slots), [1] norm slots, [2] replica slots, [3] clamp of the low digit, [4 .. 27] dimension of replica slot i (-1 unused). */
#define DPH_AUX_LAYOUT_INTS 28
int dph_index_get_aux_layout(dph_index* h, int32_t* layout);
int dph_index_set_aux_layout(dph_index* h, const int32_t* layout);
int64_t dph_index_ntotal(const dph_index* h);      /* faiss Index.ntotal (index.py:34,128) */
int     dph_index_dim(const dph_index* h);         /* faiss Index.d      (index.py:32)     */
int     dph_index_device(const dph_index* h);
/* device pointer of the resident rows (for callers that fill the shard themselves, e.g. from a torch tensor) */
void*   dph_index_rows_dev(dph_index* h);

/* ---- faiss Index.search (index.py:200) ---------------------------------------------------------------
 * x: [n,768] fp32 row-major; D: [n,k] fp32 descending; I: [n,k] int64 global ids (id_base + local row);
 * fewer than k rows -> I = -1, D = -FLT_MAX (FAISS padding).  Exact inner product over the fp32
 * de-quantised rows, ties ordered (score desc, id asc).  Host-pointer form: synchronous, retries wider /
 * exact scans until every row is certified exact; returns DPH_E_UNCERTIFIED only if that is impossible.
 * NON-FINITE query rows (a NaN / Inf element; index.py:195-200 hands FAISS whatever the encoder produced): FAISS' flat search
 * answers such a row with ids -1 / scores -FLT_MAX (no score compares greater than its heap's threshold) and the other rows as
 * ever.  Same here, on every index type: that row gets I = -1, D = -FLT_MAX, status DPH_ROW_NONFINITE (device forms) and is counted
 * in dph_search_stats.nonfinite; the call returns DPH_OK and the row costs what an ordinary row costs.  Finite rows of any
 * magnitude (1e30, all zeros) are searched exactly. */
int dph_search(dph_index* h, const float* x, int64_t n, int k, float* D, int64_t* I);
/* device-pointer form, asynchronous on `stream`, no host round trip: the first attempt, then -- gated by a
 * device-side count, a few empty launches when nothing failed -- a retry scan of the uncertified rows under a bound
 * derived from their own k-th best integer score, then the fp64 full scan for up to 32 rows that still fail.
 * status_dev [n] int32 receives 0 = certified exact, 1 = not certified (only if more than 32 rows needed the fp64
 * scan, or boundary ties exceed its 1 M-row buffer; dph_search settles those too), 3 = non-finite query row.  n <= 2^20. */
int dph_search_dev(dph_index* h, const float* x_dev, int64_t n, int k, float* D_dev, int64_t* I_dev,
                   int32_t* status_dev, void* stream);
/* statistics of the last search on the handle; after a device-pointer call this synchronises the device */
int dph_search_get_stats(dph_index* h, dph_search_stats* out);

/* ---- IVF with exact in-list inner product (BASELINE.json configs[3]; the reference's index is an IndexIVFPQ whose
 * coarse quantizer is an IndexFlatIP searched with nprobe = 256: build_phrase_index.py:99,113-116, index.py:53,62).
 * The shard is stored LIST-MAJOR: rows of one inverted list are contiguous and every list is padded to a multiple of
 * 32 rows, so each 32-row tile belongs to exactly one list.
 *   dph_index_set_row_ids: row_ids[n_rows] = global id of every stored row (-1 = padding), a permutation of
 *                          [id_base, id_base + n_ids); afterwards ntotal = n_ids and idx2id is indexed by id - id_base.
 *                          Call before set_idx2id / finalize.  dph_search on such a shard is still the exact search.
 *   dph_index_set_ivf:     centroids [nlist,768] fp32 (nlist <= 2^20) and tile_list[ceil(n_rows/32)] = the list of every tile.
 *   dph_search_ivf(_dev):  per query row the nprobe lists with the largest <q, centroid> are probed (scores on the matrix
 *                          cores, v_mfma_f32_32x32x2_f32; the lists inside the fp32 error band around the nprobe-th score
 *                          are re-ranked in float64, ties by list id, so the probed set is the float64 oracle's);
 *                          result = exact top-k over the rows of those lists, same ordering, padding, certificate and
 *                          retry rules as dph_search.  Two scans serve it (tuning key "ivf_units"): the unit scan
 *                          groups the work by list -- a list meets only the query rows that probe it, 1024 rows per
 *                          HBM-bound read of the probed lists -- and the masked scan streams every tile under a per-tile
 *                          probe mask (many short lists).  Tuning key "nprobe" (dph_index_set_tuning) makes the entry points
 *                          WITHOUT an nprobe argument -- dph_search(_dev), the sharded sample / bounded pair -- search
 *                          IVF too, which is how a list-major dump is served range-sharded over several GPUs.
 *   dph_ivf_assign_dev:    list assignment for the list builder / a k-means step (build_phrase_index.py:96-153 does this
 *                          with faiss): best[r] = arg-max_l <x_r, c_l> + bias[l] (bias may be NULL; -||c_l||^2/2 gives the
 *                          L2 assignment), gap[r] = distance to the runner-up (re-check near-ties in float64), with the
 *                          same MFMA tile, fused with the arg-max: a workgroup owns 128 rows and walks the list tiles, no
 *                          [n, nlist] score matrix is written (scores_dev is ignored and may be NULL); device pointers.
 *   dph_index_assign_dev:  the same for rows [row0, row0+n) of the RESIDENT shard: the int8 rows are de-quantised through the
 *                          shard's LUT while they are staged, so a whole dump is assigned where it lies in HBM. */
int dph_index_set_row_ids(dph_index* h, const int64_t* row_ids, int64_t n_ids);
int dph_index_set_ivf(dph_index* h, int nlist, const float* centroids, const int32_t* tile_list);
int dph_search_ivf(dph_index* h, const float* x, int64_t n, int k, int nprobe, float* D, int64_t* I);
int dph_search_ivf_dev(dph_index* h, const float* x_dev, int64_t n, int k, int nprobe, float* D_dev, int64_t* I_dev,
                       int32_t* status_dev, void* stream);
int dph_ivf_assign_dev(int device, const float* x_dev, int64_t n, const float* centroids_dev, int nlist, const float* bias_dev,
                       float* scores_dev, int32_t* best_dev, float* gap_dev, void* stream);
int dph_index_assign_dev(dph_index* h, int64_t row0, int64_t n, const float* centroids_dev, int nlist, const float* bias_dev,
                         int32_t* best_dev, float* gap_dev, void* stream);
/* ---- training the coarse quantizer where the rows lie (build_phrase_index.py:60-142: sample_data + faiss IndexIVF::train;
 * FAISS runs k-means with the quantizer -- an IndexFlatIP -- as the assignment index and normalises the centroids of
 * inner-product indexes: ClusteringParameters::spherical).  All pointers are device pointers.
 *   dph_index_gather_rows_dev: sample_dev[i] = int8 row rows_idx_dev[i] (stored-row index) of the resident shard, [m,768].
 *   dph_kmeans_step_dev:       ONE Lloyd iteration over int8 rows [m,768] in the shard's codec: assignment = arg-max_l
 *                              <x_r, c_l> + bias[l] (the fused MFMA GEMM + arg-max of dph_ivf_assign_dev; bias NULL = the
 *                              inner-product assignment FAISS uses for these indexes), update = de-quantised mean of the
 *                              members from exact integer sums (64-bit atomics), L2-normalised when `spherical`; empty
 *                              lists keep their centroid and report count 0 (the caller splits a large list into them).
 *                              centroids_dev [nlist,768] is read and overwritten; assign_dev [m], gap_dev [m] (scratch),
 *                              counts_dev [nlist] are outputs. */
int dph_index_gather_rows_dev(dph_index* h, const int64_t* rows_idx_dev, int64_t m, int8_t* sample_dev, void* stream);
int dph_kmeans_step_dev(dph_index* h, const int8_t* rows_dev, int64_t m, float* centroids_dev, int nlist, const float* bias_dev,
                        int spherical, int32_t* assign_dev, float* gap_dev, uint32_t* counts_dev, void* stream);
/* rows the shard STORES (ntotal + the padding rows of a list-major shard) */
int64_t dph_index_stored_rows(const dph_index* h);
/* The device-side list builder: a FLAT shard whose rows are resident (uploaded or generated) becomes a list-major IVF
 * shard without the rows leaving the GPU -- assign_dev[n_rows] (device, e.g. from dph_index_assign_dev) names the list
 * of every row; the rows are sorted by (list, id), every list padded to whole tiles, row_ids / tile_list / centroids set
 * as dph_index_set_row_ids + dph_index_set_ivf would (centroids: HOST pointer [nlist,768]).  Ids do not change.  Needs
 * room for a second copy of the rows while it runs.  Call dph_index_finalize afterwards.  (The add-to-index step of
 * build_phrase_index.py:145-153.) */
int dph_index_make_list_major(dph_index* h, const int32_t* assign_dev, int nlist, const float* centroids, void* stream);
/* Move the resident rows into a freshly allocated buffer (device-to-device copy, the old buffer is freed afterwards).
 * Written to test whether the buffer dph_index_make_list_major allocates next to the original streams slower (it does
 * not: profiles/r03_list_major_probe_after_fix.json, the slow full-size list-major scans of round 2 were a truncated
 * gather); kept as a compaction utility.  Needs room for a second copy of the rows while it runs. */
int dph_index_rehome_rows(dph_index* h, void* stream);

/* ---- the reference's OWN index type in HBM: IndexPreTransform(OPQMatrix(768, M)) -> IndexIVFPQ(IndexFlatIP, 768, nlist, M, 8 bits,
 * METRIC_INNER_PRODUCT, by_residual) as build_phrase_index.py:108-116 trains it, add_to_index / merge_indexes (:145-153, :282-338)
 * fill it and index.py:30-33 reads it with faiss.read_index(..., IO_FLAG_ONDISK_SAME_DIR).  densephrases_amd/faiss_io.py parses
 * index.faiss / merged.invdata; these calls take the pieces (host pointers):
 *   dph_index_create_pq          an empty handle for ntotal codes in nlist lists, M sub-quantisers (M a multiple of 16 dividing 768)
 *   dph_index_set_pq             A [768,768] row-major and b [768] of the pre-transform x' = A x + b (NULL = identity / no bias;
 *                                index.py:32 reads A as `R`), coarse centroids [nlist,768], PQ codewords [M,256,768/M]
 *   dph_index_set_pq_list_sizes  sizes[nlist]; the codes of list l occupy positions [sum(sizes[:l]), sum(sizes[:l+1]))
 *   dph_index_upload_pq_codes    codes [n,M] uint8 and ids [n] int64 for positions [pos0, pos0+n)  (any chunking)
 *   dph_index_set_idx2id / set_id_groups / set_f2o  as for a raw-dump shard (idx2id is indexed by local id), then dph_index_finalize
 *                                (builds the id -> position direct map: FAISS DirectMap.Hashtable, build_phrase_index.py:138-142).
 * On such a handle dph_search(_dev) / dph_search_ivf(_dev) are FAISS' IVFPQ search (index.py:200): x' = A x + b, the nprobe lists of
 * largest <x', centroid> (tuning key "nprobe", default 256 like index.py:53,62), score = <x', c_list> + sum_m LUT[m][code[m]] with
 * LUT[m][j] = <x'_m, codeword[m][j]> -- x', <x', c> and the LUT accumulated in float64 and rounded once, the code sum in fp32
 * sequentially in m -- top-k in (score desc, id asc) order with -1 / -FLT_MAX padding; status 0 = the exact top-k of the probed
 * lists.  dph_reconstruct is IndexIVFPQ::reconstruct (centroid + decoded residual, ROTATED space, DPH_E_NOTFOUND for unknown
 * ids: index.py:31,285-288) and dph_rescore(_dev) re-scores windows of reconstructed vectors un-rotated by R = A like
 * index.py:340,365 (as <A q, v'>); with `vecs`, [c,0,:] = v'_own R and [c,1,:] = v'_argmax R R -- the reference multiplies its
 * pred_*_vecs by R a second time (:345,370 then :381-389), replicated as is.
 * dph_index_get_transform returns A (identity on raw-dump shards): what index.py:32 calls self.R. */
int dph_index_create_pq(int device, int64_t ntotal, int nlist, int M, dph_index** out);
int dph_index_set_pq(dph_index* h, const float* A, const float* b, const float* centroids, const float* pq_centroids, int by_residual);
int dph_index_set_pq_list_sizes(dph_index* h, const int64_t* sizes);
int dph_index_upload_pq_codes(dph_index* h, int64_t pos0, int64_t n, const uint8_t* codes, const int64_t* ids);
int dph_index_get_transform(dph_index* h, float* A_out /* [768*768] */);

/* ---- faiss reconstruct (index.py:31, 286, 296) ---- de-quantised fp32 row of a global id */
int dph_reconstruct(dph_index* h, int64_t id, float* out768);

/* ---- MIPS.get_idxs (index.py:124-141) ---- ids are clipped to [0, ntotal) like the reference */
int dph_id2docword(dph_index* h, const int64_t* I, int64_t n, int32_t* doc, int32_t* word);

/* ---- start/end window re-scoring (index.py:323-370) --------------------------------------------------
 * For each of n candidates c (flattened [B*k]) with global id ids[c], (doc[c], word[c]) and first-stage
 * score first[c], and the query half qhalf[c / k] (the END half for direction 0, the START half for 1):
 *   direction 0 ("find end for start", :323-346): slot i in [0,L) -> row ids[c]+i, word[c]+i
 *   direction 1 ("find start for end", :348-371): slot s in [0,L) -> i = L-1-s, row ids[c]-i, word[c]-i
 * score[c,s] = (double)first[c] + (double)(float)dot(qhalf, dequant(row)) + (valid ? 0 : -1e9), rows outside
 * the shard count as zero vectors (reconstruct failure, :285-288); valid = valid_phrase(:305-321) from the
 * f2o CSR.  Outputs: pred_word[c] = word of the arg-max slot or -1, best[c] = max score (double),
 * argslot[c]; if vecs != NULL, vecs[c,0,:] = candidate's own de-quantised row and vecs[c,1,:] = the
 * arg-max slot's row (fp32, [n,2,768]) for return_idxs (:381-389).  Host pointers, synchronous. */
int dph_rescore(dph_index* h, int direction, const float* qhalf /*[n_q,768]*/, int64_t n_q, int k, int L,
                const int64_t* ids, const int32_t* doc, const int32_t* word, const float* first,
                int32_t* pred_word, double* best, int32_t* argslot, float* vecs);
/* device-pointer form, asynchronous on `stream` (ids/doc/word/first and all outputs are device pointers;
 * doc/word may be NULL: they are then looked up from the resident idx2id) */
int dph_rescore_dev(dph_index* h, int direction, const float* qhalf_dev, int64_t n_q, int k, int L,
                    const int64_t* ids_dev, const int32_t* doc_dev, const int32_t* word_dev,
                    const float* first_dev, int32_t* pred_word_dev, double* best_dev, int32_t* argslot_dev,
                    float* vecs_dev, void* stream);

/* ---- start/end-vector scoring of densephrases/encoder.py (BASELINE.json north_star) ----------
 * out[b, m] = <q[b, :], vecs[b, m, :]>, fp32, device pointers, asynchronous on `stream`:
 *   train_query (encoder.py:383-386): q = query_start [B,768] (the [B,1,768] CLS vector), vecs = start_vecs [B,M,768]
 *     -- the vectors MIPS.search(return_idxs=True) returns --, out = start_logits [B,M]; the same for the end side,
 *     logits = start_logits + end_logits is an add the caller does;
 *   forward (encoder.py:206-207): q = query_start [bs,768], vecs = start [bs,T,768], out = start_logits [bs,T].
 * dph_score_vecs_bwd_dev is its gradient w.r.t. q (grad_q[b,:] = sum_m grad[b,m] * vecs[b,m,:]), what query-side
 * fine-tuning back-propagates into the query encoder (train_query.py:208-275).
 * dph_dense_logits_dev: out[b,i,j] = start_logits[b,i] + end_logits[b,j]  (encoder.py:208). */
int dph_score_vecs_dev(int device, const float* q_dev, const float* vecs_dev, int64_t n_b, int64_t m, float* out_dev, void* stream);
int dph_score_vecs_bwd_dev(int device, const float* grad_dev, const float* vecs_dev, int64_t n_b, int64_t m, float* grad_q_dev,
                           void* stream);
int dph_dense_logits_dev(int device, const float* start_logits_dev, const float* end_logits_dev, int64_t n_b, int64_t T,
                         float* out_dev, void* stream);

/* ---- two-phase search of a range-sharded dump (no reference counterpart) -----------------
 * Every shard's scan is only as cheap as its pre-pass bound is tight, and the bound that matters for the MERGED top-k
 * is the one over all shards.  Phase 1 returns, per query row, the DPH_SAMPLE_KEEP (16) best integer scores of this
 * shard's pre-pass sample (INT32_MIN padded; identical x gives identical score units on every rank).  After an
 * all-gather of those [n,16] arrays dph_union_bounds_dev takes the 16-th best of the union, minus one -- a lower bound
 * of the 16-th best row of the whole dump.  Phase 2 scans under these bounds; a shard may then hold fewer than k rows
 * above the bound, so besides D/I (padded with -1) it returns bound_dev[r] = an upper bound of the reference score of
 * every row it did NOT return, and status 0 (own top-k closed) or 2 (decided by dph_merge_records_dev). */
int dph_search_sample_dev(dph_index* h, const float* x_dev, int64_t n, int32_t* top_dev /* [n,16] */, void* stream);
int dph_union_bounds_dev(int device, const int32_t* top_parts /* [n_parts,n,16] */, int n_parts, int64_t n,
                         int32_t* tau_dev /* [n] */, void* stream);
int dph_search_bounded_dev(dph_index* h, const float* x_dev, int64_t n, int k, const int32_t* tau_dev,
                           float* D_dev, int64_t* I_dev, int32_t* status_dev, double* bound_dev, void* stream);

/* ---- multi-GPU merge (no reference counterpart) --------------------------------
 * D_parts/I_parts: per-shard results [n,k], part p at byte offset p*part_stride_bytes from each base pointer
 * (device pointers, e.g. views into one packed all-gather buffer);
 * writes the global top-k in (score desc, id asc) order to D_out/I_out [n,k] and, if src_out != NULL,
 * the (part, column) each winner came from as part*k+col (int32 [n,k]).  Asynchronous on `stream`. */
int dph_merge_topk_dev(int device, const float* D_parts, const int64_t* I_parts, int n_parts,
                       int64_t part_stride_bytes /* 0 = dense [n_parts,n,k] */, int64_t n, int k,
                       float* D_out, int64_t* I_out, int32_t* src_out, void* stream);

/* The whole post-all-gather step of a sharded search in one launch: the same merge, and every winner takes the
 * window re-score results of its home shard along -- best_parts f64 [n,k], pred_parts i32 [n,k] (dph_rescore_dev
 * outputs), status_parts i32 [n] (dph_search_dev status), all with the same part stride.  Padding slots get
 * best = -1e9, pred = -1; status_out[r] = 0 iff every shard certified the row, 3 when the shards flagged it non-finite, else 1.
 * bound_parts (f64 [n] per part, may be NULL) are the dph_search_bounded_dev bounds: a part with status 2 ("decided
 * after the merge") is certified iff the merged k-th score beats its bound. */
int dph_merge_records_dev(int device, const float* D_parts, const int64_t* I_parts, const double* best_parts,
                          const int32_t* pred_parts, const int32_t* status_parts, const double* bound_parts,
                          int n_parts, int64_t part_stride_bytes /* 0 = dense per-field arrays */, int64_t n, int k,
                          float* D_out, int64_t* I_out, double* best_out, int32_t* pred_out, int32_t* status_out,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DPH_H */
