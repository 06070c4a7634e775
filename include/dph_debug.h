/* dph_debug.h -- the part of libdph's C ABI that is NOT the drop-in contract (include/dph.h): tuning keys, measurement hooks, the
 * two-batches-in-flight plumbing (twin handles, CU-range streams, the two-stage search) and the dph_debug_* hooks the tests and
 * tools use to look inside a search.  Same library, same DPH_ABI_VERSION; a host that only serves searches never includes this.
 * None of these has a counterpart in the reference (FAISS exposes no such hooks through densephrases/index.py).  */
#ifndef DPH_DEBUG_H
#define DPH_DEBUG_H

#include "dph.h"

#ifdef __cplusplus
extern "C" {
#endif

/* tuning knobs of the search pipeline (all have defaults; values are int32):
 *   "ladder"        explicit pre-pass strides, coarse -> fine (empty = derived from the shard size, {0} = none)
 *   "fine_stride"   stride of the finest sampled level when the ladder is derived (0 = default: 32 / 16)
 *   "sample_kp"     a level's bound is its kp-th best sampled score (default 16)
 *   "max_qb"        1 = passes of 128 query rows only, 2 = passes of 256 rows when more than 128 are left (default)
 *   "nprobe"        > 0 on a shard with IVF data: entry points without an nprobe argument search IVF with this nprobe
 *   "ivf_units"     IVF scan of a list-major shard whose lists are contiguous runs of tiles: 1 = unit scan (work queue of
 *                   (list chunk, segment) units with gathered query fragments, up to 1024 query rows per pass, unprobed
 *                   lists never read), 0 = masked scan (every tile, 256 rows per pass), -1 = unit scan when the lists
 *                   average >= 64 tiles (default)
 *   "ivf_spread"    1 = a chunk's query rows are dealt over the four scan waves first (default), 0 = packed
 *   "scan_seg"      shortest segment (tiles) the flat scan's work queue deals (default 64)
 *   "ladder_fuse"   1 (default): on flat shards the full scan skips the tiles the finest sampled level already scanned and
 *                   accumulates into that level's buckets -- the dump is read once per batch, not 1 + 1/32 times; 0 = off
 *   "retry_chain"   1 (default): rows the first attempt cannot certify are re-scanned on the device under their own bound, then through
 *                   the fp64 scan; 0 = first attempt only, such rows come back with status 1 (measurements, diagnostics)
 *   "aux"           aux rows of the shard (dph_index_get_aux_layout): -1 (default) = dph_index_finalize decides from the rows, 0 = none
 *                   (one shard-wide norm bound), 4 = per-row norm codes, 16 / 32 = norm codes + 12 / 24 replica digits of the rogue dimensions
 *   "scan_grid"     persistent workgroups of the scan kernels, one per CU: 0 (default) = the device's CU count, fewer leave CUs idle for
 *                   other streams (set before the first search; tools/scan_grid_probe.py)
 *   "side_grid"     scan workgroups of the sampled levels in dph_search_prepare_dev (0 = scan_grid): the CUs of the side stream
 *   "scan_sched"    hand-over schedule of the flat full scan, one value for both kernels or two (128-row, 256-row kernel):
 *                   0 = every wave stages its pieces of a tile right behind the tile's barrier, 1 = one wave after the other
 *                   (default), 2 = interleaved, one wave per k-step (same results; profiles/r04_scan_scheds_170M.json)
 *   "coarse_filter" PQ index with >= 2^16 lists: 1 .. 5 = the coarse quantizer (index.py:53 nprobe lists by <x', c>) runs a
 *                   one-product bf16 filter with the threshold test in its epilogue in front of the float64 re-rank.  5 (default,
 *                   round 5): a filter SCAN -- the centroids as 24 KiB pieces with the byte layout of an int8 tile through the flat
 *                   scan's feed, 128 query rows per read (0.27 ms = 0.76 of the HBM peak for 2^20 centroids); 3: a GEMM with the
 *                   centroids straight into MFMA operand registers from a fragment-major image (0.36 ms); 1 / 2: centroids and
 *                   queries staged through LDS, 2 with non-temporal loads; 4: 3 on contiguous runs of tiles; 0 = the three-product
 *                   bf16 GEMM over the whole score matrix (the fail-over chain) alone; same probe set, same candidate pool
 *   "coarse_teams"  PQ index, filter scan (coarse_filter 5), a pass of more than 128 query rows: 1 (default) = ONE launch in which teams of
 *                   2 / 4 / 8 workgroups of one XCD stream the same run of centroid tiles, each against its own group of 128 rows, and share
 *                   them through the XCD's L2 (the image leaves HBM once per pass); 0 = one launch -- one read of the image -- per 128 rows
 *   "pq_split_lut"  PQ index, OPQ96, row-major ADC scan: 1 = the last sixteen look-up tables are gathered from global memory instead
 *                   of LDS (an experiment to relieve the bank-conflict-bound LDS: measured slower, default 0; same results) */
int dph_index_set_tuning(dph_index* h, const char* key, const int32_t* values, int n_values);

/* (row, query row) pairs the LAST scan launch on the handle emitted and how often a wave took its emit path */
int dph_scan_counters(dph_index* h, int64_t* pairs_out, int64_t* triggers_out);
/* the same per scan wave: pairs_out[w] for the n_waves = 4 * (scan workgroups) waves of the last scan launch of the
 * first attempt (image 0) or of the retry passes (image 1); out_cap = entries of pairs_out.  The pairs of all waves share
 * one pool of chunks (csrc/dph_internal.h), so a large count in one wave costs nothing but its share of the pool. */
int dph_debug_wave_pairs(dph_index* h, int image, uint32_t* pairs_out, int out_cap, int* n_waves);

/* ---- two batches in flight on one shard (no reference counterpart: FAISS runs one search at a time, index.py:200) ------------
 * Of the ~21 ms a batch of 64 takes on a 170 M-row shard, ~1.3 ms in front of the full scan are a chain of small dependent launches
 * (quantise, three sampled levels each with refine + threshold): latency, not bandwidth.  They only need the batch's queries, so they
 * can run for batch t+1 WHILE batch t's full scan streams the dump -- on a few CUs set aside for them, since the scan kernel fills
 * every CU it is given (one persistent workgroup per CU, 132 KiB of LDS):
 *   - dph_index_create_twin: a second handle over the SAME rows, metadata and shard constants with search scratch of its own (one
 *     handle per batch in flight; the twin owns only that scratch, is destroyed before the index, and an index with twins -- or a twin
 *     -- refuses every call that would change rows or metadata; flat shards only);
 *   - dph_stream_create_cu_range: a HIP stream whose kernels run on CUs [first_cu, first_cu + n_cus) of the CU-mask bit order only
 *     (MI355X: bit i is a CU of XCD i % 8 -- a run of 8 bits is one CU of every XCD; profiles/r04_cu_mask_probe.txt);
 *   - dph_search_prepare_dev (quantise + sampled levels, scan launches of `side_grid` workgroups: tuning key, = the side stream's
 *     CUs) and dph_search_finish_dev (full scan fused with the finest level, refine, select, retry chain) enqueue together exactly
 *     the launches of dph_search_dev on the same handle: same D / I / status.  One pass per call (n <= 128 x max_qb rows); the caller
 *     orders the two stages of a batch, and the re-use of a handle by the batch after next, with events.
 * densephrases_amd.dist.PipelinedSearcher drives it: side stream = 8 CUs, main stream = the other 248 (tuning key "scan_grid"). */
int dph_index_create_twin(dph_index* index, dph_index** twin_out);
int dph_stream_create_cu_range(int device, int first_cu, int n_cus, void** stream_out);
int dph_stream_destroy(void* stream);
int dph_search_prepare_dev(dph_index* h, const float* x_dev, int64_t n, int k, void* stream);
int dph_search_finish_dev(dph_index* h, const float* x_dev, int64_t n, int k, float* D_dev, int64_t* I_dev, int32_t* status_dev,
                          void* stream);

/* ---- measurement hook (bench.py): when on, every scan launch (PQ index: the coarse quantizer's filter GEMM, its dominant
 * kernel) is bracketed by HIP events on its stream;
 * dph_profile_read synchronises those events and returns the summed kernel time and launch count since the
 * last read (roofline: algorithmic bytes per launch / average launch duration). */
int dph_profile_enable(dph_index* h, int on);
int dph_profile_read(dph_index* h, double* scan_ms_total, int* scan_launches);
/* The same read, with the scan launches of the ladder levels (the sampled pre-passes that find the bounds, index.py:200's
 * search has no such step) reported next to the full scans: per batch, all HBM-bound scan time = both sums. */
int dph_profile_read_all(dph_index* h, double* scan_ms_total, int* scan_launches, double* ladder_ms_total, int* ladder_launches);
/* Every bracketed launch since the last read on its own, in launch order: ms_out[0 .. min(*n_out, cap)) (which = 0: the full scans -- on a
 * PQ index the coarse filter scans, each launch alone inside its pair --, 1: the ladder levels' scans).  What `rocprofv3 --kernel-trace`
 * reports per dispatch of the same kernel: bench.py --trace_out writes both side by side. */
int dph_profile_read_each(dph_index* h, int which, double* ms_out, int cap, int* n_out);

/* ---- debug / test hooks.  dph_debug_scan_buckets runs the quantiser, ONE filter-scan launch over every
 * `tile_stride`-th tile for the first n <= 256 rows of x (under the per-row integer bounds tau_host, or cold when
 * NULL) and the refine step, and returns each row's bucket: keys_host [n][32768] uint64 keys
 * ((score ^ 0x80000000) << 32 | ~row: the exact integer score 128*<q1,n> + <q2,n> of a database row) and
 * counts_host[n] (bit 31 set = pairs were lost).  Every visited row r with 128*H(r) + lmax > tau must be there.
 * dph_debug_lmax returns the low-digit bounds the last quantiser run computed. */
int dph_debug_scan_buckets(dph_index* h, const float* x, int64_t n, const int32_t* tau_host, int tile_stride,
                           uint64_t* keys_host, uint32_t* counts_host);
int dph_debug_lmax(dph_index* h, int64_t n, int32_t* lmax_host);
/* Aux rows [row0, row0 + n_rows) of the shard (stride bytes each), the aux digits [n_q][32] of the last quantiser run, and
 * info[4] = {stride, norm unit, low-digit clamp, replica slots}; dph_debug_mu: the per-dimension mean codes [768].  With aux rows a
 * visited row is emitted iff  <q1, n> + sum_s aux[row][s] * qaux[q][s]  >  floor((tau - lmax) / 128). */
int dph_debug_aux(dph_index* h, int64_t row0, int64_t n_rows, int8_t* aux_host, int64_t n_q, int8_t* qaux_host, int32_t* info);
int dph_debug_mu(dph_index* h, int32_t* mu_out);
/* Timing hook (tools/scan_diag.py): quantise the first n <= 256 rows of x (host) and launch the full filter scan `iters`
 * times under a bound nothing reaches -- every tile is streamed and multiplied, nothing is emitted -- each launch bracketed
 * by HIP events; ms_out[iters] receives the launch durations.  The kernel of index.py:200's faiss search, alone. */
int dph_debug_scan_time(dph_index* h, const float* x, int64_t n, int iters, float* ms_out);
/* Candidate buckets of the LAST pass scanned on this handle: raw_out[n] = keys the refine step counted per query row of the pass (more
 * than 8192 cannot be sorted, more than 32768 do not fit: "lost pairs"), overflow_out[n] != 0: the pair pool ran dry for that row.
 * With tuning key "retry_chain" = 0 the last pass is the first attempt's. */
int dph_debug_bucket_counts(dph_index* h, int64_t n, uint32_t* raw_out, uint32_t* overflow_out);
/* PQ index, coarse quantizer of the LAST pass searched (tuning key "coarse_filter"): out[0] = 1 when the filter form failed over to
 * the three-product chain (0xFFFFFFFF: the filter form has not run), out[1] = (row, list) candidates its GEMM epilogue emitted. */
int dph_debug_pq_coarse(dph_index* h, uint32_t out[2]);
/* PQ index, bookkeeping of the LAST pass of the ADC scan: info[8] = {list-major pairs, work items taken, row-major units, 0, candidate
 * capacity per row, unit capacity, pair capacity, rows the scratch is sized for}; per_row[n_rows][3] = {candidates appended, overflow
 * flag (any stage: coarse band, work queue, candidates), the row's final bound key}.  What a row with status 1 ran into. */
int dph_debug_pq_pass(dph_index* h, int32_t info[8], uint32_t* per_row, int n_rows);
/* the (list, score key) pairs [cap][2] and query rows [cap] of the candidate pool that pass left behind; *count = triples in the pool */
int dph_debug_pq_pool(dph_index* h, uint32_t* lk_host, uint16_t* q_host, int64_t cap, int64_t* count);
/* Phase clocks of the PQ search chain, 100 MHz ticks.  The first call (out may be null) arms a clock; later calls copy what the LAST
 * launch left.  which = 0, the row-major ADC scan (many short lists): out[wg][8] = {start, end, table + list offsets, dis0, look-up
 * sums, k-th selection + append, units taken, codes summed} per workgroup, *n_wgs = workgroups of the launch; costs one barrier per
 * segment while armed.  which = 1, the probe selection (dph_coarse_select_kernel): out[row][8] = stamps {start, query norm, candidates
 * in LDS, nprobe-th candidate, marking, float64 band dots, band ranks} and [7] = band lists | candidates << 16 | lists still needed
 * << 40 per query row of the pass; armed for the process (every handle on the device). */
int dph_debug_pq_phases(dph_index* h, int which, uint64_t* out, int cap_wgs, int* n_wgs);
/* Work queue of the last IVF unit-scan pass: out[0] = chunks, out[1] = units, out[2] = capacity error flag,
 * out[3] = units taken by the full scan (>= out[1] + workgroups when the queue was drained). */
int dph_debug_units(dph_index* h, int32_t out[4]);
/* Segment `u` of the flat scan's work queue over n_tiles visited tiles (host twin of the device function the kernel
 * calls; needs no GPU): returns its first tile (>= n_tiles: the queue is empty from this u on; -1: bad arguments) and
 * its length in *len. */
int64_t dph_debug_guided_segment(int64_t u, int64_t n_tiles, int grid, int seg_min, int64_t* len);
/* The full scan behind a fused finest ladder level of stride `stride` (tuning key "ladder_fuse"; host twin of the plan
 * dph_search makes and of the kernel's index arithmetic; needs no GPU): *visit_out = tiles the full scan visits (0: a shard
 * of n_tiles tiles is not fused at this stride), returns the tile the v-th visit reads (-1: not fused, or v outside
 * [0, *visit_out)).  The visited tiles are exactly the tiles that are not multiples of `stride`, in ascending order. */
int64_t dph_debug_fused_tile(int64_t n_tiles, int stride, int64_t v, int64_t* visit_out);

#ifdef __cplusplus
}
#endif
#endif /* DPH_DEBUG_H */
