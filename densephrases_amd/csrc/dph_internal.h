// dph_internal.h -- shared constants / device structs of libdph (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include "../../include/dph_debug.h"      // (includes dph.h: the contract + the tuning / profiling / debug entry points)

// ---------------------------------------------------------------- geometry of the scan
// A scan workgroup is 4 waves, one per SIMD, one workgroup per CU.  Wave w owns QB groups of 32 query rows for the
// whole launch and keeps their HIGH int8 digit in registers (24 k-steps x 4 registers per group).  One launch ("pass")
// therefore serves 128*QB query rows: QB = 1 is the reference's eval batch (2B = 128 rows, index.py:196-197), QB = 2
// serves 256 rows with one read of the dump (BASELINE configs[3]/[4]: batches of 256 / 512 queries).
// Database rows stream HBM -> hand-owned AGPRs -> LDS in tiles of 32 rows = 24 KiB; every wave reads every tile.
#define DPH_KSTEPS 24               // 768 / 32 int8 per v_mfma_i32_32x32x32_i8
#define DPH_QGROUP 32               // query rows per MFMA column block
#define DPH_QROWS 128               // query rows per pass at QB = 1 (4 waves x 32)
#define DPH_MAX_QB 2
#define DPH_TILE_ROWS 32
#define DPH_TILE_BYTES (DPH_TILE_ROWS * DPH_DIM)     // 24576
#define DPH_SCAN_THREADS 256
#define DPH_QGROUP_FRAG_BYTES (DPH_KSTEPS * 64 * 16) // 24576: [kstep][lane][16 B] fragment image of one 32-row group
#define DPH_CENTER 40               // the code of x = 0 at the default codec ((0 - offset) * scale): centre of the synthetic dumps

// ---------------------------------------------------------------- bounds of the filter (round 5)
// Every bound is a Cauchy-Schwarz bound around the shard's PER-DIMENSION mean code mu (integers, dph_index_finalize):
//   <v, n> = <v, n - mu> + <v, mu> <= ||v||_2 * ||n - mu||_2 + <v, mu>,   <v, mu> exact per query row (dph_quantize_kernel).
// Shards whose rows are NOT alike -- rogue dimensions, heavy-tailed row norms -- additionally carry AUX ROWS (dph_scan.hip "the aux
// k-step"): `stride` bytes per stored row that a 25th k-step of the scan multiplies with the query row's aux digits, slot by slot.
#define DPH_AUX_SLOTS 32            // int8 slots of the aux k-step = bytes of a query row's aux digits
#define DPH_AUX_REP_MAX 24          // replica slots (further high digits of rogue dimensions)
struct dph_aux_layout {
    int stride;                     // bytes per stored row of the aux array: 0 = the shard has none, 4 = 4 norm slots, 16 = 4 norm + 12 replica slots, 32 = 8 + 24
    int n_norm;                     // slots [0, n_norm): codes of the row's centred norm, their sum = ceil(|| n - mu ||_2 / norm_unit)
    int n_rep;                      // slots [n_norm, n_norm + n_rep): the row's raw code in dimension rep_dim[s - n_norm]
    int q2max;                      // clamp of the low digit (64 unless the norm unit is large: the norm slots' query digit must fit int8)
    short rep_dim[DPH_AUX_REP_MAX]; // a dimension with R replica slots has 1 + R high digits: | q1 | <= 127 (1 + R)
};

// what the scan emits: one (row, query row) pair per database row whose high-digit score may beat the row's bound
// Pairs live in a POOL shared by all scan waves of a launch, handed out in chunks: a wave claims a chunk with one atomic
// when it first emits and again whenever its chunk is full, so a burst of hits in one wave (a workgroup streaming a whole
// inverted list of near neighbours, a run of near-duplicate rows) takes what it needs from the common store instead of
// overflowing a fixed per-wave region.  Rows lose pairs only when the whole pool is exhausted.
#define DPH_CHUNK_PAIRS 256         // pairs per chunk (2 KiB)
#define DPH_POOL_CHUNKS 32768       // chunks in the pool: 8 Mi pairs = 64 MiB
#define DPH_BUCKET_CAP 32768        // exact-integer-score keys per query row after the refine step (256 KiB)
#define DPH_POOL_MAX 8192           // keys the select kernel sorts in LDS
#define DPH_SELECT_C_MAX 2048       // candidates a retry pass re-scores in fp64 (first attempt: max(2k, k+32))
#define DPH_EXACT_ROWS_DEV 32       // rows per call the on-device fp64 fallback serves (rounds 1-4: 8; the rest: host loop / status 1)
#define DPH_EXACT_HEAD 1024         // bytes in front of the fallback's hit buffers: counters | redo flags | tightened thresholds
#define DPH_EXACT_HITS (1u << 20)   // boundary hits per such row the fallback's buffer holds (as before; 32 x 2^20 x 16 B = 512 MiB of 288 GB:
                                    // an all-zero query ties with every row of a shard of up to a million rows and must still be served)

// IVF unit scan (dph_scan_units_kernel): a pass serves up to DPH_PASS_MAX query rows.  Every probed inverted list is cut
// into CHUNKS of at most 128 probing query rows ("slots": the 4 x 32 MFMA columns of a scan workgroup) and into
// SEGMENTS of DPH_UNIT_TILES contiguous tiles; a UNIT = (chunk, segment) is what a scan workgroup takes from the
// work queue: it multiplies the tiles of the segment with the gathered high digits of the chunk's query rows only.
#define DPH_PASS_MAX 1024           // query rows per pass of the unit scan (and the size of every per-pass array)
#define DPH_UNIT_WORDS (DPH_PASS_MAX / 32)   // probe-mask words per list in that pass
#define DPH_UNIT_TILES 256          // tiles per segment (6 MiB of the dump; 1024 measured no better: the tail grows)
#define DPH_UNIT_SLOTS 128
#define DPH_UNIT_LAUNCHES 8         // work-queue counters per pass (one per scan launch: ladder levels + the full scan)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// per query row scalars produced by the quantiser, consumed by the select/certify kernel (all float64)
struct dph_qinfo {
    double sc;        // q_j ~= sc * (128*q1_j + q2_j)
    double e_norm2;   // || e ||_2,  e_j = q_j - sc*(128*q1_j+q2_j)
    double e_mu;      // <e, mu>  (mu = the shard's per-dimension mean code)
    double q_sum;     // sum_j q_j
    double q_l1;      // sum_j |q_j|
};

// 64-bit candidate key: high word = score biased to unsigned, low word = ~local_row, so that unsigned
// descending order is (score desc, row asc).  0 is the "empty" sentinel.
__host__ __device__ static inline uint64_t dph_make_key(int32_t score, uint32_t row) {
    return ((uint64_t)((uint32_t)score ^ 0x80000000u) << 32) | (uint64_t)(0xFFFFFFFFu - row);
}
__host__ __device__ static inline int32_t dph_key_score(uint64_t key) {
    return (int32_t)((uint32_t)(key >> 32) ^ 0x80000000u);
}
__host__ __device__ static inline uint32_t dph_key_row(uint64_t key) {
    return 0xFFFFFFFFu - (uint32_t)key;
}

// id <-> stored-row translation of a shard built from several sub-indexes (the reference merges per-dump indexes whose
// ids are offset + local, offsets spaced max_idx apart: scripts/parallel/add_to_index.py:42-51, index.py:135-140):
// group g holds stored rows [row_starts[g], row_starts[g+1]) with ids id_offsets[g] + (row - row_starts[g]).
// n_groups = 0: ids are id_base + row.
struct dph_idmap {
    const int64_t* id_offsets; const int64_t* row_starts; int n_groups; int64_t id_base; int64_t n_ids;
};
__device__ __forceinline__ int64_t dph_id_of_row(const dph_idmap& m, int64_t row) {
    if (m.n_groups == 0) return m.id_base + row;
    int lo = 0, hi = m.n_groups - 1;                    // last group whose first row is <= row
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (m.row_starts[mid] <= row) lo = mid; else hi = mid - 1; }
    return m.id_offsets[lo] + (row - m.row_starts[lo]);
}
// stored row (flat / grouped shards) or local id (list-major shards: the caller maps it through inv_row), -1 = not here
__device__ __forceinline__ int64_t dph_local_of_id(const dph_idmap& m, int64_t id) {
    if (m.n_groups == 0) { const int64_t l = id - m.id_base; return (l < 0 || l >= m.n_ids) ? -1 : l; }
    if (id < m.id_offsets[0]) return -1;
    int lo = 0, hi = m.n_groups - 1;                    // last group whose offset is <= id
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (m.id_offsets[mid] <= id) lo = mid; else hi = mid - 1; }
    const int64_t r = id - m.id_offsets[lo];
    return r < m.row_starts[lo + 1] - m.row_starts[lo] ? m.row_starts[lo] + r : -1;
}

// device-side gate of the retry launches: a kernel whose `gate` pointer is non-NULL serves rows
// [gate_base, min(*gate, gate_base + rows of the pass)) and exits at once when that range is empty, so the retry chain
// can be enqueued without a host round trip (it costs a few empty launches when every row certified).
__device__ __forceinline__ int dph_gated_rows(const int* gate, int gate_base, int n_q) {
    if (!gate) return n_q;
    const int left = *gate - gate_base;
    return left < 0 ? 0 : (left < n_q ? left : n_q);
}

// Work queue of the flat / masked scan (dph_scan.hip): segment `u` of `n_tiles` visited tiles for a grid of `grid`
// workgroups -- guided self-scheduling in phases of 2*grid segments, every phase dealing out half of what is left, so
// the segment length halves from n_tiles/(4*grid) down to `seg_min`.  Returns the segment's first tile (>= n_tiles: the
// queue is empty, and so it is for every later u) and its length in *len.  Shared by the kernel and, through
// dph_debug_guided_segment, by the CPU tests.
__host__ __device__ static inline int64_t dph_guided_segment(int64_t u, int64_t n_tiles, int grid, int seg_min, int64_t* len) {
    const int64_t per_phase = 2 * (int64_t)grid;
    int64_t first = 0, l = seg_min;
    for (int64_t k = u / per_phase;; --k) {
        const int64_t left = n_tiles - first;
        l = left / (2 * per_phase);
        l = l > (int64_t)seg_min ? l : (int64_t)seg_min;
        if (k == 0 || left <= 0) break;
        first += per_phase * l;
    }
    first += (u % per_phase) * l;
    *len = (first < n_tiles && n_tiles - first < l) ? n_tiles - first : l;
    return first;
}

// everything a pass of the search pipeline shares (filled by dph_api.hip, consumed by the launchers)
struct dph_pass {
    // shard
    const int8_t* db; int64_t n_rows; int64_t n_tiles; int64_t id_base; const int64_t* row_ids; dph_idmap idmap;
    int grid;                           // scan workgroups (= CUs)
    // the pass
    int qb;                             // 1 or 2: 128*qb query rows
    int q0;                             // first query row of the pass inside the call
    int n_q;                            // rows of the pass (host view; the gate may shrink it)
    const int* gate; int gate_base;     // device-side row count (retry passes), or NULL
    // query images (whole call)
    const float* x; const int8_t* qfrag_hi; const int8_t* q1; const int8_t* q2; const dph_qinfo* qinfo; const int* lmax;
    // the aux k-step: the shard's aux rows (NULL = none), their layout, the query rows' aux digits [row of the call][DPH_AUX_SLOTS]
    const int8_t* aux; dph_aux_layout aux_lay; const int8_t* qaux;
    // IVF
    const unsigned* tilemask;           // [n_tiles][8] words or NULL
    // IVF unit scan (unit_recs != NULL): work queue + gathered query fragments, see DPH_PASS_MAX above
    const int4* unit_recs;              // {first tile of the list, first tile of the segment, end tile, chunk}
    const int4* unit_list_recs;         // the same per chunk for the whole list (what the ladder levels walk)
    const int* unit_counts;             // [0] chunks, [1] units (device side, written by dph_units_build_kernel)
    int* unit_next;                     // [DPH_UNIT_LAUNCHES] work-queue heads
    int unit_launch;                    // which head the next scan launch uses
    const int* slot_q;                  // [chunk][128] query row of the pass in that MFMA column, -1 = empty
    const int8_t* unit_frags;           // [chunk][4][DPH_QGROUP_FRAG_BYTES] high digits in fragment order
    const unsigned* listmask;           // [nlist][mask_words]
    const int32_t* tile_list;           // [n_tiles]
    int mask_words;
    // outlier rows of the shard (sorted stored-row indices): scored against every query row, never bounded
    const unsigned* outliers; int n_out;
    // scratch
    uint2* pairs; unsigned* chunk_fill;         // [DPH_POOL_CHUNKS][DPH_CHUNK_PAIRS] pair pool, pairs in every claimed chunk
    unsigned* wave_counts;                      // [grid*4][2] (pairs, emit-path triggers) of every scan wave: statistics
    uint64_t* buckets; unsigned* bucket_counts; // [DPH_PASS_MAX][DPH_BUCKET_CAP], [DPH_PASS_MAX]
    unsigned* overflow;                         // [DPH_PASS_MAX] row lost pairs (the pair pool ran dry)
    int* queue_head;                            // [0] work-queue head of the flat / masked scan, [1] chunks claimed from the
                                                // pair pool; zeroed (with the bucket counts and overflow flags) by every scan launch
    int seg_tiles;                              // shortest queue segment in tiles (full scans; sampled levels: fewer)
    int sched;                                  // hand-over schedule of the flat full scan (dph_scan.hip hand_pos): 0 lock step, 1 wave after wave, 2 interleaved
    // the full scan behind a FUSED finest ladder level of stride S: skip_m = ceil(2^29 / (S - 1)) makes the scan skip the tiles
    // that level visited (0 = visit every tile), accumulate = keep the buckets / flags that level's refine filled
    unsigned skip_m; bool accumulate;
};

// launchers (defined in the .hip files, called from dph_api.hip)
struct dph_index;
// mu: the shard's per-dimension mean codes [768] (device); lay / norm_unit: its aux layout (stride 0: lmax carries the shard-wide norm
// bound ||q2|| rmax + <q2, mu>; otherwise lmax = <q2, mu> and the norm part rides in the aux digits qaux [row][DPH_AUX_SLOTS])
void dph_launch_quantize(const float* x_dev, int64_t n_rows, const int* gate, int8_t* qfrag_hi, int8_t* q1, int8_t* q2,
                         dph_qinfo* qinfo_dev, double rmax, int* lmax_dev, int8_t* qaux, const int* mu_dev, const dph_aux_layout& lay,
                         int norm_unit, hipStream_t st, float* xc = nullptr, int32_t* nonfinite = nullptr);
void dph_launch_nonfinite_fix(const int32_t* flag, int64_t n, int k, float* D, int64_t* I, int32_t* status, double* bound, int* count,
                              hipStream_t st);
// per-dimension sums of the stored rows (row_ids != NULL: padding rows skipped): sums[0..767] = sum n_j, sums[768..1535] = sum n_j^2
void dph_launch_colstats(const int8_t* db, int64_t n_rows, const int64_t* row_ids, long long* sums, hipStream_t st);
// aux rows of a shard: [n_tiles * 32][lay.stride] (padding rows zero)
void dph_launch_aux_build(const int8_t* db, int64_t n_rows, int64_t n_rows_padded, const int64_t* row_ids, const int* mu_dev,
                          const dph_aux_layout& lay, int norm_unit, int8_t* aux, hipStream_t st);
// filter scan of every `tile_stride`-th tile (n_tiles_visit of them) under the per-row bounds tau (NULL = none: every
// row of the visited tiles is emitted -- cold start of the ladder / tiny shards)
void dph_launch_scan(const dph_pass& p, bool sample, int64_t n_tiles_visit, int tile_stride, const int* tau, int nset,
                     hipStream_t st);
// unit scan: every unit visits the tiles of its segment whose index inside the list is a multiple of `tile_stride`;
// `rowmask` (bit r = accumulator register r of a lane, 2 rows each) restricts what a visited tile may emit (cold level)
void dph_launch_scan_units(const dph_pass& p, bool sample, int tile_stride, unsigned rowmask, const int* tau, hipStream_t st);
void dph_clear_pass_counters(const dph_pass& p, hipStream_t st);     // issued by every scan launch
void dph_launch_refine(const dph_pass& p, hipStream_t st);
#define DPH_SAMPLE_KEEP 16          // scores per query row a rank shares for the union bound
void dph_launch_threshold(const dph_pass& p, int kp, const int* floor_tau, int* tau_out, int* top_out, hipStream_t st);
void dph_launch_union_bounds(const int* top_parts, int n_parts, int64_t n, int* tau_out, hipStream_t st);
int  dph_scan_grid(int device);
struct dph_select_args {
    const float* lut; int k; int C; double rmax; double rmax_all; double delta_max; float offset; float scale;
    const int* tau;                     // bound the pass was scanned under (NULL = none)
    const int* rowmap;                  // retry passes: slot -> row of the call (outputs go to rowmap[slot]); NULL = identity
    float* D; int64_t* I; int32_t* status; double* bound_out; int32_t* ik_out; int32_t* fail_out;
    const int32_t* only_failed;         // != NULL: a second look at the same buckets -- only rows whose flag is set (and whose pairs are all there)
    int* reselect_count;                // ... counting the rows it certifies
};
void dph_launch_select(const dph_pass& p, const dph_select_args& a, hipStream_t st);
// listmask[nlist][mask_words]; tilemask (8 words per tile, mask_words == 8 only) may be NULL.  cs_slot: the caller's (one per index
// handle) slot for the candidate scratch of the one-pass probe selection -- allocated here on first use, hipFree'd by the owner;
// NULL = no scratch, long score rows take the full three-pass selection
void dph_launch_coarse(const float* x_dev, int q0, int n_q, const int* gate, int gate_base, const float* centroids, int nlist,
                       int nprobe, double cnorm_max, float* scores, unsigned* listmask, int mask_words, const int32_t* tile_list,
                       int64_t n_tiles, unsigned* tilemask, void** cs_slot, hipStream_t st);
void dph_launch_coarse_lists(const float* x_dev, int q0, int n_q, const int* gate, int gate_base, const float* centroids, int nlist,
                             int nprobe, double cnorm_max, float* scores, unsigned* listmask, int mask_words,
                             const int32_t* tile_list, int64_t n_tiles, unsigned* tilemask, int* probe_out, int probe_stride,
                             void** cs_slot, hipStream_t st);
void dph_launch_coarse_presplit(const float* x_dev, int q0, int n_q, const int* gate, int gate_base, const float* centroids, int nlist,
                                int nprobe, double cnorm_max, float* scores, unsigned* listmask, int mask_words,
                                const int32_t* tile_list, int64_t n_tiles, unsigned* tilemask, int* probe_out, int probe_stride,
                                const unsigned* c_pk, const unsigned* x_pk, void** cs_slot, hipStream_t st, bool clear_mask = true,
                                unsigned* row_fail = nullptr);
// The int8 scans multiply with v_mfma_i32_16x16x64_i8 (dph_scan.hip, round 6), and the high-digit query fragments are written in
// that instruction's operand order: entry [2 s + h][lane] (16 bytes) of a group of 32 query rows = bytes 64 s + 16 (lane >> 4) .. + 15
// of query row 16 h + (lane & 15).  (DPH_SCAN_X16 = 0 builds the 32 x 32 x 32 kernels of rounds 1-5 for A/B timing: entry [ks][lane] =
// bytes 32 ks + 16 (lane >> 5) .. + 15 of query row lane & 31.)  Both are [24][64][16 B] per group.
#ifndef DPH_SCAN_X16
#define DPH_SCAN_X16 1
#endif
__host__ __device__ static inline bool dph_frag_x16(int aux_stride) { (void)aux_stride; return DPH_SCAN_X16 != 0; }
// hands out a fresh (start, stop) event pair per bracketed launch (profiling; next == NULL: none)
struct dph_event_source { void* ctx; void (*next)(void* ctx, hipEvent_t* a, hipEvent_t* b); };
// long quantizers, the filter form: one-product bf16 GEMM with the threshold test in its epilogue (dph_ivf.hip)
void dph_launch_coarse_filter(const float* x_dev, int n_q, const float* centroids, const unsigned short* c_hi, const unsigned short* x_hi,
                              const unsigned* c_pk, const unsigned* x_pk, int nlist, int nprobe, double cnorm_max, float* scores,
                              unsigned* listmask, int mask_words, int* probe_out, int probe_stride, void** cs_slot, void** cf_slot,
                              hipStream_t st, dph_event_source prof = dph_event_source{nullptr, nullptr}, unsigned* row_fail = nullptr,
                              int variant = 1 /* 2: the centroid stream loaded non-temporal, 3: centroids straight into registers from c_frag,
                                                 5: the filter scan over c_pieces (<= 128 query rows; more: variant 3) */,
                              const unsigned short* c_frag = nullptr, const unsigned short* c_pieces = nullptr,
                              const float* cnorm = nullptr /* [nlist] ||c_l||_2 (device) */, double cnorm_cap = 0.0 /* lists longer than this get exact scores */,
                              int coarse_teams = 1 /* variant 5, more than 128 rows: one launch of workgroup teams sharing tiles through L2 (dph_scan.hip MODE 4) */);
void dph_launch_coarse_scan_teams(const void* img, int64_t n_lists, const void* qfrag, int n_q, int n_groups, const unsigned* est_keys, uint2* pairs,
                                  unsigned* chunk_fill, unsigned* wave_counts, int* counters, int grid, hipStream_t st);
// variant 5: the coarse quantizer as a filter SCAN (dph_scan.hip MODE 3) over the piece-major bf16 image
// [tile of 32 lists][half of k][32 rows][384 bf16] (dph_launch_bf16_pieces; dph_bf16_piece_rows(n) rows allocated); hits land in
// chunks of the pair pool as (list | query row << 20, score key)
void dph_launch_bf16_pieces(const float* v, int64_t n_rows, unsigned short* out, hipStream_t st);
int64_t dph_bf16_piece_rows(int64_t n_rows);
void dph_launch_coarse_scan(const void* img, int64_t n_lists, const void* qfrag, int n_q, const unsigned* est_keys, uint2* pairs,
                            unsigned* chunk_fill, unsigned* wave_counts, int* counters, int grid, hipStream_t st);
// the fragment-major bf16 image variant 3 streams (dph_bf16_frag_rows(n) rows allocated)
void dph_launch_bf16_frag(const float* v, int64_t n_rows, unsigned short* out, hipStream_t st);
int64_t dph_bf16_frag_rows(int64_t n_rows);
int dph_coarse_filter_debug(void* cf_slot, unsigned out[2]);
long long dph_coarse_filter_debug_pool(void* cf_slot, unsigned* lk_host, unsigned short* q_host, long long cap);
// bf16 image of a [n_rows, 768] fp32 matrix; tiled = 1: the tile-major layout the filter GEMM streams (dph_bf16_hi_rows(n, 1) rows allocated)
void dph_launch_bf16_hi(const float* v, int64_t n_rows, int tiled, unsigned short* hi, hipStream_t st);
int64_t dph_bf16_hi_rows(int64_t n_rows, int tiled);
void dph_launch_bf16_split(const float* v, int64_t n_elems, unsigned* packed, hipStream_t st);
// work queue of a unit-scan pass from the probe masks: chunks, slot tables, unit records, gathered fragments
void dph_launch_units_build(const unsigned* listmask, int nlist, const int* list_tile0, const int8_t* q1, int q0,
                            int chunk_cap, int unit_cap, int* unit_counts, int* unit_next, int* slot_q, int4* unit_recs,
                            int4* unit_list_recs, int8_t* unit_frags, int2* unit_offsets, int spread, hipStream_t st, bool x16);
// list assignment (arg-max over the centroids, fused: no score matrix); rows = fp32 [n,768] or int8 rows + the shard's LUT
void dph_launch_assign(const void* rows, bool rows_int8, const float* lut, int64_t n, const float* centroids, int nlist,
                       const float* bias, int32_t* best, float* gap, hipStream_t st);
// k-means centroid update over int8 rows (dph_kmeans.hip): sums [nlist,768] int64 and counts [nlist] are scratch + output
void dph_launch_kmeans_update(const int8_t* rows, const int32_t* assign, int64_t m, int nlist, float offset, float scale,
                              int spherical, long long* sums, unsigned* counts, float* centroids, hipStream_t st);
void dph_launch_gather_sample(const int8_t* db, int64_t n_rows, const int64_t* idx, int64_t m, int8_t* out, hipStream_t st);
// retry plumbing: compact the failing rows of a call (fail flags -> rows[], *count), gather their query vectors
void dph_launch_compact_failing(const int32_t* fail, int64_t n, int match, const float* x, int32_t* rows_out, int* count_out,
                                float* x_out, int max_rows, hipStream_t st);
void dph_launch_gather_rows(const float* x, const int32_t* rows, const int* count, float* x_out, int max_rows, hipStream_t st);
void dph_launch_retry_tau(const int* gate, int64_t n_max, const int32_t* rows, const int32_t* ik, const dph_qinfo* qinfo,
                          double rmax, double delta_max, float scale, int* tau_out, hipStream_t st);

void dph_launch_exact(const int8_t* db, int64_t n_rows, dph_idmap idmap, const float* x_dev, const float* lut_dev,
                      const int32_t* rows_dev, const int* n_fail_dev, int n_fail_max, int k, const int64_t* row_ids,
                      const unsigned* tilemask, float* D, int64_t* I, int32_t* status, void* scratch,
                      size_t scratch_bytes, hipStream_t st, int tighten_rounds = 1);
// device-side list builder (dph_build.hip): rows sorted by (list, row) + the first sorted position of every list
int dph_list_major_sort(const int32_t* assign_dev, int64_t n, int nlist, uint64_t** keys_out, int64_t* starts_host, hipStream_t st);
void dph_launch_list_major_gather(const int8_t* src, const uint64_t* keys, int64_t n, const int64_t* src_start_dev,
                                  const int64_t* dst_start_dev, int64_t id_base, int8_t* dst, int64_t* row_ids,
                                  int32_t* inv_row, hipStream_t st);
void dph_launch_fill(int8_t* db, int64_t n_rows, int64_t id_base, uint64_t seed, int kind, hipStream_t st);
#define DPH_NORM_BINS 8192          // histogram of squared centred row norms: bins of DPH_NORM_BIN_W (max 768*168^2 < 2^25)
#define DPH_NORM_BIN_W 4096u
#define DPH_OUTLIER_MAX 1024        // rows per shard that may be treated as outliers (always scored, never bounded)
void dph_launch_rownorm(const int8_t* db, int64_t n_rows, const int64_t* row_ids, const int* mu_dev, unsigned long long* max_out,
                        unsigned* hist, unsigned long long cut2, unsigned* out_rows, unsigned* out_count, unsigned out_cap,
                        hipStream_t st);
void dph_launch_window(int direction, const int8_t* db, int64_t n_rows, dph_idmap idmap, const float* lut_dev,
                       const float* qhalf, int64_t n_cand, int k, int L, const int64_t* ids, const int32_t* doc,
                       const int32_t* word, const float* first, const int32_t* row2doc, const int32_t* row2word,
                       const int32_t* doc_ids, int64_t n_docs, const int64_t* f2o_off, const int32_t* f2o,
                       const int32_t* inv_row, int32_t* pred_word, double* best, int32_t* argslot, float* vecs,
                       hipStream_t st);
void dph_launch_merge(const float* D_parts, const int64_t* I_parts, const double* best_parts, const int32_t* pred_parts,
                      const int32_t* status_parts, const double* bound_parts, int n_parts, int64_t stride_bytes, int64_t n,
                      int k, float* D_out, int64_t* I_out, int32_t* src_out, double* best_out, int32_t* pred_out,
                      int32_t* status_out, hipStream_t st);
void dph_launch_score_vecs(const float* q, const float* vecs, int64_t n_b, int64_t m, float* out, hipStream_t st);
void dph_launch_score_vecs_bwd(const float* grad, const float* vecs, int64_t n_b, int64_t m, float* grad_q, hipStream_t st);
void dph_launch_dense_logits(const float* s, const float* e, int64_t n_b, int64_t T, float* out, hipStream_t st);

// ---- the reference's own index type, IndexPreTransform(OPQMatrix) -> IndexIVFPQ, resident in HBM (dph_pq.hip)
struct dph_pq;
const char* dph_pq_error();
int dph_pq_alloc(dph_pq** out, int device, int64_t ntotal, int nlist, int M);
void dph_pq_free(dph_pq* p);
int dph_pq_set_params(dph_pq* p, const float* A, const float* b, const float* centroids, const float* pq_centroids, int by_residual);
int dph_pq_set_list_sizes(dph_pq* p, const int64_t* sizes);
int dph_pq_upload(dph_pq* p, int64_t pos0, int64_t n, const uint8_t* codes, const int64_t* ids);
int dph_pq_finalize(dph_pq* p, hipStream_t st);
bool dph_pq_ready(const dph_pq* p);
int64_t dph_pq_ntotal(const dph_pq* p);
int dph_pq_nlist(const dph_pq* p);
const float* dph_pq_A_host(const dph_pq* p);
void dph_pq_set_coarse_teams(dph_pq* p, int on);               // tuning key "coarse_teams"
void dph_pq_set_coarse_filter(dph_pq* p, int on);              // tuning key "coarse_filter"
void dph_pq_set_split_lut(dph_pq* p, int on);                  // tuning key "pq_split_lut"
// measurement hook: HIP events around the coarse quantizer's dominant GEMM launch of every pass (dph_profile_enable / _read on a PQ index)
int dph_pq_coarse_debug(dph_pq* p, unsigned out[2]);
int dph_pq_coarse_debug_pool(dph_pq* p, unsigned* lk_host, unsigned short* q_host, long long cap, long long* count);
int dph_pq_debug_phases(dph_pq* p, int which, unsigned long long* out, int cap);
int dph_pq_debug_pass(dph_pq* p, int* info, unsigned* per_row, int n);
int dph_coarse_select_clock(unsigned long long* out, int cap_rows);
int dph_pq_profile(dph_pq* p, int on);
int dph_pq_profile_read(dph_pq* p, double* ms_total, int* launches);
int dph_pq_profile_read_each(dph_pq* p, double* ms_out, int cap, int* n_out);
int dph_pq_search_dev(dph_pq* p, const float* x_dev, int64_t n, int k, int nprobe, float* D, int64_t* I, int32_t* status, hipStream_t st);
int dph_pq_reconstruct_dev(dph_pq* p, const int64_t* ids_dev, int64_t n, float* out_dev, int32_t* found_dev, hipStream_t st);
int dph_pq_window(dph_pq* p, int direction, dph_idmap idmap, const float* qhalf, int64_t n_q, int k, int L, const int64_t* ids,
                  const int32_t* doc, const int32_t* word, const float* first, const int32_t* row2doc, const int32_t* row2word,
                  const int32_t* doc_ids, int64_t n_docs, const int64_t* f2o_off, const int32_t* f2o, int32_t* pred_word,
                  double* best, int32_t* argslot, float* vecs, hipStream_t st);
