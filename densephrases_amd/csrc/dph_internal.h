// dph_internal.h -- shared constants / device structs of libdph (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dph.h"

// ---------------------------------------------------------------- geometry of the scan
// One scan pass serves DPH_QROWS query rows (= 2B at the reference's eval batch of 64: index.py:196-197).
// A workgroup is 4 waves, one per SIMD; wave w owns query rows [32w, 32w+32) for the whole launch and keeps
// their two int8 digits in registers (2 digits x 24 k-steps x 4 VGPR = 192 registers).  Database rows stream
// HBM -> LDS (LDS-DMA, 16 B per lane) in tiles of 32 rows = 24 KiB and every wave reads every tile.
#define DPH_KSTEPS 24               // 768 / 32 int8 per v_mfma_i32_32x32x32_i8
#define DPH_QROWS 128               // query rows per scan pass
#define DPH_TILE_ROWS 32
#define DPH_TILE_BYTES (DPH_TILE_ROWS * DPH_DIM)     // 24576
#define DPH_SCAN_THREADS 256
#define DPH_QFRAG_BYTES (2 * 4 * DPH_KSTEPS * 64 * 16)   // 196608: [digit][wave][kstep][lane][16]
#define DPH_CENTER 40               // c of the centred norm bound: n - c, c = (0 - offset) * scale at defaults

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// per query row scalars produced by the quantiser, consumed by the select/certify kernel (all float64)
struct dph_qinfo {
    double sc;        // q_j ~= sc * (128*q1_j + q2_j)
    double e_norm2;   // || e ||_2,  e_j = q_j - sc*(128*q1_j+q2_j)
    double e_sum;     // sum_j e_j
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

// launchers (defined in the .hip files, called from dph_api.hip)
struct dph_index;
void dph_launch_quantize(const float* x_dev, int64_t n_rows, int8_t* qfrag_dev, dph_qinfo* qinfo_dev, double rmax,
                         int* lmax_dev, hipStream_t st);
void dph_launch_scan(int kp, bool sample, const int8_t* db, int64_t n_rows, int64_t n_tiles, int tile_stride,
                     const int8_t* qfrag, const int* tau_init, const int* lmax_q, const unsigned* tilemask,
                     const int64_t* row_ids, uint64_t* lists, int grid, hipStream_t st);
void dph_launch_coarse(const float* x_dev, int q0, int n_q, const float* centroids, int nlist, int nprobe,
                       unsigned* listmask, const int32_t* tile_list, int64_t n_tiles, unsigned* tilemask, hipStream_t st);
#define DPH_THRESHOLD_MAX_KEYS(KP) (512 * (KP))       // lists of up to 256 scan workgroups x 2 lanes per query row
#define DPH_SAMPLE_KEEP 16          // scores per query row a rank shares for the union bound (= KP of the first attempt)
int dph_launch_threshold(int kp, const uint64_t* lists, int grid, const int* floor_tau, int* tau_out, int n_q, int* top_out,
                         hipStream_t st);
void dph_launch_union_bounds(const int* top_parts, int n_parts, int64_t n, int* tau_out, hipStream_t st);
#define DPH_SAMPLE_STRIDE 32        // the threshold pre-pass scans every 32nd tile (3.1 % of the shard)
int  dph_scan_grid(int device);
void dph_launch_select(int kp, int grid, const uint64_t* lists, const int8_t* db, int64_t n_rows,
                       int64_t id_base, const float* x_dev, const dph_qinfo* qinfo, const float* lut_dev,
                       int q0, int n_q, int k, double rmax, double delta_max, float offset, float scale,
                       const int* tau_init, const int64_t* row_ids, float* D, int64_t* I, int32_t* status,
                       double* bound_out, hipStream_t st);
void dph_launch_exact(const int8_t* db, int64_t n_rows, int64_t id_base, const float* x_dev, const float* lut_dev,
                      const int32_t* rows_dev, int n_fail, int k, const int64_t* row_ids, const unsigned* tilemask,
                      float* D, int64_t* I, int32_t* status, void* scratch, size_t scratch_bytes, hipStream_t st);
void dph_launch_fill(int8_t* db, int64_t n_rows, int64_t id_base, uint64_t seed, hipStream_t st);
void dph_launch_rownorm(const int8_t* db, int64_t n_rows, const int64_t* row_ids, unsigned long long* max_out,
                        hipStream_t st);
void dph_launch_window(int direction, const int8_t* db, int64_t n_rows, int64_t id_base, const float* lut_dev,
                       const float* qhalf, int64_t n_cand, int k, int L, const int64_t* ids, const int32_t* doc,
                       const int32_t* word, const float* first, const int32_t* row2doc, const int32_t* row2word,
                       const int32_t* doc_ids, int64_t n_docs, const int64_t* f2o_off, const int32_t* f2o,
                       const int32_t* inv_row, int64_t n_ids, int32_t* pred_word, double* best, int32_t* argslot,
                       float* vecs, hipStream_t st);
void dph_launch_merge(const float* D_parts, const int64_t* I_parts, const double* best_parts, const int32_t* pred_parts,
                      const int32_t* status_parts, const double* bound_parts, int n_parts, int64_t stride_bytes, int64_t n,
                      int k, float* D_out, int64_t* I_out, int32_t* src_out, double* best_out, int32_t* pred_out,
                      int32_t* status_out, hipStream_t st);
