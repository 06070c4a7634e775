// dph_quant.hip -- query quantiser and shard utilities (synthetic fills, centred row norms, outlier rows).
#include "dph_internal.h"

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ------------------------------------------------------------------------------------------ quantiser
// One workgroup per (padded) query row r of the call.  Writes
//   qfrag_hi : the HIGH int8 digit in the register-fragment order of the scan, [r/32][24][lane][16 B]
//              (aux shards, 32 x 32 x 32 MFMAs: element j of row r at entry j>>5, lane 32*((j>>4)&1) + r%32, byte j&15; shards without
//              aux rows, 16 x 16 x 64 MFMAs: entry 2*(j>>6) + (r%32)/16, lane 16*((j>>4)&3) + r%16, byte j&15 -- dph_internal.h),
//   q1, q2   : both digits row-major [r][768] (what dph_refine_kernel dots against a database row),
//   qaux     : the aux digits [r][DPH_AUX_SLOTS] (dph_internal.h dph_aux_layout; zero where the layout has no slot),
//   qinfo    : the row's fp64 scalars for the certificate, lmax: what the scan subtracts from the bound.
// Digits: q_j = sc * (128 * Q1_j + q2_j) + e_j with Q1_j = q1_j + (the replica digits of j), |q1| <= 127, every replica digit
// |.| <= 127, |q2| <= q2max.  A dimension with R replica slots (a ROGUE dimension of the shard: its raw code is a slot of every aux
// row) has the range 127 (1 + R), so the scale is set by max_j |q_j| / (1 + R_j): the bulk of the dimensions keeps its resolution when
// a few dimensions are several times larger than the rest -- the shape BERT-family vectors have.
// Reference: the query is the fp32 cast of index.py:195; the digits are an internal representation.
__global__ __launch_bounds__(256) void dph_quantize_kernel(const float* __restrict__ x, int64_t n, const int* __restrict__ gate,
                                                           int8_t* __restrict__ qfrag_hi, int8_t* __restrict__ q1o,
                                                           int8_t* __restrict__ q2o, dph_qinfo* __restrict__ qinfo,
                                                           double rmax, int* __restrict__ lmax_out, int8_t* __restrict__ qaux,
                                                           const int* __restrict__ mu, dph_aux_layout lay, int norm_unit,
                                                           float* __restrict__ xc, int32_t* __restrict__ nonfinite) {
    __shared__ double red[6][4];
    __shared__ double redf[4];
    const int r = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int64_t n_live = gate ? (int64_t)*gate : n;
    if (gate && (int64_t)(r / DPH_QGROUP) * DPH_QGROUP >= n_live) return;      // whole group unused by a gated retry
    float v[3];
    int reps[3];                           // replica slots of this thread's dimensions
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        v[i] = (r < n_live) ? x[(int64_t)r * DPH_DIM + t + 256 * i] : 0.f;
        reps[i] = 0;
        for (int s = 0; s < lay.n_rep; ++s) reps[i] += (lay.rep_dim[s] == t + 256 * i) ? 1 : 0;
    }
    // NON-FINITE rows (a NaN / Inf element: fp16 / bf16 encoders overflow in practice).  FAISS' flat search answers such a row with
    // -1 / -FLT_MAX (no score compares greater than the heap's threshold) and the other rows as ever (index.py:195-200 passes whatever
    // the encoder gave).  Here: the row is flagged, searched as a stand-in pattern -- bounded, neither zero nor aligned with anything, so it
    // costs what any row costs and certifies like any row (a NaN in `e2` would fail every certificate and walk the whole retry chain
    // into fp64 full scans) -- and dph_nonfinite_fix_kernel overwrites its result with ids -1, scores -FLT_MAX, status
    // DPH_ROW_NONFINITE at the end of the search.  xc = the rows as every later stage reads them (stand-in included).
    {
        const bool bad_here = !(isfinite(v[0]) && isfinite(v[1]) && isfinite(v[2]));
        const bool bad = __syncthreads_or(bad_here ? 1 : 0) != 0;
        if (bad) {
#pragma unroll
            for (int i = 0; i < 3; ++i) v[i] = (float)(((t + 256 * i) * 37) % 64) * (1.0f / 64.0f) - 0.4921875f;
        }
        if (nonfinite && t == 0 && r < n_live) nonfinite[r] = bad ? 1 : 0;
        if (xc && r < n_live) {
#pragma unroll
            for (int i = 0; i < 3; ++i) xc[(int64_t)r * DPH_DIM + t + 256 * i] = v[i];
        }
    }
    // (in double: without replicas this is exactly max |q_j|, the scale of rounds 1-4)
    double am = fmax(fabs((double)v[0]) / (double)(1 + reps[0]), fmax(fabs((double)v[1]) / (double)(1 + reps[1]), fabs((double)v[2]) / (double)(1 + reps[2])));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = fmax(am, __shfl_xor(am, o));
    if (lane == 0) redf[w] = am;
    if (qaux && t < DPH_AUX_SLOTS) qaux[(int64_t)r * DPH_AUX_SLOTS + t] = 0;
    __syncthreads();
    am = fmax(fmax(redf[0], redf[1]), fmax(redf[2], redf[3]));
    const double s = am > 0.0 ? am / 127.0 : 1.0;
    const double sc = s / 128.0;
    const double q2lim = (double)lay.q2max;
    double e2 = 0, em = 0, qs = 0, ql1 = 0, q2m = 0, q2n = 0;
    const int g = r / DPH_QGROUP, col = r % DPH_QGROUP;
    int8_t* base = qfrag_hi + (int64_t)g * DPH_QGROUP_FRAG_BYTES;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = t + 256 * i;
        const double cap = 127.0 * (double)(1 + reps[i]);
        const double u = (double)v[i] / s;
        double Q1 = rint(u);
        Q1 = fmin(cap, fmax(-cap, Q1));
        double q2 = rint((u - Q1) * 128.0);
        q2 = fmin(q2lim, fmax(-q2lim, q2));
        const double e = (double)v[i] - sc * (128.0 * Q1 + q2);
        const double q1 = fmin(127.0, fmax(-127.0, Q1));
        const double m = (double)mu[j];
        e2 += e * e; em += e * m; qs += (double)v[i]; ql1 += fabs((double)v[i]);
        q2m += q2 * m; q2n += q2 * q2;
        if (dph_frag_x16(lay.stride)) {            // the 16 x 16 x 64 operand order (dph_internal.h)
            base[((int64_t)((2 * (j >> 6) + (col >> 4)) * 64 + ((j >> 4) & 3) * 16 + (col & 15))) * 16 + (j & 15)] = (int8_t)(int)q1;
        } else {
            const int ks = j >> 5, half = (j >> 4) & 1, byte = j & 15;
            base[((int64_t)(ks * 64 + half * 32 + col)) * 16 + byte] = (int8_t)(int)q1;
        }
        q1o[(int64_t)r * DPH_DIM + j] = (int8_t)(int)q1;
        q2o[(int64_t)r * DPH_DIM + j] = (int8_t)(int)q2;
        if (reps[i] > 0) {                 // the part of Q1 beyond the first digit goes to the dimension's replica slots, 127 at a time
            int left = (int)(Q1 - q1);
            for (int sl = 0; sl < lay.n_rep; ++sl) {
                if (lay.rep_dim[sl] != j) continue;
                const int c = left > 127 ? 127 : (left < -127 ? -127 : left);
                qaux[(int64_t)r * DPH_AUX_SLOTS + lay.n_norm + sl] = (int8_t)c;
                left -= c;
            }
        }
    }
    e2 = wave_sum_f64(e2); em = wave_sum_f64(em); qs = wave_sum_f64(qs); ql1 = wave_sum_f64(ql1);
    q2m = wave_sum_f64(q2m); q2n = wave_sum_f64(q2n);
    if (lane == 0) { red[0][w] = e2; red[1][w] = em; red[2][w] = qs; red[3][w] = ql1; red[4][w] = q2m; red[5][w] = q2n; }
    __syncthreads();
    if (t == 0) {
        dph_qinfo qi;
        qi.sc = sc;
        qi.e_norm2 = sqrt(red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        qi.e_mu = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        qi.q_sum = red[2][0] + red[2][1] + red[2][2] + red[2][3];
        qi.q_l1 = red[3][0] + red[3][1] + red[3][2] + red[3][3];
        qinfo[r] = qi;
        // the low-digit term of a row: L = <q2, n> = <q2, n - mu> + <q2, mu> <= ||q2||_2 * ||n - mu||_2 + <q2, mu>   (<q2, mu>: an exact
        // integer, |.| < 2^23).  Without aux rows the norm is the shard's rmax and the whole bound is lmax; with them the scan
        // adds (norm code of the row) x (digit below) >= ||q2|| ||n - mu|| / 128 to the row's high-digit score itself and lmax
        // is <q2, mu> alone.
        const double q2mu = red[4][0] + red[4][1] + red[4][2] + red[4][3];
        const double q2nrm = sqrt(red[5][0] + red[5][1] + red[5][2] + red[5][3]);
        double lm;
        if (lay.stride > 0) {
            lm = q2mu;
            int bq = (int)ceil((double)norm_unit * q2nrm / 128.0 * (1.0 + 1e-9));
            bq = bq > 127 ? 127 : bq;      // (never: q2max is chosen so that norm_unit * q2max * sqrt(768) / 128 <= 126)
            for (int sl = 0; sl < lay.n_norm; ++sl) qaux[(int64_t)r * DPH_AUX_SLOTS + sl] = (int8_t)bq;
        } else {
            lm = ceil(q2nrm * rmax * (1.0 + 1e-9) + q2mu) + 1.0;
        }
        lmax_out[r] = lm > 1.0e9 ? 1000000000 : (lm < -1.0e9 ? -1000000000 : (int)lm);
    }
}

void dph_launch_quantize(const float* x_dev, int64_t n_rows, const int* gate, int8_t* qfrag_hi, int8_t* q1, int8_t* q2,
                         dph_qinfo* qinfo_dev, double rmax, int* lmax_dev, int8_t* qaux, const int* mu_dev, const dph_aux_layout& lay,
                         int norm_unit, hipStream_t st, float* xc, int32_t* nonfinite) {
    const int64_t padded = (n_rows + DPH_QROWS - 1) / DPH_QROWS * DPH_QROWS;
    if (padded <= 0) return;
    hipLaunchKernelGGL(dph_quantize_kernel, dim3((unsigned)padded), dim3(256), 0, st, x_dev, n_rows, gate, qfrag_hi, q1,
                       q2, qinfo_dev, rmax, lmax_dev, qaux, mu_dev, lay, norm_unit, xc, nonfinite);
}

// the result of a flagged row (dph_quantize_kernel): ids -1, scores -FLT_MAX, status DPH_ROW_NONFINITE, no claim on unseen rows
__global__ __launch_bounds__(256) void dph_nonfinite_fix_kernel(const int32_t* __restrict__ flag, int64_t n, int k, float* __restrict__ D,
                                                                int64_t* __restrict__ I, int32_t* __restrict__ status,
                                                                double* __restrict__ bound, int* __restrict__ count) {
    const int64_t r = blockIdx.x;
    if (r >= n || !flag[r]) return;
    for (int j = threadIdx.x; j < k; j += 256) { D[r * k + j] = -3.402823466e38f; I[r * k + j] = -1; }
    if (threadIdx.x == 0) {
        status[r] = DPH_ROW_NONFINITE;
        if (bound) bound[r] = -1.7976931348623157e308;
        if (count) atomicAdd(count, 1);
    }
}
void dph_launch_nonfinite_fix(const int32_t* flag, int64_t n, int k, float* D, int64_t* I, int32_t* status, double* bound, int* count,
                              hipStream_t st) {
    if (n > 0 && flag) hipLaunchKernelGGL(dph_nonfinite_fix_kernel, dim3((unsigned)n), dim3(256), 0, st, flag, n, k, D, I, status, bound, count);
}

// ------------------------------------------------------------------------------------------ synthetic fills
// kind 0 -- BASELINE.md config 2: rows i.i.d. float_to_int8(N(0, 0.6^2), -2, 20) ~ 40 + 12 z.
// kind 1 -- SURVEY 8(d) config 4 data: a mixture of 4096 Gaussians (sigma_between 0.5, sigma_within 0.25:
//           n = 40 + 10 z_c(j) + 5 z_r(j), cluster c = hash(row) mod 4096) in which every row with
//           hash(row) mod 999983 == 0 is a SATURATED outlier (bytes +127 / -128 by a per-row sign pattern): the shape
//           real phrase dumps have (dense neighbourhoods, a few extreme rows) and the i.i.d. dump does not.
// kind 3 -- kind 1 without the saturated rows: exactly SURVEY 8(d)'s config-4 data.  (An inner-product IVF cannot find a
//           saturated row through the lists a query probes -- its norm, not its direction, makes it a top hit -- so recall
//           against the exact search is measured on this kind; kind 1 stays the certificate's stress test.)
// kind 2 -- a DOCUMENT-ORDERED dump: the rows come in runs of 56..200 consecutive near-duplicates (the tokens of one
//           paragraph: n = 40 + 11.5 z_run(j) + 3.4 z_row(j), cosine ~0.92 inside a run; runs = the two parts of every
//           block of 256 rows, split at 56 + hash(block) mod 145).  What a real dump looks like to a scan that walks it in
//           id order: a query that likes one row of a run likes all of it, so the hits come in bursts.
// kind 4 -- an ANISOTROPIC dump, the shape BERT-family phrase vectors have (embed_utils.py:141-149 clips them into int8):
//           kind 2's document-ordered runs of near-duplicates, every row scaled by a log-normal factor (sigma ~0.35: a
//           per-run factor times a per-row factor, 2^(e/16) with e the sum of two Irwin-Hall draws; ~2 % of the rows at
//           twice the median norm, up to 3.5 x), and five "rogue" dimensions (DPH_ROGUE_DIMS) whose code sits near
//           -105 / +110 / -95 / +100 / -110 for EVERY row (+- 6 x the row's factor, saturating at the int8 range).
// Integer-only generators (Irwin-Hall sum of 4 hashed bytes) so that densephrases_amd/synth.py reproduces them
// bit-for-bit on the host.
__device__ __forceinline__ unsigned dph_hash32(unsigned lo, unsigned hi, unsigned seed) {
    unsigned h = lo * 0x9E3779B1u ^ (hi * 0x85EBCA77u + seed);
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ int dph_ih4(unsigned h) {       // sum of the 4 bytes - 510: ~ 147.8 * N(0,1)
    return (int)(h & 255u) + (int)((h >> 8) & 255u) + (int)((h >> 16) & 255u) + (int)(h >> 24) - 510;
}
// 4096 * 2^(i/16): the fractional part of kind 4's row factor
__device__ __constant__ int dph_pow2_16th[16] = {4096, 4277, 4467, 4664, 4871, 5087, 5312, 5547, 5793, 6049, 6317, 6597, 6889, 7194, 7512, 7845};
__device__ __forceinline__ int dph_rogue_slot(unsigned j) {     // DPH_ROGUE_DIMS of the kind-4 dump
    return j == 77u ? 0 : (j == 138u ? 1 : (j == 381u ? 2 : (j == 588u ? 3 : (j == 729u ? 4 : -1))));
}
template <int KIND>
__global__ __launch_bounds__(256) void dph_fill_kernel(int8_t* __restrict__ db, int64_t n_bytes, int64_t byte_base,
                                                       unsigned seed_lo, unsigned seed_hi) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 16;
    for (int64_t o = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; o < n_bytes; o += stride) {
        unsigned w[4];
        const uint64_t e0 = (uint64_t)(byte_base + o);
        const uint64_t row = e0 / DPH_DIM;                 // 16 | 768: the 16 bytes are of one row
        const unsigned j0 = (unsigned)(e0 % DPH_DIM);
        unsigned hr = 0, cluster = 0;
        bool outlier = false;
        if (KIND == 1 || KIND == 3) {
            hr = dph_hash32((unsigned)row, (unsigned)(row >> 32) ^ 0x5bd1e995u, seed_lo ^ seed_hi);
            cluster = hr & 4095u;
            outlier = KIND == 1 && (hr % 999983u) == 0u;
        }
        int s12 = 4096;                                    // kind 4: the row's factor, 12 fractional bits
        if (KIND == 2 || KIND == 4) {
            const uint64_t block = row >> 8;
            const unsigned split = 56u + dph_hash32((unsigned)block, (unsigned)(block >> 32) ^ 0x2545F491u, seed_lo ^ seed_hi) % 145u;
            cluster = (unsigned)(2u * (unsigned)block + (((unsigned)row & 255u) >= split ? 1u : 0u));     // the run
            if (KIND == 4) {
                const int er = (dph_ih4(dph_hash32(cluster, 0xA5u + (cluster >> 20), seed_lo + 0x7A11u)) * 2560 + 32768) >> 16;
                const int ew = (dph_ih4(dph_hash32((unsigned)row, (unsigned)(row >> 32) ^ 0x1B873593u, seed_lo ^ seed_hi ^ 0x3C6Eu)) * 2560 + 32768) >> 16;
                const int e16 = er + ew, ip = e16 >> 4;
                const int base = dph_pow2_16th[e16 & 15];
                s12 = ip >= 0 ? base << ip : base >> (-ip);
            }
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            unsigned word = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint64_t e = e0 + d * 4 + b;
                const unsigned h = dph_hash32((unsigned)e, (unsigned)(e >> 32) ^ seed_hi, seed_lo);
                int v;
                if (KIND == 0) {
                    v = DPH_CENTER + ((dph_ih4(h) * 5321 + 32768) >> 16);
                } else if (KIND == 4) {
                    const unsigned j = j0 + d * 4 + b;
                    const int rs = dph_rogue_slot(j);
                    if (rs < 0) {
                        const unsigned hc = dph_hash32(cluster * 768u + j, 0xD7u + (cluster >> 20), seed_lo + 0x51EDu);
                        const int ab = ((dph_ih4(hc) * 5099 + 32768) >> 16) + ((dph_ih4(h) * 1508 + 32768) >> 16);
                        v = DPH_CENTER + ((ab * s12) >> 12);
                    } else {
                        const int m = rs == 0 ? -105 : (rs == 1 ? 110 : (rs == 2 ? -95 : (rs == 3 ? 100 : -110)));
                        v = m + ((((dph_ih4(h) * 2661 + 32768) >> 16) * s12) >> 12);
                    }
                } else if (KIND == 2) {
                    const unsigned j = j0 + d * 4 + b;
                    const unsigned hc = dph_hash32(cluster * 768u + j, 0xD7u + (cluster >> 20), seed_lo + 0x51EDu);
                    v = DPH_CENTER + ((dph_ih4(hc) * 5099 + 32768) >> 16) + ((dph_ih4(h) * 1508 + 32768) >> 16);
                } else {
                    const unsigned j = j0 + d * 4 + b;
                    const unsigned hc = dph_hash32(cluster * 768u + j, 0xC1u, seed_lo + 0x9E37u);
                    v = DPH_CENTER + ((dph_ih4(hc) * 4434 + 32768) >> 16) + ((dph_ih4(h) * 2217 + 32768) >> 16);
                    if (outlier) v = ((hr >> (j & 15u)) & 1u) ? 127 : -128;
                }
                v = v < -128 ? -128 : (v > 127 ? 127 : v);
                word |= ((unsigned)v & 255u) << (8 * b);
            }
            w[d] = word;
        }
        *(uint4*)(db + o) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
void dph_launch_fill(int8_t* db, int64_t n_rows, int64_t id_base, uint64_t seed, int kind, hipStream_t st) {
    const int64_t n_bytes = n_rows * DPH_DIM;
    if (kind == 4)
        hipLaunchKernelGGL(dph_fill_kernel<4>, dim3(256 * 8), dim3(256), 0, st, db, n_bytes, id_base * DPH_DIM,
                           (unsigned)seed, (unsigned)(seed >> 32));
    else if (kind == 3)
        hipLaunchKernelGGL(dph_fill_kernel<3>, dim3(256 * 8), dim3(256), 0, st, db, n_bytes, id_base * DPH_DIM,
                           (unsigned)seed, (unsigned)(seed >> 32));
    else if (kind == 2)
        hipLaunchKernelGGL(dph_fill_kernel<2>, dim3(256 * 8), dim3(256), 0, st, db, n_bytes, id_base * DPH_DIM,
                           (unsigned)seed, (unsigned)(seed >> 32));
    else if (kind == 1)
        hipLaunchKernelGGL(dph_fill_kernel<1>, dim3(256 * 8), dim3(256), 0, st, db, n_bytes, id_base * DPH_DIM,
                           (unsigned)seed, (unsigned)(seed >> 32));
    else
        hipLaunchKernelGGL(dph_fill_kernel<0>, dim3(256 * 8), dim3(256), 0, st, db, n_bytes, id_base * DPH_DIM,
                           (unsigned)seed, (unsigned)(seed >> 32));
}

// ------------------------------------------------------------------------------------------ shard statistics
// Per-dimension sums of the stored rows: sums[j] = sum n_j, sums[768 + j] = sum n_j^2 (exact integers; the host takes the mean code
// mu_j and the spread of every dimension from them: dph_index_finalize).  A lane owns 16 dimensions for the whole launch.
__global__ __launch_bounds__(256) void dph_colstats_kernel(const int8_t* __restrict__ db, int64_t n_rows,
                                                           const int64_t* __restrict__ row_ids, long long* __restrict__ sums) {
    __shared__ long long part[4][48][32];               // [wave][lane][16 sums | 16 sums of squares]
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    long long a1[16], a2[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a1[i] = 0; a2[i] = 0; }
    for (int64_t row = wave0; row < n_rows; row += nwaves) {
        if (row_ids && row_ids[row] < 0) continue;      // list padding: not a row of the dump
        if (lane < 48) {
            const uint4 v = *(const uint4*)(db + row * DPH_DIM + lane * 16);
            const unsigned wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int n = (int)(int8_t)(wd[d] >> (8 * b));
                    a1[d * 4 + b] += n;
                    a2[d * 4 + b] += n * n;
                }
        }
    }
    if (lane < 48) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { part[w][lane][i] = a1[i]; part[w][lane][16 + i] = a2[i]; }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 48 * 32; e += 256) {
        const int l = e / 32, i = e % 32;
        const long long v = part[0][l][i] + part[1][l][i] + part[2][l][i] + part[3][l][i];
        if (v) atomicAdd((unsigned long long*)&sums[(i < 16 ? 0 : DPH_DIM) + l * 16 + (i & 15)], (unsigned long long)v);
    }
}
void dph_launch_colstats(const int8_t* db, int64_t n_rows, const int64_t* row_ids, long long* sums, hipStream_t st) {
    hipLaunchKernelGGL(dph_colstats_kernel, dim3(256 * 4), dim3(256), 0, st, db, n_rows, row_ids, sums);
}

// squared centred norm sum_j (n_j - mu_j)^2 of a row, summed over the wave (lanes 0..47 hold 16 dimensions each)
__device__ __forceinline__ int dph_row_norm2(const int8_t* __restrict__ db, int64_t row, const int (&mu16)[16], int lane) {
    int acc = 0;
    if (lane < 48) {
        const uint4 v = *(const uint4*)(db + row * DPH_DIM + lane * 16);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int n = (int)(int8_t)(w[d] >> (8 * b)) - mu16[d * 4 + b];
                acc += n * n;
            }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    return acc;
}

// The aux rows of a shard (dph_scan.hip "the aux k-step", layout: dph_internal.h): one wave per stored row.
//   norm slots     their sum is the smallest code c with (c * unit)^2 >= || n - mu ||^2, dealt out 127 at a time;
//   replica slots  the row's raw code in the slot's rogue dimension;
//   padding rows (beyond n_rows, list padding) and unused slots: zero -- their high-digit score stays 0 whatever the query.
__global__ __launch_bounds__(256) void dph_aux_build_kernel(const int8_t* __restrict__ db, int64_t n_rows, int64_t n_padded,
                                                            const int64_t* __restrict__ row_ids, const int* __restrict__ mu,
                                                            dph_aux_layout lay, int unit, int8_t* __restrict__ aux) {
    const int lane = threadIdx.x & 63;
    int mu16[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) mu16[i] = lane < 48 ? mu[lane * 16 + i] : 0;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t row = wave0; row < n_padded; row += nwaves) {
        const bool real = row < n_rows && !(row_ids && row_ids[row] < 0);
        int code = 0;
        if (real) {
            const long long n2 = dph_row_norm2(db, row, mu16, lane);
            long long c = (long long)(sqrt((double)n2) / (double)unit);
            while (c > 0 && (c - 1) * unit * (c - 1) * unit >= n2) --c;          // smallest c with (c unit)^2 >= n2, exactly
            while (c * unit * c * unit < n2) ++c;
            code = (int)c;
        }
        if (lane < lay.stride) {
            int b = 0;
            if (real) {
                if (lane < lay.n_norm) {
                    const int left = code - 127 * lane;
                    b = left > 127 ? 127 : (left > 0 ? left : 0);
                } else if (lane < lay.n_norm + lay.n_rep) {
                    b = (int)db[row * DPH_DIM + lay.rep_dim[lane - lay.n_norm]];
                }
            }
            aux[row * lay.stride + lane] = (int8_t)b;
        }
    }
}
void dph_launch_aux_build(const int8_t* db, int64_t n_rows, int64_t n_rows_padded, const int64_t* row_ids, const int* mu_dev,
                          const dph_aux_layout& lay, int norm_unit, int8_t* aux, hipStream_t st) {
    if (n_rows_padded <= 0 || lay.stride <= 0) return;
    hipLaunchKernelGGL(dph_aux_build_kernel, dim3(256 * 8), dim3(256), 0, st, db, n_rows, n_rows_padded, row_ids, mu_dev, lay,
                       norm_unit, aux);
}

// ------------------------------------------------------------------------------------------ centred row norms
// Squared centred norm sum_j (n_j - mu_j)^2 of every real row (exact integer; mu = the shard's per-dimension mean codes).  mode 0:
// global max + a histogram with DPH_NORM_BINS bins of DPH_NORM_BIN_W (the host picks the outlier cut and the median from it);
// mode 1: append every row whose squared norm exceeds `cut2` to `out_rows` (the shard's outlier rows: always scored exactly,
// never bounded).
__global__ __launch_bounds__(256) void dph_rownorm_kernel(const int8_t* __restrict__ db, int64_t n_rows,
                                                          const int64_t* __restrict__ row_ids, const int* __restrict__ mu,
                                                          unsigned long long* __restrict__ max_out,
                                                          unsigned* __restrict__ hist, unsigned long long cut2,
                                                          unsigned* __restrict__ out_rows, unsigned* __restrict__ out_count,
                                                          unsigned out_cap) {
    __shared__ unsigned lhist[DPH_NORM_BINS];          // per-workgroup histogram (every row of an i.i.d. dump hits the same few bins)
    const int lane = threadIdx.x & 63;
    const bool collect = out_rows != nullptr;
    if (!collect && hist) {
        for (int i = threadIdx.x; i < DPH_NORM_BINS; i += 256) lhist[i] = 0;
        __syncthreads();
    }
    int mu16[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) mu16[i] = lane < 48 ? mu[lane * 16 + i] : 0;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    int best = 0;
    for (int64_t row = wave0; row < n_rows; row += nwaves) {
        if (row_ids && row_ids[row] < 0) continue;      // list padding: not a row of the dump
        const int acc = dph_row_norm2(db, row, mu16, lane);
        if (collect) {
            if (lane == 0 && (unsigned long long)acc > cut2) {
                const unsigned s = atomicAdd(out_count, 1u);
                if (s < out_cap) out_rows[s] = (unsigned)row;
            }
        } else {
            best = max(best, acc);
            if (lane == 0 && hist) {
                const unsigned b = (unsigned)acc / DPH_NORM_BIN_W;
                atomicAdd(&lhist[b < DPH_NORM_BINS ? b : DPH_NORM_BINS - 1], 1u);
            }
        }
    }
    if (!collect) {
        if (lane == 0 && best > 0) atomicMax(max_out, (unsigned long long)best);
        if (hist) {
            __syncthreads();
            for (int i = threadIdx.x; i < DPH_NORM_BINS; i += 256)
                if (lhist[i]) atomicAdd(&hist[i], lhist[i]);
        }
    }
}
void dph_launch_rownorm(const int8_t* db, int64_t n_rows, const int64_t* row_ids, const int* mu_dev, unsigned long long* max_out,
                        unsigned* hist, unsigned long long cut2, unsigned* out_rows, unsigned* out_count, unsigned out_cap,
                        hipStream_t st) {
    hipLaunchKernelGGL(dph_rownorm_kernel, dim3(256 * 8), dim3(256), 0, st, db, n_rows, row_ids, mu_dev, max_out, hist, cut2,
                       out_rows, out_count, out_cap);
}
