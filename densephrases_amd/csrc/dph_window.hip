// dph_window.hip -- start/end window re-scoring (replaces /root/reference/densephrases/index.py:276-370: the
// 2*B*k*L python-level faiss.reconstruct calls, the zeros[B*k, L, 768] fill, end @ R, the (q * end).sum(2)
// and the valid_phrase mask) with one gather + de-quantise + dot + mask + arg-max kernel over the resident shard.
// One wavefront per candidate; rows are read exactly once (768 B each), scores accumulate in fp64.
#include "dph_internal.h"

__global__ __launch_bounds__(256) void dph_window_kernel(
    int direction, const int8_t* __restrict__ db, int64_t n_rows, dph_idmap idmap, const float* __restrict__ lut,
    const float* __restrict__ qhalf, int64_t n_cand, int k, int L, const int64_t* __restrict__ ids,
    const int32_t* __restrict__ doc_in, const int32_t* __restrict__ word_in, const float* __restrict__ first,
    const int32_t* __restrict__ row2doc, const int32_t* __restrict__ row2word, const int32_t* __restrict__ doc_ids,
    int64_t n_docs, const int64_t* __restrict__ f2o_off, const int32_t* __restrict__ f2o,
    const int32_t* __restrict__ inv_row, int32_t* __restrict__ pred_word, double* __restrict__ best,
    int32_t* __restrict__ argslot, float* __restrict__ vecs) {
    __shared__ float lut_lds[256];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int j = tid; j < 256; j += 256) lut_lds[j] = lut[j];
    __syncthreads();
    const int64_t c = (int64_t)blockIdx.x * 4 + (tid >> 6);
    if (c >= n_cand) return;

    const int64_t id = ids[c];
    // stored row of id + delta: consecutive GLOBAL ids like the reference's reconstruct loop (index.py:282-300); an id
    // this shard does not hold counts as a zero vector (:285-288).  List-major (IVF) shards go through inv_row.
    auto row_of = [&](int64_t delta) -> int64_t {
        const int64_t l = dph_local_of_id(idmap, id + delta);
        return (l < 0) ? -1 : (inv_row ? (int64_t)inv_row[l] : l);
    };
    // (doc, word) of the candidate when the caller did not pass them: idx2id is indexed by local id (= stored row on
    // flat and grouped shards); ids outside the shard are clipped like get_idxs (index.py:128-133)
    int64_t lc = dph_local_of_id(idmap, id);
    // false for an id this shard does not hold.  Range-sharded: a candidate of another rank -- its vectors come back zero here and the
    // SUM all-reduce fills them in.  Single rank: only FAISS' -1 padding (fewer than k rows) gets here, and zero is what the reference's
    // RAM branch returns for it too: reconst_fn(-1) raises and index.py:285-288 substitutes np.zeros (the HDF5 branch, which would read
    // the row of the CLIPPED id, is not the one this path follows: DESIGN.md section 2 "reference quirks").
    const bool mine = lc >= 0;
    if (lc < 0) {
        const int64_t first = idmap.n_groups ? idmap.id_offsets[0] : idmap.id_base;
        lc = id < first ? 0 : idmap.n_ids - 1;
        if (lc < 0) lc = 0;
    }
    const int d = doc_in ? doc_in[c] : row2doc[lc];
    const int w = word_in ? word_in[c] : row2word[lc];

    // document -> f2o CSR slot (binary search over the sorted doc ids)
    int64_t lo = 0, hi = n_docs;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (doc_ids[mid] < d) lo = mid + 1; else hi = mid; }
    const bool have_doc = d >= 0 && lo < n_docs && doc_ids[lo] == d;
    const int64_t fbase = have_doc ? f2o_off[lo] : 0;
    const int64_t flen = have_doc ? f2o_off[lo + 1] - fbase : 0;

    float q[12];
    const float* qp = qhalf + (c / k) * DPH_DIM + lane * 12;
#pragma unroll
    for (int j = 0; j < 12; ++j) q[j] = qp[j];

    double best_s = 0.0;
    int best_slot = -1;
    int best_word = -1;
    const float f0 = first[c];
    for (int s = 0; s < L; ++s) {
        const int i = direction == 0 ? s : (L - 1 - s);
        const int64_t ww = direction == 0 ? (int64_t)w + i : (int64_t)w - i;
        const int64_t row = row_of(direction == 0 ? (int64_t)i : -(int64_t)i);
        bool valid = have_doc && w >= 0 && w < flen && ww >= 0 && ww < flen;          // index.py:305-321
        if (valid) {
            const int64_t gap = direction == 0 ? (int64_t)f2o[fbase + ww] - (int64_t)f2o[fbase + w]
                                               : (int64_t)f2o[fbase + w] - (int64_t)f2o[fbase + ww];
            valid = gap >= 0 && gap <= L;
        }
        double dot = 0.0;
        if (row >= 0 && row < n_rows) {                 // outside the shard: zero vector (index.py:285-288)
            const unsigned* p = (const unsigned*)(db + row * DPH_DIM + lane * 12);
            const unsigned wv[3] = {p[0], p[1], p[2]};
#pragma unroll
            for (int dd = 0; dd < 3; ++dd)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    dot += (double)q[dd * 4 + b] * (double)lut_lds[(int)(int8_t)(wv[dd] >> (8 * b)) + 128];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
        }
        // fp32 + fp32 in fp32, then the float64 mask: np.expand_dims(start_scores, 1) + new_end_scores + end_mask (:343 / :368)
        const double sc = (double)(f0 + (float)dot) + (valid ? 0.0 : -1e9);
        if (best_slot < 0 || sc > best_s) { best_s = sc; best_slot = s; best_word = valid ? (int)ww : -1; }
    }
    if (lane == 0) {
        pred_word[c] = best_word;
        best[c] = best_s;
        argslot[c] = best_slot;
    }
    if (vecs) {
        // [c,0,:] the candidate's own row, [c,1,:] the arg-max slot's row (index.py:345,370,381-389)
        const int bi = direction == 0 ? best_slot : (L - 1 - best_slot);
        const int64_t rows2[2] = {row_of(0), row_of(direction == 0 ? (int64_t)bi : -(int64_t)bi)};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float* o = vecs + (c * 2 + t) * DPH_DIM + lane * 12;
            // (a candidate of another rank contributes zero vectors even when one of its window slots lies in this shard: the ranks'
            //  vectors are summed, only the candidate's own rank may speak)
            if (mine && rows2[t] >= 0 && rows2[t] < n_rows) {
                const unsigned* p = (const unsigned*)(db + rows2[t] * DPH_DIM + lane * 12);
                const unsigned wv[3] = {p[0], p[1], p[2]};
#pragma unroll
                for (int dd = 0; dd < 3; ++dd)
#pragma unroll
                    for (int b = 0; b < 4; ++b) o[dd * 4 + b] = lut_lds[(int)(int8_t)(wv[dd] >> (8 * b)) + 128];
            } else {
#pragma unroll
                for (int j = 0; j < 12; ++j) o[j] = 0.f;
            }
        }
    }
}

void dph_launch_window(int direction, const int8_t* db, int64_t n_rows, dph_idmap idmap, const float* lut_dev,
                       const float* qhalf, int64_t n_cand, int k, int L, const int64_t* ids, const int32_t* doc,
                       const int32_t* word, const float* first, const int32_t* row2doc, const int32_t* row2word,
                       const int32_t* doc_ids, int64_t n_docs, const int64_t* f2o_off, const int32_t* f2o,
                       const int32_t* inv_row, int32_t* pred_word, double* best, int32_t* argslot, float* vecs,
                       hipStream_t st) {
    if (n_cand <= 0) return;
    hipLaunchKernelGGL(dph_window_kernel, dim3((unsigned)((n_cand + 3) / 4)), dim3(256), 0, st, direction, db,
                       n_rows, idmap, lut_dev, qhalf, n_cand, k, L, ids, doc, word, first, row2doc, row2word,
                       doc_ids, n_docs, f2o_off, f2o, inv_row, pred_word, best, argslot, vecs);
}

// ------------------------------------------------------------------------------------------ encoder.py scoring
// The start/end-vector scoring of /root/reference/densephrases/encoder.py:
//   train_query (:383-386)  start_logits = query_start.matmul(start_vecs.transpose(1, 2)).squeeze(1)   [B,1,768]x[B,768,M]
//   forward     (:206-208)  start_logits = start.matmul(query_start.transpose(1, 2)).squeeze(-1)        [bs,T,768]x[bs,768,1]
//                           dense_logits = start_logits.unsqueeze(2) + end_logits.unsqueeze(1)
// Both contractions are out[b, m] = <q[b, :], vecs[b, m, :]> -- the dot of the window kernel without the gather: one
// wavefront per (b, m), 12 contiguous floats per lane, fp32 FMA chain in a fixed order, butterfly.  HBM-bound
// (3 KiB read per output); the vectors are the [B, 2k, 768] arrays MIPS.search(return_idxs=True) returns.
__global__ __launch_bounds__(256) void dph_score_vecs_kernel(const float* __restrict__ q, const float* __restrict__ vecs,
                                                             int64_t n_b, int64_t m, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n_b * m) return;
    const int64_t b = c / m;
    const float4* qp = (const float4*)(q + b * DPH_DIM + lane * 12);
    const float4* vp = (const float4*)(vecs + c * DPH_DIM + lane * 12);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float4 a = qp[i], v = vp[i];
        acc = fmaf(a.x, v.x, acc); acc = fmaf(a.y, v.y, acc); acc = fmaf(a.z, v.z, acc); acc = fmaf(a.w, v.w, acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) out[c] = acc;
}
// gradient of the above w.r.t. q (what query-side fine-tuning back-propagates into the query encoder,
// train_query.py:208-275): grad_q[b, j] = sum_m grad[b, m] * vecs[b, m, j]; one workgroup per b, 3 columns per thread
__global__ __launch_bounds__(256) void dph_score_vecs_bwd_kernel(const float* __restrict__ grad, const float* __restrict__ vecs,
                                                                 int64_t m, float* __restrict__ grad_q) {
    const int64_t b = blockIdx.x;
    float acc[3] = {0.f, 0.f, 0.f};
    for (int64_t i = 0; i < m; ++i) {
        const float g = grad[b * m + i];
        const float* v = vecs + (b * m + i) * DPH_DIM;
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[t] = fmaf(g, v[threadIdx.x + 256 * t], acc[t]);
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) grad_q[b * DPH_DIM + threadIdx.x + 256 * t] = acc[t];
}
// dense_logits[b, i, j] = start_logits[b, i] + end_logits[b, j]   (encoder.py:208)
__global__ __launch_bounds__(256) void dph_dense_logits_kernel(const float* __restrict__ s, const float* __restrict__ e,
                                                               int64_t n_b, int64_t T, float* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_b * T * T) return;
    const int64_t b = idx / (T * T), r = idx % (T * T);
    out[idx] = s[b * T + r / T] + e[b * T + r % T];
}

void dph_launch_score_vecs(const float* q, const float* vecs, int64_t n_b, int64_t m, float* out, hipStream_t st) {
    if (n_b * m <= 0) return;
    hipLaunchKernelGGL(dph_score_vecs_kernel, dim3((unsigned)((n_b * m + 3) / 4)), dim3(256), 0, st, q, vecs, n_b, m, out);
}
void dph_launch_score_vecs_bwd(const float* grad, const float* vecs, int64_t n_b, int64_t m, float* grad_q, hipStream_t st) {
    if (n_b <= 0) return;
    hipLaunchKernelGGL(dph_score_vecs_bwd_kernel, dim3((unsigned)n_b), dim3(256), 0, st, grad, vecs, m, grad_q);
}
void dph_launch_dense_logits(const float* s, const float* e, int64_t n_b, int64_t T, float* out, hipStream_t st) {
    if (n_b * T * T <= 0) return;
    hipLaunchKernelGGL(dph_dense_logits_kernel, dim3((unsigned)((n_b * T * T + 255) / 256)), dim3(256), 0, st, s, e, n_b, T, out);
}
