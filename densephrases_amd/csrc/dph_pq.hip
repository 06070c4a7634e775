// dph_pq.hip -- the reference's OWN index type served from HBM: IndexPreTransform(OPQMatrix) -> IndexIVFPQ (inner product,
// by_residual, 8-bit codes) as build_phrase_index.py:108-116 trains it and index.py:30-33,200,286 uses it (SURVEY.md 8 row
// a3 / f2, BASELINE.json north_star "(or PQ-compressed)").  gfx950 only.
//
//   x'          = A x (+ b)                                   pq_transform_kernel      (float64 accumulation, one rounding)
//   coarse      = top-nprobe lists by <x', c_l>               dph_launch_coarse        (f32 MFMA GEMM + radix select + float64 band re-rank)
//   LUT[m][j]   = <x'_m, codeword[m][j]>                      pq_lut_kernel            (float64 accumulation, one rounding)
//   score(code) = <x', c_list> + sum_m LUT[m][code[m]]        pq_adc_kernel            (fp32, summed SEQUENTIALLY in m like FAISS' scalar scan)
//   top-k       = k largest, (score desc, id asc)             pq_adc_kernel + pq_final_kernel
//   reconstruct = c_list + concat_m codeword[m][code[m]]      pq_reconstruct_kernel / pq_window_kernel
//
// ADC scan.  Work = (query row, probed list) pairs from the coarse probe masks, taken from a work queue by workgroups of
// 1024 threads.  A workgroup keeps the row's LUT in LDS (M x 256 fp32 = 96 KiB at OPQ96) and walks the list in segments of
// PQ_SEG codes: every thread sums its codes (16-byte loads of the code bytes, one LDS gather per sub-quantiser), the
// segment's scores go to LDS as order-preserving uint32 keys, a 4-pass radix select finds the segment's k-th largest, and
// only keys >= max(that, the row's running bound) are appended to the row's candidate list in HBM; the row's bound is
// raised with atomicMax.  The bound is always the k-th largest of SOME subset of the row's scores, i.e. a lower bound of
// the true k-th best, so no member of the true top-k is ever dropped.  pq_final_kernel selects the k best candidates of a
// row exactly (radix select over its candidate list, then a sort of the survivors by (score desc, id asc)).
// LDS gathers with random codes hit ~4-way bank conflicts: the kernel is LDS-gather-bound, the roofline DESIGN.md prices
// it against is M gathers per code at 64 lanes x 1 gather / (conflict factor) per clock per CU.
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "dph_internal.h"

#define PQ_THREADS 1024
#define PQ_SEG 12288                  // codes per segment (M <= 96: 96 KiB of LUT + 48 KiB of keys); halved for larger M (LDS budget).  12288: a row-major unit of the released shape (64 lists, ~10 k codes) is ONE segment -- one k-th selection per unit instead of two (8192: r1-r4)
#define PQ_FINAL_CAP 4096             // candidates the final sort takes (k + boundary ties)

static thread_local std::string g_pq_err;
const char* dph_pq_error() { return g_pq_err.c_str(); }
static int pq_fail(int code, const std::string& m) {
    g_pq_err = m;
    if (code == DPH_E_HIP || code == DPH_E_NOMEM) (void)hipGetLastError();       // (consumed: see fail() in dph_api.hip)
    return code;
}
#define PQCHK(expr)                                                                                     \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return pq_fail(e_ == hipErrorOutOfMemory ? DPH_E_NOMEM : DPH_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

__device__ __forceinline__ unsigned pq_key(float s) {
    const unsigned b = __float_as_uint(s);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);          // larger float <=> larger key; key 0 is below every score
}
__device__ __forceinline__ float pq_unkey(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// ---------------------------------------------------------------------------------------------- x' = A x (+ b)
// At = A transposed ([d_in][d_out]): lane j walks column j, coalesced over the lanes, t ascending
// out_hi / out_pk (optional): the row once more as bf16 and as packed bf16 hi << 16 | lo -- the operands of the coarse quantizer's
// filter GEMM and of its bf16x3 fail-over chain (dph_ivf.hip), written here instead of by two more launches
__device__ __forceinline__ unsigned short pq_bf16_rne(float v) {
    const unsigned u = __float_as_uint(v);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
// grid (ceil(rows / 8), DPH_DIM / 64) x 256 threads: a workgroup owns 8 rows x 64 output columns -- wave w rows 2w, 2w + 1, lane =
// column.  The 64 x 64 block of A^T of a step is loaded ONCE per workgroup (16 values per thread, coalesced), converted to float64
// once and shared through LDS (two buffers, one barrier per step); the blocks of the next three steps are in flight in registers.
// One column of one row per thread (rounds 1-4) read all of A per row -- 302 MB through the L2 for 128 rows, 28-38 us however
// the loads were batched -- and converted both factors of every term.  Same arithmetic: fma(double(A_jt), double(x_t), acc), t
// ascending, one rounding to float at the end.
#define PQT_ROWS 8
#define PQT_COLS 64
#define PQT_TB 64
__global__ __launch_bounds__(256) void pq_transform_kernel(const float* __restrict__ x, const float* __restrict__ At, int64_t n_rows,
                                                           const float* __restrict__ b, float* __restrict__ out,
                                                           unsigned short* __restrict__ out_hi = nullptr, unsigned* __restrict__ out_pk = nullptr,
                                                           int32_t* __restrict__ nonfinite = nullptr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pqt_smem[];
    __shared__ int pqt_bad[PQT_ROWS];
    double (*xs)[DPH_DIM] = (double (*)[DPH_DIM])pqt_smem;                                       // [PQT_ROWS][768]
    double (*as_)[PQT_TB][PQT_COLS] = (double (*)[PQT_TB][PQT_COLS])(pqt_smem + sizeof(double) * PQT_ROWS * DPH_DIM);   // [2][64][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * PQT_ROWS;
    const int j = blockIdx.y * PQT_COLS + lane;
    constexpr int NST = DPH_DIM / PQT_TB, RING = 4, PER = PQT_TB / 4;          // 12 steps; a thread loads PER = 16 values of a step's block
    float ring[RING][PER];
    auto issue = [&](int st, float (&dst)[PER]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PER; ++i) dst[i] = At[(int64_t)(st * PQT_TB + wave * PER + i) * DPH_DIM + j];
    };
    auto stage = [&](int st, const float (&src)[PER]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PER; ++i) as_[st & 1][wave * PER + i][lane] = (double)src[i];
    };
    if (At) {
#pragma unroll
        for (int st = 0; st < RING - 1; ++st) issue(st, ring[st]);
    }
    {
        float xr[PQT_ROWS][DPH_DIM / 256];                 // the 8 rows, all 24 loads of a thread in flight together
#pragma unroll
        for (int rr = 0; rr < PQT_ROWS; ++rr)
#pragma unroll
            for (int q = 0; q < DPH_DIM / 256; ++q) xr[rr][q] = x[(r0 + rr < n_rows ? r0 + rr : n_rows - 1) * DPH_DIM + tid + 256 * q];      // (a row past the end: the last row again, never stored)
        if (nonfinite) {
            // a row with a NaN / Inf element (search only: `nonfinite` set): flagged, searched as a bounded stand-in pattern so that the
            // coarse filter and the ADC scan see ordinary numbers (a NaN estimate fails the filter and sends the WHOLE pass down the
            // bf16x3 chain), answered with -1 / -FLT_MAX / status DPH_ROW_NONFINITE by pq_final_kernel -- what FAISS returns for it
            if (tid < PQT_ROWS) pqt_bad[tid] = 0;
            __syncthreads();
#pragma unroll
            for (int rr = 0; rr < PQT_ROWS; ++rr) {
                bool ok = true;
#pragma unroll
                for (int q = 0; q < DPH_DIM / 256; ++q) ok = ok && isfinite(xr[rr][q]);
                if (!ok) pqt_bad[rr] = 1;
            }
            __syncthreads();
#pragma unroll
            for (int rr = 0; rr < PQT_ROWS; ++rr) {
                if (pqt_bad[rr]) {
#pragma unroll
                    for (int q = 0; q < DPH_DIM / 256; ++q) xr[rr][q] = (float)(((tid + 256 * q) * 37) % 64) * (1.0f / 64.0f) - 0.4921875f;
                }
                if (blockIdx.y == 0 && tid == 0 && r0 + rr < n_rows) nonfinite[r0 + rr] = pqt_bad[rr];
            }
        }
#pragma unroll
        for (int rr = 0; rr < PQT_ROWS; ++rr)
#pragma unroll
            for (int q = 0; q < DPH_DIM / 256; ++q) xs[rr][tid + 256 * q] = (double)xr[rr][q];
    }
    double acc[2] = {0.0, 0.0};
    if (At) {
        stage(0, ring[0]);
        __syncthreads();
        // Rolled: three trips of four steps (the ring index must be a constant inside a trip), the two halves of a step a loop too.
        // Fully unrolled this was 29 KB of straight-line code run once per wave -- instruction fetch, not arithmetic or loads, was what
        // every earlier form of this kernel waited for (28-38 us whatever else changed).
#pragma unroll 1
        for (int st0 = 0; st0 < NST; st0 += RING) {
#pragma unroll
            for (int sr = 0; sr < RING; ++sr) {
                const int st = st0 + sr;
                // (always issued -- past the last step the last block once more, unused: a conditional load makes the compiler wait for
                // the count of the path without it)
                issue(st + RING - 1 < NST ? st + RING - 1 : NST - 1, ring[(sr + RING - 1) % RING]);
#pragma unroll 1
                for (int h = 0; h < PQT_TB; h += 32) {
                    double av[32];
                    double2 xv[2][16];
#pragma unroll
                    for (int u = 0; u < 32; ++u) av[u] = as_[sr & 1][h + u][lane];
#pragma unroll
                    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                        for (int u = 0; u < 16; ++u) xv[rr][u] = *(const double2*)&xs[2 * wave + rr][st * PQT_TB + h + 2 * u];
                    __builtin_amdgcn_sched_barrier(0);    // (left alone, the scheduler sinks every LDS read next to its use and waits for each)
#pragma unroll
                    for (int u = 0; u < 16; ++u)
#pragma unroll
                        for (int rr = 0; rr < 2; ++rr) {
                            acc[rr] = fma(av[2 * u], xv[rr][u].x, acc[rr]);
                            acc[rr] = fma(av[2 * u + 1], xv[rr][u].y, acc[rr]);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (st + 1 < NST) {                       // the other buffer: its readers left at the barrier that ended step st - 1
#pragma unroll
                    for (int i = 0; i < PER; ++i) as_[(sr + 1) & 1][wave * PER + i][lane] = (double)ring[(sr + 1) % RING][i];
                }
                __syncthreads();
            }
        }
    } else {
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) acc[rr] = xs[2 * wave + rr][j];
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int64_t r = r0 + 2 * wave + rr;
        if (r >= n_rows) break;
        double s_ = acc[rr];
        if (b) s_ += (double)b[j];
        const float v = (float)s_;
        out[r * DPH_DIM + j] = v;
        if (out_hi) {
            const unsigned short hi = pq_bf16_rne(v);
            out_hi[r * DPH_DIM + j] = hi;
            if (out_pk) out_pk[r * DPH_DIM + j] = ((unsigned)hi << 16) | (unsigned)pq_bf16_rne(v - __uint_as_float((unsigned)hi << 16));
        }
    }
}
static constexpr size_t PQT_LDS = sizeof(double) * (PQT_ROWS * DPH_DIM + 2 * PQT_TB * PQT_COLS);       // 48 KiB + 64 KiB
static int pq_launch_transform(const float* x, const float* At, int64_t n_rows, const float* b, float* out, unsigned short* out_hi, unsigned* out_pk,
                               int device, hipStream_t st, int32_t* nonfinite = nullptr) {
    static std::atomic<bool> attr[64];
    if (device < 0 || device >= 64 || !attr[device]) {
        const hipError_t e = hipFuncSetAttribute((const void*)pq_transform_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PQT_LDS);
        if (e != hipSuccess) return 1;
        if (device >= 0 && device < 64) attr[device] = true;
    }
    hipLaunchKernelGGL(pq_transform_kernel, dim3((unsigned)((n_rows + PQT_ROWS - 1) / PQT_ROWS), DPH_DIM / PQT_COLS), dim3(256), PQT_LDS, st, x, At, n_rows, b, out,
                       out_hi, out_pk, nonfinite);
    return 0;
}

// ---------------------------------------------------------------------------------------------- LUT[r][m][j]
// grid (rows, M / 8): a workgroup fills eight sub-quantisers' tables of a row (one workgroup per table was 12288 launches of
// 2 K multiply-adds at the released shape: 50 us of launch overhead).  DSUB > 0 (a multiple of 4): the thread's eight centroid
// vectors are loaded as float4 BEFORE the first sum starts -- with dsub a run-time value the loop issued one scalar load per
// component and waited for each (64 exposed L2 round trips per thread: 48 us for 64 rows); the sums themselves are unchanged
// (float64, t ascending)
template <int DSUB>
__global__ __launch_bounds__(256) void pq_lut_kernel(const float* __restrict__ xp, const float* __restrict__ pqc, int M, int dsub,
                                                     float* __restrict__ lut) {
    const int64_t r = blockIdx.x;
    const int j = threadIdx.x;
    if constexpr (DSUB > 0) {
        const int m0 = blockIdx.y * 8;
        float4 c[8][DSUB / 4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int m = m0 + u < M ? m0 + u : M - 1;
            const float4* cp = (const float4*)(pqc + ((int64_t)m * 256 + j) * DSUB);
#pragma unroll
            for (int v = 0; v < DSUB / 4; ++v) c[u][v] = cp[v];
        }
        __builtin_amdgcn_sched_barrier(0);                // (no branch and no scheduling across: the loads would be sunk next to their uses)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int m = m0 + u < M ? m0 + u : M - 1;     // (a clamped table is computed again and stored again: same value)
            const float* q = xp + r * DPH_DIM + m * DSUB;
            double acc = 0.0;
#pragma unroll
            for (int v = 0; v < DSUB / 4; ++v) {
                acc += (double)q[4 * v] * (double)c[u][v].x;
                acc += (double)q[4 * v + 1] * (double)c[u][v].y;
                acc += (double)q[4 * v + 2] * (double)c[u][v].z;
                acc += (double)q[4 * v + 3] * (double)c[u][v].w;
            }
            lut[(r * M + m) * 256 + j] = (float)acc;
        }
    } else {
        for (int m = blockIdx.y * 8; m < M && m < blockIdx.y * 8 + 8; ++m) {
            const float* c = pqc + ((int64_t)m * 256 + j) * dsub;
            const float* q = xp + r * DPH_DIM + m * dsub;
            double acc = 0.0;
            for (int t = 0; t < dsub; ++t) acc += (double)q[t] * (double)c[t];
            lut[(r * M + m) * 256 + j] = (float)acc;
        }
    }
}

// ---------------------------------------------------------------------------------------------- (list, row) pairs
// one thread per list: every query row of the pass whose bit is set in the list's probe mask becomes a pair -- one pair per CHUNK of
// PQ_LM_CHUNK segments of the list (round 6: work cut by code count; a list of millions of codes among lists of thousands is many
// units of many workgroups, not one workgroup's).  Record: (list, row | chunk << 10); the rows of a pass fit 10 bits (DPH_PASS_MAX).
#define PQ_LM_CHUNK 4
static_assert(DPH_PASS_MAX <= 1024, "pair records keep the row in 10 bits");
__global__ __launch_bounds__(256) void pq_pairs_kernel(const unsigned* __restrict__ listmask, int mask_words, int nlist,
                                                       const int64_t* __restrict__ list_off, int seg, int2* __restrict__ pairs,
                                                       int* __restrict__ n_pairs, int cap, unsigned* __restrict__ overflow) {
    const int l = blockIdx.x * 256 + threadIdx.x;
    if (l >= nlist) return;
    const long long len = list_off[l + 1] - list_off[l];
    if (len == 0) return;                                       // empty list: nothing to scan
    const int nch = (int)((len + (long long)seg * PQ_LM_CHUNK - 1) / ((long long)seg * PQ_LM_CHUNK));
    for (int w = 0; w < mask_words; ++w) {
        unsigned m = listmask[(int64_t)l * mask_words + w];
        while (m) {
            const int bit = __builtin_ctz(m);
            m &= m - 1u;
            const int slot = atomicAdd(n_pairs, nch);
            for (int c = 0; c < nch; ++c) {
                if (slot + c < cap) pairs[slot + c] = make_int2(l, (32 * w + bit) | (c << 10));
                else overflow[32 * w + bit] = 1u;               // (cannot happen: the capacity is sized from the longest lists)
            }
        }
    }
}

// k-th largest of the n keys key_at(0 .. n-1), n >= k >= 1, by four 8-bit radix passes.  All PQ_THREADS threads call it; hist = 256 + 4
// words of LDS.  The bin of each pass comes from suffix sums over one wave (lane l owns bins 4l .. 4l+3).
template <class F>
__device__ __forceinline__ unsigned pq_select_kth(F key_at, int n, int k, unsigned* hist) {
    unsigned prefix = 0;
    int left = k;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int i = threadIdx.x; i < 256; i += PQ_THREADS) hist[i] = 0;
        __syncthreads();
        const unsigned hi_mask = shift == 24 ? 0u : (0xFFFFFFFFu << (shift + 8));
        for (int i = threadIdx.x; i < n; i += PQ_THREADS) {
            const unsigned key = key_at(i);
            if ((key & hi_mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x < 64) {
            const int l = threadIdx.x;
            const unsigned h0 = hist[4 * l], h1 = hist[4 * l + 1], h2 = hist[4 * l + 2], h3 = hist[4 * l + 3];
            const unsigned mine = h0 + h1 + h2 + h3;
            unsigned incl = mine;                                // sum over lanes >= l
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned v = __shfl_down(incl, o);
                if (l + o < 64) incl += v;
            }
            const unsigned above = incl - mine;                  // keys in bins of higher lanes
            if (above < (unsigned)left && (unsigned)left <= incl) {
                unsigned need = (unsigned)left - above;          // rank inside this lane's four bins, from the top
                int bin;
                if (need <= h3) { bin = 4 * l + 3; }
                else if (need <= h3 + h2) { bin = 4 * l + 2; need -= h3; }
                else if (need <= h3 + h2 + h1) { bin = 4 * l + 1; need -= h3 + h2; }
                else { bin = 4 * l; need -= h3 + h2 + h1; }
                hist[256] = (unsigned)bin;
                hist[257] = need;
            }
        }
        __syncthreads();
        prefix |= hist[256] << shift;
        left = (int)hist[257];
        __syncthreads();
    }
    return prefix;
}
__device__ unsigned pq_select_kth_lds(const unsigned* keys, int n, int k, unsigned* hist) {
    return pq_select_kth([&](int i) { return keys[i]; }, n, k, hist);
}

// ---------------------------------------------------------------------------------------------- ADC scan
struct pq_scan_args {
    const float* xp; const float* cent; const float* lut; const uint8_t* codes; const int64_t* list_off;
    const int2* pairs; const int* n_pairs; int* next; int pair_cap;
    int M; int seg; int k; int by_residual;
    unsigned* bound; unsigned* cand_count; uint2* cand; int cand_cap; unsigned* overflow;
    int split_lut;                  // row-major scan: the last sixteen tables are read from global memory (pq_adc_sum1 GL)
    const int2* units; const int* n_units; int unit_cap;      // row-major scan: (row * units_per_row + group, segment) records of pq_units_kernel
    unsigned long long* prof;       // debugging (dph_debug_pq_phases): 8 words per workgroup of the row-major scan, null otherwise
};

// What happens to a segment's scores once they sit in LDS as keys: how many reach the row's running bound?  Fewer than k: the
// segment cannot raise it, only those are appended.  Otherwise the segment's k-th largest (radix select) raises the bound
// (atomicMax) and keys >= max(old bound, k-th) are appended.  pos_of(i) = position of key i in the code array.
template <bool SAMPLE, class P>
__device__ __forceinline__ void pq_segment_finish(const pq_scan_args& a, int r, const unsigned* keys_s, int n, unsigned* hist, P pos_of) {
    // SAMPLE (the row-major scan, segments of ~10 k codes): the bound comes from a strided SAMPLE of PQ_THREADS keys -- its k-th largest is
    // the k-th largest of a subset of the row's scores like every other bound here, just a lower one: the segment appends ~k n / 1024
    // keys instead of ~k (pq_final_kernel selects among a few hundred per row instead of a few dozen), and the four radix passes walk
    // one key per thread instead of twelve: 14 us per unit -> 4 (a sixth of the kernel).  The list-major scan keeps the exact k-th: its
    // rows meet thousands of segments, whose appended keys must stay ~k each.
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned bound0 = __hip_atomic_load(&a.bound[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int ns = SAMPLE && n > 2 * PQ_THREADS && a.k <= PQ_THREADS / 8 ? PQ_THREADS : n;     // (a large k wants more than a sample holds: exact)
    // sample key i sits at floor(i n / 1024): the positions span the WHOLE segment.  (i * floor(n / 1024), rounds 5-6, left up to 1023
    // keys at the end of the segment unsampled -- a segment is a sequence of LISTS, each with its own <x', centroid> term, and when the
    // row's best list lay in that tail every code of it beat the sample's k-th: hundreds of candidates from one segment, a row with
    // status 1 once in a few hundred random index shapes, tests/test_fuzz_gpu.py.)
    static_assert(PQ_THREADS == 1024, "the sample positions shift by 10");
    const bool sampled = ns < n;
    auto skey = [&](int i) { return keys_s[sampled ? (int)(((unsigned)i * (unsigned)n) >> 10) : i]; };
    int mine = 0;
    for (int i = tid; i < ns; i += PQ_THREADS) mine += skey(i) >= bound0 ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    __syncthreads();
    if (tid == 0) hist[258] = 0;
    __syncthreads();
    if (lane == 0 && mine) atomicAdd(&hist[258], (unsigned)mine);
    __syncthreads();
    const int n_ge = (int)hist[258];                                          // sample keys at or above the row's bound
    unsigned T = bound0;
    if (n_ge >= a.k) {
        const unsigned kth = pq_select_kth(skey, ns, a.k, hist);
        if (kth > T) T = kth;
        if (tid == 0 && kth > bound0) atomicMax(&a.bound[r], kth);
    }
    if (n_ge > 0 || ns < n) {                                                 // (a sample without a key at the bound says nothing about the rest)
        for (int i = tid; i < n; i += PQ_THREADS) {
            const unsigned key = keys_s[i];
            if (key >= T) {
                const unsigned slot = atomicAdd(&a.cand_count[r], 1u);
                if (slot < (unsigned)a.cand_cap) a.cand[(size_t)r * a.cand_cap + slot] = make_uint2(key, (unsigned)pos_of(i));
                else a.overflow[r] = 1u;
            }
        }
    }
}

// Two codes at a time: all code bytes of both (up to 2 x 8 uint4) are loaded BEFORE the look-up chains start, and the two chains --
// each the same sequential fp32 sum as pq_adc_sum -- are interleaved.  pq_adc_sum loads 16 code bytes, walks 16 dependent LDS gathers,
// loads the next 16: six exposed memory latencies per code of an OPQ96 index, and one gather chain per thread in flight; the
// row-major scan of the released shape spent 300 us per batch of 64 that way.
// NG = uint4 per code the kernel is compiled for (6: M <= 96, the released OPQ96; 8: up to M = 128), ng = M / 16 <= NG.
template <int NG>
__device__ __forceinline__ void pq_load_code(const uint8_t* __restrict__ codes, int64_t pos, int M, int ng, uint4 (&c)[NG]) {
    const uint4* cp = (const uint4*)(codes + (size_t)pos * M);
#pragma unroll
    for (int g = 0; g < NG; ++g) c[g] = g < ng ? cp[g] : make_uint4(0u, 0u, 0u, 0u);
}
template <int NG>
__device__ __forceinline__ void pq_adc_sum2(const uint4 (&c0)[NG], const uint4 (&c1)[NG], int ng, const float* lut_s, float& acc0, float& acc1) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g < ng) {
            const unsigned w0[4] = {c0[g].x, c0[g].y, c0[g].z, c0[g].w}, w1[4] = {c1[g].x, c1[g].y, c1[g].z, c1[g].w};
#pragma unroll
            for (int w = 0; w < 4; ++w)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int m = g * 16 + w * 4 + b;
                    acc0 = __fadd_rn(acc0, lut_s[m * 256 + ((w0[w] >> (8 * b)) & 255u)]);
                    acc1 = __fadd_rn(acc1, lut_s[m * 256 + ((w1[w] >> (8 * b)) & 255u)]);
                }
        }
    }
}

// One code: the sixteen table entries of a uint4 of code bytes are read TOGETHER, then added in order (the same sequential fp32 sum).
// Left to the scheduler every ds_read was followed by s_waitcnt lgkmcnt(0) and its add -- 96 exposed LDS latencies per code.
template <int NG, bool GL>
__device__ __forceinline__ float pq_adc_sum1(const uint4 (&c)[NG], int ng, const float* lut_s, const float* __restrict__ lut_g, float acc) {
    // GL (M = 96 only, tuning key "pq_split_lut", OFF by default): the sixteen entries of the LAST uint4 (m = 80 .. 95) come from the
    // table's copy in global memory -- through the vector L1, a data path of its own -- while the LDS serves the other eighty: the LDS
    // is the bottleneck of this loop (6 cycles per gather instruction, 65 % of them bank-conflict replays) and a sixth of its work
    // would leave it.  Same values, same order -- but measured slower: 64 divergent dwords cost the texture path more than the bank
    // conflicts cost the LDS (look-up sums 148 us per workgroup against 124, 0.754 ms per batch against 0.724).  Kept as the record.
    float vg[16];
    if constexpr (GL) {
        const unsigned w[4] = {c[NG - 1].x, c[NG - 1].y, c[NG - 1].z, c[NG - 1].w};
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int b = 0; b < 4; ++b) vg[q * 4 + b] = lut_g[((NG - 1) * 16 + q * 4 + b) * 256 + ((w[q] >> (8 * b)) & 255u)];
    }
#pragma unroll
    for (int g = 0; g < (GL ? NG - 1 : NG); ++g) {
        if (g < ng) {
            const unsigned w[4] = {c[g].x, c[g].y, c[g].z, c[g].w};
            float v[16];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int b = 0; b < 4; ++b) v[q * 4 + b] = lut_s[(g * 16 + q * 4 + b) * 256 + ((w[q] >> (8 * b)) & 255u)];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc = __fadd_rn(acc, v[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if constexpr (GL) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc = __fadd_rn(acc, vg[i]);
    }
    return acc;
}

// LIST-MAJOR work (long lists): (list, row) pairs in list order -- the rows probing a list scan it one after the other
// while its codes are hot in L2 / MALL.
template <int NG>
__global__ __launch_bounds__(PQ_THREADS, 1) void pq_adc_kernel(pq_scan_args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pq_smem[];
    const int M = a.M;
    float* const lut_s = (float*)pq_smem;                                  // [M][256]
    unsigned* const keys_s = (unsigned*)(lut_s + (size_t)M * 256);          // [seg]
    unsigned* const hist = keys_s + a.seg;                                 // [256 + 8]
    double* const red = (double*)(hist + 264);                             // [16] wave partials
    int* const cur = (int*)(red + 16);                                     // [2] the pair taken from the queue
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_pairs = min(*a.n_pairs, a.pair_cap);
    for (;;) {
        __syncthreads();
        if (tid == 0) cur[0] = atomicAdd(a.next, 1);
        __syncthreads();
        const int p = cur[0];
        if (p >= n_pairs) break;
        const int2 pr = a.pairs[p];
        const int l = pr.x, r = pr.y & 1023, chunk = pr.y >> 10;
        // the row's table into LDS, and dis0 = fl32(<x'_r, c_l>) in float64 (IVFPQ by_residual with METRIC_INNER_PRODUCT)
        const float4* src = (const float4*)(a.lut + (size_t)r * M * 256);
        for (int i = tid; i < M * 64; i += PQ_THREADS) ((float4*)lut_s)[i] = src[i];
        double part = 0.0;
        if (a.by_residual && tid < DPH_DIM) part = (double)a.xp[(size_t)r * DPH_DIM + tid] * (double)a.cent[(size_t)l * DPH_DIM + tid];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if (lane == 0) red[wave] = part;
        __syncthreads();
        double d0 = 0.0;
        for (int w = 0; w < DPH_DIM / 64; ++w) d0 += red[w];
        const float dis0 = (float)d0;
        const int64_t begin = a.list_off[l], len = min(a.list_off[l + 1] - begin, (int64_t)(chunk + 1) * a.seg * PQ_LM_CHUNK);
        for (int64_t s0 = (int64_t)chunk * a.seg * PQ_LM_CHUNK; s0 < len; s0 += a.seg) {
            const int n = (int)min((int64_t)a.seg, len - s0);
            __syncthreads();                                               // keys_s / hist of the previous segment are done with
            for (int i = tid; i < n; i += 2 * PQ_THREADS) {
                const int i1 = i + PQ_THREADS < n ? i + PQ_THREADS : i;         // (a lone code is summed twice)
                uint4 c0[NG], c1[NG];
                pq_load_code(a.codes, begin + s0 + i, M, M / 16, c0);
                pq_load_code(a.codes, begin + s0 + i1, M, M / 16, c1);
                float acc0 = dis0, acc1 = dis0;
                pq_adc_sum2(c0, c1, M / 16, lut_s, acc0, acc1);
                keys_s[i] = pq_key(acc0);
                keys_s[i1] = pq_key(acc1);
            }
            __syncthreads();
            pq_segment_finish<false>(a, r, keys_s, n, hist, [&](int i) { return begin + s0 + i; });
        }
    }
}

// ROW-MAJOR work (many short lists -- the reference's 2^20): a group = (query row, PQ_GROUP of its probed lists).  The lists'
// codes are ONE virtual sequence (prefix sums of their lengths in LDS, a binary search per code), so a segment of 12288 codes
// spans dozens of lists and the bound / select step runs once per segment, not once per 160-code list.
// The work is cut by CODE COUNT (round 6): a unit = one segment of <= seg codes of a group's sequence.  A k-means quantizer over
// token vectors (build_phrase_index.py:113-116,156-279) leaves lists hundreds of times the mean, and they are the most probed:
// with one workgroup per group a 600 k-code list kept ONE workgroup busy for 3 ms while 255 idled
// (profiles/r05_pq_ivf1M_b256_one_giant_list_phases.json).  pq_units_kernel writes the (group, segment) records once the probe
// lists are known; a workgroup of the scan pops a record, loads the row's table, rebuilds the group's prefix sums (64 offsets)
// and scores its segment only -- dis0 of just the lists that segment touches.  A uniform index gives one unit per group: the
// same work as before.
#define PQ_GROUP 64
// one wave per (row, group): total codes of the group's lists -> ceil(total / seg) unit records (none for an empty group)
__global__ __launch_bounds__(256) void pq_units_kernel(const int* __restrict__ probe, int probe_stride, int units_per_row, int n_groups,
                                                       const int64_t* __restrict__ list_off, int seg, int2* __restrict__ units,
                                                       int* __restrict__ n_units, int unit_cap, unsigned* __restrict__ overflow) {
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= n_groups) return;
    const int r = p / units_per_row, g0 = (p - r * units_per_row) * PQ_GROUP;
    const int l = g0 + lane < probe_stride ? probe[(size_t)r * probe_stride + g0 + lane] : -1;
    long long tot = l >= 0 ? (long long)(list_off[l + 1] - list_off[l]) : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
    const int nseg = (int)((tot + seg - 1) / seg);
    if (nseg == 0) return;
    int base = 0;
    if (lane == 0) base = atomicAdd(n_units, nseg);
    base = __shfl(base, 0);
    for (int s_ = lane; s_ < nseg; s_ += 64) {
        if (base + s_ < unit_cap) units[base + s_] = make_int2(p, s_);
        else overflow[r] = 1u;                                     // (cannot happen: the capacity is sized from the longest lists)
    }
}
template <int NG>
__global__ __launch_bounds__(PQ_THREADS, 1) void pq_adc_rows_kernel(pq_scan_args a, const int* __restrict__ probe, int probe_stride,
                                                                    int units_per_row) {
    extern __shared__ __attribute__((aligned(16))) unsigned char pq_smem[];
    const int M = a.M;
    float* const lut_s = (float*)pq_smem;
    unsigned* const keys_s = (unsigned*)(lut_s + (size_t)M * 256);
    unsigned* const hist = keys_s + a.seg;
    double* const red = (double*)(hist + 264);                             // (unused here; keeps the layout of pq_adc_kernel)
    int* const cur = (int*)(red + 16);
    __shared__ int g_list[PQ_GROUP];
    __shared__ long long g_beg[PQ_GROUP];
    __shared__ int g_pre[PQ_GROUP + 1];
    __shared__ float g_dis0[PQ_GROUP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // phase clock of thread 0 (100 MHz wall clock), only when a.prof is set: [start, end, table + lists, dis0, sums, select + append, units, codes]
    const bool prof = a.prof != nullptr && tid == 0;
    __shared__ unsigned long long pt[9];                                   // ([8]: the last stamp)
    if (prof) { for (int i = 1; i < 8; ++i) pt[i] = 0; pt[0] = pt[8] = wall_clock64(); }
    auto lap = [&](int slot) __attribute__((always_inline)) { if (prof) { const unsigned long long t = wall_clock64(); pt[slot] += t - pt[8]; pt[8] = t; } };
    const int n_units = min(*a.n_units, a.unit_cap);
    for (;;) {
        __syncthreads();
        if (tid == 0) cur[0] = atomicAdd(a.next, 1);
        __syncthreads();
        const int u = cur[0];
        if (u >= n_units) break;
        if (prof) pt[8] = wall_clock64();
        const int2 urec = a.units[u];
        const int r = urec.x / units_per_row, g0 = (urec.x - r * units_per_row) * PQ_GROUP;
        const float4* src = (const float4*)(a.lut + (size_t)r * M * 256);
        const float* const lut_g = a.lut + (size_t)r * M * 256;
        const bool gl = a.split_lut && M == NG * 16;            // (wave-uniform)
        for (int i = tid; i < M * 64; i += PQ_THREADS) ((float4*)lut_s)[i] = src[i];
        if (tid < PQ_GROUP) {
            const int l = g0 + tid < probe_stride ? probe[(size_t)r * probe_stride + g0 + tid] : -1;
            g_list[tid] = l;
            const long long beg = l >= 0 ? (long long)a.list_off[l] : 0;
            g_beg[tid] = beg;
            // the lists' lengths by 64 threads at once (one thread walking them paid 64 global round trips: ~30 us per unit), then
            // their exclusive prefix sums over the wave
            const int len = l >= 0 ? (int)((long long)a.list_off[l + 1] - beg) : 0;
            int incl = len;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (tid >= o) incl += v; }
            g_pre[tid] = incl - len;
            if (tid == PQ_GROUP - 1) g_pre[PQ_GROUP] = incl;
        }
        __syncthreads();                                                   // g_list is read by every wave below
        lap(2);
        const int total = g_pre[PQ_GROUP];
        const int s0 = urec.y * a.seg, n = min(a.seg, total - s0);       // this unit's segment of the group's sequence (n >= 1: pq_units_kernel)
        // dis0 of the lists this segment touches, float64: a wave takes two lists per trip and has their 2 x 12 centroid values in
        // flight together (one list after the other, each load waited for: 22 us per unit, a sixth of the kernel); sums as before --
        // t ascending per lane, then the butterfly
        {
            float xv[DPH_DIM / 64];
#pragma unroll
            for (int i = 0; i < DPH_DIM / 64; ++i) xv[i] = a.xp[(size_t)r * DPH_DIM + lane + 64 * i];
            for (int j0 = wave * 2; j0 < PQ_GROUP; j0 += (PQ_THREADS / 64) * 2) {
                // (wave-uniform: LDS values) lists j0, j0 + 1 hold virtual codes [pre[j0], pre[j0 + 2]) -- skip the pair when the segment lies elsewhere
                if (g_pre[j0] >= s0 + n || g_pre[j0 + 2 < PQ_GROUP ? j0 + 2 : PQ_GROUP] <= s0) continue;
                float cv[2][DPH_DIM / 64];
                bool on[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int l = __builtin_amdgcn_readfirstlane(j0 + u < PQ_GROUP ? g_list[j0 + u] : -1);      // (wave-uniform: a scalar row base)
                    const float* crow = a.cent + (size_t)(l >= 0 ? l : 0) * DPH_DIM;          // (an empty slot reads list 0 and is zeroed below: no branch around the loads)
#pragma unroll
                    for (int i = 0; i < DPH_DIM / 64; ++i) cv[u][i] = crow[lane + 64 * i];
                    on[u] = a.by_residual && l >= 0;
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    double part = 0.0;
#pragma unroll
                    for (int i = 0; i < DPH_DIM / 64; ++i) part += (double)xv[i] * (double)cv[u][i];
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
                    if (lane == 0 && j0 + u < PQ_GROUP) g_dis0[j0 + u] = on[u] ? (float)part : 0.f;
                }
            }
        }
        __syncthreads();
        lap(3);
        if (prof) { pt[6] += 1; pt[7] += (unsigned long long)n; }
        auto locate = [&](int v, int& j) {                                 // list of virtual code v: last j with pre[j] <= v
            int lo = 0, hi = PQ_GROUP - 1;
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (g_pre[mid] <= v) lo = mid; else hi = mid - 1; }
            j = lo;
            return (int64_t)g_beg[lo] + (v - g_pre[lo]);
        };
        {
            {
                // one code per thread and trip, the NEXT trip's code bytes already on their way while this one's look-ups run (two codes
                // per trip, loaded and then summed: every trip began with an exposed HBM round trip -- random 96-byte reads)
                int i = tid, jn = 0;
                uint4 cn[NG];
                if (i < n) pq_load_code(a.codes, locate(s0 + i, jn), M, M / 16, cn);
                for (; i < n; i += PQ_THREADS) {
                    uint4 cc[NG];
#pragma unroll
                    for (int g = 0; g < NG; ++g) cc[g] = cn[g];
                    const int jc = jn;
                    if (i + PQ_THREADS < n) pq_load_code(a.codes, locate(s0 + i + PQ_THREADS, jn), M, M / 16, cn);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (NG == 6) {           // (the released OPQ96; the M = 128 build has no registers to spare)
                        keys_s[i] = pq_key(gl ? pq_adc_sum1<NG, true>(cc, NG, lut_s, lut_g, g_dis0[jc]) : pq_adc_sum1<NG, false>(cc, M / 16, lut_s, lut_g, g_dis0[jc]));
                    } else {
                        keys_s[i] = pq_key(pq_adc_sum1<NG, false>(cc, M / 16, lut_s, lut_g, g_dis0[jc]));
                    }
                }
            }
            __syncthreads();
            lap(4);
            pq_segment_finish<true>(a, r, keys_s, n, hist, [&](int i) { int j; return locate(s0 + i, j); });
            if (a.prof) { __syncthreads(); lap(5); }
        }
    }
    if (prof) {
        pt[1] = wall_clock64();
#pragma unroll
        for (int i = 0; i < 8; ++i) a.prof[(size_t)blockIdx.x * 8 + i] = pt[i];
    }
}

// ---------------------------------------------------------------------------------------------- exact top-k of a row's candidates
__global__ __launch_bounds__(PQ_THREADS) void pq_final_kernel(const uint2* __restrict__ cand, const unsigned* __restrict__ cand_count,
                                                              int cand_cap, const unsigned* __restrict__ overflow,
                                                              const int64_t* __restrict__ ids, int q0, int k, float* __restrict__ D,
                                                              int64_t* __restrict__ I, int32_t* __restrict__ status,
                                                              const int32_t* __restrict__ nonfinite) {
    __shared__ unsigned hist[264];
    __shared__ unsigned skey[PQ_FINAL_CAP];
    __shared__ long long sid[PQ_FINAL_CAP];
    __shared__ unsigned n_keep;
    const int r = blockIdx.x, tid = threadIdx.x;
    if (nonfinite && nonfinite[r]) {                 // (block-uniform) the stand-in of a NaN / Inf row was searched; its answer is FAISS'
        for (int j = tid; j < k; j += PQ_THREADS) { D[(int64_t)(q0 + r) * k + j] = -FLT_MAX; I[(int64_t)(q0 + r) * k + j] = -1; }
        if (tid == 0) status[q0 + r] = DPH_ROW_NONFINITE;
        return;
    }
    const uint2* c = cand + (size_t)r * cand_cap;
    const int n = (int)min(cand_count[r], (unsigned)cand_cap);
    bool bad = overflow[r] != 0u;
    // k-th largest key of the candidates (all of them when fewer than k).  (One thread walking the 256 bins of each of the four
    // passes was 40 of this kernel's 48 us.)
    unsigned T = 0;
    if (n >= k) T = pq_select_kth([&](int i) { return c[i].x; }, n, k, hist);
    if (tid == 0) n_keep = 0;
    __syncthreads();
    for (int i = tid; i < n; i += PQ_THREADS) {
        const uint2 e = c[i];
        if (e.x >= T) {
            const unsigned slot = atomicAdd(&n_keep, 1u);
            if (slot < PQ_FINAL_CAP) { skey[slot] = e.x; sid[slot] = ids[e.y]; }
        }
    }
    __syncthreads();
    int m = (int)n_keep;
    if (m > PQ_FINAL_CAP) { bad = true; m = PQ_FINAL_CAP; }
    int pow2 = 1;
    while (pow2 < m) pow2 <<= 1;
    for (int i = m + tid; i < pow2; i += PQ_THREADS) { skey[i] = 0u; sid[i] = 0x7FFFFFFFFFFFFFFFll; }
    __syncthreads();
    // bitonic sort, (key desc, id asc)
    for (int size = 2; size <= pow2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < pow2 / 2; i += PQ_THREADS) {
                const int lo = 2 * i - (i & (stride - 1)), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const unsigned ka = skey[lo], kb = skey[hi];
                const long long ia = sid[lo], ib = sid[hi];
                const bool a_first = ka > kb || (ka == kb && ia < ib);
                if (a_first != up) { skey[lo] = kb; skey[hi] = ka; sid[lo] = ib; sid[hi] = ia; }
            }
            __syncthreads();
        }
    for (int j = tid; j < k; j += PQ_THREADS) {
        const int64_t o = (int64_t)(q0 + r) * k + j;
        if (j < m) { D[o] = pq_unkey(skey[j]); I[o] = sid[j]; }
        else { D[o] = -FLT_MAX; I[o] = -1; }
    }
    if (tid == 0) status[q0 + r] = bad ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------- direct map, reconstruct
__device__ __forceinline__ int64_t pq_pos_of_id(const int64_t* dm_ids, const unsigned* dm_pos, int64_t n, int64_t id) {
    int64_t lo = 0, hi = n;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (dm_ids[mid] < id) lo = mid + 1; else hi = mid; }
    return (lo < n && dm_ids[lo] == id) ? (int64_t)dm_pos[lo] : -1;
}
__device__ __forceinline__ int pq_list_of_pos(const int64_t* list_off, int nlist, int64_t pos) {
    int lo = 0, hi = nlist - 1;                                  // last list whose offset is <= pos and that is not empty there
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (list_off[mid] <= pos) lo = mid; else hi = mid - 1; }
    return lo;
}
struct pq_store {
    const float* cent; const float* pqc; const uint8_t* codes; const int64_t* list_off; const int64_t* dm_ids; const unsigned* dm_pos;
    int64_t ntotal; int nlist; int M; int dsub; int by_residual;
};
// component j of the reconstruction of the code at `pos` (rotated space, fp32: decode, then one add of the centroid)
__device__ __forceinline__ float pq_component(const pq_store& s, int64_t pos, int list, int j) {
    const int m = j / s.dsub, t = j - m * s.dsub;
    float v = s.pqc[((int64_t)m * 256 + s.codes[pos * s.M + m]) * s.dsub + t];
    if (s.by_residual) v = __fadd_rn(v, s.cent[(int64_t)list * DPH_DIM + j]);
    return v;
}
__global__ __launch_bounds__(256) void pq_reconstruct_kernel(pq_store s, const int64_t* __restrict__ ids, int64_t n,
                                                             float* __restrict__ out, int32_t* __restrict__ found) {
    const int64_t c = blockIdx.x;
    const int64_t pos = pq_pos_of_id(s.dm_ids, s.dm_pos, s.ntotal, ids[c]);
    if (threadIdx.x == 0 && found) found[c] = pos >= 0 ? 1 : 0;
    const int list = pos >= 0 ? pq_list_of_pos(s.list_off, s.nlist, pos) : 0;
    for (int j = threadIdx.x; j < DPH_DIM; j += 256) out[c * DPH_DIM + j] = pos >= 0 ? pq_component(s, pos, list, j) : 0.f;
}

// ---------------------------------------------------------------------------------------------- window re-score (index.py:276-370, PQ branch)
// One wavefront per candidate, like dph_window_kernel, over RECONSTRUCTED vectors: slot row = reconstruct(id +- i) in the
// rotated space, zero when the id is unknown (index.py:285-288); the reference un-rotates it (`end.matmul(self.R)`, :340)
// and takes the fp32 dot with the query half -- <q, v' R> = <R q, v'> = <A q, v'>, so the dot is taken with the ROTATED
// query half qrot = A q (pq_transform_kernel without the bias) in float64 and rounded once, no un-rotation of 2*B*k*L vectors.
__global__ __launch_bounds__(256) void pq_window_kernel(
    int direction, pq_store s, dph_idmap idmap, const float* __restrict__ qrot, int64_t n_cand, int k, int L,
    const int64_t* __restrict__ ids, const int32_t* __restrict__ doc_in, const int32_t* __restrict__ word_in,
    const float* __restrict__ first, const int32_t* __restrict__ row2doc, const int32_t* __restrict__ row2word,
    const int32_t* __restrict__ doc_ids, int64_t n_docs, const int64_t* __restrict__ f2o_off, const int32_t* __restrict__ f2o,
    int32_t* __restrict__ pred_word, double* __restrict__ best, int32_t* __restrict__ argslot, float* __restrict__ vecs) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t c = (int64_t)blockIdx.x * 4 + (tid >> 6);
    if (c >= n_cand) return;
    const int64_t id = ids[c];
    int64_t lc = dph_local_of_id(idmap, id);
    const bool mine = lc >= 0;          // range-sharded: a candidate of another rank (FAISS padding, single rank: an unknown id)
    if (lc < 0) {
        const int64_t first_id = idmap.n_groups ? idmap.id_offsets[0] : idmap.id_base;
        lc = id < first_id ? 0 : idmap.n_ids - 1;
        if (lc < 0) lc = 0;
    }
    const int d = doc_in ? doc_in[c] : row2doc[lc];
    const int w = word_in ? word_in[c] : row2word[lc];
    int64_t lo = 0, hi = n_docs;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (doc_ids[mid] < d) lo = mid + 1; else hi = mid; }
    const bool have_doc = d >= 0 && lo < n_docs && doc_ids[lo] == d;
    const int64_t fbase = have_doc ? f2o_off[lo] : 0;
    const int64_t flen = have_doc ? f2o_off[lo + 1] - fbase : 0;
    float q[12];
    const float* qp = qrot + (c / k) * DPH_DIM + lane * 12;
#pragma unroll
    for (int j = 0; j < 12; ++j) q[j] = qp[j];
    double best_s = 0.0;
    int best_slot = -1, best_word = -1;
    const float f0 = first[c];
    // Where the window's codes lie: lane sl resolves slot sl -- the binary search of its id in the direct map (28 dependent loads at
    // 170 M codes) and of its position in the list offsets (20) -- for up to 64 slots AT ONCE; round 4 walked the slots one after the
    // other, 2 * B * k * L searches in series per wave: 127 us per direction at B = 64, k = 10, L = 10 (profiles/r05_kernel_trace_pq_e2e_b64.csv)
    int64_t my_pos = -1;
    int my_list = 0;
    for (int sl = 0; sl < L; ++sl) {
        if ((sl & 63) == 0) {
            const int mine_sl = sl + lane;
            my_pos = -1;
            my_list = 0;
            if (mine_sl < L) {
                const int i = direction == 0 ? mine_sl : (L - 1 - mine_sl);
                my_pos = pq_pos_of_id(s.dm_ids, s.dm_pos, s.ntotal, direction == 0 ? id + i : id - i);
                if (my_pos >= 0) my_list = pq_list_of_pos(s.list_off, s.nlist, my_pos);
            }
        }
        const int i = direction == 0 ? sl : (L - 1 - sl);
        const int64_t ww = direction == 0 ? (int64_t)w + i : (int64_t)w - i;
        const int64_t pos = __shfl(my_pos, sl & 63);
        const int list = __shfl(my_list, sl & 63);
        bool valid = have_doc && w >= 0 && w < flen && ww >= 0 && ww < flen;
        if (valid) {
            const int64_t gap = direction == 0 ? (int64_t)f2o[fbase + ww] - (int64_t)f2o[fbase + w]
                                               : (int64_t)f2o[fbase + w] - (int64_t)f2o[fbase + ww];
            valid = gap >= 0 && gap <= L;
        }
        double dot = 0.0;
        if (pos >= 0) {
#pragma unroll
            for (int j = 0; j < 12; ++j) dot += (double)q[j] * (double)pq_component(s, pos, list, lane * 12 + j);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
        }
        const double sc = (double)(f0 + (float)dot) + (valid ? 0.0 : -1e9);       // fp32 + fp32, then the float64 mask (:343 / :368)
        if (best_slot < 0 || sc > best_s) { best_s = sc; best_slot = sl; best_word = valid ? (int)ww : -1; }
    }
    if (lane == 0) { pred_word[c] = best_word; best[c] = best_s; argslot[c] = best_slot; }
    if (vecs) {
        // rotated-space vectors for now: [c,0,:] the candidate's own, [c,1,:] the arg-max slot's; pq_unrotate_kernel finishes them
        const int bi = direction == 0 ? best_slot : (L - 1 - best_slot);
        const int64_t idv[2] = {id, direction == 0 ? id + bi : id - bi};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            // a candidate this shard does not hold contributes ZERO vectors, also for a window slot that happens to lie in this shard's
            // range (the first rows of the next rank's range): the ranks' vectors are SUMMED (index.py _window_vectors), exactly one
            // rank -- the candidate's -- may speak
            const int64_t pos = mine ? pq_pos_of_id(s.dm_ids, s.dm_pos, s.ntotal, idv[t]) : -1;
            const int list = pos >= 0 ? pq_list_of_pos(s.list_off, s.nlist, pos) : 0;
            float* o = vecs + (c * 2 + t) * DPH_DIM + lane * 12;
#pragma unroll
            for (int j = 0; j < 12; ++j) o[j] = pos >= 0 ? pq_component(s, pos, list, lane * 12 + j) : 0.f;
        }
    }
}
// out[v, :] = in[v, :] @ R, R = A [768,768] row-major (index.py:340 `end.matmul(self.R)`, :381-389 `.dot(self.R)`), float64
// accumulation; `twice[v & 1]` vectors get it applied a second time (the reference's pred_*_vecs are taken from the already
// un-rotated window tensor and then multiplied by R again, :345,370 then :381-389 -- replicated as is)
__global__ __launch_bounds__(256) void pq_unrotate_kernel(float* __restrict__ vecs, const float* __restrict__ A, int64_t n_vecs) {
    __shared__ float v[DPH_DIM];
    const int64_t c = blockIdx.x;
    const int reps = (c & 1) ? 2 : 1;
    for (int rep = 0; rep < reps; ++rep) {
        __syncthreads();
        for (int j = threadIdx.x; j < DPH_DIM; j += 256) v[j] = vecs[c * DPH_DIM + j];
        __syncthreads();
        for (int j = threadIdx.x; j < DPH_DIM; j += 256) {
            double acc = 0.0;
            for (int t = 0; t < DPH_DIM; ++t) acc += (double)v[t] * (double)A[(int64_t)t * DPH_DIM + j];
            vecs[c * DPH_DIM + j] = (float)acc;
        }
    }
}

__global__ __launch_bounds__(256) void pq_iota_kernel(unsigned* p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = (unsigned)i;
}
__global__ __launch_bounds__(256) void pq_dup_kernel(const int64_t* sorted_ids, int64_t n, int* dup) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i + 1 < n && sorted_ids[i] == sorted_ids[i + 1]) *dup = 1;
}

// ============================================================================================== host side
void dph_pq_free(dph_pq* p);
int dph_sort_pairs_i64_u32(const int64_t* keys_in, const unsigned* vals_in, int64_t* keys_out, unsigned* vals_out, int64_t n, hipStream_t st);

struct dph_pq {
    int device = 0, nlist = 0, M = 0, dsub = 0, by_residual = 1;
    int64_t ntotal = 0;
    float *A = nullptr, *At = nullptr, *b = nullptr, *cent = nullptr, *pqc = nullptr;
    unsigned* cent_pk = nullptr;                           // bf16 hi << 16 | lo of the centroids (long quantizers: the bf16x3 coarse GEMM)
    unsigned* xp_pk = nullptr;                             // ... and of the rotated query rows of a pass (scratch)
    unsigned short* cent_hi = nullptr;                     // the centroids as plain bf16: the coarse quantizer's one-product filter GEMM
    unsigned short* cent_frag = nullptr;                   // ... once more in MFMA fragment order (filter GEMM variant 3)
    unsigned short* cent_pieces = nullptr;                 // ... and as 24 KiB pieces with the byte layout of an int8 tile (the filter SCAN, variant 5)
    unsigned short* xp_hi = nullptr;                       // ... and the rotated query rows of a pass (scratch)
    void* coarse_cf = nullptr;                             // scratch of the filter form (dph_launch_coarse_filter)
    int split_lut = 0;                                     // tuning key "pq_split_lut": measured SLOWER (sums 148 us against 124 per workgroup), off
    int coarse_filter = 5;                                 // 0: the bf16x3 chain alone, 1 / 2: filter GEMM staging centroids and queries through LDS (2: centroid stream
                                                           // non-temporal), 3: centroids straight into registers from the fragment-major image, 4: 3 on contiguous tile runs,
                                                           // 5 (default, round 5): the filter as a SCAN -- bf16 centroid pieces through the flat scan's feed (dph_scan.hip MODE 3)
    int coarse_teams = 1;                                  // tuning key "coarse_teams": passes of more than 128 rows run the filter scan as ONE launch of workgroup teams
    bool profile = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events, prof_free;
    std::vector<float> h_A;
    uint8_t* codes = nullptr; int64_t* ids = nullptr; int64_t* list_off = nullptr;
    std::vector<int64_t> h_list_off;
    int64_t* dm_ids = nullptr; unsigned* dm_pos = nullptr;
    double cnorm_max = 0.0, cnorm_cap = 0.0;               // longest centroid; the length above which a list is HEAVY for the coarse filter (dph_coarse_select_kernel)
    float* cnorm = nullptr;                                // [nlist] ||c_l||_2
    int64_t max_list = 0;
    bool lists_set = false, params_set = false, finalized = false;
    // scratch, grown on demand
    int cap_rows = 0, cap_k = 0, cap_nprobe = 0;
    float *xp = nullptr, *lut = nullptr, *scores = nullptr, *qrot = nullptr;
    unsigned *listmask = nullptr, *bound = nullptr, *cand_count = nullptr, *overflow = nullptr;
    int2* pairs = nullptr; int* counters = nullptr; uint2* cand = nullptr; int cand_cap = 0; int pair_cap = 0;
    int* probe = nullptr;                                  // [rows][nprobe] probed lists of every row (row-major scan)
    int32_t* nf = nullptr;                                 // [rows] query rows with a NaN / Inf element (pq_transform_kernel; lives behind `counters`)
    int2* units = nullptr; int unit_cap = 0;               // ... and its work queue: (group, segment) records (pq_units_kernel)
    std::vector<int64_t> h_top_prefix;                     // [i] = codes in the i longest lists: what a row can meet at most with nprobe = i
    int64_t qrot_rows = 0;
    void* coarse_cs = nullptr;                             // candidate scratch of the one-pass probe selection (dph_launch_coarse_presplit)
    unsigned long long* phase_prof = nullptr;              // dph_debug_pq_phases: [workgroups][8], allocated by the first call
    int phase_wgs = 0;
};

static void pq_free_scratch(dph_pq* p) {
    void* v[] = {p->xp, p->lut, p->scores, p->listmask, p->pairs, p->counters, p->cand, p->probe, p->xp_pk, p->xp_hi, p->units};      // (bound / cand_count / overflow live behind counters)
    for (void* q : v) if (q) (void)hipFree(q);
    p->xp = p->lut = p->scores = nullptr; p->listmask = p->bound = p->cand_count = p->overflow = nullptr; p->nf = nullptr;
    p->pairs = nullptr; p->counters = nullptr; p->cand = nullptr; p->probe = nullptr; p->xp_pk = nullptr; p->xp_hi = nullptr; p->units = nullptr; p->cap_rows = 0;
}

int dph_pq_alloc(dph_pq** out, int device, int64_t ntotal, int nlist, int M) {
    if (!out || ntotal < 0 || nlist <= 0 || nlist > (1 << 20) || M <= 0 || DPH_DIM % M || M % 16 || M > 128)
        return pq_fail(DPH_E_ARG, "PQ index: need 0 < nlist <= 2^20 and M a multiple of 16 that divides 768, M <= 128");
    if (ntotal >= (int64_t)0xFFFFFFF0ll) return pq_fail(DPH_E_ARG, "PQ index: at most 2^32-16 codes per GPU");
    PQCHK(hipSetDevice(device));
    dph_pq* p = new dph_pq();
    p->device = device; p->ntotal = ntotal; p->nlist = nlist; p->M = M; p->dsub = DPH_DIM / M;
    const size_t nt = (size_t)(ntotal > 0 ? ntotal : 1);
    if (hipMalloc((void**)&p->cent, (size_t)nlist * DPH_DIM * 4) != hipSuccess || hipMalloc((void**)&p->pqc, (size_t)256 * DPH_DIM * 4) != hipSuccess ||
        hipMalloc((void**)&p->codes, nt * M + 16) != hipSuccess || hipMalloc((void**)&p->ids, nt * 8) != hipSuccess ||
        hipMalloc((void**)&p->list_off, ((size_t)nlist + 1) * 8) != hipSuccess || hipMalloc((void**)&p->dm_ids, nt * 8) != hipSuccess ||
        hipMalloc((void**)&p->dm_pos, nt * 4) != hipSuccess) {
        dph_pq_free(p);
        return pq_fail(DPH_E_NOMEM, "PQ index: hipMalloc failed");
    }
    *out = p;
    return DPH_OK;
}

void dph_pq_set_split_lut(dph_pq* p, int on) { if (p) p->split_lut = on ? 1 : 0; }
void dph_pq_set_coarse_filter(dph_pq* p, int on) { if (p) p->coarse_filter = on < 0 ? 0 : (on > 5 ? 5 : on); }
void dph_pq_set_coarse_teams(dph_pq* p, int on) { if (p) p->coarse_teams = on ? 1 : 0; }
int dph_pq_coarse_debug(dph_pq* p, unsigned out[2]) {
    if (!p) return pq_fail(DPH_E_ARG, "null");
    PQCHK(hipSetDevice(p->device));
    PQCHK(hipDeviceSynchronize());
    return dph_coarse_filter_debug(p->coarse_cf, out) ? pq_fail(DPH_E_HIP, "coarse debug: copy failed") : DPH_OK;
}
int dph_pq_coarse_debug_pool(dph_pq* p, unsigned* lk_host, unsigned short* q_host, long long cap, long long* count) {
    if (!p || !count) return pq_fail(DPH_E_ARG, "null");
    PQCHK(hipSetDevice(p->device));
    PQCHK(hipDeviceSynchronize());
    *count = dph_coarse_filter_debug_pool(p->coarse_cf, lk_host, q_host, cap);
    return *count < 0 ? pq_fail(DPH_E_HIP, "coarse debug: copy failed") : DPH_OK;
}
// what the LAST pass left in the scan's bookkeeping: info[8] = {counters[0..3] (pairs, next, units, -), cand_cap, unit_cap, pair_cap, rows of
// the scratch}, per_row[n][3] = {appended candidates, overflow flag, bound key} of the first n rows.  (tests/test_fuzz_gpu.py prints it
// for a round that comes back uncertified.)
int dph_pq_debug_pass(dph_pq* p, int* info, unsigned* per_row, int n) {
    if (!p || !info || n < 0) return pq_fail(DPH_E_ARG, "null");
    if (!p->counters) return pq_fail(DPH_E_STATE, "PQ debug: no search yet");
    PQCHK(hipSetDevice(p->device));
    PQCHK(hipDeviceSynchronize());
    PQCHK(hipMemcpy(info, p->counters, 16, hipMemcpyDeviceToHost));
    info[4] = p->cand_cap; info[5] = p->unit_cap; info[6] = p->pair_cap; info[7] = p->cap_rows;
    n = std::min(n, p->cap_rows);
    std::vector<unsigned> a((size_t)n), b((size_t)n), c((size_t)n);
    if (n > 0 && per_row) {
        PQCHK(hipMemcpy(a.data(), p->cand_count, (size_t)n * 4, hipMemcpyDeviceToHost));
        PQCHK(hipMemcpy(b.data(), p->overflow, (size_t)n * 4, hipMemcpyDeviceToHost));
        PQCHK(hipMemcpy(c.data(), p->bound, (size_t)n * 4, hipMemcpyDeviceToHost));
        for (int r = 0; r < n; ++r) { per_row[3 * r] = a[(size_t)r]; per_row[3 * r + 1] = b[(size_t)r]; per_row[3 * r + 2] = c[(size_t)r]; }
    }
    return DPH_OK;
}
// phase clock of the row-major ADC scan: the first call (out = null) arms it, later calls copy the last launch's [workgroups][8]
// 100 MHz ticks: start, end, table + lists, dis0, sums, select + append, units, codes.  Returns the number of workgroups.
// which = 1: dph_coarse_select_kernel's stamps instead (dph_ivf.hip).
int dph_pq_debug_phases(dph_pq* p, int which, unsigned long long* out, int cap_wgs) {
    if (!p) return -1;
    if (hipSetDevice(p->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return -1;
    if (which == 1) return dph_coarse_select_clock(out, cap_wgs);        // the probe selection's stamps, one record per query row
    if (!p->phase_prof) {
        int cus = 256;
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, p->device);
        if (hipMalloc((void**)&p->phase_prof, (size_t)cus * 64) != hipSuccess) { p->phase_prof = nullptr; return -1; }
        (void)hipMemset(p->phase_prof, 0, (size_t)cus * 64);
        p->phase_wgs = cus;
    }
    const int n = cap_wgs < p->phase_wgs ? cap_wgs : p->phase_wgs;
    if (out && n > 0 && hipMemcpy(out, p->phase_prof, (size_t)n * 64, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return p->phase_wgs;
}
int dph_pq_profile(dph_pq* p, int on) {
    if (!p) return pq_fail(DPH_E_ARG, "null");
    PQCHK(hipSetDevice(p->device));
    p->profile = on != 0;
    while (p->profile && p->prof_free.size() < 64) {
        std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
        PQCHK(hipEventCreate(&ev.first));
        PQCHK(hipEventCreate(&ev.second));
        p->prof_free.push_back(ev);
    }
    return DPH_OK;
}
int dph_pq_profile_read(dph_pq* p, double* ms_total, int* launches) {
    if (!p || !ms_total || !launches) return pq_fail(DPH_E_ARG, "null");
    PQCHK(hipSetDevice(p->device));
    double total = 0.0;
    int cnt = 0;
    for (auto& ev : p->prof_events) {
        PQCHK(hipEventSynchronize(ev.second));
        float ms = 0.f;
        PQCHK(hipEventElapsedTime(&ms, ev.first, ev.second));
        total += ms;
        ++cnt;
        p->prof_free.push_back(ev);
    }
    p->prof_events.clear();
    *ms_total = total;
    *launches = cnt;
    return DPH_OK;
}

int dph_pq_profile_read_each(dph_pq* p, double* ms_out, int cap, int* n_out) {
    if (!p || !n_out) return pq_fail(DPH_E_ARG, "null");
    PQCHK(hipSetDevice(p->device));
    int n = 0;
    for (auto& ev : p->prof_events) {
        PQCHK(hipEventSynchronize(ev.second));
        float ms = 0.f;
        PQCHK(hipEventElapsedTime(&ms, ev.first, ev.second));
        if (n < cap && ms_out) ms_out[n] = ms;
        ++n;
        p->prof_free.push_back(ev);
    }
    p->prof_events.clear();
    *n_out = n;
    return DPH_OK;
}

void dph_pq_free(dph_pq* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    for (auto& ev : p->prof_events) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (auto& ev : p->prof_free) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    pq_free_scratch(p);
    void* v[] = {p->A, p->At, p->b, p->cent, p->pqc, p->codes, p->ids, p->list_off, p->dm_ids, p->dm_pos, p->qrot, p->cent_pk, p->coarse_cs, p->cent_hi, p->coarse_cf, p->cent_frag, p->cent_pieces, p->phase_prof, p->cnorm};
    for (void* q : v) if (q) (void)hipFree(q);
    delete p;
}

int dph_pq_set_params(dph_pq* p, const float* A, const float* b, const float* centroids, const float* pq_centroids, int by_residual) {
    if (!p || !centroids || !pq_centroids) return pq_fail(DPH_E_ARG, "PQ index: null parameter");
    PQCHK(hipSetDevice(p->device));
    if (p->A) { (void)hipFree(p->A); p->A = nullptr; }
    if (p->At) { (void)hipFree(p->At); p->At = nullptr; }
    if (p->b) { (void)hipFree(p->b); p->b = nullptr; }
    p->h_A.assign((size_t)DPH_DIM * DPH_DIM, 0.f);
    if (A) {
        p->h_A.assign(A, A + (size_t)DPH_DIM * DPH_DIM);
        std::vector<float> At((size_t)DPH_DIM * DPH_DIM);
        for (int i = 0; i < DPH_DIM; ++i) for (int j = 0; j < DPH_DIM; ++j) At[(size_t)j * DPH_DIM + i] = A[(size_t)i * DPH_DIM + j];
        PQCHK(hipMalloc((void**)&p->A, At.size() * 4));
        PQCHK(hipMalloc((void**)&p->At, At.size() * 4));
        PQCHK(hipMemcpy(p->A, A, At.size() * 4, hipMemcpyHostToDevice));
        PQCHK(hipMemcpy(p->At, At.data(), At.size() * 4, hipMemcpyHostToDevice));
    } else {
        for (int i = 0; i < DPH_DIM; ++i) p->h_A[(size_t)i * DPH_DIM + i] = 1.f;
    }
    if (b) {
        PQCHK(hipMalloc((void**)&p->b, DPH_DIM * 4));
        PQCHK(hipMemcpy(p->b, b, DPH_DIM * 4, hipMemcpyHostToDevice));
    }
    PQCHK(hipMemcpy(p->cent, centroids, (size_t)p->nlist * DPH_DIM * 4, hipMemcpyHostToDevice));
    PQCHK(hipMemcpy(p->pqc, pq_centroids, (size_t)256 * DPH_DIM * 4, hipMemcpyHostToDevice));
    if (p->nlist >= (1 << 16)) {                    // CG_BF16X3_MIN of dph_ivf.hip: the coarse GEMM of long quantizers runs on bf16 hi / lo parts
        if (!p->cent_pk) PQCHK(hipMalloc((void**)&p->cent_pk, (size_t)p->nlist * DPH_DIM * 4));
        dph_launch_bf16_split(p->cent, (int64_t)p->nlist * DPH_DIM, p->cent_pk, nullptr);
        if (!p->cent_hi) PQCHK(hipMalloc((void**)&p->cent_hi, (size_t)dph_bf16_hi_rows(p->nlist, 1) * DPH_DIM * 2));
        dph_launch_bf16_hi(p->cent, p->nlist, 1, p->cent_hi, nullptr);
        if (!p->cent_frag) PQCHK(hipMalloc((void**)&p->cent_frag, (size_t)dph_bf16_frag_rows(p->nlist) * DPH_DIM * 2));
        dph_launch_bf16_frag(p->cent, p->nlist, p->cent_frag, nullptr);
        if (!p->cent_pieces) PQCHK(hipMalloc((void**)&p->cent_pieces, (size_t)dph_bf16_piece_rows(p->nlist) * DPH_DIM * 2));
        dph_launch_bf16_pieces(p->cent, p->nlist, p->cent_pieces, nullptr);
        PQCHK(hipDeviceSynchronize());
    }
    double mx = 0.0;
    std::vector<float> cn((size_t)p->nlist);
    for (int l = 0; l < p->nlist; ++l) {
        double s = 0.0;
        for (int j = 0; j < DPH_DIM; ++j) s += (double)centroids[(size_t)l * DPH_DIM + j] * (double)centroids[(size_t)l * DPH_DIM + j];
        mx = std::max(mx, s);
        cn[(size_t)l] = (float)(sqrt(s) * (1.0 + 1e-6));          // (rounded up: it bounds an error)
    }
    p->cnorm_max = sqrt(mx);
    {
        // heavy lists: centroids longer than 1.1 x the 99-th percentile (none when the lengths are alike: the cap is then the maximum)
        std::vector<float> srt(cn);
        const size_t q99 = (size_t)((double)(p->nlist - 1) * 0.99);
        std::nth_element(srt.begin(), srt.begin() + (std::ptrdiff_t)q99, srt.end());
        p->cnorm_cap = std::min(p->cnorm_max, 1.1 * (double)srt[q99]);
        if (!p->cnorm) PQCHK(hipMalloc((void**)&p->cnorm, (size_t)p->nlist * 4));
        PQCHK(hipMemcpy(p->cnorm, cn.data(), (size_t)p->nlist * 4, hipMemcpyHostToDevice));
    }
    p->by_residual = by_residual ? 1 : 0;
    p->params_set = true;
    return DPH_OK;
}

int dph_pq_set_list_sizes(dph_pq* p, const int64_t* sizes) {
    if (!p || !sizes) return pq_fail(DPH_E_ARG, "PQ index: null list sizes");
    PQCHK(hipSetDevice(p->device));
    p->h_list_off.assign((size_t)p->nlist + 1, 0);
    p->max_list = 0;
    for (int l = 0; l < p->nlist; ++l) {
        if (sizes[l] < 0) return pq_fail(DPH_E_ARG, "PQ index: negative list size");
        p->h_list_off[(size_t)l + 1] = p->h_list_off[(size_t)l] + sizes[l];
        p->max_list = std::max(p->max_list, sizes[l]);
    }
    if (p->h_list_off[(size_t)p->nlist] != p->ntotal) return pq_fail(DPH_E_ARG, "PQ index: list sizes do not add up to ntotal");
    PQCHK(hipMemcpy(p->list_off, p->h_list_off.data(), ((size_t)p->nlist + 1) * 8, hipMemcpyHostToDevice));
    {
        std::vector<int64_t> srt(sizes, sizes + p->nlist);
        std::sort(srt.begin(), srt.end(), [](int64_t x, int64_t y) { return x > y; });
        p->h_top_prefix.assign((size_t)p->nlist + 1, 0);
        for (int l = 0; l < p->nlist; ++l) p->h_top_prefix[(size_t)l + 1] = p->h_top_prefix[(size_t)l] + srt[(size_t)l];
    }
    pq_free_scratch(p);                                    // (the scratch is sized from the list lengths)
    p->lists_set = true;
    p->finalized = false;
    return DPH_OK;
}

int dph_pq_upload(dph_pq* p, int64_t pos0, int64_t n, const uint8_t* codes, const int64_t* ids) {
    if (!p || n < 0 || pos0 < 0 || pos0 + n > p->ntotal || (n > 0 && (!codes || !ids))) return pq_fail(DPH_E_ARG, "PQ index: bad code range");
    PQCHK(hipSetDevice(p->device));
    if (n > 0) {
        PQCHK(hipMemcpy(p->codes + (size_t)pos0 * p->M, codes, (size_t)n * p->M, hipMemcpyHostToDevice));
        PQCHK(hipMemcpy(p->ids + pos0, ids, (size_t)n * 8, hipMemcpyHostToDevice));
    }
    p->finalized = false;
    return DPH_OK;
}

int dph_pq_finalize(dph_pq* p, hipStream_t st) {
    if (!p || !p->params_set || !p->lists_set) return pq_fail(DPH_E_STATE, "PQ index: parameters / list sizes not set");
    PQCHK(hipSetDevice(p->device));
    if (p->ntotal > 0) {
        unsigned* iota = nullptr;
        int* dup = nullptr;
        PQCHK(hipMalloc((void**)&iota, (size_t)p->ntotal * 4));
        if (hipMalloc((void**)&dup, 4) != hipSuccess) { (void)hipFree(iota); return pq_fail(DPH_E_NOMEM, "PQ index: hipMalloc"); }
        (void)hipMemsetAsync(dup, 0, 4, st);
        hipLaunchKernelGGL(pq_iota_kernel, dim3((unsigned)((p->ntotal + 255) / 256)), dim3(256), 0, st, iota, p->ntotal);
        int rc = dph_sort_pairs_i64_u32(p->ids, iota, p->dm_ids, p->dm_pos, p->ntotal, st);
        int dup_h = 0;
        if (rc == 0) {
            hipLaunchKernelGGL(pq_dup_kernel, dim3((unsigned)((p->ntotal + 255) / 256)), dim3(256), 0, st, p->dm_ids, p->ntotal, dup);
            if (hipMemcpyAsync(&dup_h, dup, 4, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) rc = 1;
        }
        (void)hipFree(iota); (void)hipFree(dup);
        if (rc) return pq_fail(DPH_E_HIP, "PQ index: building the direct map failed");
        if (dup_h) return pq_fail(DPH_E_ARG, "PQ index: duplicate ids in the inverted lists");
    }
    p->finalized = true;
    return DPH_OK;
}

bool dph_pq_ready(const dph_pq* p) { return p && p->finalized; }
int64_t dph_pq_ntotal(const dph_pq* p) { return p ? p->ntotal : 0; }
int dph_pq_nlist(const dph_pq* p) { return p ? p->nlist : 0; }
const float* dph_pq_A_host(const dph_pq* p) { return p ? p->h_A.data() : nullptr; }

static int pq_seg(const dph_pq* p) { return p->M <= 96 ? PQ_SEG : PQ_SEG / 2; }
// many short lists ON AVERAGE: the work is grouped by query row (pq_adc_rows_kernel) and cut by code count, so a list hundreds of times
// the mean is several units of several workgroups.  (A group's 64 lists are one sequence indexed with 32-bit ints: 2^24 codes per list.)
static bool pq_by_rows(const dph_pq* p) { return p->ntotal / p->nlist < 2048 && p->max_list < ((int64_t)1 << 24); }
// units / segments a query row can meet at most: one (partial) segment per group + the codes of its nprobe longest lists cut into segments
static int64_t pq_row_units_max(const dph_pq* p, int nprobe) {
    const int np_ = std::min(nprobe, p->nlist);
    const int64_t top = p->h_top_prefix.empty() ? p->ntotal : p->h_top_prefix[(size_t)np_];
    return (nprobe + PQ_GROUP - 1) / PQ_GROUP + top / pq_seg(p) + 1;
}
static size_t pq_lds_bytes(const dph_pq* p) { return (size_t)p->M * 1024 + (size_t)pq_seg(p) * 4 + 264 * 4 + 16 * 8 + 16; }

static int pq_ensure(dph_pq* p, int rows, int k, int nprobe) {
    if (rows <= p->cap_rows && k <= p->cap_k && nprobe <= p->cap_nprobe) return DPH_OK;
    rows = std::max(rows, p->cap_rows); k = std::max(k, p->cap_k); nprobe = std::max(nprobe, p->cap_nprobe);
    pq_free_scratch(p);
    // every segment may append its k best -- or, in the row-major scan (pq_segment_finish<true>: the bound of a long segment is the k-th of
    // a 1024-key sample), about k * seg / 1024 keys: twice that is reserved
    const bool by_rows = pq_by_rows(p);
    // (the scratch serves every later call with k' <= k without a new allocation: the sampled rule's share for min(k, 128) counts even when
    // THIS k is beyond it)
    const int64_t per_seg = by_rows ? std::max<int64_t>(k, (int64_t)std::min(k, PQ_THREADS / 8) * (pq_seg(p) / PQ_THREADS) * 2) : k;
    // segments a row can meet: row-major, (partial) segments of its groups -- from the nprobe LONGEST lists, not nprobe x the longest;
    // list-major, every probed list on its own
    const int64_t segs_row = by_rows ? pq_row_units_max(p, nprobe) : (int64_t)nprobe + (p->h_top_prefix.empty() ? p->ntotal : p->h_top_prefix[(size_t)std::min(nprobe, p->nlist)]) / pq_seg(p);
    // + 512: the count above the k-th of a 1-in-12 sample has a geometric tail -- for k = 1, P(more than m) = (11/12)^m: m = 64 was
    // crossed once in 4000 cold segments, 512 once in 10^19
    int64_t cap = segs_row * per_seg + 512;
    const int64_t budget = ((int64_t)2 << 30) / 8 / rows;                // at most 2 GiB of candidates per pass
    cap = std::max<int64_t>(std::min(cap, budget), 4 * (int64_t)k + 64);
    p->cand_cap = (int)std::min<int64_t>(cap, 1 << 30);
    // list-major pairs: one per chunk of PQ_LM_CHUNK segments of a probed list -- at most nprobe + (codes of the nprobe longest lists) / chunk per row
    p->pair_cap = (int)std::min<int64_t>((int64_t)rows * (nprobe + (by_rows ? 0 : (p->h_top_prefix.empty() ? p->ntotal : p->h_top_prefix[(size_t)std::min(nprobe, p->nlist)]) / ((int64_t)pq_seg(p) * PQ_LM_CHUNK))), 1 << 28);
    p->unit_cap = by_rows ? (int)std::min<int64_t>((int64_t)rows * pq_row_units_max(p, nprobe), 1 << 28) : 0;
    if (hipMalloc((void**)&p->xp, (size_t)rows * DPH_DIM * 4) != hipSuccess || hipMalloc((void**)&p->lut, (size_t)rows * p->M * 1024) != hipSuccess ||
        hipMalloc((void**)&p->scores, (size_t)rows * p->nlist * 4) != hipSuccess || hipMalloc((void**)&p->listmask, (size_t)p->nlist * DPH_UNIT_WORDS * 4) != hipSuccess ||
        hipMalloc((void**)&p->counters, 64 + (size_t)4 * rows * 4) != hipSuccess ||        // counters | bound | cand_count | overflow (one memset per pass) | non-finite flags
        hipMalloc((void**)&p->pairs, (size_t)p->pair_cap * 8) != hipSuccess ||
        hipMalloc((void**)&p->cand, (size_t)rows * p->cand_cap * 8) != hipSuccess ||
        hipMalloc((void**)&p->probe, (size_t)rows * nprobe * 4) != hipSuccess ||
        (p->unit_cap > 0 && hipMalloc((void**)&p->units, (size_t)p->unit_cap * 8) != hipSuccess) ||
        hipMalloc((void**)&p->xp_pk, (size_t)rows * DPH_DIM * 4) != hipSuccess ||
        hipMalloc((void**)&p->xp_hi, (size_t)rows * DPH_DIM * 2) != hipSuccess) {
        pq_free_scratch(p);
        return pq_fail(DPH_E_NOMEM, "PQ search: scratch allocation failed");
    }
    p->bound = (unsigned*)(p->counters + 16);
    p->cand_count = p->bound + rows;
    p->overflow = p->cand_count + rows;
    p->nf = (int32_t*)(p->overflow + rows);
    p->cap_rows = rows; p->cap_k = k; p->cap_nprobe = nprobe;
    return DPH_OK;
}

// x_dev [n,768] -> D/I [n,k], status [n] (0 = exact top-k of the probed lists, 1 = candidate buffers overflowed); asynchronous on st
int dph_pq_search_dev(dph_pq* p, const float* x_dev, int64_t n, int k, int nprobe, float* D, int64_t* I, int32_t* status, hipStream_t st) {
    if (!p || !p->finalized) return pq_fail(DPH_E_STATE, "PQ search: index not finalized");
    PQCHK(hipSetDevice(p->device));
    nprobe = std::max(1, std::min(nprobe, p->nlist));
    const int pass = (int)std::min<int64_t>(n, DPH_PASS_MAX);
    int rc = pq_ensure(p, pass, k, nprobe);
    if (rc) return rc;
    static std::atomic<size_t> attr_bytes[64];                // dynamic LDS the kernel is allowed on that device so far
    if (p->device >= 64 || attr_bytes[p->device] < pq_lds_bytes(p)) {
        hipError_t e = hipSuccess;
        const void* ks[4] = {(const void*)pq_adc_kernel<6>, (const void*)pq_adc_kernel<8>, (const void*)pq_adc_rows_kernel<6>, (const void*)pq_adc_rows_kernel<8>};
        for (const void* kf : ks) if (e == hipSuccess) e = hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pq_lds_bytes(p));
        if (e != hipSuccess) return pq_fail(DPH_E_HIP, std::string("PQ search: hipFuncSetAttribute: ") + hipGetErrorString(e));
        if (p->device < 64) attr_bytes[p->device] = pq_lds_bytes(p);
    }
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, p->device);
    for (int64_t q0 = 0; q0 < n; q0 += DPH_PASS_MAX) {
        const int nq = (int)std::min<int64_t>(n - q0, DPH_PASS_MAX);
        if (pq_launch_transform(x_dev + q0 * DPH_DIM, p->At, nq, p->b, p->xp, p->cent_pk ? p->xp_hi : (unsigned short*)nullptr, p->cent_pk ? p->xp_pk : (unsigned*)nullptr, p->device, st, p->nf))
            return pq_fail(DPH_E_HIP, "PQ search: hipFuncSetAttribute(pq_transform_kernel)");
        {
            const dim3 lg(nq, (p->M + 7) / 8);
            if (p->dsub == 8) hipLaunchKernelGGL(pq_lut_kernel<8>, lg, dim3(256), 0, st, p->xp, p->pqc, p->M, p->dsub, p->lut);             // M = 96 (the released OPQ96)
            else if (p->dsub == 12) hipLaunchKernelGGL(pq_lut_kernel<12>, lg, dim3(256), 0, st, p->xp, p->pqc, p->M, p->dsub, p->lut);       // M = 64
            else if (p->dsub == 16) hipLaunchKernelGGL(pq_lut_kernel<16>, lg, dim3(256), 0, st, p->xp, p->pqc, p->M, p->dsub, p->lut);       // M = 48
            else hipLaunchKernelGGL(pq_lut_kernel<0>, lg, dim3(256), 0, st, p->xp, p->pqc, p->M, p->dsub, p->lut);
        }
        // many short lists on average: group the work by query row, cut by code count (pq_units_kernel)
        const bool by_rows = pq_by_rows(p);
        PQCHK(hipMemsetAsync(p->counters, 0, 64 + (size_t)3 * p->cap_rows * 4, st));      // counters, bounds, candidate counts, overflow flags (the coarse
                                                                                          // quantizer flags a row whose error band overflows there too)
        unsigned* const lmask = by_rows ? nullptr : p->listmask;       // the row-major scan walks the probe lists, not the masks (134 MB to clear at 2^20 lists)
        if (p->cent_hi && p->coarse_filter) {
            // profiling: an event pair of its own around EVERY launch of the filter's dominant kernel (the scan; the GEMM forms: the one
            // GEMM launch) -- what a rocprofv3 kernel trace reports per dispatch, so the two can be held against each other
            dph_event_source src{nullptr, nullptr};
            if (p->profile) {
                src.ctx = p;
                src.next = [](void* ctx, hipEvent_t* a, hipEvent_t* b) {
                    dph_pq* q = (dph_pq*)ctx;
                    std::pair<hipEvent_t, hipEvent_t> ev{nullptr, nullptr};
                    if (!q->prof_free.empty()) { ev = q->prof_free.back(); q->prof_free.pop_back(); }
                    else { (void)hipEventCreate(&ev.first); (void)hipEventCreate(&ev.second); }
                    q->prof_events.push_back(ev);
                    *a = ev.first; *b = ev.second;
                };
            }
            dph_launch_coarse_filter(p->xp, nq, p->cent, p->cent_hi, p->xp_hi, p->cent_pk, p->xp_pk, p->nlist, nprobe, p->cnorm_max, p->scores, lmask,
                                     DPH_UNIT_WORDS, by_rows ? p->probe : nullptr, nprobe, &p->coarse_cs, &p->coarse_cf, st, src, p->overflow, p->coarse_filter, p->cent_frag, p->cent_pieces, p->cnorm, p->cnorm_cap, p->coarse_teams);
        }
        else
            dph_launch_coarse_presplit(p->xp, 0, nq, nullptr, 0, p->cent, p->nlist, nprobe, p->cnorm_max, p->scores, lmask, DPH_UNIT_WORDS,
                                       nullptr, 0, nullptr, by_rows ? p->probe : nullptr, nprobe, p->cent_pk, p->cent_pk ? p->xp_pk : nullptr, &p->coarse_cs, st,
                                       true, p->overflow);
        pq_scan_args a;
        a.xp = p->xp; a.cent = p->cent; a.lut = p->lut; a.codes = p->codes; a.list_off = p->list_off;
        a.pairs = p->pairs; a.n_pairs = p->counters + 0; a.next = p->counters + 1; a.pair_cap = p->pair_cap;
        a.M = p->M; a.seg = pq_seg(p); a.k = k; a.by_residual = p->by_residual;
        a.bound = p->bound; a.cand_count = p->cand_count; a.cand = p->cand; a.cand_cap = p->cand_cap; a.overflow = p->overflow;
        a.prof = p->phase_prof && p->phase_wgs >= cus ? p->phase_prof : nullptr;
        a.split_lut = p->split_lut;
        a.units = p->units; a.n_units = p->counters + 2; a.unit_cap = p->unit_cap;
        if (by_rows) {
            const int upr = (nprobe + PQ_GROUP - 1) / PQ_GROUP;
            hipLaunchKernelGGL(pq_units_kernel, dim3((nq * upr + 3) / 4), dim3(256), 0, st, p->probe, nprobe, upr, nq * upr, p->list_off, a.seg, p->units,
                               p->counters + 2, p->unit_cap, p->overflow);
            if (p->M <= 96) hipLaunchKernelGGL(pq_adc_rows_kernel<6>, dim3(cus), dim3(PQ_THREADS), pq_lds_bytes(p), st, a, p->probe, nprobe, upr);
            else hipLaunchKernelGGL(pq_adc_rows_kernel<8>, dim3(cus), dim3(PQ_THREADS), pq_lds_bytes(p), st, a, p->probe, nprobe, upr);
        } else {
            hipLaunchKernelGGL(pq_pairs_kernel, dim3((p->nlist + 255) / 256), dim3(256), 0, st, p->listmask, DPH_UNIT_WORDS, p->nlist,
                               p->list_off, a.seg, p->pairs, p->counters + 0, p->pair_cap, p->overflow);
            if (p->M <= 96) hipLaunchKernelGGL(pq_adc_kernel<6>, dim3(cus), dim3(PQ_THREADS), pq_lds_bytes(p), st, a);
            else hipLaunchKernelGGL(pq_adc_kernel<8>, dim3(cus), dim3(PQ_THREADS), pq_lds_bytes(p), st, a);
        }
        hipLaunchKernelGGL(pq_final_kernel, dim3(nq), dim3(PQ_THREADS), 0, st, p->cand, p->cand_count, p->cand_cap, p->overflow, p->ids,
                           (int)q0, k, D, I, status, (const int32_t*)p->nf);
    }
    PQCHK(hipGetLastError());
    return DPH_OK;
}

int dph_pq_reconstruct_dev(dph_pq* p, const int64_t* ids_dev, int64_t n, float* out_dev, int32_t* found_dev, hipStream_t st) {
    if (!p || !p->finalized) return pq_fail(DPH_E_STATE, "PQ reconstruct: index not finalized");
    PQCHK(hipSetDevice(p->device));
    if (n <= 0) return DPH_OK;
    pq_store s{p->cent, p->pqc, p->codes, p->list_off, p->dm_ids, p->dm_pos, p->ntotal, p->nlist, p->M, p->dsub, p->by_residual};
    hipLaunchKernelGGL(pq_reconstruct_kernel, dim3((unsigned)n), dim3(256), 0, st, s, ids_dev, n, out_dev, found_dev);
    PQCHK(hipGetLastError());
    return DPH_OK;
}

int dph_pq_window(dph_pq* p, int direction, dph_idmap idmap, const float* qhalf, int64_t n_q, int k, int L, const int64_t* ids,
                  const int32_t* doc, const int32_t* word, const float* first, const int32_t* row2doc, const int32_t* row2word,
                  const int32_t* doc_ids, int64_t n_docs, const int64_t* f2o_off, const int32_t* f2o, int32_t* pred_word,
                  double* best, int32_t* argslot, float* vecs, hipStream_t st) {
    if (!p || !p->finalized) return pq_fail(DPH_E_STATE, "PQ window: index not finalized");
    PQCHK(hipSetDevice(p->device));
    if (n_q <= 0) return DPH_OK;
    if (n_q > p->qrot_rows) {
        if (p->qrot) { (void)hipFree(p->qrot); p->qrot = nullptr; p->qrot_rows = 0; }
        PQCHK(hipMalloc((void**)&p->qrot, (size_t)n_q * DPH_DIM * 4));
        p->qrot_rows = n_q;
    }
    if (pq_launch_transform(qhalf, p->At, n_q, nullptr, p->qrot, nullptr, nullptr, p->device, st)) return pq_fail(DPH_E_HIP, "PQ window: hipFuncSetAttribute(pq_transform_kernel)");
    pq_store s{p->cent, p->pqc, p->codes, p->list_off, p->dm_ids, p->dm_pos, p->ntotal, p->nlist, p->M, p->dsub, p->by_residual};
    const int64_t n_cand = n_q * k;
    hipLaunchKernelGGL(pq_window_kernel, dim3((unsigned)((n_cand + 3) / 4)), dim3(256), 0, st, direction, s, idmap, p->qrot, n_cand, k, L,
                       ids, doc, word, first, row2doc, row2word, doc_ids, n_docs, f2o_off, f2o, pred_word, best, argslot, vecs);
    if (vecs && p->A) hipLaunchKernelGGL(pq_unrotate_kernel, dim3((unsigned)(2 * n_cand)), dim3(256), 0, st, vecs, p->A, 2 * n_cand);
    PQCHK(hipGetLastError());
    return DPH_OK;
}
