// dph_scan.hip -- the headline kernel: brute-force inner-product candidate scan over the int8 phrase dump
// (replaces faiss Index.search at /root/reference/densephrases/index.py:200), plus the query quantiser and the
// shard utilities (synthetic fill, centred row-norm bound).  gfx950 / CDNA4 only.
//
// Arithmetic.  The reference searches fp32 vectors x = n/20 - 2 de-quantised from int8 n
// (embed_utils.py:141-149).  <q, x> = (<q, n>)/20 - 2*sum(q), so ranking rows by <q, n> is ranking by score.
// Each query row is written as a two-digit fixed-point number  q_j = sc*(128*q1_j + q2_j) + e_j  with
// int8 digits |q1| <= 127, |q2| <= 64, and the scan computes the EXACT integer
//     I(row) = 128*<q1, n> + <q2, n>            (|I| < 2^31)
// with v_mfma_i32_32x32x32_i8 -- the database bytes are MFMA operands as they lie in HBM, no conversion.
// The residual e is known exactly, |<e, n - c>| <= ||e||_2 * max_row ||n - c||_2, so the select kernel
// (dph_select.hip) can PROVE that the exact top-k is inside the candidate lists this kernel emits, re-rank the
// candidates with the exact fp64 score and certify the result -- or report that it could not.
//
// Structure.  256 threads = 4 waves, one per SIMD, one workgroup per CU (the register file is spent on the
// query: 2 digits x 24 k-steps x 4 VGPR = 192 registers per lane hold this wave's 32 query rows for the
// whole launch).  Database tiles of 32 rows (24 KiB, contiguous in HBM) stream HBM -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no VGPR round trip, NBUF-1 tiles in flight per CU) and every wave reads every
// tile with ds_read_b128; the 768-byte row stride would put a whole lane group on one 16-byte bank slot
// (16-way conflict), so the DMA *source* address is XOR-swizzled per row (k order inside a dot product is
// free as long as query and database agree) and the LDS image is conflict-free.
// MFMA tile: A = 32 database rows x 32 k, B = 32 k x 32 query rows, so after the k loop lane l holds, for
// query row (l & 31), the 16 scores of database rows i(r) = (r&3) + 8*(r>>2) + 4*(l>>5), r = 0..15.
// Top-k: every lane keeps its own candidate list (capacity CAP, pruned to the KP best by a wave-cooperative
// rank-by-counting when full) and a threshold tau = its KP-th best; the hot path is 16 adds, 15 max and one
// compare per tile, the list code only runs when some lane beats its threshold.
#include "dph_internal.h"
#include <type_traits>

// ------------------------------------------------------------------------------------------ helpers
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ------------------------------------------------------------------------------------------ quantiser
// one workgroup per (padded) query row; writes the two int8 digits in the register-fragment order of the
// scan ([pass][digit][wave][kstep][lane][16 B]) and the row's fp64 scalars.
__global__ __launch_bounds__(256) void dph_quantize_kernel(const float* __restrict__ x, int64_t n,
                                                           int8_t* __restrict__ qfrag,
                                                           dph_qinfo* __restrict__ qinfo) {
    __shared__ double red[5][4];
    __shared__ float redf[4];
    const int r = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    float v[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) v[i] = (r < n) ? x[(int64_t)r * DPH_DIM + t + 256 * i] : 0.f;
    float am = fmaxf(fabsf(v[0]), fmaxf(fabsf(v[1]), fabsf(v[2])));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
    if (lane == 0) redf[w] = am;
    __syncthreads();
    am = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
    const double s = am > 0.f ? (double)am / 127.0 : 1.0;
    const double sc = s / 128.0;
    double e2 = 0, es = 0, qs = 0, ql1 = 0;
    const int pass = r / DPH_QROWS, rr = r % DPH_QROWS, qw = rr >> 5, col = rr & 31;
    int8_t* base = qfrag + (int64_t)pass * DPH_QFRAG_BYTES;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int j = t + 256 * i;
        const double u = (double)v[i] / s;
        double q1 = rint(u);
        q1 = fmin(127.0, fmax(-127.0, q1));
        double q2 = rint((u - q1) * 128.0);
        q2 = fmin(64.0, fmax(-64.0, q2));
        const double e = (double)v[i] - sc * (128.0 * q1 + q2);
        e2 += e * e; es += e; qs += (double)v[i]; ql1 += fabs((double)v[i]);
        const int ks = j >> 5, half = (j >> 4) & 1, byte = j & 15;
        const int64_t off = ((int64_t)((qw * DPH_KSTEPS + ks) * 64 + half * 32 + col)) * 16 + byte;
        base[off] = (int8_t)(int)q1;                                        // digit 0
        base[off + (int64_t)4 * DPH_KSTEPS * 64 * 16] = (int8_t)(int)q2;    // digit 1
    }
    e2 = wave_sum_f64(e2); es = wave_sum_f64(es); qs = wave_sum_f64(qs); ql1 = wave_sum_f64(ql1);
    if (lane == 0) { red[0][w] = e2; red[1][w] = es; red[2][w] = qs; red[3][w] = ql1; }
    __syncthreads();
    if (t == 0) {
        dph_qinfo qi;
        qi.sc = sc;
        qi.e_norm2 = sqrt(red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        qi.e_sum = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        qi.q_sum = red[2][0] + red[2][1] + red[2][2] + red[2][3];
        qi.q_l1 = red[3][0] + red[3][1] + red[3][2] + red[3][3];
        qinfo[r] = qi;
    }
}

void dph_launch_quantize(const float* x_dev, int64_t n_rows, int8_t* qfrag_dev, dph_qinfo* qinfo_dev,
                         hipStream_t st) {
    const int64_t padded = (n_rows + DPH_QROWS - 1) / DPH_QROWS * DPH_QROWS;
    hipLaunchKernelGGL(dph_quantize_kernel, dim3((unsigned)padded), dim3(256), 0, st, x_dev, n_rows, qfrag_dev,
                       qinfo_dev);
}

// ------------------------------------------------------------------------------------------ scan
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    // counted wait for the LDS-DMA queue (hipcc does not count it for us across the raw barrier)
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else static_assert(N < 0, "unsupported vmcnt");
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
// Fragment reads are issued by hand (inline asm) so that their completion can be awaited with a COUNTED
// lgkmcnt: hipcc's own bookkeeping waits lgkmcnt(0) in this loop, i.e. for the read it issued a moment ago.
// Form (ii) of the asm-load discipline: "=v" load, then a wait statement that names the destination "+v",
// which is also what keeps the consuming MFMA below the wait.
template <int OFF>
__device__ __forceinline__ void ds_read16(v4i& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
// One 32-bit lane of an accumulator tuple, read straight out of its AGPR (hipcc otherwise copies the whole
// 16-register tuple to VGPRs at its first use, which costs 32 VGPRs for the length of a tile).  The MFMA that
// wrote the register must have retired: the callers read a tile's accumulators >= 4 k-steps (8 MFMAs) into
// the NEXT tile, the matrix pipe is in-order, so the producer is long done -- no wait states needed here.
__device__ __forceinline__ int acc_lane(int a) {
    int v;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
    return v;
}
template <int N>
__device__ __forceinline__ void wait_lgkm(v4i& dst) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(dst) : "i"(N));
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Wave-cooperative prune of lane X's list: rank every entry by counting (keys are distinct), keep the KP
// largest in sorted order at the head of the list, return the score of the KP-th (the new threshold).
template <int KP, int CAP>
__device__ __forceinline__ int prune_list(uint64_t* L, int cntX, int lane, int& kept) {
    static_assert(CAP <= 64, "one entry per lane");
    uint64_t key = (lane < cntX) ? L[lane] : 0ull;
    const unsigned lo = (unsigned)key, hi = (unsigned)(key >> 32);
    int rank = 0;
#pragma unroll
    for (int j = 0; j < CAP; ++j) {
        const unsigned jl = __builtin_amdgcn_readlane(lo, j), jh = __builtin_amdgcn_readlane(hi, j);
        const uint64_t kj = ((uint64_t)jh << 32) | jl;
        rank += (kj > key) ? 1 : 0;
    }
    const bool valid = lane < cntX;
    if (valid && rank < KP) L[rank] = key;
    kept = cntX < KP ? cntX : KP;
    const unsigned long long m = __builtin_amdgcn_ballot_w64(valid && rank == KP - 1);
    int newtau = (int)0x80000000;
    if (m) {
        const int src = __builtin_ctzll(m);
        newtau = (int)(__builtin_amdgcn_readlane(hi, src) ^ 0x80000000u);
    }
    return newtau;
}

// PF = how many k-steps ahead the ds_read_b128 of a database fragment is issued (register ring of PF+1),
// KSYNC = the k-step of tile `it` at which the hand-over for tile it+1 happens (wait for its DMA, barrier,
// issue the DMA of tile it+NBUF-1 into the buffer tile it-1 just vacated).  Doing the hand-over mid-tile keeps
// the matrix pipe fed across tile boundaries: the first fragments of tile it+1 are already in registers when
// tile it ends, and the (rare-path) threshold test of tile it rides on the k-loop of tile it+1.
#define DPH_PF 3
#define DPH_KSYNC 12

// SAMPLE = true is the threshold pre-pass: the same kernel over every `tile_stride`-th tile of the shard (its own
// name in a profile).  Its candidate lists only serve dph_threshold_kernel, which turns them into a per-query-row
// lower bound of the KP-th best integer score of the WHOLE shard (the sample is a subset); the full scan then starts
// every lane's threshold there (`tau_init`), which makes the list code ~100x rarer -- and since one wave in its rare
// path holds the other three at the tile barrier, that is what keeps the matrix pipe busy.
template <int KP, int CAP, int NBUF, bool SAMPLE>
__global__ __launch_bounds__(DPH_SCAN_THREADS, 1) void dph_scan_kernel(const int8_t* __restrict__ db,
                                                                       int64_t n_rows, int64_t n_tiles,
                                                                       int tile_stride,
                                                                       const int8_t* __restrict__ qfrag,
                                                                       const int* __restrict__ tau_init,
                                                                       uint64_t* __restrict__ lists_out) {
    static_assert(NBUF >= 3, "mid-tile hand-over needs three LDS buffers");
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [NBUF][24576] tiles | [256][CAP] u64 lists
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint64_t* const lists = (uint64_t*)(smem + NBUF * DPH_TILE_BYTES);
    uint64_t* const mylist = lists + tid * CAP;

    const int64_t t0 = ((int64_t)blockIdx.x * n_tiles) / gridDim.x;
    const int64_t t1 = ((int64_t)(blockIdx.x + 1) * n_tiles) / gridDim.x;
    const int nt = (int)(t1 - t0);

    // ---- this wave's 32 query rows, both digits, resident in registers for the whole launch
    v4i qh[DPH_KSTEPS], ql[DPH_KSTEPS];
    {
        const v4i* qf = (const v4i*)qfrag;
#pragma unroll
        for (int ks = 0; ks < DPH_KSTEPS; ++ks) {
            qh[ks] = qf[((0 * 4 + wave) * DPH_KSTEPS + ks) * 64 + lane];
            ql[ks] = qf[((1 * 4 + wave) * DPH_KSTEPS + ks) * 64 + lane];
        }
    }

    // ---- LDS-DMA source offsets: LDS unit u = i*256 + tid holds chunk c of row (u / 48), where
    //      c = (c' & 0x30) | ((c' ^ row) & 15), c' = u % 48  (XOR swizzle applied on the SOURCE side)
    unsigned goff[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const unsigned u = i * 256 + tid, row = u / 48, cp = u % 48;
        const unsigned c = (cp & 0x30u) | ((cp ^ row) & 15u);
        goff[i] = row * DPH_DIM + c * 16;
    }
    // ---- fragment read addresses: lane reads row (lane&31), chunk 2ks + (lane>>5)
    unsigned faddr[8];
    {
        const unsigned row = lane & 31, h = (unsigned)(lane >> 5) ^ (row & 15u);
#pragma unroll
        for (int m = 0; m < 8; ++m) faddr[m] = row * DPH_DIM + (((2u * m) ^ h) << 4);
    }

    auto issue_dma = [&](int64_t tile, int buf) {
        const int8_t* src = db + tile * (int64_t)tile_stride * (int64_t)DPH_TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            __builtin_amdgcn_global_load_lds((gptr_t)(src + goff[i]),
                                             (lptr_t)(smem + buf * DPH_TILE_BYTES + (i * 256 + wave * 64) * 16),
                                             16, 0, 0);
        }
    };
    const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)smem;
#pragma unroll
    for (int m = 0; m < 8; ++m) faddr[m] += lds_base;

    // nothing can be <= INT_MIN: without a pre-pass bound the first rows always enter
    int tau = tau_init ? tau_init[wave * 32 + (lane & 31)] : (int)0x80000000;
    int cnt = 0;

    // ---- prologue: NBUF-1 tiles in flight, wait for tile 0, pre-load its first fragments
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the Q loads share the vm counter with the DMA
#pragma unroll
    for (int p = 0; p < NBUF - 1; ++p)
        if (p < nt) issue_dma(t0 + p, p);
    if (nt >= NBUF - 1) wait_vmcnt<6 * (NBUF - 2)>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    v4i bq[DPH_PF + 1];
    ds_read16<0>(bq[0], faddr[0]);
    ds_read16<0>(bq[1], faddr[1]);
    ds_read16<0>(bq[2], faddr[2]);
    static_assert(DPH_PF == 3, "prologue and ring indexing assume a 3-deep prefetch");

    // Two accumulator sets (A, B) alternate between "being accumulated" and "being tested", so the threshold
    // test of tile it-1 (AGPR reads, adds, max) interleaves with the MFMAs of tile it instead of stalling them.
    v16i accA_h = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, accA_l = accA_h, accB_h = accA_h, accB_l = accA_h;

    // step `it` multiplies tile it into (ch, cl) and tests the scores of tile it-1 held in (ph, pl);
    // tiles >= nt are phantoms (stale LDS bytes, results never tested) that only flush the pipeline.
    auto tile_step = [&](v16i& ch, v16i& cl, const v16i& ph, const v16i& pl, const int it)
                         __attribute__((always_inline)) {
        const unsigned tb = (unsigned)(it % NBUF) * DPH_TILE_BYTES;
        const unsigned tn = (unsigned)((it + 1) % NBUF) * DPH_TILE_BYTES;
        const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        int mx = (int)0x80000000;
        unsigned fa[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) fa[m] = faddr[m] + tb;
        static_for<0, DPH_KSTEPS>([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            if constexpr (ks == DPH_KSYNC) {
                // hand-over: tile it+1 must have landed (ours: counted wait, everyone's: barrier)
                if (it + NBUF - 2 < nt) wait_vmcnt<6 * (NBUF - 3)>();
                else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if (it + NBUF - 1 < nt) issue_dma(t0 + it + NBUF - 1, (it + NBUF - 1) % NBUF);
            }
            constexpr int p = ks + DPH_PF;
            if constexpr (p < DPH_KSTEPS) ds_read16<(p >> 3) * 256>(bq[p & DPH_PF], fa[p & 7]);
            else ds_read16<0>(bq[p & DPH_PF], faddr[p - DPH_KSTEPS] + tn);
            wait_lgkm<DPH_PF>(bq[ks & DPH_PF]);
            if constexpr (ks == 0) {
                ch = __builtin_amdgcn_mfma_i32_32x32x32_i8(bq[ks & DPH_PF], qh[ks], zero, 0, 0, 0);
                cl = __builtin_amdgcn_mfma_i32_32x32x32_i8(bq[ks & DPH_PF], ql[ks], zero, 0, 0, 0);
            } else {
                ch = __builtin_amdgcn_mfma_i32_32x32x32_i8(bq[ks & DPH_PF], qh[ks], ch, 0, 0, 0);
                cl = __builtin_amdgcn_mfma_i32_32x32x32_i8(bq[ks & DPH_PF], ql[ks], cl, 0, 0, 0);
            }
            if constexpr (ks >= 4 && ks < 20) {
                mx = max(mx, (acc_lane(ph[ks - 4]) << 7) + acc_lane(pl[ks - 4]));
                asm volatile("" : "+v"(mx));   // keep the running max a chain (a re-associated tree holds 32 VGPRs)
            }
            // pin the software pipeline: one fragment read PF steps ahead, two MFMAs and one slice of the
            // previous tile's threshold test per k-step
            __builtin_amdgcn_sched_barrier(0);
        });

        if (it >= 1 && it <= nt && __builtin_amdgcn_ballot_w64(mx > tau) != 0ull) {
            // ---------------- rare path: some lane has a row of tile it-1 that beats its threshold
            const unsigned rowbase = (unsigned)((t0 + it - 1) * tile_stride * DPH_TILE_ROWS) + 4u * (unsigned)(lane >> 5);
            unsigned done = 0;
            bool again;
            do {
                bool blocked = false;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int s = (acc_lane(ph[r]) << 7) + acc_lane(pl[r]);
                    const bool hit = (s > tau) && !((done >> r) & 1u);
                    if (__builtin_amdgcn_ballot_w64(hit) != 0ull) {
                        if (hit) {
                            const unsigned row = rowbase + (unsigned)((r & 3) + 8 * (r >> 2));
                            if ((int64_t)row >= n_rows) {
                                done |= 1u << r;                 // padding row of the last tile
                            } else if (cnt < CAP) {
                                mylist[cnt] = dph_make_key(s, row);
                                ++cnt;
                                done |= 1u << r;
                            } else {
                                blocked = true;
                            }
                        }
                    }
                }
                again = __builtin_amdgcn_ballot_w64(blocked) != 0ull;
                unsigned long long full = __builtin_amdgcn_ballot_w64(cnt >= CAP);
                while (full) {
                    const int X = __builtin_ctzll(full);
                    full &= full - 1;
                    int kept;
                    const int nt_ = prune_list<KP, CAP>(lists + (wave * 64 + X) * CAP, CAP, lane, kept);
                    if (lane == X) { tau = nt_; cnt = kept; }
                }
            } while (again);
        }
    };

    for (int it = 0; it <= nt; it += 2) {
        tile_step(accA_h, accA_l, accB_h, accB_l, it);
        tile_step(accB_h, accB_l, accA_h, accA_l, it + 1);
    }

    // ---- drain: reduce every list to its KP best (sorted), publish [block][thread][KP], 0 = empty slot
    {
        unsigned long long over = __builtin_amdgcn_ballot_w64(cnt > KP);
        while (over) {
            const int X = __builtin_ctzll(over);
            over &= over - 1;
            const int cx = __builtin_amdgcn_readlane(cnt, X);
            int kept;
            const int nt_ = prune_list<KP, CAP>(lists + (wave * 64 + X) * CAP, cx, lane, kept);
            if (lane == X) { tau = nt_; cnt = kept; }
        }
        uint64_t* out = lists_out + ((int64_t)blockIdx.x * DPH_SCAN_THREADS + tid) * KP;
#pragma unroll
        for (int i = 0; i < KP; ++i) out[i] = (i < cnt) ? mylist[i] : 0ull;
    }
}

int dph_scan_grid(int device) {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    return cus > 0 ? cus : 256;
}

template <int KP, int CAP, int NBUF, bool SAMPLE>
static void launch_scan_t(const int8_t* db, int64_t n_rows, int64_t n_tiles, int tile_stride, const int8_t* qfrag,
                          const int* tau_init, uint64_t* lists, int grid, hipStream_t st) {
    const size_t lds = (size_t)NBUF * DPH_TILE_BYTES + (size_t)DPH_SCAN_THREADS * CAP * 8;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)dph_scan_kernel<KP, CAP, NBUF, SAMPLE>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((dph_scan_kernel<KP, CAP, NBUF, SAMPLE>), dim3(grid), dim3(DPH_SCAN_THREADS), lds, st, db, n_rows,
                       n_tiles, tile_stride, qfrag, tau_init, lists);
}

// n_tiles = number of tiles this launch visits (tile i of the launch is shard tile i*tile_stride)
void dph_launch_scan(int kp, bool sample, const int8_t* db, int64_t n_rows, int64_t n_tiles, int tile_stride,
                     const int8_t* qfrag, const int* tau_init, uint64_t* lists, int grid, hipStream_t st) {
    if (kp == 16) {
        if (sample) launch_scan_t<16, 24, 4, true>(db, n_rows, n_tiles, tile_stride, qfrag, tau_init, lists, grid, st);
        else launch_scan_t<16, 24, 4, false>(db, n_rows, n_tiles, tile_stride, qfrag, tau_init, lists, grid, st);
    } else {
        if (sample) launch_scan_t<32, 40, 3, true>(db, n_rows, n_tiles, tile_stride, qfrag, tau_init, lists, grid, st);
        else launch_scan_t<32, 40, 3, false>(db, n_rows, n_tiles, tile_stride, qfrag, tau_init, lists, grid, st);
    }
}

// ------------------------------------------------------------------------------------------ pre-pass threshold
// One workgroup per query row of the pass: the KP-th largest integer score in that row's sample lists, minus one
// (rows scoring exactly the KP-th value must still enter), or INT_MIN when the sample holds fewer than KP rows.
template <int KP>
__global__ __launch_bounds__(256) void dph_threshold_kernel(const uint64_t* __restrict__ lists, int grid,
                                                            int* __restrict__ tau_out) {
    __shared__ unsigned cnt_sh[4];
    const int qi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int qw = qi >> 5, qc = qi & 31;
    const int n_keys = grid * 2 * KP;
    // each thread keeps its share of the biased scores in registers (<= 8192 / 256 = 32 at grid 256, KP 16)
    constexpr int PER = 64;
    unsigned u[PER];                                 // statically indexed: stays in registers
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int e = tid + 256 * j;
        unsigned v = 0;
        if (e < n_keys) {
            const int l = e / KP, i = e % KP, blk = l >> 1, half = l & 1;
            v = (unsigned)(lists[((int64_t)blk * DPH_SCAN_THREADS + qw * 64 + half * 32 + qc) * KP + i] >> 32);
        }
        u[j] = v;                                    // 0 for empty slots, >= 1 for real scores
    }
    unsigned ans = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned cand = ans | (1u << bit);
        unsigned c = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) c += (u[j] >= cand) ? 1u : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        if (lane == 0) cnt_sh[wv] = c;
        __syncthreads();
        const unsigned total = cnt_sh[0] + cnt_sh[1] + cnt_sh[2] + cnt_sh[3];
        __syncthreads();
        if (total >= (unsigned)KP) ans = cand;
    }
    if (tid == 0) {
        const int kth = (int)(ans ^ 0x80000000u);
        tau_out[qi] = (ans == 0u || kth == (int)0x80000000) ? (int)0x80000000 : kth - 1;
    }
}

void dph_launch_threshold(int kp, const uint64_t* lists, int grid, int* tau_out, hipStream_t st) {
    if (kp == 16) hipLaunchKernelGGL((dph_threshold_kernel<16>), dim3(DPH_QROWS), dim3(256), 0, st, lists, grid, tau_out);
    else hipLaunchKernelGGL((dph_threshold_kernel<32>), dim3(DPH_QROWS), dim3(256), 0, st, lists, grid, tau_out);
}

// ------------------------------------------------------------------------------------------ synthetic fill
// rows i.i.d. float_to_int8(N(0, 0.6^2), -2, 20) ~ 40 + 12 z.  Integer-only generator
// (Irwin-Hall sum of 4 hashed bytes) so that densephrases_amd/synth.py reproduces it bit-for-bit on the host.
__device__ __forceinline__ unsigned dph_hash32(unsigned lo, unsigned hi, unsigned seed) {
    unsigned h = lo * 0x9E3779B1u ^ (hi * 0x85EBCA77u + seed);
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h;
}
__global__ __launch_bounds__(256) void dph_fill_kernel(int8_t* __restrict__ db, int64_t n_bytes, int64_t byte_base,
                                                       unsigned seed_lo, unsigned seed_hi) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 16;
    for (int64_t o = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; o < n_bytes; o += stride) {
        unsigned w[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            unsigned word = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint64_t e = (uint64_t)(byte_base + o + d * 4 + b);
                const unsigned h = dph_hash32((unsigned)e, (unsigned)(e >> 32) ^ seed_hi, seed_lo);
                const int sum = (int)(h & 255u) + (int)((h >> 8) & 255u) + (int)((h >> 16) & 255u) + (int)(h >> 24);
                int v = DPH_CENTER + (((sum - 510) * 5321 + 32768) >> 16);
                v = v < -128 ? -128 : (v > 127 ? 127 : v);
                word |= ((unsigned)v & 255u) << (8 * b);
            }
            w[d] = word;
        }
        *(uint4*)(db + o) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
void dph_launch_fill(int8_t* db, int64_t n_rows, int64_t id_base, uint64_t seed, hipStream_t st) {
    const int64_t n_bytes = n_rows * DPH_DIM;
    hipLaunchKernelGGL(dph_fill_kernel, dim3(256 * 8), dim3(256), 0, st, db, n_bytes, id_base * DPH_DIM,
                       (unsigned)seed, (unsigned)(seed >> 32));
}

// ------------------------------------------------------------------------------------------ centred row norm
// max over real rows of sum_j (n_j - c)^2 (exact integer): the shard constant of the certificate.
__global__ __launch_bounds__(256) void dph_rownorm_kernel(const int8_t* __restrict__ db, int64_t n_rows,
                                                          unsigned long long* __restrict__ max_out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    int best = 0;
    for (int64_t row = wave0; row < n_rows; row += nwaves) {
        int acc = 0;
        if (lane < 48) {
            const uint4 v = *(const uint4*)(db + row * DPH_DIM + lane * 16);
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int n = (int)(int8_t)(w[d] >> (8 * b)) - DPH_CENTER;
                    acc += n * n;
                }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        best = max(best, acc);
    }
    if (lane == 0 && best > 0) atomicMax(max_out, (unsigned long long)best);
}
void dph_launch_rownorm(const int8_t* db, int64_t n_rows, unsigned long long* max_out, hipStream_t st) {
    hipLaunchKernelGGL(dph_rownorm_kernel, dim3(256 * 8), dim3(256), 0, st, db, n_rows, max_out);
}
