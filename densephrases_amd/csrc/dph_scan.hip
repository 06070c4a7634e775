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
#include <stdlib.h>
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
                                                           dph_qinfo* __restrict__ qinfo, double rmax,
                                                           int* __restrict__ lmax_out) {
    __shared__ double red[6][4];
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
    double e2 = 0, es = 0, qs = 0, ql1 = 0, q2s = 0, q2n = 0;
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
        q2s += q2; q2n += q2 * q2;
        const int ks = j >> 5, half = (j >> 4) & 1, byte = j & 15;
        const int64_t off = ((int64_t)((qw * DPH_KSTEPS + ks) * 64 + half * 32 + col)) * 16 + byte;
        base[off] = (int8_t)(int)q1;                                        // digit 0
        base[off + (int64_t)4 * DPH_KSTEPS * 64 * 16] = (int8_t)(int)q2;    // digit 1
    }
    e2 = wave_sum_f64(e2); es = wave_sum_f64(es); qs = wave_sum_f64(qs); ql1 = wave_sum_f64(ql1);
    q2s = wave_sum_f64(q2s); q2n = wave_sum_f64(q2n);
    if (lane == 0) { red[0][w] = e2; red[1][w] = es; red[2][w] = qs; red[3][w] = ql1; red[4][w] = q2s; red[5][w] = q2n; }
    __syncthreads();
    if (t == 0) {
        dph_qinfo qi;
        qi.sc = sc;
        qi.e_norm2 = sqrt(red[0][0] + red[0][1] + red[0][2] + red[0][3]);
        qi.e_sum = red[1][0] + red[1][1] + red[1][2] + red[1][3];
        qi.q_sum = red[2][0] + red[2][1] + red[2][2] + red[2][3];
        qi.q_l1 = red[3][0] + red[3][1] + red[3][2] + red[3][3];
        qinfo[r] = qi;
        // upper bound of the low-digit term L = <q2, n> over every real row of the shard (lazy scan):
        // <q2, n - c> + c*sum(q2) <= ||q2||_2 * rmax + c*sum(q2), rounded up
        const double q2sum = red[4][0] + red[4][1] + red[4][2] + red[4][3];
        const double q2nrm = sqrt(red[5][0] + red[5][1] + red[5][2] + red[5][3]);
        const double lm = ceil(q2nrm * rmax + (double)DPH_CENTER * q2sum) + 1.0;
        lmax_out[r] = lm > 1.0e9 ? 1000000000 : (lm < -1.0e9 ? -1000000000 : (int)lm);
    }
}

void dph_launch_quantize(const float* x_dev, int64_t n_rows, int8_t* qfrag_dev, dph_qinfo* qinfo_dev, double rmax,
                         int* lmax_dev, hipStream_t st) {
    const int64_t padded = (n_rows + DPH_QROWS - 1) / DPH_QROWS * DPH_QROWS;
    hipLaunchKernelGGL(dph_quantize_kernel, dim3((unsigned)padded), dim3(256), 0, st, x_dev, n_rows, qfrag_dev,
                       qinfo_dev, rmax, lmax_dev);
}

// ------------------------------------------------------------------------------------------ scan
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    // counted wait for the LDS-DMA queue (hipcc does not count it for us across the raw barrier)
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if constexpr (N == 30) asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
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
// Low-digit MFMA of the lazy path: the query fragment lives in AGPRs (it is touched on ~2 % of the tiles, the VGPRs go
// to the high digit), so the instruction is written by hand with an "a" operand.  Dependent MFMAs on one accumulator
// need no wait states between them; mfma_settle() covers the MFMA -> v_accvgpr_read distance before the result is read.
__device__ __forceinline__ void mfma_lo(v16i& acc, const v4i& frag, const v4i& q) {
    asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(acc) : "v"(frag), "a"(q));
}
__device__ __forceinline__ void mfma_settle(v16i& acc) {
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc));
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

// ---- database feed: HBM -> (hand-owned AGPRs) -> LDS ------------------------------------------------------------
// Each wave streams a quarter of every tile (6 x 1 KiB, perfectly coalesced) with plain global_load_dwordx4 into
// accumulator registers it owns by hand, two tiles ahead, and writes them into the swizzled LDS image with
// ds_write_b128 one hand-over later.  (The first version used LDS-DMA, global_load_lds: those pieces count on lgkmcnt
// as well as vmcnt, so a wave that issues them stalls its own counted ds_read waits on HBM latency -- PMC: +45 %
// time, all of it "waiting"; see DESIGN.md.  Plain loads only touch vmcnt, LDS ops stay in order.)
// Staging registers: set S in 0..NSET-1, piece i in 0..5 -> a[STG0 + 24*S + 4*i .. +3].  They are named literally in the
// asm and listed as clobbers; tests/test_abi.py audits the ISA for compiler traffic in that range.
#define DPH_NSET 4                      // staging sets = tiles in flight per wave (6 KiB each)
#define DPH_STG0 (256 - 24 * DPH_NSET)  // a[160:255]
template <int S, int I>
__device__ __forceinline__ void stage_load(unsigned lane16, const int8_t* base) {
    constexpr int r = DPH_STG0 + 24 * S + 4 * I;
    asm volatile("global_load_dwordx4 a[%c2:%c3], %0, %1" ::"v"(lane16), "s"(base), "i"(r), "i"(r + 3) : "memory");
}
template <int S, int I>
__device__ __forceinline__ void stage_write(unsigned lds_addr) {
    constexpr int r = DPH_STG0 + 24 * S + 4 * I;
    asm volatile("ds_write_b128 %0, a[%c1:%c2]" ::"v"(lds_addr), "i"(r), "i"(r + 3) : "memory");
}
// IVF probe mask of a tile: one dword per wave (bit j = query row 32*wave + j probes the tile's list), fetched with a
// hand-written load into a hand-owned AGPR (a156 for even tiles, a157 for odd ones) one tile ahead of its use.  It must
// not be a compiler-visible load: hipcc would wait for it with a vmcnt that also drains the staged tiles in flight.
template <int PARITY>
__device__ __forceinline__ void mask_load(unsigned zero_off, const unsigned* addr) {
    asm volatile("global_load_dword a[%c2], %0, %1" ::"v"(zero_off), "s"(addr), "i"(156 + PARITY) : "memory");
}
// `newer` = VMEM operations issued after that load (13 in steady state: two hand-overs of 6 and the next tile's mask);
// when the tail of the launch issued fewer, the caller asks for a full drain instead.
template <int PARITY>
__device__ __forceinline__ unsigned mask_read(bool steady) {
    unsigned m;
    if (steady) asm volatile("s_waitcnt vmcnt(13)\n\tv_accvgpr_read_b32 %0, a[%c1]" : "=v"(m) : "i"(156 + PARITY) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\tv_accvgpr_read_b32 %0, a[%c1]" : "=v"(m) : "i"(156 + PARITY) : "memory");
    return m;
}
// tells the compiler the hand-owned range exists (kernel descriptor) and is off limits at this point
__device__ __forceinline__ void stage_claim() {
    asm volatile("" ::: "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171",
                 "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184",
                 "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197",
                 "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219",
                 "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232",
                 "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245",
                 "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255");
}

// SAMPLE = true is the threshold pre-pass: the same kernel over every `tile_stride`-th tile of the shard (its own
// name in a profile).  Its candidate lists only serve dph_threshold_kernel, which turns them into a per-query-row
// lower bound of the KP-th best integer score of the WHOLE shard (the sample is a subset); the full scan then starts
// every lane's threshold there (`tau_init`), which makes the list code ~100x rarer -- and since one wave in its rare
// path holds the other three at the tile barrier, that matters more than the list code's own cost.
// LAZY = true is a scan behind a pre-pass threshold: only the HIGH digit is multiplied for every tile
// (24 MFMAs instead of 48).  I = 128*H + L and L = <q2, n> <= lmax := ||q2||_2 * max_row||n - c||_2 + c*sum(q2)
// (Cauchy-Schwarz, the same shard constant as the certificate), so a row with 128*H + lmax <= tau cannot beat the
// lane's threshold; only when some lane has H > floor((tau - lmax) / 128) does the wave compute the low digit of
// that tile (from AGPR-resident fragments, re-reading the tile from LDS) and run the exact test.  With the pre-pass
// threshold that happens on ~2 % of the tiles; the skipped rows have I <= tau, exactly what the lists promise.
// IVF = true: the shard is stored list-major (every tile belongs to one inverted list, padding rows have id -1) and
// `tilemask[4*tile + wave]` says which of the wave's 32 query rows probe that tile's list; rows of unprobed lists are
// ignored by the threshold test and never enter a list -- exact in-list inner product over the probed lists only
// (FAISS IndexIVFFlat semantics).  At 2B = 128 query rows and nprobe/nlist = 1/16 every list is probed by some row,
// so the whole shard is still streamed once per batch; the mask decides who may keep what.
template <int KP, int CAP, bool SAMPLE, bool LAZY, bool IVF = false>
__global__ __launch_bounds__(DPH_SCAN_THREADS, 1) void dph_scan_kernel(const int8_t* __restrict__ db,
                                                                       int64_t n_rows, int64_t n_tiles,
                                                                       int tile_stride,
                                                                       const int8_t* __restrict__ qfrag,
                                                                       const int* __restrict__ tau_init,
                                                                       const int* __restrict__ lmax_q,
                                                                       const unsigned* __restrict__ tilemask,
                                                                       const int64_t* __restrict__ row_ids,
                                                                       uint64_t* __restrict__ lists_out) {
    // tile t lives in LDS buffer t % NBUF: being read | published | being written (| kept for the lazy low digit)
    constexpr int NBUF = LAZY ? 4 : 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [3][24576] tiles | [256][CAP] u64 lists
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint64_t* const lists = (uint64_t*)(smem + NBUF * DPH_TILE_BYTES);
    uint64_t* const mylist = lists + tid * CAP;
    stage_claim();

    // Tiles are dealt round-robin: launch-tile j of this workgroup is tile j*grid + block.  At any moment the 256 CUs
    // then stream one contiguous ~6 MB window of the shard (neighbouring DRAM pages are opened by neighbouring CUs at
    // about the same time) instead of 256 windows half a gigabyte apart.  Rows still reach every lane in increasing id
    // order, which the strict `>` threshold relies on for (score desc, id asc) lists.
    const int64_t grid_n = gridDim.x;
    const int nt = (int)((n_tiles - (int64_t)blockIdx.x + grid_n - 1) / grid_n);
    auto tile_of = [&](int j) { return (int64_t)j * grid_n + (int64_t)blockIdx.x; };

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

    const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)smem;
    // ---- LDS write addresses of this lane's six staged 16-byte units.  Piece p = 4i + wave covers units
    //      u = 64p + lane of the tile; unit u is chunk c = u % 48 of row u / 48 and is stored at chunk
    //      c' = (c & 0x30) | ((c ^ row) & 15) of that row: the XOR swizzle that makes the fragment reads conflict-free
    unsigned waddr[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const unsigned u = (unsigned)(4 * i + wave) * 64u + (unsigned)lane, row = u / 48, c = u % 48;
        waddr[i] = lds_base + (row * 48 + ((c & 0x30u) | ((c ^ row) & 15u))) * 16;
    }
    const unsigned lane16 = (unsigned)lane * 16u;
    // ---- fragment read addresses: lane reads row (lane&31), chunk 2ks + (lane>>5)
    unsigned faddr[8];
    {
        const unsigned row = lane & 31, h = (unsigned)(lane >> 5) ^ (row & 15u);
#pragma unroll
        for (int m = 0; m < 8; ++m) faddr[m] = lds_base + row * DPH_DIM + (((2u * m) ^ h) << 4);
    }

    const int64_t tile_bytes = (int64_t)tile_stride * (int64_t)DPH_TILE_BYTES;
    // this wave's first piece of launch-tile j (wave-uniform); piece i is 4 KiB further
    auto piece_base = [&](int j) { return db + tile_of(j) * tile_bytes + (int64_t)wave * 1024; };

    // nothing can be <= INT_MIN: without a pre-pass bound the first rows always enter
    int tau = tau_init ? tau_init[wave * 32 + (lane & 31)] : (int)0x80000000;
    int cnt = 0;
    // lazy path: high-digit threshold thi = floor((tau - lmax) / 128); H > thi  <=>  128*H + lmax > tau
    const int lmax = LAZY ? lmax_q[wave * 32 + (lane & 31)] : 0;
    auto hi_threshold = [&](int t) {
        if (t == (int)0x80000000) return (int)0x80000000;
        const long long d = ((long long)t - (long long)lmax) >> 7;      // arithmetic shift = floor division
        return d < -2147483647ll ? (int)0x80000000 : (int)d;
    };
    int thi = hi_threshold(tau);

    // ---- prologue: tiles 0 and 1 into LDS, tiles 2 .. NSET+1 into flight (tile t travels in staging set t % NSET),
    //      pre-load the first fragments of tile 0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the Q loads share the vm counter with the feed
    {
        const int8_t* b0 = piece_base(0);
        const int8_t* b1 = piece_base(1);
        if (nt > 0) static_for<0, 6>([&](auto ic) { constexpr int i = decltype(ic)::value; stage_load<0, i>(lane16, b0 + i * 4096); });
        if (nt > 1) static_for<0, 6>([&](auto ic) { constexpr int i = decltype(ic)::value; stage_load<1, i>(lane16, b1 + i * 4096); });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (nt > 0) static_for<0, 6>([&](auto ic) { constexpr int i = decltype(ic)::value; stage_write<0, i>(waddr[i]); });
        if (nt > 1) static_for<0, 6>([&](auto ic) { constexpr int i = decltype(ic)::value; stage_write<1, i>(waddr[i] + DPH_TILE_BYTES); });
        asm volatile("s_nop 1" ::: "memory");           // the stores have read their data registers
        static_for<0, DPH_NSET>([&](auto sc) {
            constexpr int t = 2 + decltype(sc)::value;              // tile t -> set t % NSET
            if (t < nt) {
                const int8_t* bt = piece_base(t);
                static_for<0, 6>([&](auto ic) { constexpr int i = decltype(ic)::value; stage_load<t % DPH_NSET, i>(lane16, bt + i * 4096); });
            }
        });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    v4i bq[DPH_PF + 1];
    ds_read16<0>(bq[0], faddr[0]);
    ds_read16<0>(bq[1], faddr[1]);
    ds_read16<0>(bq[2], faddr[2]);
    static_assert(DPH_PF == 3, "prologue and ring indexing assume a 3-deep prefetch");

    // Two accumulator sets (A, B) alternate between "being accumulated" and "being tested", so the threshold
    // test of tile it-1 (AGPR reads, adds, max) interleaves with the MFMAs of tile it instead of stalling them.
    v16i accA_h = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, accA_l = accA_h, accB_h = accA_h, accB_l = accA_h;

    // step `it` (SET = (it+2) % NSET, a compile-time constant of the unrolled loop) multiplies tile it into (ch, cl)
    // and tests the scores of tile it-1 held in (ph, pl); tiles >= nt are phantoms (stale LDS bytes, results never
    // tested) that only flush the pipeline.
    auto tile_step = [&](auto setc, v16i& ch, v16i& cl, const v16i& ph, const v16i& pl, const int it)
                         __attribute__((always_inline)) {
        constexpr int SET = decltype(setc)::value;
        constexpr int PARITY = SET & 1;                 // (it + 2) % NSET has the parity of it
        if constexpr (IVF)
            if (it < nt) mask_load<PARITY>(0u * (unsigned)lane, tilemask + (tile_of(it) * tile_stride) * 4 + wave);
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
                // hand-over.  Staging set SET holds tile it+2 (loaded NSET hand-overs ago); the other sets hold the
                // NSET-1 younger tiles it+3 .. it+NSET+1, which stay in flight across the wait.
                if (it + DPH_NSET + 1 < nt) wait_vmcnt<6 * (DPH_NSET - 1)>();
                else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();      // tile it-1 is fully consumed (its buffer is free), tile it+1 is published
                asm volatile("" ::: "memory");
                if (it + 2 < nt) {
                    const unsigned wb = (unsigned)((it + 2) % NBUF) * DPH_TILE_BYTES;
                    static_for<0, 6>([&](auto ic) { constexpr int i = decltype(ic)::value; stage_write<SET, i>(waddr[i] + wb); });
                }
                if (it + 2 + DPH_NSET < nt) {
                    const int8_t* b4 = piece_base(it + 2 + DPH_NSET);
                    asm volatile("s_nop 1" ::: "memory");   // ds_write has read a[..] before the reload is issued
                    static_for<0, 6>([&](auto ic) { constexpr int i = decltype(ic)::value; stage_load<SET, i>(lane16, b4 + i * 4096); });
                }
            }
            constexpr int p = ks + DPH_PF;
            if constexpr (p < DPH_KSTEPS) ds_read16<(p >> 3) * 256>(bq[p & DPH_PF], fa[p & 7]);
            else ds_read16<0>(bq[p & DPH_PF], faddr[p - DPH_KSTEPS] + tn);
            wait_lgkm<DPH_PF>(bq[ks & DPH_PF]);
            if constexpr (ks == 0) {
                ch = __builtin_amdgcn_mfma_i32_32x32x32_i8(bq[ks & DPH_PF], qh[ks], zero, 0, 0, 0);
                if constexpr (!LAZY) cl = __builtin_amdgcn_mfma_i32_32x32x32_i8(bq[ks & DPH_PF], ql[ks], zero, 0, 0, 0);
            } else {
                ch = __builtin_amdgcn_mfma_i32_32x32x32_i8(bq[ks & DPH_PF], qh[ks], ch, 0, 0, 0);
                if constexpr (!LAZY) cl = __builtin_amdgcn_mfma_i32_32x32x32_i8(bq[ks & DPH_PF], ql[ks], cl, 0, 0, 0);
            }
            if constexpr (ks >= 4 && ks < 20) {
                if constexpr (LAZY) mx = max(mx, acc_lane(ph[ks - 4]));                    // high digit only
                else mx = max(mx, (acc_lane(ph[ks - 4]) << 7) + acc_lane(pl[ks - 4]));
                asm volatile("" : "+v"(mx));   // keep the running max a chain (a re-associated tree holds 32 VGPRs)
            }
            // pin the software pipeline: one fragment read PF steps ahead, two MFMAs and one slice of the
            // previous tile's threshold test per k-step
            __builtin_amdgcn_sched_barrier(0);
        });

        bool probed = true;
        if constexpr (IVF) {
            if (it >= 1 && it <= nt) {
                const unsigned m = mask_read<1 - PARITY>(it + 2 + DPH_NSET < nt);
                probed = ((m >> (lane & 31)) & 1u) != 0u;
                if (!probed) mx = (int)0x80000000;
            }
        }
        if (it >= 1 && it <= nt && __builtin_amdgcn_ballot_w64(mx > (LAZY ? thi : tau)) != 0ull) {
            // ---------------- rare path: some lane has a row of tile it-1 that beats (or, lazy: may beat) its threshold
            v16i lo = zero;
            if constexpr (LAZY) {
                // the low digit of tile it-1, whose LDS buffer is still intact (NBUF = 4): 24 fragment reads + 24 MFMAs,
                // software-pipelined like the main loop (reads three k-steps ahead, counted waits); the three fragments
                // already prefetched for the NEXT tile sit older in the LDS queue and simply complete first
                const unsigned pb = (unsigned)((it - 1) % NBUF) * DPH_TILE_BYTES;
                v4i f[DPH_PF + 1];
                ds_read16<0>(f[0], faddr[0] + pb);
                ds_read16<0>(f[1], faddr[1] + pb);
                ds_read16<0>(f[2], faddr[2] + pb);
                static_for<0, DPH_KSTEPS>([&](auto kc) {
                    constexpr int k2 = decltype(kc)::value, p2 = k2 + DPH_PF;
                    if constexpr (p2 < DPH_KSTEPS) {
                        ds_read16<(p2 >> 3) * 256>(f[p2 & DPH_PF], faddr[p2 & 7] + pb);
                        wait_lgkm<DPH_PF>(f[k2 & DPH_PF]);
                    } else {
                        wait_lgkm<DPH_KSTEPS - 1 - k2>(f[k2 & DPH_PF]);
                    }
                    mfma_lo(lo, f[k2 & DPH_PF], ql[k2]);
                });
                mfma_settle(lo);
            }
            const unsigned rowbase = (unsigned)(tile_of(it - 1) * tile_stride * DPH_TILE_ROWS) + 4u * (unsigned)(lane >> 5);
            unsigned done = 0;
            bool again;
            do {
                bool blocked = false;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int s = (acc_lane(ph[r]) << 7) + (LAZY ? acc_lane(lo[r]) : acc_lane(pl[r]));
                    const bool hit = (s > tau) && probed && !((done >> r) & 1u);
                    if (__builtin_amdgcn_ballot_w64(hit) != 0ull) {
                        if (hit) {
                            const unsigned row = rowbase + (unsigned)((r & 3) + 8 * (r >> 2));
                            if ((int64_t)row >= n_rows || (IVF && row_ids[row] < 0)) {
                                done |= 1u << r;                 // padding row (end of the shard / end of a list)
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
            thi = hi_threshold(tau);
        }
    };

    static_assert(DPH_NSET == 2 || DPH_NSET == 4, "the unrolled tile loop below assumes 2 or 4 staging sets");
    for (int it = 0; it <= nt; it += DPH_NSET) {
        tile_step(std::integral_constant<int, 2 % DPH_NSET>{}, accA_h, accA_l, accB_h, accB_l, it);
        tile_step(std::integral_constant<int, 3 % DPH_NSET>{}, accB_h, accB_l, accA_h, accA_l, it + 1);
        if constexpr (DPH_NSET == 4) {
            tile_step(std::integral_constant<int, 0>{}, accA_h, accA_l, accB_h, accB_l, it + 2);
            tile_step(std::integral_constant<int, 1>{}, accB_h, accB_l, accA_h, accA_l, it + 3);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // nothing of the feed may still be in flight at exit

    // ---- drain: reduce every list to its KP best (sorted), publish [block][thread][KP], 0 = empty slot
    {
        // (cnt == KP included: the select kernel reads slot KP-1 of a full list as "everything dropped is below this")
        unsigned long long over = __builtin_amdgcn_ballot_w64(cnt >= KP);
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
    return cus > 0 && cus < 256 ? cus : 256;     // one workgroup per CU; the select / threshold images hold 256
}

template <int KP, int CAP, bool SAMPLE, bool LAZY, bool IVF = false>
static void launch_scan_t(const int8_t* db, int64_t n_rows, int64_t n_tiles, int tile_stride, const int8_t* qfrag,
                          const int* tau_init, const int* lmax_q, const unsigned* tilemask, const int64_t* row_ids,
                          uint64_t* lists, int grid, hipStream_t st) {
    const size_t lds = (size_t)(LAZY ? 4 : 3) * DPH_TILE_BYTES + (size_t)DPH_SCAN_THREADS * CAP * 8;
    static bool attr_set[64] = {};       // the attribute is per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        (void)hipFuncSetAttribute((const void*)dph_scan_kernel<KP, CAP, SAMPLE, LAZY, IVF>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL((dph_scan_kernel<KP, CAP, SAMPLE, LAZY, IVF>), dim3(grid), dim3(DPH_SCAN_THREADS), lds, st, db,
                       n_rows, n_tiles, tile_stride, qfrag, tau_init, lmax_q, tilemask, row_ids, lists);
}

// n_tiles = number of tiles this launch visits (tile i of the launch is shard tile i*tile_stride).
// sample = a threshold pre-pass over a strided sample (its own kernel name in a profile).  A launch that has a
// threshold to start from (tau_init and lmax_q given) runs the lazy-low-digit kernel, one without runs the eager
// two-digit kernel (first-level pre-pass, small shards).  tilemask != NULL selects the IVF kernels (kp 16 only).
// DPH_SCAN_EAGER=1 forces the eager kernel everywhere (an A/B switch; both kernels return the same lists).
void dph_launch_scan(int kp, bool sample, const int8_t* db, int64_t n_rows, int64_t n_tiles, int tile_stride,
                     const int8_t* qfrag, const int* tau_init, const int* lmax_q, const unsigned* tilemask,
                     const int64_t* row_ids, uint64_t* lists, int grid, hipStream_t st) {
    static const bool force_eager = [] { const char* e = getenv("DPH_SCAN_EAGER"); return e && atoi(e) != 0; }();
    const bool lazy = tau_init != nullptr && lmax_q != nullptr && !force_eager;
#define DPH_ARGS db, n_rows, n_tiles, tile_stride, qfrag, tau_init, lmax_q, tilemask, row_ids, lists, grid, st
    if (tilemask) {
        if (sample && lazy) launch_scan_t<16, 24, true, true, true>(DPH_ARGS);
        else if (sample) launch_scan_t<16, 32, true, false, true>(DPH_ARGS);
        else if (lazy) launch_scan_t<16, 24, false, true, true>(DPH_ARGS);
        else launch_scan_t<16, 32, false, false, true>(DPH_ARGS);
    } else if (kp == 16) {
        if (sample && lazy) launch_scan_t<16, 24, true, true>(DPH_ARGS);
        else if (sample) launch_scan_t<16, 32, true, false>(DPH_ARGS);
        else if (lazy) launch_scan_t<16, 24, false, true>(DPH_ARGS);
        else launch_scan_t<16, 32, false, false>(DPH_ARGS);
    } else {
        if (sample) launch_scan_t<32, 40, true, false>(DPH_ARGS);
        else launch_scan_t<32, 40, false, false>(DPH_ARGS);
    }
#undef DPH_ARGS
}

// ------------------------------------------------------------------------------------------ pre-pass threshold
// One workgroup per query row of the pass: the KP-th largest integer score in that row's sample lists, minus one
// (rows scoring exactly the KP-th value must still enter), or INT_MIN when the sample holds fewer than KP rows.
// 1024 threads per row (8 or 16 keys per thread in registers).  The 32 bit-steps are a latency chain, so each step is
// kept short: wave counts come from ballots + scalar pop-counts (a ds_bpermute butterfly cost ~0.7 us per step:
// 25-30 us per launch in profiles/r01_kernel_trace_21M.csv, and the pre-pass calls this once per ladder level) and
// the cross-wave sum is double-buffered so a step has one barrier.
template <int KP, int THREADS>
__global__ __launch_bounds__(THREADS) void dph_threshold_kernel(const uint64_t* __restrict__ lists, int grid,
                                                                const int* __restrict__ floor_tau,
                                                                int* __restrict__ tau_out, int n_q,
                                                                int* __restrict__ top_out) {
    __shared__ unsigned cnt_sh[2][THREADS / 64];
    __shared__ unsigned top_cnt;
    const int qi = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int qw = qi >> 5, qc = qi & 31;
    const int n_keys = grid * 2 * KP;
    // each thread keeps its share of the biased scores in registers (the launcher checks n_keys <= THREADS * PER)
    constexpr int PER = DPH_THRESHOLD_MAX_KEYS(KP) / THREADS;
    unsigned u[PER];                                 // statically indexed: stays in registers
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int e = tid + THREADS * j;
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
        unsigned c = 0;                              // wave-uniform: ballots + scalar pop-counts, no cross-lane shuffles
#pragma unroll
        for (int j = 0; j < PER; ++j) c += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(u[j] >= cand));
        unsigned total = c;
        if constexpr (THREADS > 64) {
            // double-buffered by bit parity: one barrier per step (a wave can be at most one step ahead)
            if (lane == 0) cnt_sh[bit & 1][wv] = c;
            __syncthreads();
            total = 0;
#pragma unroll
            for (int w = 0; w < THREADS / 64; ++w) total += cnt_sh[bit & 1][w];
        }
        if (total >= (unsigned)KP) ans = cand;
    }
    if (tid == 0) {
        const int kth = (int)(ans ^ 0x80000000u);
        int t = (ans == 0u || kth == (int)0x80000000) ? (int)0x80000000 : kth - 1;
        if (floor_tau) t = max(t, floor_tau[qi]);      // a second-level sample only saw rows above the first-level bound
        if (tau_out) tau_out[qi] = t;
    }
    if (top_out && qi < n_q) {                       // (block-uniform) rows past n_q are padding of the pass
        // the KP best sampled scores themselves (any order; INT_MIN where the sample holds fewer): what a rank shares
        // so that the bound can be taken over the union of all ranks' samples (dph_union_bounds_kernel)
        if (tid == 0) top_cnt = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            if (u[j] != 0u && u[j] >= ans) {
                const unsigned slot = atomicAdd(&top_cnt, 1u);
                if (slot < (unsigned)KP) top_out[(int64_t)qi * KP + slot] = (int)(u[j] ^ 0x80000000u);
            }
        }
        __syncthreads();
        for (unsigned t = top_cnt + tid; t < (unsigned)KP; t += THREADS) top_out[(int64_t)qi * KP + t] = (int)0x80000000;
    }
}

// bound over the union of n_parts samples: the KEEP-th largest of the n_parts*KEEP shared scores of a row, minus one
// (INT_MIN when the union holds fewer).  One wave per row, rank by counting (n_parts*KEEP <= 64*PER values).
__global__ __launch_bounds__(64) void dph_union_bounds_kernel(const int* __restrict__ top_parts, int n_parts,
                                                              int64_t n, int* __restrict__ tau_out) {
    const int64_t row = blockIdx.x;
    const int lane = threadIdx.x, m = n_parts * DPH_SAMPLE_KEEP;
    int best = (int)0x80000000;
    for (int c = lane; c < m; c += 64) {
        const int p = c / DPH_SAMPLE_KEEP, i = c % DPH_SAMPLE_KEEP;
        const int v = top_parts[((int64_t)p * n + row) * DPH_SAMPLE_KEEP + i];
        if (v == (int)0x80000000) continue;
        int rank = 0;                                  // entries strictly better, ties broken by position
        for (int u = 0; u < m; ++u) {
            const int w = top_parts[((int64_t)(u / DPH_SAMPLE_KEEP) * n + row) * DPH_SAMPLE_KEEP + (u % DPH_SAMPLE_KEEP)];
            rank += (w != (int)0x80000000 && (w > v || (w == v && u < c))) ? 1 : 0;
        }
        if (rank == DPH_SAMPLE_KEEP - 1) best = v - 1; // exactly one entry has this rank
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o));
    if (lane == 0) tau_out[row] = best;
}

void dph_launch_union_bounds(const int* top_parts, int n_parts, int64_t n, int* tau_out, hipStream_t st) {
    hipLaunchKernelGGL(dph_union_bounds_kernel, dim3((unsigned)n), dim3(64), 0, st, top_parts, n_parts, n, tau_out);
}

int dph_launch_threshold(int kp, const uint64_t* lists, int grid, const int* floor_tau, int* tau_out, int n_q, int* top_out,
                         hipStream_t st) {
    if (grid * 2 * kp > DPH_THRESHOLD_MAX_KEYS(kp)) return -1;       // more workgroups than the register image holds
    if (top_out && kp != DPH_SAMPLE_KEEP) return -1;
    if (kp == 16) hipLaunchKernelGGL((dph_threshold_kernel<16, 1024>), dim3(DPH_QROWS), dim3(1024), 0, st, lists, grid, floor_tau, tau_out, n_q, top_out);
    else hipLaunchKernelGGL((dph_threshold_kernel<32, 1024>), dim3(DPH_QROWS), dim3(1024), 0, st, lists, grid, floor_tau, tau_out, n_q, top_out);
    return 0;
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
                                                          const int64_t* __restrict__ row_ids,
                                                          unsigned long long* __restrict__ max_out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    int best = 0;
    for (int64_t row = wave0; row < n_rows; row += nwaves) {
        if (row_ids && row_ids[row] < 0) continue;      // list padding: not a row of the dump
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
void dph_launch_rownorm(const int8_t* db, int64_t n_rows, const int64_t* row_ids, unsigned long long* max_out,
                        hipStream_t st) {
    hipLaunchKernelGGL(dph_rownorm_kernel, dim3(256 * 8), dim3(256), 0, st, db, n_rows, row_ids, max_out);
}
