// dph_scan.hip -- the headline kernel: brute-force inner-product FILTER scan over the int8 phrase dump (replaces the
// inner loop of faiss Index.search at /root/reference/densephrases/index.py:200).  gfx950 / CDNA4 only.
//
// Arithmetic.  The reference searches fp32 vectors x = n/20 - 2 de-quantised from int8 n (embed_utils.py:141-149).
// <q, x> = <q, n>/20 - 2*sum(q), so ranking rows by <q, n> is ranking by score.  Each query row is a two-digit
// fixed-point number  q_j = sc*(128*q1_j + q2_j) + e_j  with int8 digits |q1| <= 127, |q2| <= 64; the exact integer
// score of a row is I = 128*H + L, H = <q1, n>, L = <q2, n>, and L <= lmax := ||q2||_2 * max_row||n - c||_2 + c*sum(q2)
// (Cauchy-Schwarz with a shard constant).  A row whose HIGH digit alone satisfies 128*H + lmax <= tau cannot have
// I > tau, so this kernel multiplies ONLY the high digit (one v_mfma_i32_32x32x32_i8 per 32 rows x 32 queries x 32 k:
// the database bytes are MFMA operands as they lie in HBM, no conversion) and EMITS every (row, query row) pair with
// H > floor((tau - lmax)/128).  Everything else provably has I <= tau.  The pairs (a few thousand per query row behind
// a sampled bound, see the ladder in dph_api.hip) get their exact integer score in dph_refine_kernel and their exact
// fp64 score + exactness certificate in dph_select_kernel; nothing in here keeps lists, sorts or prunes.
//
// Structure.  256 threads = 4 waves, one per SIMD, one workgroup per CU; wave w owns QB groups of 32 query rows for
// the whole launch (QB = 1: 128 rows per pass, high digit in 96 VGPRs; QB = 2: 256 rows per pass, the second group's
// digit in AGPRs).  Database tiles of 32 rows (24 KiB, contiguous in HBM) reach the workgroups in contiguous segments from a
// work queue (below).
// Feed: every wave loads a quarter of every tile (6 x 1 KiB, coalesced global_load_dwordx4) into AGPRs it owns BY HAND
// (NSET staging sets = NSET tiles in flight per wave), and writes them into an XOR-swizzled LDS image with
// ds_write_b128 one hand-over later; the hand-over (counted s_waitcnt vmcnt, raw s_barrier) sits mid-tile.  (LDS-DMA
// was measured and dropped: its pieces count on lgkmcnt too and stall the counted ds_read waits)
// The 768-byte row stride would put a whole ds_read_b128 lane group on one bank slot; the k order inside a dot product
// is free as long as query and database agree, so the image is swizzled per row (chunk' = chunk ^ (row & 15) inside
// each 256-byte third) and the fragment reads are conflict-free.
// MFMA tile: A = 32 database rows x 32 k (LDS), B = 32 k x 32 query rows (registers): lane l ends a tile holding, for
// query row (l & 31) of each of its groups, the 16 scores of rows i(r) = (r&3) + 8*(r>>2) + 4*(l>>5), r = 0..15.
// Two accumulator sets alternate so the threshold test of tile t (one max per score) rides between the MFMAs of t+1.
#include "dph_internal.h"
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
// Fragment reads are issued by hand so that their completion can be awaited with a COUNTED lgkmcnt: hipcc's own
// bookkeeping waits lgkmcnt(0) in this loop, i.e. for the read it issued a moment ago.  "=v" load, then a wait
// statement that names the destination "+v", which is also what keeps the consuming MFMA below the wait.
template <int OFF>
__device__ __forceinline__ void ds_read16(v4i& dst, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
}
template <int N>
__device__ __forceinline__ void wait_lgkm(v4i& dst) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(dst) : "i"(N));
}
// The MFMAs are written by hand as well, for register placement: the accumulators must be VGPRs (the threshold test
// reads them with plain v_max; hipcc puts them in AGPRs and pays a v_accvgpr_read per score) and the second query
// group must be AGPRs (there is no room for 192 query registers next to the accumulators in the 256 VGPRs).  With one
// wave per SIMD the instruction stream of that wave is the limit as soon as the matrix pipe is busy > 50 %: the loop
// below issues ~150 instructions per 48 MFMAs.  Hazards hipcc no longer sees: an accumulator is read by the VALU at
// least four k-steps (8 MFMAs) after the MFMA that wrote it, and dependent MFMAs on one accumulator need no wait.
#if DPH_SCAN_DIAG & 1024
// timing experiment: the matrix work of a k-step as TWO v_mfma_i32_16x16x64_i8 (the shape the guide's int8 ceiling was
// measured with: 4 accumulator registers, 64 k per instruction) on the same operand registers -- same MACs, same feed, garbage results
// (the real accumulators stay zero: nothing is emitted)
__device__ v4i dph_diag_sink;
template <bool FIRST, bool B_IN_AGPR>
__device__ __forceinline__ void mfma_i8(v16i& acc, const v4i& a, const v4i& b) {
    (void)acc;
    static_assert(sizeof(v4i) == 16, "");
    v4i d0, d1;
    asm volatile("v_mov_b32 %0, 0" : "=v"(d0[0]));      // (never read: the registers only have to exist)
    if constexpr (B_IN_AGPR) {
        asm volatile("v_mfma_i32_16x16x64_i8 %0, %2, %3, 0\n\tv_mfma_i32_16x16x64_i8 %1, %2, %3, 0" : "=&v"(d0), "=&v"(d1) : "v"(a), "a"(b));
    } else {
        asm volatile("v_mfma_i32_16x16x64_i8 %0, %2, %3, 0\n\tv_mfma_i32_16x16x64_i8 %1, %2, %3, 0" : "=&v"(d0), "=&v"(d1) : "v"(a), "v"(b));
    }
    asm volatile("" ::"v"(d0), "v"(d1));
}
#else
template <bool FIRST, bool B_IN_AGPR>
__device__ __forceinline__ void mfma_i8(v16i& acc, const v4i& a, const v4i& b) {
#if DPH_SCAN_DIAG & 512         // timing experiment: raise the wave's priority around its MFMAs (one wave per SIMD: expected to change nothing)
    asm volatile("s_setprio 3" ::: "memory");
#endif
    if constexpr (FIRST) {
        if constexpr (B_IN_AGPR) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b));
        else asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
    } else {
        if constexpr (B_IN_AGPR) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
        else asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    }
#if DPH_SCAN_DIAG & 512
    asm volatile("s_setprio 0" ::: "memory");
#endif
}
#endif

// Round 6: the int8 scans without aux rows issue their matrix work as v_mfma_i32_16x16x64_i8 -- the shape the guide's int8 ceiling is
// measured with.  Same MACs, same operand bytes, same number of fragment reads per tile; measured on the 256-row pass with the product
// feed and discarded results (tools/scan_diag.py variant 1024): 25.9 ms against 30.2 per 170 M-row launch, matrix work alone 13.6 ms
// against 18.7 (profiles/r06_scan_diag_256rows_variants.json).  A tile of 32 rows x a group of 32 query rows is four 16 x 16 blocks
// (database rows 0-15 / 16-31 x query rows 0-15 / 16-31), each accumulated over twelve slabs of 64 k.  Tile step k-step ks multiplies
// slab ks >> 1 of the row half ks & 1: ONE fragment read as before (lane l: row 16 (ks & 1) + (l & 15), bytes 64 (ks >> 1) +
// 16 (l >> 4) .. + 15 of it), two MFMAs per query group (left / right query half).  Lane l ends a tile holding, per block, rows
// 4 (l >> 4) + {0, 1, 2, 3} of the block's row half for query row l & 15 of its query half.  Aux shards (the 25th k-step has its
// operand in the 32 x 32 x 32 fragment order) and the bf16 coarse filter keep the 32 x 32 forms.
template <bool FIRST, bool B_IN_AGPR>
__device__ __forceinline__ void mfma_i8x16(v4i& acc, const v4i& a, const v4i& b) {
    if constexpr (FIRST) {
        if constexpr (B_IN_AGPR) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b));
        else asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
    } else {
        if constexpr (B_IN_AGPR) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
        else asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    }
}
// the accumulators of one query group: sixteen registers either way -- one 32 x 32 block, or four 16 x 16 blocks [row half][query half]
struct acc_x16 { v4i s[4]; };
template <bool X16> struct acc_of { using type = v16i; };
template <> struct acc_of<true> { using type = acc_x16; };
__device__ __forceinline__ int acc_get(const v16i& a, int i) { return a[i]; }
__device__ __forceinline__ int acc_get(const acc_x16& a, int i) { return a.s[i >> 2][i & 3]; }

// MODE 3 (the coarse quantizer of a PQ index as a filter scan, below): the same tile bytes are 32 rows x 384 bf16, the MFMA is
// v_mfma_f32_32x32x16_bf16 (A = 32 rows x 16 k from LDS, B = 16 k x 32 query rows from registers), fp32 accumulators in the same VGPRs
template <bool FIRST, bool B_IN_AGPR>
__device__ __forceinline__ void mfma_bf16(v16i& acc, const v4i& a, const v4i& b) {
    if constexpr (FIRST) {
        if constexpr (B_IN_AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "a"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
    } else {
        if constexpr (B_IN_AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    }
}

// PF = how many k-steps ahead the ds_read_b128 of a database fragment is issued (register ring of PF+1; a depth of 5
// was measured and changes nothing: the LDS latency is covered),
// KSYNC = the k-step of tile `it` at which the hand-over for tile it+1 happens.
#define DPH_PF 3
#define DPH_KSYNC 12
// Timing experiments only (tools/scan_diag.py builds copies of the library with -DDPH_SCAN_DIAG=bits; the product is built
// with 0): leave one ingredient of the streaming loop out to see what it costs.  Results of such a build are garbage.
//   1 no hand-over barrier | 2 no vmcnt wait | 4 no global loads | 8 no staging writes | 16 no fragment reads | 32 no threshold max
//   64 staging writes from arch VGPRs instead of AGPRs | 128 staging writes as two ds_write_b64 | 256 tile loads WITHOUT the non-temporal hint
//   512 s_setprio 3 / 0 around every MFMA | 1024 the matrix work as two v_mfma_i32_16x16x64_i8 per k-step and group (results discarded)
#ifndef DPH_SCAN_DIAG
#define DPH_SCAN_DIAG 0
#endif
// four tile buffers | the work-queue slots | a scratch kilobyte (the prologue's stand-in staging writes, SCHED != 0)
#define DPH_SCAN_LDS_BYTES ((size_t)4 * DPH_TILE_BYTES + 64 + 1024)
constexpr int staged_at(int ks) { return (ks >= DPH_KSYNC && ks < DPH_KSYNC + 6) ? 1 : 0; }   // a ds_write_b128 in that k-step

// ---- hand-over schedules ------------------------------------------------------------------------------------------
// When does wave W hand piece i of tile j+2 over (ds_write to LDS, then re-load of its staging registers one k-step later)?
// As a position counted in k-steps from the start of tile j: the window is [KSYNC, KSYNC + 24) -- after the barrier of tile j
// (tile j-2, whose buffer is written, is consumed by everyone) and before the barrier of tile j+1 (which publishes tile j+2).
//   SCHED 0  12 + i            every wave right behind the barrier (rounds 1-3)
//   SCHED 1  12 + 6 W + i      one wave after the other, six k-steps each
//   SCHED 2  12 + 4 i + W      interleaved: in every k-step exactly ONE wave of the workgroup writes a piece
// Why: the four waves run in lock step (one barrier per tile), so under SCHED 0 all four issue their global_load_dwordx4 /
// ds_write_b128 in the same k-steps and queue behind each other on the CU's one vector-memory / LDS-store path while their
// matrix pipes drain: at 256 query rows the feed instructions cost 12 of the launch's 30 ms (profiles/r04_scan_diag_*: 18.0 ms
// without loads and staging writes, 22.0 without the loads, 26.2 without the writes).  A position >= 24 lies in the NEXT tile step:
// there the piece belongs to the tile after the one being multiplied.
constexpr int hand_pos(int sched, int w, int i) { return sched == 1 ? DPH_KSYNC + 6 * w + i : (sched == 2 ? DPH_KSYNC + 4 * i + w : DPH_KSYNC + i); }
// piece written / re-loaded in k-step ks of a tile step, for the tile `2 - d` ahead of the one being multiplied (-1: none)
constexpr int wr_piece(int sched, int w, int ks, int d) {
    for (int i = 0; i < 6; ++i) if (hand_pos(sched, w, i) == ks + 24 * d) return i;
    return -1;
}
constexpr int ld_piece(int sched, int w, int ks, int d) {
    for (int i = 0; i < 6; ++i) if (hand_pos(sched, w, i) + 1 == ks + 24 * d) return i;
    return -1;
}
// loads younger than the load of piece i of the set being handed over, at the moment that piece is written: the rest of that set,
// the three younger sets, and the re-loads this hand-over has issued already (in an earlier k-step: within a k-step the write comes
// first) -- the count of the s_waitcnt vmcnt in front of the write
constexpr int younger_loads(int sched, int w, int i, int nset) {
    int n = 6 * (nset - 1) + (5 - i);
    for (int j = 0; j < 6; ++j) if (hand_pos(sched, w, j) + 1 < hand_pos(sched, w, i)) ++n;
    return n;
}
constexpr int writes_at(int sched, int w, int ks) {      // ds_write_b128 of this wave in k-step ks (any integer: periodic in 24)
    const int k = ((ks % 24) + 24) % 24;
    return (wr_piece(sched, w, k, 0) >= 0 || wr_piece(sched, w, k, 1) >= 0) ? 1 : 0;
}

// ---- hand-owned accumulator registers -------------------------------------------------------------------------------
// staging set S in 0..NSET-1, piece i in 0..5 -> a[STG0 + 24*S + 4*i .. +3], STG0 = 256 - 24*NSET; the IVF probe masks
// of the even / odd tile live in a[STG0-4 .. STG0-1].  The range is named literally in the asm below and listed as
// clobbered once (stage_claim) so that the kernel descriptor covers it; tools/audit_scan_isa.py (run by
// tests/test_abi.py) checks that the compiler never touches it outside the asm statements.
template <int NSET>
struct stg {
    static constexpr int STG0 = 256 - 24 * NSET;
    static constexpr int MASK0 = STG0 - 4;
    static constexpr int AUX0 = MASK0 - 32;      // the A operands of the aux k-step (below), a ring of four tiles x two row halves: a[124:155]
                                                 // (32 x 32 x 32 kernels: one operand per tile, the first 16 bytes of each pair)
};
// ---- the aux k-step (round 5) -----------------------------------------------------------------------------------------
// A 25th k-step whose A operand does not come from the dump but from the shard's AUX ROWS (dph_quant.hip dph_aux_build_kernel:
// `stride` bytes per stored row, written by dph_index_finalize) and whose B operand is the query row's aux digits
// (dph_quantize_kernel): slot s of both.
//   norm slots     A = a code of the row's centred norm || n - mu ||_2 (their sum = ceil(norm / unit)), B = ceil(unit * ||q2||_2 / 128):
//                  the product bounds the row's OWN low-digit term <q2, n - mu> / 128 from above (Cauchy-Schwarz per row instead of
//                  one shard constant: a heavy-tailed norm distribution loosens the filter for the heavy rows only);
//   replica slots  A = the raw code of a ROGUE dimension (one whose mean code sits far from the others' for every row, as in
//                  BERT-family vectors), B = a further high digit of the query in that dimension: the query's scale is then set by
//                  the bulk of its dimensions, not by the few rogue ones (dph_quant.hip; exact part of the score, refined alike).
// So the accumulators end a tile holding H' = <q1, n> + sum_s A_s B_s and the test H' > floor((tau - <q2, mu>) / 128) is a PER-ROW
// bound.  Lane l loads 16 bytes: row l & 31, slots 16 (l >> 5) .. +15 (stride 4: slots 0..3 are the row's, what follows belongs
// to the next rows and meets zero digits).  Issued at the top of tile step `it` for tile it + 3 into a ring of four A operands
// (S = tile mod 4), awaited with a counted vmcnt at the top of the step that multiplies it: three steps of 6 staged pieces and two
// aux loads younger than it stay in flight.
template <int NSET, int S, int RH = 0>
__device__ __forceinline__ void aux_load(unsigned voff, const int8_t* base) {
    constexpr int r = stg<NSET>::AUX0 + 8 * S + 4 * RH;
    asm volatile("global_load_dwordx4 a[%c2:%c3], %0, %1" ::"v"(voff), "s"(base), "i"(r), "i"(r + 3) : "memory");
}
template <int NSET, int S, bool B_IN_AGPR>
__device__ __forceinline__ void mfma_aux(v16i& acc, const v4i& b) {
    constexpr int r = stg<NSET>::AUX0 + 8 * S;
    if constexpr (B_IN_AGPR) asm volatile("v_mfma_i32_32x32x32_i8 %0, a[%c1:%c2], %3, 0" : "=&v"(acc) : "i"(r), "i"(r + 3), "a"(b));
    else asm volatile("v_mfma_i32_32x32x32_i8 %0, a[%c1:%c2], %3, 0" : "=&v"(acc) : "i"(r), "i"(r + 3), "v"(b));
}
// 16 x 16 x 64 kernels: the aux rows of a tile are TWO operands (row half RH: lane l holds slots 16 (l >> 4) .. + 15 of row
// 16 RH + (l & 15) -- slots past the layout's belong to following rows and meet zero digits, as in the 32 x 32 x 32 form), each multiplied
// with the left and the right query half's aux digits: the four blocks of a group open their accumulation with these
template <int NSET, int S, int RH, bool B_IN_AGPR>
__device__ __forceinline__ void mfma_aux16(v4i& acc, const v4i& b) {
    constexpr int r = stg<NSET>::AUX0 + 8 * S + 4 * RH;
    if constexpr (B_IN_AGPR) asm volatile("v_mfma_i32_16x16x64_i8 %0, a[%c1:%c2], %3, 0" : "=&v"(acc) : "i"(r), "i"(r + 3), "a"(b));
    else asm volatile("v_mfma_i32_16x16x64_i8 %0, a[%c1:%c2], %3, 0" : "=&v"(acc) : "i"(r), "i"(r + 3), "v"(b));
}
// aux loads (issued at the top of every tile step, in front of its k-step 0) younger than the load of piece i -- re-loaded at
// k-position hand_pos + 1 four tile steps before it is written at hand_pos -- at the moment of that write
constexpr int aux_younger(int sched, int w, int i) { return (hand_pos(sched, w, i) + 96) / 24 - (hand_pos(sched, w, i) + 1) / 24; }
template <int NSET, int S, int I, bool NT = true>
__device__ __forceinline__ void stage_load(unsigned voff, const int8_t* base) {
    constexpr int r = stg<NSET>::STG0 + 24 * S + 4 * I;
    if constexpr (!NT) {     // MODE 4 (teams of workgroups re-read the same tiles through their XCD's L2): default cache policy
        asm volatile("global_load_dwordx4 a[%c2:%c3], %0, %1" ::"v"(voff), "s"(base), "i"(r), "i"(r + 3) : "memory");
        return;
    }
    // non-temporal: every byte of the dump is read once per launch, by one CU -- nothing to keep in L2 / MALL (round 4: 20.37 ->
    // 19.91 ms per 170 M-row launch at 128 query rows = 6.56 TB/s, 30.25 -> 29.94 at 256; MI355X_MICROARCH.md "nt-weights")
#if DPH_SCAN_DIAG & 256         // timing experiment: the default cache policy of rounds 1-3
    asm volatile("global_load_dwordx4 a[%c2:%c3], %0, %1" ::"v"(voff), "s"(base), "i"(r), "i"(r + 3) : "memory");
#else
    asm volatile("global_load_dwordx4 a[%c2:%c3], %0, %1 nt" ::"v"(voff), "s"(base), "i"(r), "i"(r + 3) : "memory");
#endif
}
template <int NSET, int S, int I, int OFF>
__device__ __forceinline__ void stage_write(unsigned lds_addr) {
    constexpr int r = stg<NSET>::STG0 + 24 * S + 4 * I;
#if DPH_SCAN_DIAG & 64          // timing experiment: the same store from arch VGPRs (whatever they hold)
    asm volatile("ds_write_b128 %0, v[%c1:%c2] offset:%c3" ::"v"(lds_addr), "i"(64 + 4 * I), "i"(64 + 4 * I + 3), "i"(OFF) : "memory");
#elif DPH_SCAN_DIAG & 128       // timing experiment: two 8-byte stores instead of one 16-byte store
    asm volatile("ds_write_b64 %0, a[%c1:%c2] offset:%c3\n\tds_write_b64 %0, a[%c4:%c5] offset:%c6" ::"v"(lds_addr), "i"(r), "i"(r + 1), "i"(OFF),
                 "i"(r + 2), "i"(r + 3), "i"(OFF + 8) : "memory");
#else
    asm volatile("ds_write_b128 %0, a[%c1:%c2] offset:%c3" ::"v"(lds_addr), "i"(r), "i"(r + 3), "i"(OFF) : "memory");
#endif
}
// IVF probe mask of a tile: QB dwords per wave (bit j of dword g = query row 32*(wave*QB+g) + j probes the tile's
// list), fetched with a hand-written load one tile ahead of its use.  It must not be a compiler-visible load: hipcc
// would wait for it with a vmcnt that also drains the staged tiles in flight.
template <int NSET, int QB, int PARITY>
__device__ __forceinline__ void mask_load(unsigned zero_off, const unsigned* addr) {
    constexpr int r = stg<NSET>::MASK0 + 2 * PARITY;
    if constexpr (QB == 1) asm volatile("global_load_dword a[%c2], %0, %1" ::"v"(zero_off), "s"(addr), "i"(r) : "memory");
    else asm volatile("global_load_dwordx2 a[%c2:%c3], %0, %1" ::"v"(zero_off), "s"(addr), "i"(r), "i"(r + 1) : "memory");
}
// 13 VMEM loads are issued after that load before it is read (two hand-overs of 6 and the next tile's mask) -- always:
// the feed never stops loading, past the end of the shard it re-loads the last tile.  (Pair stores of the emit path
// also count on vmcnt; they only make a counted wait more conservative, never too short: loads return in order.)
// (AUX: the aux rows loaded at the top of the tile step that reads the mask are one more)
// (AUXL = aux loads per tile step: 0, 1, or 2 in the 16 x 16 x 64 kernels)
template <int NSET, int PARITY, int G, int AUXL>
__device__ __forceinline__ unsigned mask_read() {
    constexpr int r = stg<NSET>::MASK0 + 2 * PARITY + G;
    unsigned m;
    asm volatile("s_waitcnt vmcnt(%c2)\n\tv_accvgpr_read_b32 %0, a[%c1]" : "=v"(m) : "i"(r), "i"(13 + AUXL) : "memory");
    return m;
}
#define DPH_A10(d) "a" #d "0", "a" #d "1", "a" #d "2", "a" #d "3", "a" #d "4", "a" #d "5", "a" #d "6", "a" #d "7", "a" #d "8", "a" #d "9"
#define DPH_A160_255 DPH_A10(16), DPH_A10(17), DPH_A10(18), DPH_A10(19), DPH_A10(20), DPH_A10(21), DPH_A10(22), DPH_A10(23), \
                     DPH_A10(24), "a250", "a251", "a252", "a253", "a254", "a255"
// tells the compiler the hand-owned range exists (kernel descriptor) and is off limits at this point
template <int NSET>
__device__ __forceinline__ void stage_claim() {
    static_assert(NSET == 4, "staging sets");
    asm volatile("" ::: "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139",
                 "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154",
                 "a155", "a156", "a157", "a158", "a159", DPH_A160_255);
}

// MODE 0: flat shard.  MODE 1: the shard is stored list-major (every tile belongs to one inverted list) and
// `tilemask[8*tile + g]` says which rows of query group g probe that tile's list; rows of unprobed lists are never
// emitted -- exact in-list inner product over the probed lists only (FAISS IndexIVFFlat semantics).
// MODE 2: the same semantics from a WORK QUEUE of units (dph_internal.h): the workgroup takes (chunk, segment) units
// until the queue is empty; a unit multiplies the tiles of its list segment with the high digits of the <= 128 query
// rows that PROBE that list (gathered into fragment order by dph_units_gather_kernel), so a pass serves up to
// DPH_PASS_MAX query rows with the matrix work of 128 -- and lists nobody probes are never read.
// MODE 3: the COARSE QUANTIZER of a PQ index (dph_ivf.hip dph_launch_coarse_filter, tuning key coarse_filter = 5) through the same
// feed.  The "database" is the bf16 image of the coarse centroids, [tile of 32 lists][half of k][32 rows][384 bf16]: every 24 KiB
// piece has the byte layout of an int8 tile (32 rows x 768 bytes), so loads, staging, swizzle, fragment reads and every counted wait
// are the flat scan's; a tile of lists is TWO consecutive pieces (k 0..383, then 384..767) accumulated into one set of fp32
// accumulators (even step: first MFMA clears them and the previous tile is tested, odd step: accumulate), the wave's 32 query rows
// have one fragment group per half (QB = 2: half 0 in VGPRs, half 1 in AGPRs), the threshold is a float per query row
// (<x~, c~> >= estimate, the filter GEMM's epilogue test) and a hit is emitted as (list | query row << 20, order-preserving key of
// the score).  What the one-product GEMM kernels could not do at 4.4 TB/s -- stream 1.6 GB of centroids at the scan's rate.
template <int QB, int NSET, int MODE, int ROLE, int SCHED = 0, bool AUX = false>
__device__ __forceinline__ void dph_scan_body(
    const int8_t* __restrict__ db, int64_t n_rows, int64_t n_tiles, int tile_stride, const int8_t* __restrict__ qfrag,
    int n_q_host, const int* __restrict__ gate, int gate_base, const int* __restrict__ tau, const int* __restrict__ lmax_q,
    const unsigned* __restrict__ tilemask, uint2* __restrict__ pairs, unsigned* __restrict__ wave_counts,
    const int4* __restrict__ unit_recs, const int* __restrict__ unit_counts, int* __restrict__ unit_next,
    const int* __restrict__ slot_q, unsigned rowmask, int seg_tiles, const int64_t* __restrict__ row_ids,
    unsigned* __restrict__ pool_head, unsigned* __restrict__ chunk_fill, unsigned* __restrict__ overflow, unsigned skip_m,
    const int8_t* __restrict__ aux, int aux_stride, const int8_t* __restrict__ qaux) {
    constexpr bool IVF = MODE == 1;
    constexpr bool UNITS = MODE == 2;
    constexpr bool CFM = MODE == 3 || MODE == 4;
    // MODE 4 = MODE 3 for a pass of MORE than 128 query rows in ONE launch (round 6).  The CU cannot hold more than 128 rows of bf16
    // query fragments (192 registers per lane), so a longer pass used to be one launch -- one read of the 1.6 GB image from HBM -- per
    // 128 rows: 8 reads for the 1024 rows of a batch of 512.  Here the grid is cut into TEAMS of G workgroups on one XCD (workgroups
    // b, b + 8, ... share an XCD; G = groups of 128 rows, gate_base carries it): the team's members stream the SAME run of tiles,
    // each against its own group of query rows, at the same time -- the first to ask brings a tile into the XCD's L2, the others hit
    // it (default cache policy instead of `nt`).  No synchronisation: a member that drifts ahead just pays HBM for what it reads
    // first.  Static runs (one per team) instead of the work queue; hits carry the row's number in the whole pass.
    constexpr bool TEAMS = MODE == 4;
    constexpr bool NT_LOADS = !TEAMS;
    constexpr bool X16 = DPH_SCAN_X16 != 0 && !CFM;               // 16 x 16 x 64 MFMAs (above); the query fragments are in that order (dph_quantize_kernel)
    constexpr int AUXL = AUX ? (X16 ? 2 : 1) : 0;                 // aux loads a tile step issues: every counted wait below has them in its count
    constexpr int NH = X16 ? 2 : 1;                                // query rows a lane holds scores of, per group: (l & 15) of the left / right half
    using acc_t = typename acc_of<X16>::type;
    static_assert(!UNITS || QB == 1, "a unit is 128 slots");
    static_assert(!(MODE == 3 || MODE == 4) || (QB == 2 && !AUX && SCHED == 0), "the coarse filter scan: two k halves, no aux rows");
    static_assert(SCHED == 0 || MODE == 0, "the staggered hand-over schedules are built for the flat scan");
    static_assert(SCHED >= 0 && SCHED <= 2, "hand-over schedule");
    // tile t lives in LDS buffer t % 4: being read | published | being written | free.  Four buffers (not three) make
    // every buffer index a compile-time constant of the loop unrolled over the NSET staging sets: all LDS addresses
    // are a base register + an immediate offset.
    constexpr int NBUF = 4;
    static_assert(NSET % NBUF == 0, "the unrolled loop must cycle the LDS buffers");
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [4][24576]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_q = TEAMS ? n_q_host : dph_gated_rows(gate, gate_base, n_q_host);
    const int n_groups = TEAMS ? gate_base : 1;
    const int grp = TEAMS ? (int)((blockIdx.x >> 3) % (unsigned)n_groups) : 0;
    const int team = TEAMS ? (int)((blockIdx.x >> 3) / (unsigned)n_groups) * 8 + (int)(blockIdx.x & 7u) : 0;
    unsigned* const my_counts = wave_counts + ((int64_t)blockIdx.x * 4 + wave) * 2;
    if (n_q <= 0) {                       // gated retry pass with nothing to do
        if (lane == 0) { my_counts[0] = 0; my_counts[1] = 0; }
        return;
    }
    stage_claim<NSET>();
    // ---- where this wave's pairs go: chunks of DPH_CHUNK_PAIRS claimed from the launch-wide pool (dph_internal.h), one
    //      atomic per claim.  All wave-uniform: `chunk` = the chunk being filled (none yet: ~0u), `fill` = pairs in it
    //      (DPH_CHUNK_PAIRS while there is none, so the first emit claims), `cnt` = pairs emitted in total (statistics).
    unsigned chunk = 0xFFFFFFFFu, fill = DPH_CHUNK_PAIRS;
    unsigned cnt = 0, triggers = 0;       // wave-uniform
    // one ballot's worth of pairs (at most one per lane).  A ballot that does not fit the chunk fills it up, closes it
    // (its fill count is what the refine step reads), claims the next one and puts the rest there; the returning atomic
    // drains the feed once per DPH_CHUNK_PAIRS pairs of this wave (~1 us), i.e. about once per launch behind a sampled
    // bound.  With the pool exhausted the pairs are dropped and their query rows flagged: they go through the retry.
    auto emit_pairs = [&](bool emit, unsigned row, unsigned qrow) {
        const unsigned long long e = __builtin_amdgcn_ballot_w64(emit);
        const unsigned ne = (unsigned)__builtin_popcountll(e);
        if (ne == 0u) return;
        const unsigned pos = fill + __builtin_amdgcn_mbcnt_hi((unsigned)(e >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)e, 0u));
        if (fill + ne <= (unsigned)DPH_CHUNK_PAIRS) {
            if (emit) pairs[(size_t)chunk * DPH_CHUNK_PAIRS + pos] = make_uint2(row, qrow);
            fill += ne;
        } else {
            if (emit && pos < (unsigned)DPH_CHUNK_PAIRS) pairs[(size_t)chunk * DPH_CHUNK_PAIRS + pos] = make_uint2(row, qrow);
            if (chunk != 0xFFFFFFFFu && lane == 0) chunk_fill[chunk] = DPH_CHUNK_PAIRS;
            unsigned c = 0;
            if (lane == 0) c = atomicAdd(pool_head, 1u);
            c = (unsigned)__builtin_amdgcn_readfirstlane((int)c);
            if (c < (unsigned)DPH_POOL_CHUNKS) {
                if (emit && pos >= (unsigned)DPH_CHUNK_PAIRS) pairs[(size_t)c * DPH_CHUNK_PAIRS + (pos - DPH_CHUNK_PAIRS)] = make_uint2(row, qrow);
                chunk = c;
                fill = fill + ne - DPH_CHUNK_PAIRS;
            } else {
                if (emit && pos >= (unsigned)DPH_CHUNK_PAIRS) overflow[CFM ? 0u : qrow] = 1u;     // (MODE 3: `qrow` carries the score key)
                chunk = 0xFFFFFFFFu;
                fill = DPH_CHUNK_PAIRS;
            }
        }
        cnt += ne;
    };
    const unsigned n_rows_u = (unsigned)n_rows;
    // the segment taken from the queue, double-buffered: [2][8] ints, [0] = unit number
    int* const s_unit = (int*)(smem + 4 * DPH_TILE_BYTES);
    // Thread 0 runs the queue AHEAD of the workgroup so that no trip waits for the pop: the atomic that takes the segment
    // of trip t+1 is issued inside trip t, right after its prologue (whose vmcnt(0) it must not sit in front of), and its
    // result is read a whole segment later.  (dph_scan.hip is compiled with the atomic optimizer off: its wave-reduction
    // form reads the result back at once.)  It is a compiler-visible memory operation in flight across the hand-scheduled
    // loop: the loop's counted vmcnt waits only become more conservative (older operations complete first).
    int q_next = 0;                       // thread 0: the unit number of the coming trip (atomic in flight)
    // SCHED != 0 (flat shards: the probe-mask registers a[156:157] of the hand-owned range are free): the atomic is written by hand
    // and returns into a156, read back behind a hand-written wait at the top of the next trip.  As a compiler-visible atomic its
    // pending result made hipcc drain the feed (s_waitcnt vmcnt(0)) inside the streaming loop of the wave that holds thread 0,
    // where it re-used the result's register (tools/audit_scan_isa.py caught it).
    if constexpr (SCHED != 0) asm volatile("v_accvgpr_write_b32 a157, 1" ::: "memory");
    auto queue_pop = [&]() {
        if constexpr (TEAMS) return;                  // (static runs: no queue)
        if constexpr (SCHED != 0) {
            if (tid == 0) asm volatile("global_atomic_add a156, %0, a157, %1 sc0" ::"v"(0u), "s"(unit_next) : "memory");
        } else {
            if (tid == 0) q_next = atomicAdd(unit_next, 1);
        }
    };
    auto queue_result = [&]() {
        if constexpr (TEAMS) return 0;
        if constexpr (SCHED != 0) asm volatile("s_waitcnt vmcnt(0)\n\tv_accvgpr_read_b32 %0, a156" : "=v"(q_next) : : "memory");
        return q_next;
    };
    queue_pop();
    // ---- this wave's query groups (high digit), resident in registers: group 0 in VGPRs, group 1 (QB = 2) in AGPRs --
    //      the "a" operand of its MFMAs puts it there.  MODE 0 / 1: loaded once per launch; MODE 2: per unit (the
    //      gathered groups of the unit's chunk).
    // ---- per query row: emit a database row iff its high-digit score H > thi  <=>  128*H + lmax > tau.
    //      No bound (cold start) = everything; rows past n_q (padding of the pass) and empty columns = nothing.
    v4i qh[QB][DPH_KSTEPS];
    v4i qa[QB][NH];                       // AUX: the B operand of the aux k-step (slots 16 (lane >> 5 | lane >> 4) .. +15 of the lane's query row(s))
    int thi[QB][NH];
    int my_qrow[QB][NH];                  // query row of the pass in this lane's MFMA column (X16: of the left and of the right query half)
    auto load_queries = [&](int chunk) {
        const v4i* qf = (const v4i*)qfrag;
        if constexpr (UNITS) qf += (int64_t)chunk * (4 * DPH_KSTEPS * 64);           // the chunk's four gathered groups
        if constexpr (TEAMS) qf += (int64_t)grp * (4 * QB * DPH_KSTEPS * 64);         // this workgroup's group of 128 query rows
#pragma unroll
        for (int g = 0; g < QB; ++g)
#pragma unroll
            for (int ks = 0; ks < DPH_KSTEPS; ++ks) qh[g][ks] = qf[((wave * QB + g) * DPH_KSTEPS + ks) * 64 + lane];
#pragma unroll
        for (int g = 0; g < QB; ++g)
#pragma unroll
        for (int hq = 0; hq < NH; ++hq) {
            int qrow = (wave * QB + g) * DPH_QGROUP + (X16 ? 16 * hq + (lane & 15) : (lane & 31));
            if constexpr (UNITS) qrow = slot_q[chunk * DPH_UNIT_SLOTS + qrow];       // -1 = empty column
            if constexpr (CFM) qrow = grp * DPH_QROWS + wave * DPH_QGROUP + (lane & 31);   // both groups are the two k halves of the same 32 rows
            my_qrow[g][hq] = qrow;
            int t = (int)0x80000000;
            if constexpr (CFM) {
                // tau[row] = order-preserving key of the row's score estimate (0xFFFFFFFF: none): thi holds the float itself
                const unsigned key = qrow < n_q ? (unsigned)tau[qrow] : 0xFFFFFFFFu;
                t = key == 0xFFFFFFFFu ? 0x7f800000 : (int)((key & 0x80000000u) ? (key & 0x7fffffffu) : ~key);
            } else if (qrow >= n_q || qrow < 0) {
                t = 0x7fffffff;
            } else if (tau) {
                const int tq = tau[qrow];
                if (tq != (int)0x80000000) {
                    const long long d = ((long long)tq - (long long)lmax_q[qrow]) >> 7;      // arithmetic shift = floor
                    t = d < -2147483647ll ? (int)0x80000000 : (d > 2147483646ll ? 0x7ffffffe : (int)d);
                }
            }
            thi[g][hq] = t;
            if constexpr (AUX) {
                qa[g][hq] = v4i{0, 0, 0, 0};
                if constexpr (X16) {           // (a row has DPH_AUX_SLOTS = 32 slots: the lanes of k-chunks 2 and 3 hold zeros)
                    if (qrow < n_q && qrow >= 0 && lane < 32) qa[g][hq] = *(const v4i*)(qaux + (int64_t)qrow * DPH_AUX_SLOTS + (lane >> 4) * 16);
                } else {
                    if (qrow < n_q && qrow >= 0) qa[g][hq] = *(const v4i*)(qaux + (int64_t)qrow * DPH_AUX_SLOTS + (lane >> 5) * 16);
                }
            }
        }
        // The compiler does not see the hand-written s_waitcnt of the prologue: unless these loads are complete IN ITS OWN
        // BOOKKEEPING before the streaming loop starts, it waits for them at their first uses inside the loop -- vmcnt(23) ..
        // vmcnt(0) in the loop header, i.e. a full drain of the staged tiles in flight once per NSET tiles (measured: scan
        // 20.2 -> 21.5 ms at 170 M rows when a change elsewhere re-ordered these loads behind the bound loads).  Naming every
        // fragment register as an asm operand here makes it wait now, once.
#pragma unroll
        for (int ks = 0; ks < DPH_KSTEPS; ++ks) {
            asm volatile("" : "+v"(qh[0][ks]));
            if constexpr (QB == 2) asm volatile("" : "+a"(qh[QB - 1][ks]));
        }
#pragma unroll
        for (int g = 0; g < QB; ++g)
#pragma unroll
            for (int hq = 0; hq < NH; ++hq) asm volatile("" : "+v"(thi[g][hq]), "+v"(my_qrow[g][hq]));
        if constexpr (AUX) {
#pragma unroll
            for (int hq = 0; hq < NH; ++hq) {
                asm volatile("" : "+v"(qa[0][hq]));
                if constexpr (QB == 2) asm volatile("" : "+a"(qa[QB - 1][hq]));
            }
        }
    };
    if constexpr (!UNITS) load_queries(0);

    // The tiles a workgroup multiplies come from a WORK QUEUE of contiguous segments: a workgroup pops a segment
    // (thread 0, one atomic), streams its tiles front to back and pops the next -- every CU walks its own window of the
    // shard sequentially and the queue balances the tail (measured against round 1's round-robin deal of single tiles:
    // 20.7 vs 21.1 ms per 170 M-row scan at 128 query rows, 30.9 vs 32.2 ms at 256).  MODE 0 / 1: segments of the visited
    // tiles, lengths by guided self-scheduling (below); MODE 2: unit records (a segment of one inverted list + the chunk
    // of query rows probing it).
    for (int trip = 0;; ++trip) {
    int nt = 0;
    int64_t unit_first = 0;               // first visited tile of the segment
    {
        // the barrier also says every wave is done with the LDS tiles of the previous segment.  Double-buffered slot:
        // thread 0 can be at most one trip ahead of the slowest reader.
        int* const slot = s_unit + 8 * (trip & 1);
        if (tid == 0) slot[0] = queue_result();
        __syncthreads();
        const int u = __builtin_amdgcn_readfirstlane(slot[0]);
        if constexpr (UNITS) {
            if (u >= unit_counts[0]) break;         // the count of the table this launch walks
            const int4 rec = unit_recs[u];
            // the tiles of segment [rec.y, rec.z) whose index inside the list (first tile rec.x) is a multiple of tile_stride
            const int rel = rec.y - rec.x;
            const int first = rec.x + (rel + tile_stride - 1) / tile_stride * tile_stride;
            nt = first < rec.z ? (rec.z - first + tile_stride - 1) / tile_stride : 0;
            if (nt <= 0) { queue_pop(); continue; }
            unit_first = first;
            load_queries(rec.w);
        } else {
            // guided self-scheduling (dph_guided_segment): lengths halve from n_tiles/(4*grid) (127 MiB of a 170 M-row
            // shard) down to seg_tiles -- ~20 pops per workgroup instead of 80 equal ones, and a tail of seg_tiles tiles
            int64_t len;
            if constexpr (TEAMS) {
                // one run of seg_tiles tiles of lists per team, the same for all its members
                unit_first = trip == 0 ? 2 * (int64_t)team * seg_tiles : n_tiles;
                len = 2 * (int64_t)seg_tiles;
                if (unit_first < n_tiles && n_tiles - unit_first < len) len = n_tiles - unit_first;
            } else if constexpr (CFM) {
                // (the queue deals whole tiles of lists = pairs of pieces)
                unit_first = 2 * dph_guided_segment(u, n_tiles / 2, (int)gridDim.x, seg_tiles, &len);
                len *= 2;
            } else {
                unit_first = dph_guided_segment(u, n_tiles, (int)gridDim.x, seg_tiles, &len);
            }
            if (unit_first >= n_tiles) break;
            nt = (int)len;
        }
    }
    // launch-tile j of the segment (past its end: its last tile again -- the feed never stops loading, which keeps every
    // vmcnt of the loop a compile-time constant; those phantom tiles are never tested)
    // MODE 0 with skip_m != 0 (the full scan behind a FUSED finest ladder level of stride S, dph_api.hip run_pass): the tiles
    // whose index is a multiple of S were scanned -- and their pairs refined into the buckets -- by that level already, so the
    // full scan visits only the others: visited index v -> tile v + v / (S - 1) + 1, the division as a multiply by
    // skip_m = ceil(2^29 / (S - 1)) (exact for every v the host admits).
    auto tile_of = [&](int j) {
        int64_t v = unit_first + (int64_t)(j < nt ? j : nt - 1) * (int64_t)(UNITS ? tile_stride : 1);
        if constexpr (MODE == 0) { if (skip_m) v += (int64_t)(((uint64_t)v * skip_m) >> 29) + 1; }
        return v;
    };

    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    // ---- LDS write addresses of this lane's six staged 16-byte units.  Piece p = 4i + wave covers units
    //      u = 64p + lane of the tile; unit u is chunk c = u % 48 of row u / 48 and is stored at chunk
    //      c' = (c & 0x30) | ((c ^ row) & 15) of that row: the XOR swizzle that makes the fragment reads conflict-free.
    //      [0] = buffers 0/1, [1] = buffers 2/3 (the odd buffer of a pair is the +24576 immediate)
    unsigned waddr[2][6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const unsigned u = (unsigned)(4 * i + wave) * 64u + (unsigned)lane, row = u / 48, c = u % 48;
        waddr[0][i] = lds_base + (row * 48 + ((c & 0x30u) | ((c ^ row) & 15u))) * 16;
        waddr[1][i] = waddr[0][i] + 2 * DPH_TILE_BYTES;
    }
    unsigned voff[6];                       // byte offset of this lane's 16 bytes of piece i from the wave's piece 0
#pragma unroll
    for (int i = 0; i < 6; ++i) voff[i] = (unsigned)lane * 16u + (unsigned)i * 4096u;
    // ---- fragment read addresses: lane reads row (lane&31), chunk 2ks + (lane>>5)
    unsigned faddr[2][8];
    if constexpr (X16) {
        // k-step p of a tile step: row half p & 1, slab p >> 1; inside the 256-byte third p >> 3 (the immediate offset below) the lane's
        // 16 bytes are chunk 4 ((p >> 1) & 3) + (l >> 4) of row 16 (p & 1) + (l & 15), swizzled like every chunk (^ row & 15).  The four
        // 16-lane groups a ds_read_b128 is served in hit 64 distinct banks (rows r and chunks c, c + 1 of one aligned group of four:
        // c ^ r and (c ^ 1) ^ r' over disjoint row sets).
        const unsigned r15 = lane & 15, q4 = (unsigned)(lane >> 4);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            faddr[0][m] = lds_base + (16u * (m & 1) + r15) * DPH_DIM + ((((4u * (m >> 1)) + q4) ^ r15) << 4);
            faddr[1][m] = faddr[0][m] + 2 * DPH_TILE_BYTES;
        }
    } else {
        const unsigned row = lane & 31, h = (unsigned)(lane >> 5) ^ (row & 15u);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            faddr[0][m] = lds_base + row * DPH_DIM + (((2u * m) ^ h) << 4);
            faddr[1][m] = faddr[0][m] + 2 * DPH_TILE_BYTES;
        }
    }

    const int64_t tile_bytes = (int64_t)tile_stride * (int64_t)DPH_TILE_BYTES;
    // this wave's first piece of launch-tile j (wave-uniform; piece i is 4 KiB further: voff[i])
    auto piece_base = [&](int j) {
        int64_t t = tile_of(j);
        if constexpr (UNITS) return db + t * (int64_t)DPH_TILE_BYTES + (int64_t)wave * 1024;
        return db + t * tile_bytes + (int64_t)wave * 1024;
    };

    // aux rows of launch-tile j (wave-uniform base; the lane's 16 bytes: aux_voff)
    const unsigned aux_voff = X16 ? (unsigned)(lane & 15) * (unsigned)aux_stride + (unsigned)(lane >> 4) * 16u
                                  : (unsigned)(lane & 31) * (unsigned)aux_stride + (unsigned)(lane >> 5) * 16u;
    const unsigned aux_voff_lo = aux_voff + 16u * (unsigned)aux_stride;        // X16: the lower row half (rows 16 .. 31 of the tile)
    auto aux_base = [&](int j) {
        int64_t t = tile_of(j);
        if constexpr (!UNITS) t *= (int64_t)tile_stride;
        return aux + t * (int64_t)(DPH_TILE_ROWS * aux_stride);
    };
    auto aux_loads = [&](auto sc, int j) __attribute__((always_inline)) {       // the aux operand(s) of launch-tile j into ring slot S
        constexpr int S_ = decltype(sc)::value;
        aux_load<NSET, S_, 0>(aux_voff, aux_base(j));
        if constexpr (X16) aux_load<NSET, S_, 1>(aux_voff_lo, aux_base(j));
    };

    // Everything that depends on the hand-over schedule -- prologue, counted waits, the k-steps in which this wave writes and
    // re-loads its pieces -- is compiled once per wave number W (SCHED != 0: `switch (wave)` below; four copies of the
    // streaming loop, ~6 KiB each, well inside the instruction cache) so that every wait stays a compile-time constant.
    auto stream = [&](auto wc) __attribute__((always_inline)) {
    constexpr int W = decltype(wc)::value;
    // ---- prologue: tiles 0 and 1 into LDS, tiles 2 .. NSET+1 into flight (tile t travels in staging set t % NSET),
    //      pre-load the first fragments of tile 0.  SCHED != 0: exactly the state the steady loop has at the start of a
    //      tile step -- of tile 1 only the pieces this wave hands over before position 24 are in LDS (the others wait in
    //      their staging registers for the first tile step), and only the re-loads issued before position 24 are in flight.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the query / bound loads share the vm counter with the feed
    {
        const int8_t* b0 = piece_base(0);
        const int8_t* b1 = piece_base(1);
        static_for<0, 6>([&](auto ic) { constexpr int i = decltype(ic)::value; stage_load<NSET, 0, i, NT_LOADS>(voff[i], b0); });
        static_for<0, 6>([&](auto ic) { constexpr int i = decltype(ic)::value; stage_load<NSET, 1, i, NT_LOADS>(voff[i], b1); });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        static_for<0, 6>([&](auto ic) { constexpr int i = decltype(ic)::value; stage_write<NSET, 0, i, 0>(waddr[0][i]); });
        static_for<0, 6>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (hand_pos(SCHED, W, i) < 24) stage_write<NSET, 1, i, DPH_TILE_BYTES>(waddr[0][i]);
        });
        asm volatile("s_nop 1" ::: "memory");           // the stores have read their data registers
        static_for<0, NSET>([&](auto sc) {
            constexpr int t = 2 + decltype(sc)::value;              // tile t -> set t % NSET
            const int8_t* bt = piece_base(t);
            static_for<0, 6>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                // tile NSET+1 re-uses the registers of tile 1: only the pieces whose re-load comes before position 24
                if constexpr (t < NSET + 1 || hand_pos(SCHED, W, i) + 1 < 24) stage_load<NSET, t % NSET, i, NT_LOADS>(voff[i], bt);
            });
        });
        if constexpr (AUX) {
            // the aux rows of tiles 0 .. 2 behind everything else, then a drain: whatever the first tile steps await has landed,
            // and from tile step 3 on every counted wait sees the steady state (a few microseconds per queue segment)
            aux_loads(std::integral_constant<int, 0>{}, 0);
            aux_loads(std::integral_constant<int, 1>{}, 1);
            aux_loads(std::integral_constant<int, 2>{}, 2);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    queue_pop();
    constexpr int PF = DPH_PF;
    static_assert(PF == 3, "ring indexing assumes a 3-deep prefetch");
    constexpr int RING = 4;
    v4i bq[RING];
    // the first fragment reads stand where k-steps 21 .. 23 of a previous tile step would have issued them; the staging
    // writes a schedule places there are issued too (into a scratch kilobyte) so that the counted lgkmcnt waits of the first
    // k-steps see the queue they assume
    static_for<0, PF>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (SCHED != 0 && writes_at(SCHED, W, DPH_KSTEPS - PF + i) != 0)
            asm volatile("ds_write_b128 %0, %1" ::"v"(lds_base + 4 * DPH_TILE_BYTES + 64 + (unsigned)lane * 16u), "v"(qh[0][0]) : "memory");
        ds_read16<0>(bq[i], faddr[0][i]);
    });

    acc_t accA[QB], accB[QB];
#pragma unroll
    for (int g = 0; g < QB; ++g) {
        if constexpr (X16) {
#pragma unroll
            for (int b = 0; b < 4; ++b) accA[g].s[b] = v4i{0, 0, 0, 0};
        } else {
            accA[g] = v16i{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        }
        accB[g] = accA[g];
    }

    // step `it` = S mod NSET (S a compile-time constant of the unrolled loop) multiplies tile it into `cur` and tests
    // the scores of tile it-1 held in `prev`; tiles >= nt are phantoms (stale LDS bytes, results never tested) that
    // only flush the pipeline.  Buffers: tile it in S%4, it+1 in (S+1)%4, the hand-over writes it+2 into (S+2)%4 from
    // staging set (S+2)%NSET and re-loads that set with tile it+2+NSET (SCHED != 0: in the k-steps before the barrier the
    // pieces of tile it+1 that the schedule placed at positions >= 24 of the previous tile step).
    auto tile_step = [&](auto sc, acc_t (&cur)[QB], const acc_t (&prev)[QB], const int it) __attribute__((always_inline)) {
        constexpr int S = decltype(sc)::value;
        constexpr int SET = (S + 2) % NSET;
        constexpr int PARITY = S & 1;
        constexpr int BC = S % NBUF, BN = (S + 1) % NBUF, BW = (S + 2) % NBUF;
        if constexpr (AUX) {
            // aux rows of tile `it` (loaded at the top of step it-3; younger: 3 x 6 staged pieces, the aux rows of it+1 and it+2, on
            // list-major shards three probe masks), then the load for tile it+3 into the operand step it-1 consumed
            wait_vmcnt<3 * 6 + 2 * AUXL + (IVF ? 3 : 0)>();
            aux_loads(std::integral_constant<int, (S + 3) % 4>{}, it + 3);
        }
        if constexpr (IVF) {
            const int64_t t = tile_of(it);
            mask_load<NSET, QB, PARITY>(0u * (unsigned)lane, tilemask + (t * tile_stride) * 8 + wave * QB);
        }
        const int8_t* const b4 = piece_base(it + 2 + NSET);
        const int8_t* b3 = b4;
        if constexpr (SCHED != 0) b3 = piece_base(it + 1 + NSET);
        int mx[QB][NH];
#pragma unroll
        for (int g = 0; g < QB; ++g)
#pragma unroll
            for (int hq = 0; hq < NH; ++hq) mx[g][hq] = (int)0x80000000;
        constexpr int HALF = S & 1;             // MODE 3: which k half of the tile of lists this piece is
        float mxf = -__builtin_inff();
        if constexpr (AUX && X16) {
            static_for<0, 4>([&](auto bc) {
                constexpr int b = decltype(bc)::value;              // block [row half b >> 1][query half b & 1]
                mfma_aux16<NSET, S % 4, (b >> 1), false>(cur[0].s[b], qa[0][b & 1]);
                if constexpr (QB == 2) mfma_aux16<NSET, S % 4, (b >> 1), true>(cur[QB - 1].s[b], qa[QB - 1][b & 1]);
            });
        } else if constexpr (AUX) {
            mfma_aux<NSET, S % 4, false>(cur[0], qa[0][0]);
            if constexpr (QB == 2) mfma_aux<NSET, S % 4, true>(cur[QB - 1], qa[QB - 1][0]);
        }
        static_for<0, DPH_KSTEPS>([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            if constexpr (ks == DPH_KSYNC) {
                // hand-over.  Staging set SET holds tile it+2 (loaded NSET hand-overs ago); the NSET-1 younger sets
                // (and, on list-major shards, at least one mask dword) stay in flight across the wait.
                if constexpr (!(DPH_SCAN_DIAG & 2) && SCHED == 0) wait_vmcnt<6 * (NSET - 1) + (IVF ? 1 : 0) + 4 * AUXL>();
                if constexpr (!(DPH_SCAN_DIAG & 1)) __builtin_amdgcn_s_barrier();      // tile it-1 is fully consumed (its buffer is free), tile it+1 is published
                asm volatile("" ::: "memory");
            }
            if constexpr (SCHED == 0) {
                // ... spread over the next k-steps, one piece each, so the matrix pipe is never left without work: piece i
                // goes to LDS at k-step KSYNC+i and its registers are re-loaded with tile it+2+NSET one k-step later.
                if constexpr (ks >= DPH_KSYNC && ks < DPH_KSYNC + 6 && !(DPH_SCAN_DIAG & 8))
                    stage_write<NSET, SET, ks - DPH_KSYNC, (BW & 1) * DPH_TILE_BYTES>(waddr[BW >> 1][ks - DPH_KSYNC]);
                if constexpr (ks > DPH_KSYNC && ks <= DPH_KSYNC + 6 && !(DPH_SCAN_DIAG & 4)) stage_load<NSET, SET, ks - DPH_KSYNC - 1, NT_LOADS>(voff[ks - DPH_KSYNC - 1], b4);
            } else {
                // this wave's pieces by its schedule: d = 0 the tile two ahead (set SET, buffer BW), d = 1 the next tile (set
                // (S+1)%NSET, buffer BN).  One counted wait per piece (younger_loads): only THAT piece's load has to have landed.
                constexpr int w0 = wr_piece(SCHED, W, ks, 0), w1 = wr_piece(SCHED, W, ks, 1);
                constexpr int l0 = ld_piece(SCHED, W, ks, 0), l1 = ld_piece(SCHED, W, ks, 1);
                static_assert(w0 < 0 || w1 < 0, "one staging write per k-step");
                if constexpr (w0 >= 0) { wait_vmcnt<younger_loads(SCHED, W, w0 >= 0 ? w0 : 0, NSET) + AUXL * aux_younger(SCHED, W, w0 >= 0 ? w0 : 0)>(); stage_write<NSET, SET, w0, (BW & 1) * DPH_TILE_BYTES>(waddr[BW >> 1][w0]); }
                if constexpr (w1 >= 0) { wait_vmcnt<younger_loads(SCHED, W, w1 >= 0 ? w1 : 0, NSET) + AUXL * aux_younger(SCHED, W, w1 >= 0 ? w1 : 0)>(); stage_write<NSET, (S + 1) % NSET, w1, (BN & 1) * DPH_TILE_BYTES>(waddr[BN >> 1][w1]); }
                if constexpr (l0 >= 0) stage_load<NSET, SET, l0, NT_LOADS>(voff[l0], b4);
                if constexpr (l1 >= 0) stage_load<NSET, (S + 1) % NSET, l1, NT_LOADS>(voff[l1], b3);
            }
            constexpr int p = ks + PF;
            if constexpr (!(DPH_SCAN_DIAG & 16)) {
                if constexpr (p < DPH_KSTEPS) ds_read16<(BC & 1) * DPH_TILE_BYTES + (p >> 3) * 256>(bq[p & (RING - 1)], faddr[BC >> 1][p & 7]);
                else ds_read16<(BN & 1) * DPH_TILE_BYTES>(bq[p & (RING - 1)], faddr[BN >> 1][p - DPH_KSTEPS]);
            }
            // LDS operations younger than the fragment read awaited here: the PF reads issued since, plus the staging
            // writes of k-steps ks-2 .. ks
            constexpr int staged = SCHED == 0 ? staged_at(ks - 2) + staged_at(ks - 1) + staged_at(ks)
                                              : writes_at(SCHED, W, ks - 2) + writes_at(SCHED, W, ks - 1) + writes_at(SCHED, W, ks);
            constexpr int younger = (DPH_SCAN_DIAG & 16) ? 0 : PF + ((DPH_SCAN_DIAG & 8) ? 0 : staged);
            if constexpr (!(DPH_SCAN_DIAG & 16)) wait_lgkm<younger>(bq[ks & (RING - 1)]);
            if constexpr (CFM) {
                // one accumulator set per tile of lists: half 0 clears it, half 1 adds to it; the previous tile's scores are compared
                // while half 0 of this one is multiplied
                if constexpr (HALF == 0) mfma_bf16<ks == 0, false>(cur[0], bq[ks & (RING - 1)], qh[0][ks]);
                else mfma_bf16<false, true>(cur[0], bq[ks & (RING - 1)], qh[QB - 1][ks]);
                if constexpr (HALF == 0 && ks >= 4 && ks < 20) {
                    mxf = fmaxf(mxf, __int_as_float(prev[0][ks - 4]));       // (by value: __builtin_bit_cast of a vector ELEMENT reads element 0)
                    asm volatile("" : "+v"(mxf));
                }
            } else if constexpr (X16) {
                // row half ks & 1 of slab ks >> 1 against the left and the right query half: blocks [2 (ks & 1)] and [2 (ks & 1) + 1]
                constexpr int RH = ks & 1, SL = ks >> 1;
                constexpr bool FIRST = ks < 2 && !AUX;      // (aux shards: the aux MFMAs above opened the blocks)
                mfma_i8x16<FIRST, false>(cur[0].s[2 * RH], bq[ks & (RING - 1)], qh[0][2 * SL]);
                mfma_i8x16<FIRST, false>(cur[0].s[2 * RH + 1], bq[ks & (RING - 1)], qh[0][2 * SL + 1]);
                if constexpr (QB == 2) {
                    mfma_i8x16<FIRST, true>(cur[QB - 1].s[2 * RH], bq[ks & (RING - 1)], qh[QB - 1][2 * SL]);
                    mfma_i8x16<FIRST, true>(cur[QB - 1].s[2 * RH + 1], bq[ks & (RING - 1)], qh[QB - 1][2 * SL + 1]);
                }
            } else {
            mfma_i8<ks == 0 && !AUX, false>(cur[0], bq[ks & (RING - 1)], qh[0][ks]);
            if constexpr (QB == 2) mfma_i8<ks == 0 && !AUX, true>(cur[QB - 1], bq[ks & (RING - 1)], qh[QB - 1][ks]);
            }
            if constexpr (!CFM && ks >= 4 && ks < 20 && !(DPH_SCAN_DIAG & 32)) {
#pragma unroll
                for (int g = 0; g < QB; ++g) {
                    // (X16: score ks - 4 belongs to block (ks - 4) >> 2, whose query half is its low bit)
                    constexpr int HQ = X16 ? (((ks - 4) >> 2) & 1) : 0;
                    mx[g][HQ] = max(mx[g][HQ], acc_get(prev[g], ks - 4));
                    asm volatile("" : "+v"(mx[g][HQ]));   // keep the running max a chain (a re-associated tree holds 32 VGPRs)
                }
            }
            // pin the software pipeline: one fragment read PF steps ahead, QB MFMAs and one slice of the previous
            // tile's threshold test per k-step
            __builtin_amdgcn_sched_barrier(0);
        });

        if constexpr (CFM) {
            if constexpr (HALF == 0) {
                const float thr = __int_as_float(thi[0][0]);
                if (it >= 2 && it <= nt && __builtin_amdgcn_ballot_w64(mxf >= thr) != 0ull) {
                    // ---------------- emit path: some lane holds a list of tile it/2 - 1 whose score reaches its row's estimate
                    ++triggers;
                    const unsigned rowbase = (unsigned)(tile_of(it - 2) / 2) * DPH_TILE_ROWS + 4u * (unsigned)(lane >> 5);
                    float t = thr;
                    asm volatile("" : "+v"(t));             // the compares below belong to this branch: do not hoist them
                    unsigned bits = 0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) bits |= (__int_as_float(prev[0][r]) >= t) ? (1u << r) : 0u;
                    const unsigned qrow = (unsigned)my_qrow[0][0];
                    while (__builtin_amdgcn_ballot_w64(bits != 0u) != 0ull) {
                        unsigned payload = 0, key = 0;
                        bool emit = false;
                        if (bits != 0u) {
                            const int r = __builtin_ctz(bits);
                            bits &= bits - 1u;
                            const unsigned row = rowbase + (unsigned)((r & 3) + 8 * (r >> 2));
                            emit = row < n_rows_u;             // lists past the end of the quantizer are zero padding
                            payload = row | (qrow << 20);
                            // (the accumulator is picked with a chain of selects: a runtime index would put the set in scratch)
                            unsigned u = 0;
#pragma unroll
                            for (int rr = 0; rr < 16; ++rr) u = rr == r ? (unsigned)prev[0][rr] : u;
                            key = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
                        }
                        emit_pairs(emit, payload, key);
                    }
                }
            }
            return;
        }
        bool probed[QB][NH];
        bool any = false;
#pragma unroll
        for (int g = 0; g < QB; ++g) {
            unsigned m = 0xFFFFFFFFu;
            if constexpr (IVF) {
                if (g == 0) m = mask_read<NSET, 1 - PARITY, 0, AUXL>();
                else m = mask_read<NSET, 1 - PARITY, QB - 1, AUXL>();
            }
#pragma unroll
            for (int hq = 0; hq < NH; ++hq) {
                probed[g][hq] = ((m >> (X16 ? 16 * hq + (lane & 15) : (lane & 31))) & 1u) != 0u;
                any = any || (probed[g][hq] && mx[g][hq] > thi[g][hq]);
            }
        }
        if (it >= 1 && it <= nt && __builtin_amdgcn_ballot_w64(any) != 0ull) {
            // ---------------- emit path: some lane holds a row of tile it-1 whose high digit passes its bound
            ++triggers;
            const unsigned rowbase = (unsigned)(tile_of(it - 1) * (UNITS ? 1 : tile_stride) * DPH_TILE_ROWS) + 4u * (unsigned)(X16 ? lane >> 4 : lane >> 5);
            // score register r of a lane -> its row inside the tile (without the lane's 4 (l >> 4 | l >> 5)) and its query half
            auto row_of = [](int r) { return X16 ? (unsigned)((r & 3) + 16 * (r >> 3)) : (unsigned)((r & 3) + 8 * (r >> 2)); };
#pragma unroll
            for (int g = 0; g < QB; ++g) {
                int t[NH];
#pragma unroll
                for (int hq = 0; hq < NH; ++hq) {
                    t[hq] = thi[g][hq];
                    asm volatile("" : "+v"(t[hq]));     // the compares below belong to this branch: do not hoist them
                }
                unsigned bits = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) bits |= (acc_get(prev[g], r) > t[X16 ? (r >> 2) & 1 : 0]) ? (1u << r) : 0u;
                if constexpr (X16) {
                    if (!probed[g][0]) bits &= 0xF0F0u;            // (blocks 0 and 2 are the left query half)
                    if (!probed[g][NH - 1]) bits &= 0x0F0Fu;
                } else {
                    if (!probed[g][0]) bits = 0;
                }
                if constexpr (UNITS) bits &= rowmask;
                if constexpr (MODE != 0) {
                    // list-major shards pad every list to whole tiles with all-zero rows: H == 0 for every query row,
                    // which passes any negative bound -- tens of thousands of dead pairs per query row on a 4096-list
                    // shard, enough to overflow the pair regions.  A candidate with H == 0 is therefore looked up
                    // (row_ids < 0 = padding) before it is emitted; real rows with H == 0 are rare.
                    unsigned zero = 0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) zero |= (acc_get(prev[g], r) == 0) ? (1u << r) : 0u;
                    zero &= bits;
                    while (zero != 0u) {
                        const int r = __builtin_ctz(zero);
                        zero &= zero - 1u;
                        const unsigned row = rowbase + row_of(r);
                        if (row < n_rows_u && row_ids[row] < 0) bits &= ~(1u << r);
                    }
                }
                // (two named values, pinned: left as a select between two ELEMENTS of my_qrow the compiler indexes the array with the
                // condition -- and an indexed array lives in scratch)
                int q_left = my_qrow[g][0], q_right = my_qrow[g][NH - 1];
                asm volatile("" : "+v"(q_left), "+v"(q_right));
                while (__builtin_amdgcn_ballot_w64(bits != 0u) != 0ull) {
                    unsigned row = 0, qrow = 0;
                    bool emit = false;
                    if (bits != 0u) {
                        const int r = __builtin_ctz(bits);
                        bits &= bits - 1u;
                        row = rowbase + row_of(r);
                        qrow = (unsigned)((X16 && ((r >> 2) & 1)) ? q_right : q_left);
                        emit = row < n_rows_u;                 // rows past the end of the shard are zero padding
                    }
                    emit_pairs(emit, row, qrow);
                }
            }
        }
    };

    for (int it = 0; it <= nt; it += NSET) {
        static_for<0, NSET / 2>([&](auto hc) {
            constexpr int s = 2 * decltype(hc)::value;
            if constexpr (CFM) {
                // both halves of a tile of lists into the same accumulators; the sets alternate per tile
                if constexpr ((s / 2) % 2 == 0) {
                    tile_step(std::integral_constant<int, s>{}, accA, accB, it + s);
                    tile_step(std::integral_constant<int, s + 1>{}, accA, accB, it + s + 1);
                } else {
                    tile_step(std::integral_constant<int, s>{}, accB, accA, it + s);
                    tile_step(std::integral_constant<int, s + 1>{}, accB, accA, it + s + 1);
                }
            } else {
                tile_step(std::integral_constant<int, s>{}, accA, accB, it + s);
                tile_step(std::integral_constant<int, s + 1>{}, accB, accA, it + s + 1);
            }
        });
    }
    };      // stream
    if constexpr (SCHED == 0) {
        stream(std::integral_constant<int, 0>{});
    } else {
        switch (wave) {
            case 0: stream(std::integral_constant<int, 0>{}); break;
            case 1: stream(std::integral_constant<int, 1>{}); break;
            case 2: stream(std::integral_constant<int, 2>{}); break;
            default: stream(std::integral_constant<int, 3>{}); break;
        }
    }
    // the phantom loads of the last hand-overs are still in flight here: they land in staging registers the next
    // segment's prologue re-loads anyway (loads return in order) and are awaited there, behind the queue pop
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // nothing of the feed may still be in flight at exit
    if (lane == 0) {
        my_counts[0] = cnt; my_counts[1] = triggers;
        if (chunk != 0xFFFFFFFFu) chunk_fill[chunk] = fill;
    }
}

// ROLE only gives the launches their own name in a profile: 0 = the full scan of a pass (the one bench.py prices), 1 = a
// pre-pass over every `tile_stride`-th tile, 2 = the gated retry scan (a no-op launch when nothing failed).
template <int QB, int NSET, bool IVF, int ROLE, int SCHED = 0, bool AUX = false>
__global__ __launch_bounds__(DPH_SCAN_THREADS, 1) void dph_scan_kernel(
    const int8_t* __restrict__ db, int64_t n_rows, int64_t n_tiles, int tile_stride, const int8_t* __restrict__ qfrag,
    int n_q_host, const int* __restrict__ gate, int gate_base, const int* __restrict__ tau, const int* __restrict__ lmax_q,
    const unsigned* __restrict__ tilemask, uint2* __restrict__ pairs, unsigned* __restrict__ wave_counts,
    int* __restrict__ queue_head, int seg_tiles, const int64_t* __restrict__ row_ids, unsigned* __restrict__ chunk_fill,
    unsigned* __restrict__ overflow, unsigned skip_m, const int8_t* __restrict__ aux, int aux_stride, const int8_t* __restrict__ qaux) {
    dph_scan_body<QB, NSET, IVF ? 1 : 0, ROLE, SCHED, AUX>(db, n_rows, n_tiles, tile_stride, qfrag, n_q_host, gate, gate_base, tau, lmax_q,
                                                    tilemask, pairs, wave_counts, nullptr, nullptr, queue_head, nullptr, 0xFFFFu,
                                                    seg_tiles, row_ids, (unsigned*)queue_head + 1, chunk_fill, overflow, skip_m, aux,
                                                    aux_stride, qaux);
}
// the unit scan of a list-major shard (MODE 2 above); ROLE 0 = full scan of the pass, 1 = a ladder level
template <int ROLE, bool AUX = false>
__global__ __launch_bounds__(DPH_SCAN_THREADS, 1) void dph_scan_units_kernel(
    const int8_t* __restrict__ db, int64_t n_rows, int tile_stride, unsigned rowmask, const int8_t* __restrict__ unit_frags,
    int n_q, const int* __restrict__ tau, const int* __restrict__ lmax_q, const int4* __restrict__ unit_recs,
    const int* __restrict__ unit_counts, int* __restrict__ unit_next, const int* __restrict__ slot_q,
    uint2* __restrict__ pairs, unsigned* __restrict__ wave_counts, const int64_t* __restrict__ row_ids,
    unsigned* __restrict__ pool_head, unsigned* __restrict__ chunk_fill, unsigned* __restrict__ overflow,
    const int8_t* __restrict__ aux, int aux_stride, const int8_t* __restrict__ qaux) {
    dph_scan_body<1, 4, 2, ROLE, 0, AUX>(db, n_rows, 0, tile_stride, unit_frags, n_q, nullptr, 0, tau, lmax_q, nullptr, pairs,
                                         wave_counts, unit_recs, unit_counts, unit_next, slot_q, rowmask, 0, row_ids, pool_head,
                                         chunk_fill, overflow, 0u, aux, aux_stride, qaux);
}

// MODE 3 above: the coarse quantizer's filter over the bf16 centroid image (n_pieces = 2 per tile of 32 lists), <= 128 query rows
__global__ __launch_bounds__(DPH_SCAN_THREADS, 1) void dph_coarse_scan_kernel(
    const int8_t* __restrict__ img, int64_t n_lists, int64_t n_pieces, const int8_t* __restrict__ qfrag, int n_q,
    const int* __restrict__ est_keys, uint2* __restrict__ pairs, unsigned* __restrict__ wave_counts, int* __restrict__ queue_head,
    int seg_tiles, unsigned* __restrict__ chunk_fill, unsigned* __restrict__ overflow) {
    dph_scan_body<2, 4, 3, 0, 0, false>(img, n_lists, n_pieces, 1, qfrag, n_q, nullptr, 0, est_keys, nullptr, nullptr, pairs, wave_counts,
                                        nullptr, nullptr, queue_head, nullptr, 0xFFFFu, seg_tiles, nullptr, (unsigned*)queue_head + 1,
                                        chunk_fill, overflow, 0u, nullptr, 0, nullptr);
}
// MODE 4 above: the same filter for a pass of n_groups x 128 query rows in one launch (teams of n_groups workgroups per run of tiles)
__global__ __launch_bounds__(DPH_SCAN_THREADS, 1) void dph_coarse_scan_teams_kernel(
    const int8_t* __restrict__ img, int64_t n_lists, int64_t n_pieces, const int8_t* __restrict__ qfrag, int n_q, int n_groups,
    const int* __restrict__ est_keys, uint2* __restrict__ pairs, unsigned* __restrict__ wave_counts, int* __restrict__ queue_head,
    int seg_tiles, unsigned* __restrict__ chunk_fill, unsigned* __restrict__ overflow) {
    dph_scan_body<2, 4, 4, 0, 0, false>(img, n_lists, n_pieces, 1, qfrag, n_q, nullptr, n_groups, est_keys, nullptr, nullptr, pairs, wave_counts,
                                        nullptr, nullptr, queue_head, nullptr, 0xFFFFu, seg_tiles, nullptr, (unsigned*)queue_head + 1,
                                        chunk_fill, overflow, 0u, nullptr, 0, nullptr);
}
// n_groups in {2, 4, 8, 16, 32} and grid a multiple of 8 * n_groups; qfrag: the fragments of all groups, group after group
void dph_launch_coarse_scan_teams(const void* img, int64_t n_lists, const void* qfrag, int n_q, int n_groups, const unsigned* est_keys, uint2* pairs,
                                  unsigned* chunk_fill, unsigned* wave_counts, int* counters, int grid, hipStream_t st) {
    const size_t lds = DPH_SCAN_LDS_BYTES;
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        const hipError_t e = hipFuncSetAttribute((const void*)dph_coarse_scan_teams_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) fprintf(stderr, "libdph: hipFuncSetAttribute(dph_coarse_scan_teams_kernel, %zu B of LDS): %s\n", lds, hipGetErrorString(e));
        if (dev >= 0 && dev < 64) attr_set[dev] = e == hipSuccess;
    }
    const int64_t n_tiles = (n_lists + DPH_TILE_ROWS - 1) / DPH_TILE_ROWS;
    const int n_teams = grid / n_groups;
    const int seg = (int)std::max<int64_t>(1, (n_tiles + n_teams - 1) / n_teams);
    (void)hipMemsetAsync(counters, 0, 16, st);
    hipLaunchKernelGGL(dph_coarse_scan_teams_kernel, dim3(grid), dim3(DPH_SCAN_THREADS), lds, st, (const int8_t*)img, n_lists, 2 * n_tiles,
                       (const int8_t*)qfrag, n_q, n_groups, (const int*)est_keys, pairs, wave_counts, counters, seg, chunk_fill,
                       (unsigned*)counters + 2);
}
// counters: [0] work-queue head, [1] chunks claimed from the pool, [2] overflow flag, [3] spare (cleared here)
void dph_launch_coarse_scan(const void* img, int64_t n_lists, const void* qfrag, int n_q, const unsigned* est_keys, uint2* pairs,
                            unsigned* chunk_fill, unsigned* wave_counts, int* counters, int grid, hipStream_t st) {
    const size_t lds = DPH_SCAN_LDS_BYTES;
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        const hipError_t e = hipFuncSetAttribute((const void*)dph_coarse_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) fprintf(stderr, "libdph: hipFuncSetAttribute(dph_coarse_scan_kernel, %zu B of LDS): %s\n", lds, hipGetErrorString(e));
        if (dev >= 0 && dev < 64) attr_set[dev] = e == hipSuccess;
    }
    const int64_t n_tiles = (n_lists + DPH_TILE_ROWS - 1) / DPH_TILE_ROWS;
    // ONE segment per workgroup (a quarter millisecond of streaming: every further pop is a pipeline refill of ~3 us, and a queue's
    // last segment an imbalance of its own length); long quantizers only, so the segments are thousands of rows
    const int seg = (int)std::max<int64_t>(1, (n_tiles + grid - 1) / grid);
    (void)hipMemsetAsync(counters, 0, 16, st);
    hipLaunchKernelGGL(dph_coarse_scan_kernel, dim3(grid), dim3(DPH_SCAN_THREADS), lds, st, (const int8_t*)img, n_lists, 2 * n_tiles,
                       (const int8_t*)qfrag, n_q, (const int*)est_keys, pairs, wave_counts, counters, seg, chunk_fill,
                       (unsigned*)counters + 2);
}

// [4] work-queue head, chunks claimed from the pair pool (+ padding) | [DPH_PASS_MAX] bucket counts | [DPH_PASS_MAX]
// overflow flags: one allocation (dph_api.hip), cleared by one memset in front of every scan launch -- the scan counts
// and flags, the outlier / refine kernels behind it fill the buckets, threshold / select read all of it.
void dph_clear_pass_counters(const dph_pass& p, hipStream_t st) {
    // an ACCUMULATING scan (the full scan behind a fused ladder level) keeps the buckets and flags of that level: only the
    // work-queue head and the pair pool start over
    (void)hipMemsetAsync(p.queue_head, 0, p.accumulate ? (size_t)4 * 4 : (size_t)(4 + 2 * DPH_PASS_MAX) * 4, st);
}

int dph_scan_grid(int device) {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    return cus > 0 && cus < 256 ? cus : 256;     // one workgroup per CU
}

template <int QB, int NSET, bool IVF, int ROLE, int SCHED = 0, bool AUX = false>
static void launch_scan_t(const dph_pass& p, int64_t n_tiles_visit, int tile_stride, const int* tau, hipStream_t st) {
    const size_t lds = DPH_SCAN_LDS_BYTES;
    static std::atomic<bool> attr_set[64];       // the attribute is per device
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        const hipError_t e = hipFuncSetAttribute((const void*)dph_scan_kernel<QB, NSET, IVF, ROLE, SCHED, AUX>,
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) fprintf(stderr, "libdph: hipFuncSetAttribute(dph_scan_kernel, %zu B of LDS): %s\n", lds, hipGetErrorString(e));
        if (dev >= 0 && dev < 64) attr_set[dev] = e == hipSuccess;
    }
    const int8_t* qf = p.qfrag_hi + (int64_t)(p.q0 / DPH_QGROUP) * DPH_QGROUP_FRAG_BYTES;
    dph_clear_pass_counters(p, st);
    // shortest segment the queue deals: p.seg_tiles at the end of a full scan, down to one tile on the sampled levels (the
    // cold level visits one tile per workgroup)
    const int64_t fair = n_tiles_visit / ((int64_t)p.grid * 4);
    const int seg = (int)std::max<int64_t>(1, std::min<int64_t>(p.seg_tiles, fair));
    hipLaunchKernelGGL((dph_scan_kernel<QB, NSET, IVF, ROLE, SCHED, AUX>), dim3(p.grid), dim3(DPH_SCAN_THREADS), lds, st, p.db,
                       p.n_rows, n_tiles_visit, tile_stride, qf, p.n_q, p.gate, p.gate_base, tau,
                       p.lmax ? p.lmax + p.q0 : nullptr, p.tilemask, p.pairs, p.wave_counts, p.queue_head, seg, p.row_ids,
                       p.chunk_fill, p.overflow, p.skip_m, p.aux, p.aux_lay.stride,
                       p.qaux ? p.qaux + (int64_t)p.q0 * DPH_AUX_SLOTS : nullptr);
}

// nset (staging sets = tiles in flight per wave) is 4 everywhere: 8 sets measured the same on the 128-row kernel
// (21.58 vs 21.48 ms at 170 M rows) and the 256-row kernel has no registers for more
template <bool AUX>
static void launch_scan_a(const dph_pass& p, bool sample, int64_t n_tiles_visit, int tile_stride, const int* tau, hipStream_t st) {
#define DPH_GO(QB, NS, IVF)                                                                         \
    do {                                                                                            \
        if (sample) launch_scan_t<QB, NS, IVF, 1, 0, AUX>(p, n_tiles_visit, tile_stride, tau, st);       \
        else if (p.gate) launch_scan_t<QB, NS, IVF, 2, 0, AUX>(p, n_tiles_visit, tile_stride, tau, st);  \
        else launch_scan_t<QB, NS, IVF, 0, 0, AUX>(p, n_tiles_visit, tile_stride, tau, st);              \
    } while (0)
    if (p.tilemask) {
        if (p.qb == 1) DPH_GO(1, 4, true);
        else DPH_GO(2, 4, true);
    } else if (!sample && !p.gate && p.sched != 0) {
        // the full scan of a pass under a staggered hand-over schedule (tuning key `scan_sched`; dph_scan.hip, hand_pos)
        if (p.qb == 1) { if (p.sched == 1) launch_scan_t<1, 4, false, 0, 1, AUX>(p, n_tiles_visit, tile_stride, tau, st); else launch_scan_t<1, 4, false, 0, 2, AUX>(p, n_tiles_visit, tile_stride, tau, st); }
        else { if (p.sched == 1) launch_scan_t<2, 4, false, 0, 1, AUX>(p, n_tiles_visit, tile_stride, tau, st); else launch_scan_t<2, 4, false, 0, 2, AUX>(p, n_tiles_visit, tile_stride, tau, st); }
    } else if (p.qb == 1) {
        DPH_GO(1, 4, false);
    } else {
        DPH_GO(2, 4, false);
    }
#undef DPH_GO
}
void dph_launch_scan(const dph_pass& p, bool sample, int64_t n_tiles_visit, int tile_stride, const int* tau, int nset,
                     hipStream_t st) {
    (void)nset;
    if (p.unit_recs) { dph_launch_scan_units(p, sample, tile_stride, 0xFFFFu, tau, st); return; }
    // a shard with aux rows (dph_index_finalize: rogue dimensions or heavy-tailed row norms) runs the instantiations with the aux k-step
    if (p.aux) launch_scan_a<true>(p, sample, n_tiles_visit, tile_stride, tau, st);
    else launch_scan_a<false>(p, sample, n_tiles_visit, tile_stride, tau, st);
}

template <bool AUX>
static void launch_scan_units_a(const dph_pass& p, bool sample, int tile_stride, unsigned rowmask, const int* tau, hipStream_t st) {
    const size_t lds = DPH_SCAN_LDS_BYTES;
    static std::atomic<bool> attr_set[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)dph_scan_units_kernel<0, AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)dph_scan_units_kernel<1, AUX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) fprintf(stderr, "libdph: hipFuncSetAttribute(dph_scan_units_kernel, %zu B of LDS): %s\n", lds, hipGetErrorString(e));
        if (dev >= 0 && dev < 64) attr_set[dev] = e == hipSuccess;
    }
    int* next = p.unit_next + p.unit_launch;
    const int* lmax = p.lmax ? p.lmax + p.q0 : nullptr;
    const int8_t* qaux = p.qaux ? p.qaux + (int64_t)p.q0 * DPH_AUX_SLOTS : nullptr;
    dph_clear_pass_counters(p, st);
    unsigned* const pool_head = (unsigned*)p.queue_head + 1;
    // a ladder level walks the table of WHOLE lists (one unit per chunk: a few strided tiles each -- cutting those into
    // segments would only multiply the per-unit start-up), the full scan the table of segments
    if (sample)
        hipLaunchKernelGGL((dph_scan_units_kernel<1, AUX>), dim3(p.grid), dim3(DPH_SCAN_THREADS), lds, st, p.db, p.n_rows, tile_stride,
                           rowmask, p.unit_frags, p.n_q, tau, lmax, p.unit_list_recs, p.unit_counts + 0, next, p.slot_q, p.pairs,
                           p.wave_counts, p.row_ids, pool_head, p.chunk_fill, p.overflow, p.aux, p.aux_lay.stride, qaux);
    else
        hipLaunchKernelGGL((dph_scan_units_kernel<0, AUX>), dim3(p.grid), dim3(DPH_SCAN_THREADS), lds, st, p.db, p.n_rows, tile_stride,
                           rowmask, p.unit_frags, p.n_q, tau, lmax, p.unit_recs, p.unit_counts + 1, next, p.slot_q, p.pairs,
                           p.wave_counts, p.row_ids, pool_head, p.chunk_fill, p.overflow, p.aux, p.aux_lay.stride, qaux);
}
void dph_launch_scan_units(const dph_pass& p, bool sample, int tile_stride, unsigned rowmask, const int* tau, hipStream_t st) {
    if (p.aux) launch_scan_units_a<true>(p, sample, tile_stride, rowmask, tau, st);
    else launch_scan_units_a<false>(p, sample, tile_stride, rowmask, tau, st);
}
