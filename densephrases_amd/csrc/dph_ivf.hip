// dph_ivf.hip -- the coarse quantizer of the IVF path (BASELINE.json configs[3]: IVF-4096 + in-list exact IP).
// Replaces the IndexFlatIP coarse search inside faiss IndexIVF that the reference offloads to the GPU
// (/root/reference/densephrases/index.py:52-56, nprobe 256 at :53/:62; nlist up to 2^20: model.py:18): scores =
// q . centroid^T, the nprobe best lists per query row in (score desc, list id asc) order, emitted as a list-major bit
// mask for the scan.
//   dph_coarse_gemm_kernel   scores[q][l] = <x_q, c_l> on the matrix cores: v_mfma_f32_32x32x2_f32 (exact fp32 products,
//                            fp32 accumulation -- the f32-in MFMA runs at the fp32 vector rate, which is plenty: 412 GFLOP
//                            for 256 query rows x 2^20 lists), LDS-tiled 128 lists x 128 query rows per workgroup
//   dph_coarse_select_kernel one workgroup per query row: the nprobe-th largest fp32 score by a 3-pass radix select;
//                            lists clearly above it are probed, the lists within the fp32 error band around it are
//                            re-scored in fp64 and ranked (score desc, list id asc) -- so the probed SET is the one the
//                            float64 oracle (and FAISS up to its own rounding) picks, not merely a similar one
//   dph_tilemask_kernel      tilemask[tile] = listmask[list of tile] (32 B per 24 KiB tile: what the scan reads)
#include <stdio.h>
#include <algorithm>
#include "dph_internal.h"

typedef float v16f __attribute__((ext_vector_type(16)));

#define CG_LISTS 128        // lists per workgroup tile (4 waves x 32)
#define CG_QROWS 128        // query rows per workgroup tile (4 MFMA column blocks)
#define CG_KCHUNK 32        // k per LDS stage
#define CG_LD (CG_KCHUNK + 1)   // padded row stride (floats): lane&31 walks rows, so an odd stride is conflict-free

__global__ __launch_bounds__(256) void dph_coarse_gemm_kernel(const float* __restrict__ x, int q0, int n_q_host,
                                                              const int* __restrict__ gate, int gate_base,
                                                              const float* __restrict__ centroids, int nlist,
                                                              float* __restrict__ scores) {
    __shared__ float a_lds[CG_LISTS * CG_LD];
    __shared__ float b_lds[CG_QROWS * CG_LD];
    const int n_q = dph_gated_rows(gate, gate_base, n_q_host);
    const int qb0 = blockIdx.y * CG_QROWS;
    if (qb0 >= n_q) return;
    const int l0 = blockIdx.x * CG_LISTS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    v16f acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int k0 = 0; k0 < DPH_DIM; k0 += CG_KCHUNK) {
        // stage: 128 rows x 32 floats each for centroids and queries: thread t loads float4 #(t&7) of rows t>>3 + 32*i
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 3) + 32 * i, c4 = (tid & 7) * 4;
            const int l = l0 + row, q = qb0 + row;
            const float4 av = l < nlist ? *(const float4*)(centroids + (int64_t)l * DPH_DIM + k0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 bv = q < n_q ? *(const float4*)(x + (int64_t)(q0 + q) * DPH_DIM + k0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            float* ap = a_lds + row * CG_LD + c4;
            float* bp = b_lds + row * CG_LD + c4;
            ap[0] = av.x; ap[1] = av.y; ap[2] = av.z; ap[3] = av.w;
            bp[0] = bv.x; bp[1] = bv.y; bp[2] = bv.z; bp[3] = bv.w;
        }
        __syncthreads();
        // A[i = lane&31][k = lane>>5] = centroid row, B[k = lane>>5][j = lane&31] = query row (cdna_hip_programming.md section 3)
        const float* ap = a_lds + (wave * 32 + (lane & 31)) * CG_LD + (lane >> 5);
        const float* bp = b_lds + (lane & 31) * CG_LD + (lane >> 5);
#pragma unroll
        for (int kk = 0; kk < CG_KCHUNK; kk += 2) {
            const float a = ap[kk];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[j * 32 * CG_LD + kk], acc[j], 0, 0, 0);
        }
        __syncthreads();
    }
    // C[row i = (r&3) + 8*(r>>2) + 4*(lane>>5)][col j = lane&31]: i = list inside the wave's 32, j = query row of block jb
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = qb0 + j * 32 + (lane & 31);
        if (q >= n_q) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int l = l0 + wave * 32 + 8 * g + 4 * (lane >> 5);
            float* o = scores + (int64_t)q * nlist + l;
            if (l + 3 < nlist && (nlist & 3) == 0) {
                *(float4*)o = make_float4(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (l + r < nlist) o[r] = acc[j][4 * g + r];
            }
        }
    }
}

// ---- the same scores through the bf16 matrix cores, for the long coarse quantizers (the reference's 2^20 lists: 2*10^11 flop per
// 128 query rows is 2 ms at the f32 MFMA rate).  Every fp32 operand is split into two bf16 numbers, v = hi + lo + O(2^-16 |v|),
// and <x, c> is taken as hi*hi + hi*lo + lo*hi with three v_mfma_f32_32x32x16_bf16 per tile step (16x the f32-in rate each) into
// ONE fp32 accumulator: the dropped lo*lo and rounding residues are <= 3.1 * 2^-16 * ||x|| * ||c||, the accumulation of 2304
// products <= 2304 * 2^-24 * ||x|| * ||c|| -- the select kernel widens its float64 re-rank band by exactly that
// (CG_BF16X3_ERR), so the probed set stays the float64 oracle's.
typedef short v8s __attribute__((ext_vector_type(8)));
#define CG3_LD 40                    // bf16 per LDS row: 32 of the k-chunk + 8 of padding (80 B: 16-byte aligned, conflict-free)
#define CG_BF16X3_MIN (1 << 16)      // lists from which the bf16x3 kernel is used
#define CG_F32_ERR (768.0 * 5.97e-8)
#define CG_BF16X3_ERR (2304.0 * 5.97e-8 + 3.1 / 65536.0)

__device__ __forceinline__ unsigned short bf16_rne(float v) {
    const unsigned u = __float_as_uint(v);
    return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ void bf16_split(float v, unsigned short& hi, unsigned short& lo) {
    hi = bf16_rne(v);
    lo = bf16_rne(v - __uint_as_float((unsigned)hi << 16));        // exact difference, then rounded
}

// split of a whole fp32 matrix [n, 768] into its bf16 hi / lo parts (the centroids once per index, the rotated queries once per
// pass): with both operands pre-split the GEMM's staging is plain copies -- in-kernel splitting (~10 VALU ops per element and
// stage) kept the VALU busier than the matrix pipe
// (packed as hi << 16 | lo in one 32-bit word per element: the same 128-byte runs per row and k-chunk as the fp32 matrix, and
// two v_perm per pair of elements unpack them -- separate hi / lo arrays halved the run length and ran slower than splitting in
// the kernel)
__global__ __launch_bounds__(256) void dph_bf16_split_kernel(const float* __restrict__ v, int64_t n_elems, unsigned* __restrict__ packed) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_elems; i += (int64_t)gridDim.x * 256) {
        unsigned short hi, lo;
        bf16_split(v[i], hi, lo);
        packed[i] = ((unsigned)hi << 16) | (unsigned)lo;
    }
}
void dph_launch_bf16_split(const float* v, int64_t n_elems, unsigned* packed, hipStream_t st) {
    if (n_elems > 0)
        hipLaunchKernelGGL(dph_bf16_split_kernel, dim3((unsigned)std::min<int64_t>((n_elems + 255) / 256, 1 << 16)), dim3(256), 0, st, v, n_elems, packed);
}

template <bool PRE>
__global__ __launch_bounds__(256) void dph_coarse_gemm_bf16x3_kernel(const float* __restrict__ x, int q0, int n_q_host,
                                                                     const int* __restrict__ gate, int gate_base,
                                                                     const float* __restrict__ centroids, int nlist,
                                                                     float* __restrict__ scores, const unsigned* __restrict__ c_pk,
                                                                     const unsigned* __restrict__ x_pk) {
    __shared__ __attribute__((aligned(16))) unsigned short a_hi[CG_LISTS * CG3_LD], a_lo[CG_LISTS * CG3_LD];
    __shared__ __attribute__((aligned(16))) unsigned short b_hi[CG_QROWS * CG3_LD], b_lo[CG_QROWS * CG3_LD];
    const int n_q = dph_gated_rows(gate, gate_base, n_q_host);
    const int qb0 = blockIdx.y * CG_QROWS;
    if (qb0 >= n_q) return;
    const int l0 = blockIdx.x * CG_LISTS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    v16f acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int k0 = 0; k0 < DPH_DIM; k0 += CG_KCHUNK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 3) + 32 * i, c4 = (tid & 7) * 4;
            const int l = l0 + row, q = qb0 + row;
            if constexpr (PRE) {
                const uint4 z = make_uint4(0u, 0u, 0u, 0u);
                const uint4 av = l < nlist ? *(const uint4*)(c_pk + (int64_t)l * DPH_DIM + k0 + c4) : z;
                const uint4 bv = q < n_q ? *(const uint4*)(x_pk + (int64_t)(q0 + q) * DPH_DIM + k0 + c4) : z;
                // word = hi << 16 | lo: the hi halves of two elements / their lo halves into one register each
                *(uint2*)(a_hi + row * CG3_LD + c4) = make_uint2(__builtin_amdgcn_perm(av.y, av.x, 0x07060302u), __builtin_amdgcn_perm(av.w, av.z, 0x07060302u));
                *(uint2*)(a_lo + row * CG3_LD + c4) = make_uint2(__builtin_amdgcn_perm(av.y, av.x, 0x05040100u), __builtin_amdgcn_perm(av.w, av.z, 0x05040100u));
                *(uint2*)(b_hi + row * CG3_LD + c4) = make_uint2(__builtin_amdgcn_perm(bv.y, bv.x, 0x07060302u), __builtin_amdgcn_perm(bv.w, bv.z, 0x07060302u));
                *(uint2*)(b_lo + row * CG3_LD + c4) = make_uint2(__builtin_amdgcn_perm(bv.y, bv.x, 0x05040100u), __builtin_amdgcn_perm(bv.w, bv.z, 0x05040100u));
                continue;
            }
            const float4 av = l < nlist ? *(const float4*)(centroids + (int64_t)l * DPH_DIM + k0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 bv = q < n_q ? *(const float4*)(x + (int64_t)(q0 + q) * DPH_DIM + k0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            unsigned short h[4], lw[4];
            bf16_split(av.x, h[0], lw[0]); bf16_split(av.y, h[1], lw[1]); bf16_split(av.z, h[2], lw[2]); bf16_split(av.w, h[3], lw[3]);
            *(uint2*)(a_hi + row * CG3_LD + c4) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
            *(uint2*)(a_lo + row * CG3_LD + c4) = make_uint2((unsigned)lw[0] | ((unsigned)lw[1] << 16), (unsigned)lw[2] | ((unsigned)lw[3] << 16));
            bf16_split(bv.x, h[0], lw[0]); bf16_split(bv.y, h[1], lw[1]); bf16_split(bv.z, h[2], lw[2]); bf16_split(bv.w, h[3], lw[3]);
            *(uint2*)(b_hi + row * CG3_LD + c4) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
            *(uint2*)(b_lo + row * CG3_LD + c4) = make_uint2((unsigned)lw[0] | ((unsigned)lw[1] << 16), (unsigned)lw[2] | ((unsigned)lw[3] << 16));
        }
        __syncthreads();
        // A[i = lane&31][k = 8*(lane>>5) .. +7] = centroid row, B[k = 8*(lane>>5) .. +7][j = lane&31] = query row
        const int ko = 8 * (lane >> 5);
        const unsigned short* ap_h = a_hi + (wave * 32 + (lane & 31)) * CG3_LD + ko;
        const unsigned short* ap_l = a_lo + (wave * 32 + (lane & 31)) * CG3_LD + ko;
#pragma unroll
        for (int kk = 0; kk < CG_KCHUNK; kk += 16) {
            const v8s ah = *(const v8s*)(ap_h + kk), al = *(const v8s*)(ap_l + kk);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const v8s bh = *(const v8s*)(b_hi + (j * 32 + (lane & 31)) * CG3_LD + ko + kk);
                const v8s bl = *(const v8s*)(b_lo + (j * 32 + (lane & 31)) * CG3_LD + ko + kk);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = qb0 + j * 32 + (lane & 31);
        if (q >= n_q) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int l = l0 + wave * 32 + 8 * g + 4 * (lane >> 5);
            float* o = scores + (int64_t)q * nlist + l;
            if (l + 3 < nlist && (nlist & 3) == 0) {
                *(float4*)o = make_float4(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (l + r < nlist) o[r] = acc[j][4 * g + r];
            }
        }
    }
}

// The pre-split form, software-pipelined: k-chunks of 64 (each MFMA step is fed from LDS twice as long per barrier), the NEXT
// chunk's global loads are issued into registers before the current chunk is multiplied, so HBM latency overlaps the matrix work
// inside a workgroup instead of relying on two co-resident workgroups (the plain form above: 16 % matrix-pipe utilisation).
#define CG3P_K 64
#define CG3P_LD 72                   // bf16 per LDS row: 64 + 8 of padding (144 B: 16-byte aligned, conflict-free for ds_read_b128)
__global__ __launch_bounds__(256, 2) void dph_coarse_gemm_bf16x3_pipe_kernel(int q0, int n_q_host, const int* __restrict__ gate, int gate_base,
                                                                             int nlist, float* __restrict__ scores,
                                                                             const unsigned* __restrict__ c_pk,
                                                                             const unsigned* __restrict__ x_pk) {
    extern __shared__ __attribute__((aligned(16))) unsigned short cg_lds[];       // a_hi | a_lo | b_hi | b_lo, each [128][CG3P_LD]
    unsigned short* const a_hi = cg_lds;
    unsigned short* const a_lo = a_hi + CG_LISTS * CG3P_LD;
    unsigned short* const b_hi = a_lo + CG_LISTS * CG3P_LD;
    unsigned short* const b_lo = b_hi + CG_QROWS * CG3P_LD;
    const int n_q = dph_gated_rows(gate, gate_base, n_q_host);
    const int qb0 = blockIdx.y * CG_QROWS;
    if (qb0 >= n_q) return;
    const int l0 = blockIdx.x * CG_LISTS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = tid & 15, row0 = tid >> 4;                    // 16 uint4 (64 packed elements) per row, rows row0 + 16 i
    v16f acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    uint4 ra[8], rb[8];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = row0 + 16 * i, l = l0 + row, q = qb0 + row;
            ra[i] = l < nlist ? *(const uint4*)(c_pk + (int64_t)l * DPH_DIM + k0 + 4 * col) : make_uint4(0u, 0u, 0u, 0u);
            rb[i] = q < n_q ? *(const uint4*)(x_pk + (int64_t)(q0 + q) * DPH_DIM + k0 + 4 * col) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < DPH_DIM; k0 += CG3P_K) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int o = (row0 + 16 * i) * CG3P_LD + 4 * col;
            *(uint2*)(a_hi + o) = make_uint2(__builtin_amdgcn_perm(ra[i].y, ra[i].x, 0x07060302u), __builtin_amdgcn_perm(ra[i].w, ra[i].z, 0x07060302u));
            *(uint2*)(a_lo + o) = make_uint2(__builtin_amdgcn_perm(ra[i].y, ra[i].x, 0x05040100u), __builtin_amdgcn_perm(ra[i].w, ra[i].z, 0x05040100u));
            *(uint2*)(b_hi + o) = make_uint2(__builtin_amdgcn_perm(rb[i].y, rb[i].x, 0x07060302u), __builtin_amdgcn_perm(rb[i].w, rb[i].z, 0x07060302u));
            *(uint2*)(b_lo + o) = make_uint2(__builtin_amdgcn_perm(rb[i].y, rb[i].x, 0x05040100u), __builtin_amdgcn_perm(rb[i].w, rb[i].z, 0x05040100u));
        }
        __syncthreads();
        if (k0 + CG3P_K < DPH_DIM) fetch(k0 + CG3P_K);            // in flight while this chunk is multiplied
        const int ko = 8 * (lane >> 5);
        const unsigned short* ap_h = a_hi + (wave * 32 + (lane & 31)) * CG3P_LD + ko;
        const unsigned short* ap_l = a_lo + (wave * 32 + (lane & 31)) * CG3P_LD + ko;
#pragma unroll
        for (int kk = 0; kk < CG3P_K; kk += 16) {
            const v8s ah = *(const v8s*)(ap_h + kk), al = *(const v8s*)(ap_l + kk);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const v8s bh = *(const v8s*)(b_hi + (j * 32 + (lane & 31)) * CG3P_LD + ko + kk);
                const v8s bl = *(const v8s*)(b_lo + (j * 32 + (lane & 31)) * CG3P_LD + ko + kk);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[j], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = qb0 + j * 32 + (lane & 31);
        if (q >= n_q) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int l = l0 + wave * 32 + 8 * g + 4 * (lane >> 5);
            float* o = scores + (int64_t)q * nlist + l;
            if (l + 3 < nlist && (nlist & 3) == 0) {
                *(float4*)o = make_float4(acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) if (l + r < nlist) o[r] = acc[j][4 * g + r];
            }
        }
    }
}

__device__ __forceinline__ unsigned f32_key(float v) {          // order-preserving map to unsigned
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_f32(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

#define CS_THREADS 1024
#define CS_BPT (2048 / CS_THREADS)     // histogram bins per thread
#define CS_BAND_CAP 2048
#define CS_BROWS 6                   // band lists whose rows a wave has in flight together (float64 re-rank)
#define CS_SAMPLE 8192               // scores sampled for the threshold estimate of the one-pass path
#define CS_CAND 8192                 // candidates that path keeps in LDS (the filter form's wider band wants ~4 x nprobe + noise)
#define CS_FAST_MIN 8192             // lists from which the one-pass path is tried
// np-th largest of n order-preserving keys in three passes of 11 / 11 / 10 bits (2048-bin histogram in LDS).  `key_at(i)` reads
// key i; all CS_THREADS threads call it.  Keys fall into few distinct bins in the first pass (sign, exponent, two mantissa
// bits), so a wave first merges equal bins with ballots and issues ONE LDS atomic per distinct bin instead of 64 colliding ones.
template <class F>
__device__ __forceinline__ unsigned cs_select_kth(F key_at, int n, int want, unsigned* hist, unsigned* sh) {
    const int tid = threadIdx.x;
    unsigned prefix = 0, pmask = 0;
    const int shifts[3] = {21, 10, 0}, widths[3] = {11, 11, 10};
    for (int p = 0; p < 3; ++p) {
        for (int i = tid; i < 2048; i += CS_THREADS) hist[i] = 0;
        __syncthreads();
        const unsigned dm = (1u << widths[p]) - 1u;
        for (int i0 = 0; i0 < n; i0 += CS_THREADS) {
            const int i = i0 + tid;
            unsigned bin = 0xFFFFFFFFu;                  // not a member
            if (i < n) { const unsigned k = key_at(i); if ((k & pmask) == prefix) bin = (k >> shifts[p]) & dm; }
            // at most four merge rounds: keys spread over many bins (the later passes, a sample of a whole score row) are counted
            // one atomic per key after that -- 64 rounds of ballots per wave and trip otherwise (2 us per trip of 1024 keys)
            unsigned long long todo = __builtin_amdgcn_ballot_w64(bin != 0xFFFFFFFFu);
            for (int round = 0; round < 4 && todo; ++round) {
                const int leader = __builtin_ctzll(todo);
                const unsigned b = (unsigned)__builtin_amdgcn_readlane((int)bin, leader);
                const unsigned long long same = __builtin_amdgcn_ballot_w64(bin == b);
                if ((tid & 63) == leader) atomicAdd(&hist[b], (unsigned)__builtin_popcountll(same));
                todo &= ~same;
            }
            if ((todo >> (tid & 63)) & 1ull) atomicAdd(&hist[bin], 1u);
        }
        __syncthreads();
        {
            // the bin holding the want-th largest key: thread t owns bins CS_BPT t .. CS_BPT t + CS_BPT - 1, suffix sums from the
            // top bin down over the lanes (shuffles) and the waves (LDS) -- one thread walking 2048 bins cost 25 us per pass
            const int lane = tid & 63, wv = tid >> 6;
            unsigned h[CS_BPT];
            unsigned mine = 0;
#pragma unroll
            for (int u = 0; u < CS_BPT; ++u) { h[u] = hist[CS_BPT * tid + u]; mine += h[u]; }
            unsigned incl = mine;                                  // sum over this wave's lanes >= lane
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const unsigned v = __shfl_down(incl, o); if (lane + o < 64) incl += v; }
            if (lane == 0) sh[8 + wv] = incl;                      // the wave's total
            if (tid == 0) { sh[0] = 0; sh[1] = (unsigned)want; }   // (no bin reaches the rank: bin 0 takes it, as before)
            __syncthreads();
            unsigned above = incl - mine;
            for (int w = wv + 1; w < CS_THREADS / 64; ++w) above += sh[8 + w];
            if (above < (unsigned)want && (unsigned)want <= above + mine) {
                unsigned need = (unsigned)want - above;            // rank inside this thread's bins, counted from its top bin down
                int bin = CS_BPT * tid;
                unsigned acc = 0;
                bool found = false;
#pragma unroll
                for (int u = CS_BPT - 1; u >= 1; --u) {
                    if (!found && need <= acc + h[u]) { bin = CS_BPT * tid + u; need -= acc; found = true; }
                    acc += h[u];
                }
                if (!found) need -= acc;                           // the thread's lowest bin
                if (bin > 0) { sh[0] = (unsigned)bin; sh[1] = need; }
                else { sh[0] = 0; sh[1] = (unsigned)want - (above + acc); }
            }
        }
        __syncthreads();
        prefix |= sh[0] << shifts[p];
        pmask |= dm << shifts[p];
        want = (int)sh[1];
        __syncthreads();
    }
    return prefix;
}


// ---------------------------------------------------------------------------------------------------------------------
// Long quantizers, the filter form (round 4).  The bf16x3 GEMM above reads 4 bytes per centroid component and writes a
// [rows, nlist] score matrix that two more passes read back (estimate + collect): at 2^20 lists that is 3.2 GB + 2 x 0.5 GB
// per batch of 64.  The probe set only needs the scores NEAR the nprobe-th one exactly, so the first pass is a FILTER:
//   * the centroids once more as plain bf16 (the hi parts: 2 bytes per component, 1.6 GB at 2^20 lists), the rotated queries
//     likewise; <x~, c~> with ONE v_mfma_f32_32x32x16_bf16 per tile step.  |<x~,c~> - <x,c>| <= (2^-8 + 2^-18) sum|x_j c_j|
//     (two roundings to 8 significant bits) + the fp32 accumulation <= CF_HI_ERR * ||x|| * ||c||;
//   * a threshold estimate per query row from the scores of every (nlist/8192)-th list (the same kernel over a strided sample:
//     12.6 MB), an order statistic chosen so that ~4-5 x nprobe lists beat it;
//   * the threshold test sits in the GEMM's epilogue: a (row, list, key) triple per score at or above the row's estimate goes
//     to a pool (LDS-aggregated, one global atomic per workgroup tile) -- the score matrix is never written;
//   * dph_coarse_bucket_kernel deals the pool into per-row candidate lists, dph_coarse_select_kernel selects the nprobe-th
//     candidate, marks what is clearly above the error band and re-ranks the band in float64 exactly as before -- the band is
//     wider (the 2^-8 error: a few hundred lists around the 256-th of 2^20), which is what the float64 re-rank is for.
// A row whose estimate was too high (fewer than nprobe candidates, or the band reaching below it), whose candidates or band
// overflow, fails the whole pass over to the bf16x3 chain (gated on the device: no host round trip; empty launches otherwise).
#define CF_K 128                     // k-chunk of the filter GEMM = one run of the tile-major centroid image
#define CF_HI_ERR (1.0 / 256.0 + 1.0 / 262144.0 + 800.0 * 5.97e-8)
#define CF_HIT_CAP 4096              // (row, list, key) triples one workgroup tile can hold before the pass fails over
#define CF_SAMPLE 8192               // lists of the threshold sample

// The bf16 image of a [n, 768] fp32 matrix the filter GEMM reads.  tiled = 0: row-major (the query rows of a pass).  tiled = 1 (the
// centroids): CHUNK-MAJOR -- [k-chunk of 128][tile of 128 rows][row][k]: the 32 KiB a workgroup stages per step are one contiguous run,
// and the runs the workgroups of a launch read in the same step (tiles w, w+1, ... of one k-chunk) follow each other -- hundreds of
// workgroups sweep one 16 MiB window together.  Row-major, a step gathered 128 pieces of 256 B at a stride of 1536 B (4.1 TB/s whatever
// was in flight); tile-major ([tile][k-chunk]...: runs 192 KiB apart, every workgroup on the same few memory channels at the same
// moment) was three times slower than that.  Rows past n in the last tile are written as zeros (the allocation holds whole tiles).
__host__ __device__ inline int64_t dph_cf_tiled_index(int64_t row, int k, int64_t n_tiles) {
    return ((int64_t)(k / CF_K) * n_tiles + row / CG_LISTS) * (int64_t)(CG_LISTS * CF_K) + (row % CG_LISTS) * CF_K + k % CF_K;
}
__global__ __launch_bounds__(256) void dph_bf16_hi_kernel(const float* __restrict__ v, int64_t n_rows, int64_t n_rows_padded, int tiled,
                                                          unsigned short* __restrict__ hi) {
    const int64_t n_elems = n_rows_padded * DPH_DIM;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_elems; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / DPH_DIM;
        const int k = (int)(i % DPH_DIM);
        const unsigned short h = row < n_rows ? bf16_rne(v[i]) : (unsigned short)0;
        hi[tiled ? dph_cf_tiled_index(row, k, n_rows_padded / CG_LISTS) : i] = h;
    }
}
int64_t dph_bf16_hi_rows(int64_t n_rows, int tiled) { return tiled ? (n_rows + CG_LISTS - 1) / CG_LISTS * CG_LISTS : n_rows; }
void dph_launch_bf16_hi(const float* v, int64_t n_rows, int tiled, unsigned short* hi, hipStream_t st) {
    const int64_t n_elems = dph_bf16_hi_rows(n_rows, tiled) * DPH_DIM;
    if (n_elems > 0)
        hipLaunchKernelGGL(dph_bf16_hi_kernel, dim3((unsigned)std::min<int64_t>((n_elems + 255) / 256, 1 << 16)), dim3(256), 0, st, v, n_rows,
                           dph_bf16_hi_rows(n_rows, tiled), tiled, hi);
}

// SAMPLE: lists i * list_stride, i < n_lists, scores written to sample_scores[q][i].  Otherwise: all n_lists lists, hits to the pool.
// Persistent workgroups (two per CU: grid.x = 2 x CUs, or fewer tiles): workgroup w multiplies the list tiles w, w + grid.x, ... as ONE
// stream of k-chunks of 128 -- the centroid chunk two steps ahead and the query chunk one step ahead are in flight in registers while
// the current chunk is multiplied out of LDS, across tile boundaries (the epilogue of a tile runs under the next tile's loads).  One
// chunk ahead kept 64 KiB per CU in flight: 4.1 TB/s at the ~4 us the loads take under load; two chunks ahead doubles that.
typedef unsigned cf_v4u __attribute__((ext_vector_type(4)));
// every byte of the centroid image is read once per launch by one workgroup: tuning key "coarse_nt" = 1 marks those loads non-temporal
#define CF_STREAM_LOAD(p) (NT ? __builtin_nontemporal_load(p) : *(p))
template <bool SAMPLE, bool NT = false>
__global__ __launch_bounds__(256, 2) void dph_coarse_filter_gemm_kernel(int n_q, int n_lists, int list_stride, int64_t image_tiles,
                                                                        const unsigned short* __restrict__ c_hi,
                                                                        const unsigned short* __restrict__ x_hi,
                                                                        float* __restrict__ sample_scores, const unsigned* __restrict__ est,
                                                                        uint2* __restrict__ pool_lk, unsigned short* __restrict__ pool_q,
                                                                        unsigned* __restrict__ pool_count, unsigned pool_cap,
                                                                        unsigned* __restrict__ fail) {
    constexpr int CK = CF_K;
    constexpr int LD = CK + 8;                                    // bf16 per LDS row (CK + 8 of padding: 16-byte aligned rows, conflict-free ds_read_b128)
    constexpr int NF = CK / 16;                                   // uint4 per thread and operand per chunk (256 threads, 128 rows of CK bf16)
    constexpr int CPR = CK / 8;                                   // uint4 per row
    constexpr int NCH = DPH_DIM / CK;                             // chunks per tile
    constexpr unsigned HIT_CAP = (unsigned)CF_HIT_CAP;            // (row, list, key) triples the staging area holds afterwards (10 bytes each)
    extern __shared__ __attribute__((aligned(16))) unsigned short cf_lds[];       // a | b, each [128][LD]; afterwards the hit list
    unsigned short* const a_s = cf_lds;
    unsigned short* const b_s = a_s + CG_LISTS * LD;
    __shared__ unsigned hit_n;
    __shared__ unsigned hit_base;
    const int qb0 = blockIdx.y * CG_QROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = tid % CPR, row0 = tid / CPR;                  // rows row0 + (256 / CPR) i
    constexpr int RSTEP = 256 / CPR;
    const int n_tiles = (n_lists + CG_LISTS - 1) / CG_LISTS;
    const int my_tiles = (int)blockIdx.x < n_tiles ? (n_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int n_chunks = my_tiles * NCH;                          // of this workgroup's stream
    v16f acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // staging registers: the centroid chunks two steps ahead (RA0 / RA1 alternate) and the query chunk one step ahead.  Plain named
    // arrays touched through MACROS with constant indices: behind a reference parameter of a lambda hipcc left them in scratch
    // memory (256 bytes per thread, every chunk through it: the GEMM took 1.19 ms instead of 0.4)
    cf_v4u ra0[NF], ra1[NF], rb[NF];        // (native vectors: arrays of HIP's uint4 struct were not always split into registers either)
#define CF_FETCH_A(RA, C)                                                                                                        \
    do {                                                                                                                         \
        const int c_ = (C);                                                                                                      \
        const int l0_ = ((int)blockIdx.x + (c_ / NCH) * (int)gridDim.x) * CG_LISTS, k0_ = (c_ % NCH) * CK;                        \
        if constexpr (SAMPLE) {                                                                                                  \
            /* the sample's lists are every list_stride-th list of the index: 256-byte pieces of the chunk-major image */        \
            _Pragma("unroll") for (int i = 0; i < NF; ++i) {                                                                     \
                const int l = l0_ + row0 + RSTEP * i;                                                                            \
                RA[i] = l < n_lists ? *(const cf_v4u*)(c_hi + dph_cf_tiled_index((int64_t)l * list_stride, k0_ + 8 * col, image_tiles)) \
                                    : cf_v4u{0u, 0u, 0u, 0u};                                                                \
            }                                                                                                                    \
        } else {                                                                                                                 \
            /* chunk (k0 / CK, tile) of the chunk-major image: 32 KiB in one run (rows past n_lists in the last tile are zeros) */ \
            const cf_v4u* src_ = (const cf_v4u*)(c_hi + ((int64_t)(c_ % NCH) * image_tiles + l0_ / CG_LISTS) * (int64_t)(CG_LISTS * CK)); \
            _Pragma("unroll") for (int i = 0; i < NF; ++i) RA[i] = CF_STREAM_LOAD(src_ + (row0 + RSTEP * i) * CPR + col);           \
        }                                                                                                                        \
    } while (0)
    auto fetch_b = [&](int c) __attribute__((always_inline)) {
        const int k0 = (c % NCH) * CK;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int q = qb0 + row0 + RSTEP * i;
            rb[i] = q < n_q ? *(const cf_v4u*)(x_hi + (int64_t)q * DPH_DIM + k0 + 8 * col) : cf_v4u{0u, 0u, 0u, 0u};
        }
    };
    // (always_inline: as an out-of-line call the lambda takes the accumulators by reference, i.e. through scratch memory -- 3x the kernel time)
    auto epilogue = [&](int tile) __attribute__((always_inline)) {
        const int l0 = tile * CG_LISTS;
        if constexpr (SAMPLE) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = qb0 + j * 32 + (lane & 31);
                if (q >= n_q) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int l = l0 + wave * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                    if (l < n_lists) sample_scores[(int64_t)q * n_lists + l] = acc[j][r];
                }
            }
        } else {
            // every score at or above its row's estimate becomes a (row, list, key) triple; the tile's triples are gathered in LDS
            // (the staging area is free: the chunk loop ended on a barrier) and leave with ONE global atomic
            uint2* const hit_lk = (uint2*)cf_lds;                     // [HIT_CAP]
            unsigned short* const hit_q = (unsigned short*)(hit_lk + HIT_CAP);
            if (tid == 0) hit_n = 0;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int q = qb0 + j * 32 + (lane & 31);
                const unsigned e = q < n_q ? est[q] : 0xFFFFFFFFu;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int l = l0 + wave * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                    const unsigned key = f32_key(acc[j][r]);
                    const bool hit = q < n_q && l < n_lists && key >= e;
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
                    if (m == 0ull) continue;
                    unsigned base = 0;
                    if (lane == __builtin_ctzll(m)) base = atomicAdd(&hit_n, (unsigned)__builtin_popcountll(m));
                    base = (unsigned)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(m));
                    if (hit) {
                        const unsigned slot = base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                        if (slot < HIT_CAP) { hit_lk[slot] = make_uint2((unsigned)l, key); hit_q[slot] = (unsigned short)q; }
                    }
                }
            }
            __syncthreads();
            const unsigned n = hit_n;
            if (tid == 0) {
                unsigned b = 0;
                if (n > 0) b = atomicAdd(pool_count, n < HIT_CAP ? n : HIT_CAP);
                if (n > HIT_CAP || (n > 0 && b + n > pool_cap)) atomicOr(fail, 1u);
                hit_base = b;
            }
            __syncthreads();
            const unsigned b = hit_base, nn = n < HIT_CAP ? n : HIT_CAP;
            for (unsigned i = tid; i < nn; i += 256)
                if (b + i < pool_cap) { pool_lk[b + i] = hit_lk[i]; pool_q[b + i] = hit_q[i]; }
            __syncthreads();                                          // the staging area goes back to the chunk stream
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    };
    // one chunk: stage it (the centroid registers first, re-loaded at once), multiply, and at a tile's last chunk run its epilogue
#define CF_STEP(RA, C)                                                                                                           \
    do {                                                                                                                         \
        const int cc_ = (C);                                                                                                     \
        _Pragma("unroll") for (int i = 0; i < NF; ++i) *(cf_v4u*)(a_s + (row0 + RSTEP * i) * LD + 8 * col) = RA[i];                \
        if (cc_ + 2 < n_chunks) CF_FETCH_A(RA, cc_ + 2);                                                                         \
        _Pragma("unroll") for (int i = 0; i < NF; ++i) *(cf_v4u*)(b_s + (row0 + RSTEP * i) * LD + 8 * col) = rb[i];                \
        if (cc_ + 1 < n_chunks) fetch_b(cc_ + 1);                                                                                \
        __syncthreads();                                                                                                         \
        const int ko = 8 * (lane >> 5);                                                                                          \
        const unsigned short* ap = a_s + (wave * 32 + (lane & 31)) * LD + ko;                                                    \
        _Pragma("unroll") for (int kk = 0; kk < CK; kk += 16) {                                                                  \
            const v8s a = *(const v8s*)(ap + kk);                                                                                \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                      \
                const v8s b = *(const v8s*)(b_s + (j * 32 + (lane & 31)) * LD + ko + kk);                                        \
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);                                         \
            }                                                                                                                    \
        }                                                                                                                        \
        __syncthreads();                                                                                                         \
        if (cc_ % NCH == NCH - 1) epilogue((int)blockIdx.x + (cc_ / NCH) * (int)gridDim.x);                                      \
    } while (0)
    if (n_chunks == 0) return;
    CF_FETCH_A(ra0, 0);
    CF_FETCH_A(ra1, 1);                                              // (NCH >= 2: chunk 1 exists)
    fetch_b(0);
    for (int c = 0; c < n_chunks; c += 2) {                          // NCH is even: the stream has an even number of chunks
        CF_STEP(ra0, c);
        CF_STEP(ra1, c + 1);
    }
#undef CF_STEP
#undef CF_FETCH_A
}

// ---- the filter GEMM, third form (tuning key "coarse_filter" = 3): the centroids never touch LDS.
// One workgroup of 8 waves per CU; a tile is 256 lists, wave w multiplies lists 32 w .. 32 w + 31 of it with all (<= 128) query rows.
// The centroid image is FRAGMENT-MAJOR -- [tile][wave][k-step of 16][lane][8 bf16]: lane l of wave w loads, for k-step ks, the 16 bytes
// the MFMA wants from it (list 32 w + (l & 31), k = 16 ks + 8 (l >> 5) ..), so every load instruction of a wave is one contiguous
// kilobyte, a wave's share of a tile 48 KiB in one run -- straight into MFMA operand registers: a ring of three k-chunks of 128 (two in
// flight, one being multiplied).  Only the QUERY chunk goes through LDS (double-buffered, ONE barrier per chunk), staged once per 256
// lists: a quarter of the LDS stores of the forms above, which staged 32 KiB of centroids and the same 32 KiB of queries again per 128
// lists -- and a ds_write_b128 costs ~54 cycles of matrix-pipe bubble (dph_scan.hip): the suspect for their 4 TB/s.
#define CF2_LISTS 256
#define CF2_THREADS 512
__host__ __device__ inline int64_t dph_cf_frag_index(int64_t row, int k) {
    const int64_t tile = row / CF2_LISTS;
    const int w = (int)(row % CF2_LISTS) / 32, r = (int)(row % 32), ks = k / 16, h = (k % 16) / 8, e = k % 8;
    return ((((tile * 8 + w) * (DPH_DIM / 16) + ks) * 64) + (h * 32 + r)) * 8 + e;
}
__global__ __launch_bounds__(256) void dph_bf16_frag_kernel(const float* __restrict__ v, int64_t n_rows, int64_t n_rows_padded,
                                                            unsigned short* __restrict__ out) {
    const int64_t n_elems = n_rows_padded * DPH_DIM;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_elems; i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / DPH_DIM;
        const int k = (int)(i % DPH_DIM);
        out[dph_cf_frag_index(row, k)] = row < n_rows ? bf16_rne(v[i]) : (unsigned short)0;
    }
}
int64_t dph_bf16_frag_rows(int64_t n_rows) { return (n_rows + CF2_LISTS - 1) / CF2_LISTS * CF2_LISTS; }
void dph_launch_bf16_frag(const float* v, int64_t n_rows, unsigned short* out, hipStream_t st) {
    const int64_t n_elems = dph_bf16_frag_rows(n_rows) * DPH_DIM;
    if (n_elems > 0)
        hipLaunchKernelGGL(dph_bf16_frag_kernel, dim3((unsigned)std::min<int64_t>((n_elems + 255) / 256, 1 << 16)), dim3(256), 0, st, v, n_rows,
                           dph_bf16_frag_rows(n_rows), out);
}

// ---- the filter as a SCAN (tuning key "coarse_filter" = 5; dph_scan.hip MODE 3): the centroids as 24 KiB pieces with the byte layout
// of an int8 tile -- [tile of 32 lists][half of k][32 rows][384 bf16] -- streamed by the flat scan's feed at the flat scan's rate.
__global__ __launch_bounds__(256) void dph_bf16_pieces_kernel(const float* __restrict__ v, int64_t n_rows, int64_t n_rows_padded,
                                                              unsigned short* __restrict__ out) {
    const int64_t n_elems = n_rows_padded * DPH_DIM;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_elems; i += (int64_t)gridDim.x * 256) {
        // out index i = ((tile * 2 + half) * 32 + r) * 384 + kk
        const int kk = (int)(i % 384), r = (int)((i / 384) % 32), half = (int)((i / (384 * 32)) % 2);
        const int64_t tile = i / (384 * 32 * 2), row = tile * 32 + r;
        out[i] = row < n_rows ? bf16_rne(v[row * DPH_DIM + 384 * half + kk]) : (unsigned short)0;
    }
}
int64_t dph_bf16_piece_rows(int64_t n_rows) { return (n_rows + 31) / 32 * 32; }
void dph_launch_bf16_pieces(const float* v, int64_t n_rows, unsigned short* out, hipStream_t st) {
    const int64_t n_elems = dph_bf16_piece_rows(n_rows) * DPH_DIM;
    if (n_elems > 0)
        hipLaunchKernelGGL(dph_bf16_pieces_kernel, dim3((unsigned)std::min<int64_t>((n_elems + 255) / 256, 1 << 16)), dim3(256), 0, st, v, n_rows,
                           dph_bf16_piece_rows(n_rows), out);
}
// query fragments of a pass of <= 128 rows for that scan: [group of 32 rows][half][k-step of 16][lane][8 bf16], lane l = query row
// 32 g + (l & 31), k = 384 half + 16 ks + 8 (l >> 5) .. +7 -- the B operand of v_mfma_f32_32x32x16_bf16 as the wave loads it.
// (Round 6 tried v_mfma_f32_16x16x32_bf16 here as well -- the shape that took 8 % off the int8 256-row pass: bit-identical pools, and no
// faster: coarse stage 0.306 / 1.045 / 1.993 ms at batch 64 / 256 / 512 against 0.289 / 1.004 / 1.86-1.97.  Not kept.)
// (n_groups > 1, the teams form: the blocks of all groups of 128 rows one after the other, g counts on through them)
__global__ __launch_bounds__(256) void dph_cf_qfrag_kernel(const unsigned short* __restrict__ x_hi, int n_q, uint4* __restrict__ qfrag, int n_groups = 1) {
    const int i = blockIdx.x * 256 + threadIdx.x;              // one 16-byte fragment each: 4 * 2 * 24 * 64 of them per group of 128 rows
    if (i >= n_groups * 4 * 2 * 24 * 64) return;
    const int lane = i % 64, ks = (i / 64) % 24, half = (i / (64 * 24)) % 2, g = i / (64 * 24 * 2);
    const int q = 32 * g + (lane & 31), k = 384 * half + 16 * ks + 8 * (lane >> 5);
    qfrag[i] = q < n_q ? *(const uint4*)(x_hi + (int64_t)q * DPH_DIM + k) : make_uint4(0u, 0u, 0u, 0u);
}
// the chunks the scan claimed from its pair pool -> the linear (list, key) / query row pool the bucket step reads
__global__ __launch_bounds__(256) void dph_cf_flatten_kernel(const uint2* __restrict__ pairs, const unsigned* __restrict__ chunk_fill,
                                                             const int* __restrict__ counters, uint2* __restrict__ pool_lk,
                                                             unsigned short* __restrict__ pool_q, unsigned* __restrict__ pool_count,
                                                             unsigned pool_cap, unsigned* __restrict__ fail, unsigned q_base) {
    // a chunk per WAVE and trip (lane 0 reserves the run, the wave copies it): a chunk per workgroup and trip with two barriers around
    // the atomic was 6-8 dependent global round trips per workgroup, 26 us for the ~1500 chunks of a pass of 128 rows
    const unsigned claimed = (unsigned)counters[1];
    const unsigned used = claimed < (unsigned)DPH_POOL_CHUNKS ? claimed : (unsigned)DPH_POOL_CHUNKS;
    if (blockIdx.x == 0 && threadIdx.x == 0 && (claimed > (unsigned)DPH_POOL_CHUNKS || counters[2] != 0)) atomicOr(fail, 1u);
    const unsigned lane = threadIdx.x & 63u, gw = blockIdx.x * 4u + (threadIdx.x >> 6), n_waves = gridDim.x * 4u;
    for (unsigned ch = gw; ch < used; ch += n_waves) {
        const unsigned raw = chunk_fill[ch];
        const unsigned n = raw < (unsigned)DPH_CHUNK_PAIRS ? raw : (unsigned)DPH_CHUNK_PAIRS;
        if (n == 0) continue;
        unsigned b = 0;
        if (lane == 0) {
            b = atomicAdd(pool_count, n);
            if (b + n > pool_cap) atomicOr(fail, 1u);
        }
        b = (unsigned)__builtin_amdgcn_readfirstlane((int)b);
#pragma unroll
        for (unsigned i0 = 0; i0 < (unsigned)DPH_CHUNK_PAIRS; i0 += 64u) {
            const unsigned i = i0 + lane;
            if (i < n && b + i < pool_cap) {
                const uint2 pr = pairs[(size_t)ch * DPH_CHUNK_PAIRS + i];
                pool_lk[b + i] = make_uint2(pr.x & 0xFFFFFu, pr.y);
                pool_q[b + i] = (unsigned short)(q_base + (pr.x >> 20));
            }
        }
    }
}

__global__ __launch_bounds__(CF2_THREADS, 1) void dph_coarse_filter_gemm2_kernel(int n_q, int n_lists, const unsigned short* __restrict__ c_frag,
                                                                                const unsigned short* __restrict__ x_hi,
                                                                                const unsigned* __restrict__ est, uint2* __restrict__ pool_lk,
                                                                                unsigned short* __restrict__ pool_q, unsigned* __restrict__ pool_count,
                                                                                unsigned pool_cap, unsigned* __restrict__ fail, int contiguous) {
    constexpr int CK = CF_K, LD = CK + 8, NCH = DPH_DIM / CK, KS = CK / 16;      // 128, 136, 6 chunks per tile, 8 k-steps per chunk
    constexpr unsigned HIT_CAP = (unsigned)CF_HIT_CAP;
    extern __shared__ __attribute__((aligned(16))) unsigned short cf2_lds[];      // b[2][128][LD] | hit list
    unsigned short* const b_s = cf2_lds;
    uint2* const hit_lk = (uint2*)(cf2_lds + 2 * CG_QROWS * LD);
    unsigned short* const hit_q = (unsigned short*)(hit_lk + HIT_CAP);
    __shared__ unsigned hit_n;
    __shared__ unsigned hit_base;
    const int qb0 = blockIdx.y * CG_QROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_tiles = (n_lists + CF2_LISTS - 1) / CF2_LISTS;
    // the tiles of this workgroup: every gridDim.x-th one, or (contiguous) a run of ceil(n_tiles / gridDim.x) consecutive tiles -- each CU
    // then walks its own window of the image front to back, like the scan kernel walks its segments
    const int per = (n_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int tile0 = contiguous ? (int)blockIdx.x * per : (int)blockIdx.x, tstep = contiguous ? 1 : (int)gridDim.x;
    const int my_tiles = contiguous ? max(0, min(per, n_tiles - tile0)) : ((int)blockIdx.x < n_tiles ? (n_tiles - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0);
    const int n_chunks = my_tiles * NCH;
    if (n_chunks == 0) return;
    v16f acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // centroid chunks: three register sets of 8 k-steps (named, constant indices only: see the scratch lesson above)
    cf_v4u a0[KS], a1[KS], a2[KS];
    cf_v4u rb[4];                                                  // this thread's 64 bytes of the next query chunk (512 threads x 64 B = 32 KiB)
    const int brow = tid >> 2, bcol = tid & 3;                     // query row of the chunk, quarter of its 256 bytes
#define CF2_FETCH_A(RA, C)                                                                                                       \
    do {                                                                                                                         \
        const int c_ = (C);                                                                                                      \
        const int64_t tile_ = (int64_t)tile0 + (int64_t)(c_ / NCH) * tstep;                                                      \
        const cf_v4u* src_ = (const cf_v4u*)(c_frag + (((tile_ * 8 + wave) * (DPH_DIM / 16) + (c_ % NCH) * KS) * 64) * 8) + lane; \
        _Pragma("unroll") for (int i = 0; i < KS; ++i) RA[i] = __builtin_nontemporal_load(src_ + i * 64);                         \
    } while (0)
    auto fetch_b = [&](int c) __attribute__((always_inline)) {
        const int k0 = (c % NCH) * CK, q = qb0 + brow;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            rb[i] = q < n_q ? *(const cf_v4u*)(x_hi + (int64_t)q * DPH_DIM + k0 + 32 * bcol + 8 * i) : cf_v4u{0u, 0u, 0u, 0u};
    };
    auto stage_b = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *(cf_v4u*)(b_s + (buf * CG_QROWS + brow) * LD + 32 * bcol + 8 * i) = rb[i];
    };
    auto epilogue = [&](int tile) __attribute__((always_inline)) {
        if (tid == 0) hit_n = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = qb0 + j * 32 + (lane & 31);
            const unsigned e = q < n_q ? est[q] : 0xFFFFFFFFu;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int l = tile * CF2_LISTS + wave * 32 + 8 * (r >> 2) + 4 * (lane >> 5) + (r & 3);
                const unsigned key = f32_key(acc[j][r]);
                const bool hit = q < n_q && l < n_lists && key >= e;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
                if (m == 0ull) continue;
                unsigned base = 0;
                if (lane == __builtin_ctzll(m)) base = atomicAdd(&hit_n, (unsigned)__builtin_popcountll(m));
                base = (unsigned)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(m));
                if (hit) {
                    const unsigned slot = base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    if (slot < HIT_CAP) { hit_lk[slot] = make_uint2((unsigned)l, key); hit_q[slot] = (unsigned short)q; }
                }
            }
        }
        __syncthreads();
        const unsigned n = hit_n;
        if (tid == 0) {
            unsigned b = 0;
            if (n > 0) b = atomicAdd(pool_count, n < HIT_CAP ? n : HIT_CAP);
            if (n > HIT_CAP || (n > 0 && b + n > pool_cap)) atomicOr(fail, 1u);
            hit_base = b;
        }
        __syncthreads();
        const unsigned b = hit_base, nn = n < HIT_CAP ? n : HIT_CAP;
        for (unsigned i = tid; i < nn; i += CF2_THREADS)
            if (b + i < pool_cap) { pool_lk[b + i] = hit_lk[i]; pool_q[b + i] = hit_q[i]; }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    };
    // chunk C (k-chunk KC of its tile, both known at compile time modulo the tile) out of register set RA and LDS buffer KC & 1; the centroid
    // chunk two steps on goes into the set two steps on, the query chunk one step on into the other LDS buffer -- whose last readers left at
    // the barrier that ended the previous step
#define CF2_STEP(RA, RA2, C, KC)                                                                                                 \
    do {                                                                                                                         \
        const int cc_ = (C);                                                                                                     \
        if (cc_ + 2 < n_chunks) CF2_FETCH_A(RA2, cc_ + 2);                                                                        \
        if (cc_ + 1 < n_chunks) { stage_b(((KC) + 1) & 1); if (cc_ + 2 < n_chunks) fetch_b(cc_ + 2); }                            \
        const unsigned short* bp_ = b_s + (((KC) & 1) * CG_QROWS + (lane & 31)) * LD + 8 * (lane >> 5);                          \
        _Pragma("unroll") for (int i = 0; i < KS; ++i) {                                                                         \
            const v8s a_ = __builtin_bit_cast(v8s, RA[i]);                                                                       \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                      \
                const v8s b_ = *(const v8s*)(bp_ + j * 32 * LD + 16 * i);                                                        \
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, b_, acc[j], 0, 0, 0);                                       \
            }                                                                                                                    \
        }                                                                                                                        \
        __syncthreads();                                                                                                         \
    } while (0)
    // prologue: centroid chunks 0 and 1 in flight, query chunk 0 staged, query chunk 1 in registers
    CF2_FETCH_A(a0, 0);
    CF2_FETCH_A(a1, 1);
    fetch_b(0);
    stage_b(0);
    fetch_b(1);
    __syncthreads();
    for (int c = 0; c < n_chunks; c += NCH) {                        // one tile per trip: k-chunks 0 .. 5, register sets 0 1 2 0 1 2
        CF2_STEP(a0, a2, c + 0, 0);
        CF2_STEP(a1, a0, c + 1, 1);
        CF2_STEP(a2, a1, c + 2, 2);
        CF2_STEP(a0, a2, c + 3, 3);
        CF2_STEP(a1, a0, c + 4, 4);
        CF2_STEP(a2, a1, c + 5, 5);
        epilogue(tile0 + (c / NCH) * tstep);
    }
#undef CF2_STEP
#undef CF2_FETCH_A
}

#define CB_THREADS 512
// pool -> per-row candidate lists [row][cand_cap] (list, key) + counts.  Every workgroup takes a contiguous slice of the pool, counts its
// rows in LDS, reserves each row's run with one global atomic and scatters: n_wg x rows atomics instead of one per triple.
__global__ __launch_bounds__(CB_THREADS) void dph_coarse_bucket_kernel(const uint2* __restrict__ pool_lk, const unsigned short* __restrict__ pool_q,
                                                                       const unsigned* __restrict__ pool_count, unsigned pool_cap, int n_q,
                                                                       uint2* __restrict__ cand_glob, unsigned* __restrict__ cand_cnt, int cand_cap) {
    __shared__ unsigned cnt[DPH_PASS_MAX];
    __shared__ unsigned base[DPH_PASS_MAX];
    const int tid = threadIdx.x;
    unsigned n = *pool_count;
    if (n > pool_cap) n = pool_cap;
    const unsigned per = (n + gridDim.x - 1) / gridDim.x;
    const unsigned lo = blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    for (int i = tid; i < n_q; i += CB_THREADS) cnt[i] = 0;
    __syncthreads();
    for (unsigned i = lo + tid; i < hi; i += CB_THREADS) atomicAdd(&cnt[pool_q[i]], 1u);
    __syncthreads();
    for (int i = tid; i < n_q; i += CB_THREADS) { base[i] = cnt[i] ? atomicAdd(&cand_cnt[i], cnt[i]) : 0u; cnt[i] = 0; }
    __syncthreads();
    for (unsigned i = lo + tid; i < hi; i += CB_THREADS) {
        const unsigned q = pool_q[i];
        const unsigned slot = base[q] + atomicAdd(&cnt[q], 1u);
        if (slot < (unsigned)cand_cap) cand_glob[(int64_t)q * cand_cap + slot] = pool_lk[i];
    }
}

// The same straight from the chunks the filter SCAN claimed (variant 5): no linear pool in between (dph_cf_flatten_kernel, 27 us for a
// pass of 128 rows: ~2500 returning atomics on one counter).  A workgroup takes a contiguous range of chunks, a wave a chunk at a time;
// pool_count only keeps the statistic (dph_debug_pq_coarse).
__global__ __launch_bounds__(CB_THREADS) void dph_coarse_bucket_chunks_kernel(const uint2* __restrict__ pairs, const unsigned* __restrict__ chunk_fill,
                                                                              const int* __restrict__ counters, int n_q_total, unsigned q_base,
                                                                              uint2* __restrict__ cand_glob, unsigned* __restrict__ cand_cnt, int cand_cap,
                                                                              unsigned* __restrict__ pool_count, unsigned* __restrict__ fail) {
    __shared__ unsigned cnt[DPH_PASS_MAX];
    __shared__ unsigned base[DPH_PASS_MAX];
    __shared__ unsigned total;
    const int tid = threadIdx.x;
    const unsigned lane = tid & 63u, wave = (unsigned)tid >> 6, n_waves = CB_THREADS / 64;
    const unsigned claimed = (unsigned)counters[1];
    const unsigned used = claimed < (unsigned)DPH_POOL_CHUNKS ? claimed : (unsigned)DPH_POOL_CHUNKS;
    if (blockIdx.x == 0 && tid == 0 && (claimed > (unsigned)DPH_POOL_CHUNKS || counters[2] != 0)) atomicOr(fail, 1u);
    const unsigned per = (used + gridDim.x - 1) / gridDim.x;
    const unsigned lo = blockIdx.x * per, hi = lo + per < used ? lo + per : used;
    for (int i = tid; i < n_q_total; i += CB_THREADS) cnt[i] = 0;
    if (tid == 0) total = 0;
    __syncthreads();
    unsigned mine = 0;
    for (unsigned ch = lo + wave; ch < hi; ch += n_waves) {
        const unsigned raw = chunk_fill[ch];
        const unsigned n = raw < (unsigned)DPH_CHUNK_PAIRS ? raw : (unsigned)DPH_CHUNK_PAIRS;
        mine += n;
#pragma unroll
        for (unsigned i0 = 0; i0 < (unsigned)DPH_CHUNK_PAIRS; i0 += 64u)
            if (i0 + lane < n) atomicAdd(&cnt[q_base + (pairs[(size_t)ch * DPH_CHUNK_PAIRS + i0 + lane].x >> 20)], 1u);
    }
    if (lane == 0 && mine) atomicAdd(&total, mine);
    __syncthreads();
    for (int i = tid; i < n_q_total; i += CB_THREADS) { base[i] = cnt[i] ? atomicAdd(&cand_cnt[i], cnt[i]) : 0u; cnt[i] = 0; }
    if (tid == 0 && total) atomicAdd(pool_count, total);
    __syncthreads();
    for (unsigned ch = lo + wave; ch < hi; ch += n_waves) {
        const unsigned raw = chunk_fill[ch];
        const unsigned n = raw < (unsigned)DPH_CHUNK_PAIRS ? raw : (unsigned)DPH_CHUNK_PAIRS;
#pragma unroll
        for (unsigned i0 = 0; i0 < (unsigned)DPH_CHUNK_PAIRS; i0 += 64u)
            if (i0 + lane < n) {
                const uint2 pr = pairs[(size_t)ch * DPH_CHUNK_PAIRS + i0 + lane];
                const unsigned q = q_base + (pr.x >> 20);
                const unsigned slot = base[q] + atomicAdd(&cnt[q], 1u);
                if (slot < (unsigned)cand_cap) cand_glob[(int64_t)q * cand_cap + slot] = make_uint2(pr.x & 0xFFFFFu, pr.y);
            }
    }
}

// threshold estimate of every row from its sample scores [n_q][m] (lists i * stride): the smallest sample rank r whose population count
// r * stride is >= target at -3 sigma
__global__ __launch_bounds__(CS_THREADS) void dph_coarse_estimate_sample_kernel(const float* __restrict__ sample_scores, int n_q, int m, int stride,
                                                                                int target, unsigned* __restrict__ est_out) {
    __shared__ unsigned hist[2048];
    __shared__ unsigned sh[32];
    __shared__ unsigned smp[CF_SAMPLE];
    const int qi = blockIdx.x;
    if (qi >= n_q) return;
    const int tid = threadIdx.x;
    for (int i = tid; i < m; i += CS_THREADS) smp[i] = f32_key(sample_scores[(int64_t)qi * m + i]);
    __syncthreads();
    int r = 1;
    while (((double)r - 3.0 * sqrt((double)r)) * (double)stride < (double)target && r < m) ++r;
    const unsigned est = cs_select_kth([&](int i) { return smp[i]; }, m, r, hist, sh);
    if (tid == 0) est_out[qi] = est;
}

// gate of the fail-over chain: all rows of the pass when any row (or the pool) failed, none otherwise
__global__ void dph_coarse_gate_kernel(const unsigned* __restrict__ fail, int n_q, int* __restrict__ gate_out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) gate_out[0] = fail[0] ? n_q : 0;
}

// Long score rows, first half of the one-pass path: dph_coarse_estimate_kernel takes the threshold estimate of every row from a
// strided sample, dph_coarse_collect_kernel spreads the pass over the row across several workgroups (one workgroup streams its
// 4 MiB row at HBM latency: 0.6 ms for 128 rows x 2^20 lists) that append the scores at or above the estimate to the row's
// candidate list in global memory; dph_coarse_select_kernel then works on the candidates only.
__global__ __launch_bounds__(CS_THREADS) void dph_coarse_estimate_kernel(const float* __restrict__ scores, int n_q_host,
                                                                         const int* __restrict__ gate, int gate_base, int nlist, int nprobe,
                                                                         unsigned* __restrict__ est_out) {
    __shared__ unsigned hist[2048];
    __shared__ unsigned sh[32];
    __shared__ unsigned smp[CS_SAMPLE];
    const int n_q = dph_gated_rows(gate, gate_base, n_q_host);
    const int qi = blockIdx.x;
    if (qi >= n_q) return;
    const int tid = threadIdx.x;
    const float* s = scores + (int64_t)qi * nlist;
    const int np = nprobe < nlist ? nprobe : nlist;
    const int stride = nlist / CS_SAMPLE > 0 ? nlist / CS_SAMPLE : 1;
    const int m = nlist / stride < CS_SAMPLE ? nlist / stride : CS_SAMPLE;
    for (int i = tid; i < m; i += CS_THREADS) smp[i] = f32_key(s[(int64_t)i * stride]);
    __syncthreads();
    int r = 1;                                       // smallest sample rank whose population count is >= np at -3 sigma
    while (((double)r - 3.0 * sqrt((double)r)) * (double)stride < (double)np && r < m) ++r;
    const unsigned est = cs_select_kth([&](int i) { return smp[i]; }, m, r, hist, sh);
    if (tid == 0) est_out[qi] = est;
}

__global__ __launch_bounds__(CS_THREADS) void dph_coarse_collect_kernel(const float* __restrict__ scores, int n_q_host,
                                                                        const int* __restrict__ gate, int gate_base, int nlist,
                                                                        uint2* __restrict__ cand_glob, unsigned* __restrict__ cand_cnt,
                                                                        const unsigned* __restrict__ est_in) {
    __shared__ unsigned sh[32];
    __shared__ unsigned smp[CS_SAMPLE];             // this workgroup's candidates (CS_SAMPLE / 2 pairs)
    const int n_q = dph_gated_rows(gate, gate_base, n_q_host);
    const int qi = blockIdx.x, slice = blockIdx.y, n_slices = gridDim.y;
    if (qi >= n_q) return;
    const int tid = threadIdx.x;
    const float* s = scores + (int64_t)qi * nlist;
    const unsigned est = est_in[qi];
    // candidates go to LDS first and to the row's global list with ONE atomic per workgroup: thousands of returning atomics on
    // counters that share cache lines serialise in the L2 (measured: 0.9 ms for 128 rows)
    uint2* const loc = (uint2*)smp;
    constexpr unsigned LOC_CAP = CS_SAMPLE / 2;
    if (tid == 0) sh[4] = 0;
    __syncthreads();                                 // (also: everybody is done reading the sample)
    auto take = [&](float v, int i) {
        const unsigned k = f32_key(v);
        if (k >= est) { const unsigned slot = atomicAdd(&sh[4], 1u); if (slot < LOC_CAP) loc[slot] = make_uint2((unsigned)i, k); }
    };
    const int per = ((nlist + n_slices - 1) / n_slices + 3) & ~3;
    const int i_lo = slice * per, i_hi = min(nlist, i_lo + per);
    if ((nlist & 3) == 0 && (((uintptr_t)s) & 15) == 0) {
        const float4* s4 = (const float4*)s;
        const int a4 = i_lo / 4, b4 = (i_hi > i_lo ? i_hi : i_lo) / 4;        // i_lo is a multiple of 4, i_hi too (nlist and per are)
        for (int i0 = a4; i0 < b4; i0 += 4 * CS_THREADS) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i0 + u * CS_THREADS + tid; v[u] = i < b4 ? s4[i] : make_float4(-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f); }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * CS_THREADS + tid;
                if (i < b4) { take(v[u].x, 4 * i); take(v[u].y, 4 * i + 1); take(v[u].z, 4 * i + 2); take(v[u].w, 4 * i + 3); }
            }
        }
    } else {
        for (int i = i_lo + tid; i < i_hi; i += CS_THREADS) take(s[i], i);
    }
    __syncthreads();
    const unsigned n_loc = sh[4];
    if (tid == 0) sh[5] = atomicAdd(&cand_cnt[qi], n_loc > LOC_CAP ? (unsigned)(2 * CS_CAND) : n_loc);      // too many here: the row falls back
    __syncthreads();
    const unsigned base = sh[5];
    if (n_loc <= LOC_CAP)
        for (unsigned i = tid; i < n_loc; i += CS_THREADS) if (base + i < CS_CAND) cand_glob[(int64_t)qi * CS_CAND + base + i] = loc[i];
}

// debugging (dph_debug_pq_phases, which = 1): 100 MHz stamps of dph_coarse_select_kernel per query row of its last launch with rows --
// start, query norm, candidates in LDS, nprobe-th candidate, marking, float64 band dots, band ranks; [7] = band | candidates << 16 | need << 40
__device__ unsigned long long dph_cs_clock[DPH_PASS_MAX * 8];
__device__ int dph_cs_clock_on;
int dph_coarse_select_clock(unsigned long long* out, int cap_rows) {
    const int on = 1;                                    // (armed by every call, read or not)
    if (hipMemcpyToSymbol(HIP_SYMBOL(dph_cs_clock_on), &on, sizeof on) != hipSuccess) return -1;
    const int n = cap_rows < DPH_PASS_MAX ? cap_rows : DPH_PASS_MAX;
    if (out && n > 0 && hipMemcpyFromSymbol(out, HIP_SYMBOL(dph_cs_clock), (size_t)n * 64) != hipSuccess) return -1;
    return DPH_PASS_MAX;
}

// The nprobe lists of a query row.  Short lists of scores (nlist < CS_FAST_MIN): radix select over the whole score row
// (three passes) + one marking pass.  Long ones (the reference's 2^20 lists): ONE pass -- a strided sample of CS_SAMPLE
// scores gives a threshold estimate that ~8 x nprobe scores beat (an order statistic of the sample: the count above it is
// nprobe .. CS_CAND with overwhelming probability), one pass collects those candidates into LDS, and select + marking run on
// the candidates; whenever the estimate turns out too high, too low, or closer to the nprobe-th score than the fp32 error
// band, the row falls back to the full passes.  Either way the lists inside the band are re-ranked in float64.
__global__ __launch_bounds__(CS_THREADS) void dph_coarse_select_kernel(
    const float* __restrict__ x, int q0, int n_q_host, const int* __restrict__ gate, int gate_base,
    const float* __restrict__ centroids, const float* __restrict__ scores, int nlist, int nprobe, double cnorm_max,
    unsigned* __restrict__ listmask, int mask_words, int* __restrict__ probe_out, int probe_stride, double err_rel,
    const uint2* __restrict__ cand_glob, const unsigned* __restrict__ cand_cnt, const unsigned* __restrict__ est_in,
    unsigned* __restrict__ fail_out, unsigned* __restrict__ row_fail, const float* __restrict__ cnorm = nullptr, double cnorm_cap = 0.0) {
    __shared__ unsigned hist[2048];
    __shared__ float q_lds[DPH_DIM];
    __shared__ int band_id[CS_BAND_CAP];
    __shared__ double band_s[CS_BAND_CAP];
    __shared__ unsigned sh[32];
    __shared__ double qn_sh[CS_THREADS / 64];
    extern __shared__ unsigned cs_dyn[];             // [CS_CAND] candidate ids | [CS_CAND] keys
    const int n_q = dph_gated_rows(gate, gate_base, n_q_host);
    const int qi = blockIdx.x;
    if (qi >= n_q) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool clk = tid == 0 && dph_cs_clock_on != 0;
    auto stamp = [&](int slot) __attribute__((always_inline)) { if (clk) dph_cs_clock[qi * 8 + slot] = wall_clock64(); };
    stamp(0);
    const float* s = scores + (int64_t)qi * nlist;
    const int np = nprobe < nlist ? nprobe : nlist;
    double qn = 0.0;
    for (int j = tid; j < DPH_DIM; j += CS_THREADS) { const float v = x[(int64_t)(q0 + qi) * DPH_DIM + j]; q_lds[j] = v; qn += (double)v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) qn += __shfl_xor(qn, o);
    if (lane == 0) qn_sh[wv] = qn;
    __syncthreads();
    double qnorm = 0.0;
    for (int w = 0; w < CS_THREADS / 64; ++w) qnorm += qn_sh[w];
    // |MFMA dot - exact| <= err_rel * ||x|| * max||c|| (f32-in: 768 * 2^-24 * sum|x_j c_j|; bf16x3 and the one-product filter: see
    // the kernels; the callers fold their slack into err_rel)
    // HEAVY lists (round 6).  The bound is per list: err_rel * ||x|| * ||c_l||.  A quantizer trained on token vectors has a few
    // centroids several times longer than the rest -- and under inner product exactly those crowd the top of every score row.  With
    // max||c|| for every list the band grew with the ONE longest centroid (x 2.4: the band overflowed and every pass failed over to
    // the bf16x3 chain, 1.3 ms per 128 rows).  So: lists with ||c_l|| > cnorm_cap (the caller's choice, ~1.1 x the 99-th percentile)
    // are heavy; a heavy CANDIDATE gets its exact float64 score at once (a few dozen 3 KiB rows per query row) in place of its
    // estimate -- error ~0 <= delta -- and the band logic below runs with delta from cnorm_cap.  Heavy lists OUTSIDE the candidate
    // set are covered by the pool test: est + delta_max must stay below the band (`fast`).
    const bool heavy_on = cnorm != nullptr && cnorm_cap > 0.0 && cnorm_cap < cnorm_max;
    const float delta = (float)(err_rel * sqrt(qnorm) * (heavy_on ? cnorm_cap : cnorm_max)) + 1e-30f;
    const float delta_max = (float)(err_rel * sqrt(qnorm) * cnorm_max) + 1e-30f;
    stamp(1);
    const unsigned word = (unsigned)qi >> 5, bitv = 1u << (qi & 31);
    int* const cand_id = (int*)cs_dyn;
    unsigned* const cand_key = cs_dyn + CS_CAND;
    bool fast = false;
    float t = 0.f;
    int n_cand = 0;
    if (nlist >= CS_FAST_MIN && cand_glob) {
        // ---- candidates collected by dph_coarse_collect_kernel (every score at or above its sampled estimate)
        const unsigned est = est_in[qi];
        n_cand = (int)cand_cnt[qi];
        if (n_cand <= CS_CAND)
            for (int i = tid; i < n_cand; i += CS_THREADS) { const uint2 c = cand_glob[(int64_t)qi * CS_CAND + i]; cand_id[i] = (int)c.x; cand_key[i] = c.y; }
        __syncthreads();
        bool heavy_ok = true;
        if (heavy_on && n_cand <= CS_CAND) {
            // exact scores for the heavy candidates (their positions collected in band_id, CS_BROWS rows per wave in flight)
            if (tid == 0) sh[6] = 0;
            __syncthreads();
            for (int i = tid; i < n_cand; i += CS_THREADS)
                if (cnorm[cand_id[i]] > (float)cnorm_cap) { const unsigned b = atomicAdd(&sh[6], 1u); if (b < CS_BAND_CAP) band_id[b] = i; }
            __syncthreads();
            const int nh = (int)sh[6];
            heavy_ok = nh <= CS_BAND_CAP;
            for (int b0 = wv * CS_BROWS; heavy_ok && b0 < nh; b0 += (CS_THREADS / 64) * CS_BROWS) {
                float cv[CS_BROWS][12];
#pragma unroll
                for (int u = 0; u < CS_BROWS; ++u) {
                    const int b = b0 + u < nh ? b0 + u : nh - 1;
                    const float4* cp = (const float4*)(centroids + (int64_t)cand_id[band_id[b]] * DPH_DIM + lane * 12);
                    const float4 c0 = cp[0], c1 = cp[1], c2 = cp[2];
                    cv[u][0] = c0.x; cv[u][1] = c0.y; cv[u][2] = c0.z; cv[u][3] = c0.w; cv[u][4] = c1.x; cv[u][5] = c1.y; cv[u][6] = c1.z; cv[u][7] = c1.w;
                    cv[u][8] = c2.x; cv[u][9] = c2.y; cv[u][10] = c2.z; cv[u][11] = c2.w;
                }
#pragma unroll
                for (int u = 0; u < CS_BROWS; ++u) {
                    double acc = 0.0;
#pragma unroll
                    for (int j = 0; j < 12; ++j) acc += (double)q_lds[lane * 12 + j] * (double)cv[u][j];
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
                    if (lane == 0 && b0 + u < nh) cand_key[band_id[b0 + u]] = f32_key((float)acc);
                }
            }
            __syncthreads();
        }
        stamp(2);
        if (heavy_ok && n_cand >= np && n_cand <= CS_CAND) {
            const unsigned kth = cs_select_kth([&](int i) { return cand_key[i]; }, n_cand, np, hist, sh);
            t = key_f32(kth);
            // the whole error band lies inside the candidate set: a list outside it has estimate < est, hence score < est + delta_max,
            // and the true nprobe-th score is >= t - delta
            fast = (t - delta - delta_max) >= key_f32(est);
        }
    }
    if (!fast && !scores) {                          // filter form: there is no score matrix to fall back on -- the pass fails over
        if (tid == 0 && fail_out) atomicOr(fail_out, 1u);
        return;
    }
    if (!fast) t = key_f32(cs_select_kth([&](int i) { return f32_key(s[i]); }, nlist, np, hist, sh));
    const float hi = t + 2.f * delta, lo = t - 2.f * delta;
    stamp(3);
    // ---- lists clearly above the band are probed; the band is collected for the fp64 re-rank
    if (tid == 0) { sh[2] = 0; sh[3] = 0; sh[5] = 0; }
    __syncthreads();
    if (probe_out) {
        for (int i = tid; i < probe_stride; i += CS_THREADS) probe_out[(int64_t)qi * probe_stride + i] = -1;
        __syncthreads();                              // (block-uniform branch) the fills land before any real entry
    }
    unsigned n_in = 0;
    const int n_mark = fast ? n_cand : nlist;
    for (int i = tid; i < n_mark; i += CS_THREADS) {
        const int l = fast ? cand_id[i] : i;
        const float v = fast ? key_f32(cand_key[i]) : s[i];
        if (v > hi) {
            if (listmask) atomicOr(&listmask[(int64_t)l * mask_words + word], bitv);
            ++n_in;
            if (probe_out) { const unsigned o = atomicAdd(&sh[5], 1u); if ((int)o < probe_stride) probe_out[(int64_t)qi * probe_stride + o] = l; }
        }
        else if (v >= lo) { const unsigned b = atomicAdd(&sh[3], 1u); if (b < CS_BAND_CAP) band_id[b] = l; }
    }
    atomicAdd(&sh[2], n_in);
    __syncthreads();
    const int nb = (int)(sh[3] < (unsigned)CS_BAND_CAP ? sh[3] : (unsigned)CS_BAND_CAP);
    if (sh[3] > (unsigned)CS_BAND_CAP) {        // more lists inside the error band than the re-rank holds
        // filter form: the pass fails over to the chain with the narrow band.  There (thousands of centroids within 10^-4 of the
        // nprobe-th score: copies of one vector) the row is flagged -- the caller reports it like an overflowed candidate buffer --
        // instead of being answered from a truncated band; without a flag array (flat-IVF callers) the band is cut as before
        if (fail_out) { if (tid == 0) atomicOr(fail_out, 1u); return; }
        if (row_fail && tid == 0) row_fail[qi] = 1u;
    }
    int need = np - (int)sh[2];                  // band lists still to probe
    stamp(4);
    if (clk) dph_cs_clock[qi * 8 + 7] = (unsigned long long)nb | ((unsigned long long)n_cand << 16) | ((unsigned long long)(need > 0 ? need : 0) << 40);
    if (need <= 0 || nb == 0) return;
    // float64 dot of every band list: CS_BROWS lists per wave and trip, their 3 KiB rows in flight together (the one-product filter
    // leaves a few hundred lists in the band; one row at a time was 2 us of latency each)
    for (int b0 = wv * CS_BROWS; b0 < nb; b0 += (CS_THREADS / 64) * CS_BROWS) {
        float cv[CS_BROWS][12];
#pragma unroll
        for (int u = 0; u < CS_BROWS; ++u) {
            const int b = b0 + u < nb ? b0 + u : nb - 1;
            const float4* cp = (const float4*)(centroids + (int64_t)band_id[b] * DPH_DIM + lane * 12);
            const float4 c0 = cp[0], c1 = cp[1], c2 = cp[2];
            cv[u][0] = c0.x; cv[u][1] = c0.y; cv[u][2] = c0.z; cv[u][3] = c0.w; cv[u][4] = c1.x; cv[u][5] = c1.y; cv[u][6] = c1.z; cv[u][7] = c1.w;
            cv[u][8] = c2.x; cv[u][9] = c2.y; cv[u][10] = c2.z; cv[u][11] = c2.w;
        }
#pragma unroll
        for (int u = 0; u < CS_BROWS; ++u) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < 12; ++j) acc += (double)q_lds[lane * 12 + j] * (double)cv[u][j];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
            if (lane == 0 && b0 + u < nb) band_s[b0 + u] = acc;
        }
    }
    __syncthreads();
    stamp(5);
    // the `need` best of the band: bitonic sort of (score desc, list id asc) in LDS -- 55 barrier steps for a band of 535 .. 1024.
    // (Counting, for every list, the lists ahead of it was 535 x 535 float64 compares per row: 23-40 us.)
    int pow2 = 1;
    while (pow2 < nb) pow2 <<= 1;                    // <= CS_BAND_CAP
    for (int i = nb + tid; i < pow2; i += CS_THREADS) { band_s[i] = -__builtin_huge_val(); band_id[i] = 0x7FFFFFFF; }
    __syncthreads();
    for (int size = 2; size <= pow2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < pow2 / 2; i += CS_THREADS) {
                const int lo_ = 2 * i - (i & (stride - 1)), hi_ = lo_ + stride;
                const bool up = (lo_ & size) == 0;
                const double sa = band_s[lo_], sb = band_s[hi_];
                const int ia = band_id[lo_], ib = band_id[hi_];
                const bool a_first = (sa > sb) | ((sa == sb) & (ia < ib));
                if (a_first != up) { band_s[lo_] = sb; band_s[hi_] = sa; band_id[lo_] = ib; band_id[hi_] = ia; }
            }
            __syncthreads();
        }
    const int take = need < nb ? need : nb;
    for (int b = tid; b < take; b += CS_THREADS) {
        const int id = band_id[b];
        if (listmask) atomicOr(&listmask[(int64_t)id * mask_words + word], bitv);
        if (probe_out) { const unsigned o = atomicAdd(&sh[5], 1u); if ((int)o < probe_stride) probe_out[(int64_t)qi * probe_stride + o] = id; }
    }
    if (dph_cs_clock_on) { __syncthreads(); stamp(6); }
}

__global__ __launch_bounds__(256) void dph_tilemask_kernel(const int32_t* __restrict__ tile_list, int64_t n_tiles,
                                                           const uint4* __restrict__ listmask, uint4* __restrict__ tilemask,
                                                           const int* __restrict__ gate, int gate_base) {
    if (dph_gated_rows(gate, gate_base, 1) <= 0) return;          // gated retry pass with nothing to do
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t < n_tiles) {
        const int64_t l = tile_list[t];
        tilemask[2 * t] = listmask[2 * l];
        tilemask[2 * t + 1] = listmask[2 * l + 1];
    }
}

void dph_launch_coarse(const float* x_dev, int q0, int n_q, const int* gate, int gate_base, const float* centroids, int nlist,
                       int nprobe, double cnorm_max, float* scores, unsigned* listmask, int mask_words, const int32_t* tile_list,
                       int64_t n_tiles, unsigned* tilemask, void** cs_slot, hipStream_t st) {
    dph_launch_coarse_lists(x_dev, q0, n_q, gate, gate_base, centroids, nlist, nprobe, cnorm_max, scores, listmask, mask_words,
                            tile_list, n_tiles, tilemask, nullptr, 0, cs_slot, st);
}

// the same, and additionally the probed lists of every query row as a list: probe_out [n_q][probe_stride] (-1 padded, order
// unspecified) -- what a scan that groups its work by query row walks (dph_pq.hip, many short lists)
void dph_launch_coarse_lists(const float* x_dev, int q0, int n_q, const int* gate, int gate_base, const float* centroids, int nlist,
                             int nprobe, double cnorm_max, float* scores, unsigned* listmask, int mask_words,
                             const int32_t* tile_list, int64_t n_tiles, unsigned* tilemask, int* probe_out, int probe_stride,
                             void** cs_slot, hipStream_t st) {
    dph_launch_coarse_presplit(x_dev, q0, n_q, gate, gate_base, centroids, nlist, nprobe, cnorm_max, scores, listmask, mask_words,
                               tile_list, n_tiles, tilemask, probe_out, probe_stride, nullptr, nullptr, cs_slot, st, true, nullptr);
}

// ... and with the packed bf16 hi / lo images (dph_launch_bf16_split) of the centroids [nlist,768] and of the query rows
// [>= q0 + n_q, 768] when the caller keeps them (dph_pq.hip): the long-quantizer GEMM then unpacks instead of splitting
void dph_launch_coarse_presplit(const float* x_dev, int q0, int n_q, const int* gate, int gate_base, const float* centroids, int nlist,
                                int nprobe, double cnorm_max, float* scores, unsigned* listmask, int mask_words,
                                const int32_t* tile_list, int64_t n_tiles, unsigned* tilemask, int* probe_out, int probe_stride,
                                const unsigned* c_pk, const unsigned* x_pk, void** cs_slot, hipStream_t st, bool clear_mask,
                                unsigned* row_fail) {
    if (clear_mask && listmask) (void)hipMemsetAsync(listmask, 0, (size_t)nlist * mask_words * 4, st);
    const bool bf16x3 = nlist >= CG_BF16X3_MIN;
    const dim3 gg((nlist + CG_LISTS - 1) / CG_LISTS, (n_q + CG_QROWS - 1) / CG_QROWS);
    if (bf16x3 && c_pk && x_pk) {
        const size_t lds3 = (size_t)4 * CG_LISTS * CG3P_LD * 2;
        static std::atomic<bool> attr3[64];      // (statics: zero-initialised; handles are searched from several threads)
        int dev3 = 0;
        (void)hipGetDevice(&dev3);
        if (dev3 < 0 || dev3 >= 64 || !attr3[dev3]) {
            const hipError_t e = hipFuncSetAttribute((const void*)dph_coarse_gemm_bf16x3_pipe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3);
            if (e != hipSuccess) fprintf(stderr, "libdph: hipFuncSetAttribute(dph_coarse_gemm_bf16x3_pipe_kernel, %zu B of LDS): %s\n", lds3, hipGetErrorString(e));
            if (dev3 >= 0 && dev3 < 64) attr3[dev3] = e == hipSuccess;
        }
        hipLaunchKernelGGL(dph_coarse_gemm_bf16x3_pipe_kernel, gg, dim3(256), lds3, st, q0, n_q, gate, gate_base, nlist, scores, c_pk, x_pk);
    }
    else if (bf16x3)
        hipLaunchKernelGGL(dph_coarse_gemm_bf16x3_kernel<false>, gg, dim3(256), 0, st, x_dev, q0, n_q, gate, gate_base, centroids, nlist, scores,
                           (const unsigned*)nullptr, (const unsigned*)nullptr);
    else
        hipLaunchKernelGGL(dph_coarse_gemm_kernel, dim3((nlist + CG_LISTS - 1) / CG_LISTS, (n_q + CG_QROWS - 1) / CG_QROWS), dim3(256), 0,
                           st, x_dev, q0, n_q, gate, gate_base, centroids, nlist, scores);
    // long score rows: candidates collected by several workgroups per row first (scratch: one allocation per index handle, sized
    // for DPH_PASS_MAX rows, in the caller's slot -- two handles searched from two threads never share it)
    uint2* cand_glob = nullptr; unsigned* cand_cnt = nullptr; unsigned* est = nullptr;
    if (nlist >= CS_FAST_MIN && n_q <= DPH_PASS_MAX && cs_slot) {
        const size_t cand_bytes = (size_t)DPH_PASS_MAX * CS_CAND * sizeof(uint2);
        if (!*cs_slot && hipMalloc(cs_slot, cand_bytes + 2 * DPH_PASS_MAX * 4) != hipSuccess) { *cs_slot = nullptr; (void)hipGetLastError(); }
        if (*cs_slot) {
            cand_glob = (uint2*)*cs_slot;
            cand_cnt = (unsigned*)((char*)*cs_slot + cand_bytes);
            est = cand_cnt + DPH_PASS_MAX;
            (void)hipMemsetAsync(cand_cnt, 0, (size_t)n_q * 4, st);
            int slices = 2048 / (n_q > 0 ? n_q : 1);
            slices = slices < 1 ? 1 : (slices > 16 ? 16 : slices);
            hipLaunchKernelGGL(dph_coarse_estimate_kernel, dim3(n_q), dim3(CS_THREADS), 0, st, scores, n_q, gate, gate_base, nlist, nprobe, est);
            hipLaunchKernelGGL(dph_coarse_collect_kernel, dim3(n_q, slices), dim3(CS_THREADS), 0, st, scores, n_q, gate, gate_base, nlist,
                               cand_glob, cand_cnt, est);
        }
    }
    const size_t cs_lds = (size_t)2 * CS_CAND * 4;
    static std::atomic<bool> cs_attr[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !cs_attr[dev]) {      // static + dynamic LDS of the select kernel exceed 64 KiB
        const hipError_t e = hipFuncSetAttribute((const void*)dph_coarse_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)cs_lds);
        if (e != hipSuccess) fprintf(stderr, "libdph: hipFuncSetAttribute(dph_coarse_select_kernel, %zu B of LDS): %s\n", cs_lds, hipGetErrorString(e));
        if (dev >= 0 && dev < 64) cs_attr[dev] = e == hipSuccess;
    }
    hipLaunchKernelGGL(dph_coarse_select_kernel, dim3(n_q), dim3(CS_THREADS), cs_lds, st, x_dev, q0, n_q, gate, gate_base, centroids,
                       scores, nlist, nprobe, cnorm_max, listmask, mask_words, probe_out, probe_stride,
                       1.5 * (bf16x3 ? CG_BF16X3_ERR : CG_F32_ERR), cand_glob, cand_cnt, est, (unsigned*)nullptr, row_fail);
    if (tilemask && mask_words == 8)
    hipLaunchKernelGGL(dph_tilemask_kernel, dim3((unsigned)((n_tiles + 255) / 256)), dim3(256), 0, st, tile_list, n_tiles,
                       (const uint4*)listmask, (uint4*)tilemask, gate, gate_base);
}

// The filter form for long quantizers (comment above dph_coarse_filter_gemm_kernel).  x_dev: the (rotated) query rows of the pass
// [n_q][768] fp32, x_hi / c_hi their and the centroids' bf16 images, x_pk / c_pk the packed hi|lo images the fail-over chain multiplies.
// Scratch in *cf_slot (one allocation per index handle).
void dph_launch_coarse_filter(const float* x_dev, int n_q, const float* centroids, const unsigned short* c_hi, const unsigned short* x_hi,
                              const unsigned* c_pk, const unsigned* x_pk, int nlist, int nprobe, double cnorm_max, float* scores,
                              unsigned* listmask, int mask_words, int* probe_out, int probe_stride, void** cs_slot, void** cf_slot,
                              hipStream_t st, dph_event_source prof, unsigned* row_fail, int variant,
                              const unsigned short* c_frag, const unsigned short* c_pieces, const float* cnorm, double cnorm_cap, int coarse_teams) {
    const int m = nlist < CF_SAMPLE ? nlist : CF_SAMPLE;
    const int stride = nlist / m;
    const size_t b_sample = (size_t)DPH_PASS_MAX * CF_SAMPLE * 4, b_pool_lk = (size_t)DPH_PASS_MAX * CS_CAND * 8,
                 b_pool_q = (size_t)DPH_PASS_MAX * CS_CAND * 2, b_cand = (size_t)DPH_PASS_MAX * CS_CAND * 8, b_small = (size_t)(2 * DPH_PASS_MAX + 16) * 4;
    // the scan form's own scratch behind the rest: pair pool, chunk fill counts, per-wave statistics, queue counters, query fragments
    int cus_scan = 256;
    {
        int d = 0;
        (void)hipGetDevice(&d);
        (void)hipDeviceGetAttribute(&cus_scan, hipDeviceAttributeMultiprocessorCount, d);
        if (cus_scan <= 0 || cus_scan > 256) cus_scan = 256;
    }
    const size_t b_pairs = (size_t)DPH_POOL_CHUNKS * DPH_CHUNK_PAIRS * 8, b_fill = (size_t)DPH_POOL_CHUNKS * 4, b_wc = (size_t)256 * 4 * 2 * 4,
                 b_cnt = 256, b_qfrag = (size_t)(DPH_PASS_MAX / DPH_QROWS) * 4 * 2 * DPH_QGROUP_FRAG_BYTES;      // (the fragments of a whole pass: the teams form)
    // (always reserved, whether or not the first call brings the piece image: a later call that does must find it -- ADVICE r5)
    const size_t b_scan = b_pairs + b_fill + b_wc + b_cnt + b_qfrag;
    if (!*cf_slot && hipMalloc(cf_slot, b_sample + b_pool_lk + b_pool_q + b_cand + b_small + 256 + b_scan) != hipSuccess) { *cf_slot = nullptr; (void)hipGetLastError(); }
    if (!*cf_slot || n_q > DPH_PASS_MAX) {          // no scratch: the bf16x3 chain alone
        dph_launch_coarse_presplit(x_dev, 0, n_q, nullptr, 0, centroids, nlist, nprobe, cnorm_max, scores, listmask, mask_words, nullptr, 0, nullptr,
                                   probe_out, probe_stride, c_pk, x_pk, cs_slot, st, true, row_fail);
        return;
    }
    char* base = (char*)*cf_slot;
    float* sample = (float*)base;
    uint2* pool_lk = (uint2*)(base + b_sample);
    unsigned short* pool_q = (unsigned short*)(base + b_sample + b_pool_lk);
    uint2* cand = (uint2*)(base + b_sample + b_pool_lk + b_pool_q);
    unsigned* small = (unsigned*)(base + b_sample + b_pool_lk + b_pool_q + b_cand);
    unsigned* cand_cnt = small;                          // [DPH_PASS_MAX]
    unsigned* est = small + DPH_PASS_MAX;                // [DPH_PASS_MAX]
    unsigned* pool_count = small + 2 * DPH_PASS_MAX;     // [1]
    unsigned* fail = pool_count + 1;                     // [1]
    int* gate = (int*)(pool_count + 2);                  // [1]
    const unsigned pool_cap = (unsigned)((size_t)n_q * CS_CAND);
    if (listmask) (void)hipMemsetAsync(listmask, 0, (size_t)nlist * mask_words * 4, st);       // (NULL: the caller walks probe_out only)
    (void)hipMemsetAsync(small, 0, b_small, st);
    const size_t lds128 = (size_t)2 * CG_LISTS * (CF_K + 8) * 2;
    const size_t lds_v3 = (size_t)2 * CG_QROWS * (CF_K + 8) * 2 + (size_t)CF_HIT_CAP * 10;
    static std::atomic<bool> attr[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !attr[dev]) {
        hipError_t e = hipFuncSetAttribute((const void*)dph_coarse_filter_gemm_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds128);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)dph_coarse_filter_gemm_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds128);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)dph_coarse_filter_gemm_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds128);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)dph_coarse_filter_gemm2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_v3);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)dph_coarse_select_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)2 * CS_CAND * 4));
        if (e != hipSuccess) fprintf(stderr, "libdph: hipFuncSetAttribute(coarse filter kernels): %s\n", hipGetErrorString(e));
        if (dev >= 0 && dev < 64) attr[dev] = e == hipSuccess;
    }
    const int qt = (n_q + CG_QROWS - 1) / CG_QROWS;
    const int np = nprobe < nlist ? nprobe : nlist;
    // enough candidates for the nprobe lists, the error band below them and the sampling noise; never more than the lists there are
    int target = 4 * np + 64;
    if (target > nlist) target = nlist;
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int tiles_s = (m + CG_LISTS - 1) / CG_LISTS, tiles_f = (nlist + CG_LISTS - 1) / CG_LISTS;
    // persistent: two workgroups per CU share the list tiles (fewer when the pass has several query tiles: grid.y)
    const int wg = std::max(1, 2 * cus / qt);
    hipLaunchKernelGGL(dph_coarse_filter_gemm_kernel<true>, dim3(std::min(tiles_s, wg), qt), dim3(256), lds128, st, n_q, m, stride, (int64_t)tiles_f, c_hi, x_hi,
                       sample, (const unsigned*)nullptr, (uint2*)nullptr, (unsigned short*)nullptr, (unsigned*)nullptr, 0u, (unsigned*)nullptr);
    hipLaunchKernelGGL(dph_coarse_estimate_sample_kernel, dim3(n_q), dim3(CS_THREADS), 0, st, sample, n_q, m, stride, target, est);
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    auto bracket_begin = [&]() { ev0 = ev1 = nullptr; if (prof.next) { prof.next(prof.ctx, &ev0, &ev1); (void)hipEventRecord(ev0, st); } };
    auto bracket_end = [&]() { if (ev1) (void)hipEventRecord(ev1, st); };
    if (!(variant == 5 && c_pieces)) bracket_begin();
    if (variant == 5 && c_pieces) {
        // 128 query rows per read of the centroid image (one group of 32 per scan wave); a pass of more rows reads it once per 128 --
        // 4 x 0.27 ms against the GEMM's 1.36 ms at 512 rows
        char* sb = base + ((b_sample + b_pool_lk + b_pool_q + b_cand + b_small + 255) / 256) * 256;
        uint2* pairs = (uint2*)sb;
        unsigned* chunk_fill = (unsigned*)(sb + b_pairs);
        unsigned* wave_counts = (unsigned*)(sb + b_pairs + b_fill);
        int* counters = (int*)(sb + b_pairs + b_fill + b_wc);
        uint4* qfrag = (uint4*)(sb + b_pairs + b_fill + b_wc + b_cnt);
        // more than 128 rows: ONE launch of teams (dph_scan.hip MODE 4) -- the groups of 128 rows share every tile through the XCDs' L2
        // instead of each reading the image from HBM.  Groups rounded up to a power of two that divides the 32 workgroups of an XCD.
        int n_groups = 1;
        while (n_groups * (int)DPH_QROWS < n_q) n_groups *= 2;
        static const int teams_env = getenv("DPH_CF_TEAMS") ? atoi(getenv("DPH_CF_TEAMS")) : -1;     // (-1: the handle's setting)
        const bool teams = (teams_env >= 0 ? teams_env != 0 : coarse_teams != 0) && n_groups >= 2 && n_groups <= 8 && cus_scan % (8 * n_groups) == 0;
        if (teams) {
            hipLaunchKernelGGL(dph_cf_qfrag_kernel, dim3((n_groups * 4 * 2 * 24 * 64 + 255) / 256), dim3(256), 0, st, x_hi, n_q, qfrag, n_groups);
            bracket_begin();
            dph_launch_coarse_scan_teams(c_pieces, nlist, qfrag, n_q, n_groups, est, pairs, chunk_fill, wave_counts, counters, cus_scan, st);
            bracket_end();
            hipLaunchKernelGGL(dph_coarse_bucket_chunks_kernel, dim3(64), dim3(CB_THREADS), 0, st, (const uint2*)pairs, (const unsigned*)chunk_fill, (const int*)counters,
                               n_q, 0u, cand, cand_cnt, (int)CS_CAND, pool_count, fail);
        }
        for (int q0 = 0; !teams && q0 < n_q; q0 += DPH_QROWS) {
            const int nq = std::min(n_q - q0, (int)DPH_QROWS);
            hipLaunchKernelGGL(dph_cf_qfrag_kernel, dim3((4 * 2 * 24 * 64 + 255) / 256), dim3(256), 0, st, x_hi + (int64_t)q0 * DPH_DIM, nq, qfrag);
            bracket_begin();
            dph_launch_coarse_scan(c_pieces, nlist, qfrag, nq, est + q0, pairs, chunk_fill, wave_counts, counters, cus_scan, st);
            bracket_end();
            // straight into the per-row candidate lists; DPH_CF_KEEP_POOL=1 (tools/debug_coarse_scan.py) writes the linear pool of the GEMM forms too
            static const bool keep_pool = getenv("DPH_CF_KEEP_POOL") && atoi(getenv("DPH_CF_KEEP_POOL")) != 0;
            if (keep_pool)
                hipLaunchKernelGGL(dph_cf_flatten_kernel, dim3(256), dim3(256), 0, st, (const uint2*)pairs, (const unsigned*)chunk_fill, (const int*)counters,
                                   pool_lk, pool_q, pool_count, pool_cap, fail, (unsigned)q0);
            hipLaunchKernelGGL(dph_coarse_bucket_chunks_kernel, dim3(64), dim3(CB_THREADS), 0, st, (const uint2*)pairs, (const unsigned*)chunk_fill, (const int*)counters,
                               n_q, (unsigned)q0, cand, cand_cnt, (int)CS_CAND, keep_pool ? fail + 2 : pool_count, fail);
        }
    } else if (variant >= 3 && c_frag)
        hipLaunchKernelGGL(dph_coarse_filter_gemm2_kernel, dim3(std::min((nlist + CF2_LISTS - 1) / CF2_LISTS, std::max(1, cus / qt)), qt), dim3(CF2_THREADS), lds_v3, st,
                           n_q, nlist, c_frag, x_hi, est, pool_lk, pool_q, pool_count, pool_cap, fail, variant == 4 ? 1 : 0);
    else if (variant >= 2)
        hipLaunchKernelGGL((dph_coarse_filter_gemm_kernel<false, true>), dim3(std::min(tiles_f, wg), qt), dim3(256), lds128, st, n_q, nlist, 1, (int64_t)tiles_f, c_hi, x_hi,
                           (float*)nullptr, est, pool_lk, pool_q, pool_count, pool_cap, fail);
    else
        hipLaunchKernelGGL((dph_coarse_filter_gemm_kernel<false, false>), dim3(std::min(tiles_f, wg), qt), dim3(256), lds128, st, n_q, nlist, 1, (int64_t)tiles_f, c_hi, x_hi,
                           (float*)nullptr, est, pool_lk, pool_q, pool_count, pool_cap, fail);
    if (!(variant == 5 && c_pieces)) bracket_end();
    if (!(variant == 5 && c_pieces))
        hipLaunchKernelGGL(dph_coarse_bucket_kernel, dim3(64), dim3(CB_THREADS), 0, st, pool_lk, pool_q, pool_count, pool_cap, n_q, cand, cand_cnt, (int)CS_CAND);
    hipLaunchKernelGGL(dph_coarse_select_kernel, dim3(n_q), dim3(CS_THREADS), (size_t)2 * CS_CAND * 4, st, x_dev, 0, n_q, (const int*)nullptr, 0, centroids,
                       (const float*)nullptr, nlist, nprobe, cnorm_max, listmask, mask_words, probe_out, probe_stride, 1.02 * CF_HI_ERR,
                       (const uint2*)cand, (const unsigned*)cand_cnt, (const unsigned*)est, fail, (unsigned*)nullptr, cnorm, cnorm_cap);
    hipLaunchKernelGGL(dph_coarse_gate_kernel, dim3(1), dim3(64), 0, st, fail, n_q, gate);
    // fail-over: the whole pass again through the bf16x3 chain, gated on the device (empty launches when nothing failed); it ORs into
    // the masks the filter form has set (a row that succeeded marks the same lists again) and rewrites the probe lists it covers
    dph_launch_coarse_presplit(x_dev, 0, n_q, gate, 0, centroids, nlist, nprobe, cnorm_max, scores, listmask, mask_words, nullptr, 0, nullptr,
                               probe_out, probe_stride, c_pk, x_pk, cs_slot, st, false, row_fail);
}

// what the filter form did in the last pass over this scratch: out[0] = 1 when the pass failed over to the bf16x3 chain, out[1] = triples
// its GEMM epilogue put into the pool (synchronises the device)
int dph_coarse_filter_debug(void* cf_slot, unsigned out[2]) {
    if (!cf_slot) { out[0] = out[1] = 0xFFFFFFFFu; return 0; }
    const size_t off = (size_t)DPH_PASS_MAX * CF_SAMPLE * 4 + (size_t)DPH_PASS_MAX * CS_CAND * 8 + (size_t)DPH_PASS_MAX * CS_CAND * 2 +
                       (size_t)DPH_PASS_MAX * CS_CAND * 8 + (size_t)2 * DPH_PASS_MAX * 4;
    unsigned v[2] = {0, 0};
    if (hipMemcpy(v, (char*)cf_slot + off, 8, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    out[0] = v[1];          // fail
    out[1] = v[0];          // pool_count
    return 0;
}

// the candidate pool the filter left behind: (list, key) and query row of up to `cap` triples; returns the pool count (-1: copy failed)
long long dph_coarse_filter_debug_pool(void* cf_slot, unsigned* lk_host, unsigned short* q_host, long long cap) {
    if (!cf_slot) return 0;
    const size_t o_lk = (size_t)DPH_PASS_MAX * CF_SAMPLE * 4, o_q = o_lk + (size_t)DPH_PASS_MAX * CS_CAND * 8,
                 o_cnt = o_q + (size_t)DPH_PASS_MAX * CS_CAND * 2 + (size_t)DPH_PASS_MAX * CS_CAND * 8 + (size_t)2 * DPH_PASS_MAX * 4;
    unsigned n = 0;
    if (hipMemcpy(&n, (char*)cf_slot + o_cnt, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    const long long m = (long long)n < cap ? (long long)n : cap;
    if (m > 0) {
        if (hipMemcpy(lk_host, (char*)cf_slot + o_lk, (size_t)m * 8, hipMemcpyDeviceToHost) != hipSuccess) return -1;
        if (hipMemcpy(q_host, (char*)cf_slot + o_q, (size_t)m * 2, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    }
    return (long long)n;
}

// ---- work queue of the unit scan (dph_internal.h: DPH_PASS_MAX).  One wave per inverted list: the probing query rows
// of the list (set bits of its DPH_UNIT_WORDS mask words, ascending) are dealt into chunks of 128 slots; every chunk
// gets a slot table, its gathered high-digit fragments (written by the kernel below) and one unit record per segment
// of DPH_UNIT_TILES tiles.  Chunks and units are numbered in list order (dph_units_offsets_kernel: an exclusive scan
// over the lists), i.e. in address order: consecutive pops of the scan's work queue stream neighbouring segments.
#define UO_THREADS 1024
__global__ __launch_bounds__(UO_THREADS) void dph_units_offsets_kernel(const unsigned* __restrict__ listmask, int nlist,
                                                                      const int* __restrict__ list_tile0,
                                                                      int2* __restrict__ offsets, int* __restrict__ counts) {
    __shared__ int2 wsum[UO_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per = (nlist + UO_THREADS - 1) / UO_THREADS;
    const int l0 = tid * per, l1 = min(nlist, l0 + per);
    int2 mine = make_int2(0, 0);                      // chunks, units of this thread's lists
    for (int l = l0; l < l1; ++l) {
        const uint4* m = (const uint4*)(listmask + (int64_t)l * DPH_UNIT_WORDS);
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < DPH_UNIT_WORDS / 4; ++i) {
            const uint4 v = m[i];
            cnt += __builtin_popcount(v.x) + __builtin_popcount(v.y) + __builtin_popcount(v.z) + __builtin_popcount(v.w);
        }
        const int ntiles = list_tile0[l + 1] - list_tile0[l];
        const int chunks = (cnt > 0 && ntiles > 0) ? (cnt + DPH_UNIT_SLOTS - 1) / DPH_UNIT_SLOTS : 0;
        mine.x += chunks;
        mine.y += chunks * ((ntiles + DPH_UNIT_TILES - 1) / DPH_UNIT_TILES);
    }
    int2 incl = mine;                                 // inclusive scan over the workgroup
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int vx = __shfl_up(incl.x, o), vy = __shfl_up(incl.y, o);
        if (lane >= o) { incl.x += vx; incl.y += vy; }
    }
    if (lane == 63) wsum[wv] = incl;
    __syncthreads();
    int2 base = make_int2(0, 0);
    for (int w = 0; w < wv; ++w) { base.x += wsum[w].x; base.y += wsum[w].y; }
    int2 run = make_int2(base.x + incl.x - mine.x, base.y + incl.y - mine.y);       // exclusive prefix of this thread
    for (int l = l0; l < l1; ++l) {
        // (recomputed: cheaper than keeping `per` values in registers)
        const uint4* m = (const uint4*)(listmask + (int64_t)l * DPH_UNIT_WORDS);
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < DPH_UNIT_WORDS / 4; ++i) {
            const uint4 v = m[i];
            cnt += __builtin_popcount(v.x) + __builtin_popcount(v.y) + __builtin_popcount(v.z) + __builtin_popcount(v.w);
        }
        const int ntiles = list_tile0[l + 1] - list_tile0[l];
        const int chunks = (cnt > 0 && ntiles > 0) ? (cnt + DPH_UNIT_SLOTS - 1) / DPH_UNIT_SLOTS : 0;
        offsets[l] = run;
        run.x += chunks;
        run.y += chunks * ((ntiles + DPH_UNIT_TILES - 1) / DPH_UNIT_TILES);
    }
    if (tid == UO_THREADS - 1) { counts[0] = run.x; counts[1] = run.y; }
}

__global__ __launch_bounds__(256) void dph_units_build_kernel(const unsigned* __restrict__ listmask, int nlist,
                                                              const int* __restrict__ list_tile0, int chunk_cap, int unit_cap,
                                                              const int2* __restrict__ offsets, int* __restrict__ counts,
                                                              int* __restrict__ slot_q, int4* __restrict__ unit_recs,
                                                              int4* __restrict__ list_recs, int spread) {
    const int l = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (l >= nlist) return;
    const unsigned w = lane < DPH_UNIT_WORDS ? listmask[(int64_t)l * DPH_UNIT_WORDS + lane] : 0u;
    const int pc = __builtin_popcount(w);
    int incl = pc;                                   // inclusive prefix sum over the lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    const int cnt = __shfl(incl, 63);
    const int t0 = list_tile0[l], ntiles = list_tile0[l + 1] - t0;
    if (cnt == 0 || ntiles <= 0) return;
    const int chunks = (cnt + DPH_UNIT_SLOTS - 1) / DPH_UNIT_SLOTS;
    const int segs = (ntiles + DPH_UNIT_TILES - 1) / DPH_UNIT_TILES;
    const int c0 = offsets[l].x, u0 = offsets[l].y;
    if (c0 + chunks > chunk_cap || u0 + chunks * segs > unit_cap) { if (lane == 0) atomicOr(&counts[2], 1); return; }   // cannot happen: caps are worst case
    // slot tables: the i-th probing row (ascending) goes to chunk i / 128
    int i = incl - pc;
    unsigned bits = w;
    while (bits) {
        const int j = __builtin_ctz(bits);
        bits &= bits - 1u;
        // dealt over the four column groups (= scan waves) first, so that a half-empty chunk still loads all four waves
        // with pairs: their pair regions fill evenly and so do the refine workgroups behind them
        const int within = i % DPH_UNIT_SLOTS;
        const int col = spread ? (within & 3) * DPH_QGROUP + (within >> 2) : within;
        slot_q[(int64_t)c0 * DPH_UNIT_SLOTS + (i - within) + col] = 32 * lane + j;
        ++i;
    }
    for (int e = cnt + lane; e < chunks * DPH_UNIT_SLOTS; e += 64) {
        const int within = e % DPH_UNIT_SLOTS;
        const int col = spread ? (within & 3) * DPH_QGROUP + (within >> 2) : within;
        slot_q[(int64_t)c0 * DPH_UNIT_SLOTS + (e - within) + col] = -1;
    }
    for (int c = lane; c < chunks; c += 64) list_recs[c0 + c] = make_int4(t0, t0, t0 + ntiles, c0 + c);
    for (int e = lane; e < chunks * segs; e += 64) {
        const int c = e / segs, sg = e % segs;
        const int first = t0 + sg * DPH_UNIT_TILES;
        const int end = min(t0 + ntiles, first + DPH_UNIT_TILES);
        unit_recs[u0 + e] = make_int4(t0, first, end, c0 + c);
    }
}

// fragment image of chunk c, group g (dph_quantize_kernel's order: [kstep][lane][16 B], lane = 32*half + column):
// the 16 bytes of lane (half, column) at k-step ks are bytes 32 ks + 16 half .. + 15 of the column's query row
__global__ __launch_bounds__(256) void dph_units_gather_kernel(const int* __restrict__ counts, const int* __restrict__ slot_q,
                                                               const int8_t* __restrict__ q1, int q0,
                                                               int8_t* __restrict__ frags, bool x16) {
    const int c = blockIdx.x, g = blockIdx.y;
    if (c >= counts[0]) return;
    const int lane = threadIdx.x & 63, kq = threadIdx.x >> 6;
    uint4* out = (uint4*)(frags + ((int64_t)c * 4 + g) * DPH_QGROUP_FRAG_BYTES);
    if (x16) {
        // the 16 x 16 x 64 order (dph_internal.h): entry 2 s + h, lane = 16 (k-chunk) + column of the query half h
#pragma unroll
        for (int i = 0; i < DPH_KSTEPS / 4; ++i) {
            const int e = kq + 4 * i, s_ = e >> 1, hq = e & 1;
            const int qrow = slot_q[(int64_t)c * DPH_UNIT_SLOTS + g * DPH_QGROUP + 16 * hq + (lane & 15)];
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (qrow >= 0) v = *(const uint4*)(q1 + (int64_t)(q0 + qrow) * DPH_DIM + 64 * s_ + 16 * (lane >> 4));
            out[e * 64 + lane] = v;
        }
        return;
    }
    const int qrow = slot_q[(int64_t)c * DPH_UNIT_SLOTS + g * DPH_QGROUP + (lane & 31)];
    const int8_t* src = q1 + (int64_t)(q0 + (qrow < 0 ? 0 : qrow)) * DPH_DIM + 16 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < DPH_KSTEPS / 4; ++i) {
        const int ks = kq + 4 * i;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (qrow >= 0) v = *(const uint4*)(src + 32 * ks);
        out[ks * 64 + lane] = v;
    }
}

void dph_launch_units_build(const unsigned* listmask, int nlist, const int* list_tile0, const int8_t* q1, int q0,
                            int chunk_cap, int unit_cap, int* unit_counts, int* unit_next, int* slot_q, int4* unit_recs,
                            int4* unit_list_recs, int8_t* unit_frags, int2* unit_offsets, int spread, hipStream_t st, bool x16) {
    // unit_counts[4] and unit_next[DPH_UNIT_LAUNCHES] are one allocation
    (void)unit_next;
    (void)hipMemsetAsync(unit_counts, 0, (size_t)(4 + DPH_UNIT_LAUNCHES) * sizeof(int), st);
    hipLaunchKernelGGL(dph_units_offsets_kernel, dim3(1), dim3(UO_THREADS), 0, st, listmask, nlist, list_tile0, unit_offsets,
                       unit_counts);
    hipLaunchKernelGGL(dph_units_build_kernel, dim3((nlist + 3) / 4), dim3(256), 0, st, listmask, nlist, list_tile0, chunk_cap,
                       unit_cap, unit_offsets, unit_counts, slot_q, unit_recs, unit_list_recs, spread);
    hipLaunchKernelGGL(dph_units_gather_kernel, dim3(chunk_cap, 4), dim3(256), 0, st, unit_counts, slot_q, q1, q0, unit_frags, x16);
}

// ---- list assignment of database rows for the list builder (replaces the add-to-index step of
// build_phrase_index.py:145-153 for the exact in-list variant): best[r] = arg-max_l <x_r, c_l> (+ an optional per-list
// bias: -||c||^2/2 turns it into the L2 assignment of a k-means step), ties to the lowest list id, plus the gap to the
// runner-up so the caller can re-check near-ties in float64.  FUSED: a workgroup owns 128 rows and walks all list tiles,
// keeping the running best two scores per row in registers -- the [n, nlist] score matrix (2.8 TB for 170 M rows x 4096
// lists) is never written.  Same MFMA tile as dph_coarse_gemm_kernel (v_mfma_f32_32x32x2_f32: exact fp32 products), the
// centroids are the streamed operand (L2-resident: 12.6 MB at 4096 lists).  ROWS_INT8: the rows are int8 rows of the
// resident shard, de-quantised through the shard's LUT while they are staged (the reference's fp32 values exactly).
template <bool ROWS_INT8>
__global__ __launch_bounds__(256) void dph_assign_fused_kernel(const void* __restrict__ rows, int64_t n,
                                                               const float* __restrict__ lut,
                                                               const float* __restrict__ centroids, int nlist,
                                                               const float* __restrict__ bias, int32_t* __restrict__ best,
                                                               float* __restrict__ gap) {
    __shared__ float a_lds[CG_LISTS * CG_LD];
    __shared__ float b_lds[CG_QROWS * CG_LD];
    __shared__ float red_s1[4][CG_QROWS], red_s2[4][CG_QROWS];
    __shared__ int red_i1[4][CG_QROWS];
    const int64_t r0 = (int64_t)blockIdx.x * CG_QROWS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float b1[4], b2[4];
    int i1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { b1[j] = -3.4e38f; b2[j] = -3.4e38f; i1[j] = 0x7fffffff; }
    for (int l0 = 0; l0 < nlist; l0 += CG_LISTS) {
        v16f acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        for (int k0 = 0; k0 < DPH_DIM; k0 += CG_KCHUNK) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = (tid >> 3) + 32 * i, c4 = (tid & 7) * 4;
                const int l = l0 + row;
                const int64_t q = r0 + row;
                const float4 av = l < nlist ? *(const float4*)(centroids + (int64_t)l * DPH_DIM + k0 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q < n) {
                    if constexpr (ROWS_INT8) {
                        const unsigned w = *(const unsigned*)((const int8_t*)rows + q * DPH_DIM + k0 + c4);
                        bv = make_float4(lut[(int)(int8_t)(w) + 128], lut[(int)(int8_t)(w >> 8) + 128],
                                         lut[(int)(int8_t)(w >> 16) + 128], lut[(int)(int8_t)(w >> 24) + 128]);
                    } else {
                        bv = *(const float4*)((const float*)rows + q * DPH_DIM + k0 + c4);
                    }
                }
                float* ap = a_lds + row * CG_LD + c4;
                float* bp = b_lds + row * CG_LD + c4;
                ap[0] = av.x; ap[1] = av.y; ap[2] = av.z; ap[3] = av.w;
                bp[0] = bv.x; bp[1] = bv.y; bp[2] = bv.z; bp[3] = bv.w;
            }
            __syncthreads();
            const float* ap = a_lds + (wave * 32 + (lane & 31)) * CG_LD + (lane >> 5);
            const float* bp = b_lds + (lane & 31) * CG_LD + (lane >> 5);
#pragma unroll
            for (int kk = 0; kk < CG_KCHUNK; kk += 2) {
                const float a = ap[kk];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[j * 32 * CG_LD + kk], acc[j], 0, 0, 0);
            }
            __syncthreads();
        }
        // C[i = (r&3) + 8*(r>>2) + 4*(lane>>5)][j = lane&31]: i = list inside the wave's 32 (ascending in r), j = row
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int l = l0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (l < nlist) {
                    const float v = acc[j][r] + (bias ? bias[l] : 0.f);
                    if (v > b1[j] || (v == b1[j] && l < i1[j])) { b2[j] = b1[j]; b1[j] = v; i1[j] = l; }
                    else if (v > b2[j]) b2[j] = v;
                }
            }
    }
    // merge the two lane halves (same row, other lists), then the four waves through LDS
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float ob1 = __shfl_xor(b1[j], 32), ob2 = __shfl_xor(b2[j], 32);
        const int oi1 = __shfl_xor(i1[j], 32);
        if (ob1 > b1[j] || (ob1 == b1[j] && oi1 < i1[j])) { b2[j] = fmaxf(b1[j], ob2); b1[j] = ob1; i1[j] = oi1; }
        else b2[j] = fmaxf(b2[j], ob1);
        if (lane < 32) { red_s1[wave][j * 32 + lane] = b1[j]; red_s2[wave][j * 32 + lane] = b2[j]; red_i1[wave][j * 32 + lane] = i1[j]; }
    }
    __syncthreads();
    if (tid < CG_QROWS) {
        float s1 = red_s1[0][tid], s2 = red_s2[0][tid];
        int i = red_i1[0][tid];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float o1 = red_s1[w][tid], o2 = red_s2[w][tid];
            const int oi = red_i1[w][tid];
            if (o1 > s1 || (o1 == s1 && oi < i)) { s2 = fmaxf(s1, o2); s1 = o1; i = oi; }
            else s2 = fmaxf(s2, o1);
        }
        const int64_t q = r0 + tid;
        if (q < n) { best[q] = i; gap[q] = s1 - s2; }
    }
}

void dph_launch_assign(const void* rows, bool rows_int8, const float* lut, int64_t n, const float* centroids, int nlist,
                       const float* bias, int32_t* best, float* gap, hipStream_t st) {
    const dim3 grid((unsigned)((n + CG_QROWS - 1) / CG_QROWS));
    if (rows_int8)
        hipLaunchKernelGGL(dph_assign_fused_kernel<true>, grid, dim3(256), 0, st, rows, n, lut, centroids, nlist, bias, best, gap);
    else
        hipLaunchKernelGGL(dph_assign_fused_kernel<false>, grid, dim3(256), 0, st, rows, n, lut, centroids, nlist, bias, best, gap);
}
