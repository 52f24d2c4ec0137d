// Weight-gradient GEMM of a Linear layer for gfx950 (bf16 MFMA, fp32 accumulate):
//     gW[n][k] = sum_r dY[r][n] X[r][k],    gb[n] = sum_r dY[r][n]          (autograd of y = x W^T + b, nb:1205)
// The contraction runs over the ROWS, the slow index of both operands, and the outputs are small (at most
// 672 x 256) while R is ~20k: split over row ranges, one fp32 slab per range (deterministic, summed by
// svae_reduce_slabs).  The generic LDS-tiled GEMM ran these at 0.7 TB/s: a block's k-loop is a serial chain
// of global load -> transpose -> LDS -> barrier -> 8 MFMAs with one tile in flight and 1.5 blocks per CU.
// Here the fp32 operand tiles go to LDS untouched by LDS-DMA (buffer_load ... lds: rows beyond R read as 0),
// three tiles in flight per block; a wave builds its bf16 fragments from the row-major tile with conflict-free
// column reads (lanes = consecutive columns) -- no staging registers, no transposing stores.
//
// The all-bf16 form (bf16 dY and bf16 X: the GRU's input-weight gradient) keeps both tiles as XOR-swizzled bf16 images
// and takes a fragment with two ds_read_b64_tr_b16 (swz_off / tr_lane_off below): 16 LDS instructions per wave and
// 32 rows instead of 128 two-byte reads and their packing.  dW_ih alone at 21.8k rows, back-to-back launches
// (tools/probe/wgrad_time.py): 30.6 -> 24.1 us, which is what an experiment build with one lane-linear 16-byte read
// per fragment (wrong values, same instruction mix otherwise) had given as the bound of this change (23.3 us).  Inside
// the step the kernel's region did not move (A/B of the whole step: 0.8756 ms both): it runs beside the weight
// sweep's finish kernels and the optimizer, and what they contend for is not its LDS reads.
// What did not carry over: (1) the same reads for dY alone with an fp32 X beside it (the mixed form, dW_hh) made that
// kernel slower, 30.7 -> 33.8 us, so it keeps row-major images and column reads; (2) the first attempt at transposing
// reads in round 2 used row-major, unswizzled images whose rows all start on bank 0 (4-way conflicts on every read)
// and gained nothing.  With no tile staged after the prologue (SV_WG_EXP=2) the old kernel ran 29.3 us of its 30.7:
// the operand bytes from L2 are not the bound either; the rest is the k-loop's per-stage latency and the slab stores.
#include "common.h"

namespace {

constexpr int WM = 128, WN = 128;   // output tile: WM dY-columns x WN X-columns
constexpr int WK = 16;              // rows per stage = one MFMA k-step
constexpr int NST = 4;              // LDS stages
constexpr int TILE_BYTES = WK * WM * 4;             // 8 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;         // 16 KiB
constexpr unsigned BUF_FLAGS = 0x00020000u;

struct WgradArgs {
    const float* dy;    // [R][ldy]
    const float* x;     // [R][ldx]
    float* out;         // slab z: out + z * slab_stride, [N][K] then (colsum) [N]
    float* colsum;      // out-relative pointer of slab 0's bias gradient, or nullptr
    int R, N, K, ldy, ldx;
    int nx, ny, nz, tiles_per_split;
    long long slab_stride;
    int remap_from, remap_add;   // dY columns >= remap_from are read remap_add columns further (both multiples of 4)
};

__device__ __forceinline__ unsigned pk2(float a, float b) {
    const bf16v2 h = __builtin_convertvector((f32x2){a, b}, bf16v2);
    return __builtin_bit_cast(unsigned, h);
}

// A bf16 operand tile in LDS is [rows][128 columns] with 256-byte rows whose 16-byte chunks are XOR-swizzled by the
// row: chunk ch of row r sits at 256 r + 16 (ch ^ (((r & 3) << 2) | ((r >> 2) & 3))).  The LDS-DMA that fills it is
// lane-linear, so the swizzle is applied on the GLOBAL side (a lane fetches the chunk that belongs at its LDS
// position); the four rows of a transposing read then fall into the four 64-byte quarters of the bank row and a
// 32-lane half (two 4 x 16 blocks side by side) is conflict-free.  Unswizzled, all rows start on bank 0: 4-way.
__device__ __forceinline__ unsigned swz_off(unsigned row, unsigned ch) {
    return 256u * row + 16u * (ch ^ (((row & 3u) << 2) | ((row >> 2) & 3u)));
}
typedef short s4v __attribute__((ext_vector_type(4)));
typedef short s8v __attribute__((ext_vector_type(8)));
// ds_read_b64_tr_b16: per 16-lane group a 4-row x 16-column block of 16-bit elements, lane 4q + p supplies the address
// of row q, columns 4p .. 4p + 3, lane i receives column i of the four rows.  EXEC is full at every call below.
__device__ __forceinline__ s4v tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)p);
}
// byte offset (inside a swizzled tile, k-step 0) this lane supplies for half h (rows 8 (g >> 1) + 4 h + q) of the
// fragment of the 32-column group that starts at column cb: two such reads are the lane's 8 k-values of column
// cb + (lane & 31), in the MFMA's own order.
__device__ __forceinline__ unsigned tr_lane_off(int lane, int cb, int h) {
    const unsigned g = (unsigned)lane >> 4, q = ((unsigned)lane >> 2) & 3u, pp = (unsigned)lane & 3u;
    return swz_off(8u * (g >> 1) + 4u * (unsigned)h + q, (unsigned)cb / 8u + 2u * (g & 1u) + (pp >> 1)) + 8u * (pp & 1u);
}

// ABF: dY holds bf16 elements (the GRU's merged gradient array): its tile is 16 rows x 128 columns x 2 B = 4 KiB, one
// LDS-DMA piece (4 rows) per wave, and a fragment is built from 2-byte reads without conversion.
// hi / mid / lo parts of 8 fp32 values as three bf16 fragments: hi = RNE(v), mid = RNE(v - hi), lo = RNE(v - hi - mid);
// together they carry all 24 significant bits of v.  The split instantiation (SPLIT) contracts
// hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi: fp32-class products on the bf16 pipe.
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
    unsigned hw[4], mw[4], lw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        hw[i] = pk2(v[2 * i], v[2 * i + 1]);
        const float r0 = v[2 * i] - __uint_as_float(hw[i] << 16), r1 = v[2 * i + 1] - __uint_as_float(hw[i] & 0xFFFF0000u);
        mw[i] = pk2(r0, r1);
        lw[i] = pk2(r0 - __uint_as_float(mw[i] << 16), r1 - __uint_as_float(mw[i] & 0xFFFF0000u));
    }
    hi = __builtin_bit_cast(bf16x8, (uint4){hw[0], hw[1], hw[2], hw[3]});
    mid = __builtin_bit_cast(bf16x8, (uint4){mw[0], mw[1], mw[2], mw[3]});
    lo = __builtin_bit_cast(bf16x8, (uint4){lw[0], lw[1], lw[2], lw[3]});
}

// SPLIT (fp32 dY only): the parity instantiation -- same staging, tiling and slab order, every product carried by
// the six leading terms of (hi + mid + lo) x (hi + mid + lo).
// XBF: X holds bf16 elements as well (the embedding rows as the gather wrote them: its consumers round to bf16 anyway,
// RNE there as here, so the gradient is bit-identical): the X tile is 4 KiB, one piece per wave, and the six
// re-reads of X from L2 that bound this kernel (header) are half the bytes.
// bid = the workgroup's index inside its job (the kernel's blockIdx.x, or that minus the first job's workgroups -- a
// multiple of 8, so that id % 8 stays the XCD -- in the two-job launch below)
template <bool ABF, bool SPLIT = false, bool XBF = false>
__device__ __forceinline__ void wgrad_body(const WgradArgs& a, const int bid, char* smem) {
    static_assert(!(ABF && SPLIT), "the split instantiation reads fp32 operands");
    static_assert(!XBF || ABF, "bf16 X comes with the bf16 dY");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    // blocks of one row split sit on one XCD (id % 8), see gemm.hip
    const int xcd = bid & 7, slot = bid >> 3;
    const int gsize = a.nx * a.ny, bz = xcd + 8 * (slot / gsize), mem = slot % gsize;
    if (bz >= a.nz) return;
    const int bx = mem % a.nx, by = mem / a.nx;
    const int m0 = by * WM, n0 = bx * WN;
    // Rows per stage.  With both operands bf16 a stage of 32 rows costs the LDS of a 16-row fp32 stage and the k-loop
    // (one wait, one barrier, fragment builds and 4 MFMAs per wave and 16 rows) makes half as many trips -- measured:
    // no difference (dW_ih 19.5 us either way), like halving the X bytes before it (-1.6 us) and quartering the LDS
    // instructions in round 2: none of the three is what bounds this kernel.  Kept, harmless.
    // a.tiles_per_split counts 16-row tiles; the splits' row boundaries are the same multiples of 32 rows in every
    // instantiation (wgrad_launch), so all of them add the same rows into the same slabs.
    constexpr int WKT = XBF ? 32 : WK;
    const int ntiles = (a.R + WKT - 1) / WKT;
    const int tps = a.tiles_per_split / (WKT / WK);
    const int t_begin = bz * tps;
    int t_end = t_begin + tps;
    if (t_end > ntiles) t_end = ntiles;

    const __amdgpu_buffer_rsrc_t dyb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, (unsigned)((size_t)a.R * a.ldy * (ABF ? 2 : 4)), BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t xb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (unsigned)((size_t)a.R * a.ldx * (XBF ? 2 : 4)), BUF_FLAGS);
    // one piece = one wave-instruction = 2 tile rows x 128 columns; a tile is 8 pieces per operand, wave w
    // issues pieces w and w + 4 of both.  Columns beyond the operand's width fall into the next row (finite
    // garbage that only reaches output rows / columns that are never stored) or beyond the buffer (zeros).
    const unsigned rowl = (unsigned)(lane >> 5);
    const unsigned rowa = (unsigned)(lane >> 4);      // ABF: row of the lane inside its 4-row piece
    // bf16 tiles: the lane's tile row is 4 wave + rowa (+ 16), so its swizzle term is (rowa << 2) | wave (swz_off)
    // (the all-bf16 form only: with an fp32 X beside it the transposing reads of dY alone made the kernel SLOWER,
    // 30.7 -> 33.8 us for dW_ih at 21.8k rows, so the mixed form keeps its row-major image and 2-byte column reads)
    constexpr bool TR = XBF;
    const int chs = (lane & 15) ^ (int)((rowa << 2) | (unsigned)wave);
    const int ca = ABF ? m0 + 8 * (TR ? chs : (lane & 15)) : m0 + 4 * (lane & 31);
    const unsigned colA = (unsigned)(ca + (ca >= a.remap_from ? a.remap_add : 0)) * (ABF ? 2u : 4u), colB = XBF ? (unsigned)(n0 + 8 * chs) * 2u : (unsigned)(n0 + 4 * (lane & 31)) * 4u;
#define SV_WG_STAGE(t_, st_)                                                                                   \
    {                                                                                                          \
        char* l_ = smem + (st_) * STAGE_BYTES + wave * 1024;                                                    \
        const unsigned r0_ = (unsigned)(t_) * WK + 2 * wave + rowl;                                             \
        if (ABF) {                                                                                             \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dyb, (__attribute__((address_space(3))) void*)l_, 16,      \
                                                     ((unsigned)(t_) * WKT + 4 * wave + rowa) * (unsigned)a.ldy * 2u + colA, 0, 0, 0); \
            if (XBF)                                                                                           \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(dyb, (__attribute__((address_space(3))) void*)(l_ + 4096), 16, \
                                                         ((unsigned)(t_) * WKT + 16 + 4 * wave + rowa) * (unsigned)a.ldy * 2u + colA, 0, 0, 0); \
        } else {                                                                                               \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dyb, (__attribute__((address_space(3))) void*)l_, 16,      \
                                                     r0_ * (unsigned)a.ldy * 4u + colA, 0, 0, 0);               \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(dyb, (__attribute__((address_space(3))) void*)(l_ + 4096), 16, \
                                                     (r0_ + 8) * (unsigned)a.ldy * 4u + colA, 0, 0, 0);         \
        }                                                                                                      \
        if (XBF) {                                                                                             \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xb, (__attribute__((address_space(3))) void*)(l_ + TILE_BYTES), 16, \
                                                     ((unsigned)(t_) * WKT + 4 * wave + rowa) * (unsigned)a.ldx * 2u + colB, 0, 0, 0); \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xb, (__attribute__((address_space(3))) void*)(l_ + TILE_BYTES + 4096), 16, \
                                                     ((unsigned)(t_) * WKT + 16 + 4 * wave + rowa) * (unsigned)a.ldx * 2u + colB, 0, 0, 0); \
        } else {                                                                                               \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xb, (__attribute__((address_space(3))) void*)(l_ + TILE_BYTES), 16, \
                                                     r0_ * (unsigned)a.ldx * 4u + colB, 0, 0, 0);               \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xb, (__attribute__((address_space(3))) void*)(l_ + TILE_BYTES + 4096), 16, \
                                                     (r0_ + 8) * (unsigned)a.ldx * 4u + colB, 0, 0, 0);         \
        }                                                                                                      \
    }

    unsigned trA[2][2], trB[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            trA[i][h] = tr_lane_off(lane, wm + 32 * i, h);
            trB[i][h] = tr_lane_off(lane, wn + 32 * i, h);
        }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const bool do_cs = a.colsum != nullptr && bx == 0 && tid < WM;
    float cs = 0.f;

    // prologue: three tiles in flight (a missing tile is simply not issued; counted waits below assume 4 pieces
    // per issued tile, so the tail drains with vmcnt(0))
#pragma unroll
    for (int p = 0; p < NST - 1; ++p)
        if (t_begin + p < t_end) SV_WG_STAGE(t_begin + p, p)

    for (int t = t_begin; t < t_end; ++t) {
        const int st = (t - t_begin) & (NST - 1);
        // own pieces of tile t landed (tiles t+1, t+2 may still be in flight: 8 younger pieces)
        // (pieces per tile and wave: 2 + 2, or 1 + 2 with a bf16 dY)
        // (pieces per tile and wave: 2 + 2; 1 + 2 with a bf16 dY; 2 + 2 again with both bf16 and 32-row stages)
        if (t + 2 < t_end) {
            if (ABF && !XBF) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else if (t + 1 < t_end) {
            if (ABF && !XBF) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();       // tile t complete; everyone is done reading tile t-1's stage
        asm volatile("" ::: "memory");
#if !defined(SV_WG_EXP) || SV_WG_EXP != 2
        if (t + NST - 1 < t_end) SV_WG_STAGE(t + NST - 1, (st + NST - 1) & (NST - 1))
#endif
        const float* sA = reinterpret_cast<const float*>(smem + st * STAGE_BYTES);
        const float* sB = sA + WK * WM;
        if (do_cs) {
#pragma unroll
            for (int k = 0; k < WKT; ++k)
                cs += TR ? sv_bf2f(*reinterpret_cast<const unsigned short*>(reinterpret_cast<const char*>(sA) + swz_off((unsigned)k, (unsigned)tid >> 3) + 2u * ((unsigned)tid & 7u)))
                          : ABF ? sv_bf2f(reinterpret_cast<const unsigned short*>(sA)[k * WM + tid]) : sA[k * WM + tid];
        }
#pragma unroll
        for (int ks = 0; ks < WKT / WK; ++ks) {       // one MFMA k-step = 16 rows of the stage
        // fragment of a 32-column group: lane (li, lh) takes rows 8 lh .. 8 lh + 7 of column base + li
        bf16x8 fa[2], fb[2], fam[2], fbm[2], fal[2], fbl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            uint4 u;
            if (TR) {        // two transposing reads of the swizzled image (k-step ks = 16 rows = 4096 bytes further)
                const char* tA = reinterpret_cast<const char*>(sA) + 4096 * ks;
                fa[i] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(tr16(tA + trA[i][0]), tr16(tA + trA[i][1]), 0, 1, 2, 3, 4, 5, 6, 7));
            } else if (ABF) {
                const unsigned short* p = reinterpret_cast<const unsigned short*>(sA) + (16 * ks + 8 * lh) * WM + wm + 32 * i + li;
                u.x = (unsigned)p[0] | ((unsigned)p[WM] << 16); u.y = (unsigned)p[2 * WM] | ((unsigned)p[3 * WM] << 16);
                u.z = (unsigned)p[4 * WM] | ((unsigned)p[5 * WM] << 16); u.w = (unsigned)p[6 * WM] | ((unsigned)p[7 * WM] << 16);
                fa[i] = __builtin_bit_cast(bf16x8, u);
            } else if (SPLIT) {
                const float* p = sA + (8 * lh) * WM + wm + 32 * i + li;
                const float v[8] = {p[0], p[WM], p[2 * WM], p[3 * WM], p[4 * WM], p[5 * WM], p[6 * WM], p[7 * WM]};
                split8(v, fa[i], fam[i], fal[i]);
            } else {
                const float* p = sA + (8 * lh) * WM + wm + 32 * i + li;
                u.x = pk2(p[0], p[WM]); u.y = pk2(p[2 * WM], p[3 * WM]);
                u.z = pk2(p[4 * WM], p[5 * WM]); u.w = pk2(p[6 * WM], p[7 * WM]);
                fa[i] = __builtin_bit_cast(bf16x8, u);
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float* p = sB + (8 * lh) * WN + wn + 32 * j + li;
            if (XBF) {
                const char* tB = reinterpret_cast<const char*>(sB) + 4096 * ks;
                fb[j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(tr16(tB + trB[j][0]), tr16(tB + trB[j][1]), 0, 1, 2, 3, 4, 5, 6, 7));
            } else if (SPLIT) {
                const float v[8] = {p[0], p[WN], p[2 * WN], p[3 * WN], p[4 * WN], p[5 * WN], p[6 * WN], p[7 * WN]};
                split8(v, fb[j], fbm[j], fbl[j]);
            } else {
                uint4 u;
                u.x = pk2(p[0], p[WN]); u.y = pk2(p[2 * WN], p[3 * WN]);
                u.z = pk2(p[4 * WN], p[5 * WN]); u.w = pk2(p[6 * WN], p[7 * WN]);
                fb[j] = __builtin_bit_cast(bf16x8, u);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (SPLIT) {       // small terms first
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[i], fb[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fbl[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fam[i], fbm[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fam[i], fb[j], acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fbm[j], acc[i][j], 0, 0, 0);
                }
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        }
    }
#undef SV_WG_STAGE

    float* C = a.out + (size_t)bz * a.slab_stride;
    if (do_cs && m0 + tid < a.N) (a.colsum + (size_t)bz * a.slab_stride)[m0 + tid] = cs;
    // C layout of a 32x32 tile: col = lane&31 -> X column, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -> dY column
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn + 32 * j + li;
        if (n >= a.K) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m < a.N) C[(size_t)m * a.K + n] = acc[i][j][r];
            }
    }
}

template <bool ABF, bool SPLIT = false, bool XBF = false>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    wgrad_body<ABF, SPLIT, XBF>(a, blockIdx.x, smem);
}

// Two jobs in one launch -- the GRU's recurrent- and input-weight gradients, which share dY and follow each other on
// the step's tail: job 0 (bf16 dY, fp32 X) on workgroups [0, n0), job 1 (bf16 dY, X fp32 or bf16) on the rest.  One
// kernel boundary and one slab reduce less on that chain, and no second partial round of workgroups in between;
// every workgroup does exactly what it does in the single-job launches (same slabs, bit-identical).
template <bool XBF1>
__global__ __launch_bounds__(256) void wgrad_pair_kernel(WgradArgs a0, WgradArgs a1, int n0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x < n0) wgrad_body<true, false, false>(a0, blockIdx.x, smem);
    else wgrad_body<true, false, XBF1>(a1, (int)blockIdx.x - n0, smem);
}

// ---- The all-bf16 form with LDS reads the compiler does not see (round 3): wgrad_body<true, false, true> -- same tiles,
// same stages, same MFMA order, bit-identical weight gradients -- with its fragment reads as inline asm.  Behind the
// builtin reads the compiler puts an s_waitcnt vmcnt(0) in front of the first LDS read of every stage (it cannot tell the
// read from the LDS-DMA writes in flight): the prefetch ring of three tiles is drained at every stage, the tile requested a
// moment ago included.  Here the waits are hand-counted: vmcnt for the stage's own pieces in front of the barrier,
// lgkmcnt in front of each k-step's MFMAs (LDS returns in order).  Column sums: a thread owns one 16-byte chunk (8
// columns) of rows rg and rg + 16 -- two 16-byte reads per stage instead of 32 two-byte ones --, the 16 row groups are
// summed through LDS at the end (another summation order than wgrad_body's: the bias gradients agree to rounding).
__device__ __forceinline__ void wgrad16_body(const WgradArgs& a, const int bid, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int xcd = bid & 7, slot = bid >> 3;
    const int gsize = a.nx * a.ny, bz = xcd + 8 * (slot / gsize), mem = slot % gsize;
    if (bz >= a.nz) return;
    const int bx = mem % a.nx, by = mem / a.nx;
    const int m0 = by * WM, n0 = bx * WN;
    const int ntiles = (a.R + 31) / 32;
    const int tps = a.tiles_per_split / 2;             // (a.tiles_per_split counts 16-row tiles)
    const int t_begin = bz * tps;
    int t_end = t_begin + tps;
    if (t_end > ntiles) t_end = ntiles;

    const __amdgpu_buffer_rsrc_t dyb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, (unsigned)((size_t)a.R * a.ldy * 2), BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t xb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (unsigned)((size_t)a.R * a.ldx * 2), BUF_FLAGS);
    const unsigned rowa = (unsigned)(lane >> 4);
    const int chs = (lane & 15) ^ (int)((rowa << 2) | (unsigned)wave);
    const int ca = m0 + 8 * chs;
    const unsigned colA = (unsigned)(ca + (ca >= a.remap_from ? a.remap_add : 0)) * 2u, colB = (unsigned)(n0 + 8 * chs) * 2u;
    const unsigned ldy2 = (unsigned)a.ldy * 2u, ldx2 = (unsigned)a.ldx * 2u;
#define SV_W16_LD(rs_, dst_, off_) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (__attribute__((address_space(3))) void*)(dst_), 16, off_, 0, 0, 0);
#define SV_W16_STAGE(t_, st_)                                                                                   \
    {                                                                                                           \
        char* l_ = smem + (st_) * STAGE_BYTES + wave * 1024;                                                     \
        const unsigned ra_ = ((unsigned)(t_) * 32u + 4u * (unsigned)wave + rowa);                                \
        SV_W16_LD(dyb, l_, ra_ * ldy2 + colA) SV_W16_LD(dyb, l_ + 4096, (ra_ + 16u) * ldy2 + colA)               \
        SV_W16_LD(xb, l_ + TILE_BYTES, ra_ * ldx2 + colB) SV_W16_LD(xb, l_ + TILE_BYTES + 4096, (ra_ + 16u) * ldx2 + colB) \
    }
    const unsigned lds0 = (unsigned)(size_t)smem;
    // fragment (i = 0, 1; half h) offsets inside a k-step's image, for the A tile (column base wm + 32 i) and the B tile
    const unsigned tA00 = tr_lane_off(lane, wm, 0), tA01 = tr_lane_off(lane, wm, 1), tA10 = tr_lane_off(lane, wm + 32, 0),
                   tA11 = tr_lane_off(lane, wm + 32, 1), tB00 = tr_lane_off(lane, wn, 0), tB01 = tr_lane_off(lane, wn, 1),
                   tB10 = tr_lane_off(lane, wn + 32, 0), tB11 = tr_lane_off(lane, wn + 32, 1);
    const unsigned cs_c8 = (unsigned)tid & 15u, cs_rg = (unsigned)tid >> 4;
    const unsigned cs_a0 = swz_off(cs_rg, cs_c8), cs_a1 = swz_off(cs_rg + 16u, cs_c8);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float cs0 = 0.f, cs1 = 0.f, cs2 = 0.f, cs3 = 0.f, cs4 = 0.f, cs5 = 0.f, cs6 = 0.f, cs7 = 0.f;
    typedef unsigned u4v __attribute__((ext_vector_type(4)));

#pragma unroll
    for (int p = 0; p < NST - 1; ++p)
        if (t_begin + p < t_end) SV_W16_STAGE(t_begin + p, p)

#define SV_W16_RD(d_, base_, o0_, o1_)                                                                           \
    {                                                                                                           \
        s4v h0_, h1_;                                                                                           \
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(h0_) : "v"((base_) + (o0_)));                            \
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(h1_) : "v"((base_) + (o1_)));                            \
        d_ = __builtin_bit_cast(bf16x8, __builtin_shufflevector(h0_, h1_, 0, 1, 2, 3, 4, 5, 6, 7));              \
    }
    for (int t = t_begin; t < t_end; ++t) {
        const int st = (t - t_begin) & (NST - 1);
        // own pieces of tile t landed (4 per tile and wave; tiles t + 1, t + 2 may still be in flight)
        if (t + 2 < t_end) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (t + 1 < t_end) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();       // tile t complete; everyone is done reading tile t - 1's stage
        asm volatile("" ::: "memory");
        if (t + NST - 1 < t_end) SV_W16_STAGE(t + NST - 1, (st + NST - 1) & (NST - 1))
        const unsigned sA = lds0 + (unsigned)(st * STAGE_BYTES), sB = sA + TILE_BYTES;
        bf16x8 a00, a01, b00, b01, a10, a11, b10, b11;      // [k-step][i]
        u4v c0, c1;
        SV_W16_RD(a00, sA, tA00, tA01) SV_W16_RD(a01, sA, tA10, tA11)
        SV_W16_RD(b00, sB, tB00, tB01) SV_W16_RD(b01, sB, tB10, tB11)
        SV_W16_RD(a10, sA + 4096u, tA00, tA01) SV_W16_RD(a11, sA + 4096u, tA10, tA11)
        SV_W16_RD(b10, sB + 4096u, tB00, tB01) SV_W16_RD(b11, sB + 4096u, tB10, tB11)
        asm volatile("ds_read_b128 %0, %1" : "=v"(c0) : "v"(sA + cs_a0));
        asm volatile("ds_read_b128 %0, %1" : "=v"(c1) : "v"(sA + cs_a1));
        // 18 reads in flight, returned in order: the first k-step's 8 are there when 10 remain
        asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(a00), "+v"(a01), "+v"(b00), "+v"(b01));
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a00, b00, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a00, b01, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a01, b00, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a01, b01, acc[1][1], 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a10), "+v"(a11), "+v"(b10), "+v"(b11));
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a10, b10, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a10, b11, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a11, b10, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a11, b11, acc[1][1], 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c0), "+v"(c1));
#define SV_W16_CSADD(v_)                                                                                         \
        cs0 += __uint_as_float(v_.x << 16); cs1 += __uint_as_float(v_.x & 0xFFFF0000u);                          \
        cs2 += __uint_as_float(v_.y << 16); cs3 += __uint_as_float(v_.y & 0xFFFF0000u);                          \
        cs4 += __uint_as_float(v_.z << 16); cs5 += __uint_as_float(v_.z & 0xFFFF0000u);                          \
        cs6 += __uint_as_float(v_.w << 16); cs7 += __uint_as_float(v_.w & 0xFFFF0000u);
        SV_W16_CSADD(c0) SV_W16_CSADD(c1)
#undef SV_W16_CSADD
    }
#undef SV_W16_RD
#undef SV_W16_STAGE
#undef SV_W16_LD

    float* C = a.out + (size_t)bz * a.slab_stride;
    if (a.colsum != nullptr && bx == 0) {
        // (__syncthreads, not the bare s_barrier builtin: that one is no memory fence to the compiler, which moved the
        //  reads of `red` across it -- bias gradients that changed from run to run, found by tools/probe/soak_which.py)
        __syncthreads();                       // every wave is past its last read of the stages
        float* red = reinterpret_cast<float*>(smem);           // [16 row groups][128 columns]
        float* rp = red + cs_rg * 128 + 8 * cs_c8;
        rp[0] = cs0; rp[1] = cs1; rp[2] = cs2; rp[3] = cs3; rp[4] = cs4; rp[5] = cs5; rp[6] = cs6; rp[7] = cs7;
        __syncthreads();
        if (tid < WM) {
            float sum = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) sum += red[g * 128 + tid];
            if (m0 + tid < a.N) (a.colsum + (size_t)bz * a.slab_stride)[m0 + tid] = sum;
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn + 32 * j + li;
        if (n >= a.K) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int mb = m0 + wm + 32 * i + 4 * lh;
            if (m0 + wm + 32 * i + 31 < a.N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) C[(size_t)(mb + (r & 3) + 8 * (r >> 2)) * a.K + n] = acc[i][j][r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    if (m < a.N) C[(size_t)m * a.K + n] = acc[i][j][r];
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void wgrad_pair16_kernel(WgradArgs a0, WgradArgs a1, int n0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const bool f = (int)blockIdx.x < n0;
    WgradArgs a;
    a.dy = f ? a0.dy : a1.dy; a.x = f ? a0.x : a1.x; a.out = f ? a0.out : a1.out; a.colsum = f ? a0.colsum : a1.colsum;
    a.R = a0.R; a.N = a0.N; a.K = f ? a0.K : a1.K; a.ldy = a0.ldy; a.ldx = f ? a0.ldx : a1.ldx;
    a.nx = f ? a0.nx : a1.nx; a.ny = a0.ny; a.nz = a0.nz; a.tiles_per_split = a0.tiles_per_split;
    a.slab_stride = f ? a0.slab_stride : a1.slab_stride;
    a.remap_from = f ? a0.remap_from : a1.remap_from; a.remap_add = f ? a0.remap_add : a1.remap_add;
    wgrad16_body(a, f ? (int)blockIdx.x : (int)blockIdx.x - n0, smem);
}

// ---- The all-bf16 form on 256 x 256 tiles (round 3): the GRU's two weight gradients on the step's tail.
// The 128 x 128 kernel above is bound by its k-loop's per-stage latency (header): a wave has 8 MFMAs (256 pipe cycles) per
// 32-row stage between two waits and a barrier.  Here a workgroup of 4 waves owns 256 dY columns x 256 X columns, a wave
// 128 x 128 of it (16 accumulator tiles, 256 registers: one wave per SIMD), and a 32-row stage carries 32 MFMAs per wave
// (1024 pipe cycles) for the same wait + barrier; operand bytes per output element halve.  An operand is staged as two
// of the 128-column swizzled images of the kernel above, side by side (a stage: A0 A1 B0 B1, 8 KiB each), filled and read
// (ds_read_b64_tr_b16) exactly as there.  6 output tiles (3 x 1 per job) x 40 row splits = 240 workgroups: one round.
constexpr int BIG_T = 256;
constexpr int BIG_IMG = 32 * 256;              // one 128-column bf16 image of a 32-row stage: 8 KiB
constexpr int BIG_STAGE = 4 * BIG_IMG;         // 32 KiB
constexpr int BIG_NST = 4;

__device__ __forceinline__ void wgrad_big_body(const WgradArgs& a, const int bid, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int wm = (wave >> 1) * 128, wn = (wave & 1) * 128;
    const int xcd = bid & 7, slot = bid >> 3;
    const int gsize = a.nx * a.ny, bz = xcd + 8 * (slot / gsize), mem = slot % gsize;
    if (bz >= a.nz) return;
    const int bx = mem % a.nx, by = mem / a.nx;
    const int m0 = by * BIG_T, n0 = bx * BIG_T;
    const int ntiles = (a.R + 31) / 32;
    const int t_begin = bz * a.tiles_per_split;          // (32-row tiles here)
    int t_end = t_begin + a.tiles_per_split;
    if (t_end > ntiles) t_end = ntiles;

    const __amdgpu_buffer_rsrc_t dyb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, (unsigned)((size_t)a.R * a.ldy * 2), BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t xb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (unsigned)((size_t)a.R * a.ldx * 2), BUF_FLAGS);
    // a piece = one wave-instruction = 4 rows of one image; wave w issues rows 4 w .. 4 w + 3 and 16 + the same of all
    // four images (8 pieces per stage and wave); the lane fetches the chunk that belongs at its LDS position (swz_off)
    const unsigned rowa = (unsigned)(lane >> 4);
    const int chs = (lane & 15) ^ (int)((rowa << 2) | (unsigned)wave);
    const int ca0 = m0 + 8 * chs, ca1 = m0 + 128 + 8 * chs;
    const unsigned colA0 = (unsigned)(ca0 + (ca0 >= a.remap_from ? a.remap_add : 0)) * 2u,
                   colA1 = (unsigned)(ca1 + (ca1 >= a.remap_from ? a.remap_add : 0)) * 2u,
                   colB0 = (unsigned)(n0 + 8 * chs) * 2u, colB1 = (unsigned)(n0 + 128 + 8 * chs) * 2u;
    const unsigned ldy2 = (unsigned)a.ldy * 2u, ldx2 = (unsigned)a.ldx * 2u;
#define SV_BIG_LD(rs_, dst_, off_) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_, (__attribute__((address_space(3))) void*)(dst_), 16, off_, 0, 0, 0);
#define SV_BIG_STAGE(t_, st_)                                                                                   \
    {                                                                                                           \
        char* l_ = smem + (st_) * BIG_STAGE + wave * 1024;                                                       \
        const unsigned ra_ = ((unsigned)(t_) * 32u + 4u * (unsigned)wave + rowa);                                \
        SV_BIG_LD(dyb, l_, ra_ * ldy2 + colA0) SV_BIG_LD(dyb, l_ + 4096, (ra_ + 16u) * ldy2 + colA0)             \
        SV_BIG_LD(dyb, l_ + BIG_IMG, ra_ * ldy2 + colA1) SV_BIG_LD(dyb, l_ + BIG_IMG + 4096, (ra_ + 16u) * ldy2 + colA1) \
        SV_BIG_LD(xb, l_ + 2 * BIG_IMG, ra_ * ldx2 + colB0) SV_BIG_LD(xb, l_ + 2 * BIG_IMG + 4096, (ra_ + 16u) * ldx2 + colB0) \
        SV_BIG_LD(xb, l_ + 3 * BIG_IMG, ra_ * ldx2 + colB1) SV_BIG_LD(xb, l_ + 3 * BIG_IMG + 4096, (ra_ + 16u) * ldx2 + colB1) \
    }

    // (the same offsets serve both operands: a wave's column bases are 32 i in its image)
    const unsigned tr00 = tr_lane_off(lane, 0, 0), tr01 = tr_lane_off(lane, 0, 1), tr10 = tr_lane_off(lane, 32, 0),
                   tr11 = tr_lane_off(lane, 32, 1), tr20 = tr_lane_off(lane, 64, 0), tr21 = tr_lane_off(lane, 64, 1),
                   tr30 = tr_lane_off(lane, 96, 0), tr31 = tr_lane_off(lane, 96, 1);
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // Column sums of dY (the bias gradient), by the workgroups of the first X tile: a thread owns 8 columns (one 16-byte
    // chunk) of rows rg, rg + 8, rg + 16, rg + 24 of every stage -- four 16-byte LDS reads per stage instead of 32
    // two-byte ones --, and the eight row groups are summed through LDS once, at the end.
    const bool do_cs = a.colsum != nullptr && bx == 0;
    float cs0 = 0.f, cs1 = 0.f, cs2 = 0.f, cs3 = 0.f, cs4 = 0.f, cs5 = 0.f, cs6 = 0.f, cs7 = 0.f;
    const unsigned cs_c8 = (unsigned)tid & 31u, cs_rg = (unsigned)tid >> 5;
    const unsigned cs_a0 = (cs_c8 >> 4) * BIG_IMG + swz_off(cs_rg, cs_c8 & 15u), cs_a1 = (cs_c8 >> 4) * BIG_IMG + swz_off(cs_rg + 8u, cs_c8 & 15u),
                   cs_a2 = (cs_c8 >> 4) * BIG_IMG + swz_off(cs_rg + 16u, cs_c8 & 15u), cs_a3 = (cs_c8 >> 4) * BIG_IMG + swz_off(cs_rg + 24u, cs_c8 & 15u);

    // Software pipeline over half-stages (16 rows = one MFMA k-step): the 16 transposing reads of the NEXT k-step's
    // fragments are issued one behind each of the 16 MFMAs of the current one (one wave per SIMD: nothing else hides an
    // LDS read), and the stage's one barrier sits between its two k-steps -- by then every wave holds the stage's second
    // half in registers, so the stage is refilled at once (tiles t + 1 .. t + 3 in flight while t + 4 is requested).
    // (Known cost, not yet removed: behind the LDS-read builtins the compiler puts an s_waitcnt vmcnt(0) in front of the
    //  first read of a phase -- an LDS read it cannot tell apart from the LDS-DMA writes in flight -- which drains the
    //  prefetch ring at every stage, here and in the 128 x 128 kernels above (llvm-objdump).  proj.hip avoids it with
    //  inline-asm reads and hand-counted waits; the same change here made the register allocator spill at 100 VGPRs
    //  -- inline asm with 64-bit outputs beside 256 accumulator registers -- and ran 65 us instead of 36.)
    bf16x8 ra[4], rb[4], na[4], nb[4];     // the current / next fragments
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    u4v cv0, cv1, cv2, cv3;
#define SV_BIG_FRAG(r_, i_) r_[i_]
#define SV_BIG_RD1(da_, db_, i_, t0_, t1_)                                                                       \
    {                                                                                                           \
        s4v a0_, a1_, b0_, b1_;                                                                                 \
        a0_ = tr16(smem + bA_ + t0_); a1_ = tr16(smem + bA_ + t1_);                                               \
        b0_ = tr16(smem + bB_ + t0_); b1_ = tr16(smem + bB_ + t1_);                                               \
        da_[i_] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0_, a1_, 0, 1, 2, 3, 4, 5, 6, 7));         \
        db_[i_] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0_, b1_, 0, 1, 2, 3, 4, 5, 6, 7));         \
    }
    // (written out, on scalars: with the offsets in an array and the asm in an unrolled loop the array stayed in scratch
    //  memory and every read was preceded by a scratch load and a vmcnt(0))
#define SV_BIG_READS(da_, db_, stg_, ks_)                                                                        \
    {                                                                                                           \
        const unsigned bA_ = (unsigned)((stg_) * BIG_STAGE + (wave >> 1) * BIG_IMG + 4096 * (ks_));              \
        const unsigned bB_ = (unsigned)((stg_) * BIG_STAGE + (2 + (wave & 1)) * BIG_IMG + 4096 * (ks_));         \
        SV_BIG_RD1(da_, db_, 0, tr00, tr01) SV_BIG_RD1(da_, db_, 1, tr10, tr11)                                  \
        SV_BIG_RD1(da_, db_, 2, tr20, tr21) SV_BIG_RD1(da_, db_, 3, tr30, tr31)                                  \
    }
#define SV_BIG_MFMAS(ca_, cb_)                                                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                            \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                        \
            acc[i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(SV_BIG_FRAG(ca_, i_), SV_BIG_FRAG(cb_, j_), acc[i_][j_], 0, 0, 0);

#pragma unroll
    for (int p = 0; p < BIG_NST - 1; ++p)
        if (t_begin + p < t_end) SV_BIG_STAGE(t_begin + p, p)
    if (t_begin < t_end) {
        // first tile landed (two younger tiles may be in flight)
        if (t_begin + 2 < t_end) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (t_begin + 1 < t_end) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t_begin + BIG_NST - 1 < t_end) SV_BIG_STAGE(t_begin + BIG_NST - 1, BIG_NST - 1)
        SV_BIG_READS(ra, rb, 0, 0)
    }
    for (int t = t_begin; t < t_end; ++t) {
        const int st = (t - t_begin) & (BIG_NST - 1);
        // k-step 0 of tile t (fragments in ra / rb) beside the reads of its k-step 1 (and of the column sums' chunks)
        SV_BIG_READS(na, nb, st, 1)
        {   // (every workgroup: a branch around inline asm here cost 170 registers)
            const char* cb = smem + st * BIG_STAGE;
            cv0 = *reinterpret_cast<const u4v*>(cb + cs_a0); cv1 = *reinterpret_cast<const u4v*>(cb + cs_a1);
            cv2 = *reinterpret_cast<const u4v*>(cb + cs_a2); cv3 = *reinterpret_cast<const u4v*>(cb + cs_a3);
        }
        SV_BIG_MFMAS(ra, rb)
        // tile t + 1 landed?  (two younger tiles may be in flight)
        if (t + 1 < t_end) {
            if (t + 3 < t_end) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (t + 2 < t_end) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();       // every wave holds tile t in registers: its stage is free, tile t + 1 is complete
        asm volatile("" ::: "memory");
        if (t + BIG_NST < t_end) SV_BIG_STAGE(t + BIG_NST, st)
        if (t + 1 < t_end) SV_BIG_READS(ra, rb, (st + 1) & (BIG_NST - 1), 0)
        SV_BIG_MFMAS(na, nb)
#define SV_BIG_CSADD(v_)                                                                                         \
        cs0 += __uint_as_float(v_.x << 16); cs1 += __uint_as_float(v_.x & 0xFFFF0000u);                          \
        cs2 += __uint_as_float(v_.y << 16); cs3 += __uint_as_float(v_.y & 0xFFFF0000u);                          \
        cs4 += __uint_as_float(v_.z << 16); cs5 += __uint_as_float(v_.z & 0xFFFF0000u);                          \
        cs6 += __uint_as_float(v_.w << 16); cs7 += __uint_as_float(v_.w & 0xFFFF0000u);
        SV_BIG_CSADD(cv0) SV_BIG_CSADD(cv1) SV_BIG_CSADD(cv2) SV_BIG_CSADD(cv3)
#undef SV_BIG_CSADD
    }
#undef SV_BIG_FRAG
#undef SV_BIG_READS
#undef SV_BIG_RD1
#undef SV_BIG_MFMAS
#undef SV_BIG_STAGE
#undef SV_BIG_LD

    float* C = a.out + (size_t)bz * a.slab_stride;
    if (do_cs) {
        // sum the eight row groups' partial column sums through LDS (the stages are free: every wave is past its last read)
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);           // [8 row groups][256 columns]
        float* rp = red + cs_rg * 256 + 8 * cs_c8;
        rp[0] = cs0; rp[1] = cs1; rp[2] = cs2; rp[3] = cs3; rp[4] = cs4; rp[5] = cs5; rp[6] = cs6; rp[7] = cs7;
        __syncthreads();
        float sum = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) sum += red[g * 256 + tid];
        if (m0 + tid < a.N) (a.colsum + (size_t)bz * a.slab_stride)[m0 + tid] = sum;
    }
    // C layout of a 32x32 tile: col = lane&31 -> X column, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -> dY column.  A 32-row
    // group that lies inside N is stored without per-element checks (N = 672 = 21 groups: always, for the GRU).
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn + 32 * j + li;
        if (n >= a.K) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int mb = m0 + wm + 32 * i + 4 * lh;
            if (m0 + wm + 32 * i + 31 < a.N) {
#pragma unroll
                for (int r = 0; r < 16; ++r) C[(size_t)(mb + (r & 3) + 8 * (r >> 2)) * a.K + n] = acc[i][j][r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    if (m < a.N) C[(size_t)m * a.K + n] = acc[i][j][r];
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void wgrad_big_pair_kernel(WgradArgs a0, WgradArgs a1, int n0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // the job's arguments selected field by field (scalar selects): with the body called on a0 or a1 by reference the
    // compiler kept the structure in scratch memory and re-read its fields inside the loop
    const bool f = (int)blockIdx.x < n0;
    WgradArgs a;
    a.dy = f ? a0.dy : a1.dy; a.x = f ? a0.x : a1.x; a.out = f ? a0.out : a1.out; a.colsum = f ? a0.colsum : a1.colsum;
    a.R = a0.R; a.N = a0.N; a.K = f ? a0.K : a1.K; a.ldy = a0.ldy; a.ldx = f ? a0.ldx : a1.ldx;
    a.nx = f ? a0.nx : a1.nx; a.ny = a0.ny; a.nz = a0.nz; a.tiles_per_split = a0.tiles_per_split;
    a.slab_stride = f ? a0.slab_stride : a1.slab_stride;
    a.remap_from = f ? a0.remap_from : a1.remap_from; a.remap_add = f ? a0.remap_add : a1.remap_add;
    wgrad_big_body(a, f ? (int)blockIdx.x : (int)blockIdx.x - n0, smem);
}

// dst[i] = bf16(src[i]) (RNE), n a multiple of 8: the bf16 copy of the recurrence's previous hidden states that the
// all-bf16 weight-gradient kernels read (they round to bf16 anyway: the products are bit-identical).
__global__ void f32_to_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, long long n8) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const float4 u = reinterpret_cast<const float4*>(src)[2 * i], v = reinterpret_cast<const float4*>(src)[2 * i + 1];
        uint4 o;
        o.x = pk2(u.x, u.y); o.y = pk2(u.z, u.w); o.z = pk2(v.x, v.y); o.w = pk2(v.z, v.w);
        reinterpret_cast<uint4*>(dst)[i] = o;
    }
}

// out_j[i] = sum over job j's slabs, both jobs in one launch (blocks [0, b0): job 0); 8 slabs in flight, fixed order
__global__ void reduce_slabs_pair_kernel(const float* __restrict__ s0, float* __restrict__ o0, long long n40,
                                         long long stride0, const float* __restrict__ s1, float* __restrict__ o1,
                                         long long n41, long long stride1, int nsplit, int b0) {
    const bool first = (int)blockIdx.x < b0;
    const float* __restrict__ slabs = first ? s0 : s1;
    float* __restrict__ out = first ? o0 : o1;
    const long long n4 = first ? n40 : n41, stride = first ? stride0 : stride1;
    const int nb = first ? b0 : (int)gridDim.x - b0, b = first ? (int)blockIdx.x : (int)blockIdx.x - b0;
    for (long long i = (long long)b * blockDim.x + threadIdx.x; i < n4; i += (long long)nb * blockDim.x) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int z0 = 0; z0 < nsplit; z0 += 8) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = reinterpret_cast<const float4*>(slabs + (z0 + j < nsplit ? z0 + j : 0) * stride)[i];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (z0 + j < nsplit) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
        }
        reinterpret_cast<float4*>(out)[i] = s;
    }
}

}  // namespace

// Reference interface replaced: the weight / bias gradient of one nn.Linear (or of the GRU's input / recurrent
// weights) inside loss.backward() nb:1205.  dy [R][ldy] (N columns used), x [R][ldx] (K columns used).
// Split z of nsplit writes gW [N][K] at out + z*slab_stride and, when colsum != NULL, gb [N] at
// colsum + z*slab_stride (colsum usually = out + N*K); sum the slabs with svae_reduce_slabs.
// bf16 operands (rounded from fp32 on the fly), fp32 accumulation.
extern "C" int svae_wgrad_cols(const float* dy, const float* x, float* out, float* colsum, int R, int N, int K, int ldy,
                               int ldx, int nsplit, long long slab_stride, int remap_from, int remap_add, void* stream);
extern "C" int svae_wgrad(const float* dy, const float* x, float* out, float* colsum, int R, int N, int K, int ldy,
                          int ldx, int nsplit, long long slab_stride, void* stream) {
    return svae_wgrad_cols(dy, x, out, colsum, R, N, K, ldy, ldx, nsplit, slab_stride, N, 0, stream);
}

// The same with dY's logical column n read from stored column n + (n >= remap_from ? remap_add : 0): the GRU's
// recurrent-weight gradient takes (dr, dz, dn*r) out of the merged [R][896] array (remap_from 448, remap_add 224).
static int wgrad_launch(const void* dy, int dy_bf16, const float* x, float* out, float* colsum, int R, int N, int K, int ldy,
                        int ldx, int nsplit, long long slab_stride, int remap_from, int remap_add, void* stream);
// prec 1: bf16 operands (svae_wgrad_cols); prec 2: the split instantiation (hi/lo bf16 parts, fp32-class products)
extern "C" int svae_wgrad_cols_prec(const float* dy, const float* x, float* out, float* colsum, int R, int N, int K,
                                    int ldy, int ldx, int nsplit, long long slab_stride, int remap_from, int remap_add,
                                    int prec, void* stream) {
    SV_CHECK_ARG(prec == 1 || prec == 2, "svae_wgrad_cols_prec: prec %d (1 = bf16, 2 = split bf16)", prec);
    return wgrad_launch(dy, prec == 2 ? 2 : 0, x, out, colsum, R, N, K, ldy, ldx, nsplit, slab_stride, remap_from,
                        remap_add, stream);
}
extern "C" int svae_wgrad_cols(const float* dy, const float* x, float* out, float* colsum, int R, int N, int K, int ldy,
                               int ldx, int nsplit, long long slab_stride, int remap_from, int remap_add, void* stream) {
    return wgrad_launch(dy, 0, x, out, colsum, R, N, K, ldy, ldx, nsplit, slab_stride, remap_from, remap_add, stream);
}
// dy AND x hold bf16 elements (x: the embedding rows as svae_embed_fwd_bf16 wrote them; ldx in elements, a multiple
// of 8): the GRU's input-weight gradient with half the bytes of its re-read operand.
extern "C" int svae_wgrad_cols_bf16dy_bf16x(const void* dy, const void* x, float* out, float* colsum, int R, int N, int K,
                                            int ldy, int ldx, int nsplit, long long slab_stride, int remap_from,
                                            int remap_add, void* stream) {
    SV_CHECK_ARG((ldy % 8) == 0 && (ldx % 8) == 0 && (remap_from % 8) == 0 && (remap_add % 8) == 0,
                 "svae_wgrad_cols_bf16dy_bf16x: bad alignment");
    return wgrad_launch(dy, 3, (const float*)x, out, colsum, R, N, K, ldy, ldx, nsplit, slab_stride, remap_from, remap_add,
                        stream);
}
// dy holds bf16 elements (ldy in elements, a multiple of 8; remap_from / remap_add multiples of 8).
extern "C" int svae_wgrad_cols_bf16dy(const void* dy, const float* x, float* out, float* colsum, int R, int N, int K,
                                      int ldy, int ldx, int nsplit, long long slab_stride, int remap_from, int remap_add,
                                      void* stream) {
    SV_CHECK_ARG((ldy % 8) == 0 && (remap_from % 8) == 0 && (remap_add % 8) == 0, "svae_wgrad_cols_bf16dy: bad alignment");
    return wgrad_launch(dy, 1, x, out, colsum, R, N, K, ldy, ldx, nsplit, slab_stride, remap_from, remap_add, stream);
}

static int wgrad_args(const void* dy_, const float* x, float* out, float* colsum, int R, int N, int K, int ldy, int ldx,
                      int nsplit, long long slab_stride, int remap_from, int remap_add, WgradArgs& a, int& gsize) {

    const float* dy = (const float*)dy_;
    SV_CHECK_ARG(dy && x && out, "svae_wgrad: null pointer");
    SV_CHECK_ARG(remap_from >= 0 && remap_add >= 0 && (remap_from % 4) == 0 && (remap_add % 4) == 0 && N + remap_add <= ldy,
                 "svae_wgrad_cols: bad column remap");
    SV_CHECK_ARG(R > 0 && N > 0 && K > 0 && ldy >= N && ldx >= K && (ldy % 4) == 0 && (ldx % 4) == 0,
                 "svae_wgrad: bad shape R=%d N=%d K=%d ldy=%d ldx=%d", R, N, K, ldy, ldx);
    SV_CHECK_ARG(((uintptr_t)dy % 16) == 0 && ((uintptr_t)x % 16) == 0, "svae_wgrad: operands must be 16-byte aligned");
    // (the 4 GiB limit below is checked for the wider, fp32, element)
    SV_CHECK_ARG(nsplit >= 1 && (nsplit == 1 || slab_stride >= (long long)N * K), "svae_wgrad: bad split");
    SV_CHECK_ARG((long long)R * ldy * 4 < (1ll << 32) && (long long)R * ldx * 4 < (1ll << 32),
                 "svae_wgrad: operand larger than a 4 GiB buffer descriptor");
    a = WgradArgs{dy, x, out, colsum, R, N, K, ldy, ldx, (K + WN - 1) / WN, (N + WM - 1) / WM, nsplit, 0, slab_stride,
                  remap_from, remap_add};
    // a split = a whole number of 32-row stages (the same row boundaries for the 16- and the 32-row instantiations)
    const int n32 = (R + 31) / 32;
    a.tiles_per_split = 2 * ((n32 + nsplit - 1) / nsplit);
    gsize = a.nx * a.ny;
    return 0;
}

static int wgrad_attrs() {
    static bool attr_done = false;
    if (!attr_done) {

        hipError_t e = hipFuncSetAttribute((const void*)wgrad_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)wgrad_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)wgrad_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)wgrad_kernel<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)wgrad_pair_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE_BYTES);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)wgrad_pair_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE_BYTES);
        if (e != hipSuccess) return svae_fail((int)e, "svae_wgrad: LDS attribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    return 0;
}

static int wgrad_launch(const void* dy_, int dy_bf16, const float* x, float* out, float* colsum, int R, int N, int K, int ldy,
                        int ldx, int nsplit, long long slab_stride, int remap_from, int remap_add, void* stream) {
    WgradArgs a;
    int gsize;
    if (int rc = wgrad_args(dy_, x, out, colsum, R, N, K, ldy, ldx, nsplit, slab_stride, remap_from, remap_add, a, gsize)) return rc;
    if (int rc = wgrad_attrs()) return rc;
    if (dy_bf16 == 3)
        hipLaunchKernelGGL((wgrad_kernel<true, false, true>), dim3(8 * ((nsplit + 7) / 8) * gsize), dim3(256), NST * STAGE_BYTES,
                           (hipStream_t)stream, a);
    else if (dy_bf16 == 2)
        hipLaunchKernelGGL((wgrad_kernel<false, true>), dim3(8 * ((nsplit + 7) / 8) * gsize), dim3(256), NST * STAGE_BYTES,
                           (hipStream_t)stream, a);
    else if (dy_bf16)
        hipLaunchKernelGGL(wgrad_kernel<true>, dim3(8 * ((nsplit + 7) / 8) * gsize), dim3(256), NST * STAGE_BYTES,
                           (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(wgrad_kernel<false>, dim3(8 * ((nsplit + 7) / 8) * gsize), dim3(256), NST * STAGE_BYTES,
                           (hipStream_t)stream, a);
    SV_LAUNCH_CHECK();
    return 0;
}

// No single counterpart in the reference: the weight gradients of the GRU's recurrent and input weights (two of the
// matrix products inside loss.backward(), nb:1205) in ONE launch.  Job 0 = svae_wgrad_cols_bf16dy(dy, x0, ...), job 1 =
// svae_wgrad_cols_bf16dy(dy, x1, ...) or, with x1_bf16 != 0, svae_wgrad_cols_bf16dy_bf16x(dy, x1, ...): the same
// workgroups writing the same slabs (bit-identical to the two single launches); both jobs share dy, R, N, ldy, nsplit.
extern "C" int svae_wgrad_pair_bf16dy(const void* dy, int R, int N, int ldy, int nsplit,
                                      const float* x0, int K0, int ldx0, float* out0, float* colsum0, long long stride0,
                                      int remap_from0, int remap_add0,
                                      const void* x1, int x1_bf16, int K1, int ldx1, float* out1, float* colsum1,
                                      long long stride1, int remap_from1, int remap_add1, void* stream) {
    SV_CHECK_ARG((ldy % 8) == 0 && (remap_from0 % 8) == 0 && (remap_add0 % 8) == 0 && (remap_from1 % 8) == 0 &&
                     (remap_add1 % 8) == 0 && (!x1_bf16 || (ldx1 % 8) == 0),
                 "svae_wgrad_pair_bf16dy: bad alignment");
    WgradArgs a0, a1;
    int g0, g1;
    if (int rc = wgrad_args(dy, x0, out0, colsum0, R, N, K0, ldy, ldx0, nsplit, stride0, remap_from0, remap_add0, a0, g0)) return rc;
    if (int rc = wgrad_args(dy, (const float*)x1, out1, colsum1, R, N, K1, ldy, ldx1, nsplit, stride1, remap_from1, remap_add1, a1, g1)) return rc;
    if (int rc = wgrad_attrs()) return rc;
    const int n0 = 8 * ((nsplit + 7) / 8) * g0, n1 = 8 * ((nsplit + 7) / 8) * g1;     // n0 % 8 == 0: id % 8 stays the XCD
    if (x1_bf16)
        hipLaunchKernelGGL(wgrad_pair_kernel<true>, dim3(n0 + n1), dim3(256), NST * STAGE_BYTES, (hipStream_t)stream, a0, a1, n0);
    else
        hipLaunchKernelGGL(wgrad_pair_kernel<false>, dim3(n0 + n1), dim3(256), NST * STAGE_BYTES, (hipStream_t)stream, a0, a1, n0);
    SV_LAUNCH_CHECK();
    return 0;
}

// svae_reduce_slabs for two slab sets in one launch: out_j[0 .. n_j) = sum_z slabs_j[z * stride_j + i], fixed order.
extern "C" int svae_reduce_slabs_pair(const float* slabs0, float* out0, long long n0, long long stride0,
                                      const float* slabs1, float* out1, long long n1, long long stride1, int nsplit,
                                      void* stream) {
    SV_CHECK_ARG(slabs0 && out0 && slabs1 && out1 && n0 > 0 && n1 > 0 && (n0 % 4) == 0 && (n1 % 4) == 0 && nsplit >= 1 &&
                     (stride0 % 4) == 0 && (stride1 % 4) == 0,
                 "svae_reduce_slabs_pair: bad arguments");
    SV_CHECK_ARG(((uintptr_t)slabs0 % 16) == 0 && ((uintptr_t)slabs1 % 16) == 0 && ((uintptr_t)out0 % 16) == 0 &&
                     ((uintptr_t)out1 % 16) == 0, "svae_reduce_slabs_pair: 16-byte alignment");
    const long long n40 = n0 / 4, n41 = n1 / 4;
    int b0 = (int)((n40 + 255) / 256), b1 = (int)((n41 + 255) / 256);
    if (b0 > 1024) b0 = 1024;
    if (b1 > 1024) b1 = 1024;
    hipLaunchKernelGGL(reduce_slabs_pair_kernel, dim3(b0 + b1), dim3(256), 0, (hipStream_t)stream, slabs0, out0, n40, stride0,
                       slabs1, out1, n41, stride1, nsplit, b0);
    SV_LAUNCH_CHECK();
    return 0;
}

// No single counterpart in the reference: svae_wgrad_pair_bf16dy with BOTH X operands as bf16 elements, on 256 x 256
// tiles (wgrad_big_body).  x0 / x1 [R][ldx] bf16 (ldx multiples of 8); slabs as in the pair launch above (sum them with
// svae_reduce_slabs_pair); nsplit a multiple of 8 is best (workgroups = 8 ceil(nsplit / 8) x tiles per job).
extern "C" int svae_wgrad_pair_big(const void* dy, int R, int N, int ldy, int nsplit,
                                   const void* x0, int K0, int ldx0, float* out0, float* colsum0, long long stride0,
                                   int remap_from0, int remap_add0,
                                   const void* x1, int K1, int ldx1, float* out1, float* colsum1, long long stride1,
                                   int remap_from1, int remap_add1, void* stream) {
    SV_CHECK_ARG(dy && x0 && x1 && out0 && out1, "svae_wgrad_pair_big: null pointer");
    SV_CHECK_ARG(R > 0 && N > 0 && K0 > 0 && K1 > 0 && ldy >= N && ldx0 >= K0 && ldx1 >= K1 && nsplit >= 1,
                 "svae_wgrad_pair_big: bad shape");
    SV_CHECK_ARG((ldy % 8) == 0 && (ldx0 % 8) == 0 && (ldx1 % 8) == 0 && (remap_from0 % 8) == 0 && (remap_add0 % 8) == 0 &&
                     (remap_from1 % 8) == 0 && (remap_add1 % 8) == 0 && remap_add0 >= 0 && remap_add1 >= 0 &&
                     N + remap_add0 <= ldy && N + remap_add1 <= ldy,
                 "svae_wgrad_pair_big: bad alignment / remap");
    SV_CHECK_ARG(((uintptr_t)dy % 16) == 0 && ((uintptr_t)x0 % 16) == 0 && ((uintptr_t)x1 % 16) == 0,
                 "svae_wgrad_pair_big: operands must be 16-byte aligned");
    SV_CHECK_ARG((long long)R * ldy * 2 < (1ll << 32) && (long long)R * ldx0 * 2 < (1ll << 32) && (long long)R * ldx1 * 2 < (1ll << 32),
                 "svae_wgrad_pair_big: operand larger than a 4 GiB buffer descriptor");
    SV_CHECK_ARG(stride0 >= (long long)N * K0 && stride1 >= (long long)N * K1, "svae_wgrad_pair_big: bad slab stride");
    const int n32 = (R + 31) / 32, tps = (n32 + nsplit - 1) / nsplit;
    WgradArgs a0{(const float*)dy, (const float*)x0, out0, colsum0, R, N, K0, ldy, ldx0, (K0 + BIG_T - 1) / BIG_T,
                 (N + BIG_T - 1) / BIG_T, nsplit, tps, stride0, remap_from0, remap_add0};
    WgradArgs a1{(const float*)dy, (const float*)x1, out1, colsum1, R, N, K1, ldy, ldx1, (K1 + BIG_T - 1) / BIG_T,
                 (N + BIG_T - 1) / BIG_T, nsplit, tps, stride1, remap_from1, remap_add1};
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)wgrad_big_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           BIG_NST * BIG_STAGE);
        if (e != hipSuccess) return svae_fail((int)e, "svae_wgrad_pair_big: LDS attribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    const int z8 = 8 * ((nsplit + 7) / 8);
    const int n0 = z8 * a0.nx * a0.ny, n1 = z8 * a1.nx * a1.ny;
    hipLaunchKernelGGL(wgrad_big_pair_kernel, dim3(n0 + n1), dim3(256), BIG_NST * BIG_STAGE, (hipStream_t)stream, a0, a1, n0);
    SV_LAUNCH_CHECK();
    return 0;
}

// dst [n] bf16 = RNE(src [n] fp32); n a multiple of 8, both 16-byte aligned.
extern "C" int svae_f32_to_bf16(const float* src, void* dst, long long n, void* stream) {
    SV_CHECK_ARG(src && dst && n > 0 && (n % 8) == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0,
                 "svae_f32_to_bf16: bad arguments");
    long long nb = (n / 8 + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, src, (unsigned short*)dst, n / 8);
    SV_LAUNCH_CHECK();
    return 0;
}

// svae_wgrad_pair_bf16dy with BOTH X operands as bf16 elements (x0: the bf16 copy of the previous hidden states), on the
// hand-waited all-bf16 kernel (wgrad16_body): same workgroups, same slabs; the weight gradients are bit-identical to
// svae_wgrad_cols_bf16dy_bf16x's, the bias gradients (column sums) agree to summation order.
extern "C" int svae_wgrad_pair_bf16(const void* dy, int R, int N, int ldy, int nsplit,
                                    const void* x0, int K0, int ldx0, float* out0, float* colsum0, long long stride0,
                                    int remap_from0, int remap_add0,
                                    const void* x1, int K1, int ldx1, float* out1, float* colsum1, long long stride1,
                                    int remap_from1, int remap_add1, void* stream) {
    SV_CHECK_ARG((ldy % 8) == 0 && (ldx0 % 8) == 0 && (ldx1 % 8) == 0 && (remap_from0 % 8) == 0 && (remap_add0 % 8) == 0 &&
                     (remap_from1 % 8) == 0 && (remap_add1 % 8) == 0,
                 "svae_wgrad_pair_bf16: bad alignment");
    WgradArgs a0, a1;
    int g0, g1;
    if (int rc = wgrad_args(dy, (const float*)x0, out0, colsum0, R, N, K0, ldy, ldx0, nsplit, stride0, remap_from0, remap_add0, a0, g0)) return rc;
    if (int rc = wgrad_args(dy, (const float*)x1, out1, colsum1, R, N, K1, ldy, ldx1, nsplit, stride1, remap_from1, remap_add1, a1, g1)) return rc;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)wgrad_pair16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NST * STAGE_BYTES);
        if (e != hipSuccess) return svae_fail((int)e, "svae_wgrad_pair_bf16: LDS attribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    const int n0 = 8 * ((nsplit + 7) / 8) * g0, n1 = 8 * ((nsplit + 7) / 8) * g1;
    hipLaunchKernelGGL(wgrad_pair16_kernel, dim3(n0 + n1), dim3(256), NST * STAGE_BYTES, (hipStream_t)stream, a0, a1, n0);
    SV_LAUNCH_CHECK();
    return 0;
}
