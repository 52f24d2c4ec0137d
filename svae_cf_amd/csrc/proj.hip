// Fused |items|-wide output projection + log-softmax + next-k NLL for gfx950, forward and backward,
// WITHOUT materialising the [rows x items] logits (reference: Decoder.linear2 nb:710, F.log_softmax
// nb:802, likelihood nb:805-808, and their autograd).
//
// One kernel template, three modes.  An "X" operand is resident (32 rows of it per wave as MFMA B
// fragments in registers), a "Y" operand streams through LDS in tiles of 64 rows, written there by
// LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass):
//
//   GEMM1   C1[y][x] = sum_k Y[y][k] X[x][k]           v_mfma_f32_32x32x16_bf16, K = 160 (10 steps)
//           P[y][x]  = exp2(C1[y][x])                  -- nothing else: see "constant columns"
//   GEMM2   Z[x][n] += sum_y P[y][x] Y[y][n]           contraction over y = the REGISTER index of C1: the
//           accumulator tile is converted to bf16 in place and fed back as the A operand (no LDS round
//           trip); its B operand is a transposed bf16 copy of Y staged beside the tile.
//
//   FDH     X = decoder hidden rows h2[r], Y = W3[i].  Forward AND hidden gradient in ONE sweep over the
//           items (flash-attention style): P = exp2(logit2 - m_r) with a running stabiliser m_r,
//           Z[r][0:150] = sum_i P W3[i] (un-normalised dL/dh2), Z[r][155] = sum_i P (softmax denominator).
//           The finish kernel normalises: lse, NLL, dL/dh2 = (c_r / l_r) Z - (target part), tanh'.
//   FWD     the same sweep without GEMM2 (loss only: scoring).
//   DW      X = W3[i], Y = h2[r]:  P = c_r softmax (lse known by now);  Z[i][0:150] = dL/dW3 (dense part),
//           Z[i][150] = dL/db3 (h2's bf16 copy has a ones column).
//
// Constant columns.  K is padded 150 -> 160; the pad columns carry every additive constant of the
// exponent through the MFMA itself, so the VALU work per logit is one v_exp_f32 and half a
// v_cvt_pk_bf16_f32.  All GEMM1 exponents are in log2 units: the resident operand is a copy scaled by
// log2(e) (h2s, w3s); constants are split into bf16 hi + lo parts (16 significant bits).
//
//   column            150   151/152            153/154              155/156
//   h2b  (Y of DW)    1     1 / 1              (log2 c_r - lse2_r)  0
//   h2s  (X of FDH)   0     1 / 1              0                    (-m_r, kept in REGISTERS only)
//   w3b  (Y of FDH)   0     log2e*b_i hi/lo    0                    1 / 1
//   w3s  (X of DW)    0     log2e*b_i hi/lo    1 / 1                0
//   pad rows of h2b carry -1e30 in column 153, pad items of w3b -1e30 in column 151: P = 0 there.
//
// The running stabiliser (FDH / FWD): m_r starts at the first sub-tile's maximum and is raised lazily,
// only when a logit exceeds it by more than 2^16 (a wave-uniform, almost never taken branch): the lane
// that owns row r rewrites its own X fragment (columns 155/156) and rescales Z.  Splits use their own m;
// the finish kernel combines them.
//
// The sparse "-y" part of dlogits = w (n softmax - y) is applied by the finish kernels (targets are
// <= k per row).
//
// Grid: 1-D, one workgroup per entry of a host-built schedule (svae_proj_schedule): entry = (x tile, first
// y tile, end y tile, slab).  The number of workgroups is a whole multiple of the CU count and x tile j is
// split ns_j = q or q+1 ways (8..16) so that every CU gets the same number of y tiles: with a fixed 8-way
// split the grid was 8*ceil(nx/256) workgroups, e.g. 688 on 256 CUs = 2.7 "rounds" that take as long as 3.
// Entries are ordered so that the workgroups that share an XCD (id % 8) stream neighbouring parts of Y, which
// then stay in that XCD's L2.  Each split writes its own fp32 slab (deterministic; no atomics).
#include <algorithm>

#include <hip/hip_ext.h>

#include "common.h"

#ifdef SV_PROJ_TIMING   // diagnostic build only (tools/proj_timing.py): phase stamps of waves 0 and 4 of block 0
__device__ unsigned long long g_proj_dbg[3][2][8];
__device__ unsigned long long g_proj_blk[3][4096][4];   // per block (wave 0): realtime at start, at end, loop cycles, tiles
__device__ unsigned long long g_proj_blk2[3][4096][2];  // per block (wave 0): realtime at kernel entry, behind the epilogue's stores
__device__ unsigned long long g_proj_wv[3][8][8][8];    // block 0, per wave, tiles 8..15: cycle stamps (see SV_WSTAMP)
#define SV_WSTAMP(k_) if (blockIdx.x == 0 && j >= 8 && j < 16 && lane == 0) g_proj_wv[MODE][wave][j - 8][k_] = __builtin_amdgcn_s_memtime();
#define SV_TS(i_) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ts_acc[i_] += now_ - ts_last; ts_last = now_; __builtin_amdgcn_sched_barrier(0); }
#else
#define SV_TS(i_)
#define SV_WSTAMP(k_)
#endif
namespace {

constexpr int KP = SV_DP;          // 160
constexpr int KS1 = KP / 16;       // 10 k-steps of GEMM1
constexpr int NT2 = KP / 32;       // 5 n-tiles of GEMM2
constexpr int BX = 256;            // x rows per workgroup
constexpr int BY = 64;             // y rows per LDS tile
constexpr int NSPLIT = 8;           // nominal split count (schedule: 8..NS_MAX per x tile)
constexpr int NS_MAX = 16;          // most slabs an x tile can have
constexpr int C_ONE = 150, C_BH = 151, C_BL = 152, C_RH = 153, C_RL = 154, C_MH = 155, C_ML = 156;
// LDS images of one stage, both UNPADDED and XOR-swizzled in units of 16-byte chunks (the swizzle is
// applied to the LDS-DMA's per-lane SOURCE address; the destination is lane-linear):
//   Y   [64 rows][20 chunks]   chunk c of row r at position c ^ ((r >> 2) & 3)
//   Y^T [160 rows][8 chunks]   chunk c of row n at position c ^ ((n >> 1) & 7)
// -> every ds_read_b128 of a fragment is conflict-free (16 lanes of a read group hit 16 distinct bank quads).
constexpr int YIMG_BYTES = BY * KP * 2;        // 20480
constexpr int TIMG_BYTES = KP * BY * 2;        // 20480
constexpr int STAGE_BYTES = YIMG_BYTES + TIMG_BYTES;
constexpr int NSTAGE_P = 4;                    // software-pipelined sweeps (FDH / DW): the whole 160 KiB of the CU
constexpr int LDS_BYTES = NSTAGE_P * STAGE_BYTES;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float NEG_BIG = -1.0e30f;
constexpr float RESCALE_THR = 16.0f;           // log2 units

enum { MODE_FWD = 0, MODE_FDH = 1, MODE_DW = 2 };

struct ProjArgs {
    const unsigned short* X;    // [nx][160] bf16, scaled copy (h2s / w3s)
    const unsigned short* Y;    // [ny_pad][160] bf16 (h2b / w3b)
    const unsigned short* YT;   // [160][ldt] bf16: transposed copy of Y, columns in GEMM2's k-order
    float* out;                 // FWD: float2 (m, l) [NSPLIT][nx];  FDH / DW: slabs [NSPLIT][nx][160]
    float* mpart;               // FDH: [NSPLIT][nx] stabiliser of each split
    int nx, ny, ldt;
    const int4* sched;          // per SEGMENT: (x tile, first y tile, end y tile, slab)
    const int* wg_off;          // flat schedule: workgroup b runs segments wg_off[b] .. wg_off[b+1]-1 one after the other;
                                // NULL: one segment per workgroup (segment = blockIdx.x)
    // split instantiation only: element distance between the parts of an image (hi at the pointer, mid = RNE(v - hi)
    // one stride further, lo = RNE(v - hi - mid) two strides further; same layout each): X's, and Y's / Y^T's
    long long xps, yps;
};

// two fp32 -> packed bf16 pair (RNE): one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    const f32x2 v = {a, b};
    const bf16v2 h = __builtin_convertvector(v, bf16v2);
    return *reinterpret_cast<const unsigned*>(&h);
}

__device__ __forceinline__ void glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// ds_read_b64_tr_b16 (gfx950): a 16-lane group reads a block of 4 rows x 16 columns of 16-bit elements; lane 4q + p
// supplies the address of row q, columns 4p .. 4p+3; lane i receives column i of the four rows.  EXEC must be full.
typedef short s4v __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ s4v tr_read(unsigned addr) {
    s4v v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}

// lo part of a packed bf16 pair: RNE(a - hi(a)), RNE(b - hi(b))
__device__ __forceinline__ unsigned pack_lo(float a, float b, unsigned hw) {
    return pack_bf16(a - __uint_as_float(hw << 16), b - __uint_as_float(hw & 0xFFFF0000u));
}

// 8 waves, two per SIMD: the exp / bf16-pack VALU work of one wave issues while the partner's MFMAs execute.
//
// SPLIT = the parity instantiation of the same kernel ("prec 2"): every operand is carried as bf16 hi + mid + lo
// parts (all 24 significant bits) and every product by its six leading terms hi*hi + hi*mid + mid*hi + mid*mid +
// hi*lo + lo*hi, so the sweep delivers fp32-class logits and gradients through the same constant-column scheme,
// running stabiliser, split / slab / finish structure.  The additive constants, hi/lo pairs in two columns of the
// hi images, get their third part in the MID images (against the other operand's hi ones column): bias in column
// 151 of the items' mid images, the row constant in column 153 of h2b's mid image, the stabiliser in column 155 of
// the lane's mid X fragment; the pad columns of the lo images are zero.  Staging is the simple form (one tile: its
// hi, mid and lo images in the three stages' bytes; load, barrier, compute, barrier) since this instantiation is
// never timed.
template <int MODE, bool SPLIT = false>
__global__ __launch_bounds__(512, 1) void proj_kernel(ProjArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
#ifdef SV_PROJ_TIMING
    const unsigned long long blk_entry = __builtin_amdgcn_s_memrealtime();
#endif
    // Flat schedule (svae_proj_schedule_flat): a workgroup owns a contiguous range of the (x tile, y tile) space and
    // walks it segment by segment -- at an x-tile boundary it writes its slab, takes the next rows and starts the Y
    // pipeline again.  Every LDS read of a segment is behind its last tile's barrier, so the next segment's LDS-DMA may
    // start at once.
    const int seg0 = a.wg_off ? __builtin_amdgcn_readfirstlane(a.wg_off[blockIdx.x]) : (int)blockIdx.x;
    const int seg1 = a.wg_off ? __builtin_amdgcn_readfirstlane(a.wg_off[blockIdx.x + 1]) : seg0 + 1;
    for (int seg = seg0; seg < seg1; ++seg) {
    const int4 sc = a.sched[seg];
    const int xt = __builtin_amdgcn_readfirstlane(sc.x), split = __builtin_amdgcn_readfirstlane(sc.w);
    const int x0 = xt * BX + wave * 32;
    const int t_begin = __builtin_amdgcn_readfirstlane(sc.y);
    const int t_end = __builtin_amdgcn_readfirstlane(sc.z);

    // resident X fragments: lane (col = li -> x, half lh) element j = X[x][16 s + 8 lh + j]
    const int x = x0 + li;
    const bool xok = x < a.nx;
    bf16x8 xf[KS1], xm[SPLIT ? KS1 : 1], xl[SPLIT ? KS1 : 1];
#pragma unroll
    for (int s = 0; s < KS1; ++s) {
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0}, vm = {0, 0, 0, 0, 0, 0, 0, 0}, vl = {0, 0, 0, 0, 0, 0, 0, 0};
        if (xok) v = *reinterpret_cast<const bf16x8*>(a.X + (size_t)x * KP + 16 * s + 8 * lh);
        if (SPLIT && xok) {
            vm = *reinterpret_cast<const bf16x8*>(a.X + a.xps + (size_t)x * KP + 16 * s + 8 * lh);
            vl = *reinterpret_cast<const bf16x8*>(a.X + 2 * a.xps + (size_t)x * KP + 16 * s + 8 * lh);
        }
        xf[s] = v;
        if (SPLIT) { xm[s] = vm; xl[s] = vl; }
    }

    f32x16 Z[NT2];
    if (MODE != MODE_FWD) {
#pragma unroll
        for (int n = 0; n < NT2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) Z[n][r] = 0.f;
    }
    float m_eff = 0.f, l_run = 0.f;
    float thr = -3.0e38f;      // the first sub-tile always sets the stabiliser

    // ---- LDS-DMA staging: chunk p = tid + 512 i of an image lands at byte 16 p; its source is swizzled
    unsigned offY[3], offT[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int p = tid + 512 * i;
        const int row = p / 20, cs = p % 20;
        offY[i] = (unsigned)(row * (KP * 2) + 16 * (cs ^ ((row >> 2) & 3)));
        const int n = p >> 3, ct = p & 7;
        offT[i] = (unsigned)(n * a.ldt * 2 + 16 * (ct ^ ((n >> 1) & 7)));
    }
    const char* Yg = reinterpret_cast<const char*>(a.Y);
    const char* Tg = reinterpret_cast<const char*>(a.YT);
    // One LDS-DMA piece = one wave-instruction = 1 KiB.  A tile has 20 (+20 for Y^T) pieces; wave w issues pieces
    // w and w + 8 of each image, waves 0..3 also piece w + 16.  Piece q of wave w (q = 0..NPIECE-1) alternates
    // between the two images.  Issued ONE AT A TIME between MFMAs: a burst of all 40 right after the barrier
    // queued every wave behind the CU's address path for 700-1000 cycles per tile (measured).
    constexpr int NIMG = MODE == MODE_FWD ? 1 : 2;
    const int npiece = (wave < 4 ? 3 : 2) * NIMG;
#define SV_PIECE(q_, ybase_, tbase_, lbase_)                                                            \
    {                                                                                                  \
        const int i_ = (q_) / NIMG;                                                                    \
        if (NIMG == 1 || ((q_) & 1) == 0)                                                              \
            glds16((ybase_) + offY[i_], (lbase_) + 8192 * i_);                                         \
        else                                                                                           \
            glds16((tbase_) + offT[i_], (lbase_) + YIMG_BYTES + 8192 * i_);                            \
    }
#define SV_STAGE_ALL(t_, buf_)                                                                          \
    {                                                                                                  \
        const char* yb_ = Yg + (size_t)(t_) * (BY * KP * 2);                                           \
        const char* tb_ = Tg + (size_t)(t_) * (BY * 2);                                                \
        char* ly_ = smem + (buf_) * STAGE_BYTES + wave * 1024;                                         \
        _Pragma("unroll") for (int q_ = 0; q_ < 3 * NIMG; ++q_)                                        \
            if (q_ < npiece) SV_PIECE(q_, yb_, tb_, ly_)                                               \
    }

    // fragment read offsets inside a stage
    const int sw = (li >> 2) & 3, sw2 = (li >> 1) & 7;
    const int ra0 = li * (KP * 2) + 16 * (lh ^ sw), ra1 = li * (KP * 2) + 16 * ((2 + lh) ^ sw);
    int rb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) rb[q] = YIMG_BYTES + li * (BY * 2) + 16 * ((2 * q + lh) ^ sw2);

    // Running stabiliser of the softmax (FDH / FWD), shared by both instantiations: see the file header.
    // Returns the factor by which the lane's row was rescaled (1 when the stabiliser stayed).
    auto stabilise = [&](f32x16& c1) -> float {
    float f_ret = 1.0f;
    if (MODE != MODE_DW) {
        float tm = fmaxf(c1[0], c1[1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) tm = fmaxf(tm, fmaxf(c1[r], c1[r + 1]));
        // (the two lane halves of a row are only reconciled inside the rare branch)
        if (__any(tm > thr)) {
            tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
            const bool need = tm > thr;
            // raise the stabiliser of the rows that need it; the value used is the one the bf16 hi/lo pair
            // represents exactly, so the MFMA and the bookkeeping agree
            const float mn = fmaxf(m_eff + tm, -1.0e4f);
            const unsigned short hi = sv_f2bf(-mn);
            const float hif = sv_bf2f(hi);
            const unsigned short lo = sv_f2bf(-mn - hif);
            // (split instantiation: a third part, 24 bits in all, in the lane's mid fragment)
            const unsigned short lo2 = SPLIT ? sv_f2bf((-mn - hif) - sv_bf2f(lo)) : (unsigned short)0;
            const float mq = need ? -((hif + sv_bf2f(lo)) + sv_bf2f(lo2)) : m_eff;
            const float delta = m_eff - mq;          // <= 0 (0 for rows left alone)
            const float f = __builtin_amdgcn_exp2f(delta);
            f_ret = f;
#pragma unroll
            for (int r = 0; r < 16; ++r) c1[r] += delta;
            if (need && lh == 1) {
                xf[KS1 - 1][C_MH - 152] = (short)hi;
                xf[KS1 - 1][C_ML - 152] = (short)lo;
                if (SPLIT) xm[SPLIT ? KS1 - 1 : 0][C_MH - 152] = (short)lo2;
            }
            m_eff = mq;
            if (MODE == MODE_FWD) {
                l_run *= f;
            } else {
                // Z rows are x = (r&3) + 8 (r>>2) + 4 lh: the factor of row x sits in lane x (both halves hold it)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float f0 = __shfl(f, 8 * g + 4 * lh + 0, 64), f1 = __shfl(f, 8 * g + 4 * lh + 1, 64);
                    const float f2 = __shfl(f, 8 * g + 4 * lh + 2, 64), f3 = __shfl(f, 8 * g + 4 * lh + 3, 64);
#pragma unroll
                    for (int n = 0; n < NT2; ++n) {
                        Z[n][4 * g + 0] *= f0; Z[n][4 * g + 1] *= f1;
                        Z[n][4 * g + 2] *= f2; Z[n][4 * g + 3] *= f3;
                    }
                }
            }
            thr = RESCALE_THR;
        }
    }
    return f_ret;
    };

    if constexpr (!SPLIT && MODE != MODE_FWD) {
    // ---- software-pipelined sweep (FDH / DW) ------------------------------------------------------------------
    // Sub-tiles (32 y rows; two per tile) are numbered k = 0 .. 2n-1.  GEMM2 of sub-tile k runs one phase late,
    // TOGETHER with the vector work (exp2, bf16 pack, stabiliser maximum) of sub-tile k+1:
    //     iteration k:   A  GEMM1(k+1)                      10 MFMAs, LDS fragment reads, LDS-DMA issue in the gaps
    //                    B  GEMM2(k)  ||  exp/pack(k+1)     10 MFMAs with 2 v_exp + 1 v_cvt_pk in each gap
    // so every wave's own stream keeps the matrix pipe fed while its vector work issues (an MFMA holds the SIMD's
    // vector issue for 8 of its 32 cycles: 24 cycles of vector issue per MFMA are free).  Before, a wave alternated
    // MFMA-only and VALU-only phases and relied on its SIMD partner being in the other kind -- which the per-tile
    // barrier prevented: partners ran in lockstep, matrix pipe ~60 % busy inside the loop.
    //
    // ONE image per tile.  GEMM2's B operand is Y transposed; it is read straight out of the row-major Y image with
    // ds_read_b64_tr_b16 (gfx950: a 16-lane group reads a block of 4 rows x 16 columns and every lane gets one
    // column of it): the k-order of a converted accumulator -- element j of lane half lh is y = 16 s2 + 8 (j>>2) +
    // 4 lh + (j&3) -- is two such blocks of four consecutive rows, so a fragment is two 8-byte transposed reads, and
    // the image's chunk swizzle (position c ^ ((row >> 2) & 3), constant over a block's rows) keeps them
    // conflict-free: the four rows' 64-byte runs fall into the four quarters of the 256-byte bank window.  The
    // transposed copies (w3tb / h2tb) are not staged at all: half the L2 -> LDS traffic and half the LDS-DMA
    // instructions of the two-image form, whose loop ran at 3500 (FDH) / 3100 (DW) cycles per tile against the
    // matrix pipe's 2560 -- it was the fill, not the arithmetic, that set its pace.
    //
    // Fragments live in rotating windows of W register sets, refilled right behind the MFMA that consumed a slot:
    // the Y window runs W fragments ahead through GEMM1(k+1), GEMM1(k+2), ...; the Y^T window likewise through
    // GEMM2(k), GEMM2(k+1).  LDS: NSTP stages of 20 KiB; tile j (relative) lives in stage j % NSTP, is first read at
    // the end of iteration 2j-2 and last read in iteration 2j+1; the one barrier of a tile closes iteration 2j+1.
    // Behind it tile j+2 has landed everywhere (each wave waits for its own pieces; those of the tiles behind stay
    // in flight) and stage j % NSTP is free: tile j+NSTP is issued into it during the next tile's iterations, piece
    // by piece between MFMAs.
    //
    // (Tried: a stagger -- waves 4..7 taking the tile's barrier one phase earlier in program order, so that SIMD
    //  partners run A beside B: 3297 -> 3202 cycles per tile in the DW sweep, and 95 spilled registers in the FDH one,
    //  whose mid-iteration barrier keeps the logits alive across it; a barrier every other tile: 3194; issue priority
    //  (s_setprio) alternating between the two waves of a SIMD every half tile: the arrival spread at the barrier
    //  falls from ~650 to ~250 cycles, the tile takes as long as before; one barrier per PAIR of tiles with eight
    //  stages, tile j+6 issued during tile j and the third piece alternating between the wave halves with the tile's
    //  parity (five pieces per wave between two barriers): correct, 2 % slower on both sweeps.  None kept.)
    constexpr int W = 4;
    constexpr int PSTAGE = YIMG_BYTES;
    constexpr int NSTP = 7;                       // stages; tile j+NSTP-2 is issued during tile j, into tile j-2's stage
    constexpr int DAH = NSTP - 2;                 // how many tiles ahead the LDS-DMA runs
    const int ntile = t_end - t_begin;
    // This wave's pieces of a tile (20 pieces of 1 KiB over 8 waves): pieces w and w + 8, and waves 4..7 also piece
    // w + 12.  The odd four pieces go to the YOUNGER wave of each SIMD: with the third piece on waves 0..3 -- the ones
    // that win the issue arbitration against their SIMD partners -- a loop of nothing but this kernel's 80 MFMAs per
    // SIMD, its LDS-DMA and the barrier ran 3661 cycles per tile, with it on waves 4..7 2763, with three pieces on
    // every wave 2682, with none 2560 (tools/probe/mfma_probe.hip).
    // (In this kernel the move is worth 1 %; a dummy fourth piece on waves 0..3, three on every wave: nothing more.)
    const int npc = wave >= 4 ? 3 : 2;
    unsigned offY3;                               // source offset of the third piece: chunk tid + 768
    {
        const int p = tid + 768;
        const int row = p / 20, cs = p % 20;
        offY3 = (unsigned)(row * (KP * 2) + 16 * (cs ^ ((row >> 2) & 3)));
    }
    if (ntile > 0) {
#define SV_PIECE1(q_, ybase_, lbase_)                                                   \
    {                                                                                   \
        if ((q_) < 2) glds16((ybase_) + offY[(q_) & 1], (lbase_) + 8192 * (q_));         \
        else glds16((ybase_) + offY3, (lbase_) + 12288);                                \
    }
        // wait until all but the youngest c_ tiles' pieces of this wave have landed (c_ uniform, 0 .. NSTP-3)
#define SV_WAIT_TILES(c_)                                                                    \
    {                                                                                        \
        const int n_ = (c_) * npc;                                                           \
        if (n_ >= 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");                         \
        else if (n_ == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                    \
        else if (n_ == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                    \
        else if (n_ == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                    \
        else if (n_ == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                    \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                 \
    }
#pragma unroll
        for (int j = 0; j < DAH; ++j)
            if (j < ntile) {
                const char* yb_ = Yg + (size_t)(t_begin + j) * (BY * KP * 2);
                char* ly_ = smem + j * PSTAGE + wave * 1024;
#pragma unroll
                for (int q_ = 0; q_ < 3; ++q_)
                    if (q_ < npc) SV_PIECE1(q_, yb_, ly_)
            }
        {
            const int issued = ntile < DAH ? ntile : DAH;
            SV_WAIT_TILES(issued > 3 ? issued - 3 : 0)          // tiles 0 .. 2 landed
        }
        __syncthreads();
        // transposed fragment reads: lane 4q + p of 16-lane group gq supplies row q, columns 4p .. 4p+3 of its block
        int tb[2];
        {
            const int gq = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
            const int cc = 2 * (gq & 1) + (p4 >> 1);          // 16-byte chunk inside the n tile's 64-byte run
#pragma unroll
            for (int h = 0; h < 2; ++h)
                tb[h] = (8 * h + 4 * lh + q) * (KP * 2) + 16 * (cc ^ (2 * h + lh)) + 8 * (p4 & 1);
        }
#define SV_YA(st_, sub_, s_) \
    (*reinterpret_cast<const bf16x8*>((st_) + (sub_) * (32 * KP * 2) + 64 * ((s_) >> 1) + (((s_) & 1) ? ra1 : ra0)))
        // (st_: the per-lane LDS byte addresses {stage base + tb[0], stage base + tb[1]} of a stage, so that every read
        //  is one base register + an immediate offset)
        // The read is inline asm, not __builtin_amdgcn_ds_read_tr16_b64: behind the builtin the compiler puts an
        // s_waitcnt vmcnt(0) in front of the first read of every phase (an LDS read it cannot tell apart from the
        // LDS-DMA writes in flight), which drains the whole prefetch ring twice per tile.  The price: the compiler
        // does not count these reads, so the waits are ours -- LDS returns in order, a fragment is consumed W = 4 gaps
        // (= 6 later reads) after its two reads were issued: s_waitcnt lgkmcnt(6) in front of each MFMA of GEMM2.
#define SV_TR(st_, sub_, i_, h_)                                                                          \
    tr_read<(sub_) * (32 * KP * 2) + ((i_) & 1) * (16 * KP * 2) + 64 * ((i_) >> 1)>((st_)[h_])
#define SV_YB(st_, sub_, i_) \
    __builtin_shufflevector(SV_TR(st_, sub_, i_, 0), SV_TR(st_, sub_, i_, 1), 0, 1, 2, 3, 4, 5, 6, 7)
        const unsigned lds0 = (unsigned)(size_t)smem;
#define SV_MFMA_DS                                     \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        bf16x8 fa[W], fb[W], pa[2];
        f32x16 c1;
        // prologue: GEMM1(0), its vector work, and the two windows
        {
#pragma unroll
            for (int r = 0; r < 16; ++r) c1[r] = 0.f;
#pragma unroll
            for (int s = 0; s < KS1; ++s) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(SV_YA(smem, 0, s), xf[s], c1, 0, 0, 0);
            stabilise(c1);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                uint4 u;
                u.x = pack_bf16(__builtin_amdgcn_exp2f(c1[8 * s2 + 0]), __builtin_amdgcn_exp2f(c1[8 * s2 + 1]));
                u.y = pack_bf16(__builtin_amdgcn_exp2f(c1[8 * s2 + 2]), __builtin_amdgcn_exp2f(c1[8 * s2 + 3]));
                u.z = pack_bf16(__builtin_amdgcn_exp2f(c1[8 * s2 + 4]), __builtin_amdgcn_exp2f(c1[8 * s2 + 5]));
                u.w = pack_bf16(__builtin_amdgcn_exp2f(c1[8 * s2 + 6]), __builtin_amdgcn_exp2f(c1[8 * s2 + 7]));
                pa[s2] = *reinterpret_cast<bf16x8*>(&u);
            }
            const unsigned t0_[2] = {lds0 + (unsigned)tb[0], lds0 + (unsigned)tb[1]};
#pragma unroll
            for (int w = 0; w < W; ++w) fa[w] = SV_YA(smem, 1, w);
            static_assert(W == 4 && (2 * KS1) % W == 0, "the prologue, the waits and the slot numbering below assume this");
            fb[0] = SV_YB(t0_, 0, 0); fb[1] = SV_YB(t0_, 0, 1); fb[2] = SV_YB(t0_, 0, 2); fb[3] = SV_YB(t0_, 0, 3);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]));
        }
#ifdef SV_PROJ_TIMING
        const unsigned long long blk_rt0 = __builtin_amdgcn_s_memrealtime();
        const unsigned long long blk_c0 = __builtin_amdgcn_s_memtime();
#endif
        // The tile's barrier: tile j+2 must have landed; the tiles issued behind it (up to j+DAH) stay in flight.
        // (Tried instead of s_barrier: per-stage "tile done" counters in LDS -- a wave adds one when it is through tile
        //  j and its pieces of tile j+3 have landed, and only waits until all eight are through tile j-1, so that waves
        //  run up to a tile apart.  Correct, and 6-9 % SLOWER (FDH 0.257 vs 0.241 ms, DW 0.306 vs 0.280): the two waves
        //  of a SIMD share one matrix pipe whatever their distance, and the poll costs an LDS round trip per tile.
        //  What the in-kernel stamps say about a tile's ~3350 cycles (matrix pipe alone: 2560): without any vector work
        //  and any LDS read the loop -- 80 MFMAs per SIMD, the LDS-DMA of the next tiles, the barrier -- still takes
        //  ~3050; the exp / pack work costs ~90, the fragment reads ~160.  The LDS-DMA stream beside the MFMAs is what
        //  the remaining fifth goes to.)
#define SV_TILE_SYNC                                                                \
    {                                                                               \
        const int last_ = j + DAH < ntile ? j + DAH : ntile - 1;                    \
        SV_WAIT_TILES(last_ > j + 2 ? last_ - (j + 2) : 0)                          \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                          \
        __builtin_amdgcn_s_barrier();                                               \
        asm volatile("" ::: "memory");                                              \
    }
        int sj = 0;                                       // stage of tile j
        for (int j = 0; j < ntile; ++j) {
            const int sn = sj + 1 == NSTP ? 0 : sj + 1;   // stage of tile j+1
            const int sd = sj + DAH >= NSTP ? sj + DAH - NSTP : sj + DAH;   // stage of tile j+DAH (= of tile j-2)
            const char* st = smem + sj * PSTAGE;
            const char* stn = smem + sn * PSTAGE;
            const unsigned tst[2] = {lds0 + (unsigned)(sj * PSTAGE + tb[0]), lds0 + (unsigned)(sj * PSTAGE + tb[1])};
            const unsigned tsn[2] = {lds0 + (unsigned)(sn * PSTAGE + tb[0]), lds0 + (unsigned)(sn * PSTAGE + tb[1])};
            const bool more = j + 1 < ntile;              // a next tile exists
            const bool dma = j + DAH < ntile;             // tile j+DAH is issued during this tile's iterations
            const char* dy = Yg + (size_t)(t_begin + j + DAH) * (BY * KP * 2);
            char* dl = smem + sd * PSTAGE + wave * 1024;
#define SV_DMA(q_) if (dma && (q_) < npc) SV_PIECE1(q_, dy, dl)
            SV_WSTAMP(0)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                const bool has_next = sub == 0 || more;   // sub-tile k+1 exists
                // ---- A: GEMM1(k+1); the Y window moves on through sub-tile k+1 and into sub-tile k+2
                const char* a1 = sub == 0 ? st : stn;     // stage of sub-tile k+1 (its sub index: 1 - sub)
                if (has_next) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) c1[r] = 0.f;
#pragma unroll
                    for (int s = 0; s < KS1; ++s) {
                        // (slot of fragment s of this phase: the window turns through the 2 KS1 fragments of a tile's
                        //  two phases, and W divides 2 KS1, not KS1)
                        const int sl = (KS1 * sub + s) % W;
                        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[sl], xf[s], c1, 0, 0, 0);
                        fa[sl] = s + W < KS1 ? SV_YA(a1, 1 - sub, s + W) : SV_YA(stn, sub, s + W - KS1);
                        SV_MFMA_DS
                        if (s == 2) SV_DMA(sub)
                        if (sub == 0 && s == 6) SV_DMA(2)
                    }
                }
                SV_WSTAMP(1 + 2 * sub)
                // ---- B: GEMM2(k) beside the vector work of sub-tile k+1; the Y^T window moves on
                // Gap i = MFMA i, one transposed fragment (two reads), and for i = 1..8 two v_exp + one v_cvt_pk of
                // sub-tile k+1 (gap 0 stays bare: the logits come out of GEMM1's last MFMA, still in flight) and one
                // v_max3 of the stabiliser check.  The check itself comes AFTER the phase: a raise is only needed
                // before a logit exceeds the stabiliser by 2^127, the threshold is 2^16, so the rows that crossed it
                // in this sub-tile are raised now -- Z holds GEMM2(k) by then -- and their sixteen exponentials redone.
                {
                    unsigned pw[8];
                    float tm = -3.0e38f;
                    // Pin the order of the gaps.  The MFMAs, v_exp and v_cvt_pk are pure values to the compiler,
                    // which otherwise gathers the sixteen v_exp in front of the ten MFMAs; an empty asm that
                    // "rewrites" a gap's inputs (the MFMA's A operand, the two logits) and the gap before's output is
                    // a point they cannot cross.  It emits no instruction.  (Literal gap numbers: the transposed
                    // reads take their offsets as immediates.)
#define SV_EXP2(x_) __builtin_amdgcn_exp2f(x_)
#define SV_B_PIN(i_)                                                                                          \
    if ((i_) == 0) asm volatile("" : "+v"(pa[0]));                                                            \
    else if ((i_) == 1) asm volatile("" : "+v"(pa[1]), "+v"(c1[0]), "+v"(c1[1]));                             \
    else if ((i_) <= 8) asm volatile("" : "+v"(pa[(i_) & 1]), "+v"(c1[2 * (i_) - 2]), "+v"(c1[2 * (i_) - 1])  \
                                          : "v"(pw[(i_) >= 2 ? (i_) - 2 : 0]));                               \
    else asm volatile("" : "+v"(pa[(i_) & 1]) : "v"(pw[7]));
#define SV_SL(sub_, i_) ((KS1 * (sub_) + (i_)) % W)       /* window slot, as in phase A */
    /* (The timing builds that stripped this loop -- no LDS-DMA, no fragment reads, no exponentials, no vector work: wrong
        results by construction, figures in the comment above SV_TILE_SYNC -- are in the history of this file, round 2;  
        the production translation unit carries none of them.) */
#define SV_B_GAP(sub_, i_)                                                                                    \
    {                                                                                                         \
        SV_B_PIN(i_)                                                                                          \
        asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(fb[SV_SL(sub_, i_)]));                                     \
        Z[(i_) >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[(i_) & 1], fb[SV_SL(sub_, i_)], Z[(i_) >> 1], 0, 0, 0); \
        if ((i_) + W < KS1) fb[SV_SL(sub_, i_)] = SV_YB(tst, sub_, ((i_) + W) % KS1);                    \
        else if ((sub_) == 0) fb[SV_SL(sub_, i_)] = SV_YB(tst, 1, ((i_) + W) % KS1);                          \
        else fb[SV_SL(sub_, i_)] = SV_YB(tsn, 0, ((i_) + W) % KS1);                                           \
        if ((i_) >= 1 && (i_) <= 8) {                                                          \
            const float l0_ = c1[2 * (i_) - 2], l1_ = c1[2 * (i_) - 1];                                       \
            pw[(i_) - 1] = pack_bf16(SV_EXP2(l0_), SV_EXP2(l1_));                                             \
            /* one v_max3_f32 (fmaxf() also quiets NaNs: two more instructions per logit) */                  \
            if (MODE != MODE_DW) asm("v_max3_f32 %0, %0, %1, %2" : "+v"(tm) : "v"(l0_), "v"(l1_));           \
        }                                                                                                     \
    }
                    if (sub == 0) {
                        SV_B_GAP(0, 0) SV_B_GAP(0, 1) SV_B_GAP(0, 2) SV_B_GAP(0, 3) SV_B_GAP(0, 4)
                        SV_B_GAP(0, 5) SV_B_GAP(0, 6) SV_B_GAP(0, 7) SV_B_GAP(0, 8) SV_B_GAP(0, 9)
                    } else {
                        SV_B_GAP(1, 0) SV_B_GAP(1, 1) SV_B_GAP(1, 2) SV_B_GAP(1, 3) SV_B_GAP(1, 4)
                        SV_B_GAP(1, 5) SV_B_GAP(1, 6) SV_B_GAP(1, 7) SV_B_GAP(1, 8) SV_B_GAP(1, 9)
                    }
#undef SV_B_GAP
#undef SV_B_PIN
#undef SV_SL
                    if (MODE != MODE_DW && has_next && __any(tm > thr)) {
                        stabilise(c1);
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            pw[q] = pack_bf16(__builtin_amdgcn_exp2f(c1[2 * q]), __builtin_amdgcn_exp2f(c1[2 * q + 1]));
                    }
                    pa[0] = __builtin_bit_cast(bf16x8, (uint4){pw[0], pw[1], pw[2], pw[3]});
                    pa[1] = __builtin_bit_cast(bf16x8, (uint4){pw[4], pw[5], pw[6], pw[7]});
                }
                SV_WSTAMP(2 + 2 * sub)
            }
            SV_TILE_SYNC
            SV_WSTAMP(7)
            sj = sn;
        }
#ifdef SV_PROJ_TIMING
        if (lane == 0 && wave == 0 && blockIdx.x < 4096) {
            g_proj_blk[MODE][blockIdx.x][0] = blk_rt0;
            g_proj_blk[MODE][blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
            g_proj_blk[MODE][blockIdx.x][2] = __builtin_amdgcn_s_memtime() - blk_c0;
            g_proj_blk[MODE][blockIdx.x][3] = (unsigned long long)ntile;
        }
#endif
#undef SV_DMA
#undef SV_YA
#undef SV_YB
#undef SV_TR
#undef SV_MFMA_DS
#undef SV_PIECE1
#undef SV_WAIT_TILES
#undef SV_TILE_SYNC
    }
    } else if constexpr (!SPLIT) {
    // Three LDS stages; tile t lives in stage (t - t_begin) % 3.  An LDS-DMA piece takes 1-2 us from issue to
    // landed under load (about one tile), so a tile's pieces are issued between one and two tiles ahead:
    // pieces 0,1 of tile t+3 behind the barrier of tile t (into tile t's stage: every wave made its last read of
    // it before that barrier), pieces 2.. of tile t+2 in the first half of tile t.
    if (t_begin < t_end) SV_STAGE_ALL(t_begin, 0)
    if (t_begin + 1 < t_end) SV_STAGE_ALL(t_begin + 1, 1)
    if (t_begin + 2 < t_end) {
        const char* yb_ = Yg + (size_t)(t_begin + 2) * (BY * KP * 2);
        const char* tb_ = Tg + (size_t)(t_begin + 2) * (BY * 2);
        char* ly_ = smem + 2 * STAGE_BYTES + wave * 1024;
        if (MODE == MODE_FWD) {
            SV_STAGE_ALL(t_begin + 2, 2)
        } else {
            SV_PIECE(0, yb_, tb_, ly_)
            SV_PIECE(1, yb_, tb_, ly_)
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // Fragment reads run one phase ahead of the MFMAs that use them (an MFMA that waits for a read issued
    // just before it exposes the whole LDS latency): GEMM1 of a sub-tile issues GEMM2's Y^T fragments,
    // GEMM2 issues the next sub-tile's Y fragments.  One ds_read_b128 behind each MFMA.
#define SV_YA(st_, sub_, s_) \
    (*reinterpret_cast<const bf16x8*>((st_) + (sub_) * (32 * KP * 2) + 64 * ((s_) >> 1) + (((s_) & 1) ? ra1 : ra0)))
#define SV_YB(st_, sub_, i_) \
    (*reinterpret_cast<const bf16x8*>((st_) + ((i_) >> 1) * (32 * BY * 2) + rb[2 * (sub_) + ((i_) & 1)]))
#define SV_PAIR                                       \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    bf16x8 fa[KS1], fb[KS1];
#pragma unroll
    for (int s = 0; s < KS1; ++s) fa[s] = SV_YA(smem, 0, s);

#ifdef SV_PROJ_TIMING
    const unsigned long long blk_rt0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long blk_c0 = __builtin_amdgcn_s_memtime();
    unsigned long long ts_acc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long ts_last = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    // (Tried and dropped: putting the tile's barrier at different points of the program for waves 0..3 and
    //  4..7 so the two waves of a SIMD do not run their exp/pack phases together -- no gain, measured.)
    constexpr bool EARLY = false;
    int bufi = 0;
    for (int t = t_begin; t < t_end; ++t) {
        const char* st = smem + bufi * STAGE_BYTES;
        const int bufn = bufi == 2 ? 0 : bufi + 1;          // stage of tile t+1
        const int bufp = bufi == 0 ? 2 : bufi - 1;          // stage of tile t-1 = stage of tile t+2
        const char* stn = smem + bufn * STAGE_BYTES;
        // this iteration's DMA (see above): the rest of tile t+2 into the stage tile t-1 left, then the first
        // pieces of tile t+3 into tile t's own stage once the barrier is behind us
        const bool dma2 = t + 2 < t_end, dma3 = t + 3 < t_end;
        const char* dy2 = Yg + (size_t)(t + 2) * (BY * KP * 2);
        const char* dt2 = Tg + (size_t)(t + 2) * (BY * 2);
        char* dl2 = smem + bufp * STAGE_BYTES + wave * 1024;
        char* dl3 = smem + bufi * STAGE_BYTES + wave * 1024;
#define SV_DMA2(q_) if (dma2 && (q_) < npiece) SV_PIECE(q_, dy2, dt2, dl2)
#define SV_DMA3(q_) if (dma3 && (q_) < npiece) SV_PIECE(q_, dy2 + BY * KP * 2, dt2 + BY * 2, dl3)

        // accumulator rows of a 32x32 tile: y = (r&3) + 8*(r>>2) + 4*lh  -> groups of 4 consecutive y
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            f32x16 c1;
#pragma unroll
            for (int r = 0; r < 16; ++r) c1[r] = 0.f;
#pragma unroll
            for (int s = 0; s < KS1; ++s) {
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s], xf[s], c1, 0, 0, 0);
                if (MODE != MODE_FWD) {
                    fb[s] = SV_YB(st, sub, s);
                    SV_PAIR
                    if (sub == 0 && s == 2) SV_DMA2(2)
                    if (sub == 0 && s == 6) SV_DMA2(3)
                } else {
                    // no GEMM2: fetch the next sub-tile's Y fragments here (the next tile's stage is complete:
                    // FWD puts its barrier in the middle of the tile, see below)
                    fa[s] = sub == 0 ? SV_YA(st, 1, s) : SV_YA(stn, 0, s);
                    SV_PAIR
                    if (sub == 1 && (s == 1 || s == 4 || s == 7)) SV_DMA3(s / 3)
                }
            }
            SV_TS(0)
            if (EARLY && sub == (MODE == MODE_FWD ? 0 : 1)) {
                // The one barrier of the tile.  Behind it tile t+1 has landed everywhere (each wave waits for its
                // own pieces of it; its pieces of tile t+2 stay in flight) and every wave has made its last read
                // of tile t's stage.
                if (!dma2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (wave < 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NIMG) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NIMG) : "memory");
                SV_TS(4)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                SV_TS(5)
            }
            stabilise(c1);
            float p[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) p[r] = __builtin_amdgcn_exp2f(c1[r]);
            bf16x8 pa[2];
            if (MODE == MODE_FWD) {
                float acc = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc += p[r];
                l_run += acc;
            } else {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    uint4 u;
                    u.x = pack_bf16(p[8 * s2 + 0], p[8 * s2 + 1]);
                    u.y = pack_bf16(p[8 * s2 + 2], p[8 * s2 + 3]);
                    u.z = pack_bf16(p[8 * s2 + 4], p[8 * s2 + 5]);
                    u.w = pack_bf16(p[8 * s2 + 6], p[8 * s2 + 7]);
                    pa[s2] = *reinterpret_cast<bf16x8*>(&u);
                }
            }
            SV_TS(1)
            if (!EARLY && sub == (MODE == MODE_FWD ? 0 : 1)) {
                // The one barrier of the tile.  Behind it tile t+1 has landed everywhere (each wave waits for its
                // own pieces of it; its pieces of tile t+2 stay in flight) and every wave has made its last read
                // of tile t's stage.
                if (!dma2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else if (wave < 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NIMG) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NIMG) : "memory");
                SV_TS(4)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                SV_TS(5)
            }
            if (MODE != MODE_FWD) {
                // GEMM2: Z[x][n] += sum_y P[y][x] Y[y][n].  k-order of a converted accumulator: element j of
                // half lh is y = 16 s2 + 8 (j>>2) + 4 lh + (j&3); the Y^T copy stores its columns in that order
                // (proj_prep), so a fragment is one 16-byte read.
#pragma unroll
                for (int i = 0; i < KS1; ++i) {
                    Z[i >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[i & 1], fb[i], Z[i >> 1], 0, 0, 0);
                    fa[i] = sub == 0 ? SV_YA(st, 1, i) : SV_YA(stn, 0, i);
                    SV_PAIR
                    if (sub == 0 && i == 2) SV_DMA2(4)
                    if (sub == 0 && i == 6) SV_DMA2(5)
                    if (sub == 1 && i == 2) SV_DMA3(0)
                    if (sub == 1 && i == 6) SV_DMA3(1)
                }
                SV_TS(3)
            }
        }
        bufi = bufn;
    }
#undef SV_DMA2
#undef SV_DMA3
#ifdef SV_PROJ_TIMING
    if (lane == 0 && wave == 0 && blockIdx.x < 4096) {
        g_proj_blk[MODE][blockIdx.x][0] = blk_rt0;
        g_proj_blk[MODE][blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
        g_proj_blk[MODE][blockIdx.x][2] = __builtin_amdgcn_s_memtime() - blk_c0;
        g_proj_blk[MODE][blockIdx.x][3] = (unsigned long long)(t_end - t_begin);
    }
    if (blockIdx.x == 0 && lane == 0 && (wave == 0 || wave == 4)) {
        for (int i = 0; i < 6; ++i) g_proj_dbg[MODE][wave >> 2][i] = ts_acc[i];
        g_proj_dbg[MODE][wave >> 2][7] = (unsigned long long)(t_end - t_begin);
    }
#endif
    } else {
        // ---- split instantiation: one tile at a time; its hi / mid / lo images in the bytes of stages 0 / 1 / 2
        const char* sm_ = smem + STAGE_BYTES;
        const char* sl_ = smem + 2 * STAGE_BYTES;
        for (int t = t_begin; t < t_end; ++t) {
            __syncthreads();                      // every wave has made its last read of the tile before
#pragma unroll
            for (int part = 0; part < 3; ++part) {
                const char* yb_ = Yg + (size_t)part * a.yps * 2 + (size_t)t * (BY * KP * 2);
                const char* tb_ = Tg + (size_t)part * a.yps * 2 + (size_t)t * (BY * 2);
                char* ly_ = smem + part * STAGE_BYTES + wave * 1024;
#pragma unroll
                for (int q_ = 0; q_ < 3 * NIMG; ++q_)
                    if (q_ < npiece) SV_PIECE(q_, yb_, tb_, ly_)
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
                f32x16 c1;
#pragma unroll
                for (int r = 0; r < 16; ++r) c1[r] = 0.f;
#pragma unroll
                for (int s = 0; s < KS1; ++s) {          // small terms first
                    const bf16x8 yh = SV_YA(smem, sub, s), ym = SV_YA(sm_, sub, s), yl = SV_YA(sl_, sub, s);
                    const int s_ = SPLIT ? s : 0;
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yl, xf[s], c1, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yh, xl[s_], c1, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ym, xm[s_], c1, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ym, xf[s], c1, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yh, xm[s_], c1, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yh, xf[s], c1, 0, 0, 0);
                }
                stabilise(c1);
                float p[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) p[r] = __builtin_amdgcn_exp2f(c1[r]);
                if (MODE == MODE_FWD) {
                    float acc = 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc += p[r];
                    l_run += acc;
                } else {
                    bf16x8 ph[2], pm[2], pl[2];
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        unsigned hw[4], mw[4], lw[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float p0 = p[8 * s2 + 2 * i], p1 = p[8 * s2 + 2 * i + 1];
                            hw[i] = pack_bf16(p0, p1);
                            const float r0 = p0 - __uint_as_float(hw[i] << 16), r1 = p1 - __uint_as_float(hw[i] & 0xFFFF0000u);
                            mw[i] = pack_bf16(r0, r1);
                            lw[i] = pack_lo(r0, r1, mw[i]);
                        }
                        ph[s2] = __builtin_bit_cast(bf16x8, (uint4){hw[0], hw[1], hw[2], hw[3]});
                        pm[s2] = __builtin_bit_cast(bf16x8, (uint4){mw[0], mw[1], mw[2], mw[3]});
                        pl[s2] = __builtin_bit_cast(bf16x8, (uint4){lw[0], lw[1], lw[2], lw[3]});
                    }
#pragma unroll
                    for (int i = 0; i < KS1; ++i) {
                        const bf16x8 th = SV_YB(smem, sub, i), tm = SV_YB(sm_, sub, i), tl = SV_YB(sl_, sub, i);
                        f32x16& z = Z[i >> 1];
                        z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pl[i & 1], th, z, 0, 0, 0);
                        z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph[i & 1], tl, z, 0, 0, 0);
                        z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pm[i & 1], tm, z, 0, 0, 0);
                        z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pm[i & 1], th, z, 0, 0, 0);
                        z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph[i & 1], tm, z, 0, 0, 0);
                        z = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ph[i & 1], th, z, 0, 0, 0);
                    }
                }
            }
        }
    }
#undef SV_YA
#undef SV_YB
#undef SV_PAIR
#undef SV_STAGE_ALL
#undef SV_PIECE

    const bool any_tile = t_begin < t_end;
    if (MODE == MODE_FWD) {
        // the two lane halves hold the same x (and the same m), different y
        const float l = l_run + __shfl_xor(l_run, 32, 64);
        if (lh == 0 && xok)
            reinterpret_cast<float2*>(a.out)[(size_t)split * a.nx + x] = make_float2(any_tile ? m_eff : NEG_BIG, l);
    } else {
        // Z tile layout: col = li -> n, row = (r&3) + 8*(r>>2) + 4*lh -> x
        float* slab = a.out + (size_t)split * a.nx * KP;
#pragma unroll
        for (int n = 0; n < NT2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int xr = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (xr < a.nx) slab[(size_t)xr * KP + 32 * n + li] = Z[n][r];
            }
        if (MODE == MODE_FDH && lh == 0 && xok) a.mpart[(size_t)split * a.nx + x] = any_tile ? m_eff : NEG_BIG;
    }
    }   // segments
#ifdef SV_PROJ_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && wave == 0 && blockIdx.x < 4096) {
        g_proj_blk2[MODE][blockIdx.x][0] = blk_entry;
        g_proj_blk2[MODE][blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// operand preparation
// ------------------------------------------------------------------------------------------------
// Column position of y inside the Y^T copy: GEMM2's k-order swaps bits 2 and 3 of the index.
__device__ __forceinline__ int korder(int y) { return (y & ~12) | ((y & 4) << 1) | ((y & 8) >> 1); }

// fp32 [n][160] -> the three bf16 copies (plain [n_pad][160], scaled by log2e [n_pad][160], transposed
// [160][ldt]) with the constant columns of the file header.  ROWS = 1: decoder hidden rows (h2b, h2s, h2tb);
// ROWS = 0: items (w3b, w3s, w3tb), bias = b3.
template <int ROWS>
__global__ void proj_prep_kernel(const float* __restrict__ src, const float* __restrict__ bias,
                                 unsigned short* __restrict__ dst, unsigned short* __restrict__ dsts,
                                 unsigned short* __restrict__ dstT, int n, int n_pad, int ldt, int row0,
                                 long long ps) {
    // ps > 0 (split instantiation): also the mid and lo parts of the three copies, ps and 2 ps elements behind each
    // -- RNE(v - hi), RNE(v - hi - mid) in the data columns; zero in the constant columns except the THIRD part of
    // the bias (column 151 of the items' mid images)
    __shared__ unsigned short tile[32][KP + 2];
    __shared__ unsigned short tile_m[32][KP + 2];
    __shared__ unsigned short tile_l[32][KP + 2];
    const int r0 = row0 + blockIdx.x * 32;
    const unsigned short one = 0x3F80;
    for (int i = threadIdx.x; i < 32 * KP; i += blockDim.x) {
        const int r = i / KP, c = i % KP;
        const int row = r0 + r;
        const bool live = row < n;
        unsigned short hp = 0, hs = 0;          // plain copy, scaled copy
        unsigned short mp = 0, ms = 0;          // their mid parts
        unsigned short lp = 0, ls = 0;          // their lo parts
        if (c < SV_D) {
            if (live) {
                const float v = src[(size_t)row * KP + c];
                hp = sv_f2bf(v);
                hs = sv_f2bf(v * LOG2E);
                const float rp = v - sv_bf2f(hp), rs = v * LOG2E - sv_bf2f(hs);
                mp = sv_f2bf(rp);
                ms = sv_f2bf(rs);
                lp = sv_f2bf(rp - sv_bf2f(mp));
                ls = sv_f2bf(rs - sv_bf2f(ms));
            }
        } else if (ROWS) {
            if (c == C_ONE) hp = live ? one : 0;
            if (c == C_BH || c == C_BL) { hp = live ? one : 0; hs = live ? one : 0; }
            if (c == C_RH) hp = live ? 0 : sv_f2bf(NEG_BIG);      // live rows: written by the forward finish
        } else {
            if (c == C_BH || c == C_BL) {
                if (live) {
                    const float b = bias[row] * LOG2E;
                    const unsigned short hi = sv_f2bf(b);
                    const unsigned short lo = sv_f2bf(b - sv_bf2f(hi));
                    hp = hs = (c == C_BH) ? hi : lo;
                    if (c == C_BH) mp = ms = sv_f2bf((b - sv_bf2f(hi)) - sv_bf2f(lo));
                } else if (c == C_BH) {
                    hp = sv_f2bf(NEG_BIG);
                }
            }
            if (c == C_RH || c == C_RL) hs = live ? one : 0;
            if (c == C_MH || c == C_ML) hp = one;
        }
        tile[r][c] = hp;
        tile_m[r][c] = mp;
        tile_l[r][c] = lp;
        if (row < n_pad) {
            dst[(size_t)row * KP + c] = hp;
            dsts[(size_t)row * KP + c] = hs;
            if (ps > 0) {
                dst[ps + (size_t)row * KP + c] = mp;
                dsts[ps + (size_t)row * KP + c] = ms;
                dst[2 * ps + (size_t)row * KP + c] = lp;
                dsts[2 * ps + (size_t)row * KP + c] = ls;
            }
        }
    }
    if (dstT == nullptr) return;          // (uniform) bf16 instantiation: the sweeps transpose on the LDS read
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * KP; i += blockDim.x) {
        const int c = i / 32, r = i % 32;
        if (r0 + r < ldt) {
            dstT[(size_t)c * ldt + korder(r0 + r)] = tile[r][c];
            if (ps > 0) {
                dstT[ps + (size_t)c * ldt + korder(r0 + r)] = tile_m[r][c];
                dstT[2 * ps + (size_t)c * ldt + korder(r0 + r)] = tile_l[r][c];
            }
        }
    }
}

// Loss-only forward finish, one wave per row: lse from the NSPLIT partials (log2 units), target logits by fp32
// dot products, nll_row.
__global__ void proj_fwd_finish_kernel(const float2* __restrict__ part, const float* __restrict__ h2,
                                       const float* __restrict__ w3, const float* __restrict__ b3,
                                       const int* __restrict__ tgt, int R, int K, float* __restrict__ nll_row,
                                       float* __restrict__ lse_out, int ns_q, int ns_rem) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= R) return;
    const int ns = ns_q + ((row / BX) < ns_rem ? 1 : 0);
    float m = NEG_BIG, l = 0.f;
    if (lane < ns) {
        const float2 p = part[(size_t)lane * R + row];
        m = p.x; l = p.y;
    }
    const float M = sv_wave_max(m);
    l = sv_wave_sum(l * exp2f(m - M));
    const float lse = (M + log2f(l)) * LN2;
    float nll = 0.f;
    const float* hr = h2 + (size_t)row * KP;
    for (int k = 0; k < K; ++k) {
        const int it = tgt[(size_t)row * K + k];     // wave-uniform
        if (it < 0) continue;
        const float* wr = w3 + (size_t)it * KP;
        float d = 0.f;
        for (int c = lane; c < SV_D; c += 64) d += hr[c] * wr[c];
        d = sv_wave_sum(d) + b3[it];
        nll += lse - d;
    }
    if (lane == 0) {
        nll_row[row] = nll;
        lse_out[row] = lse;
    }
}

// FDH finish, one wave per row (float4 lanes 0..39): combine the NSPLIT (m, Z) partials, lse and NLL, the
// target part of dL/dh2, tanh', and the row constant log2(c_r) - lse2_r of the weight-gradient kernel
// (bf16 hi/lo into columns 153/154 of h2b).
__global__ void proj_fdh_finish_kernel(const float* __restrict__ slabs, const float* __restrict__ mpart,
                                       const float* __restrict__ h2, const float* __restrict__ w3,
                                       const float* __restrict__ b3, const int* __restrict__ tgt,
                                       const float* __restrict__ w_nll, int R, int K, float* __restrict__ dd1,
                                       float* __restrict__ nll_row, float* __restrict__ lse_out,
                                       unsigned short* __restrict__ h2b, int ns_q, int ns_rem,
                                       unsigned short* __restrict__ h2b_lo, const int* __restrict__ ns_tab) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= R) return;
    const int ln = lane < KP / 4 ? lane : 0;
    const int ns = ns_tab ? ns_tab[row / BX] : ns_q + ((row / BX) < ns_rem ? 1 : 0);     // wave-uniform
    // everything that depends on the row number only goes out first (K == 4: the targets in one 16-byte load)
    const int4 t4 = K == 4 ? *reinterpret_cast<const int4*>(tgt + (size_t)row * 4) : make_int4(-1, -1, -1, -1);
    const float w = w_nll[row];
    const float4 hv = reinterpret_cast<const float4*>(h2 + (size_t)row * KP)[ln];
    float ms[NS_MAX];
    float M = NEG_BIG;
#pragma unroll
    for (int z = 0; z < NS_MAX; ++z) {
        ms[z] = z < ns ? mpart[(size_t)z * R + row] : NEG_BIG;
        M = fmaxf(M, ms[z]);
    }
    float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int z0 = 0; z0 < NS_MAX; z0 += 8) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int z = z0 + j < ns ? z0 + j : 0;
            v[j] = reinterpret_cast<const float4*>(slabs + ((size_t)z * R + row) * KP)[ln];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float f = z0 + j < ns ? exp2f(ms[z0 + j] - M) : 0.f;     // fixed order: reproducible
            u.x += f * v[j].x; u.y += f * v[j].y; u.z += f * v[j].z; u.w += f * v[j].w;
        }
        if (ns <= 8) break;
    }
    const float l = __shfl(u.w, C_MH / 4, 64);            // column 155 = sum of P
    const float lse2 = M + log2f(l);
    const float lse = lse2 * LN2;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float nll = 0.f;
    int nr = 0;
    if (K == 4) {
        // next_k = 4 (every configuration of the reference): straight-line code, so that the row's targets, its h2 row
        // and weight go out with the slab loads above and the four W3 rows one round trip behind them.  The general
        // loop below is a chain of dependent loads (target -> W3 row -> reduce, four times, behind the slabs), and at
        // 8 waves per SIMD and 2.7 rounds of workgroups that chain WAS the kernel: 24 us for 60 MB.  Same operations
        // in the same order: bit-identical.
        const float4 wv[4] = {reinterpret_cast<const float4*>(w3 + (size_t)(t4.x < 0 ? 0 : t4.x) * KP)[ln],
                              reinterpret_cast<const float4*>(w3 + (size_t)(t4.y < 0 ? 0 : t4.y) * KP)[ln],
                              reinterpret_cast<const float4*>(w3 + (size_t)(t4.z < 0 ? 0 : t4.z) * KP)[ln],
                              reinterpret_cast<const float4*>(w3 + (size_t)(t4.w < 0 ? 0 : t4.w) * KP)[ln]};
        const float bb[4] = {b3[t4.x < 0 ? 0 : t4.x], b3[t4.y < 0 ? 0 : t4.y], b3[t4.z < 0 ? 0 : t4.z], b3[t4.w < 0 ? 0 : t4.w]};
        const int its[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float d = lane < KP / 4 ? hv.x * wv[k].x + hv.y * wv[k].y + hv.z * wv[k].z + hv.w * wv[k].w : 0.f;   // pads are zero
            d = sv_wave_sum(d) + bb[k];
            if (its[k] >= 0) {
                nll += lse - d;
                ++nr;
                acc.x += wv[k].x; acc.y += wv[k].y; acc.z += wv[k].z; acc.w += wv[k].w;
            }
        }
    } else
    for (int k = 0; k < K; ++k) {
        const int it = tgt[(size_t)row * K + k];     // wave-uniform
        if (it < 0) continue;
        const float4 wv = reinterpret_cast<const float4*>(w3 + (size_t)it * KP)[ln];
        float d = lane < KP / 4 ? hv.x * wv.x + hv.y * wv.y + hv.z * wv.z + hv.w * wv.w : 0.f;   // pads are zero
        d = sv_wave_sum(d) + b3[it];
        nll += lse - d;
        ++nr;
        acc.x += wv.x; acc.y += wv.y; acc.z += wv.z; acc.w += wv.w;
    }
    const float c = (float)nr * w;
    const float sc = c / l;
    if (lane < KP / 4) {
        float4 o;
        const int c0 = 4 * lane;
        o.x = c0 + 0 < SV_D ? (sc * u.x - w * acc.x) * (1.0f - hv.x * hv.x) : 0.f;
        o.y = c0 + 1 < SV_D ? (sc * u.y - w * acc.y) * (1.0f - hv.y * hv.y) : 0.f;
        o.z = c0 + 2 < SV_D ? (sc * u.z - w * acc.z) * (1.0f - hv.z * hv.z) : 0.f;
        o.w = c0 + 3 < SV_D ? (sc * u.w - w * acc.w) * (1.0f - hv.w * hv.w) : 0.f;
        reinterpret_cast<float4*>(dd1 + (size_t)row * KP)[lane] = o;
    }
    if (lane == 0) {
        nll_row[row] = nll;
        lse_out[row] = lse;
        const float rc = nr > 0 ? log2f(c) - lse2 : NEG_BIG;
        const unsigned short hi = sv_f2bf(rc);
        const unsigned short lo = sv_f2bf(rc - sv_bf2f(hi));
        h2b[(size_t)row * KP + C_RH] = hi;
        h2b[(size_t)row * KP + C_RL] = lo;
        // split instantiation: the third part rides in the MID image (against the ones column of w3s's hi image)
        if (h2b_lo) h2b_lo[(size_t)row * KP + C_RH] = nr > 0 ? sv_f2bf((rc - sv_bf2f(hi)) - sv_bf2f(lo)) : (unsigned short)0;
    }
}

// dL/dW3, dL/db3 finish.  Per item: sum the NSPLIT slabs, subtract the target part sum_r w_r h2[r] over the
// item's target occurrences (pairs sorted by item on the host: occ_row lists the data rows), write the row.
// Column 150 of the slab sum is the bias gradient (ones column), its target part is -sum w_r.
// Occurrence counts are heavily skewed, so the host lists the items by count (occ_items): the first `nlong`
// (more than OCC_LONG occurrences) get a workgroup each -- occurrences split over the 4 waves, 4 in flight per
// wave, partial sums added in a fixed order (bitwise reproducible) -- the others one wave each.
constexpr int OCC_LONG = 32;

__device__ __forceinline__ void dw_occ_sum(const float* __restrict__ h2, const float* __restrict__ w_nll,
                                           const int* __restrict__ occ_row, int b, int e, int first, int step, int ln,
                                           float4& s, float& wsum) {
    for (int o = b + first; o < e; o += 4 * step) {
        int r[4];
        float w[4];
        float4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = (o + j * step < e) ? occ_row[o + j * step] : -1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int rr = r[j] < 0 ? 0 : r[j];
            w[j] = r[j] < 0 ? 0.f : w_nll[rr];
            v[j] = reinterpret_cast<const float4*>(h2 + (size_t)rr * KP)[ln];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s.x -= w[j] * v[j].x; s.y -= w[j] * v[j].y; s.z -= w[j] * v[j].z; s.w -= w[j] * v[j].w;
            wsum += w[j];
        }
    }
}

__device__ __forceinline__ void dw_write_row(float4 t, float wsum, int it, int lane, float* __restrict__ gw3,
                                             float* __restrict__ gb3) {
    if (lane == SV_D / 4) {          // lane 37 holds columns 148..151: column 150 = bias gradient
        gb3[it] = t.z - wsum;
        t.z = 0.f;
    }
    const int c0 = 4 * lane;
    if (c0 + 0 >= SV_D) t.x = 0.f;
    if (c0 + 1 >= SV_D) t.y = 0.f;
    if (c0 + 2 >= SV_D) t.z = 0.f;
    if (c0 + 3 >= SV_D) t.w = 0.f;
    reinterpret_cast<float4*>(gw3 + (size_t)it * KP)[lane] = t;
}

constexpr int DWL_WAVES = 16;   // the busiest item has ~ k * B occurrences: 16 waves x 4 in flight walk it in ~16 rounds
__global__ __launch_bounds__(64 * DWL_WAVES) void proj_dw_finish_long_kernel(const float* __restrict__ slabs,
                                                                  const float* __restrict__ h2,
                                                                  const float* __restrict__ w_nll,
                                                                  const int* __restrict__ occ_off,
                                                                  const int* __restrict__ occ_row,
                                                                  const int* __restrict__ occ_items,
                                                                  float* __restrict__ gw3, float* __restrict__ gb3, int I,
                                                                  int ns_q, int ns_rem, const int* __restrict__ ns_tab) {
    __shared__ float4 red[DWL_WAVES][KP / 4];
    __shared__ float wred[DWL_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int it = occ_items[blockIdx.x];
    const int ln = lane < KP / 4 ? lane : 0;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    float wsum = 0.f;
    if (wave == 0) {
        const int ns = ns_tab ? ns_tab[it / BX] : ns_q + ((it / BX) < ns_rem ? 1 : 0);
        for (int z = 0; z < ns; ++z) {
            const float4 v = reinterpret_cast<const float4*>(slabs + ((size_t)z * I + it) * KP)[ln];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    dw_occ_sum(h2, w_nll, occ_row, occ_off[it], occ_off[it + 1], wave, DWL_WAVES, ln, s, wsum);
    if (lane < KP / 4) red[wave][lane] = s;
    if (lane == 0) wred[wave] = wsum;
    __syncthreads();
    if (wave != 0 || lane >= KP / 4) return;
    float4 t = red[0][lane];
    float wt = wred[0];
#pragma unroll
    for (int q = 1; q < DWL_WAVES; ++q) {
        const float4 u = red[q][lane];
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        wt += wred[q];
    }
    dw_write_row(t, wt, it, lane, gw3, gb3);
}

// one wave per entry j of [nlong, I_rows): entries below I are items (occ_items[j]), the rest pad rows (zeroed)
__global__ __launch_bounds__(256) void proj_dw_finish_kernel(const float* __restrict__ slabs,
                                                             const float* __restrict__ h2,
                                                             const float* __restrict__ w_nll,
                                                             const int* __restrict__ occ_off,
                                                             const int* __restrict__ occ_row,
                                                             const int* __restrict__ occ_items, float* __restrict__ gw3,
                                                             float* __restrict__ gb3, int nlong, int I, int I_rows,
                                                             int ns_q, int ns_rem, const int* __restrict__ ns_tab) {
    const int lane = threadIdx.x & 63;
    const int j = nlong + blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= I_rows || lane >= KP / 4) return;
    if (j >= I) {
        reinterpret_cast<float4*>(gw3 + (size_t)j * KP)[lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane == 0) gb3[j] = 0.f;
        return;
    }
    const int it = occ_items[j];
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    float wsum = 0.f;
    const int ns = ns_tab ? ns_tab[it / BX] : ns_q + ((it / BX) < ns_rem ? 1 : 0);      // wave-uniform
#pragma unroll
    for (int z0 = 0; z0 < NS_MAX; z0 += 8) {
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int z = z0 + j < ns ? z0 + j : 0;
            v[j] = reinterpret_cast<const float4*>(slabs + ((size_t)z * I + it) * KP)[lane];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (z0 + j < ns) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
        if (ns <= 8) break;
    }
    dw_occ_sum(h2, w_nll, occ_row, occ_off[it], occ_off[it + 1], 0, 1, lane, s, wsum);
    dw_write_row(s, wsum, it, lane, gw3, gb3);
}

// Optional timing of the NEXT sweep launch of this thread: the start / stop events are attached to the dispatch itself
// (hipExtLaunchKernelGGL), so they take the kernel's own begin and end timestamps and put no marker packets into the
// queue.  (Bracketing a launch with hipEventRecord cost the host 150 us per record once kernels were in flight --
// 600 us in the first GPU process of a freshly booted box -- which made a bracketed step host-bound.)
thread_local hipEvent_t g_time_start = nullptr, g_time_stop = nullptr;

template <int MODE, bool SPLIT = false>
int launch_proj(ProjArgs a, int nwg, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)proj_kernel<MODE, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return svae_fail((int)e, "svae_proj: LDS attribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    if (g_time_start != nullptr && g_time_stop != nullptr) {
        hipExtLaunchKernelGGL((proj_kernel<MODE, SPLIT>), dim3(nwg), dim3(512), LDS_BYTES, st, g_time_start, g_time_stop, 0, a);
        g_time_start = g_time_stop = nullptr;
    } else {
        hipLaunchKernelGGL((proj_kernel<MODE, SPLIT>), dim3(nwg), dim3(512), LDS_BYTES, st, a);
    }
    SV_LAUNCH_CHECK();
    return 0;
}

}  // namespace

#ifdef SV_PROJ_TIMING
extern "C" int svae_debug_proj(unsigned long long* out48) {
    return (int)hipMemcpyFromSymbol(out48, HIP_SYMBOL(g_proj_dbg), 48 * sizeof(unsigned long long));
}
extern "C" int svae_debug_proj_waves(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_proj_wv), sizeof(unsigned long long) * 3 * 8 * 8 * 8);
}
extern "C" int svae_debug_proj_blocks2(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_proj_blk2), 3 * 4096 * 2 * sizeof(unsigned long long));
}
extern "C" int svae_debug_proj_blocks(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_proj_blk), 3 * 4096 * 4 * sizeof(unsigned long long));
}
#endif
extern "C" int svae_proj_nsplit(void) { return NSPLIT; }

// Timing hooks (measurement only; no counterpart in the reference).  svae_proj_time_next arms the calling thread's next
// sweep launch (svae_proj_nll_fwd / _fwd_dh / _bwd_dw, part 1) with a start and a stop event created by
// svae_event_create; svae_event_elapsed_ms reads the kernel's duration once it has finished.
extern "C" int svae_event_create(void** out) {
    if (!out) return svae_fail(-1, "svae_event_create: null pointer");
    hipEvent_t e = nullptr;
    hipError_t rc = hipEventCreate(&e);
    if (rc != hipSuccess) return svae_fail((int)rc, "svae_event_create: %s", hipGetErrorString(rc));
    *out = (void*)e;
    return 0;
}
// Ordering events of the step's streams (host helpers, no counterpart in the reference): an event without timing,
// its record on a stream and a stream's wait for it -- the three HIP calls, without the 10-14 us of host time that
// the torch wrappers of the same calls cost (a step records 9 and waits for 11: 0.16 of its 0.45 ms of host time).
extern "C" int svae_sync_event_create(void** out) {
    if (!out) return svae_fail(-1, "svae_sync_event_create: null pointer");
    hipEvent_t e = nullptr;
    hipError_t rc = hipEventCreateWithFlags(&e, hipEventDisableTiming);
    if (rc != hipSuccess) return svae_fail((int)rc, "svae_sync_event_create: %s", hipGetErrorString(rc));
    *out = (void*)e;
    return 0;
}
extern "C" int svae_event_record(void* e, void* stream) {
    hipError_t rc = hipEventRecord((hipEvent_t)e, (hipStream_t)stream);
    if (rc != hipSuccess) return svae_fail((int)rc, "svae_event_record: %s", hipGetErrorString(rc));
    return 0;
}
extern "C" int svae_stream_wait_event(void* stream, void* e) {
    hipError_t rc = hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)e, 0);
    if (rc != hipSuccess) return svae_fail((int)rc, "svae_stream_wait_event: %s", hipGetErrorString(rc));
    return 0;
}
extern "C" int svae_event_destroy(void* e) {
    hipError_t rc = hipEventDestroy((hipEvent_t)e);
    if (rc != hipSuccess) return svae_fail((int)rc, "svae_event_destroy: %s", hipGetErrorString(rc));
    return 0;
}
extern "C" int svae_event_elapsed_ms(void* start, void* stop, float* ms) {
    if (!start || !stop || !ms) return svae_fail(-1, "svae_event_elapsed_ms: null pointer");
    hipError_t rc = hipEventSynchronize((hipEvent_t)stop);
    if (rc == hipSuccess) rc = hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
    if (rc != hipSuccess) return svae_fail((int)rc, "svae_event_elapsed_ms: %s", hipGetErrorString(rc));
    return 0;
}
extern "C" int svae_proj_time_next(void* start, void* stop) {
    g_time_start = (hipEvent_t)start;
    g_time_stop = (hipEvent_t)stop;
    return 0;
}
extern "C" int svae_proj_ns_max(void) { return NS_MAX; }

// Work division of one projection sweep (HOST function, no GPU work): nx resident rows in x tiles of 256, ny
// streamed rows in y tiles of 64, ncu compute units.  uniform = 0: the workgroup count is the whole multiple of
// ncu that the nominal 8-way split rounds up to; x tile j gets ns_j = q + (j < rem) splits (capped at 16 and at
// the number of y tiles), split s of it the y tiles [s nt / ns_j, (s+1) nt / ns_j).  uniform = 1: see below.  sched (host memory, 4 ints per
// workgroup, room for cap workgroups; cap >= 16 * ceil(nx/256) always suffices) receives (x tile, first y tile,
// end y tile, slab) in launch order: sorted by where the split starts in Y and dealt to the XCDs (id % 8) in
// contiguous runs, so the workgroups resident on one XCD at a time stream neighbouring parts of Y.
// Returns the number of workgroups (> 0) or a negative error; *ns_q, *ns_rem as above.
extern "C" int svae_proj_schedule(int nx, int ny, int ncu, int uniform, int* sched, int cap, int* ns_q, int* ns_rem) {
    SV_CHECK_ARG(nx > 0 && ny > 0 && ncu > 0 && sched && ns_q && ns_rem, "svae_proj_schedule: bad arguments");
    const int xt = (nx + BX - 1) / BX, nt = (ny + BY - 1) / BY;
    const int lim = nt < NS_MAX ? nt : NS_MAX;
    long long W;
    int q, rem = 0;
    if (uniform > 0) {
        // every x tile split the same number of ways (the nominal 8, or uniform - 1 when uniform > 1): all
        // workgroups the same length, so the rounds stay aligned and the workgroups of an XCD walk Y in lockstep.
        // Needed when a split's part of Y (40 KiB per y tile) does not fit the XCD's 4 MiB L2: a workgroup that
        // starts late then re-fetches it from HBM -- measured on the weight-gradient sweep: +16 % with mixed split
        // counts at 51k rows.  Other split counts measured there (ML-20M shape, beside the GRU's BPTT on 64 CUs):
        // 3: +31 %, 4: +10 %, 6: +12 %, 9: +10 %, 12: +10 %, 16: +15 % (more slabs, or workgroups too long to pack).
        q = uniform > 1 ? uniform - 1 : NSPLIT;
        if (q > lim) q = lim;
        W = (long long)q * xt;
    } else {
        // uniform < 0: nominal split count -uniform instead of 8 (fewer, longer workgroups: fewer slabs)
        const long long nominal = (long long)(uniform < 0 ? -uniform : NSPLIT) * xt;
        const long long rounds = (nominal + ncu - 1) / ncu;
        W = rounds * ncu;
        q = (int)(W / xt), rem = (int)(W % xt);
        if (q >= lim) { q = lim; rem = 0; }
        W = (long long)q * xt + rem;
    }
    SV_CHECK_ARG(W <= cap, "svae_proj_schedule: schedule of %lld workgroups does not fit cap %d", W, cap);
    // sort key: start of the split as a fraction of Y, s / ns_j, compared exactly as s * ns_other
    struct Ent { int j, s, ns; };
    Ent* e = new Ent[W];
    long long n = 0;
    for (int j = 0; j < xt; ++j) {
        const int ns = q + (j < rem ? 1 : 0);
        for (int s2 = 0; s2 < ns; ++s2) e[n++] = Ent{j, s2, ns};
    }
    std::stable_sort(e, e + n, [](const Ent& a, const Ent& b) {
        const long long l = (long long)a.s * b.ns, r = (long long)b.s * a.ns;
        return l != r ? l < r : a.j < b.j;
    });
    long long p = 0;
    for (int c = 0; c < 8; ++c)
        for (long long id = c; id < n; id += 8, ++p) {
            const Ent& t = e[p];
            int* o = sched + 4 * id;
            o[0] = t.j;
            o[1] = (int)((long long)t.s * nt / t.ns);
            o[2] = (int)((long long)(t.s + 1) * nt / t.ns);
            o[3] = t.s;
        }
    delete[] e;
    *ns_q = q;
    *ns_rem = rem;
    return (int)n;
}
// FLAT work division (round 3; HOST function, no GPU work).  The (x tile, y tile) space of a sweep, x-major, is cut
// into nwg contiguous ranges of equal COST, one per workgroup: a range that crosses x-tile boundaries is several
// segments, which the workgroup runs one after the other (proj_kernel's segment loop); every segment after a
// workgroup's first costs `boundary` tiles' worth of time on top of its tiles (slab out, rows in, the Y pipeline
// primed again: ~8 us = 5 tiles measured).  The balanced form above cuts every x tile into q or q + 1 equal pieces:
// its longest piece is 7 % above the mean over the bench's plans and 23 % in single launches, and that is what the
// launch then takes.  nwg is the caller's choice (<= the CUs it wants the sweep on; capped so that no x tile gets more
// than svae_proj_ns_max() slabs); every workgroup gets at least one tile.
// Launch order: ranges are handed to workgroup ids so that the 32 ids of an XCD (id % 8) hold ranges whose position
// in Y (start mod ny tiles) is adjacent -- ranges are equally long and advance at the same pace, so they stay
// neighbours in Y for their whole length (wrapping together), and what one fetches from HBM the others find in L2.
// Table layout (ints): 4 per segment (x tile, first y tile, end y tile, slab = ordinal of the segment within its x
// tile) for *nseg segments in workgroup order; then wg_off [nwg + 1] padded to a multiple of 4 ints; then ns_tab
// [x tiles] (slabs per x tile) padded likewise.  Returns the table's length in int4 units (> 0) or a negative error;
// the sweeps take it as (sched, nwg, ns_q = 0, ns_rem = *nseg).
extern "C" int svae_proj_schedule_flat(int nx, int ny, int nwg_want, int boundary, int* sched, int cap4, int* nwg_out,
                                       int* nseg_out) {
    SV_CHECK_ARG(nx > 0 && ny > 0 && nwg_want > 0 && boundary >= 0 && sched && nwg_out && nseg_out,
                 "svae_proj_schedule_flat: bad arguments");
    const int xt = (nx + BX - 1) / BX, nt = (ny + BY - 1) / BY;
    const long long total = (long long)xt * nt;
    long long W = nwg_want;
    // at most NS_MAX segments per x tile: a range must be at least nt / (NS_MAX - 1) tiles long
    const long long lmin = (nt + NS_MAX - 2) / (NS_MAX - 1);
    if (W > total / lmin) W = total / lmin;
    if (W > total) W = total;
    if (W < 1) W = 1;
    // greedy fill for a cost limit C: tiles until the limit is reached; binary search for the smallest C that needs <= W ranges
    auto ranges_for = [&](double C, long long* ends) -> long long {
        long long pos = 0, n = 0;
        while (pos < total) {
            double cost = 0;
            long long p = pos;
            while (p < total) {
                const long long to_edge = nt - (p % nt);                 // tiles to the end of this x tile
                const double extra = p > pos ? (double)boundary : 0.0;     // a further segment
                if (cost + extra + 1 > C && p > pos) break;
                long long take = (long long)(C - cost - extra);
                if (take < 1) take = 1;
                if (take > to_edge) take = to_edge;
                cost += extra + (double)take;
                p += take;
                if (take < to_edge) break;
            }
            if (ends && n < W) ends[n] = p;
            ++n;
            pos = p;
        }
        return n;
    };
    double lo = (double)total / (double)W, hi = (double)total / (double)W + (double)boundary * (nt > 0 ? (double)total / nt / W + 2 : 2) + 2;
    while (ranges_for(hi, nullptr) > W) hi *= 1.5;
    for (int it = 0; it < 40; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (ranges_for(mid, nullptr) <= W) hi = mid; else lo = mid;
    }
    long long* ends = new long long[W];
    const long long nr = ranges_for(hi, ends);
    // (fewer ranges than W can come out when the limit is loose: that is then the workgroup count)
    struct Rng { long long b, e; };
    Rng* r = new Rng[nr];
    for (long long i = 0; i < nr; ++i) r[i] = Rng{i ? ends[i - 1] : 0, ends[i]};
    // workgroup order: by position in Y of the range's start, dealt to the XCDs in contiguous runs
    long long* ord = new long long[nr];
    for (long long i = 0; i < nr; ++i) ord[i] = i;
    std::stable_sort(ord, ord + nr, [&](long long x, long long y) { return r[x].b % nt < r[y].b % nt; });
    long long* of_wg = new long long[nr];
    {
        long long p = 0;
        for (int c = 0; c < 8; ++c)
            for (long long id = c; id < nr; id += 8, ++p) of_wg[id] = ord[p];
    }
    // slab ordinals: segments of an x tile numbered in flat order
    long long nseg = 0;
    for (long long i = 0; i < nr; ++i) nseg += (r[i].e - 1) / nt - r[i].b / nt + 1;
    const long long need4 = nseg + (nr + 1 + 3) / 4 + (xt + 3) / 4;
    if (need4 > cap4) {
        delete[] ends; delete[] r; delete[] ord; delete[] of_wg;
        return svae_fail(-2, "svae_proj_schedule_flat: table of %lld int4 does not fit cap %d", need4, cap4);
    }
    int* ns_tab = sched + 4 * nseg + 4 * ((nr + 1 + 3) / 4);
    int* wg_off = sched + 4 * nseg;
    for (int j = 0; j < xt; ++j) ns_tab[j] = 0;
    // slab ordinal of the segment of range i in x tile j = number of ranges that start in x tile j before it (+1 if the
    // x tile is entered from the range before): computed by a walk in flat order
    int* ordinal_first = new int[nr];          // slab ordinal of each range's FIRST segment
    {
        for (long long i = 0; i < nr; ++i) {
            const long long j0 = r[i].b / nt, j1 = (r[i].e - 1) / nt;
            ordinal_first[i] = ns_tab[j0];
            for (long long j = j0; j <= j1; ++j) ns_tab[j] += 1;
        }
    }
    long long sidx = 0;
    for (long long id = 0; id < nr; ++id) {
        const long long i = of_wg[id];
        wg_off[id] = (int)sidx;
        const long long j0 = r[i].b / nt, j1 = (r[i].e - 1) / nt;
        for (long long j = j0; j <= j1; ++j) {
            const long long b = j == j0 ? r[i].b - j * nt : 0, e = j == j1 ? r[i].e - j * nt : nt;
            int* o = sched + 4 * sidx++;
            o[0] = (int)j;
            o[1] = (int)b;
            o[2] = (int)e;
            o[3] = j == j0 ? ordinal_first[i] : 0;     // a range enters every further x tile at its first y tile
        }
    }
    wg_off[nr] = (int)sidx;
    for (long long k = nr + 1; k < 4 * ((nr + 1 + 3) / 4); ++k) wg_off[k] = (int)sidx;
    for (int k = xt; k < 4 * ((xt + 3) / 4); ++k) ns_tab[k] = 0;
    bool ok = sidx == nseg;
    for (int j = 0; j < xt && ok; ++j) ok = ns_tab[j] >= 1 && ns_tab[j] <= NS_MAX;
    delete[] ends; delete[] r; delete[] ord; delete[] of_wg; delete[] ordinal_first;
    if (!ok) return svae_fail(-3, "svae_proj_schedule_flat: internal error (slab counts)");
    *nwg_out = (int)nr;
    *nseg_out = (int)nseg;
    return (int)need4;
}
extern "C" int svae_proj_tile(void) { return BY; }
extern "C" int svae_proj_occ_long(void) { return OCC_LONG; }

// bf16 operand copies of the decoder hidden rows (fp32 h2 [R][160], the tanh output of Decoder.linear1 nb:709):
// h2b [R_pad][160] plain, h2s [R_pad][160] scaled by log2(e), h2tb [160][R_pad] transposed; R_pad % 64 == 0.
extern "C" int svae_proj_prep_rows_range(const float* h2, int R, int R_pad, int row0, int row1, void* h2b, void* h2s,
                                         void* h2tb, void* stream);
extern "C" int svae_proj_prep_rows(const float* h2, int R, int R_pad, void* h2b, void* h2s, void* h2tb,
                                   void* stream) {
    return svae_proj_prep_rows_range(h2, R, R_pad, 0, R_pad, h2b, h2s, h2tb, stream);
}

// The same for rows [row0, row1) only (multiples of 32; all pointers are those of the whole arrays).
extern "C" int svae_proj_prep_rows_range_split(const float* h2, int R, int R_pad, int row0, int row1, void* h2b, void* h2s,
                                               void* h2tb, long long part_stride, void* stream);
extern "C" int svae_proj_prep_rows_range(const float* h2, int R, int R_pad, int row0, int row1, void* h2b, void* h2s,
                                         void* h2tb, void* stream) {
    return svae_proj_prep_rows_range_split(h2, R, R_pad, row0, row1, h2b, h2s, h2tb, 0, stream);
}
// part_stride > 0: also the mid and lo parts of the three copies for the split instantiation, part_stride and
// 2 part_stride ELEMENTS behind each copy (part_stride >= R_pad * 160; 0: bf16 copies only).
extern "C" int svae_proj_prep_rows_range_split(const float* h2, int R, int R_pad, int row0, int row1, void* h2b, void* h2s,
                                               void* h2tb, long long part_stride, void* stream) {
    SV_CHECK_ARG(h2 && h2b && h2s && (h2tb || part_stride == 0) && R > 0 && R_pad >= R && (R_pad % BY) == 0,
                 "svae_proj_prep_rows: bad arguments");
    SV_CHECK_ARG(part_stride == 0 || (part_stride >= (long long)R_pad * KP && (part_stride % 8) == 0),
                 "svae_proj_prep_rows: bad part stride %lld", part_stride);
    SV_CHECK_ARG(row0 >= 0 && row0 < row1 && row1 <= R_pad && (row0 % 32) == 0 && (row1 % 32) == 0,
                 "svae_proj_prep_rows_range: bad range [%d, %d) of %d", row0, row1, R_pad);
    hipLaunchKernelGGL(proj_prep_kernel<1>, dim3((row1 - row0) / 32), dim3(256), 0, (hipStream_t)stream, h2,
                       (const float*)nullptr, (unsigned short*)h2b, (unsigned short*)h2s, (unsigned short*)h2tb, R, R_pad,
                       R_pad, row0, part_stride);
    SV_LAUNCH_CHECK();
    return 0;
}

// bf16 operand copies of Decoder.linear2 (nb:704): w3 [I][160] fp32, b3 [I]  ->  w3b, w3s [I_pad][160],
// w3tb [160][I_pad]; I_pad % 64 == 0.  _split: part_stride as above (>= I_pad * 160).
extern "C" int svae_proj_prep_items_split(const float* w3, const float* b3, int I, int I_pad, void* w3b, void* w3s,
                                          void* w3tb, long long part_stride, void* stream) {
    SV_CHECK_ARG(w3 && b3 && w3b && w3s && (w3tb || part_stride == 0) && I > 0 && I_pad >= I && (I_pad % BY) == 0,
                 "svae_proj_prep_items: bad arguments");
    SV_CHECK_ARG(part_stride == 0 || (part_stride >= (long long)I_pad * KP && (part_stride % 8) == 0),
                 "svae_proj_prep_items: bad part stride %lld", part_stride);
    hipLaunchKernelGGL(proj_prep_kernel<0>, dim3(I_pad / 32), dim3(256), 0, (hipStream_t)stream, w3, b3,
                       (unsigned short*)w3b, (unsigned short*)w3s, (unsigned short*)w3tb, I, I_pad, I_pad, 0, part_stride);
    SV_LAUNCH_CHECK();
    return 0;
}
extern "C" int svae_proj_prep_items(const float* w3, const float* b3, int I, int I_pad, void* w3b, void* w3s,
                                    void* w3tb, void* stream) {
    return svae_proj_prep_items_split(w3, b3, I, I_pad, w3b, w3s, w3tb, 0, stream);
}

// a schedule (svae_proj_schedule) must cover every x tile of nx rows exactly ns_j times: cheap consistency check
// of the three numbers the caller hands back (a stale schedule would leave slabs unwritten)
static bool sched_ok(int nx, int nwg, int ns_q, int ns_rem) {
    const int xt = (nx + BX - 1) / BX;
    if (ns_q == 0)          // flat schedule (svae_proj_schedule_flat): ns_rem = number of segments
        return nwg >= 1 && ns_rem >= nwg && ns_rem >= xt && (long long)ns_rem <= (long long)nwg + xt;
    return ns_q >= 1 && ns_q <= NS_MAX && ns_rem >= 0 && ns_rem < xt && ns_q + (ns_rem > 0) <= NS_MAX &&
           (long long)ns_q * xt + ns_rem == nwg;
}

// Parts of a flat schedule table (ns_q == 0, ns_rem = segments): the workgroups' segment offsets and the slab counts.
static inline const int* flat_wg_off(const int* sched, int ns_q, int nseg) { return ns_q == 0 ? sched + 4 * (long long)nseg : nullptr; }
static inline const int* flat_ns_tab(const int* sched, int nwg, int ns_q, int nseg) {
    return ns_q == 0 ? sched + 4 * (long long)nseg + 4 * ((nwg + 1 + 3) / 4) : nullptr;
}

// Reference interface replaced: Decoder.linear2 + F.log_softmax + likelihood of VAELoss.forward
// (nb:710, nb:802-808), loss only (scoring).  part = float2 [8][R] scratch.  Outputs nll_row[R], lse[R].
extern "C" int svae_proj_nll_fwd_split(const void* h2s, const void* w3b, const float* h2, const float* w3,
                                       const float* b3, const int* tgt, int R, int I, int K, float* part, float* nll_row,
                                       float* lse, const int* sched, int nwg, int ns_q, int ns_rem,
                                       long long row_part_stride, long long item_part_stride, void* stream);
extern "C" int svae_proj_nll_fwd(const void* h2s, const void* w3b, const float* h2, const float* w3, const float* b3,
                                 const int* tgt, int R, int I, int K, float* part, float* nll_row, float* lse,
                                 const int* sched, int nwg, int ns_q, int ns_rem, void* stream) {
    return svae_proj_nll_fwd_split(h2s, w3b, h2, w3, b3, tgt, R, I, K, part, nll_row, lse, sched, nwg, ns_q, ns_rem,
                                   0, 0, stream);
}
// row_part_stride / item_part_stride (both > 0 or both 0): the part strides the row and item copies were prepared
// with (svae_proj_prep_*_split) -> the split instantiation (fp32-class logits); 0: bf16.
extern "C" int svae_proj_nll_fwd_split(const void* h2s, const void* w3b, const float* h2, const float* w3,
                                       const float* b3, const int* tgt, int R, int I, int K, float* part, float* nll_row,
                                       float* lse, const int* sched, int nwg, int ns_q, int ns_rem,
                                       long long row_part_stride, long long item_part_stride, void* stream) {
    SV_CHECK_ARG(h2s && w3b && h2 && w3 && b3 && tgt && part && nll_row && lse && sched, "svae_proj_nll_fwd: null pointer");
    SV_CHECK_ARG(R > 0 && I > 0 && K >= 1, "svae_proj_nll_fwd: bad shape");
    SV_CHECK_ARG((row_part_stride > 0) == (item_part_stride > 0) && row_part_stride >= 0,
                 "svae_proj_nll_fwd: pass both part strides or none");
    SV_CHECK_ARG(sched_ok(R, nwg, ns_q, ns_rem), "svae_proj_nll_fwd: schedule does not belong to R = %d", R);
    SV_CHECK_ARG(ns_q > 0, "svae_proj_nll_fwd: the loss-only sweep takes the balanced schedule (svae_proj_schedule)");
    ProjArgs a{(const unsigned short*)h2s, (const unsigned short*)w3b, nullptr, part, nullptr, R, I, 0, (const int4*)sched,
               nullptr, row_part_stride, item_part_stride};
    if (int e = row_part_stride > 0 ? launch_proj<MODE_FWD, true>(a, nwg, (hipStream_t)stream)
                                    : launch_proj<MODE_FWD>(a, nwg, (hipStream_t)stream)) return e;
    hipLaunchKernelGGL(proj_fwd_finish_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)part, h2, w3, b3, tgt, R, K, nll_row, lse, ns_q, ns_rem);
    SV_LAUNCH_CHECK();
    return 0;
}

// Reference interface replaced: the same forward AND the hidden-side half of its autograd (nb:1205) in one
// sweep over the items.  Outputs nll_row[R], lse[R], dd1 [R][160] = dL/d(pre-tanh decoder hidden) (tanh of
// Decoder.linear1, nb:709), and the row constants in columns 153/154 of h2b that svae_proj_nll_bwd_dw needs.
// slabs: scratch of 16 * R * 160 floats; mpart: 16 * R floats.  parts: 1 = sweep kernel, 2 = finish (3 = both).
// _split: the part strides of the row copies (h2s, h2b) and of the item copies (w3b, w3tb); both 0 = bf16.
extern "C" int svae_proj_nll_fwd_dh_split(const void* h2s, void* h2b, const void* w3b, const void* w3tb, int I_pad,
                                          const float* h2, const float* w3, const float* b3, const int* tgt,
                                          const float* w_nll, int R, int I, int K, float* slabs, float* mpart,
                                          float* nll_row, float* lse, float* dd1, const int* sched, int nwg, int ns_q,
                                          int ns_rem, int parts, long long row_part_stride, long long item_part_stride,
                                          void* stream) {
    // (w3tb is read by the split instantiation only: the bf16 sweep transposes on the LDS read; it may be NULL there)
    SV_CHECK_ARG(h2s && h2b && w3b && (w3tb || row_part_stride == 0) && h2 && w3 && b3 && tgt && w_nll && slabs && mpart &&
                 nll_row && lse && dd1, "svae_proj_nll_fwd_dh: null pointer");
    SV_CHECK_ARG(R > 0 && I > 0 && K >= 1 && I_pad >= I && (I_pad % BY) == 0 && parts > 0 && parts < 4,
                 "svae_proj_nll_fwd_dh: bad shape");
    const bool split = row_part_stride > 0;
    SV_CHECK_ARG(split == (item_part_stride > 0) && row_part_stride >= 0, "svae_proj_nll_fwd_dh: pass both part strides or none");
    SV_CHECK_ARG(sched && sched_ok(R, nwg, ns_q, ns_rem), "svae_proj_nll_fwd_dh: schedule does not belong to R = %d", R);
    SV_CHECK_ARG(!split || ns_q > 0, "svae_proj_nll_fwd_dh: the split instantiation takes the balanced schedule");
    hipStream_t st = (hipStream_t)stream;
    if (parts & 1) {
        ProjArgs a{(const unsigned short*)h2s, (const unsigned short*)w3b, (const unsigned short*)w3tb, slabs, mpart,
                   R, I, I_pad, (const int4*)sched, flat_wg_off(sched, ns_q, ns_rem), row_part_stride, item_part_stride};
        if (int e = split ? launch_proj<MODE_FDH, true>(a, nwg, st) : launch_proj<MODE_FDH>(a, nwg, st)) return e;
    }
    if (parts & 2) {
        hipLaunchKernelGGL(proj_fdh_finish_kernel, dim3((R + 3) / 4), dim3(256), 0, st, slabs, mpart, h2, w3, b3, tgt,
                           w_nll, R, K, dd1, nll_row, lse, (unsigned short*)h2b, ns_q, ns_rem,
                           split ? (unsigned short*)h2b + row_part_stride : (unsigned short*)nullptr,
                           flat_ns_tab(sched, nwg, ns_q, ns_rem));
        SV_LAUNCH_CHECK();
    }
    return 0;
}
extern "C" int svae_proj_nll_fwd_dh(const void* h2s, void* h2b, const void* w3b, const void* w3tb, int I_pad,
                                    const float* h2, const float* w3, const float* b3, const int* tgt,
                                    const float* w_nll, int R, int I, int K, float* slabs, float* mpart,
                                    float* nll_row, float* lse, float* dd1, const int* sched, int nwg, int ns_q,
                                    int ns_rem, int parts, void* stream) {
    return svae_proj_nll_fwd_dh_split(h2s, h2b, w3b, w3tb, I_pad, h2, w3, b3, tgt, w_nll, R, I, K, slabs, mpart, nll_row,
                                      lse, dd1, sched, nwg, ns_q, ns_rem, parts, 0, 0, stream);
}

// Reference interface replaced: the weight-side half of the autograd of the same (nb:1205): gw3 [I_pad][160],
// gb3 [I_pad].  Needs svae_proj_nll_fwd_dh of the same batch before it (row constants in h2b).
// slabs: scratch of 16 * I * 160 floats.  occ_off [I+1] / occ_row: target occurrences by item; occ_items [I]: the
// items ordered by occurrence count (descending), the first nocc_long of them having more than svae_proj_occ_long().
// I_rows >= I: rows of gw3 / gb3 (rows >= I are written as zero).  parts: 1 = sweep kernel, 2 = finish (3 = both).
// _split: the part strides of the item copy (w3s) and of the row copies (h2b, h2tb); both 0 = bf16.
extern "C" int svae_proj_nll_bwd_dw_split(const void* w3s, const void* h2b, const void* h2tb, int R, int R_pad,
                                          int I, int I_rows, const float* h2, const float* w_nll, const int* occ_off,
                                          const int* occ_row, const int* occ_items, int nocc_long, float* slabs,
                                          float* gw3, float* gb3, const int* sched, int nwg, int ns_q, int ns_rem,
                                          int parts, long long row_part_stride, long long item_part_stride,
                                          void* stream) {
    // (h2tb: as w3tb above)
    SV_CHECK_ARG(w3s && h2b && (h2tb || row_part_stride == 0) && h2 && w_nll && occ_off && occ_row && occ_items && slabs &&
                 gw3 && gb3, "svae_proj_nll_bwd_dw: null pointer");
    SV_CHECK_ARG(nocc_long >= 0 && nocc_long <= I, "svae_proj_nll_bwd_dw: bad nocc_long");
    SV_CHECK_ARG(R > 0 && I > 0 && I_rows >= I && R_pad >= R && (R_pad % BY) == 0 && parts > 0 && parts < 4,
                 "svae_proj_nll_bwd_dw: bad shape");
    const bool split = row_part_stride > 0;
    SV_CHECK_ARG(split == (item_part_stride > 0) && row_part_stride >= 0, "svae_proj_nll_bwd_dw: pass both part strides or none");
    SV_CHECK_ARG(sched && sched_ok(I, nwg, ns_q, ns_rem), "svae_proj_nll_bwd_dw: schedule does not belong to I = %d", I);
    SV_CHECK_ARG(!split || ns_q > 0, "svae_proj_nll_bwd_dw: the split instantiation takes the balanced schedule");
    hipStream_t st = (hipStream_t)stream;
    if (parts & 1) {
        ProjArgs a{(const unsigned short*)w3s, (const unsigned short*)h2b, (const unsigned short*)h2tb, slabs, nullptr,
                   I, R, R_pad, (const int4*)sched, flat_wg_off(sched, ns_q, ns_rem), item_part_stride, row_part_stride};
        if (int e = split ? launch_proj<MODE_DW, true>(a, nwg, st) : launch_proj<MODE_DW>(a, nwg, st)) return e;
    }
    if (parts & 2) {
        if (nocc_long > 0)
            hipLaunchKernelGGL(proj_dw_finish_long_kernel, dim3(nocc_long), dim3(64 * DWL_WAVES), 0, st, slabs, h2, w_nll, occ_off,
                               occ_row, occ_items, gw3, gb3, I, ns_q, ns_rem, flat_ns_tab(sched, nwg, ns_q, ns_rem));
        hipLaunchKernelGGL(proj_dw_finish_kernel, dim3((I_rows - nocc_long + 3) / 4), dim3(256), 0, st, slabs, h2, w_nll,
                           occ_off, occ_row, occ_items, gw3, gb3, nocc_long, I, I_rows, ns_q, ns_rem,
                           flat_ns_tab(sched, nwg, ns_q, ns_rem));
        SV_LAUNCH_CHECK();
    }
    return 0;
}
extern "C" int svae_proj_nll_bwd_dw(const void* w3s, const void* h2b, const void* h2tb, int R, int R_pad,
                                    int I, int I_rows, const float* h2, const float* w_nll, const int* occ_off,
                                    const int* occ_row, const int* occ_items, int nocc_long, float* slabs, float* gw3,
                                    float* gb3, const int* sched, int nwg, int ns_q, int ns_rem, int parts,
                                    void* stream) {
    return svae_proj_nll_bwd_dw_split(w3s, h2b, h2tb, R, R_pad, I, I_rows, h2, w_nll, occ_off, occ_row, occ_items,
                                      nocc_long, slabs, gw3, gb3, sched, nwg, ns_q, ns_rem, parts, 0, 0, stream);
}
