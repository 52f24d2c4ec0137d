// Row-panel GEMM for the two GRU-side contractions whose weight is small and whose rows are many (gfx950, bf16 MFMA,
// fp32 accumulate):
//     which 0   gi  [R][672] = emb [R][256] W_ih^T + b_ih     (input GEMM inside nn.GRU, call nb:759)
//     which 1   demb[R][256] = dgi [R][672] W_ih             (its data gradient at loss.backward(), nb:1205)
// Both are HBM-bound shapes (80 MB of rows against 7.5 GFLOP).  The generic LDS-tiled GEMM (gemm.hip) ran them at
// 1-1.5 TB/s: 128x128 output tiles re-read the row operand once per column block and keep one k-tile in flight.
// Here a workgroup owns 128 rows and a column chunk whose weight fragments ALL fit the LDS:
//   * a wave owns 32 rows; its row operand goes from HBM / L2 straight into MFMA B fragments (lane = row, 8
//     consecutive k per lane and k-step: two 16-byte loads);
//   * the product is computed transposed, C^T[n][r] = sum_k W[n][k] X[r][k]: the weight is the A operand, laid out
//     beforehand (rowgemm_prep, once per optimizer step) as bf16 fragments in exactly the order the kernel consumes
//     them, so a stage is one linear copy global -> registers -> LDS and every fragment read is a lane-linear,
//     conflict-free ds_read_b128 shared by the four waves;
//   * a lane ends up with 4 consecutive output columns of its own row per accumulator group: float4 stores.
// The chunk's fragments come in by one LDS-DMA burst; the row loads are plain loads issued ahead of their use (the
// compiler places their waits).
#include "common.h"

namespace {

// Waves per workgroup (32 rows each).  dL/demb runs 8 waves = 256 rows per workgroup: half as many workgroups (228 at
// 21.8k rows: one round of the GPU instead of 1.8), half the L2 -> LDS traffic of the weight fragments, two waves per
// SIMD -- its region in the step 105 -> 98 us, the step itself within the spread (0.8694 -> 0.8682 and 0.8635 -> 0.8626 ms,
// two A/Bs of three alternating pairs, end of round 2);
// bit-identical (a row's arithmetic does not depend on its workgroup).  The split instantiation stays at 4 waves (at
// 8 it needs more than the 256 registers two waves per SIMD leave it).  The gi GEMM likewise (126 + 144 workgroups for
// the two time ranges instead of 252 + 288): its region 33 -> 31 us, the step 0.8696 -> 0.8667 ms, three alternating pairs.
#ifndef SV_RG0_NW
#define SV_RG0_NW 8
#endif
#ifndef SV_RG1_NW
#define SV_RG1_NW 8
#endif

// CT = n-tiles (of 32 columns) per column chunk, chosen so that ALL fragments of a chunk (CT x K/16 KiB) fit the LDS;
// XG = k-steps per group of row loads, PDX = groups in flight ahead of the one being multiplied.
template <int WHICH> struct RgShape;
template <> struct RgShape<0> { static constexpr int K = SV_E, N = SV_G, CT = 8, XG = 4, PDX = 3, NW = SV_RG0_NW; };
template <> struct RgShape<1> { static constexpr int K = SV_G, N = SV_E, CT = 3, XG = 6, PDX = 2, NW = SV_RG1_NW; };
template <int WHICH, bool SPLIT> constexpr int rg_nw() { return SPLIT ? 4 : RgShape<WHICH>::NW; }

struct RgArgs {
    const float* x;               // [R][K]
    const unsigned short* frags;  // weight fragments, consumption order (rowgemm_prep)
    const float* bias;            // [N] or nullptr
    float* y;                     // [R][N]
    int R, nrb;                   // rows, row blocks of 32 NW
    int ldx;                      // row stride of x (floats, >= K)
    const unsigned short* frags_lo;   // SPLIT: the mid parts of the weight, same fragment order, followed (K*N elements
                                      // further) by the lo parts; read from L2, not staged
};

__device__ __forceinline__ unsigned rg_pk2(float a, float b) {
    const bf16v2 h = __builtin_convertvector((f32x2){a, b}, bf16v2);
    return __builtin_bit_cast(unsigned, h);
}

// element offset (bf16 units) of the fragment of n-tile `tile`, k-step `ks`: chunk-major, then k-step, then tile
template <int WHICH>
__host__ __device__ constexpr size_t rg_frag_off(int tile, int ks) {
    using S = RgShape<WHICH>;
    constexpr int KS = S::K / 16, NT = S::N / 32;
    const int c = tile / S::CT, t = tile % S::CT;
    const int ntc = (NT - c * S::CT) < S::CT ? (NT - c * S::CT) : S::CT;
    return ((size_t)c * S::CT * KS + (size_t)ks * ntc + t) * 512;
}

__device__ __forceinline__ void rg_glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// One workgroup = 128 rows x one column chunk.  All weight fragments of the chunk go to LDS in one LDS-DMA burst while
// the first row loads are in flight; one barrier; then MFMAs with the row operand streaming in behind them.
// (A first version walked ALL chunks per workgroup with a three-stage fragment ring: 171 workgroups of one wave per
//  SIMD, every block of 32 MFMAs waiting for a 1-2 us DMA issued two blocks earlier: 35 us however few rows.)
// Grid: id -> (row block, chunk) with the chunks of a row block 8 ids apart, i.e. on the same XCD (id % 8) and
// dispatched together: the row block is then read from HBM once and from that XCD's L2 by the other chunks.
// XB: x holds bf16 elements (ldx in elements): a k-step of a lane is one 16-byte load and needs no conversion.
// SPLIT (fp32 rows only): the parity instantiation.  Same tiling and fragment order; operands as bf16 hi + mid + lo
// parts (24 bits), a product as the six leading terms (rows split in registers, the weight's mid / lo fragments read
// straight from L2: the LDS is full of the hi fragments).
template <int WHICH, bool XB, bool SPLIT = false>
__global__ __launch_bounds__((64 * rg_nw<WHICH, SPLIT>())) void rowgemm_kernel(RgArgs a) {
    static_assert(!(XB && SPLIT), "the split instantiation reads fp32 rows");
    using S = RgShape<WHICH>;
    constexpr int K = S::K, N = S::N, CT = S::CT, XG = S::XG, PDX = S::PDX;
    constexpr int KS = K / 16, NT = N / 32, NCH = (NT + CT - 1) / CT, NG = KS / XG;
    static_assert(KS % XG == 0, "k-steps must fill whole groups");
    extern __shared__ __attribute__((aligned(16))) char wst[];           // [ntc * KS] fragments of 1 KiB

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    // id = 8 * (NCH * (rb / 8) + c) + rb % 8
    const int id = blockIdx.x, q = id >> 3;
    const int c = q % NCH, rb = 8 * (q / NCH) + (id & 7);
    if (rb >= a.nrb) return;                                             // whole workgroup
    const int ntc = (NT - c * CT) < CT ? (NT - c * CT) : CT;
    constexpr int NW = rg_nw<WHICH, SPLIT>();
    const int row = rb * (32 * NW) + wave * 32 + li;
    const bool rok = row < a.R;
    const float* xr = XB ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(a.x) +
                                                           (size_t)(rok ? row : a.R - 1) * a.ldx + 8 * lh)
                         : a.x + (size_t)(rok ? row : a.R - 1) * a.ldx + 8 * lh;

    // weight fragments of the chunk: ntc * KS pieces of 1 KiB, wave w issues pieces w, w + 4, ...
    {
        const char* src = reinterpret_cast<const char*>(a.frags) + (size_t)c * CT * KS * 1024 + lane * 16;
        const int npieces = ntc * KS;
        for (int p = wave; p < npieces; p += NW) rg_glds16(src + (size_t)p * 1024, wst + p * 1024);
    }
    float4 xraw[PDX + 1][XG][XB ? 1 : 2];
#define RG_LOADX(g_)                                                                                         \
    _Pragma("unroll") for (int s_ = 0; s_ < XG; ++s_) {                                                      \
        if (XB) {   /* 16 bf16 per k-step = 8 floats' worth of address */                                    \
            xraw[(g_) % (PDX + 1)][s_][0] = *reinterpret_cast<const float4*>(xr + 8 * ((g_) * XG + s_));     \
        } else {                                                                                             \
            xraw[(g_) % (PDX + 1)][s_][0] = *reinterpret_cast<const float4*>(xr + 16 * ((g_) * XG + s_));    \
            xraw[(g_) % (PDX + 1)][s_][1] = *reinterpret_cast<const float4*>(xr + 16 * ((g_) * XG + s_) + 4);\
        }                                                                                                    \
    }
#pragma unroll
    for (int g = 0; g < PDX && g < NG; ++g) RG_LOADX(g)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // The chunk's tile count is a compile-time value inside `run` (full chunks, or the last, narrower one).
    // Measured alone at 21.8k rows (tools/rowgemm_timing.py, launch gap included): gi 33 us, of which the 58 MB of
    // output stores are 12 us and the row loads 7 us; dL/demb 39-43 us.  In the step the generic GEMM took 45 / 137 us
    // on average for the same work and this kernel 43 / 103 us -- the second shares the GPU with the dW3 sweep.
    auto run = [&](auto ntc_c) {
        constexpr int NTC = decltype(ntc_c)::value;
        f32x16 acc[NTC];
#pragma unroll
        for (int t = 0; t < NTC; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        const uint4* st = reinterpret_cast<const uint4*>(wst) + lane;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + PDX < NG) RG_LOADX(g + PDX)
#pragma unroll
            for (int s = 0; s < XG; ++s) {
                bf16x8 xf, xm, xl;
                if (XB) {
                    xf = __builtin_bit_cast(bf16x8, xraw[g % (PDX + 1)][s][0]);
                } else if (SPLIT) {
                    const float4 lo = xraw[g % (PDX + 1)][s][0], hi = xraw[g % (PDX + 1)][s][XB ? 0 : 1];
                    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    unsigned hw[4], mw[4], lw[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        hw[i] = rg_pk2(v[2 * i], v[2 * i + 1]);
                        const float r0 = v[2 * i] - __uint_as_float(hw[i] << 16);
                        const float r1 = v[2 * i + 1] - __uint_as_float(hw[i] & 0xFFFF0000u);
                        mw[i] = rg_pk2(r0, r1);
                        lw[i] = rg_pk2(r0 - __uint_as_float(mw[i] << 16), r1 - __uint_as_float(mw[i] & 0xFFFF0000u));
                    }
                    xf = __builtin_bit_cast(bf16x8, (uint4){hw[0], hw[1], hw[2], hw[3]});
                    xm = __builtin_bit_cast(bf16x8, (uint4){mw[0], mw[1], mw[2], mw[3]});
                    xl = __builtin_bit_cast(bf16x8, (uint4){lw[0], lw[1], lw[2], lw[3]});
                } else {
                    const float4 lo = xraw[g % (PDX + 1)][s][0], hi = xraw[g % (PDX + 1)][s][1];
                    uint4 u;
                    u.x = rg_pk2(lo.x, lo.y); u.y = rg_pk2(lo.z, lo.w); u.z = rg_pk2(hi.x, hi.y); u.w = rg_pk2(hi.z, hi.w);
                    xf = __builtin_bit_cast(bf16x8, u);
                }
                const int ks = g * XG + s;
#pragma unroll
                for (int t = 0; t < NTC; ++t) {
                    const bf16x8 wf = __builtin_bit_cast(bf16x8, st[(ks * NTC + t) * 64]);
                    if (SPLIT) {
                        const unsigned short* fp_ = a.frags_lo + ((size_t)c * CT * KS + (size_t)ks * NTC + t) * 512 + lane * 8;
                        const bf16x8 wm = *reinterpret_cast<const bf16x8*>(fp_);
                        const bf16x8 wl = *reinterpret_cast<const bf16x8*>(fp_ + (size_t)K * N);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xf, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xl, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, xm, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, xf, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xm, acc[t], 0, 0, 0);
                    }
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xf, acc[t], 0, 0, 0);
                }
            }
        }
        // lane li = row, register i -> column 32 t + (i & 3) + 8 (i >> 2) + 4 lh of the chunk
        if (rok) {
            float* yr = a.y + (size_t)row * N + c * CT * 32 + 4 * lh;
#pragma unroll
            for (int t = 0; t < NTC; ++t) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 v = make_float4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]);
                    if (a.bias) {
                        const float4 bv = *reinterpret_cast<const float4*>(a.bias + c * CT * 32 + 32 * t + 8 * g + 4 * lh);
                        v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                    }
                    *reinterpret_cast<float4*>(yr + 32 * t + 8 * g) = v;
                }
            }
        }
    };
    constexpr int LAST = NT - (NCH - 1) * CT;
    if (ntc == CT) run(std::integral_constant<int, CT>{});
    else run(std::integral_constant<int, LAST>{});
#undef RG_LOADX
}

// fp32 weight -> bf16 fragments in consumption order.  Element (n, k) of the weight the kernel multiplies by is
// src[n * sn + k * sk]  (which 0: W_ih [672][256], sn = 256, sk = 1;  which 1: W_ih^T, sn = 1, sk = 256).
template <int WHICH>
__global__ void rowgemm_prep_kernel(const float* __restrict__ src, int sn, int sk, unsigned short* __restrict__ dst,
                                    unsigned short* __restrict__ dst_lo) {
    using S = RgShape<WHICH>;
    constexpr int KS = S::K / 16, NT = S::N / 32;
    const int f = blockIdx.x;                 // fragment id = tile * KS + ks
    const int tile = f / KS, ks = f % KS;
    const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
    unsigned short* d = dst + rg_frag_off<WHICH>(tile, ks) + lane * 8;
    const int n = 32 * tile + li, k0 = 16 * ks + 8 * lh;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = src[(size_t)n * sn + (size_t)(k0 + j) * sk];
        const unsigned short h = sv_f2bf(v);
        d[j] = h;
        if (dst_lo) {      // mid part, then (K*N elements further) the lo part
            const float r = v - sv_bf2f(h);
            const unsigned short m = sv_f2bf(r);
            (dst_lo + (d - dst))[j] = m;
            (dst_lo + (size_t)S::K * S::N + (d - dst))[j] = sv_f2bf(r - sv_bf2f(m));
        }
    }
    (void)NT;
}

// Head of a training step in ONE launch: the embedding gather of the first rows (one wave per row: 64 lanes x 16 B =
// one 1 KiB row) and the bf16 fragments of W_ih for the input-gate GEMM that follows it (rowgemm_prep_kernel<0>'s
// work, one wave per fragment).  They are independent of each other and both head the step's critical path: one
// launch gap instead of two.
constexpr int RG_PREP0_FRAGS = (SV_G / 32) * (SV_E / 16);
template <bool OUT16>
__global__ __launch_bounds__(256) void embed_wih_prep_kernel(const float* __restrict__ table, const int* __restrict__ x,
                                                             float* __restrict__ out, int R, int I,
                                                             const float* __restrict__ wih,
                                                             unsigned short* __restrict__ frags, int nprep) {
    const int PREP_BLOCKS = nprep;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((int)blockIdx.x < PREP_BLOCKS) {
        using S = RgShape<0>;
        constexpr int KS = S::K / 16;
        const int f = blockIdx.x * 4 + wave;              // fragment id = tile * KS + ks
        const int tile = f / KS, ks = f % KS;
        const int li = lane & 31, lh = lane >> 5;
        unsigned short* d = frags + rg_frag_off<0>(tile, ks) + lane * 8;
        const int n = 32 * tile + li, k0 = 16 * ks + 8 * lh;
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = sv_f2bf(wih[(size_t)n * SV_E + (k0 + j)]);
        return;
    }
    int row = (blockIdx.x - PREP_BLOCKS) * 4 + wave;
    const int stride = (gridDim.x - PREP_BLOCKS) * 4;
    for (; row < R; row += stride) {
        const int it = x[row];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (it >= 0 && it < I) v = reinterpret_cast<const float4*>(table + (size_t)it * SV_E)[lane];
        if (OUT16) {      // bf16 rows (512 B): what both consumers of the rows round them to anyway
            uint2 u;
            u.x = rg_pk2(v.x, v.y); u.y = rg_pk2(v.z, v.w);
            reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(out) + (size_t)row * SV_E)[lane] = u;
        } else {
            reinterpret_cast<float4*>(out + (size_t)row * SV_E)[lane] = v;
        }
    }
}

template <int WHICH, bool XB, bool SPLIT = false>
int rg_launch(RgArgs a, hipStream_t st) {
    using S = RgShape<WHICH>;
    constexpr int lds = S::CT * (S::K / 16) * 1024;
    constexpr int nch = (S::N / 32 + S::CT - 1) / S::CT;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)rowgemm_kernel<WHICH, XB, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return svae_fail((int)e, "svae_rowgemm: LDS attribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    constexpr int NW = rg_nw<WHICH, SPLIT>();
    a.nrb = (a.R + 32 * NW - 1) / (32 * NW);
    const int groups = (a.nrb + 7) / 8;                 // 8 row blocks (one per XCD) x nch chunks each
    hipLaunchKernelGGL((rowgemm_kernel<WHICH, XB, SPLIT>), dim3(groups * nch * 8), dim3(64 * NW), lds, st, a);
    SV_LAUNCH_CHECK();
    return 0;
}

}  // namespace

// svae_embed_fwd (rows [0, R) of the plan) and svae_rowgemm_prep_prec(wih, 0, frags, 1) in one launch (bf16 mode).
// Reference: nn.Embedding call nb:756; the fragments are this build's operand form of gru.weight_ih_l0 (nb:724).
// out_bf16 != 0: `out` receives bf16 rows ([R][256] elements of 2 bytes) for svae_rowgemm_gi_bf16x /
// svae_wgrad_cols_bf16dy_bf16x; wih == NULL: the gather alone (svae_embed_fwd_bf16).
extern "C" int svae_embed_fwd_wih_prep(const float* table, const int* x, void* out, int R, int I, const float* wih,
                                       void* frags, int out_bf16, void* stream) {
    SV_CHECK_ARG(table && x && out && R > 0 && I > 0 && ((wih == nullptr) == (frags == nullptr)),
                 "svae_embed_fwd_wih_prep: bad arguments");
    static_assert(RG_PREP0_FRAGS % 4 == 0, "four fragments per workgroup");
    int gather_blocks = (R + 3) / 4;
    if (gather_blocks > 8192) gather_blocks = 8192;     // grid-stride beyond that
    const int nprep = wih ? RG_PREP0_FRAGS / 4 : 0;
    if (out_bf16)
        hipLaunchKernelGGL(embed_wih_prep_kernel<true>, dim3(nprep + gather_blocks), dim3(256), 0, (hipStream_t)stream,
                           table, x, (float*)out, R, I, wih, (unsigned short*)frags, nprep);
    else
        hipLaunchKernelGGL(embed_wih_prep_kernel<false>, dim3(nprep + gather_blocks), dim3(256), 0, (hipStream_t)stream,
                           table, x, (float*)out, R, I, wih, (unsigned short*)frags, nprep);
    SV_LAUNCH_CHECK();
    return 0;
}
// The gather alone with bf16 rows out (the second time range's rows, on the side stream).
extern "C" int svae_embed_fwd_bf16(const float* table, const int* x, void* out16, int R, int I, void* stream) {
    return svae_embed_fwd_wih_prep(table, x, out16, R, I, nullptr, nullptr, 1, stream);
}

extern "C" int svae_rowgemm_frag_elems(int which) {
    return which == 0 ? SV_G * SV_E : which == 1 ? SV_E * SV_G : -1;
}

// wih: gru.weight_ih_l0 as stored, [672][256] fp32 (gate rows padded 200 -> 224 with zeros).
// prec 1: bf16 fragments (svae_rowgemm_frag_elems elements); prec 2: hi fragments followed by the mid and the lo
// fragments (v - hi and v - hi - mid, rounded to bf16) for the split instantiation: 3 * svae_rowgemm_frag_elems.
extern "C" int svae_rowgemm_prep_prec(const float* wih, int which, void* frags, int prec, void* stream) {
    SV_CHECK_ARG(wih && frags && (which == 0 || which == 1) && (prec == 1 || prec == 2), "svae_rowgemm_prep: bad arguments");
    unsigned short* lo = prec == 2 ? (unsigned short*)frags + SV_G * SV_E : nullptr;
    if (which == 0)
        hipLaunchKernelGGL(rowgemm_prep_kernel<0>, dim3((SV_G / 32) * (SV_E / 16)), dim3(64), 0, (hipStream_t)stream, wih,
                           SV_E, 1, (unsigned short*)frags, lo);
    else
        hipLaunchKernelGGL(rowgemm_prep_kernel<1>, dim3((SV_E / 32) * (SV_G / 16)), dim3(64), 0, (hipStream_t)stream, wih,
                           1, SV_E, (unsigned short*)frags, lo);
    SV_LAUNCH_CHECK();
    return 0;
}
extern "C" int svae_rowgemm_prep(const float* wih, int which, void* frags, void* stream) {
    return svae_rowgemm_prep_prec(wih, which, frags, 1, stream);
}

// which 0: y [R][672] = x [R][256] W_ih^T + bias (bias may be NULL);  which 1: y [R][256] = x [R][672] W_ih.
// frags from svae_rowgemm_prep(which) of the current W_ih.  x and y 16-byte aligned, dense rows.
extern "C" int svae_rowgemm_ld(const float* x, int ldx, const void* frags, const float* bias, float* y, int R, int which,
                               void* stream);
extern "C" int svae_rowgemm(const float* x, const void* frags, const float* bias, float* y, int R, int which,
                            void* stream) {
    return svae_rowgemm_ld(x, which == 0 ? SV_E : SV_G, frags, bias, y, R, which, stream);
}

// The same with rows of x ldx floats apart (dL/demb reads (dr, dz, dn) out of the merged [R][896] gradient array).
extern "C" int svae_rowgemm_ld_prec(const float* x, int ldx, const void* frags, const float* bias, float* y, int R,
                                    int which, int prec, void* stream);
extern "C" int svae_rowgemm_ld(const float* x, int ldx, const void* frags, const float* bias, float* y, int R, int which,
                               void* stream) {
    return svae_rowgemm_ld_prec(x, ldx, frags, bias, y, R, which, 1, stream);
}
// prec 2: the split instantiation; frags from svae_rowgemm_prep_prec(..., 2).
extern "C" int svae_rowgemm_ld_prec(const float* x, int ldx, const void* frags, const float* bias, float* y, int R,
                                    int which, int prec, void* stream) {
    SV_CHECK_ARG(prec == 1 || prec == 2, "svae_rowgemm: prec %d (1 = bf16, 2 = split bf16)", prec);
    SV_CHECK_ARG(x && frags && y && R > 0 && (which == 0 || which == 1), "svae_rowgemm: bad arguments");
    SV_CHECK_ARG(ldx >= (which == 0 ? SV_E : SV_G) && (ldx % 4) == 0, "svae_rowgemm: bad row stride %d", ldx);
    SV_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && (!bias || ((uintptr_t)bias % 16) == 0),
                 "svae_rowgemm: operands must be 16-byte aligned");
    RgArgs a{x, (const unsigned short*)frags, bias, y, R, 0, ldx, (const unsigned short*)frags + SV_G * SV_E};
    if (prec == 2)
        return which == 0 ? rg_launch<0, false, true>(a, (hipStream_t)stream) : rg_launch<1, false, true>(a, (hipStream_t)stream);
    return which == 0 ? rg_launch<0, false>(a, (hipStream_t)stream) : rg_launch<1, false>(a, (hipStream_t)stream);
}

// which 0 with x as bf16 elements (the embedding rows as svae_embed_fwd_bf16 / svae_embed_fwd_wih_prep wrote them):
// gi = emb W_ih^T + b_ih, bit-identical to the fp32-row form (which rounds the rows to bf16 itself, RNE as the gather).
extern "C" int svae_rowgemm_gi_bf16x(const void* x, int ldx, const void* frags, const float* bias, float* y, int R,
                                     void* stream) {
    SV_CHECK_ARG(x && frags && y && R > 0 && ldx >= SV_E && (ldx % 8) == 0, "svae_rowgemm_gi_bf16x: bad arguments");
    SV_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && (!bias || ((uintptr_t)bias % 16) == 0),
                 "svae_rowgemm_gi_bf16x: operands must be 16-byte aligned");
    RgArgs a{(const float*)x, (const unsigned short*)frags, bias, y, R, 0, ldx, nullptr};
    return rg_launch<0, true>(a, (hipStream_t)stream);
}

// which 1 with x as bf16 elements (the GRU's merged gradient array): ldx in elements, a multiple of 8.
extern "C" int svae_rowgemm_bf16x(const void* x, int ldx, const void* frags, float* y, int R, void* stream) {
    SV_CHECK_ARG(x && frags && y && R > 0 && ldx >= SV_G && (ldx % 8) == 0, "svae_rowgemm_bf16x: bad arguments");
    SV_CHECK_ARG(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0, "svae_rowgemm_bf16x: operands must be 16-byte aligned");
    RgArgs a{(const float*)x, (const unsigned short*)frags, nullptr, y, R, 0, ldx, nullptr};
    return rg_launch<1, true>(a, (hipStream_t)stream);
}
