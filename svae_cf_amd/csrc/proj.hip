// Fused |items|-wide output projection + log-softmax + next-k NLL for gfx950, forward and backward,
// WITHOUT materialising the [rows x items] logits (reference: Decoder.linear2 nb:710, F.log_softmax
// nb:802, likelihood nb:805-808, and their autograd).
//
// One kernel template, three modes.  An "X" operand is resident (64 rows of it per wave as MFMA B
// fragments in registers), a "Y" operand streams through LDS in tiles of 64 rows:
//
//   GEMM1   C1[y][x] = sum_k Y[y][k] X[x][k]           v_mfma_f32_32x32x16_bf16, K = 160 (10 steps)
//           accumulator initialised with a per-y constant ky[y]  (bias, or ln c_r - lse_r)
//   FWD     X = decoder hidden rows h2[r], Y = W3[i]:  per-lane online (max, sum-exp) over y -- the
//           logit's data row sits on the lane (swapped orientation), so no cross-lane reduction.
//   DH      X = h2[r], Y = W3[i]:  P = exp2(C1*log2e + kx[x]) = c_r softmax;  GEMM2  Z[x][n] += P^T Y
//           -> dL/dh2 (dense part)
//   DW      X = W3[i], Y = h2[r]:  P as above with the roles of kx / ky swapped;  GEMM2 Z[x][n] += P^T Y
//           -> dL/dW3 (dense part); column 150 of h2's bf16 copy is 1.0, so Z[:,150] = dL/db3.
//   GEMM2 contracts over y, the REGISTER index of C1: the accumulator tile is converted to bf16 in
//   place and used directly as the A operand (no LDS round trip); its B operand comes from a
//   transposed bf16 copy of Y staged next to the tile.
//
// The sparse "-y" part of dlogits = w (n softmax - y) is applied by the slab-reduce epilogues
// (targets are <= k per row; see proj_dh_finish / proj_dw_finish).
//
// Grid: 1-D, id -> (x tile = id / 8, y split = id % 8): blocks that share an XCD (id % 8) stream
// the same eighth of Y, which then stays in that XCD's L2.  Each split writes its own fp32 slab
// (deterministic; no atomics).
#include "common.h"

namespace {

constexpr int KP = SV_DP;          // 160
constexpr int KS1 = KP / 16;       // 10 k-steps of GEMM1
constexpr int NT2 = KP / 32;       // 5 n-tiles of GEMM2
constexpr int BX = 256;            // x rows per workgroup (4 waves x 64)
constexpr int BY = 64;             // y rows per LDS tile
constexpr int NSPLIT = 8;
constexpr int YLD = KP + 8;        // bf16 elements per row of the Y image (336 B)
constexpr int TLD = BY + 8;        // bf16 elements per row of the Y^T image (144 B: 16-byte aligned rows,
                                   // 36-dword stride = conflict-free 16-byte reads over 16 rows)
constexpr int STAGE_ELEMS = BY * YLD + KP * TLD;            // bf16 elements per stage (Y + Y^T)
constexpr int STAGE_BYTES = STAGE_ELEMS * 2 + BY * 4;       // + ky floats
constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG_BIG = -1.0e30f;

enum { MODE_FWD = 0, MODE_DH = 1, MODE_DW = 2 };

struct ProjArgs {
    const unsigned short* X;    // [nx][160] bf16
    const unsigned short* Y;    // [ny_pad][160] bf16 (rows >= ny are zero)
    const unsigned short* YT;   // [160][ldt] bf16 (transposed copy of Y; columns >= ny zero)
    const float* kx;            // [nx]      per-x exponent constant (log2 units); unused in FWD
    const float* ky;            // [ny_pad]  per-y accumulator initial value (natural units; pads = -1e30)
    float* out;                 // FWD: float2 partial (m, l) [NSPLIT][nx];  DH/DW: slabs [NSPLIT][nx][160]
    int nx, ny, ldt;
    int tiles_per_split;
};

// two fp32 -> packed bf16 pair (RNE): one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    const f32x2 v = {a, b};
    const bf16v2 h = __builtin_convertvector(v, bf16v2);
    return *reinterpret_cast<const unsigned*>(&h);
}

// NB = 32-row x sub-tiles per wave.  NB = 2: 4 waves (one per SIMD, 512 registers each).  NB = 1: 8 waves (two per
// SIMD): the exp / bf16-pack VALU work of one wave issues while the other wave's MFMAs execute, which a single
// wave per SIMD only gets as far as the compiler interleaves the two statically.
template <int MODE, int NB>
__global__ __launch_bounds__(512 / NB, 1) void proj_kernel(ProjArgs a) {
    constexpr int NT = 512 / NB;                 // threads
    constexpr int NCH = (1280 + NT - 1) / NT;    // 16-byte staging chunks per thread and image
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int xt = blockIdx.x / NSPLIT, split = blockIdx.x % NSPLIT;
    const int x0 = xt * BX + wave * (32 * NB);
    const int ntiles = (a.ny + BY - 1) / BY;
    const int t_begin = split * a.tiles_per_split;
    int t_end = t_begin + a.tiles_per_split;
    if (t_end > ntiles) t_end = ntiles;

    // resident X fragments: lane (col = li -> x, half lh) element j = X[x][16 s + 8 lh + j]
    bf16x8 xf[NB][KS1];
    float kxv[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int x = x0 + 32 * b + li;
        const bool ok = x < a.nx;
        kxv[b] = (MODE != MODE_FWD && ok) ? a.kx[x] : 0.f;
#pragma unroll
        for (int s = 0; s < KS1; ++s) {
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (ok) v = *reinterpret_cast<const bf16x8*>(a.X + (size_t)x * KP + 16 * s + 8 * lh);
            xf[b][s] = v;
        }
    }

    f32x16 Z[NB][NT2];
    if (MODE != MODE_FWD) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int n = 0; n < NT2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) Z[b][n][r] = 0.f;
    }
    float run_m[NB], run_l[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) { run_m[b] = NEG_BIG; run_l[b] = 0.f; }

    // ---- staging: global -> registers (issued early) -> LDS (written after the compute of the previous tile)
    // Y tile: 64 rows x 20 chunks of 16 B = 1280 chunks, 5 per thread.  Y^T tile: 160 rows x 8 chunks = 1280, 5 per thread.
    // (scalar staging registers: arrays indexed inside guarded loops end up in scratch)
    uint4 sy0, sy1, sy2, sy3, sy4, st0, st1, st2, st3, st4;
    float skv = 0.f;
#define SV_STAGE_LOAD1(i, SY, ST)                                                                         \
    if ((i) < NCH) {                                                                                      \
        const int c = tid + NT * (i);                                                                     \
        if (NT * ((i) + 1) <= 1280 || c < 1280) {                                                         \
            SY = *reinterpret_cast<const uint4*>(a.Y + (size_t)(y0_ + c / 20) * KP + 8 * (c % 20));       \
            if (MODE != MODE_FWD)                                                                         \
                ST = *reinterpret_cast<const uint4*>(a.YT + (size_t)(c >> 3) * a.ldt + y0_ + 8 * (c & 7)); \
        }                                                                                                 \
    }
#define SV_STAGE_LOAD(t_)                                \
    {                                                    \
        const int y0_ = (t_) * BY;                       \
        SV_STAGE_LOAD1(0, sy0, st0)                      \
        SV_STAGE_LOAD1(1, sy1, st1)                      \
        SV_STAGE_LOAD1(2, sy2, st2)                      \
        SV_STAGE_LOAD1(3, sy3, st3)                      \
        SV_STAGE_LOAD1(4, sy4, st4)                      \
        if (tid < BY) skv = a.ky[y0_ + tid];             \
    }
#define SV_STAGE_STORE1(i, SY, ST)                                                                       \
    if ((i) < NCH) {                                                                                     \
        const int c = tid + NT * (i);                                                                    \
        if (NT * ((i) + 1) <= 1280 || c < 1280) {                                                        \
            *reinterpret_cast<uint4*>(sY_ + (c / 20) * YLD + 8 * (c % 20)) = SY;                         \
            if (MODE != MODE_FWD) {                                                                      \
                /* y columns are stored in GEMM2's k-order: y = 16 s + 8 a + 4 h + c  ->  16 s + 8 h + 4 a + c */ \
                uint2* d = reinterpret_cast<uint2*>(sT_ + (c >> 3) * TLD + 16 * ((c & 7) >> 1) + 4 * (c & 1)); \
                d[0] = make_uint2(ST.x, ST.y);                                                           \
                d[2] = make_uint2(ST.z, ST.w);                                                           \
            }                                                                                            \
        }                                                                                                \
    }
#define SV_STAGE_STORE(buf_)                                                                         \
    {                                                                                                \
        unsigned short* sY_ = reinterpret_cast<unsigned short*>(smem + (buf_) * STAGE_BYTES);        \
        unsigned short* sT_ = sY_ + BY * YLD;                                                        \
        float* sK_ = reinterpret_cast<float*>(smem + (buf_) * STAGE_BYTES + STAGE_ELEMS * 2);        \
        SV_STAGE_STORE1(0, sy0, st0)                                                                 \
        SV_STAGE_STORE1(1, sy1, st1)                                                                 \
        SV_STAGE_STORE1(2, sy2, st2)                                                                 \
        SV_STAGE_STORE1(3, sy3, st3)                                                                 \
        SV_STAGE_STORE1(4, sy4, st4)                                                                 \
        if (tid < BY) sK_[tid] = skv;                                                                \
    }

    if (t_begin < t_end) {
        SV_STAGE_LOAD(t_begin)
        SV_STAGE_STORE(0)
    }
    __syncthreads();

    for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) & 1;
        const bool more = (t + 1) < t_end;
        if (more) SV_STAGE_LOAD(t + 1)
        const unsigned short* sY = reinterpret_cast<const unsigned short*>(smem + buf * STAGE_BYTES);
        const unsigned short* sT = sY + BY * YLD;
        const float* sK = reinterpret_cast<const float*>(smem + buf * STAGE_BYTES + STAGE_ELEMS * 2);

        // accumulator rows of a 32x32 tile: y = (r&3) + 8*(r>>2) + 4*lh  -> groups of 4 consecutive y.
        // Program order per tile (2 sub-tiles of 32 y): G1(s0) G1(s1) | P(s0) G2(s0) | P(s1) G2(s1) (FWD: softmax
        // instead of P/G2): the VALU work of one sub-tile is independent of the MFMAs next to it, so the
        // scheduler can run it under them (MFMA and VALU are separate pipes).
        f32x16 c1[2][NB];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 kv = *reinterpret_cast<const float4*>(sK + sub * 32 + 8 * g + 4 * lh);
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    c1[sub][b][4 * g + 0] = kv.x; c1[sub][b][4 * g + 1] = kv.y;
                    c1[sub][b][4 * g + 2] = kv.z; c1[sub][b][4 * g + 3] = kv.w;
                }
            }
#pragma unroll
            for (int s = 0; s < KS1; ++s) {
                const bf16x8 ya = *reinterpret_cast<const bf16x8*>(sY + (sub * 32 + li) * YLD + 16 * s + 8 * lh);
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    c1[sub][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ya, xf[b][s], c1[sub][b], 0, 0, 0);
            }
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            if (MODE == MODE_FWD) {
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    float tm = c1[sub][b][0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) tm = fmaxf(tm, c1[sub][b][r]);
                    const float mn = fmaxf(run_m[b], tm);
                    const float mnl = mn * LOG2E;
                    float acc = run_l[b] * __builtin_amdgcn_exp2f((run_m[b] - mn) * LOG2E);
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc += __builtin_amdgcn_exp2f(fmaf(c1[sub][b][r], LOG2E, -mnl));
                    run_l[b] = acc;
                    run_m[b] = mn;
                }
            } else {
                bf16x8 pa[NB][2];
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    float p[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) p[r] = __builtin_amdgcn_exp2f(fmaf(c1[sub][b][r], LOG2E, kxv[b]));
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        uint4 u;
                        u.x = pack_bf16(p[8 * s2 + 0], p[8 * s2 + 1]);
                        u.y = pack_bf16(p[8 * s2 + 2], p[8 * s2 + 3]);
                        u.z = pack_bf16(p[8 * s2 + 4], p[8 * s2 + 5]);
                        u.w = pack_bf16(p[8 * s2 + 6], p[8 * s2 + 7]);
                        pa[b][s2] = *reinterpret_cast<bf16x8*>(&u);
                    }
                }
                // GEMM2: Z[x][n] += sum_y P[y][x] Y[y][n].  k-order of a converted accumulator: element j of
                // half lh is y = 16 s2 + 8 (j>>2) + 4 lh + (j&3); the Y^T image stores its columns in that order, so a
                // fragment is one 16-byte read.
#pragma unroll
                for (int n = 0; n < NT2; ++n) {
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const bf16x8 yb = *reinterpret_cast<const bf16x8*>(sT + (32 * n + li) * TLD + sub * 32 + 16 * s2 + 8 * lh);
#pragma unroll
                        for (int b = 0; b < NB; ++b)
                            Z[b][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[b][s2], yb, Z[b][n], 0, 0, 0);
                    }
                }
            }
        }
        if (more) SV_STAGE_STORE(buf ^ 1)
        __syncthreads();
    }

    if (MODE == MODE_FWD) {
        // combine the two lane halves (same x, different y) and write the split's partial (m, l)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const float om = __shfl_xor(run_m[b], 32, 64);
            const float ol = __shfl_xor(run_l[b], 32, 64);
            const float mn = fmaxf(run_m[b], om);
            const float l = run_l[b] * __builtin_amdgcn_exp2f((run_m[b] - mn) * LOG2E) +
                            ol * __builtin_amdgcn_exp2f((om - mn) * LOG2E);
            const int x = x0 + 32 * b + li;
            if (lh == 0 && x < a.nx)
                reinterpret_cast<float2*>(a.out)[(size_t)split * a.nx + x] = make_float2(mn, l);
        }
    } else {
        // Z tile layout: col = li -> n, row = (r&3) + 8*(r>>2) + 4*lh -> x
        float* slab = a.out + (size_t)split * a.nx * KP;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int n = 0; n < NT2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int x = x0 + 32 * b + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (x < a.nx) slab[(size_t)x * KP + 32 * n + li] = Z[b][n][r];
                }
    }
}

// ------------------------------------------------------------------------------------------------
// operand preparation
// ------------------------------------------------------------------------------------------------
// fp32 [n][160] -> bf16 row-major copy [n_pad][160] and transposed copy [160][ldt]; optional "ones"
// column (col 150 := 1.0) used to get the bias gradient out of GEMM2; rows >= n are zero.
__global__ void proj_prep_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst,
                                 unsigned short* __restrict__ dstT, int n, int n_pad, int ldt, int ones_col) {
    __shared__ unsigned short tile[32][KP + 2];
    const int r0 = blockIdx.x * 32;
    for (int i = threadIdx.x; i < 32 * KP; i += blockDim.x) {
        const int r = i / KP, c = i % KP;
        float v = 0.f;
        if (r0 + r < n) {
            v = src[(size_t)(r0 + r) * KP + c];
            if (c == ones_col) v = 1.0f;
        }
        const unsigned short h = sv_f2bf(v);
        tile[r][c] = h;
        if (r0 + r < n_pad) dst[(size_t)(r0 + r) * KP + c] = h;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 32 * KP; i += blockDim.x) {
        const int c = i / 32, r = i % 32;
        if (r0 + r < ldt) dstT[(size_t)c * ldt + r0 + r] = tile[r][c];
    }
}

// ky for FWD / DH: bias (pads -> -1e30).  kx for DW: bias * log2e.
__global__ void proj_bias_consts_kernel(const float* __restrict__ b3, float* __restrict__ ky_items,
                                        float* __restrict__ kx_items, int I, int I_pad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= I_pad) return;
    const float b = i < I ? b3[i] : NEG_BIG;
    ky_items[i] = b;
    if (i < I) kx_items[i] = b * LOG2E;
}

// Forward finish, one wave per row: lse from the NSPLIT partials, target logits by fp32 dot products,
// nll_row, and the exponent constants of the backward kernels.
__global__ void proj_fwd_finish_kernel(const float2* __restrict__ part, const float* __restrict__ h2,
                                       const float* __restrict__ w3, const float* __restrict__ b3,
                                       const int* __restrict__ tgt, const float* __restrict__ w_nll, int R, int K,
                                       float* __restrict__ nll_row, float* __restrict__ lse_out,
                                       float* __restrict__ kx_rows, float* __restrict__ ky_rows, int R_pad) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= R_pad) return;
    if (row >= R) {
        if (lane == 0) ky_rows[row] = NEG_BIG;
        return;
    }
    float m = NEG_BIG, l = 0.f;
    if (lane < NSPLIT) {
        const float2 p = part[(size_t)lane * R + row];
        m = p.x; l = p.y;
    }
    const float M = sv_wave_max(m);
    l = sv_wave_sum(l * expf(m - M));
    const float lse = M + logf(l);
    float nll = 0.f;
    int nr = 0;
    const float* hr = h2 + (size_t)row * KP;
    for (int k = 0; k < K; ++k) {
        const int it = tgt[(size_t)row * K + k];     // wave-uniform
        if (it < 0) continue;
        const float* wr = w3 + (size_t)it * KP;
        float d = 0.f;
        for (int c = lane; c < SV_D; c += 64) d += hr[c] * wr[c];
        d = sv_wave_sum(d) + b3[it];
        nll += lse - d;
        ++nr;
    }
    if (lane == 0) {
        nll_row[row] = nll;
        lse_out[row] = lse;
        const float c = (float)nr * w_nll[row];
        kx_rows[row] = log2f(c) - lse * LOG2E;
        ky_rows[row] = logf(c) - lse;
    }
}

// dL/dh2 finish: sum the NSPLIT slabs, subtract the target part w_r sum_k W3[tgt], apply tanh'.
// One wave per row, float4 lanes 0..39.
__global__ void proj_dh_finish_kernel(const float* __restrict__ slabs, const float* __restrict__ w3,
                                      const int* __restrict__ tgt, const float* __restrict__ w_nll,
                                      const float* __restrict__ d1, float* __restrict__ dd1, int R, int K) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= R || lane >= KP / 4) return;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < NSPLIT; ++z) {
        const float4 v = reinterpret_cast<const float4*>(slabs + ((size_t)z * R + row) * KP)[lane];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const float w = w_nll[row];
    for (int k = 0; k < K; ++k) {
        const int it = tgt[(size_t)row * K + k];
        if (it < 0) continue;
        const float4 v = reinterpret_cast<const float4*>(w3 + (size_t)it * KP)[lane];
        s.x -= w * v.x; s.y -= w * v.y; s.z -= w * v.z; s.w -= w * v.w;
    }
    const float4 y = reinterpret_cast<const float4*>(d1 + (size_t)row * KP)[lane];
    float4 o;
    const int c0 = 4 * lane;
    o.x = c0 + 0 < SV_D ? s.x * (1.0f - y.x * y.x) : 0.f;
    o.y = c0 + 1 < SV_D ? s.y * (1.0f - y.y * y.y) : 0.f;
    o.z = c0 + 2 < SV_D ? s.z * (1.0f - y.z * y.z) : 0.f;
    o.w = c0 + 3 < SV_D ? s.w * (1.0f - y.w * y.w) : 0.f;
    reinterpret_cast<float4*>(dd1 + (size_t)row * KP)[lane] = o;
}

// dL/dW3, dL/db3 finish: one workgroup per item.  Wave 0 sums the slabs; the item's target occurrences
// (pairs sorted by item on the host: occ_row lists the data rows) are split over the 4 waves, 4 in flight
// per wave, and their partial sums are added in a fixed order (bitwise reproducible).  Column 150 of the
// slab sum is the bias gradient (ones column), its target part is -sum w_r.
__global__ __launch_bounds__(256) void proj_dw_finish_kernel(const float* __restrict__ slabs,
                                                             const float* __restrict__ h2,
                                                             const float* __restrict__ w_nll,
                                                             const int* __restrict__ occ_off,
                                                             const int* __restrict__ occ_row, float* __restrict__ gw3,
                                                             float* __restrict__ gb3, int I, int I_pad) {
    __shared__ float4 red[4][KP / 4];
    __shared__ float wred[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int it = blockIdx.x;
    const int ln = lane < KP / 4 ? lane : 0;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    float wsum = 0.f;
    if (it < I) {
        if (wave == 0) {
            for (int z = 0; z < NSPLIT; ++z) {
                const float4 v = reinterpret_cast<const float4*>(slabs + ((size_t)z * I + it) * KP)[ln];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        }
        const int b = occ_off[it], e = occ_off[it + 1];
        for (int o = b + wave; o < e; o += 16) {
            int r[4];
            float w[4];
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = (o + 4 * j < e) ? occ_row[o + 4 * j] : -1;
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
    if (lane < KP / 4) red[wave][lane] = s;
    if (lane == 0) wred[wave] = wsum;
    __syncthreads();
    if (wave != 0 || lane >= KP / 4) return;
    float4 t = red[0][lane];
#pragma unroll
    for (int q = 1; q < 4; ++q) {
        const float4 u = red[q][lane];
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    if (it < I) {
        if (lane == SV_D / 4) {          // lane 37 holds columns 148..151: column 150 = bias gradient
            gb3[it] = t.z - (wred[0] + wred[1] + wred[2] + wred[3]);
            t.z = 0.f;
        }
        const int c0 = 4 * lane;
        if (c0 + 0 >= SV_D) t.x = 0.f;
        if (c0 + 1 >= SV_D) t.y = 0.f;
        if (c0 + 2 >= SV_D) t.z = 0.f;
        if (c0 + 3 >= SV_D) t.w = 0.f;
    }
    reinterpret_cast<float4*>(gw3 + (size_t)it * KP)[lane] = t;
}

constexpr int PROJ_NB = 1;   // x sub-tiles per wave of the backward kernels (see proj_kernel)

template <int MODE>
int launch_proj(const ProjArgs& a, hipStream_t st) {
    const int lds = 2 * STAGE_BYTES;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)proj_kernel<MODE, PROJ_NB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return svae_fail((int)e, "svae_proj: LDS attribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    const int xt = (a.nx + BX - 1) / BX;
    hipLaunchKernelGGL((proj_kernel<MODE, PROJ_NB>), dim3(xt * NSPLIT), dim3(512 / PROJ_NB), lds, st, a);
    SV_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" int svae_proj_nsplit(void) { return NSPLIT; }
extern "C" int svae_proj_tile(void) { return BY; }

// bf16 operand copies.  dst [n_pad][160], dstT [160][ldt]; n_pad and ldt are multiples of 64, >= n.
extern "C" int svae_proj_prep(const float* src, void* dst, void* dstT, int n, int n_pad, int ldt, int ones_col,
                              void* stream) {
    SV_CHECK_ARG(src && dst && dstT && n > 0 && n_pad >= n && ldt >= n_pad && (n_pad % BY) == 0 && (ldt % 8) == 0,
                 "svae_proj_prep: bad arguments");
    hipLaunchKernelGGL(proj_prep_kernel, dim3((ldt + 31) / 32), dim3(256), 0, (hipStream_t)stream, src,
                       (unsigned short*)dst, (unsigned short*)dstT, n, n_pad, ldt, ones_col);
    SV_LAUNCH_CHECK();
    return 0;
}

extern "C" int svae_proj_bias_consts(const float* b3, float* ky_items, float* kx_items, int I, int I_pad,
                                     void* stream) {
    SV_CHECK_ARG(b3 && ky_items && kx_items && I > 0 && I_pad >= I, "svae_proj_bias_consts: bad arguments");
    hipLaunchKernelGGL(proj_bias_consts_kernel, dim3((I_pad + 255) / 256), dim3(256), 0, (hipStream_t)stream, b3,
                       ky_items, kx_items, I, I_pad);
    SV_LAUNCH_CHECK();
    return 0;
}

// Reference interface replaced: Decoder.linear2 + F.log_softmax + likelihood of VAELoss.forward
// (nb:710, nb:802-808), fused.  h2b = bf16 copy of the decoder hidden rows [R_pad][160], w3b / w3tb =
// bf16 copies of W3 ([I_pad][160] and [160][ldt]); part = float2 [8][R] scratch.
// Outputs: nll_row[R], lse[R], kx_rows[R], ky_rows[R_pad] (exponent constants for the backward).
extern "C" int svae_proj_nll_fwd(const void* h2b, const void* w3b, const float* ky_items, const float* h2,
                                 const float* w3, const float* b3, const int* tgt, const float* w_nll, int R,
                                 int R_pad, int I, int K, float* part, float* nll_row, float* lse, float* kx_rows,
                                 float* ky_rows, void* stream) {
    SV_CHECK_ARG(h2b && w3b && ky_items && h2 && w3 && b3 && tgt && w_nll && part && nll_row && lse && kx_rows &&
                     ky_rows, "svae_proj_nll_fwd: null pointer");
    SV_CHECK_ARG(R > 0 && I > 0 && K >= 1 && R_pad >= R && (R_pad % BY) == 0, "svae_proj_nll_fwd: bad shape");
    ProjArgs a{(const unsigned short*)h2b, (const unsigned short*)w3b, nullptr, nullptr, ky_items, part, R, I, 0, 0};
    const int ntiles = (I + BY - 1) / BY;
    a.tiles_per_split = (ntiles + NSPLIT - 1) / NSPLIT;
    if (int e = launch_proj<MODE_FWD>(a, (hipStream_t)stream)) return e;
    hipLaunchKernelGGL(proj_fwd_finish_kernel, dim3((R_pad + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)part, h2, w3, b3, tgt, w_nll, R, K, nll_row, lse, kx_rows, ky_rows, R_pad);
    SV_LAUNCH_CHECK();
    return 0;
}

// Reference interface replaced: autograd of the same (nb:1205).  Dense parts by the fused kernels, the
// sparse target parts and tanh' (of Decoder.linear1's activation, nb:709) in the finish kernels.
//   dd1 [R][160]  = dL/d(pre-tanh decoder hidden)       gw3 [I_pad][160], gb3 [I_pad]
// slabs: scratch of 8 * max(R, I) * 160 floats.  occ_off [I+1] / occ_row: target occurrences by item.
extern "C" int svae_proj_nll_bwd(const void* h2b, const void* h2tb, int ldt_rows, const void* w3b, const void* w3tb,
                                 int ldt_items, const float* kx_rows, const float* ky_rows, const float* kx_items,
                                 const float* ky_items, const float* h2, const float* w3, const int* tgt,
                                 const float* w_nll, const int* occ_off, const int* occ_row, int R, int I, int I_pad,
                                 int K, float* slabs, float* dd1, float* gw3, float* gb3, int parts, void* stream) {
    SV_CHECK_ARG(h2b && h2tb && w3b && w3tb && kx_rows && ky_rows && kx_items && ky_items && h2 && w3 && tgt &&
                     w_nll && occ_off && occ_row && slabs && dd1 && gw3 && gb3, "svae_proj_nll_bwd: null pointer");
    SV_CHECK_ARG(R > 0 && I > 0 && I_pad >= I && K >= 1 && parts > 0 && parts < 16, "svae_proj_nll_bwd: bad shape");
    hipStream_t st = (hipStream_t)stream;
    // parts: 1 = weight-gradient kernel, 2 = its finish, 4 = hidden-gradient kernel, 8 = its finish (15 = all;
    // the pieces share `slabs`, so they must be enqueued in this order on one stream)
    if (parts & 3) {   // dL/dW3, dL/db3:  X = items, Y = rows
        ProjArgs a{(const unsigned short*)w3b, (const unsigned short*)h2b, (const unsigned short*)h2tb, kx_items,
                   ky_rows, slabs, I, R, ldt_rows, 0};
        const int ntiles = (R + BY - 1) / BY;
        a.tiles_per_split = (ntiles + NSPLIT - 1) / NSPLIT;
        if (parts & 1)
            if (int e = launch_proj<MODE_DW>(a, st)) return e;
        if (parts & 2) {
            hipLaunchKernelGGL(proj_dw_finish_kernel, dim3(I_pad), dim3(256), 0, st, slabs, h2, w_nll, occ_off,
                               occ_row, gw3, gb3, I, I_pad);
            SV_LAUNCH_CHECK();
        }
    }
    if (parts & 12) {   // dL/dh2:  X = rows, Y = items
        ProjArgs a{(const unsigned short*)h2b, (const unsigned short*)w3b, (const unsigned short*)w3tb, kx_rows,
                   ky_items, slabs, R, I, ldt_items, 0};
        const int ntiles = (I + BY - 1) / BY;
        a.tiles_per_split = (ntiles + NSPLIT - 1) / NSPLIT;
        if (parts & 4)
            if (int e = launch_proj<MODE_DH>(a, st)) return e;
        if (parts & 8) {
            hipLaunchKernelGGL(proj_dh_finish_kernel, dim3((R + 3) / 4), dim3(256), 0, st, slabs, w3, tgt, w_nll, h2,
                               dd1, R, K);
            SV_LAUNCH_CHECK();
        }
    }
    return 0;
}
