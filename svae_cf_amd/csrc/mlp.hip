// The row-wise MLP chain between the GRU and the |items|-wide projection, fused, for gfx950 (bf16 MFMA,
// fp32 accumulate), forward and backward:
//
//   forward   enc = tanh(W1 h + b1)          Encoder.forward nb:693-696          [R,224] -> [R,160]
//             tmp = Wl enc + bl = (mu | ls)  Model.sample_latent nb:738-741      -> [R,128]
//             z   = mu + exp(ls) eps,  KL    nb:743-750, VAELoss nb:797          -> [R,64], kl_row[R]
//             d1  = tanh(W2 z + b2)          Decoder.linear1 + tanh nb:708-709   -> [R,160]
//   backward  dz = W2^T dd1;  dtmp = latent'(dz);  denc = (Wl^T dtmp) (1 - enc^2);  dhs = W1^T denc
//             (autograd of the same at loss.backward() nb:1205; weight gradients are separate GEMMs)
//
// One wave owns 32 rows for the whole chain and never leaves its registers between layers: every
// layer is computed TRANSPOSED, C^T[n][r] = sum_k W[n][k] act[k][r], with the weights as the MFMA A
// operand and the activations as the B operand, so a data row sits on a LANE and the features on the
// REGISTER index.  The fp32 accumulator tile of one layer, converted to bf16 in place, is exactly a B
// operand of the next layer; the k-order it implies (kmap_acc below) is baked into the weight
// fragments, which svae_mlp_prep lays out fragment-major (one fully coalesced 1 KiB load per fragment).
// As separate launches of the generic GEMM these were 8 latency-bound kernels of 35-40 us each.
#include "common.h"

#ifdef SV_MLP_TIMING
// diagnostic build only (-DSV_MLP_TIMING, read by tools/probe/mlp_time.py): phase stamps of workgroup 0, wave 0
__device__ unsigned long long g_mlp_dbg[8];
#define SV_MTS(i_) { __builtin_amdgcn_sched_barrier(0); ts_[i_] = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
#else
#define SV_MTS(i_)
#endif
namespace {

constexpr int KS_H = SV_HP / 16;    // 14 k-steps over the GRU state
constexpr int KS_D = SV_DP / 16;    // 10 k-steps over a hidden layer
constexpr int KS_L2 = SV_L2 / 16;   // 8  k-steps over (mu | ls)
constexpr int KS_L = SV_L / 16;     // 4  k-steps over z
constexpr int NT_D = SV_DP / 32;    // 5 n-tiles of a hidden layer
constexpr int NT_L2 = SV_L2 / 32;   // 4
constexpr int NT_L = SV_L / 32;     // 2
constexpr int NT_H = SV_HP / 32;    // 7
// fragment images, in units of fragments (64 lanes x 8 bf16 = 1 KiB each)
constexpr int OFF_F1 = 0;                          // [NT_D][KS_H]   W1[n][k]      k natural
constexpr int OFF_F2 = OFF_F1 + NT_D * KS_H;       // [NT_L2][KS_D]  Wl[n][k]      k in accumulator order
constexpr int OFF_F3 = OFF_F2 + NT_L2 * KS_D;      // [NT_D][KS_L]   W2[n][k]      k in accumulator order
constexpr int OFF_B1 = OFF_F3 + NT_D * KS_L;       // [NT_L][KS_D]   W2[k][n]      k natural
constexpr int OFF_B2 = OFF_B1 + NT_L * KS_D;       // [NT_D][KS_L2]  Wl[k][n]      k in accumulator order
constexpr int OFF_B3 = OFF_B2 + NT_D * KS_L2;      // [NT_H][KS_D]   W1[k][n]      k in accumulator order
constexpr int NFRAG = OFF_B3 + NT_H * KS_D;        // 260

// k index of element j of a lane's 8-element fragment (lane half lh) at k-step s
__device__ __forceinline__ int kmap_nat(int s, int lh, int j) { return 16 * s + 8 * lh + j; }
// ... when the B operand is a converted 32x32 accumulator: register rr of half lh holds row (rr&3) + 8 (rr>>2) + 4 lh
__device__ __forceinline__ int kmap_acc(int s, int lh, int j) { return 16 * s + (j & 3) + 8 * (j >> 2) + 4 * lh; }

__global__ void mlp_prep_kernel(const float* __restrict__ w1, const float* __restrict__ wl,
                                const float* __restrict__ w2, unsigned short* __restrict__ frags, int parts) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = gid >> 6, lane = gid & 63;
    if (f >= NFRAG) return;
    const int li = lane & 31, lh = lane >> 5;
    unsigned short v[8], vm[8], vl[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float x;
        if (f < OFF_F2) {
            const int nt = (f - OFF_F1) / KS_H, s = (f - OFF_F1) % KS_H;
            x = w1[(size_t)(32 * nt + li) * SV_HP + kmap_nat(s, lh, j)];
        } else if (f < OFF_F3) {
            const int nt = (f - OFF_F2) / KS_D, s = (f - OFF_F2) % KS_D;
            x = wl[(size_t)(32 * nt + li) * SV_DP + kmap_acc(s, lh, j)];
        } else if (f < OFF_B1) {
            const int nt = (f - OFF_F3) / KS_L, s = (f - OFF_F3) % KS_L;
            x = w2[(size_t)(32 * nt + li) * SV_L + kmap_acc(s, lh, j)];
        } else if (f < OFF_B2) {
            const int nt = (f - OFF_B1) / KS_D, s = (f - OFF_B1) % KS_D;
            x = w2[(size_t)kmap_nat(s, lh, j) * SV_L + 32 * nt + li];
        } else if (f < OFF_B3) {
            const int nt = (f - OFF_B2) / KS_L2, s = (f - OFF_B2) % KS_L2;
            x = wl[(size_t)kmap_acc(s, lh, j) * SV_DP + 32 * nt + li];
        } else {
            const int nt = (f - OFF_B3) / KS_D, s = (f - OFF_B3) % KS_D;
            x = w1[(size_t)kmap_acc(s, lh, j) * SV_HP + 32 * nt + li];
        }
        v[j] = sv_f2bf(x);
        const float r = x - sv_bf2f(v[j]);
        vm[j] = sv_f2bf(r);
        vl[j] = sv_f2bf(r - sv_bf2f(vm[j]));
    }
    uint4 u;
    u.x = v[0] | ((unsigned)v[1] << 16); u.y = v[2] | ((unsigned)v[3] << 16);
    u.z = v[4] | ((unsigned)v[5] << 16); u.w = v[6] | ((unsigned)v[7] << 16);
    reinterpret_cast<uint4*>(frags)[(size_t)f * 64 + lane] = u;
    if (parts == 3) {    // the split instantiation's mid and lo parts: fragments NFRAG + f and 2 NFRAG + f
        u.x = vm[0] | ((unsigned)vm[1] << 16); u.y = vm[2] | ((unsigned)vm[3] << 16);
        u.z = vm[4] | ((unsigned)vm[5] << 16); u.w = vm[6] | ((unsigned)vm[7] << 16);
        reinterpret_cast<uint4*>(frags)[(size_t)(NFRAG + f) * 64 + lane] = u;
        u.x = vl[0] | ((unsigned)vl[1] << 16); u.y = vl[2] | ((unsigned)vl[3] << 16);
        u.z = vl[4] | ((unsigned)vl[5] << 16); u.w = vl[6] | ((unsigned)vl[7] << 16);
        reinterpret_cast<uint4*>(frags)[(size_t)(2 * NFRAG + f) * 64 + lane] = u;
    }
}

__device__ __forceinline__ unsigned pk2(float a, float b) {
    const bf16v2 h = __builtin_convertvector((f32x2){a, b}, bf16v2);
    return __builtin_bit_cast(unsigned, h);
}
// ---- split instantiation (SPLIT): operands as bf16 hi + mid + lo parts (mid = RNE(v - hi), lo = RNE(v - hi - mid):
// all 24 significant bits), a product as its six leading terms hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi ->
// fp32-class results from the same kernels (the parity gates run through it).  Activations are split in registers;
// the weights' mid / lo fragments are read straight from L2 (the LDS holds the hi fragments); tanh / exp use the
// accurate library forms.  A B operand is a `B3`: only .h is computed (and alive) in the bf16 instantiation.
struct B3 { bf16x8 h, m, l; };
// 8 floats -> parts
template <bool SPLIT>
__device__ __forceinline__ void parts8(const float (&v)[8], B3& o) {
    unsigned hw[4], mw[4], lw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        hw[i] = pk2(v[2 * i], v[2 * i + 1]);
        if (SPLIT) {
            const float r0 = v[2 * i] - __uint_as_float(hw[i] << 16), r1 = v[2 * i + 1] - __uint_as_float(hw[i] & 0xFFFF0000u);
            mw[i] = pk2(r0, r1);
            lw[i] = pk2(r0 - __uint_as_float(mw[i] << 16), r1 - __uint_as_float(mw[i] & 0xFFFF0000u));
        }
    }
    o.h = __builtin_bit_cast(bf16x8, (uint4){hw[0], hw[1], hw[2], hw[3]});
    if (SPLIT) {
        o.m = __builtin_bit_cast(bf16x8, (uint4){mw[0], mw[1], mw[2], mw[3]});
        o.l = __builtin_bit_cast(bf16x8, (uint4){lw[0], lw[1], lw[2], lw[3]});
    }
}
// an accumulator tile -> the two B fragments (k-steps 2T, 2T+1) of the next layer
template <bool SPLIT>
__device__ __forceinline__ void to_frags(const f32x16& c, B3& f0, B3& f1) {
    const float a[8] = {c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]};
    const float b[8] = {c[8], c[9], c[10], c[11], c[12], c[13], c[14], c[15]};
    parts8<SPLIT>(a, f0);
    parts8<SPLIT>(b, f1);
}
// fragment f of the kernel's LDS-resident image (lane-linear 16-byte reads: conflict-free)
__device__ __forceinline__ bf16x8 frag_ld(const char* lds, int f, int lane) {
    return *reinterpret_cast<const bf16x8*>(lds + f * 1024 + lane * 16);
}
// The kernel's weight fragments (nfrag x 1 KiB, contiguous in HBM from fragment `first`) go to LDS by LDS-DMA,
// one wave-instruction per fragment, spread over the block's 4 waves.  Straight from L2 every MFMA waited
// for its own fragment load (about 130 dependent L2 round trips per wave: 40 us for a 5 us kernel).
__device__ __forceinline__ void stage_frags(const unsigned short* frags, int first, int nfrag, char* lds, int wave, int lane) {
    const char* g = reinterpret_cast<const char*>(frags) + (size_t)first * 1024 + lane * 16;
    for (int f = wave; f < nfrag; f += 4)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + f * 1024),
                                         (__attribute__((address_space(3))) void*)(lds + f * 1024), 16, 0, 0);
}
constexpr int NFRAG_FWD = OFF_B1 - OFF_F1;     // 130
constexpr int NFRAG_BWD = NFRAG - OFF_B1;      // 130
constexpr int MLP_ROWS = 128;                  // rows per block (4 waves x 32)
// Behind the forward kernel's fragments: the three bias vectors (b1 [160], bl [128], b2 [160]) as floats.  Every
// accumulator tile starts from its bias; read from global memory in front of each tile, that load sits BEHIND the
// stores of the tile before it in the wave's one in-order memory counter, so every tile waited for an HBM write
// acknowledgement.  The kernels now issue every global load before their first store and never wait for a store
// (a lone workgroup: 20.3 -> 17.8 us forward, 170 workgroups 25.7 -> 22.8 us; tools/probe/mlp_time.py).
// What the rest is (SV_MLP_TIMING stamps, cycles of one wave): 12k until the fragments and rows are there (two
// memory round trips' worth, the 130 KiB LDS-DMA per workgroup), then 9k / 7k / 13k for the three layers, which is
// the wave's own vector work -- 352 transcendentals (tanh = exp + rcp, quarter rate) and ~2000 other vector
// instructions for 32 rows, one wave per SIMD.  Tried on top, no gain, not kept: the tiles stored as whole
// 128-byte row segments through an LDS transpose (a direct store touches 32 rows per instruction; 18.6 us), and the
// fragments of a k-loop read one tile ahead instead of ds_read -> wait -> MFMA per k-step (18.5 us).  Halving the
// rows per wave (16-row waves on 16x16x32 tiles, 64-row workgroups) is what would shorten it; the backward's 21.8k
// rows then need two rounds of workgroups.
constexpr int BIAS_FLOATS = SV_DP + SV_L2 + SV_DP;
constexpr int MLP_FWD_LDS = NFRAG_FWD * 1024 + BIAS_FLOATS * 4;
constexpr int BOFF_1 = 0, BOFF_L = SV_DP, BOFF_2 = SV_DP + SV_L2;
// natural-order B fragment from an fp32 row: 8 consecutive floats
template <bool SPLIT>
__device__ __forceinline__ B3 row_frag(const float* p) {
    const float4 lo = *reinterpret_cast<const float4*>(p);
    const float4 hi = *reinterpret_cast<const float4*>(p + 4);
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    B3 o;
    parts8<SPLIT>(v, o);
    return o;
}
// c += W (fragment fl of the LDS image = fragment f of the whole set) x B
template <bool SPLIT>
__device__ __forceinline__ void mm(f32x16& c, const char* wlds, int fl, const unsigned short* frags, int f, int lane,
                                   const B3& b) {
    const bf16x8 wh = frag_ld(wlds, fl, lane);
    if (SPLIT) {       // small terms first
        const bf16x8 wm = *reinterpret_cast<const bf16x8*>(frags + ((size_t)(NFRAG + f) * 64 + lane) * 8);
        const bf16x8 wl = *reinterpret_cast<const bf16x8*>(frags + ((size_t)(2 * NFRAG + f) * 64 + lane) * 8);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, b.h, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, b.l, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, b.m, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wm, b.h, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, b.m, c, 0, 0, 0);
    }
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, b.h, c, 0, 0, 0);
}
template <bool SPLIT> __device__ __forceinline__ float mlp_tanh(float x) { return SPLIT ? tanhf(x) : sv_tanh_fast(x); }
template <bool SPLIT> __device__ __forceinline__ float mlp_exp(float x) {
    return SPLIT ? expf(x) : __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
}
// accumulator tile initialised with a per-feature bias: register group g holds features 32 nt + 8 g + 4 lh + 0..3
__device__ __forceinline__ void bias_init(f32x16& c, const float* bias, int nt, int lh) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 b = *reinterpret_cast<const float4*>(bias + 32 * nt + 8 * g + 4 * lh);
        c[4 * g + 0] = b.x; c[4 * g + 1] = b.y; c[4 * g + 2] = b.z; c[4 * g + 3] = b.w;
    }
}
__device__ __forceinline__ void tile_store(float* dst_row, int nt, int lh, const f32x16& c, bool ok) {
    if (!ok) return;
#pragma unroll
    for (int g = 0; g < 4; ++g)
        *reinterpret_cast<float4*>(dst_row + 32 * nt + 8 * g + 4 * lh) =
            make_float4(c[4 * g + 0], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]);
}

struct MlpFwdArgs {
    const float* hs;        // [R][224]
    const float* eps;       // [R][64]
    const unsigned short* frags;
    const float* b1;        // [160]
    const float* bl;        // [128]
    const float* b2;        // [160]
    float* enc;             // [R][160]
    float* tmp;             // [R][128]
    float* z;               // [R][64]
    float* d1;              // [R][160]
    float* kl_row;          // [R]
    int R;
    // PREP only: the projection sweeps' bf16 operand copies of d1 (what svae_proj_prep_rows_range makes from d1)
    unsigned short* h2b;    // [R_pad][160]  plain copy + constant columns
    unsigned short* h2s;    // [R_pad][160]  copy scaled by log2(e) + constant columns
    unsigned short* h2tb;   // [160][R_pad]  transpose of h2b, rows in the sweeps' k order
    int row0, row1, ldt;    // rows [row0, row1) are written (row1 may reach into the pad rows R..R_pad)
};

// K-pad constant columns of the projection operands (the same numbers as csrc/proj.hip's file header)
constexpr int PC_ONE = 150, PC_BH = 151, PC_BL = 152, PC_RH = 153;
constexpr float P_LOG2E = 1.4426950408889634f, P_NEG_BIG = -1.0e30f;
__device__ __forceinline__ int prep_korder(int y) { return (y & ~12) | ((y & 4) << 1) | ((y & 8) >> 1); }
// (plain, scaled) bf16 of element (row, column c) of the decoder hidden, as proj_prep_kernel<1> writes them
__device__ __forceinline__ void prep_vals(int c, float v, bool live, unsigned short& hp, unsigned short& hs) {
    const unsigned short one = 0x3F80;
    hp = 0; hs = 0;
    if (c < SV_D) {
        if (live) { hp = sv_f2bf(v); hs = sv_f2bf(v * P_LOG2E); }
    } else {
        if (c == PC_ONE) hp = live ? one : 0;
        if (c == PC_BH || c == PC_BL) { hp = live ? one : 0; hs = live ? one : 0; }
        if (c == PC_RH) hp = live ? 0 : sv_f2bf(P_NEG_BIG);
    }
}

template <bool PREP, bool SPLIT = false>
__global__ __launch_bounds__(256) void mlp_fwd_kernel(MlpFwdArgs a) {
    static_assert(!(PREP && SPLIT), "the split instantiation leaves the operand copies to svae_proj_prep_rows_range_split");
    extern __shared__ __attribute__((aligned(16))) char wlds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int r = (PREP ? a.row0 : 0) + (blockIdx.x * 4 + wave) * 32 + li;
    const bool ok = r < a.R && (!PREP || r < a.row1);
    const size_t rc = ok ? r : a.R - 1;          // loads are clamped, stores masked
    // enc / tmp / z are what the backward reads; a forward without one (scoring) passes them NULL and keeps d1 only
    const bool ok_act = ok && a.enc != nullptr;

#ifdef SV_MLP_TIMING
    unsigned long long ts_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // ---- encoder: enc^T = tanh(W1 hs^T + b1)
    SV_MTS(0)
    // the weight fragments' LDS-DMA first (130 KiB per workgroup: the longest latency of the head), the row loads
    // behind it: one round trip instead of two
    stage_frags(a.frags, OFF_F1, NFRAG_FWD, wlds, wave, lane);
    float* bias_lds = reinterpret_cast<float*>(wlds + NFRAG_FWD * 1024);
    if (threadIdx.x < BIAS_FLOATS / 4) {
        const int i4 = threadIdx.x * 4;
        const float* src = i4 < BOFF_L ? a.b1 + i4 : (i4 < BOFF_2 ? a.bl + (i4 - BOFF_L) : a.b2 + (i4 - BOFF_2));
        *reinterpret_cast<float4*>(bias_lds + i4) = *reinterpret_cast<const float4*>(src);
    }
    SV_MTS(1)
    B3 hb[KS_H];
#pragma unroll
    for (int s = 0; s < KS_H; ++s) hb[s] = row_frag<SPLIT>(a.hs + rc * SV_HP + 16 * s + 8 * lh);
    // the noise of this lane's row, needed two layers down: fetched here, ahead of every store
    float4 epsv[NT_L][4];
#pragma unroll
    for (int q = 0; q < NT_L; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            epsv[q][g] = *reinterpret_cast<const float4*>(a.eps + rc * SV_L + 32 * q + 8 * g + 4 * lh);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SV_MTS(2)
    __syncthreads();
    SV_MTS(3)
    B3 pe[KS_D];
#pragma unroll
    for (int nt = 0; nt < NT_D; ++nt) {
        f32x16 c;
        bias_init(c, bias_lds + BOFF_1, nt, lh);
#pragma unroll
        for (int s = 0; s < KS_H; ++s)
            mm<SPLIT>(c, wlds, OFF_F1 + nt * KS_H + s, a.frags, OFF_F1 + nt * KS_H + s, lane, hb[s]);
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = mlp_tanh<SPLIT>(c[i]);
        tile_store(a.enc + rc * SV_DP, nt, lh, c, ok_act);
        to_frags<SPLIT>(c, pe[2 * nt], pe[2 * nt + 1]);
    }
    SV_MTS(4)
    // ---- (mu | ls)^T = Wl enc^T + bl
    f32x16 t[NT_L2];
#pragma unroll
    for (int nt = 0; nt < NT_L2; ++nt) {
        bias_init(t[nt], bias_lds + BOFF_L, nt, lh);
#pragma unroll
        for (int s = 0; s < KS_D; ++s)
            mm<SPLIT>(t[nt], wlds, OFF_F2 + nt * KS_D + s, a.frags, OFF_F2 + nt * KS_D + s, lane, pe[s]);
        tile_store(a.tmp + rc * SV_L2, nt, lh, t[nt], ok_act);
    }
    // ---- z = mu + exp(ls) eps, KL row term (log_sigma is a log-std for sampling and a log-variance in KL:
    //      reproduced as the reference has it, nb:743 / nb:797)
    B3 pz[KS_L];
    float kl = 0.f;
#pragma unroll
    for (int q = 0; q < NT_L; ++q) {
        f32x16 zt;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 e = epsv[q][g];
            const float ev[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float mu = t[q][4 * g + c], ls = t[q + 2][4 * g + c];
                const float sg = mlp_exp<SPLIT>(ls);
                zt[4 * g + c] = mu + sg * ev[c];
                kl += 0.5f * (-ls + sg + mu * mu - 1.0f);
            }
        }
        tile_store(a.z + rc * SV_L, q, lh, zt, ok_act);
        to_frags<SPLIT>(zt, pz[2 * q], pz[2 * q + 1]);
    }
    kl += __shfl_xor(kl, 32, 64);
    if (ok && lh == 0) a.kl_row[r] = kl;
    SV_MTS(5)
    // ---- decoder hidden: d1^T = tanh(W2 z^T + b2)
#pragma unroll
    for (int nt = 0; nt < NT_D; ++nt) {
        f32x16 c;
        bias_init(c, bias_lds + BOFF_2, nt, lh);
#pragma unroll
        for (int s = 0; s < KS_L; ++s)
            mm<SPLIT>(c, wlds, OFF_F3 + nt * KS_L + s, a.frags, OFF_F3 + nt * KS_L + s, lane, pz[s]);
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = mlp_tanh<SPLIT>(c[i]);
        tile_store(a.d1 + rc * SV_DP, nt, lh, c, ok);
        if (PREP && r < a.row1) {
            // the sweeps' operand copies straight from the accumulators (same values as a conversion of the d1 just
            // stored): 8-byte row-major stores; the transpose goes out as 2-byte stores, 32 consecutive rows of one
            // column per half-wave
            const int kr = prep_korder(r);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                unsigned short hp[4], hs[4];
                const int c0 = 32 * nt + 8 * g + 4 * lh;
#pragma unroll
                for (int i = 0; i < 4; ++i) prep_vals(c0 + i, c[4 * g + i], ok, hp[i], hs[i]);
                uint2 up, us;
                up.x = hp[0] | ((unsigned)hp[1] << 16); up.y = hp[2] | ((unsigned)hp[3] << 16);
                us.x = hs[0] | ((unsigned)hs[1] << 16); us.y = hs[2] | ((unsigned)hs[3] << 16);
                *reinterpret_cast<uint2*>(a.h2b + (size_t)r * SV_DP + c0) = up;
                *reinterpret_cast<uint2*>(a.h2s + (size_t)r * SV_DP + c0) = us;
                if (a.h2tb != nullptr) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) a.h2tb[(size_t)(c0 + i) * a.ldt + kr] = hp[i];
                }
            }
        }
    }
    SV_MTS(6)
#ifdef SV_MLP_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SV_MTS(7)
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int i = 0; i < 8; ++i) g_mlp_dbg[i] = ts_[i];
#endif
}

struct MlpBwdArgs {
    const float* dd1;       // [R][160]  dL/d(pre-tanh decoder hidden)
    const float* tmp;       // [R][128]
    const float* eps;       // [R][64]
    const float* enc;       // [R][160]
    const float* w_kl;      // [R]
    const unsigned short* frags;
    float* dtmp;            // [R][128]
    float* denc;            // [R][160]  dL/d(pre-tanh encoder hidden)
    float* dhs;             // [R][224]
    float anneal;
    int R;
};

template <bool SPLIT = false>
__global__ __launch_bounds__(256) void mlp_bwd_kernel(MlpBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char wlds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int r = (blockIdx.x * 4 + wave) * 32 + li;
    const bool ok = r < a.R;
    const size_t rc = ok ? r : a.R - 1;

    // ---- dz^T = W2^T dd1^T
    // (the fragments' LDS-DMA first, then every global load of the kernel, all before its first store: see MLP_FWD_LDS)
    stage_frags(a.frags, OFF_B1, NFRAG_BWD, wlds, wave, lane);
    B3 db[KS_D];
#pragma unroll
    for (int s = 0; s < KS_D; ++s) db[s] = row_frag<SPLIT>(a.dd1 + rc * SV_DP + 16 * s + 8 * lh);
    const float kw = a.anneal * a.w_kl[rc];
    float4 epsv[NT_L][4], muv[NT_L][4], lsv[NT_L][4], encv[NT_D][4];
#pragma unroll
    for (int q = 0; q < NT_L; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            epsv[q][g] = *reinterpret_cast<const float4*>(a.eps + rc * SV_L + 32 * q + 8 * g + 4 * lh);
            muv[q][g] = *reinterpret_cast<const float4*>(a.tmp + rc * SV_L2 + 32 * q + 8 * g + 4 * lh);
            lsv[q][g] = *reinterpret_cast<const float4*>(a.tmp + rc * SV_L2 + SV_L + 32 * q + 8 * g + 4 * lh);
        }
#pragma unroll
    for (int nt = 0; nt < NT_D; ++nt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            encv[nt][g] = *reinterpret_cast<const float4*>(a.enc + rc * SV_DP + 32 * nt + 8 * g + 4 * lh);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    f32x16 dz[NT_L];
#pragma unroll
    for (int nt = 0; nt < NT_L; ++nt) {
#pragma unroll
        for (int i = 0; i < 16; ++i) dz[nt][i] = 0.f;
#pragma unroll
        for (int s = 0; s < KS_D; ++s)
            mm<SPLIT>(dz[nt], wlds, nt * KS_D + s, a.frags, OFF_B1 + nt * KS_D + s, lane, db[s]);
    }
    // ---- latent backward: dmu = dz + a w mu ;  dls = dz exp(ls) eps + a w 0.5 (exp(ls) - 1)
    B3 pd[KS_L2];
#pragma unroll
    for (int q = 0; q < NT_L; ++q) {
        f32x16 dmu, dls;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 e = epsv[q][g], m = muv[q][g], l = lsv[q][g];
            const float ev[4] = {e.x, e.y, e.z, e.w}, mv[4] = {m.x, m.y, m.z, m.w}, lv[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float sg = mlp_exp<SPLIT>(lv[c]);
                const float gz = dz[q][4 * g + c];
                dmu[4 * g + c] = gz + kw * mv[c];
                dls[4 * g + c] = gz * sg * ev[c] + kw * 0.5f * (sg - 1.0f);
            }
        }
        tile_store(a.dtmp + rc * SV_L2, q, lh, dmu, ok);
        tile_store(a.dtmp + rc * SV_L2, q + 2, lh, dls, ok);
        to_frags<SPLIT>(dmu, pd[2 * q], pd[2 * q + 1]);
        to_frags<SPLIT>(dls, pd[2 * (q + 2)], pd[2 * (q + 2) + 1]);
    }
    // ---- denc^T = (Wl^T dtmp^T) (1 - enc^2)
    B3 pn[KS_D];
#pragma unroll
    for (int nt = 0; nt < NT_D; ++nt) {
        f32x16 c;
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = 0.f;
#pragma unroll
        for (int s = 0; s < KS_L2; ++s)
            mm<SPLIT>(c, wlds, OFF_B2 - OFF_B1 + nt * KS_L2 + s, a.frags, OFF_B2 + nt * KS_L2 + s, lane, pd[s]);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 y = encv[nt][g];
            c[4 * g + 0] *= 1.0f - y.x * y.x; c[4 * g + 1] *= 1.0f - y.y * y.y;
            c[4 * g + 2] *= 1.0f - y.z * y.z; c[4 * g + 3] *= 1.0f - y.w * y.w;
        }
        tile_store(a.denc + rc * SV_DP, nt, lh, c, ok);
        to_frags<SPLIT>(c, pn[2 * nt], pn[2 * nt + 1]);
    }
    // ---- dhs^T = W1^T denc^T
#pragma unroll
    for (int nt = 0; nt < NT_H; ++nt) {
        f32x16 c;
#pragma unroll
        for (int i = 0; i < 16; ++i) c[i] = 0.f;
#pragma unroll
        for (int s = 0; s < KS_D; ++s)
            mm<SPLIT>(c, wlds, OFF_B3 - OFF_B1 + nt * KS_D + s, a.frags, OFF_B3 + nt * KS_D + s, lane, pn[s]);
        tile_store(a.dhs + rc * SV_HP, nt, lh, c, ok);
    }
}

}  // namespace

extern "C" int svae_mlp_frag_elems(void) { return NFRAG * 512; }

// bf16 weight fragments of the three Linear layers (weights padded as in the flat parameter buffer:
// w1 [160][224] Encoder.linear1 nb:688, wl [128][160] Model.linear1 nb:729, w2 [160][64] Decoder.linear1 nb:701).
// frags: svae_mlp_frag_elems() bf16 elements.  Call once per optimizer step (the weights change).
extern "C" int svae_mlp_prep_prec(const float* w1, const float* wl, const float* w2, void* frags, int prec, void* stream);
extern "C" int svae_mlp_prep(const float* w1, const float* wl, const float* w2, void* frags, void* stream) {
    return svae_mlp_prep_prec(w1, wl, w2, frags, 1, stream);
}
// prec 2: frags holds 3 * svae_mlp_frag_elems() elements: the hi fragments, then the mid parts (v - hi), then the
// lo parts (v - hi - mid), each rounded to bf16.
extern "C" int svae_mlp_prep_prec(const float* w1, const float* wl, const float* w2, void* frags, int prec, void* stream) {
    SV_CHECK_ARG(w1 && wl && w2 && frags && (prec == 1 || prec == 2), "svae_mlp_prep: bad arguments");
    hipLaunchKernelGGL(mlp_prep_kernel, dim3((NFRAG * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w1, wl, w2,
                       (unsigned short*)frags, prec == 2 ? 3 : 1);
    SV_LAUNCH_CHECK();
    return 0;
}

// Reference interface replaced: Encoder.forward nb:693-696, Model.sample_latent nb:734-750, the KL row term of
// VAELoss nb:797 and Decoder.linear1 + tanh nb:708-709, for R rows at once.
extern "C" int svae_mlp_fwd_prec(const float* hs, const float* eps, const void* frags, const float* b1, const float* bl,
                                 const float* b2, float* enc, float* tmp, float* z, float* d1, float* kl_row, int R,
                                 int prec, void* stream);
extern "C" int svae_mlp_fwd(const float* hs, const float* eps, const void* frags, const float* b1, const float* bl,
                            const float* b2, float* enc, float* tmp, float* z, float* d1, float* kl_row, int R,
                            void* stream) {
    return svae_mlp_fwd_prec(hs, eps, frags, b1, bl, b2, enc, tmp, z, d1, kl_row, R, 1, stream);
}
// prec 2: the split instantiation (frags from svae_mlp_prep_prec(..., 2)).
extern "C" int svae_mlp_fwd_prec(const float* hs, const float* eps, const void* frags, const float* b1, const float* bl,
                                 const float* b2, float* enc, float* tmp, float* z, float* d1, float* kl_row, int R,
                                 int prec, void* stream) {
    SV_CHECK_ARG(hs && eps && frags && b1 && bl && b2 && enc && tmp && z && d1 && kl_row, "svae_mlp_fwd: null pointer");
    SV_CHECK_ARG(R > 0 && (prec == 1 || prec == 2), "svae_mlp_fwd: R = %d, prec = %d", R, prec);
    if (prec == 2) {
        MlpFwdArgs a2{hs, eps, (const unsigned short*)frags, b1, bl, b2, enc, tmp, z, d1, kl_row, R, nullptr, nullptr, nullptr, 0, 0, 0};
        static bool attr2_done = false;
        if (!attr2_done) {
            hipError_t e = hipFuncSetAttribute((const void*)mlp_fwd_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_FWD_LDS);
            if (e != hipSuccess) return svae_fail((int)e, "svae_mlp_fwd: LDS attribute: %s", hipGetErrorString(e));
            attr2_done = true;
        }
        hipLaunchKernelGGL((mlp_fwd_kernel<false, true>), dim3((R + MLP_ROWS - 1) / MLP_ROWS), dim3(256), MLP_FWD_LDS, (hipStream_t)stream, a2);
        SV_LAUNCH_CHECK();
        return 0;
    }
    MlpFwdArgs a{hs, eps, (const unsigned short*)frags, b1, bl, b2, enc, tmp, z, d1, kl_row, R, nullptr, nullptr, nullptr, 0, 0, 0};
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)mlp_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_FWD_LDS);
        if (e != hipSuccess) return svae_fail((int)e, "svae_mlp_fwd: LDS attribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    hipLaunchKernelGGL(mlp_fwd_kernel<false>, dim3((R + MLP_ROWS - 1) / MLP_ROWS), dim3(256), MLP_FWD_LDS, (hipStream_t)stream, a);
    SV_LAUNCH_CHECK();
    return 0;
}

// The same chain over rows [row0, row1) of the batch, which ALSO writes the projection sweeps' bf16 operand copies of
// its last output (h2b, h2s, h2tb: what svae_proj_prep_rows_range(d1, R, R_pad, row0, row1, ...) makes from d1,
// bit for bit) -- one kernel and one read of d1 less on the decoder side's critical chain.  All pointers are the
// arrays' bases (row 0); row0 % 32 == 0; row1 <= R_pad may reach into the pad rows, which get the non-live constants.
extern "C" int svae_mlp_fwd_prep(const float* hs, const float* eps, const void* frags, const float* b1, const float* bl,
                                 const float* b2, float* enc, float* tmp, float* z, float* d1, float* kl_row, int R,
                                 int R_pad, int row0, int row1, void* h2b, void* h2s, void* h2tb, void* stream) {
    // (h2tb may be NULL: the bf16 sweeps read the transpose out of the row-major image)
    SV_CHECK_ARG(hs && eps && frags && b1 && bl && b2 && d1 && kl_row && h2b && h2s, "svae_mlp_fwd_prep: null pointer");
    // (enc, tmp, z: all three or none -- none when no backward follows: they are then not written)
    SV_CHECK_ARG((enc && tmp && z) || (!enc && !tmp && !z), "svae_mlp_fwd_prep: enc / tmp / z must be given together");
    SV_CHECK_ARG(R > 0 && R_pad >= R && (R_pad % 64) == 0 && row0 >= 0 && row0 < row1 && row1 <= R_pad && (row0 % 32) == 0 &&
                 row0 < R, "svae_mlp_fwd_prep: bad range [%d, %d) of %d (%d)", row0, row1, R, R_pad);
    MlpFwdArgs a{hs, eps, (const unsigned short*)frags, b1, bl, b2, enc, tmp, z, d1, kl_row, R, (unsigned short*)h2b,
                 (unsigned short*)h2s, (unsigned short*)h2tb, row0, row1, R_pad};
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)mlp_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, MLP_FWD_LDS);
        if (e != hipSuccess) return svae_fail((int)e, "svae_mlp_fwd_prep: LDS attribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    hipLaunchKernelGGL(mlp_fwd_kernel<true>, dim3((row1 - row0 + MLP_ROWS - 1) / MLP_ROWS), dim3(256), MLP_FWD_LDS,
                       (hipStream_t)stream, a);
    SV_LAUNCH_CHECK();
    return 0;
}

// Reference interface replaced: autograd of the same chain at loss.backward() nb:1205 (data gradients; the
// weight gradients are svae_gemm calls on dd1/z, dtmp/enc, denc/hs).  dd1 = dL/d(pre-tanh decoder hidden).
extern "C" int svae_mlp_bwd_prec(const float* dd1, const float* tmp, const float* eps, const float* enc,
                                 const float* w_kl, const void* frags, float anneal, float* dtmp, float* denc, float* dhs,
                                 int R, int prec, void* stream);
extern "C" int svae_mlp_bwd(const float* dd1, const float* tmp, const float* eps, const float* enc,
                            const float* w_kl, const void* frags, float anneal, float* dtmp, float* denc, float* dhs,
                            int R, void* stream) {
    return svae_mlp_bwd_prec(dd1, tmp, eps, enc, w_kl, frags, anneal, dtmp, denc, dhs, R, 1, stream);
}
extern "C" int svae_mlp_bwd_prec(const float* dd1, const float* tmp, const float* eps, const float* enc,
                                 const float* w_kl, const void* frags, float anneal, float* dtmp, float* denc, float* dhs,
                                 int R, int prec, void* stream) {
    SV_CHECK_ARG(dd1 && tmp && eps && enc && w_kl && frags && dtmp && denc && dhs, "svae_mlp_bwd: null pointer");
    SV_CHECK_ARG(R > 0 && (prec == 1 || prec == 2), "svae_mlp_bwd: R = %d, prec = %d", R, prec);
    MlpBwdArgs a{dd1, tmp, eps, enc, w_kl, (const unsigned short*)frags, dtmp, denc, dhs, anneal, R};
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute((const void*)mlp_bwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, NFRAG_BWD * 1024);
        if (e == hipSuccess)
            e = hipFuncSetAttribute((const void*)mlp_bwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, NFRAG_BWD * 1024);
        if (e != hipSuccess) return svae_fail((int)e, "svae_mlp_bwd: LDS attribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    if (prec == 2)
        hipLaunchKernelGGL(mlp_bwd_kernel<true>, dim3((R + MLP_ROWS - 1) / MLP_ROWS), dim3(256), NFRAG_BWD * 1024, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(mlp_bwd_kernel<false>, dim3((R + MLP_ROWS - 1) / MLP_ROWS), dim3(256), NFRAG_BWD * 1024, (hipStream_t)stream, a);
    SV_LAUNCH_CHECK();
    return 0;
}

#ifdef SV_MLP_TIMING
extern "C" int svae_debug_mlp(unsigned long long* out8) {
    return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_mlp_dbg), 8 * sizeof(unsigned long long));
}
#endif
