// Generic LDS-tiled MFMA GEMM for gfx950:  C[M,N] = epi( sum_k A(m,k) * B(k,n) )
//
// Two arithmetic modes, same tiling:
//   PREC 0: exact fp32   v_mfma_f32_32x32x2_f32   (bit-for-bit an fmaf chain; parity gates)
//   PREC 1: bf16 inputs  v_mfma_f32_32x32x16_bf16 (fp32 accumulate; operands rounded RNE while staging)
// Operand layouts (all row-major fp32 in HBM):
//   AL 0: A stored [M][K] (k contiguous)     AL 1: A stored [K][M] (m contiguous)
//   BL 0: B stored [N][K] (k contiguous)     BL 1: B stored [K][N] (n contiguous)
// so y = x W^T is (0,0), dx = dy W is (0,1), dW = dy^T x is (1,1) with A = dy, B = x.
// Tile 128x128x32, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 MFMA tiles of 32x32.
// Split-K writes one fp32 slab per split (deterministic; reduced by reduce_slabs).
//
// MFMA k-order freedom: the products only have to pair A(m,k) with B(k,n) for the same k, so a
// lane may take 4 (fp32) / 8 (bf16) CONSECUTIVE k's from one 16-byte LDS read and feed them to
// consecutive MFMA steps, as long as both operands use the same map.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LDK32 = BK + 4;    // fp32, k-contiguous tile row stride (floats): 36 -> conflict-free b128
constexpr int LDM32 = BM + 4;    // fp32, m-contiguous tile row stride (floats)
constexpr int LDK16 = BK + 8;    // bf16, k-contiguous tile row stride (elements): 40 -> 80 B rows

struct GemmArgs {
    const float* A; const float* B; float* C; const float* bias; const float* aux;
    float* colsum;      // optional (A layout 1 only): colsum[m] = sum_k A(m,k), one slab per split like C
    int M, N, K, lda, ldb, ldc, ldaux;
    int act;            // 0 none, 1 tanh, 2 multiply by (1 - aux^2)  (tanh backward)
    int ktiles_per_split;
    long long slab_stride;
    int nx, ny, nz;     // logical grid: column blocks, row blocks, k-splits
    int plain;          // block order: 0 = panel groups per XCD (see the kernel), 1 = plain (fewer groups than XCDs)
};

// Staging is split (register staging, issue-early / write-late): load() puts a thread's 4 float4 of the
// NEXT k-tile in flight before the MFMAs of the current one, store() converts and writes them to LDS after.
// The four staged float4 are separate scalars (an array member ended up in scratch) and every load is
// unconditional (address clamped, value selected), so the 8 loads of a tile issue back to back.
#define SV_R4 float4 &v0, float4 &v1, float4 &v2, float4 &v3
#define SV_CR4 const float4 &v0, const float4 &v1, const float4 &v2, const float4 &v3

__device__ __forceinline__ float4 ld4(const float* g, size_t off, bool ok) {
    const float4 t = *reinterpret_cast<const float4*>(g + (ok ? off : 0));
    return ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
}

__device__ __forceinline__ unsigned pk2(float a, float b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 h2 __attribute__((ext_vector_type(2)));
    const f2 v = {a, b};
    const h2 h = __builtin_convertvector(v, h2);
    return *reinterpret_cast<const unsigned*>(&h);
}

template <int PREC, int L>
struct Tile;  // LDS image of one operand tile (rows = m or n index, BK k's)

// k-contiguous source: thread (row = idx>>3, c4 = idx&7), idx = tid + 256 i: 8 lanes read one row's 128 B
struct TileK {
    static __device__ __forceinline__ void load(SV_R4, const float* g, int ld, int r0, int k0, int R, int K, int tid) {
        const int row = tid >> 3, gk = k0 + 4 * (tid & 7);
        const bool kok = gk < K;
        v0 = ld4(g, (size_t)(r0 + row) * ld + gk, kok && (r0 + row) < R);
        v1 = ld4(g, (size_t)(r0 + row + 32) * ld + gk, kok && (r0 + row + 32) < R);
        v2 = ld4(g, (size_t)(r0 + row + 64) * ld + gk, kok && (r0 + row + 64) < R);
        v3 = ld4(g, (size_t)(r0 + row + 96) * ld + gk, kok && (r0 + row + 96) < R);
    }
};

// ---- fp32 tiles ----
template <>
struct Tile<0, 0> : TileK {  // image [128][36] floats
    static constexpr int BYTES = 128 * LDK32 * 4;
    static __device__ __forceinline__ void store(SV_CR4, float* s, int tid) {
        float* d = s + (tid >> 3) * LDK32 + 4 * (tid & 7);
        *reinterpret_cast<float4*>(d) = v0;
        *reinterpret_cast<float4*>(d + 32 * LDK32) = v1;
        *reinterpret_cast<float4*>(d + 64 * LDK32) = v2;
        *reinterpret_cast<float4*>(d + 96 * LDK32) = v3;
    }
    // fragment for MFMA steps kb..: lane (i = l&31, kk = l>>5) -> 4 consecutive k's
    static __device__ __forceinline__ float4 frag(const float* s, int row, int kb, int kk) {
        return *reinterpret_cast<const float4*>(s + row * LDK32 + kb + 4 * kk);
    }
};
template <>
struct Tile<0, 1> {  // m-contiguous source: image [32][132] floats; thread (k = tid>>5 + 8 i, c4 = tid&31)
    static constexpr int BYTES = BK * LDM32 * 4;
    static __device__ __forceinline__ void load(SV_R4, const float* g, int ld, int r0, int k0, int R, int K, int tid) {
        const int k = k0 + (tid >> 5), gr = r0 + 4 * (tid & 31);
        const bool rok = gr < R;
        v0 = ld4(g, (size_t)k * ld + gr, rok && k < K);
        v1 = ld4(g, (size_t)(k + 8) * ld + gr, rok && (k + 8) < K);
        v2 = ld4(g, (size_t)(k + 16) * ld + gr, rok && (k + 16) < K);
        v3 = ld4(g, (size_t)(k + 24) * ld + gr, rok && (k + 24) < K);
    }
    static __device__ __forceinline__ void store(SV_CR4, float* s, int tid) {
        float* d = s + (tid >> 5) * LDM32 + 4 * (tid & 31);
        *reinterpret_cast<float4*>(d) = v0;
        *reinterpret_cast<float4*>(d + 8 * LDM32) = v1;
        *reinterpret_cast<float4*>(d + 16 * LDM32) = v2;
        *reinterpret_cast<float4*>(d + 24 * LDM32) = v3;
    }
    static __device__ __forceinline__ float4 frag(const float* s, int row, int kb, int kk) {
        const float* p = s + (kb + 4 * kk) * LDM32 + row;
        return make_float4(p[0], p[LDM32], p[2 * LDM32], p[3 * LDM32]);
    }
};

// ---- bf16 tiles: image always [128][40] bf16, k contiguous ----
template <>
struct Tile<1, 0> : TileK {
    static constexpr int BYTES = 128 * LDK16 * 2;
    static __device__ __forceinline__ void store(SV_CR4, unsigned short* s, int tid) {
        unsigned short* d = s + (tid >> 3) * LDK16 + 4 * (tid & 7);
        *reinterpret_cast<uint2*>(d) = make_uint2(pk2(v0.x, v0.y), pk2(v0.z, v0.w));
        *reinterpret_cast<uint2*>(d + 32 * LDK16) = make_uint2(pk2(v1.x, v1.y), pk2(v1.z, v1.w));
        *reinterpret_cast<uint2*>(d + 64 * LDK16) = make_uint2(pk2(v2.x, v2.y), pk2(v2.z, v2.w));
        *reinterpret_cast<uint2*>(d + 96 * LDK16) = make_uint2(pk2(v3.x, v3.y), pk2(v3.z, v3.w));
    }
};
// m-contiguous source, transposed while staging: a thread owns a 4k x 4m block, kg = tid&7, mg = tid>>3.
// A wave-load touches 8 k-rows x 128 B; the 8-byte LDS writes of a 16-lane group fall in two rows 80 dwords
// apart: conflict-free (an mg-fastest map was 16-way conflicted: SQ_LDS_BANK_CONFLICT = 77 % of LDS cycles).
template <>
struct Tile<1, 1> {
    static constexpr int BYTES = 128 * LDK16 * 2;
    static __device__ __forceinline__ void load(SV_R4, const float* g, int ld, int r0, int k0, int R, int K, int tid) {
        const int k = k0 + 4 * (tid & 7), gr = r0 + 4 * (tid >> 3);
        const bool rok = gr < R;
        v0 = ld4(g, (size_t)k * ld + gr, rok && k < K);
        v1 = ld4(g, (size_t)(k + 1) * ld + gr, rok && (k + 1) < K);
        v2 = ld4(g, (size_t)(k + 2) * ld + gr, rok && (k + 2) < K);
        v3 = ld4(g, (size_t)(k + 3) * ld + gr, rok && (k + 3) < K);
    }
    static __device__ __forceinline__ void store(SV_CR4, unsigned short* s, int tid) {
        unsigned short* d = s + (4 * (tid >> 3)) * LDK16 + 4 * (tid & 7);
        *reinterpret_cast<uint2*>(d) = make_uint2(pk2(v0.x, v1.x), pk2(v2.x, v3.x));
        *reinterpret_cast<uint2*>(d + LDK16) = make_uint2(pk2(v0.y, v1.y), pk2(v2.y, v3.y));
        *reinterpret_cast<uint2*>(d + 2 * LDK16) = make_uint2(pk2(v0.z, v1.z), pk2(v2.z, v3.z));
        *reinterpret_cast<uint2*>(d + 3 * LDK16) = make_uint2(pk2(v0.w, v1.w), pk2(v2.w, v3.w));
    }
};

template <int PREC, int AL, int BL>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using TA = Tile<PREC, AL>;
    using TB = Tile<PREC, BL>;
    using elem_t = typename std::conditional<PREC == 0, float, unsigned short>::type;
    // two LDS stages: the next k-tile is written to the other stage while this one is read -> one barrier per k-tile
    constexpr int STAGE = TA::BYTES + TB::BYTES;
    elem_t* sA = reinterpret_cast<elem_t*>(smem);
    elem_t* sB = reinterpret_cast<elem_t*>(smem + TA::BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    // XCD-aware block order (1-D grid; workgroups are dealt round-robin over the 8 XCDs, so id % 8 names an XCD):
    // the blocks that read the same operand panel -- the nx*ny blocks of one k-split, or the nx blocks of one
    // row panel when there is no split -- get consecutive slots of ONE XCD, whose L2 then serves the panel's
    // re-reads (in x-fastest order they sat on 8 different XCDs and every re-read went out to the fabric:
    // 250 MB fetched for 80 MB of weight-gradient operands).
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int bx, by, bz;
    if (a.plain) {
        // fewer panel groups than XCDs (e.g. the 256 ranked rows of a scoring batch against 20 000 items: two row
        // panels): one group per XCD would leave the other XCDs without work -- 314 workgroups on 64 CUs, 38 us for a
        // 20 MB result -- so the blocks are dealt in plain order over the whole chip
        const int gsize = a.nx * a.ny, mem = blockIdx.x % gsize;
        bz = blockIdx.x / gsize; bx = mem % a.nx; by = mem / a.nx;
    } else if (a.nz > 1) {
        const int gsize = a.nx * a.ny, g = xcd + 8 * (slot / gsize), mem = slot % gsize;
        if (g >= a.nz) return;
        bz = g; bx = mem % a.nx; by = mem / a.nx;
    } else {
        const int g = xcd + 8 * (slot / a.nx);
        if (g >= a.ny) return;
        bz = 0; by = g; bx = slot % a.nx;
    }
    const int m0 = by * BM, n0 = bx * BN;
    const int kt_begin = bz * a.ktiles_per_split;
    int kt_end = kt_begin + a.ktiles_per_split;
    const int kt_total = (a.K + BK - 1) / BK;
    if (kt_end > kt_total) kt_end = kt_total;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int li = lane & 31, kk = lane >> 5;
    // bias-gradient fold: the first column-tile's workgroups also sum their A tile over k
    const bool do_cs = (AL == 1) && a.colsum != nullptr && bx == 0 && tid < BM;
    float cs = 0.f;

    float4 a0, a1, a2, a3, b0, b1, b2, b3;
    if (kt_begin < kt_end) {
        TA::load(a0, a1, a2, a3, a.A, a.lda, m0, kt_begin * BK, a.M, a.K, tid);
        TB::load(b0, b1, b2, b3, a.B, a.ldb, n0, kt_begin * BK, a.N, a.K, tid);
        TA::store(a0, a1, a2, a3, sA, tid);
        TB::store(b0, b1, b2, b3, sB, tid);
    }
    __syncthreads();

    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool more = (kt + 1) < kt_end;
        elem_t* sAn = reinterpret_cast<elem_t*>(smem + (((kt - kt_begin) & 1) ^ 1) * STAGE);
        elem_t* sBn = reinterpret_cast<elem_t*>(smem + (((kt - kt_begin) & 1) ^ 1) * STAGE + TA::BYTES);
        if (more) {   // next tile's global loads fly under this tile's MFMAs
            TA::load(a0, a1, a2, a3, a.A, a.lda, m0, (kt + 1) * BK, a.M, a.K, tid);
            TB::load(b0, b1, b2, b3, a.B, a.ldb, n0, (kt + 1) * BK, a.N, a.K, tid);
        }
        if (AL == 1 && do_cs) {
            if constexpr (PREC == 0) {
#pragma unroll
                for (int k = 0; k < BK; ++k) cs += reinterpret_cast<const float*>(sA)[k * LDM32 + tid];
            } else {
#pragma unroll
                for (int k8 = 0; k8 < BK; k8 += 8) {
                    const bf16x8 v = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const unsigned short*>(sA) + tid * LDK16 + k8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) cs += sv_bf2f((unsigned short)v[j]);
                }
            }
        }
        if constexpr (PREC == 0) {
#pragma unroll
            for (int kb = 0; kb < BK; kb += 8) {
                float4 fa[2], fb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[i] = TA::frag(sA, wm + 32 * i + li, kb, kk);
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[j] = TB::frag(sB, wn + 32 * j + li, kb, kk);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                                reinterpret_cast<const float*>(&fa[i])[s],
                                reinterpret_cast<const float*>(&fb[j])[s], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kb = 0; kb < BK; kb += 16) {
                bf16x8 fa[2], fb[2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    fa[i] = *reinterpret_cast<const bf16x8*>(sA + (wm + 32 * i + li) * LDK16 + kb + 8 * kk);
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    fb[j] = *reinterpret_cast<const bf16x8*>(sB + (wn + 32 * j + li) * LDK16 + kb + 8 * kk);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
            }
        }
        if (more) {
            TA::store(a0, a1, a2, a3, sAn, tid);
            TB::store(b0, b1, b2, b3, sBn, tid);
        }
        __syncthreads();   // the next stage is complete; everyone is done reading this one
        sA = sAn;
        sB = sBn;
    }

    if (AL == 1 && do_cs && m0 + tid < a.M)
        a.colsum[(size_t)bz * a.slab_stride + m0 + tid] = cs;

    // epilogue: C layout of a 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    float* C = a.C + (size_t)bz * a.slab_stride;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn + 32 * j + li;
        if (n >= a.N) continue;
        const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kk;
                if (m < a.M) {
                    float v = acc[i][j][r] + bv;
                    if (a.act == 1) v = tanhf(v);
                    if (a.act == 2) { const float y = a.aux[(size_t)m * a.ldaux + n]; v *= (1.0f - y * y); }
                    C[(size_t)m * a.ldc + n] = v;
                }
            }
        }
    }
}

template <int PREC, int AL, int BL>
int launch(GemmArgs a, int nsplit, hipStream_t st) {
    a.nx = (a.N + BN - 1) / BN;
    a.ny = (a.M + BM - 1) / BM;
    a.nz = nsplit;
    const int ngroups = nsplit > 1 ? nsplit : a.ny, gsize = nsplit > 1 ? a.nx * a.ny : a.nx;
    a.plain = ngroups < 8 ? 1 : 0;
    dim3 grid(a.plain ? a.nx * a.ny * a.nz : 8 * ((ngroups + 7) / 8) * gsize);
    size_t lds = 2 * (Tile<PREC, AL>::BYTES + Tile<PREC, BL>::BYTES);
    static bool attr_done = false;
    if (!attr_done && lds > 65536) {
        hipError_t e = hipFuncSetAttribute((const void*)gemm_kernel<PREC, AL, BL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return svae_fail((int)e, "svae_gemm: LDS attribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    hipLaunchKernelGGL((gemm_kernel<PREC, AL, BL>), grid, dim3(256), lds, st, a);
    SV_LAUNCH_CHECK();
    return 0;
}

// (8 slabs in flight per thread: summed one dependent load at a time the kernel was latency-bound, 13 us for 22 MB)
__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, float* __restrict__ out, long long n4,
                                    int nsplit, long long slab_stride) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long step = (long long)gridDim.x * blockDim.x;
    for (; i < n4; i += step) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int z0 = 0; z0 < nsplit; z0 += 8) {
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int z = z0 + j < nsplit ? z0 + j : 0;
                v[j] = reinterpret_cast<const float4*>(slabs + z * slab_stride)[i];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)     // fixed order: bitwise reproducible
                if (z0 + j < nsplit) { s.x += v[j].x; s.y += v[j].y; s.z += v[j].z; s.w += v[j].w; }
        }
        reinterpret_cast<float4*>(out)[i] = s;
    }
}

}  // namespace

extern "C" int svae_gemm(const float* A, const float* B, float* C, const float* bias, const float* aux,
                         float* colsum, int M, int N, int K, int lda, int ldb, int ldc, int ldaux, int a_layout,
                         int b_layout, int act, int prec, int nsplit, long long slab_stride, void* stream) {
    SV_CHECK_ARG(A && B && C, "svae_gemm: null pointer");
    SV_CHECK_ARG(M > 0 && N > 0 && K > 0, "svae_gemm: bad shape %d %d %d", M, N, K);
    SV_CHECK_ARG((lda % 4) == 0 && (ldb % 4) == 0, "svae_gemm: lda/ldb must be multiples of 4");
    SV_CHECK_ARG((a_layout == 1 && b_layout == 1) || (K % 4) == 0, "svae_gemm: k-contiguous operands need K %% 4 == 0");
    SV_CHECK_ARG(a_layout == 0 || (M % 4) == 0, "svae_gemm: m-contiguous A needs M %% 4 == 0");
    SV_CHECK_ARG(b_layout == 0 || (N % 4) == 0, "svae_gemm: n-contiguous B needs N %% 4 == 0");
    SV_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "svae_gemm: operands must be 16-byte aligned");
    SV_CHECK_ARG(nsplit >= 1 && (nsplit == 1 || slab_stride >= (long long)M * ldc), "svae_gemm: bad split");
    SV_CHECK_ARG(nsplit == 1 || (bias == nullptr && act == 0), "svae_gemm: split-K takes no epilogue");
    SV_CHECK_ARG(act != 2 || aux, "svae_gemm: act 2 needs aux");
    SV_CHECK_ARG(colsum == nullptr || a_layout == 1, "svae_gemm: colsum needs an m-contiguous A");
    SV_CHECK_ARG(prec == 0 || prec == 1, "svae_gemm: prec must be 0 (fp32) or 1 (bf16)");
    GemmArgs a{A, B, C, bias, aux, colsum, M, N, K, lda, ldb, ldc, ldaux, act, 0, slab_stride};
    const int kt = (K + BK - 1) / BK;
    a.ktiles_per_split = (kt + nsplit - 1) / nsplit;
    hipStream_t st = (hipStream_t)stream;
    const int key = prec * 4 + a_layout * 2 + b_layout;
    switch (key) {
        case 0: return launch<0, 0, 0>(a, nsplit, st);
        case 1: return launch<0, 0, 1>(a, nsplit, st);
        case 2: return launch<0, 1, 0>(a, nsplit, st);
        case 3: return launch<0, 1, 1>(a, nsplit, st);
        case 4: return launch<1, 0, 0>(a, nsplit, st);
        case 5: return launch<1, 0, 1>(a, nsplit, st);
        case 6: return launch<1, 1, 0>(a, nsplit, st);
        case 7: return launch<1, 1, 1>(a, nsplit, st);
    }
    return svae_fail(-1, "svae_gemm: bad layout");
}

extern "C" int svae_reduce_slabs(const float* slabs, float* out, long long n, int nsplit, long long slab_stride,
                                 void* stream) {
    SV_CHECK_ARG(slabs && out && n > 0 && (n % 4) == 0 && nsplit >= 1 && (slab_stride % 4) == 0,
                 "svae_reduce_slabs: bad arguments");
    long long n4 = n / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, slabs, out, n4, nsplit,
                       slab_stride);
    SV_LAUNCH_CHECK();
    return 0;
}

// Reference interface replaced: Decoder.linear2 (nb:710) as Model.forward returns it -- the dense logits
// dec_out[r][i] = sum_k h2[r][k] W3[i][k] + b3[i] of n rows of the decoder hidden (h2 [n][160], rows of 150 values
// padded with zeros; W3 [I][160] as stored in the flat parameter buffer; logits [n][ldl], ldl >= I).  The notebook
// surface and evaluate() call it for the rows they rank; the training step never materialises logits.
// prec 0: exact fp32 MFMA (the parity mode), 1: bf16 operands.
extern "C" int svae_logits_fwd(const float* h2, const float* w3, const float* b3, float* logits, int n, int I, int ldl,
                               int prec, void* stream) {
    SV_CHECK_ARG(h2 && w3 && b3 && logits && n > 0 && I > 0 && ldl >= I, "svae_logits_fwd: bad arguments");
    return svae_gemm(h2, w3, logits, b3, nullptr, nullptr, n, I, SV_DP, SV_DP, SV_DP, ldl, 0, 0, 0, 0, prec, 1, 0, stream);
}
