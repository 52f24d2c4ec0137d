// Persistent-over-timesteps GRU recurrence for gfx950 (reference: nn.GRU call nb:759, def nb:724-727;
// gate order r,z,n; h0 = 0).
//
// Layout: a batch is TIME-MAJOR PACKED.  Users are sorted by length (descending); at timestep t the
// first nact[t] users are active and their rows are row_off[t] .. row_off[t]+nact[t]-1.  One workgroup
// owns 16 consecutive users for ALL timesteps (users are independent, so there is no inter-workgroup
// exchange).  4 waves, one per SIMD; wave w owns hidden-unit tiles jt = w, w+4, w+8, w+12 (< 14) and
// all three gates of those units, so the gate math needs no cross-wave exchange: a lane owns hidden
// unit j = 16*jt + (lane & 15) of users m = 4*(lane >> 4) + r, r = 0..3 (the MFMA C layout), keeps
// h[m][j] in a register (fp32) and publishes it to LDS as the next step's matmul operand.
//
// The input-side gates gi = emb W_ih^T + b_ih for all rows are computed beforehand by one big GEMM.
// This file: the recurrent part, gh = h_{t-1} W_hh^T + b_hh, the fused gate math, and the reverse-time
// BPTT that produces dgi / dgh for the weight-gradient GEMMs.
//
//   fp32 mode: v_mfma_f32_16x16x4_f32, W_hh streamed from L2 every step (480 KB fp32 does not fit a CU)
//   bf16 mode: v_mfma_f32_16x16x32_bf16, W_hh fragments resident in VGPRs/AGPRs for the whole sequence
//              (336 registers per lane), h / dgh exchanged through a ping-pong bf16 LDS image: ONE
//              barrier per timestep.
//
// Every per-row output buffer must have R + 1 rows: row R is a trash row that absorbs the stores of
// inactive users in the branch-free bf16 kernels.
// Saved for backward, per row: gates4 [R][224][4] = (r, z, n, gh_n) interleaved (one 16-byte access
// per owned element), hs [R][224], hprev [R][224] (= h_{t-1}, the B operand of the dW_hh GEMM).
#include "common.h"

#ifdef SV_GRU_TIMING
__device__ unsigned long long g_gru_dbg[8];
#define SV_TS(i_) { __builtin_amdgcn_sched_barrier(0); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ts_acc[i_] += now_ - ts_last; ts_last = now_; __builtin_amdgcn_sched_barrier(0); }
#else
#define SV_TS(i_)
#endif
namespace {

constexpr int MB = 16;             // MFMA M: user rows of a workgroup's tiles
constexpr int MU = 4;              // bf16 kernels: users a workgroup really owns (rows MU..15 of its tiles are
                                   // padding).  The step is bound by the CU's 64 B/clk L1<->L2 path (gi, saved
                                   // gates, dgi/dgh: >100 KB per step at 16 users), so halving the users per CU
                                   // and doubling the CUs in use shortens the recurrence's critical path.
static_assert(MU <= 4, "element roles: lane 16 u + n owns user u, so a workgroup has at most 4 users");
constexpr int NJT = SV_HP / 16;    // 14 hidden-unit tiles of 16
constexpr int HLD = SV_HP + 4;     // fp32 LDS row stride (floats) of the h image
constexpr int GLD = SV_G + 4;      // fp32 LDS row stride (floats) of the dgh image
constexpr int HLDB = SV_HP + 8;    // bf16 LDS row stride (elements) of the h image   (464 B rows)
constexpr int GLDB = SV_G + 8;     // bf16 LDS row stride (elements) of the dgh image (1360 B rows)

// Step scalars (nact[t], row_off[t]) are read through the constant address space: a uniform address there
// is an s_load (lgkmcnt), not a vector load whose s_waitcnt vmcnt would also wait for every older store
// of the step before (loads and stores retire through one in-order counter on gfx9).
typedef __attribute__((address_space(4))) const int* sv_const_int_p;
__device__ __forceinline__ sv_const_int_p sv_scalar_ptr(const int* p) {
    return (sv_const_int_p)(unsigned long long)p;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned SV_BUF_FLAGS = 0x00020000u;   // raw buffer, 32-bit data format (gfx9 word 3)
constexpr unsigned SV_BUF_OOB = 0x80000000u;     // offset of a masked lane: beyond every array (all < 2 GiB)

struct GruArgs {
    const float* gi;      // [R][672]   (fwd)   input-side pre-activations incl. b_ih
    const float* whh;     // fwd: W_hh [672][224];  bwd: W_hh^T [224][672]
    const float* bhh;     // [672]
    const int* row_off;   // [Tmax+1]
    const int* nact;      // [Tmax]
    float* hs;            // [R][224]     h_t        (fwd out)
    float* hprev;         // [R][224]     h_{t-1}    (fwd out / bwd in)
    float* gates4;        // [R][224][4]  r, z, n, gh_n  (fwd out / bwd in)
    const float* dhs;     // [R][224]  (bwd in) dL/dh_t from the encoder side
    float* dgi;           // [R][672]  (bwd out) d/d(gi) = (dr_pre, dz_pre, dn_pre)
    float* dgh;           // [R][672]  (bwd out) d/d(gh) = (dr_pre, dz_pre, dn_pre * r)
    int B, Tmax;
    int merged;           // backward, bf16 kernel: dgi is ONE array [R][4*224] = (dr, dz, dn, dn*r); dgh is unused
                          // (1: fp32 elements, 2: bf16 elements)
    int t0, t1;           // bf16 kernels: timesteps [t0, t1).  Forward: t0 > 0 continues from hs (h_{t0-1}).  Backward
                          // (runs t1-1 down to t0): t1 < Tmax continues from dh_carry, t0 > 0 leaves dL/dh_{t0-1} there
    float* dh_carry;      // [B][224]  backward only: the recurrent part of dL/dh across a range boundary
};

// forward gate math for one owned element; returns h_t
template <bool FAST = false>
__device__ __forceinline__ float gate_fwd(const GruArgs& a, size_t row, int j, float gir, float giz, float gin,
                                          float ghr, float ghz, float ghn, float hp) {
    const float rg = FAST ? sv_sigmoid_fast(gir + ghr) : sv_sigmoid(gir + ghr);
    const float zg = FAST ? sv_sigmoid_fast(giz + ghz) : sv_sigmoid(giz + ghz);
    const float ng = FAST ? sv_tanh_fast(gin + rg * ghn) : tanhf(gin + rg * ghn);
    const float hn = (1.0f - zg) * ng + zg * hp;
    a.hs[row * SV_HP + j] = hn;
    a.hprev[row * SV_HP + j] = hp;
    *reinterpret_cast<float4*>(a.gates4 + (row * SV_HP + j) * 4) = make_float4(rg, zg, ng, ghn);
    return hn;
}

// backward gate math for one owned element: d = dL/dh_t (total).  Writes dgi/dghn, returns the three
// dgh values through refs and d*z (the direct carry into dh_{t-1}).
__device__ __forceinline__ float gate_bwd(const GruArgs& a, size_t row, int j, float d, float4 gt, float hp,
                                          float& drp, float& dzp, float& dnr) {
    const float rg = gt.x, zg = gt.y, ng = gt.z, ghn = gt.w;
    const float dn = d * (1.0f - zg);
    const float dzg = d * (hp - ng);
    const float dnp = dn * (1.0f - ng * ng);
    const float drg = dnp * ghn;
    dzp = dzg * zg * (1.0f - zg);
    drp = drg * rg * (1.0f - rg);
    dnr = dnp * rg;
    float* o1 = a.dgi + row * SV_G;
    o1[j] = drp;
    o1[SV_HP + j] = dzp;
    o1[2 * SV_HP + j] = dnp;
    float* o2 = a.dgh + row * SV_G;
    o2[j] = drp;
    o2[SV_HP + j] = dzp;
    o2[2 * SV_HP + j] = dnr;
    return d * zg;
}

// ------------------------------------------------------------------------------------------------
// fp32 forward
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gru_fwd_f32_kernel(GruArgs a) {
    __shared__ __attribute__((aligned(16))) float h_lds[MB * HLD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int u0 = blockIdx.x * MB;
    const int ln = lane & 15, lk = lane >> 4;

    for (int i = tid; i < MB * HLD; i += 256) h_lds[i] = 0.f;
    float hreg[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) hreg[q][r] = 0.f;
    __syncthreads();

    for (int t = 0; t < a.Tmax; ++t) {
        const int na = a.nact[t];
        if (na <= u0) break;  // uniform: lengths are sorted, nobody in this workgroup is active any more
        const int rbase = a.row_off[t] + u0;

        f32x4 acc[4][3];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[q][g] = (f32x4){0.f, 0.f, 0.f, 0.f};

        if (t > 0) {
            for (int kb = 0; kb < SV_HP; kb += 16) {
                const float4 av = *reinterpret_cast<const float4*>(&h_lds[ln * HLD + kb + 4 * lk]);
                float4 bv[4][3];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int jt = wave + 4 * q;
                    if (jt < NJT) {
#pragma unroll
                        for (int g = 0; g < 3; ++g)
                            bv[q][g] = *reinterpret_cast<const float4*>(
                                a.whh + (size_t)(g * SV_HP + 16 * jt + ln) * SV_HP + kb + 4 * lk);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int jt = wave + 4 * q;
                    if (jt < NJT) {
#pragma unroll
                        for (int g = 0; g < 3; ++g) {
                            acc[q][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv[q][g].x, acc[q][g], 0, 0, 0);
                            acc[q][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv[q][g].y, acc[q][g], 0, 0, 0);
                            acc[q][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv[q][g].z, acc[q][g], 0, 0, 0);
                            acc[q][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv[q][g].w, acc[q][g], 0, 0, 0);
                        }
                    }
                }
            }
        }
        __syncthreads();  // every wave has finished reading h_{t-1} from LDS

        // gate math.  Loads are unconditional (row clamped to an active one) and issued for all owned
        // elements first: a load inside the per-element `if` would serialise one memory round trip per element.
        float gv[4][4][3];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int jt = wave + 4 * q;
            if (jt < NJT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 4 * lk + r;
                    const float* gp = a.gi + (size_t)(rbase + ((u0 + m < na) ? m : 0)) * SV_G + 16 * jt + ln;
#pragma unroll
                    for (int g = 0; g < 3; ++g) gv[q][r][g] = gp[g * SV_HP];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int jt = wave + 4 * q;
            if (jt < NJT) {
                const int j = 16 * jt + ln;
                const float br = a.bhh[j], bz = a.bhh[SV_HP + j], bn = a.bhh[2 * SV_HP + j];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 4 * lk + r;
                    if (u0 + m < na) {
                        const size_t row = (size_t)(rbase + m);
                        const float hn = gate_fwd(a, row, j, gv[q][r][0], gv[q][r][1], gv[q][r][2],
                                                  acc[q][0][r] + br, acc[q][1][r] + bz, acc[q][2][r] + bn, hreg[q][r]);
                        hreg[q][r] = hn;
                        h_lds[m * HLD + j] = hn;
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// fp32 backward (BPTT)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gru_bwd_f32_kernel(GruArgs a) {
    extern __shared__ __attribute__((aligned(16))) float g_lds[];  // [MB][GLD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int u0 = blockIdx.x * MB;
    const int ln = lane & 15, lk = lane >> 4;

    int Tw = 0;  // timesteps in which this workgroup has an active user (nact is non-increasing)
    while (Tw < a.Tmax && a.nact[Tw] > u0) ++Tw;

    float dh[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) dh[q][r] = 0.f;

    for (int t = Tw - 1; t >= 0; --t) {
        const int na = a.nact[t];
        const int rbase = a.row_off[t] + u0;
        float carry[4][4];
        float4 sg[4][4];
        float shp[4][4], sdh[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int jt = wave + 4 * q;
            if (jt < NJT) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 4 * lk + r;
                    const size_t e = (size_t)(rbase + ((u0 + m < na) ? m : 0)) * SV_HP + 16 * jt + ln;
                    sg[q][r] = *reinterpret_cast<const float4*>(a.gates4 + e * 4);
                    shp[q][r] = a.hprev[e];
                    sdh[q][r] = a.dhs[e];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int jt = wave + 4 * q;
            if (jt < NJT) {
                const int j = 16 * jt + ln;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 4 * lk + r;
                    float drp = 0.f, dzp = 0.f, dnr = 0.f;
                    carry[q][r] = 0.f;
                    if (u0 + m < na) {
                        const size_t row = (size_t)(rbase + m);
                        carry[q][r] = gate_bwd(a, row, j, dh[q][r] + sdh[q][r], sg[q][r], shp[q][r], drp, dzp, dnr);
                    }
                    g_lds[m * GLD + j] = drp;
                    g_lds[m * GLD + SV_HP + j] = dzp;
                    g_lds[m * GLD + 2 * SV_HP + j] = dnr;
                }
            }
        }
        __syncthreads();
        if (t > 0) {
            f32x4 acc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int kb = 0; kb < SV_G; kb += 16) {
                const float4 av = *reinterpret_cast<const float4*>(&g_lds[ln * GLD + kb + 4 * lk]);
                float4 bv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int jt = wave + 4 * q;
                    if (jt < NJT)
                        bv[q] = *reinterpret_cast<const float4*>(a.whh + (size_t)(16 * jt + ln) * SV_G + kb + 4 * lk);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int jt = wave + 4 * q;
                    if (jt < NJT) {
                        acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv[q].x, acc[q], 0, 0, 0);
                        acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv[q].y, acc[q], 0, 0, 0);
                        acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bv[q].z, acc[q], 0, 0, 0);
                        acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bv[q].w, acc[q], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) dh[q][r] = carry[q][r] + acc[q][r];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 kernels: weights resident in registers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    const bf16v2 h = __builtin_convertvector((f32x2){a, b}, bf16v2);
    return __builtin_bit_cast(unsigned, h);
}

__device__ __forceinline__ bf16x8 load_w8(const float* p) {
    const float4 lo = *reinterpret_cast<const float4*>(p);
    const float4 hi = *reinterpret_cast<const float4*>(p + 4);
    bf16x8 v;
    v[0] = (short)sv_f2bf(lo.x); v[1] = (short)sv_f2bf(lo.y); v[2] = (short)sv_f2bf(lo.z); v[3] = (short)sv_f2bf(lo.w);
    v[4] = (short)sv_f2bf(hi.x); v[5] = (short)sv_f2bf(hi.y); v[6] = (short)sv_f2bf(hi.z); v[7] = (short)sv_f2bf(hi.w);
    return v;
}

// Split instantiation ("prec 2"): an fp32 value as bf16 hi + mid + lo parts (all 24 significant bits), and a product by
// its six leading terms on the bf16 MFMA pipe, small terms first -- as in the split instantiations of the other
// fused kernels (proj.hip, mlp.hip, rowgemm.hip, wgrad.hip).
__device__ __forceinline__ void split3(float v, unsigned short& h, unsigned short& m, unsigned short& l) {
    h = sv_f2bf(v);
    const float r1 = v - sv_bf2f(h);
    m = sv_f2bf(r1);
    l = sv_f2bf(r1 - sv_bf2f(m));
}
__device__ __forceinline__ void load_w8_split(const float* p, bf16x8& h, bf16x8& m, bf16x8& l) {
    const float4 lo = *reinterpret_cast<const float4*>(p);
    const float4 hi = *reinterpret_cast<const float4*>(p + 4);
    const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        unsigned short a, b, c;
        split3(v[i], a, b, c);
        h[i] = (short)a; m[i] = (short)b; l[i] = (short)c;
    }
}
__device__ __forceinline__ f32x4 mfma_split(const bf16x8 ah, const bf16x8 am, const bf16x8 al, const bf16x8 bh,
                                            const bf16x8 bm, const bf16x8 bl, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
    return c;
}

// Re-deal of a 16x16 accumulator tile whose rows 0..3 are the real users (registers 0..3 of lanes 0..15): lane
// 16 u + n receives row u, column n.  Two row swaps and one half swap in registers (gfx950's v_permlane16_swap /
// v_permlane32_swap; semantics checked by tools/probe/permlane_probe.hip) instead of a round trip through an LDS
// scratch (24 ds_write_b32 on 16 lanes, a wave barrier, 6 ds_read_b32 and their latencies on the step's chain).
__device__ __forceinline__ float redeal_rows(const f32x4 c) {
    // permlane16_swap(a, b)[0] = (a.row0, b.row0, a.row2, b.row2); permlane32_swap(a, b)[0] = (a.row0, a.row1, b.row0, b.row1)
    const auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(c[0]), __float_as_uint(c[1]), false, false);
    const auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(c[2]), __float_as_uint(c[3]), false, false);
    const auto s = __builtin_amdgcn_permlane32_swap(p[0], q[0], false, false);
    return __uint_as_float(s[0]);
}

// Register budget: a wave that owns 4 hidden tiles would need 336 registers for its W_hh fragments alone;
// with the per-step prefetch (gi / saved gates) that spills.  So every wave keeps THREE tiles in registers
// (252) and the fourth tile of waves 0 and 1 lives in LDS in fragment order (lane-linear 16-byte reads,
// conflict-free): 2 waves x 21 fragments x 1 KiB = 42 KiB.
constexpr int WL_FRAGS = 21;                       // fragments of the LDS-resident tile (fwd: 3 gates x 7; bwd: 21 k-steps)

// Forward: 8 waves (two per SIMD).  The step is bound by the SIMD's vector issue port (gate math), not by the
// MFMA pipe, and with one wave per SIMD the two only overlap as far as the compiler interleaves them; with
// two waves the hardware does it: one wave's gate math issues while the other's MFMAs execute.
// Wave w owns hidden tiles w and w+8 (< 14): tile w in registers (84), tile w+8 in LDS in fragment order
// (lane-linear 16-byte reads, conflict-free): 6 tiles x 21 KiB = 126 KiB.
constexpr int WLF_ELEMS = 6 * WL_FRAGS * 64 * 8;
constexpr int FWD_WAVES = 8;
constexpr int XS_FLOATS = 2 * 3 * 64;              // re-deal scratch of one wave: [tile][gate][user][unit]

// SPLIT: the parity instantiation of the SAME recurrence kernel (prec 2, used by PREC_SPLIT): same work division
// (MU users per workgroup, 8 waves, wave w owns hidden tiles w and w + 8, lane 16 u + n owns user u of unit n), same
// branch-free step (out-of-range buffer offsets for inactive users, input gates fetched one step ahead, accumulator rows
// re-dealt by lane swaps, one barrier per timestep, ping-pong h image), but fp32-class arithmetic: h is published as
// bf16 hi + mid + lo images, every W_hh fragment is split the same way and every product taken by its six leading
// terms; sigmoid / tanh are the exact ones of the fp32 kernel.  The weight fragments are NOT resident (three parts of
// 301 KB): they are re-read from L2 and split every step, and the per-tile alternation is off -- this instantiation is
// never timed.  The mid / lo images live where the bf16 instantiation keeps its LDS-resident weight tiles.
template <int NQ, bool SPLIT>  // NQ: number of hidden tiles this wave owns (2 for waves 0..5; 1 for waves 6,7)
__device__ __forceinline__ void gru_fwd_bf16_body(const GruArgs& a, unsigned short* h_bf, unsigned short* w_lds,
                                                  float* xs, int wave, int lane) {
    const int u0 = blockIdx.x * MU;
    const int ln = lane & 15, lk = lane >> 4;
    constexpr int KS = SV_HP / 32;  // 7 k-steps
    constexpr int NL = SPLIT ? 0 : NQ - 1;      // tiles in LDS
    constexpr bool ALT = !SPLIT;          // per-tile alternation (below)
    constexpr int HIMG = 2 * MB * HLDB;          // elements of one ping-pong h image pair
    unsigned short* const h_mid = w_lds;         // (SPLIT only)
    unsigned short* const h_lo = w_lds + HIMG;

    bf16x8 W[SPLIT ? 1 : 3][SPLIT ? 1 : KS];
    if constexpr (!SPLIT) {
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int s = 0; s < KS; ++s)
            W[g][s] = load_w8(a.whh + (size_t)(g * SV_HP + 16 * wave + ln) * SV_HP + 32 * s + 8 * lk);
    }
    unsigned short* wl = w_lds + (size_t)wave * (WL_FRAGS * 64 * 8) + lane * 8;
    if (NL > 0) {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int s = 0; s < KS; ++s)
                *reinterpret_cast<bf16x8*>(wl + (g * KS + s) * 512) =
                    load_w8(a.whh + (size_t)(g * SV_HP + 16 * (wave + 8) + ln) * SV_HP + 32 * s + 8 * lk);
    }
    float bias[NQ][3];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int g = 0; g < 3; ++g) bias[q][g] = a.bhh[g * SV_HP + 16 * (wave + 8 * q) + ln];

    // Element roles.  The MFMA leaves the 4 real users of a tile in the four accumulator registers of lanes
    // 0..15 (rows 0..3 of the C layout); lanes 16..63 hold padding rows.  Gate math, loads and stores on that
    // layout run 8 elements per lane of which 2 are real.  So the accumulators are re-dealt through a small
    // per-wave LDS scratch: lane 16 u + n takes user u of unit n -- one element per owned tile, every lane busy,
    // a quarter of the vector and memory instructions.
    float hreg[NQ];      // h_{t-1} of the owned elements (user lk, unit 16 (wave + 8 q) + ln)
#pragma unroll
    for (int q = 0; q < NQ; ++q) hreg[q] = 0.f;

    // The step body is ONE basic block apart from the scratch write of lanes 0..15: loads of inactive users get an
    // out-of-range buffer offset (read 0, stores dropped), the matmul also runs at t = 0 (h = 0).
    const sv_const_int_p nact_s = sv_scalar_ptr(a.nact), roff_s = sv_scalar_ptr(a.row_off);
    const unsigned R = (unsigned)roff_s[a.Tmax];
    const __amdgpu_buffer_rsrc_t gi_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.gi), 0, R * (SV_G * 4u), SV_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t hs_b = __builtin_amdgcn_make_buffer_rsrc(a.hs, 0, R * (SV_HP * 4u), SV_BUF_FLAGS);
    // (hprev / gates4 NULL = a forward that no backward follows (scoring): zero-sized buffers, every store of the saved
    //  state is out of range and dropped -- 5.4 KB per row that nobody reads; the loop itself is unchanged)
    const __amdgpu_buffer_rsrc_t hp_b = __builtin_amdgcn_make_buffer_rsrc(a.hprev, 0, a.hprev ? R * (SV_HP * 4u) : 0u, SV_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t g4_b = __builtin_amdgcn_make_buffer_rsrc(a.gates4, 0, a.gates4 ? R * (SV_HP * 16u) : 0u, SV_BUF_FLAGS);
    int na = nact_s[a.t0];
    int rbase = roff_s[a.t0] + u0;
    if (a.t0 > 0) {
        // resumed recurrence (the forward is launched in two time ranges so that the decoder side of the first
        // range's rows runs beside the second): h_{t0-1} of the owned elements comes back from hs (fp32, what the
        // first launch kept in registers) and its bf16 copy goes into the image step t0 reads
        const int nprev = nact_s[a.t0 - 1];
        if (lk < MU && u0 + lk < nprev) {
            const size_t row = (size_t)(roff_s[a.t0 - 1] + u0 + lk);
            unsigned short* hinit = h_bf + (a.t0 & 1) * (MB * HLDB);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int j = 16 * (wave + 8 * q) + ln;
                const float h = a.hs[row * SV_HP + j];
                hreg[q] = h;
                if constexpr (SPLIT) {
                    unsigned short p0, p1, p2;
                    split3(h, p0, p1, p2);
                    const int o = (a.t0 & 1) * (MB * HLDB) + lk * HLDB + j;
                    h_bf[o] = p0; h_mid[o] = p1; h_lo[o] = p2;
                } else
                hinit[lk * HLDB + j] = sv_f2bf(h);
            }
        }
        __syncthreads();
    }
#ifdef SV_GRU_TIMING
    unsigned long long ts_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long ts_last = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    float gi_nx[NQ][3];          // input-side gates of the step about to run
    {
        const bool act0 = lk < MU && u0 + lk < na;
        const unsigned og0 = act0 ? ((unsigned)(rbase + lk) * SV_G + 16 * wave + ln) * 4u : SV_BUF_OOB;
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int g = 0; g < 3; ++g)
                gi_nx[q][g] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(gi_b, og0 + (g * SV_HP + 128 * q) * 4, 0, 0));
    }
    for (int t = a.t0; t < a.t1; ++t) {
        if (na <= u0) break;
        SV_TS(0)
        const unsigned short* hcur = h_bf + (t & 1) * (MB * HLDB);
        unsigned short* hnxt = h_bf + ((t + 1) & 1) * (MB * HLDB);
        // next step's scalars: fetched now, consumed after the barrier
        const int tn = (t + 1 < a.Tmax) ? t + 1 : t;
        const int na_next = (t + 1 < a.Tmax) ? nact_s[tn] : 0;
        const int rbase_next = roff_s[tn] + u0;

        // 32-bit byte offsets of this lane's row (the host bounds B*Tmax so that they stay below 2^31)
        const bool act = lk < MU && u0 + lk < na;
        const unsigned row = rbase + lk;
        const unsigned ohs = act ? (row * SV_HP + 16 * wave + ln) * 4u : SV_BUF_OOB;
        const unsigned og4 = act ? (row * SV_HP + 16 * wave + ln) * 16u : SV_BUF_OOB;

        // Input-side gates: this step's were fetched one step ago; the NEXT step's are issued now, ahead of this
        // step's stores.  VMEM operations retire in order, so a load issued behind the stores of the step before
        // (as this step's own fetch used to be) cannot be waited for without waiting for those stores too.
        float gi[NQ][3];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int g = 0; g < 3; ++g) gi[q][g] = gi_nx[q][g];
        {
            const bool actn = lk < MU && u0 + lk < na_next && t + 1 < a.t1;
            const unsigned ogn = actn ? ((unsigned)(rbase_next + lk) * SV_G + 16 * wave + ln) * 4u : SV_BUF_OOB;
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    gi_nx[q][g] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(gi_b, ogn + (g * SV_HP + 128 * q) * 4, 0, 0));
        }

        f32x4 acc[NQ][3];
        float gh[NQ][3];
        float a_er = 0.f, a_ez = 0.f, a_rg = 0.f, a_zg = 0.f, a_en = 0.f, a_ng = 0.f, a_hn = 0.f;
        if constexpr (SPLIT) {
            const int ho = (t & 1) * (MB * HLDB) + ln * HLDB + 8 * lk;
            bf16x8 Ah[KS], Am[KS], Al[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                Ah[s] = *reinterpret_cast<const bf16x8*>(h_bf + ho + 32 * s);
                Am[s] = *reinterpret_cast<const bf16x8*>(h_mid + ho + 32 * s);
                Al[s] = *reinterpret_cast<const bf16x8*>(h_lo + ho + 32 * s);
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    f32x4 c = (f32x4){bias[q][g], bias[q][g], bias[q][g], bias[q][g]};
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        bf16x8 wh, wm, wl_;
                        load_w8_split(a.whh + (size_t)(g * SV_HP + 16 * (wave + 8 * q) + ln) * SV_HP + 32 * s + 8 * lk, wh, wm, wl_);
                        c = mfma_split(Ah[s], Am[s], Al[s], wh, wm, wl_, c);
                    }
                    acc[q][g] = c;
                }
        } else {
        bf16x8 A[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) A[s] = *reinterpret_cast<const bf16x8*>(hcur + ln * HLDB + 32 * s + 8 * lk);
#ifdef SV_GRU_TIMING
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
        SV_TS(1)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int g = 0; g < 3; ++g) acc[q][g] = (f32x4){bias[q][g], bias[q][g], bias[q][g], bias[q][g]};
        // LDS-resident fragments run through a ring of PD+1 groups (one group = the 3 gate fragments of one
        // k-step), fetched PD groups ahead of the MFMAs that use them.  sched_barrier mask 0x676: everything
        // but DS reads and MFMAs may cross, i.e. those two keep this source order while the gate math and
        // the global traffic float freely.
        constexpr int PD = 3, NGRP = NL * KS;
        bf16x8 wb[PD + 1][3];
#define SV_WLOAD(i_)                                                                                        \
    _Pragma("unroll") for (int g = 0; g < 3; ++g)                                                           \
        wb[(i_) % (PD + 1)][g] = *reinterpret_cast<const bf16x8*>(wl + (g * KS + (i_)) * 512);
#define SV_PIN_ORDER __builtin_amdgcn_sched_barrier(0x676);
#pragma unroll
        for (int i = 0; i < PD && i < NGRP; ++i) SV_WLOAD(i)
        SV_PIN_ORDER
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int g = 0; g < 3; ++g)
                acc[0][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s], W[g][s], acc[0][g], 0, 0, 0);
        // Per-tile alternation (waves that own two tiles): the gate math of tile 0 is issued in slices BETWEEN the
        // k-step groups of tile 1's MFMAs -- a wave issues in order, so vector work placed behind its last MFMA only
        // starts when the pipe has taken all of them; in the gaps it costs nothing and the partner wave's MFMAs fill
        // the pipe meanwhile.  sched_barrier(0) pins each slice to its gap.  Same arithmetic, same order per element.
#define SV_SLICE_FENCE __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NGRP; ++i) {
            SV_PIN_ORDER
            if (i + PD < NGRP) SV_WLOAD(i + PD)
#pragma unroll
            for (int g = 0; g < 3; ++g)
                acc[NQ - 1][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[i], wb[i % (PD + 1)][g], acc[NQ - 1][g], 0, 0, 0);
            if (NQ == 2 && ALT) {
                SV_SLICE_FENCE
                if (i == 0) gh[0][0] = redeal_rows(acc[0][0]);
                if (i == 1) gh[0][1] = redeal_rows(acc[0][1]);
                if (i == 2) gh[0][2] = redeal_rows(acc[0][2]);
                if (i == 3) {
                    a_er = __builtin_amdgcn_exp2f((gi[0][0] + gh[0][0]) * -1.4426950408889634f) + 1.0f;
                    a_ez = __builtin_amdgcn_exp2f((gi[0][1] + gh[0][1]) * -1.4426950408889634f) + 1.0f;
                }
                if (i == 4) { a_rg = __builtin_amdgcn_rcpf(a_er); a_zg = __builtin_amdgcn_rcpf(a_ez); }
                if (i == 5) a_en = __builtin_amdgcn_exp2f(__builtin_fmaf(a_rg, gh[0][2], gi[0][2]) * -2.8853900817779268f) + 1.0f;
                if (i == 6) {
                    a_ng = __builtin_amdgcn_rcpf(a_en) * 2.0f - 1.0f;
                    a_hn = __builtin_fmaf(a_zg, hreg[0] - a_ng, a_ng);
                    if (lk < MU) hnxt[lk * HLDB + 16 * wave + ln] = sv_f2bf(a_hn);
                }
                SV_SLICE_FENCE
            }
        }
        SV_PIN_ORDER
#undef SV_SLICE_FENCE
#ifdef SV_GRU_TIMING
        asm volatile("v_mov_b32 %0, %0" : "+v"(acc[NQ - 1][2][3]));
        asm volatile("s_nop 0" ::: "memory");
#endif
        SV_TS(2)
#undef SV_WLOAD
#undef SV_PIN_ORDER
        }
        // re-deal: every lane picks its (user lk, unit ln) values out of rows 0..3 of the accumulator tiles
        constexpr bool DONE0 = NQ == 2 && ALT;      // tile 0 went through its gate math between tile 1's MFMAs
#pragma unroll
        for (int q = DONE0 ? 1 : 0; q < NQ; ++q)
#pragma unroll
            for (int g = 0; g < 3; ++g) gh[q][g] = redeal_rows(acc[q][g]);
#ifdef SV_GRU_TIMING
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        SV_TS(3)
        float s_hn[NQ], s_hp[NQ], s_rg[NQ], s_zg[NQ], s_ng[NQ];
        if (DONE0) {
            s_hn[0] = a_hn; s_hp[0] = hreg[0]; s_rg[0] = a_rg; s_zg[0] = a_zg; s_ng[0] = a_ng;
            hreg[0] = a_hn;
        }
#pragma unroll
        for (int q = DONE0 ? 1 : 0; q < NQ; ++q) {
            const int j = 16 * (wave + 8 * q) + ln;
            float rg, zg, ng, hn;
            const float hp = hreg[q];
            if constexpr (SPLIT) {      // the fp32 kernel's gate math (gate_fwd<false>)
                rg = sv_sigmoid(gi[q][0] + gh[q][0]);
                zg = sv_sigmoid(gi[q][1] + gh[q][1]);
                ng = tanhf(gi[q][2] + rg * gh[q][2]);
                hn = (1.0f - zg) * ng + zg * hp;
            } else {
                const float er = __builtin_amdgcn_exp2f((gi[q][0] + gh[q][0]) * -1.4426950408889634f) + 1.0f;
                const float ez = __builtin_amdgcn_exp2f((gi[q][1] + gh[q][1]) * -1.4426950408889634f) + 1.0f;
                rg = __builtin_amdgcn_rcpf(er), zg = __builtin_amdgcn_rcpf(ez);
                const float en = __builtin_amdgcn_exp2f(__builtin_fmaf(rg, gh[q][2], gi[q][2]) * -2.8853900817779268f) + 1.0f;
                ng = __builtin_amdgcn_rcpf(en) * 2.0f - 1.0f;
                hn = __builtin_fmaf(zg, hp - ng, ng);
            }
            hreg[q] = hn;
            s_hn[q] = hn; s_hp[q] = hp; s_rg[q] = rg; s_zg[q] = zg; s_ng[q] = ng;
            if constexpr (SPLIT) {
                if (lk < MU) {
                    unsigned short p0, p1, p2;
                    split3(hn, p0, p1, p2);
                    const int o = ((t + 1) & 1) * (MB * HLDB) + lk * HLDB + j;
                    h_bf[o] = p0; h_mid[o] = p1; h_lo[o] = p2;
                }
            } else
            if (lk < MU) hnxt[lk * HLDB + j] = sv_f2bf(hn);
        }
        SV_TS(4)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(s_hn[q]), hs_b, ohs + 512 * q, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(s_hp[q]), hp_b, ohs + 512 * q, 0, 0);
            const f32x4 g0 = {s_rg[q], s_zg[q], s_ng[q], gh[q][2]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, g0), g4_b, og4 + 2048 * q, 0, 0);
        }
        SV_TS(5)
        na = na_next;
        rbase = rbase_next;
#ifdef SV_GRU_TIMING
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        SV_TS(6)
        __syncthreads();
        SV_TS(7)  // h_t published (ping-pong image: the next step writes the other buffer)
    }
#ifdef SV_GRU_TIMING
    if (blockIdx.x == 0 && wave == 0 && lane == 0)
        for (int i = 0; i < 8; ++i) g_gru_dbg[i] = ts_acc[i];
#endif
}

template <bool SPLIT>
__global__ __launch_bounds__(64 * FWD_WAVES, 1) void gru_fwd_bf16_kernel(GruArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_f[];
    unsigned short* w_lds = smem_f;                // 6 tiles x [21][64][8]   (SPLIT: the mid and lo h images)
    unsigned short* h_bf = smem_f + WLF_ELEMS;     // [2][MB][HLDB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* xs = reinterpret_cast<float*>(h_bf + 2 * MB * HLDB) + wave * XS_FLOATS;   // per-wave re-deal scratch
    for (int i = tid; i < 2 * MB * HLDB; i += 64 * FWD_WAVES) h_bf[i] = 0;
    if (SPLIT)
        for (int i = tid; i < 4 * MB * HLDB; i += 64 * FWD_WAVES) w_lds[i] = 0;
    __syncthreads();
    if (wave < 6)
        gru_fwd_bf16_body<2, SPLIT>(a, h_bf, w_lds, xs, wave, lane);
    else
        gru_fwd_bf16_body<1, SPLIT>(a, h_bf, w_lds, xs, wave, lane);
}

// Backward: same organisation as the forward (8 waves, wave w owns hidden tiles w and w+8, the second one
// LDS-resident; MU users per workgroup, masked lanes get out-of-range buffer offsets).  The dgh image only has
// MU real rows: the MFMA rows of padding users alias them (row = ln % MU), their results are discarded.
// SPLIT: as in the forward kernel -- the dgh image as bf16 hi + mid + lo parts, the W_hh^T fragments re-read and split
// every step, six products per fragment pair; the gate-gradient arithmetic is exact fp32 in both instantiations.
template <int NQ, bool SPLIT>
__device__ __forceinline__ void gru_bwd_bf16_body(const GruArgs& a, unsigned short* g_bf, unsigned short* w_lds,
                                                  float* xs, int wave, int lane) {
    const int u0 = blockIdx.x * MU;
    const int ln = lane & 15, lk = lane >> 4;
    constexpr int KS = SV_G / 32;  // 21 k-steps
    constexpr int NL = SPLIT ? 0 : NQ - 1;
    constexpr int GIMG = 2 * MU * GLDB;          // elements of one ping-pong dgh image pair
    unsigned short* const g_mid = w_lds;         // (SPLIT only)
    unsigned short* const g_lo = w_lds + GIMG;

    // B operand of dh_{t-1} = dgh W_hh: B[k = gate column c][n = hidden j'] = W_hh[c][j'] = whh_t[j'][c]
    bf16x8 W[SPLIT ? 1 : KS];
    if constexpr (!SPLIT) {
#pragma unroll
    for (int s = 0; s < KS; ++s) W[s] = load_w8(a.whh + (size_t)(16 * wave + ln) * SV_G + 32 * s + 8 * lk);
    }
    unsigned short* wl = w_lds + (size_t)wave * (WL_FRAGS * 64 * 8) + lane * 8;
    if (NL > 0) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
            *reinterpret_cast<bf16x8*>(wl + s * 512) =
                load_w8(a.whh + (size_t)(16 * (wave + 8) + ln) * SV_G + 32 * s + 8 * lk);
    }

    const sv_const_int_p nact_s = sv_scalar_ptr(a.nact), roff_s = sv_scalar_ptr(a.row_off);
    int Tw = 0;
    while (Tw < a.Tmax && nact_s[Tw] > u0) ++Tw;

    const unsigned R = (unsigned)roff_s[a.Tmax];
    const __amdgpu_buffer_rsrc_t g4_b = __builtin_amdgcn_make_buffer_rsrc(a.gates4, 0, R * (SV_HP * 16u), SV_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t hp_b = __builtin_amdgcn_make_buffer_rsrc(a.hprev, 0, R * (SV_HP * 4u), SV_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t dh_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dhs), 0, R * (SV_HP * 4u), SV_BUF_FLAGS);
    // Two output forms.  Separate: dgi = (dr, dz, dn) and dgh = (dr, dz, dn*r), [R][672] each.  Merged: the two only
    // differ in the third gate, so ONE array [R][896] = (dr, dz, dn, dn*r) carries both (its consumers read columns
    // 0..671, or 0..447 and 672..895): four stores per element instead of six, 39 MB less per step.
    const bool merged = a.merged != 0;
    const bool out16 = a.merged == 2;      // bf16 elements: every consumer rounds to bf16 anyway (RNE here as there)
    const unsigned gld = merged ? 4u * SV_HP : (unsigned)SV_G;
    const unsigned gsz = out16 ? 2u : 4u;
    const __amdgpu_buffer_rsrc_t gi_b = __builtin_amdgcn_make_buffer_rsrc(a.dgi, 0, R * (gld * gsz), SV_BUF_FLAGS);
    const __amdgpu_buffer_rsrc_t gh_b = merged ? gi_b : __builtin_amdgcn_make_buffer_rsrc(a.dgh, 0, R * (SV_G * 4u), SV_BUF_FLAGS);

    // Element roles as in the forward kernel: lane 16 u + n owns user u of unit n of each tile this wave owns.
    float dh[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) dh[q] = 0.f;

    // Saved forward values of a step are fetched one step ahead (issued right after the barrier, consumed
    // after the matmul of the step before), so their latency sits under the MFMAs.  Masked lanes read zeros
    // (d = 0 follows) and their stores are dropped.
    u32x4 sg[NQ];
    unsigned shp[NQ], sdh[NQ];
    unsigned ohs, og4, ogo;
    bool actv;
#define SV_GRU_BWD_FETCH(t_)                                                                                \
    {                                                                                                       \
        const int na_ = nact_s[t_];                                                                         \
        const unsigned row = roff_s[t_] + u0 + lk;                                                          \
        actv = lk < MU && u0 + lk < na_;                                                                    \
        ohs = actv ? (row * SV_HP + 16 * wave + ln) * 4u : SV_BUF_OOB;                                      \
        og4 = actv ? (row * SV_HP + 16 * wave + ln) * 16u : SV_BUF_OOB;                                     \
        ogo = actv ? (row * gld + 16 * wave + ln) * gsz : SV_BUF_OOB;                                       \
        _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                                    \
            sg[q] = __builtin_amdgcn_raw_buffer_load_b128(g4_b, og4 + 2048 * q, 0, 0);                      \
            shp[q] = __builtin_amdgcn_raw_buffer_load_b32(hp_b, ohs + 512 * q, 0, 0);                       \
            sdh[q] = __builtin_amdgcn_raw_buffer_load_b32(dh_b, ohs + 512 * q, 0, 0);                       \
        }                                                                                                   \
    }
    // Range [t0, t1) of the BPTT (a whole recurrence: 0, Tmax).  A later range has left the recurrent part of
    // dL/dh_{t1-1} of the users still running at step t1 in dh_carry; this one leaves dL/dh_{t0-1} there.
    const int t_hi = Tw < a.t1 ? Tw : a.t1;
    if (a.t1 < a.Tmax && lk < MU && u0 + lk < nact_s[a.t1]) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) dh[q] = a.dh_carry[(size_t)(u0 + lk) * SV_HP + 16 * (wave + 8 * q) + ln];
    }
    if (t_hi > a.t0) SV_GRU_BWD_FETCH(t_hi - 1)
    for (int t = t_hi - 1; t >= a.t0; --t) {
        unsigned short* gcur = g_bf + (t & 1) * (MU * GLDB);
        float carry[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int j = 16 * (wave + 8 * q) + ln;
            const float rg = __uint_as_float(sg[q].x), zg = __uint_as_float(sg[q].y);
            const float ng = __uint_as_float(sg[q].z), ghn = __uint_as_float(sg[q].w);
            const float d = actv ? dh[q] + __uint_as_float(sdh[q]) : 0.f;
            const float dn = d * (1.0f - zg);
            const float dzg = d * (__uint_as_float(shp[q]) - ng);
            const float dnp = dn * (1.0f - ng * ng);
            const float dzp = dzg * zg * (1.0f - zg);
            const float drp = dnp * ghn * rg * (1.0f - rg);
            const float dnr = dnp * rg;
            carry[q] = d * zg;
            if (out16) {       // wave-uniform
                __builtin_amdgcn_raw_buffer_store_b16(sv_f2bf(drp), gi_b, ogo + 256 * q, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b16(sv_f2bf(dzp), gi_b, ogo + 256 * q + SV_HP * 2, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b16(sv_f2bf(dnp), gi_b, ogo + 256 * q + 2 * SV_HP * 2, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b16(sv_f2bf(dnr), gi_b, ogo + 256 * q + 3 * SV_HP * 2, 0, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(drp), gi_b, ogo + 512 * q, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dzp), gi_b, ogo + 512 * q + SV_HP * 4, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dnp), gi_b, ogo + 512 * q + 2 * SV_HP * 4, 0, 0);
                if (merged) {      // wave-uniform
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dnr), gi_b, ogo + 512 * q + 3 * SV_HP * 4, 0, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(drp), gh_b, ogo + 512 * q, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dzp), gh_b, ogo + 512 * q + SV_HP * 4, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dnr), gh_b, ogo + 512 * q + 2 * SV_HP * 4, 0, 0);
                }
            }
            if constexpr (SPLIT) {
                if (lk < MU) {
                    const int o = (t & 1) * (MU * GLDB) + lk * GLDB + j;
                    unsigned short p0, p1, p2;
                    split3(drp, p0, p1, p2);
                    g_bf[o] = p0; g_mid[o] = p1; g_lo[o] = p2;
                    split3(dzp, p0, p1, p2);
                    g_bf[o + SV_HP] = p0; g_mid[o + SV_HP] = p1; g_lo[o + SV_HP] = p2;
                    split3(dnr, p0, p1, p2);
                    g_bf[o + 2 * SV_HP] = p0; g_mid[o + 2 * SV_HP] = p1; g_lo[o + 2 * SV_HP] = p2;
                }
            } else
            if (lk < MU) {   // MU = 4: every lane owns an image row
                gcur[lk * GLDB + j] = sv_f2bf(drp);
                gcur[lk * GLDB + SV_HP + j] = sv_f2bf(dzp);
                gcur[lk * GLDB + 2 * SV_HP + j] = sv_f2bf(dnr);
            }
        }
        __syncthreads();  // dgh_t published (ping-pong: step t-1 writes the other buffer)
        if (t > 0) {
            SV_GRU_BWD_FETCH(t - 1)
            f32x4 acc[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (SPLIT) {
                const int go = (t & 1) * (MU * GLDB) + (ln % MU) * GLDB + 8 * lk;
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(g_bf + go + 32 * s);
                    const bf16x8 am = *reinterpret_cast<const bf16x8*>(g_mid + go + 32 * s);
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(g_lo + go + 32 * s);
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        bf16x8 wh, wm, wl_;
                        load_w8_split(a.whh + (size_t)(16 * (wave + 8 * q) + ln) * SV_G + 32 * s + 8 * lk, wh, wm, wl_);
                        acc[q] = mfma_split(ah, am, al, wh, wm, wl_, acc[q]);
                    }
                }
            } else
#pragma unroll
            for (int s0 = 0; s0 < KS; s0 += 7) {
                bf16x8 A[7];
#pragma unroll
                for (int s = 0; s < 7; ++s)
                    A[s] = *reinterpret_cast<const bf16x8*>(gcur + (ln % MU) * GLDB + 32 * (s0 + s) + 8 * lk);
#pragma unroll
                for (int s = 0; s < 7; ++s) {
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s], W[s0 + s], acc[0], 0, 0, 0);
                    if (NQ == 2) {
                        const bf16x8 wv = *reinterpret_cast<const bf16x8*>(wl + (s0 + s) * 512);
                        acc[NQ - 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[s], wv, acc[NQ - 1], 0, 0, 0);
                    }
                }
            }
            // re-deal dh_{t-1} = carry + dgh W_hh: rows 0..3 of the C tile (lanes 0..15) -> lane 16 u + n
#pragma unroll
            for (int q = 0; q < NQ; ++q) dh[q] = carry[q] + redeal_rows(acc[q]);
        }
    }
    if (a.t0 > 0 && t_hi > a.t0 && lk < MU && u0 + lk < a.B) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) a.dh_carry[(size_t)(u0 + lk) * SV_HP + 16 * (wave + 8 * q) + ln] = dh[q];
    }
#undef SV_GRU_BWD_FETCH
}

template <bool SPLIT>
__global__ __launch_bounds__(64 * FWD_WAVES, 1) void gru_bwd_bf16_kernel(GruArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned short smem_b[];
    unsigned short* w_lds = smem_b;                // 6 tiles x [21][64][8]   (SPLIT: the mid and lo dgh images)
    unsigned short* g_bf = smem_b + WLF_ELEMS;     // [2][MU][GLDB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* xs = reinterpret_cast<float*>(g_bf + 2 * MU * GLDB) + wave * 128;     // per-wave re-deal scratch
    if (wave < 6)
        gru_bwd_bf16_body<2, SPLIT>(a, g_bf, w_lds, xs, wave, lane);
    else
        gru_bwd_bf16_body<1, SPLIT>(a, g_bf, w_lds, xs, wave, lane);
}

}  // namespace

static int check_gru(const GruArgs& a, bool bwd, bool saved_optional = false) {
    SV_CHECK_ARG(a.whh && a.row_off && a.nact, "svae_gru: null pointer");
    SV_CHECK_ARG(saved_optional ? ((a.hprev == nullptr) == (a.gates4 == nullptr)) : (a.hprev && a.gates4),
                 "svae_gru: hprev / gates4 (only the bf16 and split forward kernels may leave BOTH out)");
    SV_CHECK_ARG(a.B > 0 && a.Tmax > 0, "svae_gru: empty batch (B=%d Tmax=%d)", a.B, a.Tmax);
    if (!bwd) SV_CHECK_ARG(a.gi && a.bhh && a.hs, "svae_gru_fwd: null pointer");
    if (bwd) SV_CHECK_ARG(a.dhs && a.dgi && a.dgh, "svae_gru_bwd: null pointer");
    return 0;
}

// Reference interface replaced: nn.GRU forward inside Model.forward (nb:759).
extern "C" int svae_gru_fwd_range(const float* gi, const float* whh, const float* bhh, const int* row_off,
                                  const int* nact, int B, int Tmax, int t0, int t1, float* hs, float* hprev,
                                  float* gates4, int prec, void* stream);
extern "C" int svae_gru_fwd(const float* gi, const float* whh, const float* bhh, const int* row_off,
                            const int* nact, int B, int Tmax, float* hs, float* hprev, float* gates4, int prec,
                            void* stream) {
    return svae_gru_fwd_range(gi, whh, bhh, row_off, nact, B, Tmax, 0, Tmax, hs, hprev, gates4, prec, stream);
}

// The same recurrence over timesteps [t0, t1) only.  t0 > 0 continues a recurrence whose steps < t0 have been
// run (hs holds h_{t0-1}); bf16 kernel only.  No counterpart in the reference (one cuDNN call there): the split
// lets the host overlap the second range with the decoder side of the first range's rows.
extern "C" int svae_gru_fwd_range(const float* gi, const float* whh, const float* bhh, const int* row_off,
                                  const int* nact, int B, int Tmax, int t0, int t1, float* hs, float* hprev,
                                  float* gates4, int prec, void* stream) {
    GruArgs a{gi, whh, bhh, row_off, nact, hs, hprev, gates4, nullptr, nullptr, nullptr, B, Tmax, 0, t0, t1};
    if (int e = check_gru(a, false, prec != 0)) return e;
    SV_CHECK_ARG(prec == 0 || prec == 1 || prec == 2, "svae_gru_fwd: bad prec %d", prec);
    SV_CHECK_ARG(t0 >= 0 && t0 < t1 && t1 <= Tmax, "svae_gru_fwd_range: bad range [%d, %d) of %d", t0, t1, Tmax);
    SV_CHECK_ARG(prec != 0 || (t0 == 0 && t1 == Tmax), "svae_gru_fwd_range: the fp32 kernel runs whole sequences only");
    dim3 grid((B + MB - 1) / MB);
    if (prec == 0) {
        hipLaunchKernelGGL(gru_fwd_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    } else {
        grid = dim3((B + MU - 1) / MU);
        SV_CHECK_ARG((long long)B * Tmax < (1ll << 19), "svae_gru_fwd: B*Tmax = %lld rows exceed the 32-bit addressing "
                     "of the bf16 kernel (2^19 rows); split the batch", (long long)B * Tmax);
        const int lds = (WLF_ELEMS + 2 * MB * HLDB) * sizeof(unsigned short) + FWD_WAVES * XS_FLOATS * 4;   // 156 KiB of the CU's 160
        // prec 2: the split instantiation of the same kernel (fp32-class; see gru_fwd_bf16_body)
        const void* fn = prec == 2 ? (const void*)gru_fwd_bf16_kernel<true> : (const void*)gru_fwd_bf16_kernel<false>;
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return svae_fail((int)e, "svae_gru_fwd: LDS attribute: %s", hipGetErrorString(e));
        if (prec == 2)
            hipLaunchKernelGGL(gru_fwd_bf16_kernel<true>, dim3((B + MU - 1) / MU), dim3(64 * FWD_WAVES), lds, (hipStream_t)stream, a);
        else
            hipLaunchKernelGGL(gru_fwd_bf16_kernel<false>, dim3((B + MU - 1) / MU), dim3(64 * FWD_WAVES), lds, (hipStream_t)stream, a);
    }
    SV_LAUNCH_CHECK();
    return 0;
}

extern "C" int svae_gru_bwd(const float* dhs, const float* hprev, const float* gates4, const float* whh_t,
                            const int* row_off, const int* nact, int B, int Tmax, float* dgi, float* dgh,
                            int prec, void* stream);
// bf16 kernel with the merged gradient array as bf16 elements: dg4 [R][896] bf16 = (dr, dz, dn, dn*r).
extern "C" int svae_gru_bwd_bf16out(const float* dhs, const float* hprev, const float* gates4, const float* whh_t,
                                    const int* row_off, const int* nact, int B, int Tmax, void* dg4, void* stream) {
    return svae_gru_bwd(dhs, hprev, gates4, whh_t, row_off, nact, B, Tmax, (float*)dg4, nullptr, 3, stream);
}

static int gru_bwd_launch(const float* dhs, const float* hprev, const float* gates4, const float* whh_t,
                          const int* row_off, const int* nact, int B, int Tmax, float* dgi, float* dgh, int prec,
                          int t0, int t1, float* dh_carry, void* stream);

// Reference interface replaced: autograd through nn.GRU at loss.backward() (nb:1205).
// whh_t is W_hh transposed, [224][672].
extern "C" int svae_gru_bwd(const float* dhs, const float* hprev, const float* gates4, const float* whh_t,
                            const int* row_off, const int* nact, int B, int Tmax, float* dgi, float* dgh,
                            int prec, void* stream) {
    return gru_bwd_launch(dhs, hprev, gates4, whh_t, row_off, nact, B, Tmax, dgi, dgh, prec, 0, Tmax, nullptr, stream);
}

// The BPTT over timesteps [t0, t1) only (it runs from t1-1 down to t0), bf16 kernel, merged bf16 gradient array.
// t1 < Tmax continues a BPTT whose steps >= t1 have been run with the same dh_carry [B][224]; t0 > 0 leaves the
// recurrent part of dL/dh_{t0-1} there for the next call.  The rows of timesteps >= t0 of dg4 are final when the call
// has run.  No counterpart in the reference (one autograd call there): the split lets the host start the weight and
// embedding gradients of the late timesteps' rows while the BPTT of the early ones is still running.
extern "C" int svae_gru_bwd_range_bf16out(const float* dhs, const float* hprev, const float* gates4, const float* whh_t,
                                          const int* row_off, const int* nact, int B, int Tmax, int t0, int t1,
                                          void* dg4, float* dh_carry, void* stream) {
    SV_CHECK_ARG(t0 >= 0 && t0 < t1 && t1 <= Tmax, "svae_gru_bwd_range: bad range [%d, %d) of %d", t0, t1, Tmax);
    SV_CHECK_ARG(dh_carry || (t0 == 0 && t1 == Tmax), "svae_gru_bwd_range: a partial range needs dh_carry");
    return gru_bwd_launch(dhs, hprev, gates4, whh_t, row_off, nact, B, Tmax, (float*)dg4, nullptr, 3, t0, t1, dh_carry,
                          stream);
}

static int gru_bwd_launch(const float* dhs, const float* hprev, const float* gates4, const float* whh_t,
                          const int* row_off, const int* nact, int B, int Tmax, float* dgi, float* dgh, int prec,
                          int t0, int t1, float* dh_carry, void* stream) {
    // dgh == NULL (bf16 kernel only): merged output, dgi is [R][896] = (dr, dz, dn, dn*r)
    const int out16 = prec == 3 ? 1 : 0;      // internal: svae_gru_bwd_bf16out
    if (out16) prec = 1;
    const int merged = dgh == nullptr ? 1 + out16 : 0;
    GruArgs a{nullptr, whh_t, nullptr, row_off, nact, nullptr, const_cast<float*>(hprev),
              const_cast<float*>(gates4), dhs, dgi, merged ? dgi : dgh, B, Tmax, merged, t0, t1, dh_carry};
    if (int e = check_gru(a, true)) return e;
    SV_CHECK_ARG(prec != 0 || (t0 == 0 && t1 == Tmax), "svae_gru_bwd: time ranges need the bf16 kernel");
    SV_CHECK_ARG(prec == 0 || prec == 1 || prec == 2, "svae_gru_bwd: bad prec %d", prec);
    SV_CHECK_ARG(!merged || prec != 0, "svae_gru_bwd: the merged gradient array needs the bf16 kernel template");
    SV_CHECK_ARG(prec != 2 || merged != 2, "svae_gru_bwd: the split instantiation writes fp32 gradients");
    dim3 grid((B + MB - 1) / MB);
    if (prec == 0) {
        hipLaunchKernelGGL(gru_bwd_f32_kernel, grid, dim3(256), MB * GLD * sizeof(float), (hipStream_t)stream, a);
    } else {
        grid = dim3((B + MU - 1) / MU);
        SV_CHECK_ARG((long long)B * Tmax < (1ll << 19), "svae_gru_bwd: B*Tmax = %lld rows exceed the 32-bit addressing "
                     "of the bf16 kernel (2^19 rows); split the batch", (long long)B * Tmax);
        const int lds = (WLF_ELEMS + 2 * MU * GLDB) * sizeof(unsigned short) + FWD_WAVES * 128 * 4;
        const void* fn = prec == 2 ? (const void*)gru_bwd_bf16_kernel<true> : (const void*)gru_bwd_bf16_kernel<false>;
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return svae_fail((int)e, "svae_gru_bwd: LDS attribute: %s", hipGetErrorString(e));
        if (prec == 2)
            hipLaunchKernelGGL(gru_bwd_bf16_kernel<true>, grid, dim3(64 * FWD_WAVES), lds, (hipStream_t)stream, a);
        else
            hipLaunchKernelGGL(gru_bwd_bf16_kernel<false>, grid, dim3(64 * FWD_WAVES), lds, (hipStream_t)stream, a);
    }
    SV_LAUNCH_CHECK();
    return 0;
}

#ifdef SV_GRU_TIMING
extern "C" int svae_debug_gru(unsigned long long* out8) {
    return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_gru_dbg), 8 * sizeof(unsigned long long));
}
#endif
