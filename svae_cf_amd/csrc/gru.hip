// Persistent-over-timesteps GRU recurrence for gfx950 (reference: nn.GRU call nb:759, def nb:724-727;
// gate order r,z,n; h0 = 0).
//
// Layout: a batch is TIME-MAJOR PACKED.  Users are sorted by length (descending); at timestep t the
// first nact[t] users are active and their rows are row_off[t] .. row_off[t]+nact[t]-1.  One workgroup
// owns 16 consecutive users for ALL timesteps (users are independent, so there is no inter-workgroup
// exchange); the hidden state lives in registers (fp32, owner lane) and in LDS (matmul operand).
//
// The input-side gates gi = emb W_ih^T + b_ih for all rows are computed beforehand by one big GEMM.
// This file: the recurrent part, gh = h_{t-1} W_hh^T + b_hh, the fused gate math, and the reverse-time
// BPTT that produces dgi / dgh for the weight-gradient GEMMs.
//
//   fp32 mode: v_mfma_f32_16x16x4_f32, W_hh streamed from L2 every step (480 KB fp32 does not fit a CU)
//   bf16 mode: v_mfma_f32_16x16x32_bf16, W_hh fragments resident in VGPRs for the whole sequence
#include "common.h"

namespace {

constexpr int MB = 16;             // users per workgroup (MFMA M)
constexpr int NJT = SV_HP / 16;    // 14 hidden-unit tiles of 16
constexpr int HLD = SV_HP + 4;     // LDS row stride (floats) of the h image
constexpr int GLD = SV_G + 4;      // LDS row stride (floats) of the dgh image

struct GruArgs {
    const float* gi;      // [R][672]   (fwd)   input-side pre-activations incl. b_ih
    const float* whh;     // fwd: W_hh [672][224];  bwd: W_hh^T [224][672]
    const float* bhh;     // [672]
    const int* row_off;   // [Tmax+1]
    const int* nact;      // [Tmax]
    float* hs;            // [R][224]  hidden states (fwd out / bwd in)
    float* gates;         // [R][5][224]  r, z, n, gh_n, h_{t-1}  (fwd out / bwd in)
    const float* dhs;     // [R][224]  (bwd in) dL/dh_t from the encoder side
    float* dgi;           // [R][672]  (bwd out)
    float* dgh;           // [R][672]  (bwd out)
    int B, Tmax;
};

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

        // gate math: lane owns hidden unit j = 16*jt + ln for users m = 4*lk + r
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
                        const float* gp = a.gi + row * SV_G;
                        const float ghn = acc[q][2][r] + bn;
                        const float rg = sv_sigmoid(gp[j] + acc[q][0][r] + br);
                        const float zg = sv_sigmoid(gp[SV_HP + j] + acc[q][1][r] + bz);
                        const float ng = tanhf(gp[2 * SV_HP + j] + rg * ghn);
                        const float hp = hreg[q][r];
                        const float hn = (1.0f - zg) * ng + zg * hp;
                        hreg[q][r] = hn;
                        a.hs[row * SV_HP + j] = hn;
                        float* gs = a.gates + row * (5 * SV_HP);
                        gs[4 * SV_HP + j] = hp;
                        gs[j] = rg;
                        gs[SV_HP + j] = zg;
                        gs[2 * SV_HP + j] = ng;
                        gs[3 * SV_HP + j] = ghn;
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
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int jt = wave + 4 * q;
            if (jt < NJT) {
                const int j = 16 * jt + ln;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 4 * lk + r;
                    float drp = 0.f, dzp = 0.f, dnp = 0.f, dnr = 0.f;
                    carry[q][r] = 0.f;
                    if (u0 + m < na) {
                        const size_t row = (size_t)(rbase + m);
                        const float* gs = a.gates + row * (5 * SV_HP);
                        const float rg = gs[j], zg = gs[SV_HP + j], ng = gs[2 * SV_HP + j], ghn = gs[3 * SV_HP + j];
                        const float hprev = gs[4 * SV_HP + j];
                        const float d = dh[q][r] + a.dhs[row * SV_HP + j];
                        const float dn = d * (1.0f - zg);
                        const float dzg = d * (hprev - ng);
                        dnp = dn * (1.0f - ng * ng);
                        const float drg = dnp * ghn;
                        dzp = dzg * zg * (1.0f - zg);
                        drp = drg * rg * (1.0f - rg);
                        dnr = dnp * rg;
                        carry[q][r] = d * zg;
                        float* o1 = a.dgi + row * SV_G;
                        float* o2 = a.dgh + row * SV_G;
                        o1[j] = drp; o1[SV_HP + j] = dzp; o1[2 * SV_HP + j] = dnp;
                        o2[j] = drp; o2[SV_HP + j] = dzp; o2[2 * SV_HP + j] = dnr;
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

}  // namespace

static int check_gru(const GruArgs& a, bool bwd) {
    SV_CHECK_ARG(a.whh && a.row_off && a.nact && a.hs && a.gates, "svae_gru: null pointer");
    SV_CHECK_ARG(a.B > 0 && a.Tmax > 0, "svae_gru: empty batch (B=%d Tmax=%d)", a.B, a.Tmax);
    if (!bwd) SV_CHECK_ARG(a.gi && a.bhh, "svae_gru_fwd: null pointer");
    if (bwd) SV_CHECK_ARG(a.dhs && a.dgi && a.dgh, "svae_gru_bwd: null pointer");
    return 0;
}

// Reference interface replaced: nn.GRU forward inside Model.forward (nb:759).
extern "C" int svae_gru_fwd(const float* gi, const float* whh, const float* bhh, const int* row_off,
                            const int* nact, int B, int Tmax, float* hs, float* gates, int prec, void* stream) {
    GruArgs a{gi, whh, bhh, row_off, nact, hs, gates, nullptr, nullptr, nullptr, B, Tmax};
    if (int e = check_gru(a, false)) return e;
    SV_CHECK_ARG(prec == 0, "svae_gru_fwd: prec %d not built", prec);
    dim3 grid((B + MB - 1) / MB);
    hipLaunchKernelGGL(gru_fwd_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    SV_LAUNCH_CHECK();
    return 0;
}

// Reference interface replaced: autograd through nn.GRU at loss.backward() (nb:1205).
// whh_t is W_hh transposed, [224][672].
extern "C" int svae_gru_bwd(const float* dhs, const float* hs, const float* gates, const float* whh_t,
                            const int* row_off, const int* nact, int B, int Tmax, float* dgi, float* dgh,
                            int prec, void* stream) {
    GruArgs a{nullptr, whh_t, nullptr, row_off, nact, const_cast<float*>(hs), const_cast<float*>(gates),
              dhs, dgi, dgh, B, Tmax};
    if (int e = check_gru(a, true)) return e;
    SV_CHECK_ARG(prec == 0, "svae_gru_bwd: prec %d not built", prec);
    dim3 grid((B + MB - 1) / MB);
    hipLaunchKernelGGL(gru_bwd_f32_kernel, grid, dim3(256), MB * GLD * sizeof(float), (hipStream_t)stream, a);
    SV_LAUNCH_CHECK();
    return 0;
}
