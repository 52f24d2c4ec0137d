// Shared definitions for the SVAE gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <type_traits>

// ---- padded model geometry (reference hyper_params nb:82-85: E=256, H=200, Hd=150, L=64) ----
// Every K dimension is padded to a multiple of 32 so bf16 MFMA (K=32 / 16 per step) and 16-byte
// vector loads never need tail code.  Pad weights are zero and stay zero under Adam (g = p = 0).
#define SV_E 256       // item_embed_size
#define SV_H 200       // rnn_size
#define SV_HP 224      // rnn_size padded
#define SV_G (3 * SV_HP)  // 672: gate columns, gate g / unit j at column g*224 + j
#define SV_D 150       // hidden_size
#define SV_DP 160      // hidden_size padded
#define SV_L 64        // latent_size
#define SV_L2 128      // 2 * latent_size

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16v2 __attribute__((ext_vector_type(2)));

extern thread_local char g_svae_err[512];
int svae_fail(int code, const char* fmt, ...);
// ops.hip: per-device address of the static scratch of svae_masked_topk_static (current device); out may be NULL
int svae_scratch_setup(float** out);

#define SV_CHECK_ARG(cond, ...)                         \
    do {                                                \
        if (!(cond)) return svae_fail(-1, __VA_ARGS__); \
    } while (0)

#define SV_LAUNCH_CHECK()                                                            \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) return svae_fail((int)e__, "%s: launch failed: %s", \
                                                 __func__, hipGetErrorString(e__)); \
    } while (0)

__device__ __forceinline__ float sv_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// fast forms for the bf16 kernels: v_exp_f32 / v_rcp_f32 (about 1 ulp each)
__device__ __forceinline__ float sv_sigmoid_fast(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float sv_tanh_fast(float x) {
    return 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * x)) - 1.0f;
}

// round-to-nearest-even f32 -> bf16 (plain cast form keeps NaN a NaN on gfx950)
__device__ __forceinline__ unsigned short sv_f2bf(float x) {
    __hip_bfloat16 b = __float2bfloat16(x);
    return *reinterpret_cast<unsigned short*>(&b);
}
__device__ __forceinline__ float sv_bf2f(unsigned short u) {
    return __uint_as_float(((unsigned)u) << 16);
}

__device__ __forceinline__ float sv_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float sv_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
