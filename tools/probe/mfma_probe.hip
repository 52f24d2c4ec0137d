// Probe: how fast does one CU run v_mfma_f32_32x32x16_bf16 from two waves per SIMD, in the dependency pattern of
// the projection sweep (a chain of 10 on one accumulator, then 10 over five accumulators in pairs), with and
// without a workgroup barrier every 40?   hipcc -O3 --offload-arch=gfx950 mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void glds16(const char* g, char* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// MODE 4: sweep pattern + barrier + the LDS-DMA traffic of the sweep (20 KiB per iteration and workgroup, 5 tiles in
// flight, counted vmcnt wait in front of the barrier).  MODE 5: the same without the barrier.
template <int MODE, int WAVES, int PIECES = 3, int NOWAIT = 0, int DEPTH = 5>
__global__ __launch_bounds__(64 * WAVES, 1) void kd(unsigned long long* out, float* sink, const char* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int npc = PIECES == 3 ? (wave < 4 ? 3 : 2) : PIECES == 5 ? (wave >= 4 ? 3 : 2) : PIECES == 6 ? ((wave & 1) ? 2 : 3) : (PIECES == 4 ? 3 : PIECES);
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
    a[0] = (short)threadIdx.x;
    f32x16 c1, Z[5];
    for (int r = 0; r < 16; ++r) { c1[r] = 0.f; for (int n = 0; n < 5; ++n) Z[n][r] = 0.f; }
    const char* g = src + (size_t)(blockIdx.x % 16) * 20480 * 8 + threadIdx.x * 16;
    for (int j = 0; j < DEPTH; ++j)
        for (int q = 0; q < npc; ++q) glds16(g + j * 20480 + q * 8192, smem + (j % 7) * 20480 + wave * 1024 + q * 8192);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int sj = 5;
    for (int it = 0; it < iters; ++it) {
        const char* gs = g + ((it + 5) & 255) * 20480;
        char* ls = smem + sj * 20480 + wave * 1024;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int s = 0; s < 10; ++s) {
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                if (s == 2 && sub < npc) glds16(gs + sub * 8192, ls + sub * 8192);
                if (s == 6 && sub == 0 && npc == 3) glds16(gs + 2 * 8192, ls + 2 * 8192);
            }
#pragma unroll
            for (int i = 0; i < 10; ++i) Z[i >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, Z[i >> 1], 0, 0, 0);
        }
        if (NOWAIT) {}
        else {      // all but the youngest DEPTH - 2 tiles' pieces of this wave have landed
            const int n = (DEPTH - 2) * npc;
            if (n == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (n == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
            else if (n == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (MODE == 4) __builtin_amdgcn_s_barrier();
        sj = sj == 6 ? 0 : sj + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
    for (int r = 0; r < 16; ++r) { acc += c1[r]; for (int n = 0; n < 5; ++n) acc += Z[n][r]; }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k(unsigned long long* out, float* sink, int iters) {
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
    a[0] = (short)threadIdx.x;
    f32x16 c1, Z[5];
    for (int r = 0; r < 16; ++r) { c1[r] = 0.f; for (int n = 0; n < 5; ++n) Z[n][r] = 0.f; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            if (MODE == 0 || MODE == 1) {
#pragma unroll
                for (int s = 0; s < 10; ++s) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 10; ++i) Z[i >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, Z[i >> 1], 0, 0, 0);
            } else {   // independent: round robin over six accumulators
#pragma unroll
                for (int s = 0; s < 20; ++s) {
                    if (s % 6 == 5) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                    else Z[s % 6] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, Z[s % 6], 0, 0, 0);
                }
            }
        }
        if (MODE == 0 || MODE == 2) __builtin_amdgcn_s_barrier();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
    for (int r = 0; r < 16; ++r) { acc += c1[r]; for (int n = 0; n < 5; ++n) acc += Z[n][r]; }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
}


// One wave per SIMD: 4 waves, each with TWO row blocks (two chains of 10, then 20 over ten accumulators) = the same 80
// MFMAs per SIMD and tile, the same 20 KiB of LDS-DMA per tile (5 pieces per wave), one barrier per tile over 4 waves.
template <int BARRIER, int DMA>
__global__ __launch_bounds__(256, 1) void kd4(unsigned long long* out, float* sink, const char* src, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
    a[0] = (short)threadIdx.x;
    f32x16 c1a, c1b, Z[10];
    for (int r = 0; r < 16; ++r) { c1a[r] = 0.f; c1b[r] = 0.f; for (int n = 0; n < 10; ++n) Z[n][r] = 0.f; }
    const char* g = src + (size_t)(blockIdx.x % 16) * 20480 * 8 + (threadIdx.x & 63) * 16 + wave * 5 * 1024;
    if (DMA)
        for (int j = 0; j < 5; ++j)
            for (int q = 0; q < 5; ++q) glds16(g + j * 20480 + q * 1024, smem + (j % 7) * 20480 + wave * 5120 + q * 1024);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    int sj = 5;
    for (int it = 0; it < iters; ++it) {
        const char* gs = g + ((it + 5) & 255) * 20480;
        char* ls = smem + sj * 20480 + wave * 5120;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
            for (int s = 0; s < 10; ++s) {
                c1a = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1a, 0, 0, 0);
                c1b = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1b, 0, 0, 0);
                if (DMA && s == 2) glds16(gs + (2 * sub) * 1024, ls + (2 * sub) * 1024);
                if (DMA && s == 6) glds16(gs + (2 * sub + 1) * 1024, ls + (2 * sub + 1) * 1024);
            }
#pragma unroll
            for (int i = 0; i < 20; ++i) {
                Z[i >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, Z[i >> 1], 0, 0, 0);
                if (DMA && sub == 0 && i == 10) glds16(gs + 4 * 1024, ls + 4 * 1024);
            }
        }
        if (DMA) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        if (BARRIER) __builtin_amdgcn_s_barrier();
        sj = sj == 6 ? 0 : sj + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0.f;
    for (int r = 0; r < 16; ++r) { acc += c1a[r] + c1b[r]; for (int n = 0; n < 10; ++n) acc += Z[n][r]; }
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int BARRIER, int DMA>
void rund4(const char* name, int blocks) {
    unsigned long long* out; float* sink; char* src;
    hipMalloc(&out, blocks * 8 * 8); hipMalloc(&sink, blocks * 8 * 64 * 4); hipMalloc(&src, 16 * 20480 * 8 + 300 * 20480);
    hipMemset(src, 0, 16 * 20480 * 8 + 300 * 20480);
    hipFuncSetAttribute((const void*)kd4<BARRIER, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 7 * 20480);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kd4<BARRIER, DMA>), dim3(blocks), dim3(256), 7 * 20480, 0, out, sink, src, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((kd4<BARRIER, DMA>), dim3(blocks), dim3(256), 7 * 20480, 0, out, sink, src, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[16];
    hipMemcpy(h, out, 4 * 8, hipMemcpyDeviceToHost);
    printf("%-34s waves 4 blocks %4d: cycles/iter wave0 %.0f wave3 %.0f, wall %.3f ms -> %.1f ns/iter\n", name, blocks,
           (double)h[0] / iters, (double)h[3] / iters, ms, ms * 1e6 / iters);
}

template <int MODE, int WAVES>
void run(const char* name, int blocks) {
    unsigned long long* out; float* sink;
    hipMalloc(&out, blocks * WAVES * 8); hipMalloc(&sink, blocks * WAVES * 64 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, out, sink, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, out, sink, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[16];
    hipMemcpy(h, out, WAVES * 8, hipMemcpyDeviceToHost);
    const double per_simd = 40.0 * (WAVES / 4);           // MFMAs per SIMD per iteration
    printf("%-34s waves %d blocks %4d: cycles/iter wave0 %.0f (= %.1f per MFMA of the SIMD), wall %.3f ms -> %.1f ns/iter\n",
           name, WAVES, blocks, (double)h[0] / iters, (double)h[0] / iters / per_simd, ms, ms * 1e6 / iters);
    hipFree(out); hipFree(sink);
}

template <int MODE, int PIECES = 3, int NOWAIT = 0, int DEPTH = 5>
void rund(const char* name, int blocks) {
    unsigned long long* out; float* sink; char* src;
    hipMalloc(&out, blocks * 8 * 8); hipMalloc(&sink, blocks * 8 * 64 * 4); hipMalloc(&src, 16 * 20480 * 8 + 300 * 20480);
    hipMemset(src, 0, 16 * 20480 * 8 + 300 * 20480);
    hipFuncSetAttribute((const void*)kd<MODE, 8, PIECES, NOWAIT, DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 7 * 20480);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((kd<MODE, 8, PIECES, NOWAIT, DEPTH>), dim3(blocks), dim3(512), 7 * 20480, 0, out, sink, src, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((kd<MODE, 8, PIECES, NOWAIT, DEPTH>), dim3(blocks), dim3(512), 7 * 20480, 0, out, sink, src, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[16];
    hipMemcpy(h, out, 8 * 8, hipMemcpyDeviceToHost);
    printf("%-34s waves 8 blocks %4d: cycles/iter wave0 %.0f wave4 %.0f, wall %.3f ms -> %.1f ns/iter\n", name, blocks,
           (double)h[0] / iters, (double)h[4] / iters, ms, ms * 1e6 / iters);
}

int main() {
    for (int blocks : {1, 256}) {
        rund<4>("sweep + LDS-DMA 20K + barrier", blocks);
        rund<4, 2>("sweep + LDS-DMA 16K + barrier", blocks);
        rund<4, 1>("sweep + LDS-DMA 8K + barrier", blocks);
        rund<4, 0>("sweep + LDS-DMA 0K + barrier", blocks);
        rund<4, 4>("sweep + LDS-DMA 24K (3 each) + barrier", blocks);
        rund<4, 5>("20K, waves 4-7 three pieces", blocks);
        rund<4, 6>("20K, even waves three pieces", blocks);
        rund<4, 3, 0, 6>("sweep + LDS-DMA 20K depth 6 + barrier", blocks);
        rund<4, 3, 0, 4>("sweep + LDS-DMA 20K depth 4 + barrier", blocks);
        rund<4, 2, 0, 4>("sweep + LDS-DMA 16K depth 4 + barrier", blocks);
        rund<4, 5, 0, 3>("20K, waves 4-7 three, depth 3", blocks);
        rund<4, 5, 0, 4>("20K, waves 4-7 three, depth 4", blocks);
        rund<4, 5, 0, 6>("20K, waves 4-7 three, depth 6", blocks);
        rund<4, 4, 0, 3>("24K (3 each), depth 3", blocks);
        rund<4, 4, 0, 4>("24K (3 each), depth 4", blocks);
        rund<4, 2, 0, 3>("16K, depth 3", blocks);
        rund<4, 5, 1>("20K, waves 4-7 three, nowait", blocks);
        rund<4, 3, 1>("sweep + LDS-DMA 20K nowait + barrier", blocks);
        rund<4, 1, 1>("sweep + LDS-DMA 8K nowait + barrier", blocks);
    }
    for (int blocks : {1, 256}) {
        rund4<1, 1>("4 waves: 20K + barrier", blocks);
        rund4<0, 1>("4 waves: 20K, no barrier", blocks);
        rund4<1, 0>("4 waves: no DMA, barrier", blocks);
        rund4<0, 0>("4 waves: no DMA, no barrier", blocks);
    }
    for (int blocks : {1, 256}) {
        run<0, 8>("sweep pattern + barrier", blocks);
        run<1, 8>("sweep pattern, no barrier", blocks);
        run<2, 8>("independent accs + barrier", blocks);
        run<3, 8>("independent accs, no barrier", blocks);
        run<1, 4>("sweep pattern, no barrier", blocks);
        run<3, 4>("independent accs, no barrier", blocks);
    }
    return 0;
}
