// Probe of ds_read_b64_tr_b16 (gfx950): which element lands where.  Build: hipcc --offload-arch=gfx950 tr16_probe.hip -o tr16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 160];
    for (int i = threadIdx.x; i < 64 * 160; i += 64) lds[i] = (short)((i / 160) * 256 + (i % 160));   // row * 256 + col
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    // group g reads the block rows 4g .. 4g+3, columns 32 .. 47: lane 4q + p supplies row q, columns 4p .. 4p+3
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + (4 * g + q) * 160 + 32 + 4 * p));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
    short* d; short h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, i = lane & 15;
        for (int j = 0; j < 4; ++j) {
            const int want = (4 * g + j) * 256 + 32 + i;      // element j of lane i of the group = row j, column i of the block
            if (h[lane * 4 + j] != want) { if (bad < 8) printf("lane %d elem %d: got row %d col %d, want row %d col %d\n", lane, j, h[lane*4+j] >> 8, h[lane*4+j] & 255, want >> 8, want & 255); ++bad; }
        }
    }
    printf("tr16 probe: %d mismatches\n", bad);
    return bad != 0;
}
