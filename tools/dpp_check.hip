// dpp_check: the DPP / permlane-swap xor-butterfly of vox_device.h against the __shfl_xor one, bit for bit, on random data.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define VOX_DPP_BUTTERFLY 1
#include "../vox_serve_amd/csrc/vox_device.h"

template <int WIDTH>
__device__ float butterfly_shfl(float s) {
#pragma unroll
    for (int off = WIDTH / 2; off >= 1; off >>= 1) s = s + __shfl_xor(s, off, 64);
    return s;
}
__global__ void k(const float* x, float* a, float* b, int width) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    const float s = x[i];
    float ra, rb;
    if (width == 64) { ra = butterfly<64>(s); rb = butterfly_shfl<64>(s); }
    else if (width == 16) { ra = butterfly<16>(s); rb = butterfly_shfl<16>(s); }
    else { ra = butterfly<8>(s); rb = butterfly_shfl<8>(s); }
    a[i] = ra; b[i] = rb;
}
int main() {
    const int n = 64 * 4096;
    float *h = (float*)malloc(n * 4), *ha = (float*)malloc(n * 4), *hb = (float*)malloc(n * 4);
    srand(1);
    for (int i = 0; i < n; ++i) h[i] = ((rand() % 2000001) - 1000000) * 1e-3f * (1 + (rand() % 1000) * 1e-3f);
    float *x, *a, *b;
    hipMalloc(&x, n * 4); hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
    hipMemcpy(x, h, n * 4, hipMemcpyHostToDevice);
    int bad = 0;
    for (int w : {64, 16, 8}) {
        hipLaunchKernelGGL(k, dim3(n / 64), dim3(64), 0, 0, x, a, b, w);
        hipMemcpy(ha, a, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb, b, n * 4, hipMemcpyDeviceToHost);
        int m = 0;
        for (int i = 0; i < n; ++i) m += memcmp(ha + i, hb + i, 4) != 0;
        printf("width %d: %d mismatches of %d\n", w, m, n);
        bad += m;
    }
    return bad != 0;
}
