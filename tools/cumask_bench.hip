// Development microbench: does hipExtStreamCreateWithCUMask confine a kernel?  A compute-bound kernel (one block per CU-slot)
// timed on an unrestricted stream and on streams with every 2nd / 4th / 8th CU bit set, plus the set of (XCC, CU) ids seen.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#include <set>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
__global__ __launch_bounds__(256) void k_spin(float* out, unsigned* ids, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) a = fmaf(a, b, 1e-6f);
    if (threadIdx.x == 0) {
        unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID
        unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); // XCC_ID
        ids[blockIdx.x] = (xcc & 0xf) << 16 | (hw & 0xffff);
    }
    out[blockIdx.x * 256 + threadIdx.x] = a;
}
int main() {
    const int blocks = 2048, iters = 20000;
    float* out; unsigned* ids; CK(hipMalloc(&out, blocks * 256 * 4)); CK(hipMalloc(&ids, blocks * 4));
    std::vector<unsigned> h(blocks);
    for (int stride : {1, 2, 4, 8}) {
        hipStream_t st;
        if (stride == 1) CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        else {
            uint32_t mask[8] = {0};
            for (int cu = 0; cu < 256; cu += stride) mask[cu >> 5] |= 1u << (cu & 31);
            CK(hipExtStreamCreateWithCUMask(&st, 8, mask));
        }
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, st, out, ids, iters);
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(a, st));
        hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(256), 0, st, out, ids, iters);
        CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        CK(hipMemcpy(h.data(), ids, blocks * 4, hipMemcpyDeviceToHost));
        std::set<unsigned> cus; std::set<unsigned> xccs;
        for (unsigned v : h) { cus.insert(((v >> 16) << 16) | ((v >> 8) & 0xff) | ((v >> 13) & 0x7) << 12); xccs.insert(v >> 16); }
        printf("stride %d: %.3f ms, distinct (xcc, se/cu) ids %zu, xccs %zu\n", stride, ms, cus.size(), xccs.size());
        CK(hipStreamDestroy(st));
    }
    return 0;
}
