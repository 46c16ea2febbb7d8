// Development microbench: does the MFMA D-layout store pattern of the conv epilogues (16 lanes x 4 B = 64-byte segments, 4 rows per
// instruction) cost HBM bandwidth against fully coalesced 16-byte-per-lane stores?  Tile 128 x 96 fp32 per block as in k_conv_gemm<32,4,3>.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/store_pattern_probe tools/store_pattern_probe.hip && tools/bin/store_pattern_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int N = 96, BM = 128;

// out = res + f(tile): D-layout pattern (mode 0) or row-major float4 pattern (mode 1); two outputs when TWO
template <int MODE, bool RES, bool TWO>
__global__ __launch_bounds__(256) void k_pat(const float* res, float* out, float* out2, int M) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t m0 = (size_t)blockIdx.x * BM;
    if (MODE == 0) {
        const int wm = (wave >> 1) * 64, wn = (wave & 1) * 48, l15 = lane & 15, rg = (lane >> 4) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float rs[3][4];
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) rs[j][r] = RES ? res[(m0 + wm + i * 16 + rg + r) * N + wn + j * 16 + l15] : 1.0f;
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const size_t o = (m0 + wm + i * 16 + rg + r) * N + wn + j * 16 + l15;
                    const float v = rs[j][r] * 1.0001f + (float)lane;
                    out[o] = v;
                    if (TWO) out2[o] = v * 0.5f;
                }
        }
    } else {
        // 128 x 96 floats = 3072 float4 = 12 per thread, consecutive lanes consecutive 16-byte chunks
        float4 rs[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) rs[k] = RES ? reinterpret_cast<const float4*>(res + m0 * N)[tid + 256 * k] : make_float4(1.f, 1.f, 1.f, 1.f);
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            float4 v = rs[k];
            v.x = v.x * 1.0001f + (float)lane; v.y += 1.f; v.z += 2.f; v.w += 3.f;
            reinterpret_cast<float4*>(out + m0 * N)[tid + 256 * k] = v;
            if (TWO) { v.x *= 0.5f; reinterpret_cast<float4*>(out2 + m0 * N)[tid + 256 * k] = v; }
        }
    }
}

template <int MODE, bool RES, bool TWO>
static void run(hipStream_t st, const float* res, float* out, float* out2, int M, const char* tag) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_pat<MODE, RES, TWO>), dim3(M / BM), dim3(256), 0, st, res, out, out2, M);
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k_pat<MODE, RES, TWO>), dim3(M / BM), dim3(256), 0, st, res, out, out2, M);
    CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)M * N * 4 * ((RES ? 1 : 0) + 1 + (TWO ? 1 : 0));
    printf("  %-44s %7.1f us  %6.2f TB/s\n", tag, ms * 100.f, bytes / (ms * 1e-4) / 1e12);
}

int main() {
    const int M = 614400;
    hipStream_t st; CK(hipStreamCreate(&st));
    float *res, *out, *out2;
    CK(hipMalloc(&res, (size_t)M * N * 4)); CK(hipMalloc(&out, (size_t)M * N * 4)); CK(hipMalloc(&out2, (size_t)M * N * 4));
    CK(hipMemset(res, 0, (size_t)M * N * 4));
    printf("tile 128 x 96 fp32 per block, %d rows (236 MB per tensor)\n", M);
    run<0, false, false>(st, res, out, out2, M, "D layout: 1 store");
    run<1, false, false>(st, res, out, out2, M, "row-major float4: 1 store");
    run<0, true, false>(st, res, out, out2, M, "D layout: residual load + 1 store");
    run<1, true, false>(st, res, out, out2, M, "row-major float4: residual load + 1 store");
    run<0, true, true>(st, res, out, out2, M, "D layout: residual load + 2 stores");
    run<1, true, true>(st, res, out, out2, M, "row-major float4: residual load + 2 stores");
    return 0;
}
