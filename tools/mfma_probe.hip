// mfma_probe: D = v_mfma_f32_16x16x32_bf16(A, B, C) for a file of cases, one wave per case.  Used to characterise the
// matrix core's accumulation arithmetic so that oracle/voxref.c can restate it bit for bit (tools/mfma_model.py).
//   mfma_probe in.bin out.bin [chain]
// in.bin : int32 n_cases, then per case A[16][32] bf16 (row-major, k fastest), B[32][16] bf16 ([k][col]), C[16][16] f32
// out.bin: per case D[16][16] f32.   chain > 1: the case's MFMA is applied `chain` times (D fed back as C).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

struct Case {
    uint16_t A[16][32];
    uint16_t B[32][16];
    float C[16][16];
};

__global__ __launch_bounds__(64) void k_probe(const Case* in, float* out, int n, int chain) {
    const int c = blockIdx.x, lane = threadIdx.x;
    if (c >= n) return;
    const Case& cs = in[c];
    const int fr = lane & 15, g = lane >> 4;
    union { uint16_t u[8]; bf16x8_t v; } a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a.u[i] = cs.A[fr][g * 8 + i];
        b.u[i] = cs.B[g * 8 + i][fr];
    }
    f32x4_t acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = cs.C[g * 4 + r][fr];
    for (int i = 0; i < chain; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) out[(size_t)c * 256 + (g * 4 + r) * 16 + fr] = acc[r];
}

int main(int argc, char** argv) {
    if (argc < 3) { fprintf(stderr, "usage: mfma_probe in.bin out.bin [chain]\n"); return 2; }
    const int chain = argc > 3 ? atoi(argv[3]) : 1;
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int32_t n = 0;
    if (fread(&n, 4, 1, f) != 1 || n <= 0) { fprintf(stderr, "bad header\n"); return 1; }
    std::vector<Case> cases(n);
    if (fread(cases.data(), sizeof(Case), n, f) != (size_t)n) { fprintf(stderr, "short file\n"); return 1; }
    fclose(f);
    Case* din = nullptr;
    float* dout = nullptr;
    if (hipMalloc(&din, sizeof(Case) * n) != hipSuccess || hipMalloc(&dout, (size_t)n * 256 * 4) != hipSuccess) return 1;
    hipMemcpy(din, cases.data(), sizeof(Case) * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_probe, dim3(n), dim3(64), 0, 0, din, dout, n, chain);
    if (hipDeviceSynchronize() != hipSuccess) { fprintf(stderr, "kernel failed\n"); return 1; }
    std::vector<float> out((size_t)n * 256);
    hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    f = fopen(argv[2], "wb");
    fwrite(out.data(), 4, out.size(), f);
    fclose(f);
    printf("mfma_probe: %d cases, chain %d\n", n, chain);
    return 0;
}
