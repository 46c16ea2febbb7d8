// Development microbench (round 6): does touching ONE dword per cache line of a weight matrix pull it into the Infinity Cache, so that a
// later streaming read of it (the next GEMM) runs faster than from HBM?  Sequence per trial: evict (stream 1 GiB) -> [touch W] -> timed
// stream of W (nt loads, 256 blocks x 512 threads, the GEMMs' shape).  Sizes: 8 .. 128 MB.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mall_prefetch_probe tools/mall_prefetch_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
template <int NT>
__global__ __launch_bounds__(512) void k_stream(const u32x4_t* p, size_t n16, unsigned* sink) {
    unsigned acc = 0;
    const size_t stride = (size_t)gridDim.x * 512;
    size_t i = (size_t)blockIdx.x * 512 + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        u32x4_t a, b, c, d;
        if (NT) { a = __builtin_nontemporal_load(p + i); b = __builtin_nontemporal_load(p + i + stride); c = __builtin_nontemporal_load(p + i + 2 * stride); d = __builtin_nontemporal_load(p + i + 3 * stride); }
        else { a = p[i]; b = p[i + stride]; c = p[i + 2 * stride]; d = p[i + 3 * stride]; }
        acc ^= a.x ^ b.y ^ c.z ^ d.w;
    }
    for (; i < n16; i += stride) acc ^= p[i].x;
    if (acc == 0x12345678u) *sink = acc;
}
// one dword per `step` bytes, fire and forget (the value is never used: the wave ends when the loads have returned)
__global__ __launch_bounds__(256) void k_touch(const char* p, size_t bytes, int step, unsigned* sink) {
    // (the destination register stays tied to `v` until the wait: an asm load the compiler believes complete would otherwise land in a
    // register it has already given to something else)
    unsigned v = 0;
    for (size_t o = ((size_t)blockIdx.x * 256 + threadIdx.x) * step; o < bytes; o += (size_t)gridDim.x * 256 * step)
        asm volatile("global_load_dword %0, %1, off" : "+v"(v) : "v"(p + o) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) :: "memory");
    if (v == 0x12345679u) *sink = v;
}
int main() {
    const size_t EV = (size_t)1 << 30;
    char *ev, *w; unsigned* sink;
    CK(hipMalloc(&ev, EV)); CK(hipMalloc(&w, (size_t)128 << 20)); CK(hipMalloc(&sink, 4));
    CK(hipMemset(ev, 1, EV)); CK(hipMemset(w, 2, (size_t)128 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timed_stream = [&](size_t bytes, int nt) {
        CK(hipEventRecord(e0));
        if (nt) hipLaunchKernelGGL(k_stream<1>, dim3(256), dim3(512), 0, 0, (const u32x4_t*)w, bytes / 16, sink);
        else hipLaunchKernelGGL(k_stream<0>, dim3(256), dim3(512), 0, 0, (const u32x4_t*)w, bytes / 16, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e3f;
    };
    auto evict = [&]() { hipLaunchKernelGGL(k_stream<0>, dim3(1024), dim3(512), 0, 0, (const u32x4_t*)ev, EV / 16, sink); CK(hipDeviceSynchronize()); };
    for (size_t mb : {8, 16, 25, 50, 100, 128}) {
        const size_t bytes = mb << 20;
        float cold = 0, hot = 0, t128 = 0, t64 = 0, tt = 0;
        const int R = 5;
        for (int r = 0; r < R; ++r) {
            evict(); cold += timed_stream(bytes, 1);
            hot += timed_stream(bytes, 1);                         // right after a full (nt) read of it
            evict();
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, 0, w, bytes, 128, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tt += ms * 1e3f;
            t128 += timed_stream(bytes, 1);
            evict();
            hipLaunchKernelGGL(k_touch, dim3(256), dim3(256), 0, 0, w, bytes, 64, sink);
            CK(hipDeviceSynchronize());
            t64 += timed_stream(bytes, 1);
        }
        printf("%4zu MB: stream cold %7.2f us (%.2f TB/s) | again right after %7.2f us (%.2f TB/s) | after touch/128B %7.2f us (%.2f TB/s; the touch itself %.2f us) | after touch/64B %7.2f us\n",
               mb, cold / R, bytes / (cold / R) * 1e-6, hot / R, bytes / (hot / R) * 1e-6, t128 / R, bytes / (t128 / R) * 1e-6, tt / R, t64 / R);
    }
    return 0;
}
