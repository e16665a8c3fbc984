// Diagnostic: shader clock under light/heavy load and basic dependent-op latencies on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void chain(double* out, unsigned long long* stamps, int iters, int mode) {
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    double v = threadIdx.x * 1e-3 + 1.0, a = 1.0000001;
    uint64_t s = 88172645463325252ULL + blockIdx.x;
    int acc = 0;
    if (mode == 0) { for (int i = 0; i < iters; i++) v = v * a + 1e-9; }           // dependent f64 mul+add
    else if (mode == 1) { for (int i = 0; i < iters; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; s = s * 5 + (s >> 60);} acc = (int)s; } // scalar-ish u64
    else if (mode == 2) { for (int i = 0; i < iters; i++) { int l = (int)(s & 63); s = s * 6364136223846793005ULL + 1442695040888963407ULL; int lo = __builtin_amdgcn_readlane(__double2loint(v), l); v += (double)lo * 1e-30; } }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = v + acc;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}
int main() {
    double* out; unsigned long long* st; hipMalloc(&out, 8 * 64 * 4096); hipMalloc(&st, 16 * 4096);
    unsigned long long h[2];
    for (int blocks : {1, 8, 512, 2048}) for (int mode = 0; mode < 3; mode++) {
        int iters = 2000000;
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        chain<<<blocks, 64>>>(out, st, 1000, mode); hipDeviceSynchronize();
        hipEventRecord(a); chain<<<blocks, 64>>>(out, st, iters, mode); hipEventRecord(b); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, a, b); hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
        printf("blocks %4d mode %d: %.2f ms, %.1f ns/iter, shader cycles/iter %.1f, clock %.0f MHz\n", blocks, mode, ms, ms * 1e6 / iters,
               (double)h[0] / iters, (double)h[0] / (double)h[1] * 100.0);
    }
    return 0;
}
