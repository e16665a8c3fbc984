// Diagnostic: cycles per xoshiro256** draw on the scalar unit, and per DPP-shift serial-scan step (f64).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
__global__ void rng(uint64_t* out, unsigned long long* stamps, int iters, uint64_t seed) {
    uint64_t s0 = seed, s1 = seed * 3 + 1, s2 = seed ^ 0x9e3779b97f4a7c15ULL, s3 = ~seed;
    s0 = __builtin_amdgcn_readfirstlane((uint32_t)s0) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(s0 >> 32)) << 32);
    s1 = __builtin_amdgcn_readfirstlane((uint32_t)s1) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(s1 >> 32)) << 32);
    s2 = __builtin_amdgcn_readfirstlane((uint32_t)s2) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(s2 >> 32)) << 32);
    s3 = __builtin_amdgcn_readfirstlane((uint32_t)s3) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(s3 >> 32)) << 32);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    uint64_t acc = 0; uint32_t cnt = 0;
    for (int i = 0; i < iters; i++) {
        uint64_t r = rotl(s1 * 5, 7) * 9, t = s1 << 17;
        s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3; s2 ^= t; s3 = rotl(s3, 45);
        cnt += (uint32_t)(r >> 63);     // the top-bit test of gen_index(1)
        acc ^= r;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) { out[blockIdx.x] = acc + cnt; stamps[blockIdx.x] = t1 - t0; }
}
__device__ __forceinline__ double shr1(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x111, 0xF, 0xF, true);  // row_shr:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x111, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__global__ void scan(double* out, unsigned long long* stamps, int iters) {
    double term = threadIdx.x * 0.25 + 1.0, acc = term;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 1; k < 8; k++) { double v = shr1(acc); acc = ((threadIdx.x & 7) == k) ? v + term : acc; }
        term += 1e-9;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) stamps[0] = t1 - t0;
}
int main() {
    uint64_t* out; unsigned long long* st; double* dout;
    hipMalloc(&out, 8 * 4096); hipMalloc(&st, 8 * 4096); hipMalloc(&dout, 8 * 64);
    unsigned long long h;
    int iters = 1000000;
    rng<<<1, 64>>>(out, st, 1000, 12345); hipDeviceSynchronize();
    rng<<<1, 64>>>(out, st, iters, 12345); hipDeviceSynchronize();
    hipMemcpy(&h, st, 8, hipMemcpyDeviceToHost);
    printf("xoshiro next_u64 on SALU: %.1f cycles/draw\n", (double)h / iters);
    scan<<<1, 64>>>(dout, st, 1000); hipDeviceSynchronize();
    scan<<<1, 64>>>(dout, st, iters / 10); hipDeviceSynchronize();
    hipMemcpy(&h, st, 8, hipMemcpyDeviceToHost);
    printf("DPP serial scan: %.1f cycles per 7-step scan = %.1f per step\n", (double)h / (iters / 10), (double)h / (iters / 10) / 7);
    return 0;
}
