// Diagnostic: cycles per hop of the rejected-proposal walk (v_readlane chain with an exit test), lone wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
// A: the shipped pattern
__device__ __forceinline__ uint32_t walk_a(uint32_t hopw, uint32_t &p) {
    uint32_t steps = 22;
#pragma unroll
    for (uint32_t k = 0; k < 22; k++) {
        const uint32_t hv = uni((uint32_t)__builtin_amdgcn_readlane((int)hopw, (int)p));
        if (!(hv & 64u)) { steps = k; break; }
        p = hv & 63u;
    }
    return steps;
}
// B: no test at all, fixed 8 hops
__device__ __forceinline__ uint32_t walk_b(uint32_t hopw, uint32_t &p) {
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) p = (uint32_t)__builtin_amdgcn_readlane((int)hopw, (int)p) & 63u;
    return 8;
}
// C: 8 hops, stop bits OR-ed and tested once
__device__ __forceinline__ uint32_t walk_c(uint32_t hopw, uint32_t &p, uint32_t &bad) {
    uint32_t acc = 64u;
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) { const uint32_t hv = (uint32_t)__builtin_amdgcn_readlane((int)hopw, (int)p); acc &= hv; p = hv & 63u; }
    bad = !(acc & 64u);
    return 8;
}
template <int MODE>
__global__ void probe(unsigned long long *stamps, uint32_t *out, int iters, uint32_t stop_every) {
    const uint32_t lane = threadIdx.x;
    // chain: nxt = (lane + 3) & 63, rejected unless lane % stop_every == 0
    uint32_t hopw = ((lane + 3) & 63u) | ((lane % stop_every) ? 64u : 0u);
    uint32_t p = 1, total = 0, bad = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) total += walk_a(hopw, p);
        if (MODE == 1) total += walk_b(hopw, p);
        if (MODE == 2) { total += walk_c(hopw, p, bad); }
        p = (p + 1 + bad) & 63u;  // "event": move on
        hopw ^= (total & 1u) << 8;  // keep the compiler honest
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { stamps[0] = t1 - t0; out[0] = total; out[1] = p; }
}
int main() {
    unsigned long long *st; uint32_t *out;
    hipMalloc(&st, 64); hipMalloc(&out, 64);
    const int iters = 200000;
    for (uint32_t se : {64u, 16u, 5u}) {
        unsigned long long h; uint32_t o[2];
        probe<0><<<1, 64>>>(st, out, iters, se); hipDeviceSynchronize();
        hipMemcpy(&h, st, 8, hipMemcpyDeviceToHost); hipMemcpy(o, out, 8, hipMemcpyDeviceToHost);
        printf("A stop_every=%u: %.1f cycles/call, %.2f hops/call -> %.1f cycles/hop\n", se, (double)h / iters, (double)o[0] / iters, (double)h / (o[0] + iters));
    }
    unsigned long long h; uint32_t o[2];
    probe<1><<<1, 64>>>(st, out, iters, 64); hipDeviceSynchronize();
    hipMemcpy(&h, st, 8, hipMemcpyDeviceToHost); hipMemcpy(o, out, 8, hipMemcpyDeviceToHost);
    printf("B (8 untested hops): %.1f cycles/hop\n", (double)h / o[0]);
    probe<2><<<1, 64>>>(st, out, iters, 64); hipDeviceSynchronize();
    hipMemcpy(&h, st, 8, hipMemcpyDeviceToHost); hipMemcpy(o, out, 8, hipMemcpyDeviceToHost);
    printf("C (8 hops, one test): %.1f cycles/hop\n", (double)h / o[0]);
    return 0;
}
