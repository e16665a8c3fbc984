// Diagnostic (GPU box): what a host buffer costs to move.  Pageable vs pinned hipMemcpyAsync H2D / D2H at the sizes one slice of the
// headline workload moves, alone and from four host threads at once; the price of hipHostMalloc / hipHostRegister; first-touch
// of a fresh std::vector.  hipcc --offload-arch=gfx950 -O2 -o scripts/bin/pcie_probe scripts/pcie_probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
static void copy_test(const char *tag, void *h, void *d, size_t bytes, hipStream_t st) {
    for (int dir = 0; dir < 2; dir++) {
        double best = 1e9;
        for (int rep = 0; rep < 3; rep++) {
            const double t0 = now();
            if (dir == 0) (void)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st);
            else (void)hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, st);
            (void)hipStreamSynchronize(st);
            best = std::min(best, now() - t0);
        }
        printf("%-28s %s %4zu MB: %7.2f ms  %6.1f GB/s\n", tag, dir ? "D2H" : "H2D", bytes >> 20, best * 1e3, bytes / best / 1e9);
    }
}
int main() {
    CK(hipSetDevice(0));
    const size_t MB = 1 << 20;
    for (size_t bytes : {64 * MB, 256 * MB}) {
        void *d = nullptr;
        CK(hipMalloc(&d, bytes));
        hipStream_t st;
        CK(hipStreamCreate(&st));
        double t0 = now();
        std::vector<unsigned char> v(bytes);            // value-initialised: first touch
        printf("std::vector first touch      %4zu MB: %7.2f ms\n", bytes >> 20, (now() - t0) * 1e3);
        copy_test("pageable", v.data(), d, bytes, st);
        t0 = now();
        void *p = nullptr;
        CK(hipHostMalloc(&p, bytes, hipHostMallocDefault));
        printf("hipHostMalloc                %4zu MB: %7.2f ms\n", bytes >> 20, (now() - t0) * 1e3);
        t0 = now();
        memset(p, 1, bytes);
        printf("memset of the pinned block   %4zu MB: %7.2f ms\n", bytes >> 20, (now() - t0) * 1e3);
        copy_test("pinned", p, d, bytes, st);
        t0 = now();
        memcpy(p, v.data(), bytes);
        printf("memcpy pageable -> pinned    %4zu MB: %7.2f ms  %6.1f GB/s\n", bytes >> 20, (now() - t0) * 1e3, bytes / (now() - t0) / 1e9);
        t0 = now();
        CK(hipHostFree(p));
        printf("hipHostFree                  %4zu MB: %7.2f ms\n", bytes >> 20, (now() - t0) * 1e3);
        t0 = now();
        CK(hipHostRegister(v.data(), bytes, hipHostRegisterDefault));
        printf("hipHostRegister              %4zu MB: %7.2f ms\n", bytes >> 20, (now() - t0) * 1e3);
        copy_test("registered", v.data(), d, bytes, st);
        t0 = now();
        CK(hipHostUnregister(v.data()));
        printf("hipHostUnregister            %4zu MB: %7.2f ms\n", bytes >> 20, (now() - t0) * 1e3);
        CK(hipFree(d));
        CK(hipStreamDestroy(st));
    }
    // four threads, each its own stream and pageable buffers of 100 MB in + 55 MB out (one slice of the headline workload)
    for (int pinned = 0; pinned < 2; pinned++) {
        const size_t in_b = 100 * MB, out_b = 55 * MB;
        std::vector<std::thread> th;
        std::vector<double> secs(4);
        std::vector<void *> hp(4, nullptr);
        std::vector<std::vector<unsigned char>> hv(4);
        for (int i = 0; i < 4; i++) {
            if (pinned) { CK(hipHostMalloc(&hp[i], in_b, hipHostMallocDefault)); memset(hp[i], 1, in_b); }
            else { hv[i].assign(in_b, 1); hp[i] = hv[i].data(); }
        }
        const double t0 = now();
        for (int i = 0; i < 4; i++)
            th.emplace_back([&, i]() {
                (void)hipSetDevice(0);
                void *d = nullptr;
                hipStream_t st;
                (void)hipMalloc(&d, in_b);
                (void)hipStreamCreate(&st);
                const double a = now();
                (void)hipMemcpyAsync(d, hp[i], in_b, hipMemcpyHostToDevice, st);
                (void)hipStreamSynchronize(st);
                (void)hipMemcpyAsync(hp[i], d, out_b, hipMemcpyDeviceToHost, st);
                (void)hipStreamSynchronize(st);
                secs[i] = now() - a;
                (void)hipFree(d);
                (void)hipStreamDestroy(st);
            });
        for (auto &t : th) t.join();
        printf("4 threads x (100 MB in + 55 MB out), %s: wall %.1f ms, per thread %.1f %.1f %.1f %.1f ms\n", pinned ? "pinned" : "pageable",
               (now() - t0) * 1e3, secs[0] * 1e3, secs[1] * 1e3, secs[2] * 1e3, secs[3] * 1e3);
        if (pinned) for (int i = 0; i < 4; i++) (void)hipHostFree(hp[i]);
    }
    return 0;
}
