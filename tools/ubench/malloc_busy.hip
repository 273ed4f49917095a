// Micro-benchmark (round 6): what hipMalloc of one 512-MiB pool chunk costs (a) on an idle device from the calling thread, (b) from the
// calling thread while a long kernel runs, (c) from a SECOND thread while the first thread launches and waits for short kernels --
// and what (c) does to the first thread's launch + synchronise round trips.  The pool's helper thread (csrc/obsgp_host.cpp) is (c).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/malloc_busy.hip -o /tmp/malloc_busy -lpthread && /tmp/malloc_busy
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#include <algorithm>

__global__ void spin(long long cycles, int* out) {
    const long long t0 = wall_clock64();          // 100 MHz
    while (wall_clock64() - t0 < cycles) {}
    if (out && threadIdx.x == 0 && blockIdx.x == 0) *out = 1;
}
// a kernel that keeps the memory system busy for `cycles` ticks of the 100 MHz clock: every workgroup streams over `n` floats again and again
__global__ void stream(long long cycles, float* buf, size_t n) {
    const long long t0 = wall_clock64();
    float acc = 0.f;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    while (wall_clock64() - t0 < cycles) {
        for (int r = 0; r < 64; ++r) { acc += buf[i]; i += (size_t)gridDim.x * blockDim.x; if (i >= n) i -= n; }
    }
    if (acc == 123.456f) buf[0] = acc;
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static const size_t kChunk = (size_t)512 << 20;

int main() {
    hipStream_t s; hipStreamCreate(&s);
    int* d; hipMalloc(&d, 4);
    spin<<<1, 64, 0, s>>>(1000, d); hipStreamSynchronize(s);
    std::vector<void*> held;
    auto timed_malloc = [&](const char* what) {
        void* p = nullptr; const double t0 = now_ms(); hipError_t e = hipMalloc(&p, kChunk); const double t1 = now_ms();
        printf("%-58s %7.2f ms%s\n", what, t1 - t0, e == hipSuccess ? "" : "  FAILED"); if (p) held.push_back(p);
    };
    for (int i = 0; i < 12; ++i) timed_malloc("hipMalloc 512 MiB, idle device");
    // (b) a 256-workgroup kernel of ~40 ms in flight (s_memtime counts at 100 MHz: 4e6 ticks)
    for (int i = 0; i < 3; ++i) {
        spin<<<256, 256, 0, s>>>(4000000, d);
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
        timed_malloc("hipMalloc 512 MiB, same thread, 40 ms kernel in flight");
        const double t0 = now_ms(); hipStreamSynchronize(s); printf("   ... kernel finished %.2f ms later\n", now_ms() - t0);
    }
    // (b2) the same under a kernel that streams 4 GB of global memory for 40 ms
    {
        float* big = nullptr; const size_t nbig = (size_t)1 << 30; hipMalloc(&big, nbig * 4); hipMemsetAsync(big, 0, nbig * 4, s); hipStreamSynchronize(s);
        for (int i = 0; i < 4; ++i) {
            stream<<<1024, 256, 0, s>>>(4000000, big, nbig);
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
            timed_malloc("hipMalloc 512 MiB, same thread, 40 ms STREAMING kernel");
            const double t0 = now_ms(); hipStreamSynchronize(s); printf("   ... kernel finished %.2f ms later\n", now_ms() - t0);
        }
        // and from a second thread while this thread waits for such kernels
        std::atomic<bool> go{false}, done{false};
        std::vector<double> mall;
        std::thread th([&] {
            hipSetDevice(0);
            while (!go.load()) std::this_thread::yield();
            for (int i = 0; i < 8; ++i) { void* p = nullptr; const double t0 = now_ms(); hipMalloc(&p, kChunk); mall.push_back(now_ms() - t0); if (p) held.push_back(p); }
            done.store(true);
        });
        std::vector<double> rt;
        go.store(true);
        while (!done.load()) { const double t0 = now_ms(); stream<<<1024, 256, 0, s>>>(500000, big, nbig); hipStreamSynchronize(s); rt.push_back(now_ms() - t0); }
        th.join();
        std::sort(rt.begin(), rt.end());
        printf("round trips of a 5 ms STREAMING kernel while a second thread allocates 8 chunks: %zu, median %.2f ms, max %.2f ms; hipMalloc calls:", rt.size(), rt[rt.size() / 2], rt.back());
        for (double m : mall) printf(" %.2f", m);
        printf("\n");
        hipFree(big);
    }
    // (c) a second thread allocates eight chunks back to back while this thread does launch + synchronise round trips of a ~20 us kernel
    for (int with = 0; with < 2; ++with) {
        std::atomic<bool> go{false}, done{false};
        std::vector<double> mall;
        std::thread th([&] {
            hipSetDevice(0);
            while (!go.load()) std::this_thread::yield();
            if (with) for (int i = 0; i < 16; ++i) { void* p = nullptr; const double t0 = now_ms(); hipMalloc(&p, kChunk); mall.push_back(now_ms() - t0); if (p) held.push_back(p); }
            else std::this_thread::sleep_for(std::chrono::milliseconds(30));
            done.store(true);
        });
        std::vector<double> rt;
        go.store(true);
        while (!done.load()) { const double t0 = now_ms(); spin<<<64, 256, 0, s>>>(2000, d); hipStreamSynchronize(s); rt.push_back(now_ms() - t0); }
        th.join();
        std::sort(rt.begin(), rt.end());
        printf("round trips of a 20 us kernel %s: %zu, median %.3f ms, max %.2f ms", with ? "while a second thread allocates 16 chunks" : "alone", rt.size(), rt[rt.size() / 2], rt.back());
        if (with) { printf("; the second thread's hipMalloc calls:"); for (double m : mall) printf(" %.2f", m); }
        printf("\n");
    }
    // (d) the same with a long kernel per round trip (a training in flight: ~5 ms)
    {
        std::atomic<bool> go{false}, done{false};
        std::vector<double> mall;
        std::thread th([&] {
            hipSetDevice(0);
            while (!go.load()) std::this_thread::yield();
            for (int i = 0; i < 8; ++i) { void* p = nullptr; const double t0 = now_ms(); hipMalloc(&p, kChunk); mall.push_back(now_ms() - t0); if (p) held.push_back(p); }
            done.store(true);
        });
        std::vector<double> rt;
        go.store(true);
        while (!done.load()) { const double t0 = now_ms(); spin<<<256, 256, 0, s>>>(500000, d); hipStreamSynchronize(s); rt.push_back(now_ms() - t0); }
        th.join();
        std::sort(rt.begin(), rt.end());
        printf("round trips of a 5 ms kernel while a second thread allocates 8 chunks: %zu, median %.2f ms, max %.2f ms; hipMalloc calls:", rt.size(), rt[rt.size() / 2], rt.back());
        for (double m : mall) printf(" %.2f", m);
        printf("\n");
    }
    // (e) does the cost depend on how much the process already holds?  96 more chunks (to ~70 GB), every call above 0.1 ms printed;
    //     then touch the newest chunk from a kernel for the first time
    {
        double worst = 0; int slow = 0;
        for (int i = 0; i < 96; ++i) {
            void* p = nullptr; const double t0 = now_ms(); hipMalloc(&p, kChunk); const double dt = now_ms() - t0; if (p) held.push_back(p);
            if (dt > 0.1) { ++slow; printf("   chunk %zu (%.1f GB held): %.2f ms\n", held.size(), held.size() * 0.5, dt); }
            worst = std::max(worst, dt);
        }
        printf("96 more chunks: %d calls above 0.1 ms, worst %.2f ms, %.1f GB held\n", slow, worst, held.size() * 0.5);
        const double t0 = now_ms(); hipMemsetAsync(held.back(), 0, kChunk, s); hipStreamSynchronize(s);
        const double t1 = now_ms(); hipMemsetAsync(held.back(), 0, kChunk, s); hipStreamSynchronize(s);
        printf("first memset of a fresh chunk %.2f ms, second %.2f ms\n", t1 - t0, now_ms() - t1);
    }
    { const double t0 = now_ms(); for (void* p : held) hipFree(p); printf("hipFree of %zu chunks: %.1f ms\n", held.size(), now_ms() - t0); }
    return 0;
}
