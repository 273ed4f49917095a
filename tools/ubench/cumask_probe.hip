// Probe: which CUs a stream created with hipExtStreamCreateWithCUMask uses (XCC id + hardware CU / SE ids per workgroup),
// and whether an unmasked high-priority stream gets the CUs a masked stream leaves out while the masked one is busy.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ab/cumask_probe tools/ubench/cumask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void probe(unsigned* out, long long spin) {
    unsigned xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}
static void run(const char* name, hipStream_t s, unsigned* d, int nwg) {
    hipLaunchKernelGGL(probe, dim3(nwg), dim3(512), 0, s, d, 20000LL);   // 200 us at 100 MHz: every CU gets workgroups
    hipStreamSynchronize(s);
    std::vector<unsigned> h(2 * nwg);
    hipMemcpy(h.data(), d, 8 * nwg, hipMemcpyDeviceToHost);
    std::set<unsigned> cus; int per_xcc[8] = {0};
    std::set<unsigned> cu_per_xcc[8];
    for (int i = 0; i < nwg; ++i) {
        const unsigned xcc = h[2 * i] & 0xF, hw = h[2 * i + 1];
        const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        cus.insert((xcc << 16) | (se << 8) | (sh << 4) | cu);
        cu_per_xcc[xcc & 7].insert((se << 8) | (sh << 4) | cu);
    }
    printf("%-28s distinct CUs used %zu | per XCC:", name, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %zu", cu_per_xcc[x].size());
    printf(" | first 16 workgroups -> XCC:");
    for (int i = 0; i < 16; ++i) printf(" %u", h[2 * i] & 0xF);
    printf("\n");
}
int main() {
    unsigned* d; const int nwg = 2048;
    hipMalloc(&d, 8 * nwg);
    hipStream_t s0; hipStreamCreate(&s0);
    run("no mask", s0, d, nwg);
    uint32_t m[8];
    for (int i = 0; i < 8; ++i) m[i] = 0xFFFFFFFFu;
    m[7] = 0; hipStream_t s1; if (hipExtStreamCreateWithCUMask(&s1, 8, m) != hipSuccess) { printf("mask stream failed\n"); return 1; }
    run("bits 0..223", s1, d, nwg);
    for (int i = 0; i < 8; ++i) m[i] = 0x7F7F7F7Fu;      // every eighth bit cleared
    hipStream_t s2; hipExtStreamCreateWithCUMask(&s2, 8, m); run("bit i%8==7 cleared", s2, d, nwg);
    for (int i = 0; i < 8; ++i) m[i] = 0xFFFFFFFFu; m[0] = 0xFFFFFF00u;
    hipStream_t s3; hipExtStreamCreateWithCUMask(&s3, 8, m); run("bits 0..7 cleared", s3, d, nwg);
    // concurrency: masked stream busy for ~5 ms, then time a tiny kernel on an unmasked high-priority stream
    int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStream_t sp; hipStreamCreateWithPriority(&sp, hipStreamNonBlocking, hi);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int which = 0; which < 2; ++which) {
        hipStream_t busy = which ? s1 : s0;
        hipLaunchKernelGGL(probe, dim3(1024), dim3(512), 0, busy, d, 200000LL);   // 4 rounds x 2 ms on every CU the stream may use
        hipEventRecord(e0, sp);
        hipLaunchKernelGGL(probe, dim3(64), dim3(256), 0, sp, d + 4096 - 128, 100LL);
        hipEventRecord(e1, sp); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipDeviceSynchronize();
        printf("small high-priority kernel beside a busy %s stream: %.3f ms\n", which ? "MASKED (224 CUs)" : "unmasked", ms);
    }
    return 0;
}
