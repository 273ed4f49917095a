// Packed cluster models for the multi-GPU exchange (per-cluster training shards across the GPUs of a
// node; what test() needs from a trained cluster travels in ONE padded record per model so that an RCCL all-gather of
// equal-size slots assembles the map on every rank).  The reference has no counterpart (single process).
//
// Record layout (stride bytes, multiple of 256):
//   [0, 64)      header: int dim, N, ng, K, ld, nb; float scale; int reserved[9]
//   [64, ..)     rowinfo[ld] (int)  |  x4[N][4] (float)  |  Xt tiles [ld/32 (ld/32 + 1) / 2][1024] (float), each piece 256-B aligned
// Only what K4 reads is shipped (Xt carries alpha in row K): 2 K^2 + 20 K bytes instead of the 10 K^2 of a trained model.
#include "ongpis.h"

namespace gpis {

__host__ __device__ inline size_t pk_align(size_t v) { return (v + 255) & ~(size_t)255; }
__host__ __device__ inline size_t pk_off_ri() { return 64; }
__host__ __device__ inline size_t pk_off_x4(int ld) { return pk_align(64 + sizeof(int) * (size_t)ld); }
__host__ __device__ inline size_t pk_off_xt(int ld, int N) { return pk_align(pk_off_x4(ld) + 16 * (size_t)N); }
__host__ __device__ inline size_t pk_bytes(int ld, int N) {
    const size_t nbx = ld / 32;
    return pk_align(pk_off_xt(ld, N) + sizeof(float) * 1024 * nbx * (nbx + 1) / 2);
}
size_t packed_model_bytes(int ld, int N) { return pk_bytes(ld, N); }

// grid = models, block = 256: model -> record (PACK) or record -> model (the descriptor must already point at allocated
// storage of the right size)
template <bool PACK>
__global__ __launch_bounds__(256) void model_pack_kernel(const ClusterModel* __restrict__ models, const int* __restrict__ slots,
                                                         char* __restrict__ buf, size_t stride) {
    const ClusterModel m = models[slots[blockIdx.x]];
    char* rec = buf + (size_t)blockIdx.x * stride;
    const int tid = threadIdx.x;
    if (PACK && tid == 0) {
        int* h = reinterpret_cast<int*>(rec);
        h[0] = m.dim; h[1] = m.N; h[2] = m.ng; h[3] = m.K; h[4] = m.ld; h[5] = m.nb;
        reinterpret_cast<float*>(rec)[6] = m.scale;
        for (int i = 7; i < 16; ++i) h[i] = 0;
    }
    const int ld = m.ld, N = m.N;
    int* ri = reinterpret_cast<int*>(rec + pk_off_ri());
    float4* x4 = reinterpret_cast<float4*>(rec + pk_off_x4(ld));
    float4* xt = reinterpret_cast<float4*>(rec + pk_off_xt(ld, N));
    const size_t nxt = (size_t)256 * (ld / 32) * (ld / 32 + 1) / 2;   // float4s
    if (PACK) {
        for (int i = tid; i < ld; i += 256) ri[i] = m.rowinfo[i];
        for (int i = tid; i < N; i += 256) x4[i] = reinterpret_cast<const float4*>(m.x4)[i];
        for (size_t i = tid; i < nxt; i += 256) xt[i] = reinterpret_cast<const float4*>(m.Xt)[i];
    } else {
        for (int i = tid; i < ld; i += 256) m.rowinfo[i] = ri[i];
        for (int i = tid; i < N; i += 256) reinterpret_cast<float4*>(m.x4)[i] = x4[i];
        for (size_t i = tid; i < nxt; i += 256) reinterpret_cast<float4*>(m.Xt)[i] = xt[i];
    }
}

void model_pack_launch(bool pack, const ClusterModel* d_models, const int* d_slots, int n, char* d_buf, size_t stride, hipStream_t s) {
    if (n <= 0) return;
    if (pack) hipLaunchKernelGGL((model_pack_kernel<true>), dim3(n), dim3(256), 0, s, d_models, d_slots, d_buf, stride);
    else hipLaunchKernelGGL((model_pack_kernel<false>), dim3(n), dim3(256), 0, s, d_models, d_slots, d_buf, stride);
}

}  // namespace gpis
