// Packed cluster models for the multi-GPU exchange (per-cluster training shards across the GPUs of a
// node; what test() needs from a trained cluster travels in ONE record per model).  Records have their OWN size --
// pk_bytes(ld, N), a multiple of 256 -- and sit back to back at the byte offsets the caller gives (every rank can compute
// every rank's offsets: the sizes of all jobs of a frame are known everywhere), so an exchange moves the records' bytes and
// nothing else (round 3 padded every record to the frame's largest: 4.5 x the bytes at F = 5).  The reference has no
// counterpart (single process).
//
// Record layout:
//   [0, 64)      header: int dim, N, ng, K, ld, nb; float scale; int kind; int reserved[8]
//   [64, ..)     rowinfo[ld] (int)  |  x4[N][4] (float)  |  tiles [ld/32 (ld/32 + 1) / 2][1024] (float)  |  alpha[ld] (float), each piece 256-B aligned
// kind 0 (prediction record): the tiles are Xt -- only what K4 reads is shipped (Xt carries alpha in row K): 2 K^2 + 24 K bytes instead
// of the 10 K^2 of a trained model; the alpha piece is unused.
// kind 1 (factor record, round 5): the tiles are Lt (-L re-tiled, inverted diagonal blocks) and alpha travels beside them -- the same
// bytes; the receiver computes X = L^-1 itself (K3b) when it first predicts with the model, so a sharded update() keeps the lazy
// inverse: a cluster retrained in several consecutive updates is never inverted in between, on any rank.  Every record has the
// same size whatever its kind: every rank can still lay out every rank's buffer from (ld, N) alone.
#include "ongpis.h"

namespace gpis {

__host__ __device__ inline size_t pk_align(size_t v) { return (v + 255) & ~(size_t)255; }
__host__ __device__ inline size_t pk_off_ri() { return 64; }
__host__ __device__ inline size_t pk_off_x4(int ld) { return pk_align(64 + sizeof(int) * (size_t)ld); }
__host__ __device__ inline size_t pk_off_xt(int ld, int N) { return pk_align(pk_off_x4(ld) + 16 * (size_t)N); }
__host__ __device__ inline size_t pk_off_alpha(int ld, int N) {
    const size_t nbx = ld / 32;
    return pk_align(pk_off_xt(ld, N) + sizeof(float) * 1024 * nbx * (nbx + 1) / 2);
}
__host__ __device__ inline size_t pk_bytes(int ld, int N) { return pk_align(pk_off_alpha(ld, N) + sizeof(float) * (size_t)ld); }
size_t packed_model_bytes(int ld, int N) { return pk_bytes(ld, N); }

// grid = models, block = 256: model -> record (PACK) or record -> model (the descriptor must already point at allocated
// storage of the right size and kind).  Bit 30 of a slot entry = factor record (kind 1): tiles <-> Lt, alpha <-> alpha.
constexpr int kPackFactorBit = 1 << 30;
template <bool PACK>
__global__ __launch_bounds__(256) void model_pack_kernel(const ClusterModel* __restrict__ models, const int* __restrict__ slots,
                                                         char* __restrict__ buf, const unsigned long long* __restrict__ offs) {
    const int entry = slots[blockIdx.x];
    const bool factor = (entry & kPackFactorBit) != 0;
    const ClusterModel m = models[entry & ~kPackFactorBit];
    char* rec = buf + offs[blockIdx.x];
    const int tid = threadIdx.x;
    if (PACK && tid == 0) {
        int* h = reinterpret_cast<int*>(rec);
        h[0] = m.dim; h[1] = m.N; h[2] = m.ng; h[3] = m.K; h[4] = m.ld; h[5] = m.nb;
        reinterpret_cast<float*>(rec)[6] = m.scale;
        h[7] = factor ? 1 : 0;
        for (int i = 8; i < 16; ++i) h[i] = 0;
    }
    const int ld = m.ld, N = m.N;
    int* ri = reinterpret_cast<int*>(rec + pk_off_ri());
    float4* x4 = reinterpret_cast<float4*>(rec + pk_off_x4(ld));
    float4* xt = reinterpret_cast<float4*>(rec + pk_off_xt(ld, N));
    float* al = reinterpret_cast<float*>(rec + pk_off_alpha(ld, N));
    const size_t nxt = (size_t)256 * (ld / 32) * (ld / 32 + 1) / 2;   // float4s
    float* tiles = factor ? m.Lt : m.Xt;
    if (PACK) {
        for (int i = tid; i < ld; i += 256) ri[i] = m.rowinfo[i];
        for (int i = tid; i < N; i += 256) x4[i] = reinterpret_cast<const float4*>(m.x4)[i];
        for (size_t i = tid; i < nxt; i += 256) xt[i] = reinterpret_cast<const float4*>(tiles)[i];
        for (int i = tid; i < ld; i += 256) al[i] = factor ? m.alpha[i] : 0.f;
    } else {
        for (int i = tid; i < ld; i += 256) m.rowinfo[i] = ri[i];
        for (int i = tid; i < N; i += 256) reinterpret_cast<float4*>(m.x4)[i] = x4[i];
        for (size_t i = tid; i < nxt; i += 256) reinterpret_cast<float4*>(tiles)[i] = xt[i];
        if (factor) for (int i = tid; i < ld; i += 256) m.alpha[i] = al[i];
    }
}

void model_pack_launch(bool pack, const ClusterModel* d_models, const int* d_slots, int n, char* d_buf, const unsigned long long* d_offs, hipStream_t s) {
    if (n <= 0) return;
    if (pack) hipLaunchKernelGGL((model_pack_kernel<true>), dim3(n), dim3(256), 0, s, d_models, d_slots, d_buf, d_offs);
    else hipLaunchKernelGGL((model_pack_kernel<false>), dim3(n), dim3(256), 0, s, d_models, d_slots, d_buf, d_offs);
}

// the 64-byte headers of n records gathered into one array (one copy to the host instead of n)
__global__ void model_headers_kernel(const char* __restrict__ buf, const unsigned long long* __restrict__ offs, int n, int* __restrict__ out) {
    const int i = blockIdx.x * 16 + (threadIdx.x >> 4), w = threadIdx.x & 15;
    if (i < n) out[16 * i + w] = reinterpret_cast<const int*>(buf + offs[i])[w];
}
void model_headers_launch(const char* d_buf, const unsigned long long* d_offs, int n, int* d_out, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(model_headers_kernel, dim3((n + 15) / 16), dim3(256), 0, s, d_buf, d_offs, n, d_out);
}

}  // namespace gpis
