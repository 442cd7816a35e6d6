// Probe: what rate does HBM deliver for the ViT attention kernels' access pattern — per (frame, head) 3 x 257 pieces of 128 B
// at a 6144-byte stride (qkv rows [F*257][3072] bf16) — against the same bytes laid out contiguously per head?
// build: hipcc --offload-arch=gfx950 -O2 tools/probes/strided_read.hip -o tools/probes/strided_read
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
constexpr int VN = 257, VLD = 3072;
// one 512-thread workgroup per (frame, head); thread <-> (row, 16-byte chunk): 8 threads per 128-byte piece
template <bool CONTIG>
__global__ void __launch_bounds__(512) rd(const uint16_t* __restrict__ qkv, uint32_t* __restrict__ sink, int F) {
    const int f = blockIdx.x >> 4, h = blockIdx.x & 15, tid = threadIdx.x;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (int which = 0; which < 3; ++which)
        for (int s = tid; s < VN * 8; s += 512) {
            const int row = s >> 3, c = s & 7;
            const uint16_t* p = CONTIG ? qkv + ((((size_t)f * 3 + which) * 16 + h) * VN + row) * 64 + c * 8
                                       : qkv + ((size_t)f * VN + row) * VLD + which * 1024 + h * 64 + c * 8;
            const u32x4 v = *(const u32x4*)p;
            acc[0] ^= v[0]; acc[1] ^= v[1]; acc[2] ^= v[2]; acc[3] ^= v[3];
        }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[blockIdx.x] = acc[0];
}
int main() {
    const int F = 128;
    const size_t n = (size_t)F * VN * VLD;
    uint16_t* buf[3];
    for (auto& b : buf) { hipMalloc(&b, n * 2); hipMemset(b, 1, n * 2); }
    uint32_t* sink; hipMalloc(&sink, F * 16 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        float best = 1e9f, sum = 0.f;
        for (int it = 0; it < 23; ++it) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(rd<false>, dim3(F * 16), dim3(512), 0, 0, buf[it % 3], sink, F);
            else hipLaunchKernelGGL(rd<true>, dim3(F * 16), dim3(512), 0, 0, buf[it % 3], sink, F);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (it >= 3) { sum += ms; if (ms < best) best = ms; }
        }
        const double bytes = (double)F * 16 * 3 * VN * 128;
        printf("%s: mean %.1f us, best %.1f us -> %.2f TB/s (best), %.0f MB\n", mode ? "contiguous per head" : "128 B pieces at a 6 KB stride",
               sum / 20 * 1e3, best * 1e3, bytes / (best * 1e-3) / 1e12, bytes / 1e6);
    }
    return 0;
}
