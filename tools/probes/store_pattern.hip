// Probe: how much does the GEMM epilogue's store pattern cost?  Writes an M x N bf16 matrix (a) the way the
// MFMA epilogue does it (per wave-instruction: 16 rows x 32 contiguous bytes) and (b) as full 128-byte lines
// (8 lanes x 16 B per row).  hipcc --offload-arch=gfx950 -O3 store_pattern.hip -o store_pattern && ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// one workgroup (512 threads, waves 2 x 4) per 192 x 192 tile, wave tile 96 x 48 (MI = 6, NI = 3)
__global__ void __launch_bounds__(512) pat_frag(uint16_t* C, int M, int N, int tiles_n) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
    const int m0 = (blockIdx.x / tiles_n) * 192, n0 = (blockIdx.x % tiles_n) * 192;
    const int wm0 = (wave >> 2) * 96, wn0 = (wave & 3) * 48;
    for (int i = 0; i < 6; ++i) {
        const int m = m0 + wm0 + i * 16 + l15;
        if (m >= M) continue;
        for (int j = 0; j < 3; ++j) {
            const int n = n0 + wn0 + j * 16 + g * 4;
            if (n >= N) continue;
            u32x2 v = {(uint32_t)m, (uint32_t)n};
            *(u32x2*)(C + (size_t)m * N + n) = v;
        }
    }
}
// same tile per workgroup, but 16 B per lane, consecutive lanes along the row (24 chunks per 384-byte tile row)
__global__ void __launch_bounds__(512) pat_line(uint16_t* C, int M, int N, int tiles_n) {
    const int tid = threadIdx.x;
    const int m0 = (blockIdx.x / tiles_n) * 192, n0 = (blockIdx.x % tiles_n) * 192;
    for (int c = tid; c < 192 * 24; c += 512) {
        const int r = c / 24, q = c % 24;
        const int m = m0 + r, n = n0 + q * 8;
        if (m >= M || n >= N) continue;
        u32x4 v = {(uint32_t)m, (uint32_t)n, 0u, 0u};
        *(u32x4*)(C + (size_t)m * N + n) = v;
    }
}

int main() {
    const int M = 8224, N = 4096;                     // ViT fc1 output: 67 MB
    uint16_t* C;
    hipMalloc(&C, (size_t)M * N * 2);
    char* flush;
    hipMalloc(&flush, 512u << 20);
    const int tm = (M + 191) / 192, tn = (N + 191) / 192;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pat = 0; pat < 2; ++pat)
        for (int rep = 0; rep < 5; ++rep) {
            hipMemsetAsync(flush, rep, 512u << 20, 0);
            hipEventRecord(e0, 0);
            if (pat == 0) hipLaunchKernelGGL(pat_frag, dim3(tm * tn), dim3(512), 0, 0, C, M, N, tn);
            else hipLaunchKernelGGL(pat_line, dim3(tm * tn), dim3(512), 0, 0, C, M, N, tn);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%s rep %d: %.1f us  %.2f TB/s\n", pat ? "full-line 16B/lane" : "fragment 16x32B  ", rep, ms * 1e3,
                   (double)M * N * 2 / (ms * 1e-3) / 1e12);
        }
    return 0;
}
