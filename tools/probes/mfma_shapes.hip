// Probe: sustained bf16 MFMA rate on RANDOM operands for the two instruction shapes the GEMM could be built from,
// v_mfma_f32_16x16x32_bf16 (the kernels' shape) and v_mfma_f32_32x32x16_bf16 (twice the flops per operand register),
// with the whole chip busy for a few milliseconds so that the power-limited clock settles.  Register-resident
// operands (no LDS, no memory): the ceiling the kernels can approach, and whether the 32x32 shape buys clock.
// Second pair: the same loops with one ds_read_b128 per MFMA operand pair (the kernels' LDS traffic per flop for
// a 96x32 / 64x64 wave tile is ~1 b128 read per 1.5-2 MFMAs).
//   hipcc --offload-arch=gfx950 -O3 mfma_shapes.hip -o mfma_shapes && ./mfma_shapes
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

template <int SHAPE, int LDS>
__global__ void __launch_bounds__(256) mfma_loop(const uint32_t* __restrict__ seed, float* __restrict__ sink, int iters) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[256 * 4 * 4];
    const int tid = threadIdx.x;
    // random bf16 operands: 4 A and 4 B fragments per lane, mantissas from the seed buffer, exponents near 1.0
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        uint32_t w[4], v[4];
        for (int j = 0; j < 4; ++j) {
            const uint32_t s = seed[(blockIdx.x * 256 + tid) * 32 + i * 8 + j];
            const uint32_t t = seed[(blockIdx.x * 256 + tid) * 32 + i * 8 + 4 + j];
            w[j] = (s & 0x807f807fu) | 0x3f003f00u;             // two bf16 in [0.5, 1) with random sign / mantissa
            v[j] = (t & 0x807f807fu) | 0x3f003f00u;
            lds[(tid * 4 + i) * 4 + j] = w[j];
        }
        a[i] = __builtin_bit_cast(bf16x8, *(uint4*)w);
        b[i] = __builtin_bit_cast(bf16x8, *(uint4*)v);
    }
    __syncthreads();
    float out = 0.f;
    if constexpr (SHAPE == 16) {
        f32x4 c[16];
        for (int i = 0; i < 16; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if constexpr (LDS) a[i] = __builtin_bit_cast(bf16x8, *(const uint4*)&lds[((tid ^ (it & 63)) * 4 + i) * 4]);
#pragma unroll
                for (int j = 0; j < 4; ++j) c[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], c[i * 4 + j], 0, 0, 0);
            }
        }
        for (int i = 0; i < 16; ++i) out += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    } else {
        f32x16 c[4];
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if constexpr (LDS) a[i] = __builtin_bit_cast(bf16x8, *(const uint4*)&lds[((tid ^ (it & 63)) * 4 + i) * 4]);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    c[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], c[i * 2 + j], 0, 0, 0);
                    c[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i + 2], b[j + 2], c[i * 2 + j], 0, 0, 0);
                }
            }
        }
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) out += c[i][r];
    }
    if (out == 123.456f) sink[0] = out;                       // keep the loop alive
}

template <int SHAPE, int LDS>
static void run(const char* name, const uint32_t* seed, float* sink, int iters) {
    const int blocks = 256 * 4;                              // 4 workgroups x 4 waves per CU = 4 waves per SIMD
    // flops per iteration and wave: SHAPE 16: 16 MFMAs x 16*16*32*2; SHAPE 32: 8 MFMAs x 32*32*16*2 — both 262144
    const double flop = (double)blocks * 4 * iters * 16.0 * 16 * 16 * 32 * 2;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((mfma_loop<SHAPE, LDS>), dim3(blocks), dim3(256), 0, 0, seed, sink, iters / 10);   // warm
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_loop<SHAPE, LDS>), dim3(blocks), dim3(256), 0, 0, seed, sink, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s run %d: %8.3f ms  %7.1f TFLOP/s\n", name, r, ms, flop / ms / 1e9);
        if (ms < best) best = ms;
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 500;        // 500: 0.3-0.4 ms per launch; 20000: ~15 ms (settled clocks)
    const size_t words = (size_t)256 * 4 * 256 * 32;
    std::vector<uint32_t> h(words);
    uint32_t x = 0x12345678u;
    for (size_t i = 0; i < words; ++i) { x = x * 1664525u + 1013904223u; h[i] = x ^ (x >> 15); }
    uint32_t* seed;
    float* sink;
    hipMalloc(&seed, words * 4);
    hipMalloc(&sink, 4);
    hipMemcpy(seed, h.data(), words * 4, hipMemcpyHostToDevice);
    run<16, 0>("16x16x32 registers", seed, sink, iters);
    run<32, 0>("32x32x16 registers", seed, sink, iters);
    run<16, 1>("16x16x32 + ds_read_b128", seed, sink, iters);
    run<32, 1>("32x32x16 + ds_read_b128", seed, sink, iters);
    if (hipDeviceSynchronize() != hipSuccess) { printf("FAILED\n"); return 1; }
    return 0;
}
