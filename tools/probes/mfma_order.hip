// Probe (round 6): what in the ORDER of a wave's MFMAs moves the sustained (power-limited) rate?  One wave per SIMD owning 256
// accumulation registers BY NAME (as the persistent GEMMs), random operands in registers, whole chip, settled clocks; every MFMA
// is an asm statement, so the order below is the order issued (hipcc reschedules builtin MFMAs: the first version of this probe
// measured the scheduler's order, not the source's).
// 16x16x32: 64 blocks (8 x 8), a K tile = 2 K steps; 32x32x16: 16 blocks (4 x 4), a K tile = 4 K steps.  Orders:
//   0  K step outer, row i, column j inner: srcB (the i operand) held for 8 / 4 consecutive MFMAs, srcA cycling       [gemm_p4 / p32 today]
//   1  the same with the operands swapped in the instruction: srcA held, srcB cycling
//   2  block outer, K step inner: back-to-back MFMAs on the SAME accumulator (chains of 2 / 4), row operands held through the chain
//   3  K step outer, diagonal walk: neither operand repeats between consecutive MFMAs
// Every order adds the same products to the same accumulators in the same K order: the checksums must agree bit for bit.
//   hipcc --offload-arch=gfx950 -O3 mfma_order.hip -o mfma_order && ./mfma_order [iters]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;

#define A8(n) "a" #n "0", "a" #n "1", "a" #n "2", "a" #n "3", "a" #n "4", "a" #n "5", "a" #n "6", "a" #n "7", "a" #n "8", "a" #n "9"
#define ALL_AGPRS                                                                                                          \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", A8(1), A8(2), A8(3), A8(4), A8(5), A8(6), A8(7), A8(8), A8(9), A8(10), \
        A8(11), A8(12), A8(13), A8(14), A8(15), A8(16), A8(17), A8(18), A8(19), A8(20), A8(21), A8(22), A8(23), A8(24), "a250", "a251", \
        "a252", "a253", "a254", "a255"

template <int SHAPE>
__device__ __forceinline__ void mfma(int blk, const bf16x8& x, const bf16x8& y) {
    if constexpr (SHAPE == 16) asm volatile("v_mfma_f32_16x16x32_bf16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(x), "v"(y), "i"(4 * blk), "i"(4 * blk + 3));
    else asm volatile("v_mfma_f32_32x32x16_bf16 a[%2:%3], %0, %1, a[%2:%3]" ::"v"(x), "v"(y), "i"(16 * blk), "i"(16 * blk + 15));
}

template <int SHAPE, int ORDER>
__global__ void __launch_bounds__(256) mfma_loop(const uint32_t* __restrict__ seed, float* __restrict__ sink, int iters) {
    constexpr int NB = SHAPE == 16 ? 8 : 4, NS = SHAPE == 16 ? 2 : 4;      // blocks per side, K steps per K tile
    const int tid = threadIdx.x;
    bf16x8 a[16], b[16];                                     // [side block][K step]
    for (int i = 0; i < 16; ++i) {
        uint32_t w[4], v[4];
        for (int j = 0; j < 4; ++j) {
            const uint32_t s = seed[((blockIdx.x * 256 + tid) * 32 + (i & 3) * 8 + j) & 0xfffff] + i * 0x9e3779b9u;
            const uint32_t t = seed[((blockIdx.x * 256 + tid) * 32 + (i & 3) * 8 + 4 + j) & 0xfffff] ^ (i * 0x85ebca6bu);
            w[j] = (s & 0x807f807fu) | 0x3f003f00u;
            v[j] = (t & 0x807f807fu) | 0x3f003f00u;
        }
        a[i] = __builtin_bit_cast(bf16x8, *(uint4*)w);
        b[i] = __builtin_bit_cast(bf16x8, *(uint4*)v);
    }
    asm volatile("" ::: ALL_AGPRS);
#pragma unroll
    for (int r = 0; r < 256; ++r) asm volatile("v_accvgpr_write_b32 a[%0], 0" ::"i"(r));
    asm volatile("s_nop 7");
    for (int it = 0; it < iters; ++it) {
        if constexpr (ORDER == 0 || ORDER == 1) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int i = 0; i < NB; ++i)
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        if constexpr (ORDER == 0) mfma<SHAPE>(i * NB + j, b[j * NS + s], a[i * NS + s]);
                        else mfma<SHAPE>(i * NB + j, a[i * NS + s], b[j * NS + s]);
                    }
        } else if constexpr (ORDER == 2) {
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int s = 0; s < NS; ++s) mfma<SHAPE>(i * NB + j, b[j * NS + s], a[i * NS + s]);
        } else {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int d = 0; d < NB; ++d)
#pragma unroll
                    for (int i = 0; i < NB; ++i) mfma<SHAPE>(i * NB + (i + d) % NB, b[((i + d) % NB) * NS + s], a[i * NS + s]);
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    float out = 0.f;
#pragma unroll
    for (int r = 0; r < 256; ++r) {
        float v;
        asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(v) : "i"(r));
        if (ORDER == 1) out += v;                             // (the swapped form holds the transposed blocks: same multiset per wave)
        else out += v;
    }
    // one checksum per wave: the sum over lanes and registers is order independent only up to fp32 rounding of THIS final sum, so
    // sum as integers of the bit patterns instead
    if (sink) atomicAdd(&sink[0], 0.f);
    sink[1 + blockIdx.x * 256 + tid] = out;
}

template <int SHAPE, int ORDER>
static double run(const char* name, const uint32_t* seed, float* sink, int iters) {
    const int blocks = 256;
    const double flop = (double)blocks * 4 * iters * 64.0 * 32 * 32 * 16 * 2;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((mfma_loop<SHAPE, ORDER>), dim3(blocks), dim3(256), 0, 0, seed, sink, iters / 10);
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_loop<SHAPE, ORDER>), dim3(blocks), dim3(256), 0, 0, seed, sink, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // checksum of a SHORT run (8 iterations: values stay finite)
    hipLaunchKernelGGL((mfma_loop<SHAPE, ORDER>), dim3(blocks), dim3(256), 0, 0, seed, sink, 8);
    std::vector<float> h(1 + 256 * 256);
    hipMemcpy(h.data(), sink, h.size() * 4, hipMemcpyDeviceToHost);
    double cs = 0;
    for (size_t i = 1; i < h.size(); ++i) cs += (double)h[i];
    printf("%-44s %8.3f ms  %7.1f TFLOP/s   checksum %.9e\n", name, best, flop / best / 1e9, cs);
    return cs;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    const size_t words = (size_t)1 << 20;
    std::vector<uint32_t> h(words);
    uint32_t x = 0x12345678u;
    for (size_t i = 0; i < words; ++i) { x = x * 1664525u + 1013904223u; h[i] = x ^ (x >> 15); }
    uint32_t* seed;
    float* sink;
    hipMalloc(&seed, words * 4);
    hipMalloc(&sink, (1 + 256 * 256) * 4);
    hipMemcpy(seed, h.data(), words * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<16, 0>("16x16x32 srcB held x8, srcA cycling", seed, sink, iters);
        run<16, 1>("16x16x32 srcA held x8, srcB cycling", seed, sink, iters);
        run<16, 2>("16x16x32 chains of 2 on one accumulator", seed, sink, iters);
        run<16, 3>("16x16x32 no operand repeats", seed, sink, iters);
        run<32, 0>("32x32x16 srcB held x4, srcA cycling", seed, sink, iters);
        run<32, 1>("32x32x16 srcA held x4, srcB cycling", seed, sink, iters);
        run<32, 2>("32x32x16 chains of 4 on one accumulator", seed, sink, iters);
        run<32, 3>("32x32x16 no operand repeats", seed, sink, iters);
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("FAILED\n"); return 1; }
    return 0;
}
