// Probe (round 6): issue cost of VALU-class instructions on one SIMD, in shader cycles per wave instruction, with 1 and 4 waves per SIMD —
// the numbers behind DESIGN.md §4.3's accounting of the ViT attention (v_exp_f32 16, plain VALU 4, MFMA 16) and §9's note on the
// activation epilogues.  Each case is 512 independent instructions in asm (8 accumulator chains, no dependent pairs closer than 8).
//   hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY(NAME, ASM)                                                                                          \
    __global__ void __launch_bounds__(1024) NAME(unsigned long long* out, float seed, int iters) {              \
        float v0 = seed + threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7; \
        float w0 = v0 * 0.5f, w1 = w0, w2 = w0, w3 = w0, w4 = w0, w5 = w0, w6 = w0, w7 = w0;                      \
        const unsigned long long t0 = __builtin_readcyclecounter();                                             \
        for (int i = 0; i < iters; ++i) {                                                                       \
            _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                     \
                asm volatile(ASM : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7), \
                             "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3), "+v"(w4), "+v"(w5), "+v"(w6), "+v"(w7));     \
            }                                                                                                   \
        }                                                                                                       \
        const unsigned long long t1 = __builtin_readcyclecounter();                                             \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;                        \
        if (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 + w0 + w1 + w2 + w3 + w4 + w5 + w6 + w7 == 12345.f) out[0] = 0; \
    }
#define I8(OP) OP " %0, %0\n" OP " %1, %1\n" OP " %2, %2\n" OP " %3, %3\n" OP " %4, %4\n" OP " %5, %5\n" OP " %6, %6\n" OP " %7, %7\n"
#define F8(OP) OP " %0, %0, %8, %0\n" OP " %1, %1, %9, %1\n" OP " %2, %2, %10, %2\n" OP " %3, %3, %11, %3\n" OP " %4, %4, %12, %4\n" OP " %5, %5, %13, %5\n" OP " %6, %6, %14, %6\n" OP " %7, %7, %15, %7\n"
BODY(k_exp32, I8("v_exp_f32"))
BODY(k_rcp32, I8("v_rcp_f32"))
BODY(k_exp16, I8("v_exp_f16"))
BODY(k_rcp16, I8("v_rcp_f16"))
BODY(k_rsq32, I8("v_rsq_f32"))
BODY(k_fma32, F8("v_fma_f32"))
BODY(k_mov, I8("v_mov_b32"))
BODY(k_cvtpk, "v_cvt_pk_bf16_f32 %0, %0, %8\nv_cvt_pk_bf16_f32 %1, %1, %9\nv_cvt_pk_bf16_f32 %2, %2, %10\nv_cvt_pk_bf16_f32 %3, %3, %11\nv_cvt_pk_bf16_f32 %4, %4, %12\nv_cvt_pk_bf16_f32 %5, %5, %13\nv_cvt_pk_bf16_f32 %6, %6, %14\nv_cvt_pk_bf16_f32 %7, %7, %15\n")
BODY(k_max3, F8("v_max3_f32"))
// packed f32: operands are register PAIRS
__global__ void __launch_bounds__(1024) k_pkfma(unsigned long long* out, float seed, int iters) {
    typedef __attribute__((ext_vector_type(2))) float f2;
    f2 v[8], w[8];
    for (int i = 0; i < 8; ++i) { v[i] = f2{seed + threadIdx.x + i, seed - i}; w[i] = f2{0.5f, 0.25f}; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(v[j]) : "v"(w[j]));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    float s = 0; for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
    if (s == 12345.f) out[0] = 0;
}
template <typename K>
static void run(const char* name, K kern) {
    unsigned long long* d; hipMalloc(&d, 256 * 16 * 8);
    for (int waves : {4, 8, 16}) {               // per workgroup = 1, 2, 4 per SIMD (one workgroup per CU)
        const int iters = 64;
        hipLaunchKernelGGL(kern, dim3(256), dim3(waves * 64), 0, 0, d, 1.0f, iters);
        hipLaunchKernelGGL(kern, dim3(256), dim3(waves * 64), 0, 0, d, 1.0f, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(256 * 16); hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0; int n = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) { s += (double)h[b * 16 + w]; ++n; }
        const double per_wave_instr = s / n / (iters * 64.0);
        printf("%-10s %d wave(s)/SIMD: %6.2f cycles per instruction per wave  -> %5.2f cycles of SIMD time per wave instruction\n", name, waves / 4, per_wave_instr, per_wave_instr / (waves / 4));
    }
    hipFree(d);
}
int main() {
    run("v_mov_b32", k_mov); run("v_fma_f32", k_fma32); run("v_max3_f32", k_max3); run("v_pk_fma", k_pkfma); run("v_cvt_pk", k_cvtpk);
    run("v_exp_f32", k_exp32); run("v_rcp_f32", k_rcp32); run("v_rsq_f32", k_rsq32); run("v_exp_f16", k_exp16); run("v_rcp_f16", k_rcp16);
    return 0;
}
