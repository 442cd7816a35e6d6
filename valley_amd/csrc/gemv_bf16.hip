// Weight-streaming GEMV for decode: C[M<=8, N] = epi(A[M,K] · W[N,K]^T + bias) + residual.
// HBM-bound by construction: every wave owns two consecutive weight rows and streams them once with
// 16-byte loads (no LDS round trip — the operand is not shared between waves,
// cdna_hip_programming §5 "GEMV / M <= 16"); the few activation rows stay L1/L2 resident.
// Algorithmic bytes per launch = N*K*2 (weights) — activations and outputs are negligible.
#include "common.hpp"
#include "../../include/valley_hip.h"

namespace {

VLY_DEVICE float dot8(const u32x4& w, const u32x4& a) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s = fmaf(h_lo(w[i]), h_lo(a[i]), s);
        s = fmaf(h_hi(w[i]), h_hi(a[i]), s);
    }
    return s;
}

// KS = 1: every wave owns two weight rows (short rows).  KS = 4 (K >= 2048, i.e. every decode projection): the
// workgroup owns two rows and its four waves split K, partial sums meet in LDS — four times the waves, hence
// four times the loads in flight: o/down 4.0-4.5 -> 4.9-5.5 TB/s, q|k|v 5.9 -> 6.4, gate/up 6.5 -> 6.8, lm_head 6.9.
template <int MR, int EPI, int OUT, int KS>
__global__ void __launch_bounds__(256) gemv_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W,
                                                   const float* __restrict__ bias, const float* __restrict__ R,
                                                   void* __restrict__ Cv, int M, int N, int K, int lda, int ldw, int ldc, int ldr) {
    __shared__ float red[KS == 1 ? 1 : KS * 2 * MR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = (KS == 1 ? blockIdx.x * 4 + wave : blockIdx.x) * 2;
    if (n0 >= N) return;
    const uint16_t* w0 = W + (size_t)n0 * ldw;
    const uint16_t* w1 = W + (size_t)min(n0 + 1, N - 1) * ldw;
    float acc0[MR], acc1[MR];
#pragma unroll
    for (int m = 0; m < MR; ++m) { acc0[m] = 0.f; acc1[m] = 0.f; }
    const int nch = K >> 3;
#pragma unroll 4
    for (int c = (KS == 1 ? lane : wave * 64 + lane); c < nch; c += 64 * KS) {
        const u32x4 x0 = __builtin_nontemporal_load((const u32x4*)(w0 + 8 * c));
        const u32x4 x1 = __builtin_nontemporal_load((const u32x4*)(w1 + 8 * c));
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const u32x4 a = *(const u32x4*)(A + (size_t)min(m, M - 1) * lda + 8 * c);
            acc0[m] += dot8(x0, a);
            acc1[m] += dot8(x1, a);
        }
    }
#pragma unroll
    for (int m = 0; m < MR; ++m) { acc0[m] = wave_sum(acc0[m]); acc1[m] = wave_sum(acc1[m]); }
    if constexpr (KS > 1) {
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < MR; ++m) { red[(wave * MR + m) * 2] = acc0[m]; red[(wave * MR + m) * 2 + 1] = acc1[m]; }
        }
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int m = 0; m < MR; ++m) {                       // fixed order: wave 0 + 1 + 2 + ...
            float s0 = red[m * 2], s1 = red[m * 2 + 1];
#pragma unroll
            for (int w = 1; w < KS; ++w) { s0 += red[(w * MR + m) * 2]; s1 += red[(w * MR + m) * 2 + 1]; }
            acc0[m] = s0;
            acc1[m] = s1;
        }
    }
    if (lane != 0) return;
    const bool has1 = n0 + 1 < N;
    const float b0 = bias ? bias[n0] : 0.f, b1 = (bias && has1) ? bias[n0 + 1] : 0.f;
#pragma unroll
    for (int m = 0; m < MR; ++m) {
        if (m >= M) break;
        float v0 = acc0[m] + b0, v1 = acc1[m] + b1;
        if constexpr (EPI == VLY_EPI_QUICK_GELU) {
            v0 = x_sigmoid(v0, 1.702f);
            v1 = x_sigmoid(v1, 1.702f);
        }
        if constexpr (EPI == VLY_EPI_SWIGLU) {
            const float o = x_sigmoid(v0, 1.f) * v1;
            const size_t off = (size_t)m * ldc + (n0 >> 1);
            if constexpr (OUT == VLY_OUT_BF16) ((uint16_t*)Cv)[off] = f2h(o);
            else ((float*)Cv)[off] = o;
        } else {
            if (R) {
                v0 += R[(size_t)m * ldr + n0];
                if (has1) v1 += R[(size_t)m * ldr + n0 + 1];
            }
            const size_t off = (size_t)m * ldc + n0;
            if constexpr (OUT == VLY_OUT_BF16) {
                ((uint16_t*)Cv)[off] = f2h(v0);
                if (has1) ((uint16_t*)Cv)[off + 1] = f2h(v1);
            } else {
                ((float*)Cv)[off] = v0;
                if (has1) ((float*)Cv)[off + 1] = v1;
            }
        }
    }
}

template <int MR>
int launch_mr(const void* A, const void* W, const float* bias, const float* R, void* C, int M, int N, int K, int lda,
              int ldw, int ldc, int ldr, int epi, int out, hipStream_t st) {
    const bool split = K >= 2048;                            // long rows: the workgroup's four waves split K
    dim3 grid(split ? (N + 1) / 2 : (N + 7) / 8), block(256);
#define VLY_GEMV(E, O)                                                                                            \
    do {                                                                                                          \
        if (split) hipLaunchKernelGGL((gemv_kernel<MR, E, O, 4>), grid, block, 0, st, (const uint16_t*)A,         \
                                      (const uint16_t*)W, bias, R, C, M, N, K, lda, ldw, ldc, ldr);               \
        else hipLaunchKernelGGL((gemv_kernel<MR, E, O, 1>), grid, block, 0, st, (const uint16_t*)A,               \
                                (const uint16_t*)W, bias, R, C, M, N, K, lda, ldw, ldc, ldr);                     \
    } while (0)
    if (epi == VLY_EPI_NONE && out == VLY_OUT_BF16) VLY_GEMV(VLY_EPI_NONE, VLY_OUT_BF16);
    else if (epi == VLY_EPI_NONE && out == VLY_OUT_F32) VLY_GEMV(VLY_EPI_NONE, VLY_OUT_F32);
    else if (epi == VLY_EPI_QUICK_GELU && out == VLY_OUT_BF16) VLY_GEMV(VLY_EPI_QUICK_GELU, VLY_OUT_BF16);
    else if (epi == VLY_EPI_SWIGLU && out == VLY_OUT_BF16) VLY_GEMV(VLY_EPI_SWIGLU, VLY_OUT_BF16);
    else {
        vly_set_error("vly_gemv_bf16: unsupported epilogue/out_dtype combination (%d,%d)", epi, out);
        return -22;
    }
#undef VLY_GEMV
    return vly_check_launch("vly_gemv_bf16");
}

}  // namespace

extern "C" int vly_gemv_bf16(const void* A, const void* W, const float* bias, const float* residual, void* C, int M,
                             int N, int K, int lda, int ldw, int ldc, int ldr, int epilogue, int out_dtype, void* stream) {
    if (M <= 0 || M > 8 || N <= 0 || K <= 0 || K % 8 || lda % 8 || ldw % 8 || ldw <= 0 || ((uintptr_t)A & 15) || ((uintptr_t)W & 15) ||
        (epilogue == VLY_EPI_SWIGLU && (N % 2 || residual))) {
        vly_set_error("vly_gemv_bf16: unsupported shape/alignment M=%d N=%d K=%d lda=%d ldw=%d", M, N, K, lda, ldw);
        return -22;
    }
    hipStream_t st = (hipStream_t)stream;
    if (M == 1) return launch_mr<1>(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, st);
    if (M == 2) return launch_mr<2>(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, st);
    if (M <= 4) return launch_mr<4>(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, st);
    return launch_mr<8>(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, st);
}
