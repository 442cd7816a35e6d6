// Latency-optimised bf16 GEMM for a FEW rows (8 < M <= 256):  C[M,N] = epi(A[M,K] . W[N,K]^T + bias), bf16 out.
//
// Where it is used: the F-row remainder of the tall ViT GEMMs (ops.row_split: M = F*257 is cut at a multiple of 4096 rows
// so that the main launch is a whole number of workgroup rounds) — e.g. 128 x 4096 x 1024.  Such a problem is ~1 GFLOP:
// with the tile kernels it is ONE tile row whose K loop runs its 16-64 dependent global->LDS->MFMA steps at ~1 us each
// (14-17 us at K = 1024, 33 us at K = 4096, measured), i.e. pure latency.  Here the latency chain is cut instead:
//   * a workgroup owns one 32 x 32 block of C; its four waves split K four ways (no barrier inside the K loop) and
//     reduce their partial blocks through LDS at the end — 128 x 4096 gives 512 workgroups = 8 waves on every CU;
//   * no LDS staging: A is tiny (L2-resident), W is read exactly once per 32-row block of A, and every lane loads its
//     MFMA fragment (16 bytes, K-contiguous) straight from global memory, UNROLL K steps of 32 in flight per wave.
// Algorithmic work 2*M*N*K flop; the kernel is latency-, not throughput-bound by design (a few microseconds).
#include <cstdlib>
#include "common.hpp"
#include "../../include/valley_hip.h"

namespace {

constexpr int SK_UNROLL = 4;       // K steps (of 32) whose four fragment loads are issued before the first MFMA of the group

// NWV waves split K: 4 for the K = 1024 remainders, 16 (1024 threads) when K >= 2048 — the fc2 remainder 128 x 1024 x 4096 is
// only 128 workgroups, and with four waves each wave walks 32 dependent steps of ~1 us load latency
template <int EPI, int NWV>
__global__ void __launch_bounds__(NWV * 64) gemm_skinny_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W,
                                                               const float* __restrict__ bias, uint16_t* __restrict__ C, int M, int N,
                                                               int K, int lda, int ldw, int ldc) {
    __shared__ f32x4 red[NWV][4][64];                      // [wave][fragment][lane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
    const int kq = K / NWV, kbeg = wave * kq;              // this wave's share of K
    // fragment row pointers (rows past the edge re-read the last row; masked at the store)
    const uint16_t* a0 = A + (size_t)min(m0 + l15, M - 1) * lda + kbeg + g * 8;
    const uint16_t* a1 = A + (size_t)min(m0 + 16 + l15, M - 1) * lda + kbeg + g * 8;
    const uint16_t* w0 = W + (size_t)(n0 + l15) * ldw + kbeg + g * 8;
    const uint16_t* w1 = W + (size_t)(n0 + 16 + l15) * ldw + kbeg + g * 8;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Whole groups of SK_UNROLL steps (every shape the ViT remainders have: kq = 256): no guards, so the 16 loads of a group
    // leave back to back and the MFMAs count them down (round 4: with the guarded loop below hipcc peeled the wave-uniform
    // `if`s into paths that wait vmcnt(0) behind one to three loads — 17.8 us for 128 x 1024 x 4096).  With four waves
    // (256 threads, registers to spare) the NEXT group's loads leave before this group's MFMAs.
    const int groups = (kq % (32 * SK_UNROLL) == 0) ? kq / (32 * SK_UNROLL) : 0;
    if (groups > 0) {
        bf16x8 fa0[SK_UNROLL], fa1[SK_UNROLL], fw0[SK_UNROLL], fw1[SK_UNROLL];
        auto load_group = [&](bf16x8 (&xa0)[SK_UNROLL], bf16x8 (&xa1)[SK_UNROLL], bf16x8 (&xw0)[SK_UNROLL], bf16x8 (&xw1)[SK_UNROLL],
                              int k) {
#pragma unroll
            for (int u = 0; u < SK_UNROLL; ++u) {
                xw0[u] = *(const bf16x8*)(w0 + k + 32 * u);
                xw1[u] = *(const bf16x8*)(w1 + k + 32 * u);
                xa0[u] = *(const bf16x8*)(a0 + k + 32 * u);
                xa1[u] = *(const bf16x8*)(a1 + k + 32 * u);
            }
        };
        auto mfma_group = [&](const bf16x8 (&xa0)[SK_UNROLL], const bf16x8 (&xa1)[SK_UNROLL], const bf16x8 (&xw0)[SK_UNROLL],
                              const bf16x8 (&xw1)[SK_UNROLL]) {
#pragma unroll
            for (int u = 0; u < SK_UNROLL; ++u) {
                acc[0][0] = mfma16(xw0[u], xa0[u], acc[0][0]);
                acc[0][1] = mfma16(xw1[u], xa0[u], acc[0][1]);
                acc[1][0] = mfma16(xw0[u], xa1[u], acc[1][0]);
                acc[1][1] = mfma16(xw1[u], xa1[u], acc[1][1]);
            }
        };
        load_group(fa0, fa1, fw0, fw1, 0);
        if constexpr (NWV == 4) {
            bf16x8 ga0[SK_UNROLL], ga1[SK_UNROLL], gw0[SK_UNROLL], gw1[SK_UNROLL];
#pragma unroll 1
            for (int it = 0; it < groups; it += 2) {
                if (it + 1 < groups) load_group(ga0, ga1, gw0, gw1, (it + 1) * 32 * SK_UNROLL);
                mfma_group(fa0, fa1, fw0, fw1);
                if (it + 1 >= groups) break;
                if (it + 2 < groups) load_group(fa0, fa1, fw0, fw1, (it + 2) * 32 * SK_UNROLL);
                mfma_group(ga0, ga1, gw0, gw1);
            }
        } else {
#pragma unroll 1
            for (int it = 0; it < groups; ++it) {
                mfma_group(fa0, fa1, fw0, fw1);
                if (it + 1 < groups) load_group(fa0, fa1, fw0, fw1, (it + 1) * 32 * SK_UNROLL);
            }
        }
    } else
    for (int k = 0; k < kq; k += 32 * SK_UNROLL) {
        bf16x8 fa0[SK_UNROLL], fa1[SK_UNROLL], fw0[SK_UNROLL], fw1[SK_UNROLL];
#pragma unroll
        for (int u = 0; u < SK_UNROLL; ++u) {
            const int kk = min(k + 32 * u, kq - 32);       // kq % 32 == 0; a repeated last step is skipped below
            fa0[u] = *(const bf16x8*)(a0 + kk);
            fa1[u] = *(const bf16x8*)(a1 + kk);
            fw0[u] = *(const bf16x8*)(w0 + kk);
            fw1[u] = *(const bf16x8*)(w1 + kk);
        }
#pragma unroll
        for (int u = 0; u < SK_UNROLL; ++u) {
            if (k + 32 * u < kq) {                         // wave-uniform
                // W fragment first: the lane ends up with C[m = .. + l15][n = .. + 4g + r] (4 consecutive n)
                acc[0][0] = mfma16(fw0[u], fa0[u], acc[0][0]);
                acc[0][1] = mfma16(fw1[u], fa0[u], acc[0][1]);
                acc[1][0] = mfma16(fw0[u], fa1[u], acc[1][0]);
                acc[1][1] = mfma16(fw1[u], fa1[u], acc[1][1]);
            }
        }
    }
    // ---- K reduction through LDS: wave w < 4 sums fragment (i, j) = (w >> 1, w & 1) of all waves
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) red[wave][2 * i + j][lane] = acc[i][j];
    __syncthreads();
    if (wave >= 4) return;
    f32x4 v = red[0][wave][lane];
#pragma unroll
    for (int w = 1; w < NWV; ++w) v += red[w][wave][lane];
    const int m = m0 + (wave >> 1) * 16 + l15, n = n0 + (wave & 1) * 16 + 4 * g;
    if (m >= M) return;
    if (bias) v += *(const f32x4*)(bias + n);
    if constexpr (EPI == VLY_EPI_QUICK_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = x_sigmoid(v[r], 1.702f);
    }
    if constexpr (EPI == VLY_EPI_RELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
    }
    u32x2 pk;
    pk[0] = pack_h2(v[0], v[1]);
    pk[1] = pack_h2(v[2], v[3]);
    *(u32x2*)(C + (size_t)m * ldc + n) = pk;
}

}  // namespace

extern "C" int vly_gemm_skinny_bf16(const void* A, const void* W, const float* bias, void* C, int M, int N, int K, int lda, int ldw,
                                    int ldc, int epilogue, void* stream) {
    if (M <= 0 || M > 256 || N <= 0 || N % 32 || K < 128 || K % 128 || (K >= 2048 && K % 512) || lda % 8 || ldw % 8 || ldc % 4 || ((uintptr_t)A & 15) ||
        ((uintptr_t)W & 15) || ((uintptr_t)C & 7) || (bias && ((uintptr_t)bias & 15))) {
        vly_set_error("vly_gemm_skinny_bf16: unsupported shape/alignment M=%d N=%d K=%d lda=%d ldw=%d ldc=%d", M, N, K, lda, ldw, ldc);
        return -22;
    }
    static const bool narrow_only = getenv("VLY_SKINNY_WAVES") && atoi(getenv("VLY_SKINNY_WAVES")) == 4;   // A/B switch
    const bool wide = K >= 2048 && !narrow_only;            // 16 waves: kq = K / 16 is a multiple of 32 (K % 512 == 0)
    dim3 grid(N / 32, (M + 31) / 32), block(wide ? 1024 : 256);
    hipStream_t st = (hipStream_t)stream;
#define VLY_SKINNY(E)                                                                                                      \
    do {                                                                                                                   \
        if (wide)                                                                                                          \
            hipLaunchKernelGGL((gemm_skinny_kernel<E, 16>), grid, block, 0, st, (const uint16_t*)A, (const uint16_t*)W, bias,  \
                               (uint16_t*)C, M, N, K, lda, ldw, ldc);                                                      \
        else                                                                                                               \
            hipLaunchKernelGGL((gemm_skinny_kernel<E, 4>), grid, block, 0, st, (const uint16_t*)A, (const uint16_t*)W, bias,   \
                               (uint16_t*)C, M, N, K, lda, ldw, ldc);                                                      \
    } while (0)
    switch (epilogue) {
        case VLY_EPI_NONE: VLY_SKINNY(VLY_EPI_NONE); break;
        case VLY_EPI_QUICK_GELU: VLY_SKINNY(VLY_EPI_QUICK_GELU); break;
        case VLY_EPI_RELU: VLY_SKINNY(VLY_EPI_RELU); break;
        default: vly_set_error("vly_gemm_skinny_bf16: unsupported epilogue %d", epilogue); return -22;
    }
#undef VLY_SKINNY
    return vly_check_launch("vly_gemm_skinny_bf16");
}
