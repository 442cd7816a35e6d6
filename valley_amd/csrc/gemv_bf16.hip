// Weight-streaming GEMV for decode: C[M<=8, N] = epi(A[M,K] · W[N,K]^T + bias) + residual.
// HBM-bound by construction: every wave owns two consecutive weight rows and streams them once with
// 16-byte loads (no LDS round trip — the operand is not shared between waves,
// cdna_hip_programming §5 "GEMV / M <= 16"); the few activation rows stay L1/L2 resident.
// Algorithmic bytes per launch = N*K*2 (weights) — activations and outputs are negligible.
#include <cstdlib>
#include "common.hpp"
#include "../../include/valley_hip.h"

namespace {

VLY_DEVICE float dot8(const u32x4& w, const u32x4& a) { return vly_dot8(w, a); }      // common.hpp: one definition for both files

// KS = 1: every wave owns NR weight rows (short rows).  KS = 4 (K >= 2048, i.e. every decode projection): the
// workgroup owns NR rows and its four waves split K, partial sums meet in LDS — four times the waves, hence
// four times the loads in flight: o/down 4.0-4.5 -> 4.9-5.5 TB/s, q|k|v 5.9 -> 6.4, gate/up 6.5 -> 6.8, lm_head 6.9.
// NR = 2 or 4 rows per workgroup, chosen per shape by launch_mr for the BALANCE of the launch: a CU holds eight
// 4-wave workgroups of the NR = 2 kernel (<= 64 registers) — with N = 5120 (o / down of the 13B decoder) that is 2560
// workgroups on 2048 slots, a second round at a quarter of the chip (the 5.2 TB/s of those two GEMVs next to the 6.8
// of gate/up); NR = 4 makes it 1280 workgroups = exactly five per CU, all resident, every CU streaming to the end.
template <int MR, int EPI, int OUT, int KS, int NR>
__global__ void __launch_bounds__(256) gemv_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W,
                                                   const float* __restrict__ bias, const float* __restrict__ R,
                                                   void* __restrict__ Cv, int M, int N, int K, int lda, int ldw, int ldc, int ldr) {
    __shared__ float red[KS == 1 ? 1 : KS * NR * MR];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = (KS == 1 ? blockIdx.x * 4 + wave : blockIdx.x) * NR;
    if (n0 >= N) return;
    const uint16_t* w[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) w[r] = W + (size_t)min(n0 + r, N - 1) * ldw;
    float acc[NR][MR];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int m = 0; m < MR; ++m) acc[r][m] = 0.f;
    const int nch = K >> 3;
#pragma clang loop unroll_count(8 / NR)
    for (int c = (KS == 1 ? lane : wave * 64 + lane); c < nch; c += 64 * KS) {
        u32x4 x[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) x[r] = __builtin_nontemporal_load((const u32x4*)(w[r] + 8 * c));
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const u32x4 a = *(const u32x4*)(A + (size_t)min(m, M - 1) * lda + 8 * c);
#pragma unroll
            for (int r = 0; r < NR; ++r) acc[r][m] += dot8(x[r], a);
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int m = 0; m < MR; ++m) acc[r][m] = wave_sum(acc[r][m]);
    if constexpr (KS > 1) {
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < MR; ++m)
#pragma unroll
                for (int r = 0; r < NR; ++r) red[(wave * MR + m) * NR + r] = acc[r][m];
        }
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int m = 0; m < MR; ++m)                         // fixed order: wave 0 + 1 + 2 + ...
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                float s = red[m * NR + r];
#pragma unroll
                for (int wv = 1; wv < KS; ++wv) s += red[(wv * MR + m) * NR + r];
                acc[r][m] = s;
            }
    }
    if (lane != 0) return;
#pragma unroll
    for (int rp = 0; rp < NR; rp += 2) {                     // row pairs: (gate, up) under SwiGLU
        const int n = n0 + rp;
        if (n >= N) break;
        const bool has1 = n + 1 < N;
        const float b0 = bias ? bias[n] : 0.f, b1 = (bias && has1) ? bias[n + 1] : 0.f;
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            if (m >= M) break;
            float v0 = acc[rp][m] + b0, v1 = acc[rp + 1][m] + b1;
            if constexpr (EPI == VLY_EPI_QUICK_GELU) {
                v0 = x_sigmoid(v0, 1.702f);
                v1 = x_sigmoid(v1, 1.702f);
            }
            if constexpr (EPI == VLY_EPI_SWIGLU) {
                float o = x_sigmoid(v0, 1.f) * v1;
                // the product is an fp32 VALUE before it is stored: on the fp16 build hipcc otherwise folds multiply + conversion
                // into v_fma_mixlo_f16 (one rounding of the exact product), a last bit away from what a consumer of the fp32
                // value (vly_decode_layers' fp32 scratch) rounds to
                asm volatile("" : "+v"(o));
                const size_t off = (size_t)m * ldc + (n >> 1);
                if constexpr (OUT == VLY_OUT_BF16) ((uint16_t*)Cv)[off] = f2h(o);
                else ((float*)Cv)[off] = o;
            } else {
                if (R) {
                    v0 += R[(size_t)m * ldr + n];
                    if (has1) v1 += R[(size_t)m * ldr + n + 1];
                }
                const size_t off = (size_t)m * ldc + n;
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
}

int cu_count() {
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    return cus;
}

// fraction of the resident slots a launch of `wgs` equal workgroups keeps busy (whole rounds of `slots`)
float balance(int wgs, int slots) { return (float)wgs / (float)(((wgs + slots - 1) / slots) * slots); }

template <int MR>
int launch_mr(const void* A, const void* W, const float* bias, const float* R, void* C, int M, int N, int K, int lda,
              int ldw, int ldc, int ldr, int epi, int out, hipStream_t st) {
    const bool split = K >= 2048;                            // long rows: the workgroup's four waves split K
    // rows per workgroup (see gemv_kernel): the NR = 2 kernel is resident eight to a CU, NR = 4 five to a CU (<= 96 registers);
    // four rows only where they fill the rounds better (VLY_GEMV_ROWS=2|4 pins it for A/B runs)
    static const int pin = [] { const char* e = getenv("VLY_GEMV_ROWS"); return e ? atoi(e) : 0; }();
    bool four = false;
    if (split && MR <= 2 && N >= 8) {
        const int cus = cu_count();
        // (measured, profiles/history/r03/r03_gemv_rows_per_wg.jsonl: 5120 x 13824 25.3 -> 23.9 us; 5120 x 5120 10.8 -> 10.9: short rows
        // gain nothing, so K >= 8192 as well)
        four = pin ? pin == 4 : K >= 8192 && balance((N + 3) / 4, cus * 5) > balance((N + 1) / 2, cus * 8) + 0.02f;
    }
    dim3 grid(split ? (four ? (N + 3) / 4 : (N + 1) / 2) : (N + 7) / 8), block(256);
#define VLY_GEMV(E, O)                                                                                            \
    do {                                                                                                          \
        if constexpr (MR <= 2) {                                                                                  \
            if (four) {                                                                                           \
                hipLaunchKernelGGL((gemv_kernel<MR, E, O, 4, 4>), grid, block, 0, st, (const uint16_t*)A,         \
                                   (const uint16_t*)W, bias, R, C, M, N, K, lda, ldw, ldc, ldr);                  \
                break;                                                                                            \
            }                                                                                                     \
        }                                                                                                         \
        if (split) hipLaunchKernelGGL((gemv_kernel<MR, E, O, 4, 2>), grid, block, 0, st, (const uint16_t*)A,      \
                                      (const uint16_t*)W, bias, R, C, M, N, K, lda, ldw, ldc, ldr);               \
        else hipLaunchKernelGGL((gemv_kernel<MR, E, O, 1, 2>), grid, block, 0, st, (const uint16_t*)A,            \
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

// ---------------------------------------------------------------------------------------------
// 3 <= M <= 16 rows (round 5): the same weight stream on the MATRIX cores.  gemv_kernel's dot products cost 16 VALU operations per row
// and 16 bytes of weights — at eight rows (serving.ContinuousBatcher: eight live requests on one captured step) the step is VALU-
// bound at 2.3x its batch-1 time although it streams the same bytes.  One MFMA 16x16x32 takes 16 weight rows x 32 k (1 KB: one 16-byte
// load per lane, exactly the fragment the lane must supply) against 16 activation rows: 256 B/clk of weights per CU, far above what
// HBM delivers (~15 B/clk/CU), whatever M is.
//   * a workgroup owns 16 weight rows; its four waves take the 64-wide K pairs p = wave, wave + 4, ... (two MFMA steps = one whole
//     128-byte line per row and wave) and meet in LDS at the end, summed in the fixed order wave 0 + 1 + 2 + 3;
//   * two groups of two pairs in flight per wave (16 KB of weights requested ahead);
//   * weights AND activations come through buffer descriptors that end with the last valid row: rows past N / M read as zero without
//     a branch (no exec-masked loads: behind those hipcc waits vmcnt(0)), and the activation lanes past M cost no L1 traffic.
// A row's result depends on that row and the weights only (every output element is its own MFMA accumulation chain in k order):
// a request's tokens do not depend on what the other slots hold.
// Measured at M = 8 on the 13B projections (profiles/r05/r05_gemv_rows.txt): 3.7-4.1 TB/s of weights on the wide shapes against 2.4-2.65
// for the VALU kernel, 6.6-6.8 at M = 1.  What keeps THIS form from the M = 1 rate is the fragment's shape: a 16-lane group of one
// load touches SIXTEEN rows (64 accesses of 16 bytes per instruction, ~16 B/clk per CU — the same limit tools/probes/store_rate_cu.hip
// finds for the GEMM epilogue's stores); without the activation loads the stream runs 4.6 TB/s, and a non-temporal hint costs 8 %
// (each line is touched by two loads).  It is kept as the A/B form (VLY_GEMV_MFMA=1); the default is gemv_mfma2_kernel below.
template <int EPI, int OUT>
__global__ void __launch_bounds__(256) gemv_mfma_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W,
                                                        const float* __restrict__ bias, const float* __restrict__ R, void* __restrict__ Cv,
                                                        int M, int N, int K, int lda, int ldw, int ldc, int ldr) {
    __shared__ f32x4 red[3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int rows_w = min(16, N - n0);                               // valid weight rows of this block
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(W) + (size_t)n0 * ldw, 0,
                                                                         (uint32_t)(((size_t)(rows_w - 1) * ldw + K) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(A), 0, (uint32_t)(((size_t)(M - 1) * lda + K) * 2),
                                                                         0x00020000);
    // lane (l15, g): 16 bytes at [row l15][k + 8 g]; a row past the end is pushed out of the descriptor's range (-> zeros, no traffic)
    const uint32_t vw = l15 < rows_w ? (uint32_t)(l15 * ldw + g * 8) * 2u : 0x80000000u;
    const uint32_t va = l15 < M ? (uint32_t)(l15 * lda + g * 8) * 2u : 0x80000000u;
    auto ldw16 = [&](int k) { return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsW, vw, (uint32_t)k * 2u, 0)); };
    auto lda16 = [&](int k) { return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsA, va, (uint32_t)k * 2u, 0)); };
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int P = K >> 6;                                             // 64-wide pairs of MFMA steps
    struct Grp { bf16x8 w[4], a[4]; };                                // two pairs = four steps
    auto load_grp = [&](Grp& q, int p) {                              // pairs p and p + 4 (both inside K: the caller checks)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k = (p + 4 * u) << 6;
            q.w[2 * u] = ldw16(k);
            q.w[2 * u + 1] = ldw16(k + 32);
            q.a[2 * u] = lda16(k);
            q.a[2 * u + 1] = lda16(k + 32);
        }
    };
    auto mfma_grp = [&](const Grp& q) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = mfma16(q.w[u], q.a[u], acc);
    };
    int p = wave;
    if (p + 4 < P) {
        Grp q0, q1;
        load_grp(q0, p);
#pragma unroll 1
        for (;;) {
            const bool more1 = p + 12 < P;                            // a whole second group behind q0?
            if (more1) load_grp(q1, p + 8);
            mfma_grp(q0);
            p += 8;
            if (!more1) break;
            const bool more0 = p + 12 < P;
            if (more0) load_grp(q0, p + 8);
            mfma_grp(q1);
            p += 8;
            if (!more0) break;
        }
    }
    for (; p < P; p += 4) {                                           // the pairs that did not fill a group
        const int k = p << 6;
        const bf16x8 w0 = ldw16(k), w1 = ldw16(k + 32), a0 = lda16(k), a1 = lda16(k + 32);
        acc = mfma16(w0, a0, acc);
        acc = mfma16(w1, a1, acc);
    }
    if (wave) red[wave - 1][lane] = acc;
    __syncthreads();
    if (wave) return;
    acc = ((acc + red[0][lane]) + red[1][lane]) + red[2][lane];        // fixed order: wave 0 + 1 + 2 + 3
    // lane (m = l15, g) holds C[m][n .. n + 3], n = n0 + 4 g
    const int m = l15, n = n0 + 4 * g;
    if (m >= M || n >= N) return;
    f32x4 v = acc;
    if (bias) v += *(const f32x4*)(bias + n);
    if constexpr (EPI == VLY_EPI_QUICK_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = x_sigmoid(v[r], 1.702f);
    }
    if constexpr (EPI == VLY_EPI_SWIGLU) {
        float o0 = x_sigmoid(v[0], 1.f) * v[1], o1 = x_sigmoid(v[2], 1.f) * v[3];
        asm volatile("" : "+v"(o0), "+v"(o1));                       // fp32 VALUES before the conversion (see gemv_kernel)
        const size_t off = (size_t)m * ldc + (n >> 1);
        if constexpr (OUT == VLY_OUT_BF16) *(uint32_t*)((uint16_t*)Cv + off) = (uint32_t)f2h(o0) | ((uint32_t)f2h(o1) << 16);
        else *(float2*)((float*)Cv + off) = make_float2(o0, o1);
    } else {
        if (R) v += *(const f32x4*)(R + (size_t)m * ldr + n);
        const size_t off = (size_t)m * ldc + n;
        if constexpr (OUT == VLY_OUT_BF16) {
            u32x2 pk;
            pk[0] = (uint32_t)f2h(v[0]) | ((uint32_t)f2h(v[1]) << 16);
            pk[1] = (uint32_t)f2h(v[2]) | ((uint32_t)f2h(v[3]) << 16);
            *(u32x2*)((uint16_t*)Cv + off) = pk;
        } else {
            *(f32x4*)((float*)Cv + off) = v;
        }
    }
}

// ---- the same through an LDS ring (round 5, second form).  gemv_mfma_kernel loads every lane's MFMA fragment straight from global
// memory: a 16-lane group of such a load touches sixteen rows, and the CU's load path then moves ~16 B/clk (3.7-4.1 TB/s of weights
// chip-wide; tools/probes/store_rate_cu.hip measures the same cliff for stores).  Here the bytes travel as the M = 1 kernel's do —
// whole 128-byte lines, eight rows per instruction — by LDS-DMA into a per-wave ring, and the fragment shape is produced by the
// ds_read: every wave owns NSLOT slots of {16 weight rows x 128 B, 16 activation rows x 128 B} = one 64-wide K pair, keeps
// NSLOT - 1 pairs in flight (counted vmcnt: LDS-DMA retires in issue order), and needs no barrier — producer and consumer of a slot
// are the same wave.  Rows are stored with the chunk swizzle of the GEMM stages (chunk ^= row & 7, applied to the SOURCE address).
// Measured (profiles/r05/r05_gemv_rows_v3.txt, M = 8): q|k|v 5.5, gate|up 5.5, lm_head 5.7 TB/s; o 3.5 and down 4.2 (320 row blocks
// of 16: see HALF8); the 13B step of eight live requests 10.8 (VALU) -> 9.0 (fragments from global) -> 6.35 -> 6.25 ms (HALF8) =
// 1280 tokens/s, 6.0x the single request's 214; three and four requests 6.3 / 6.9 -> 5.6 / 5.7 ms.
// HALF8 (M <= 8, narrow N): a workgroup owns EIGHT weight rows and the MFMA's sixteen rows are those eight rows at two K positions
// (LDS rows 0-7: k in [128 p, 128 p + 64), rows 8-15: the same weight rows at [128 p + 64, 128 p + 128)), the sixteen activation
// columns likewise the eight activation rows at the two positions: D[n][m] (n, m < 8) and D[n + 8][m + 8] are the two halves of
// C[n][m]; the off-diagonal quadrants pair a K position with the other one and are dropped.  Half the MFMA work is wasted (it is
// 20x over-provisioned here) and the row blocks are twice as many: o / down of the 13B decoder are 640 workgroups instead of 320.
// Measured (M = 8): o 15.05 -> 14.4 us (3.48 -> 3.64 TB/s), down 34.4 -> 32.6 (4.11 -> 4.34); the 13B step of eight requests 6.35 ->
// 6.25 ms.  The narrow shapes stay well under the wide ones' 5.5 TB/s whatever the block shape and ring depth: they are 9-24 us
// kernels whose workgroups (two resident per CU) each pay their own first-byte latency and reduction tail.  Summation order differs
// from the sixteen-row form (two K positions accumulate separately), so batch invariance holds inside M <= 8 and inside M > 8.
// What the narrow shapes are NOT bound by (profiles/r05/r05_gemv_h8_variants.txt): 2 / 3 / 4 waves per workgroup with 3 / 4 / 6 slots (two to
// four workgroups per CU, 7.7-12.8 MB requested ahead chip-wide) all run o 14.3-14.8 and down 31.9-33.7 us.  What they ARE partly bound
// by (r05_gemv_a_witness.txt, timing witnesses with wrong results, since removed): without the activation fetches — every workgroup
// re-reads all of A[8][K] from L2, as many bytes as its weights — o runs 12.4 and down 27.8 us (4.2 / 5.1 TB/s; the wide shapes move
// < 3 %).  Sharing A between more weight rows needs a K split across workgroups and a fix-up pass whose tail costs what it saves on
// 14 us kernels.
#ifndef VLY_GEMV_H8_NW
#define VLY_GEMV_H8_NW 4          // waves per eight-row workgroup and slots per wave (A/B: tools/ab_lib.py build x --src gemv_bf16.hip -D...)
#endif
#ifndef VLY_GEMV_H8_NSLOT
#define VLY_GEMV_H8_NSLOT 4
#endif
template <int EPI, int OUT, int NW, bool BIGM, bool HALF8 = false>       // BIGM: 9 .. 16 activation rows (a second activation DMA per pair, 4 KB slots)
__global__ void __launch_bounds__(NW * 64) gemv_mfma2_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W,
                                                             const float* __restrict__ bias, const float* __restrict__ R,
                                                             void* __restrict__ Cv, int M, int N, int K, int lda, int ldw, int ldc, int ldr) {
    static_assert(!(BIGM && HALF8), "HALF8 is the M <= 8 form");
    constexpr bool FOUR_K = BIGM || HALF8;                            // 4 KB slots: 16 LDS rows of weights + 16 of activations
    constexpr int NSLOT = HALF8 ? VLY_GEMV_H8_NSLOT : FOUR_K ? 4 : 5, SLOT = FOUR_K ? 4096 : 3072;  // 64 / 60 KB of ring per 4-wave workgroup: two workgroups per CU
    // (HALF8 with 3 / 5 slots: 14.5 / 16.8 us against 14.4 on the 13B o projection — the ring's depth is not what bounds the narrow shapes)
    constexpr int KSH = HALF8 ? 7 : 6;                                // a slot covers 128 (HALF8) or 64 k
    __shared__ __attribute__((aligned(16))) char ring[NW][NSLOT][SLOT];
    __shared__ f32x4 red[HALF8 ? NW : NW - 1][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l15 = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * (HALF8 ? 8 : 16);
    // DMA source of this lane: row (lane >> 3) of an 8-row half, 16-byte chunk ((lane & 7) ^ (row & 7)); rows past the end re-read the last
    const int r8 = lane >> 3, ch = (lane & 7) ^ (r8 & 7);            // (row + 8 has the same low three bits)
    const uint16_t* w_lo = W + (size_t)min(n0 + r8, N - 1) * ldw + ch * 8;
    const uint16_t* w_hi = HALF8 ? w_lo + 64 : W + (size_t)min(n0 + 8 + r8, N - 1) * ldw + ch * 8;
    const uint16_t* a_lo = A + (size_t)min(r8, M - 1) * lda + ch * 8;
    [[maybe_unused]] const uint16_t* a_hi = HALF8 ? a_lo + 64 : A + (size_t)min(8 + r8, M - 1) * lda + ch * 8;
    constexpr bool two_a = FOUR_K;
    char* my = &ring[wave][0][0];
    // (asm, not the builtin: hipcc knows that the builtin writes LDS and waits vmcnt(0) in front of every ds_read that follows — the
    // ring would hold one pair in flight; M0 = the LDS address of the instruction's 1 KB, the hardware adds lane * 16)
    const uint32_t my_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)my);
    auto dma = [&](const uint16_t* src, uint32_t dst) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst) : "memory", "m0");
    };
    auto issue = [&](int p, int slot) {                              // pair p -> slot: 3 or 4 LDS-DMA instructions of 1 KB
        const uint32_t d = my_lds + (uint32_t)slot * SLOT;
        const int k = p << KSH;
        dma(w_lo + k, d);
        dma(w_hi + k, d + 1024);
        dma(a_lo + k, d + 2048);
        if (two_a) dma(a_hi + k, d + 3072);
    };
    // fragment reads: row l15, chunk (4 step + g) ^ (l15 & 7)
    const int rd0 = l15 * 128 + (((0 + g) ^ (l15 & 7)) << 4), rd1 = l15 * 128 + (((4 + g) ^ (l15 & 7)) << 4);
    // the activation piece holds 8 rows unless FOUR_K stages two: columns m >= 8 of the MFMA read row m & 7 again (their results are
    // dropped below) instead of running 1 KB past the slot
    const int ra0 = two_a ? rd0 : (l15 & 7) * 128 + (((0 + g) ^ (l15 & 7)) << 4), ra1 = two_a ? rd1 : (l15 & 7) * 128 + (((4 + g) ^ (l15 & 7)) << 4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int P = K >> KSH;
    const int n = (P - wave + NW - 1) / NW;                          // pairs of this wave: p = wave + NW i
#pragma unroll
    for (int i = 0; i < NSLOT - 1; ++i)
        if (i < n) issue(wave + NW * i, i);
    int slot = 0;
    for (int i = 0; i < n; ++i) {
        const int ahead = i + NSLOT - 1;
        if (ahead < n) {
            // the slot being refilled was read in the PREVIOUS iteration: its ds_reads must have returned before the DMA may land there
            // (only the order of issue separated them — hipcc is free to sink the MFMAs and their lgkmcnt wait below this asm)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue(wave + NW * ahead, (slot + NSLOT - 1) % NSLOT);
            // pair i has landed when at most the (NSLOT - 1) younger pairs' instructions are outstanding
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"((NSLOT - 1) * (two_a ? 4 : 3)) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the tail: nothing more to request
        }
        const char* s = my + slot * SLOT;
        const bf16x8 w0 = *(const bf16x8*)(s + rd0), w1 = *(const bf16x8*)(s + rd1);
        const bf16x8 a0 = *(const bf16x8*)(s + 2048 + ra0), a1 = *(const bf16x8*)(s + 2048 + ra1);
        acc = mfma16(w0, a0, acc);
        acc = mfma16(w1, a1, acc);
        slot = (slot + 1) % NSLOT;
    }
    if constexpr (HALF8) {
        // every wave publishes; lane (m < 8, g < 2) of wave 0 sums, wave by wave, the quadrant D[n][m] and its partner D[n + 8][m + 8]
        // (lane + 40: column m + 8, row group g + 2) — fixed order
        red[wave][lane] = acc;
        __syncthreads();
        if (wave || l15 >= 8 || g >= 2) return;
        acc = red[0][lane] + red[0][lane + 40];
#pragma unroll
        for (int wv = 1; wv < NW; ++wv) acc += red[wv][lane] + red[wv][lane + 40];
    } else {
        if (wave) red[wave - 1][lane] = acc;
        __syncthreads();
        if (wave) return;
#pragma unroll
        for (int wv = 1; wv < NW; ++wv) acc += red[wv - 1][lane];    // fixed order: wave 0 + 1 + 2 + ...
    }
    const int m = l15, nn = n0 + 4 * g;
    if (m >= M || nn >= N) return;
    f32x4 v = acc;
    if (bias) v += *(const f32x4*)(bias + nn);
    if constexpr (EPI == VLY_EPI_QUICK_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = x_sigmoid(v[r], 1.702f);
    }
    if constexpr (EPI == VLY_EPI_SWIGLU) {
        float o0 = x_sigmoid(v[0], 1.f) * v[1], o1 = x_sigmoid(v[2], 1.f) * v[3];
        asm volatile("" : "+v"(o0), "+v"(o1));
        const size_t off = (size_t)m * ldc + (nn >> 1);
        if constexpr (OUT == VLY_OUT_BF16) *(uint32_t*)((uint16_t*)Cv + off) = (uint32_t)f2h(o0) | ((uint32_t)f2h(o1) << 16);
        else *(float2*)((float*)Cv + off) = make_float2(o0, o1);
    } else {
        if (R) v += *(const f32x4*)(R + (size_t)m * ldr + nn);
        const size_t off = (size_t)m * ldc + nn;
        if constexpr (OUT == VLY_OUT_BF16) {
            u32x2 pk;
            pk[0] = (uint32_t)f2h(v[0]) | ((uint32_t)f2h(v[1]) << 16);
            pk[1] = (uint32_t)f2h(v[2]) | ((uint32_t)f2h(v[3]) << 16);
            *(u32x2*)((uint16_t*)Cv + off) = pk;
        } else {
            *(f32x4*)((float*)Cv + off) = v;
        }
    }
}

int launch_mfma_rows(const void* A, const void* W, const float* bias, const float* R, void* C, int M, int N, int K, int lda, int ldw, int ldc,
                     int ldr, int epi, int out, hipStream_t st) {
    // VLY_GEMV_MFMA: 2 (default) = the LDS-ring form, 1 = fragments straight from global memory (A/B runs)
    static const int form = [] { const char* e = getenv("VLY_GEMV_MFMA"); return e ? atoi(e) : 2; }();
    if (form != 1) {
        // eight-row blocks (HALF8) where sixteen-row blocks leave the CUs unevenly loaded: fewer than three blocks per CU, M <= 8
        static const int half_pin = [] { const char* e = getenv("VLY_GEMV_HALF8"); return e ? atoi(e) : -1; }();     // (A/B runs)
        const bool half8 = M <= 8 && K % 128 == 0 && (half_pin >= 0 ? half_pin != 0 : (N + 15) / 16 < 3 * 256);     // (by SHAPE only, not by the device's CU count:
                                                                                                                   // the same request sums in the same order on every partition of the GPU — ADVICE r5)
        dim3 g2(half8 ? (N + 7) / 8 : (N + 15) / 16), b2(256);
        // (eight waves per workgroup — one workgroup per CU — measured 3-5 % behind four on the narrow shapes: r05_gemv_rows_v2.txt)
#define VLY_GEMV_M2(E, O)                                                                                                              \
        do {                                                                                                                           \
            if (half8) hipLaunchKernelGGL((gemv_mfma2_kernel<E, O, VLY_GEMV_H8_NW, false, true>), g2, dim3(VLY_GEMV_H8_NW * 64), 0, st, (const uint16_t*)A, (const uint16_t*)W, bias, R, C, M, N, \
                                          K, lda, ldw, ldc, ldr);                                                                      \
            else if (M > 8) hipLaunchKernelGGL((gemv_mfma2_kernel<E, O, 4, true>), g2, b2, 0, st, (const uint16_t*)A, (const uint16_t*)W, bias, R, C, M, N, K, \
                                          lda, ldw, ldc, ldr);                                                                         \
            else hipLaunchKernelGGL((gemv_mfma2_kernel<E, O, 4, false>), g2, b2, 0, st, (const uint16_t*)A, (const uint16_t*)W, bias, R, C, M, N, K, lda, \
                                    ldw, ldc, ldr);                                                                                    \
        } while (0)
        if (epi == VLY_EPI_NONE && out == VLY_OUT_BF16) VLY_GEMV_M2(VLY_EPI_NONE, VLY_OUT_BF16);
        else if (epi == VLY_EPI_NONE && out == VLY_OUT_F32) VLY_GEMV_M2(VLY_EPI_NONE, VLY_OUT_F32);
        else if (epi == VLY_EPI_QUICK_GELU && out == VLY_OUT_BF16) VLY_GEMV_M2(VLY_EPI_QUICK_GELU, VLY_OUT_BF16);
        else if (epi == VLY_EPI_SWIGLU && out == VLY_OUT_BF16) VLY_GEMV_M2(VLY_EPI_SWIGLU, VLY_OUT_BF16);
        else {
            vly_set_error("vly_gemv_bf16: unsupported epilogue/out_dtype combination (%d,%d)", epi, out);
            return -22;
        }
#undef VLY_GEMV_M2
        return vly_check_launch("vly_gemv_bf16");
    }
    dim3 grid((N + 15) / 16), block(256);
#define VLY_GEMV_M(E, O)                                                                                                          \
    hipLaunchKernelGGL((gemv_mfma_kernel<E, O>), grid, block, 0, st, (const uint16_t*)A, (const uint16_t*)W, bias, R, C, M, N, K, lda, ldw, \
                       ldc, ldr)
    if (epi == VLY_EPI_NONE && out == VLY_OUT_BF16) VLY_GEMV_M(VLY_EPI_NONE, VLY_OUT_BF16);
    else if (epi == VLY_EPI_NONE && out == VLY_OUT_F32) VLY_GEMV_M(VLY_EPI_NONE, VLY_OUT_F32);
    else if (epi == VLY_EPI_QUICK_GELU && out == VLY_OUT_BF16) VLY_GEMV_M(VLY_EPI_QUICK_GELU, VLY_OUT_BF16);
    else if (epi == VLY_EPI_SWIGLU && out == VLY_OUT_BF16) VLY_GEMV_M(VLY_EPI_SWIGLU, VLY_OUT_BF16);
    else {
        vly_set_error("vly_gemv_bf16: unsupported epilogue/out_dtype combination (%d,%d)", epi, out);
        return -22;
    }
#undef VLY_GEMV_M
    return vly_check_launch("vly_gemv_bf16");
}

// ---------------------------------------------------------------------------------------------
// RMSNorm folded into the GEMV that consumes it (decode: input_layernorm -> q|k|v, post_attention_layernorm -> gate|up,
// norm -> lm_head): C = epi(rmsnorm(H; gamma, eps) · W^T + bias) + residual, H the fp32 residual stream.
// A batch-1 decode layer is short weight-streaming kernels back to back; a norm launch between two of them costs its body
// plus a kernel boundary (MI355X_MICROARCH "boundary" / "launches-baseline") and moves 30 KB.  Here the norm is computed once
// per CU — norm_row_kernel's arithmetic operation for operation, so x and everything downstream are BIT-IDENTICAL to
// vly_rmsnorm + vly_gemv_bf16 — into LDS, and the GEMV loop reads x from LDS instead of re-fetching it through the vector
// cache beside the weight stream.  Measured (13B decode, 256 tokens, same box, interleaved): 201.5 -> 207.0 tokens/s; three
// structures of the prologue (five 256-thread workgroups per CU each normalising for itself, with and without a second
// pair in flight, and this one) all land within 0.3 % of each other — profiles/history/r03/r03_decode_fuse_norm.txt.
// ---------------------------------------------------------------------------------------------
template <int CH>
struct PairRegs {
    u32x4 x0[CH], x1[CH];
};

template <int MR, int EPI, int OUT, int CH, int PRO>
__global__ void __launch_bounds__(1024) gemv_norm_kernel(const float* __restrict__ H, const float* __restrict__ gamma, float eps,
                                                         const uint16_t* __restrict__ W, const float* __restrict__ bias,
                                                         const float* __restrict__ R, void* __restrict__ Cv, int M, int N, int K,
                                                         int ldh, int ldw, int ldc, int ldr) {
    // ONE 16-wave workgroup per CU = four 256-thread GROUPS, each of which is a gemv_kernel<.., 4, 2> workgroup (same chunk
    // order per thread, same wave sums, same fixed-order sum over its four waves) walking its own row pairs; the norm is
    // computed ONCE per CU, by group 0 with norm_row_kernel's arithmetic, into LDS (256 x 40 KB of H and gamma leave L2 per
    // launch instead of 1280 x 40 KB).
    extern __shared__ __attribute__((aligned(16))) char gn_dyn[];
    uint16_t* xs = (uint16_t*)gn_dyn;                                   // [MR][K]
    __shared__ float red[2][4][4 * 2 * MR];
    __shared__ float nred[4];
    const int tid = threadIdx.x & 255, grp = threadIdx.x >> 8, lane = tid & 63, wave = tid >> 6;
    const int nvec = K >> 2, nch = K >> 3;
    // A row pair is CH 16-byte chunks per thread and row (CH = ceil(K / 2048)), all issued at once.  Two pairs are in
    // flight per group: pair i + 1's loads leave before pair i is reduced, and the FIRST pair's loads leave before the norm
    // prologue, so the weight stream starts with the kernel.
    auto load_pair = [&](PairRegs<CH>& p, int n0) {
        // UNCONDITIONAL, branch-free loads: behind exec-masked branches hipcc loses count of what is in flight and waits
        // vmcnt(0) — for the prefetched pair too.  A ragged last chunk, and the pair past the end a group "prefetches"
        // in its last trip, read W[0..7] (one cache line for the whole group) and are never accumulated.
        const bool live = n0 < N;
        const uint16_t* w0 = W + (live ? (size_t)n0 * ldw : 0);
        const uint16_t* w1 = W + (live ? (size_t)min(n0 + 1, N - 1) * ldw : 0);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int c = tid + 256 * i;
            const int off = (live && c < nch) ? 8 * c : 0;
            p.x0[i] = __builtin_nontemporal_load((const u32x4*)(w0 + off));
            p.x1[i] = __builtin_nontemporal_load((const u32x4*)(w1 + off));
        }
    };
    auto reduce_pair = [&](const PairRegs<CH>& p, int n0, float* rd) {      // (every group, every trip: the barrier is the workgroup's)
        float acc0[MR], acc1[MR];
#pragma unroll
        for (int m = 0; m < MR; ++m) { acc0[m] = 0.f; acc1[m] = 0.f; }
#pragma unroll
        for (int i = 0; i < CH; ++i) {                       // chunk order per thread as gemv_kernel<.., 4, 2>: tid, tid + 256, ...
            const int c = tid + 256 * i;
            if (c < nch) {
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    const u32x4 a = *(const u32x4*)(xs + (size_t)m * K + 8 * c);
                    acc0[m] += dot8(p.x0[i], a);
                    acc1[m] += dot8(p.x1[i], a);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < MR; ++m) { acc0[m] = wave_sum(acc0[m]); acc1[m] = wave_sum(acc1[m]); }
        if (lane == 0) {
#pragma unroll
            for (int m = 0; m < MR; ++m) { rd[(wave * MR + m) * 2] = acc0[m]; rd[(wave * MR + m) * 2 + 1] = acc1[m]; }
        }
        __syncthreads();                                     // (one barrier per pair: the two halves of red alternate)
        if (tid != 0 || n0 >= N) return;
        const bool has1 = n0 + 1 < N;
        const float b0 = bias ? bias[n0] : 0.f, b1 = (bias && has1) ? bias[n0 + 1] : 0.f;
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            if (m >= M) break;
            float s0 = rd[m * 2], s1 = rd[m * 2 + 1];        // fixed order: wave 0 + 1 + 2 + 3 (gemv_kernel's)
#pragma unroll
            for (int wv = 1; wv < 4; ++wv) { s0 += rd[(wv * MR + m) * 2]; s1 += rd[(wv * MR + m) * 2 + 1]; }
            float v0 = s0 + b0, v1 = s1 + b1;
            if constexpr (EPI == VLY_EPI_SWIGLU) {
                float o = x_sigmoid(v0, 1.f) * v1;
                // the product is an fp32 VALUE before it is stored: on the fp16 build hipcc otherwise folds multiply + conversion
                // into v_fma_mixlo_f16 (one rounding of the exact product), a last bit away from what a consumer of the fp32
                // value (vly_decode_layers' fp32 scratch) rounds to
                asm volatile("" : "+v"(o));
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
    };
    PairRegs<CH> pa, pb;
    // pair p = cu + CUs * (grp + 4 t): a CU's four groups share its pairs round-robin, so every CU streams the same number of
    // rows (round 3 gave group (cu, grp) the pairs 4 cu + grp + 4 CUs t: with 2560 pairs — the 13B o projection — the groups of
    // CUs 0..127 made three trips, those of CUs 128..255 two, and half the chip sat out the last third of the launch)
    const int first = (grp * (int)gridDim.x + (int)blockIdx.x) * 2, stride = gridDim.x * 4 * 2;
    if constexpr (PRO == 1) {
        // x = the merge of vly_decode_attention_split's partials (H = partials, ldh = heads): all 1024 threads, one or two
        // 4-wide units each — unit u is dims 4 (u & 31) .. + 3 of head u >> 5 — loads first, then the weights
        const int heads = ldh, gt = threadIdx.x;
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            const float* pb_ = H + (size_t)min(m, M - 1) * heads * (VLY_DECODE_SPLITS * 132);
            float ms[2][VLY_DECODE_SPLITS], ls[2][VLY_DECODE_SPLITS];
            float4 os[2][VLY_DECODE_SPLITS];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int u = min(gt + 1024 * i, nvec - 1);
                const float* hb = pb_ + (size_t)(u >> 5) * (VLY_DECODE_SPLITS * 132);
#pragma unroll
                for (int sp = 0; sp < VLY_DECODE_SPLITS; ++sp) {
                    ms[i][sp] = hb[sp * 132];
                    ls[i][sp] = hb[sp * 132 + 1];
                    os[i][sp] = *(const float4*)(hb + sp * 132 + 4 + 4 * (u & 31));
                }
            }
            if (m == 0) load_pair(pa, first);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int u = gt + 1024 * i;
                float mx = ms[i][0];
#pragma unroll
                for (int sp = 1; sp < VLY_DECODE_SPLITS; ++sp) mx = fmaxf(mx, ms[i][sp]);
                float L = 0.f;
                float4 O = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int sp = 0; sp < VLY_DECODE_SPLITS; ++sp) {            // split order: deterministic
                    const float w = exp2f(ms[i][sp] - mx);
                    L = fmaf(ls[i][sp], w, L);
                    O.x = fmaf(os[i][sp].x, w, O.x); O.y = fmaf(os[i][sp].y, w, O.y);
                    O.z = fmaf(os[i][sp].z, w, O.z); O.w = fmaf(os[i][sp].w, w, O.w);
                }
                if (u < nvec) {
                    u32x2 pk;
                    pk[0] = pack_h2(O.x / L, O.y / L);
                    pk[1] = pack_h2(O.z / L, O.w / L);
                    *(u32x2*)(xs + (size_t)m * K + 4 * u) = pk;
                }
            }
        }
        __syncthreads();
    } else if (grp == 0) {
        // issue order: row 0 of H and gamma FIRST, then the weights — vmcnt retires in order, so the norm below waits for its
        // own operands only and runs under the weights' HBM latency
        float4 v[2 * CH], gm[2 * CH];                                    // K <= 2048 CH: 2 CH float4 per thread cover a row
        auto load_h = [&](int m) {
            const float4* hr = (const float4*)(H + (size_t)min(m, M - 1) * ldh);
#pragma unroll
            for (int i = 0; i < 2 * CH; ++i) {
                const int c = tid + 256 * i;
                const float4 t = hr[min(c, nvec - 1)];
                v[i] = (c < nvec) ? t : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        };
        load_h(0);
#pragma unroll
        for (int i = 0; i < 2 * CH; ++i) gm[i] = ((const float4*)gamma)[min(tid + 256 * i, nvec - 1)];
        load_pair(pa, first);
#pragma unroll
        for (int m = 0; m < MR; ++m) {
            if (m > 0) load_h(m);
            float s = 0.f;                                               // norm_row_kernel's arithmetic, operation for operation
#pragma unroll
            for (int i = 0; i < 2 * CH; ++i) s += vly_sumsq4(v[i].x, v[i].y, v[i].z, v[i].w);
            s = wave_sum(s);
            if (lane == 0) nred[wave] = s;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // group 0's four waves only meet here: see below
            s = nred[0] + nred[1] + nred[2] + nred[3];
            const float rstd = rsqrtf(s / (float)K + eps);
#pragma unroll
            for (int i = 0; i < 2 * CH; ++i) {
                const int c = tid + 256 * i;
                if (c >= nvec) continue;
                float4 o;
                o.x = gm[i].x * (v[i].x * rstd); o.y = gm[i].y * (v[i].y * rstd);
                o.z = gm[i].z * (v[i].z * rstd); o.w = gm[i].w * (v[i].w * rstd);
                u32x2 pk;
                pk[0] = pack_h2(o.x, o.y);
                pk[1] = pack_h2(o.z, o.w);
                *(u32x2*)(xs + (size_t)m * K + 4 * c) = pk;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    } else {
        load_pair(pa, first);
#pragma unroll
        for (int m = 0; m < MR; ++m) {                                   // the other groups arrive at the same 2 MR barriers
            asm volatile("s_barrier" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
        }
    }
    // trips are counted for the workgroup (group 0 owns the lowest pair index, so its count is the largest): every group
    // passes every barrier, pairs past N are loaded from W[0..7] and dropped
#pragma unroll 1
    for (int n0 = first, nb = (int)blockIdx.x * 2; nb < N; n0 += 2 * stride, nb += 2 * stride) {
        const int n1 = n0 + stride;
        load_pair(pb, n1);
        reduce_pair(pa, n0, red[0][grp]);
        if (nb + stride >= N) break;
        load_pair(pa, n1 + stride);
        reduce_pair(pb, n1, red[1][grp]);
    }
}

template <int MR, int PRO>
int launch_norm_mr(const float* H, const float* gamma, float eps, const void* W, const float* bias, const float* R, void* C, int M,
                   int N, int K, int ldh, int ldw, int ldc, int ldr, int epi, int out, hipStream_t st, const char* name) {
    const size_t lds = (size_t)MR * K * 2;
    const int pairs = (N + 1) / 2, groups = (pairs + 3) / 4;
    dim3 grid(groups < cu_count() ? groups : cu_count()), block(1024);       // one 16-wave workgroup per CU
#define VLY_GEMVN_CH(E, O, CH)                                                                                     \
    hipLaunchKernelGGL((gemv_norm_kernel<MR, E, O, CH, PRO>), grid, block, lds, st, H, gamma, eps, (const uint16_t*)W, bias, R, C, \
                       M, N, K, ldh, ldw, ldc, ldr)
#define VLY_GEMVN(E, O)                                                                                           \
    do {                                                                                                          \
        if (K <= 4096) VLY_GEMVN_CH(E, O, 2);                                                                     \
        else VLY_GEMVN_CH(E, O, 3);                                                                               \
    } while (0)
    if (epi == VLY_EPI_NONE && out == VLY_OUT_BF16) VLY_GEMVN(VLY_EPI_NONE, VLY_OUT_BF16);
    else if (epi == VLY_EPI_NONE && out == VLY_OUT_F32) VLY_GEMVN(VLY_EPI_NONE, VLY_OUT_F32);
    else if (PRO == 0 && epi == VLY_EPI_SWIGLU && out == VLY_OUT_BF16) {
        if constexpr (PRO == 0) VLY_GEMVN(VLY_EPI_SWIGLU, VLY_OUT_BF16);
    } else {
        vly_set_error("%s: unsupported epilogue/out_dtype combination (%d,%d)", name, epi, out);
        return -22;
    }
#undef VLY_GEMVN_CH
#undef VLY_GEMVN
    return vly_check_launch(name);
}

}  // namespace

extern "C" int vly_gemv_bf16(const void* A, const void* W, const float* bias, const float* residual, void* C, int M,
                             int N, int K, int lda, int ldw, int ldc, int ldr, int epilogue, int out_dtype, void* stream) {
    if (M <= 0 || M > 16 || N <= 0 || K <= 0 || K % 8 || lda % 8 || ldw % 8 || ldw <= 0 || ((uintptr_t)A & 15) || ((uintptr_t)W & 15) ||
        (epilogue == VLY_EPI_SWIGLU && (N % 2 || residual))) {
        vly_set_error("vly_gemv_bf16: unsupported shape/alignment M=%d N=%d K=%d lda=%d ldw=%d (M <= 16)", M, N, K, lda, ldw);
        return -22;
    }
    hipStream_t st = (hipStream_t)stream;
    // three rows and more: the matrix-core form (gemv_mfma_kernel), when its 16-byte / 8-byte vector accesses line up; VLY_GEMV_MFMA=0
    // keeps the VALU kernels (A/B runs; M <= 8 only)
    static const bool no_mfma = getenv("VLY_GEMV_MFMA") && atoi(getenv("VLY_GEMV_MFMA")) == 0;
    const int No = epilogue == VLY_EPI_SWIGLU ? N / 2 : N;
    const bool mfma_ok = K % 64 == 0 && N % 4 == 0 && ldc % 4 == 0 && (size_t)M * lda * 2 < (1ull << 31) && (size_t)16 * ldw * 2 < (1ull << 31) &&
                         ((uintptr_t)C & (out_dtype == VLY_OUT_F32 ? 15 : (epilogue == VLY_EPI_SWIGLU ? 3 : 7))) == 0 &&
                         (!bias || ((uintptr_t)bias & 15) == 0) && (!residual || (ldr % 4 == 0 && ((uintptr_t)residual & 15) == 0)) &&
                         (epilogue != VLY_EPI_SWIGLU || (out_dtype == VLY_OUT_BF16 ? ldc % 2 == 0 : true)) && No > 0;
    static const int min_rows = [] { const char* e = getenv("VLY_GEMV_MFMA_MIN_ROWS"); return e ? atoi(e) : 3; }();      // (A/B runs)
    if (M >= min_rows && M >= 3 && mfma_ok && !(no_mfma && M <= 8))
        return launch_mfma_rows(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, st);
    if (M > 8) {
        vly_set_error("vly_gemv_bf16: 9 <= M <= 16 needs K %% 64 == 0, N %% 4 == 0 and 16-byte aligned rows (M=%d N=%d K=%d ldc=%d)", M, N, K, ldc);
        return -22;
    }
    if (M == 1) return launch_mr<1>(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, st);
    if (M == 2) return launch_mr<2>(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, st);
    if (M <= 4) return launch_mr<4>(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, st);
    return launch_mr<8>(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, st);
}

extern "C" int vly_gemv_rmsnorm_bf16(const float* H, const float* gamma, float eps, const void* W, const float* bias,
                                     const float* residual, void* C, int M, int N, int K, int ldh, int ldw, int ldc, int ldr,
                                     int epilogue, int out_dtype, void* stream) {
    if (M <= 0 || M > 2 || N <= 0 || K < 2048 || K > 6144 || K % 8 || ldh % 4 || ldw % 8 || ldw <= 0 || ((uintptr_t)H & 15) ||
        ((uintptr_t)gamma & 15) || ((uintptr_t)W & 15) || (epilogue == VLY_EPI_SWIGLU && (N % 2 || residual))) {
        vly_set_error("vly_gemv_rmsnorm_bf16: unsupported shape/alignment M=%d N=%d K=%d ldh=%d ldw=%d (M <= 2, 2048 <= K <= 6144)", M, N, K,
                      ldh, ldw);
        return -22;
    }
    {   // no aliasing of the output with H: every workgroup re-reads the whole H row for its norm while others write C
        const char *h0 = (const char*)H, *h1 = h0 + ((size_t)(M - 1) * ldh + K) * 4;
        const int No = epilogue == VLY_EPI_SWIGLU ? N / 2 : N;
        const char *c0 = (const char*)C, *c1 = c0 + ((size_t)(M - 1) * ldc + No) * (out_dtype == VLY_OUT_F32 ? 4 : 2);
        if (c0 < h1 && h0 < c1) {
            vly_set_error("vly_gemv_rmsnorm_bf16: C overlaps H (the norm re-reads H while C is written: not an in-place operation)");
            return -22;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    if (M == 1) return launch_norm_mr<1, 0>(H, gamma, eps, W, bias, residual, C, M, N, K, ldh, ldw, ldc, ldr, epilogue, out_dtype, st, "vly_gemv_rmsnorm_bf16");
    return launch_norm_mr<2, 0>(H, gamma, eps, W, bias, residual, C, M, N, K, ldh, ldw, ldc, ldr, epilogue, out_dtype, st, "vly_gemv_rmsnorm_bf16");
}

#ifndef VLY_EXPERIMENTAL
#define VLY_EXPERIMENTAL 0
#endif
#if VLY_EXPERIMENTAL      // round 3's o projection with the attention merge in its prologue (libvalley_hip_exp.so only)
extern "C" int vly_gemv_attnmerge_bf16(const float* partials, const void* W, const float* bias, const float* residual, void* C, int M,
                                       int N, int heads, int ldw, int ldc, int ldr, int out_dtype, void* stream) {
    const int K = heads * 128;
    if (M <= 0 || M > 2 || N <= 0 || heads <= 0 || K < 2048 || K > 6144 || ldw % 8 || ldw < K || ((uintptr_t)partials & 15) ||
        ((uintptr_t)W & 15)) {
        vly_set_error("vly_gemv_attnmerge_bf16: unsupported shape/alignment M=%d N=%d heads=%d ldw=%d (M <= 2, 2048 <= heads*128 <= 6144)", M,
                      N, heads, ldw);
        return -22;
    }
    hipStream_t st = (hipStream_t)stream;
    if (M == 1) return launch_norm_mr<1, 1>(partials, nullptr, 0.f, W, bias, residual, C, M, N, K, heads, ldw, ldc, ldr, VLY_EPI_NONE, out_dtype, st, "vly_gemv_attnmerge_bf16");
    return launch_norm_mr<2, 1>(partials, nullptr, 0.f, W, bias, residual, C, M, N, K, heads, ldw, ldc, ldr, VLY_EPI_NONE, out_dtype, st, "vly_gemv_attnmerge_bf16");
}
#endif
