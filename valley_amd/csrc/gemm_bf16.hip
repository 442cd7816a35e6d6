// bf16 GEMM with fused epilogues for gfx950:  C[M,N] = epi(A[M,K] · W[N,K]^T + bias) + residual.
//
// Both operands are K-contiguous (activations row-major, nn.Linear weights as stored), so one
// staging routine serves both.  Structure (MI355X_MICROARCH / cdna_hip_programming §5):
//   * tile BM x BN x 64, waves laid out (BM/WM) x (BN/WN), each wave owns WM x WN of C as
//     (WM/16) x (WN/16) MFMA 16x16x32 fragments held in registers for the whole K loop;
//   * global -> LDS by `global_load_lds_dwordx4` (no VGPR round trip): the LDS image is
//     lane-linear [row][8 x 16B], so the bank swizzle chunk ^= (row & 7) is applied to the
//     per-lane SOURCE address and again on the ds_read_b128 (rule 21: both sides or neither);
//   * two LDS stages, one barrier per K tile: the loads of tile t+1 are in flight while the
//     MFMAs of tile t run;
//   * operands are fed to the MFMA swapped (W fragment as the "A" operand), so every lane ends
//     up holding 4 consecutive n for one m: bias/residual are float4 loads and C is written as
//     8-byte (bf16) or 16-byte (fp32) pieces along the row;
//   * block id -> tile mapping gives each XCD (private 4 MiB L2) a contiguous run of tiles.
// Two kernels live here: gemm_kernel (one tile per workgroup; 4 / 8 / 16 waves, the K-loop variants PIPE 0..8) and
// gemm_p4_kernel (tile hints 197-199, the hot path's choice): ONE workgroup per CU walks its tiles, one wave per SIMD
// with a (BM/2) x 128 accumulator block, software-pipelined fragment reads and LDS-DMA pieces across K-tile AND tile
// boundaries, epilogue on registers only — see the comments at PIPE 8 and at gemm_p4_kernel, and DESIGN.md §4.
// Algorithmic work: 2*M*N*K flop per launch; HBM traffic floor (M*K + N*K)*2 + M*N*out bytes.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "common.hpp"
#include "../../include/valley_hip.h"

// Bytes that must come from beyond an XCD's L2 under each tile order (8 XCDs, each owning a contiguous
// run of tiles): n-fastest -> A once + W min(8, tiles_m) times; m-fastest -> W once + A min(8, tiles_n) times.
int vly_tile_order_m_fast(int M, int N, int K, int tiles_m, int tiles_n) {
    const double a = (double)M * K, w = (double)N * K;
    const double n_fast = a + w * (tiles_m < 8 ? tiles_m : 8);
    const double m_fast = w + a * (tiles_n < 8 ? tiles_n : 8);
    return m_fast < n_fast;
}

// Group height (in m-tiles) of the tile order inside each XCD's contiguous run: tiles are visited group by group, a
// group = gm consecutive m-tiles x all n-tiles, m fastest inside it.  gm = 1 is the n-fastest order, gm = tiles_m the
// m-fastest one.  The `conc` workgroups an XCD runs at the same time then cover a gm x (conc / gm) block of tiles, which
// per K step touches gm A-tiles + conc / gm W-tiles: minimal for gm ~ sqrt(conc * BN / BM) — e.g. 11 x 3 -> 6 x 5.3
// tiles for the 13B gate/up GEMM (14 -> 11.3 operand tiles per K step and 32 workgroups, -19 % L2 fill traffic).
// VLY_TILE_GM=<n> overrides (A/B measurements).
int vly_tile_group_height(int M, int N, int K, int tiles_m, int tiles_n, int BM, int BN, int wg_per_cu) {
    static const int forced = getenv("VLY_TILE_GM") ? atoi(getenv("VLY_TILE_GM")) : 0;
    static const int forced_max_m = getenv("VLY_TILE_GM_MAXM") ? atoi(getenv("VLY_TILE_GM_MAXM")) : 0x7fffffff;   // (only shapes up to M rows)
    if (forced > 0 && M <= forced_max_m) return forced < tiles_m ? forced : tiles_m;
    static const bool grouped = !(getenv("VLY_TILE_GROUPED") && atoi(getenv("VLY_TILE_GROUPED")) == 0);
    if (!grouped) return vly_tile_order_m_fast(M, N, K, tiles_m, tiles_n) ? tiles_m : 1;      // round-1 behaviour
    // the near-square block also serves the shapes whose byte count prefers n-fastest (tall A, small W — the ViT GEMMs):
    // measured, own process per setting (tools/gemm_time.py, cold operands): fc1 32768x4096x1024 343 us (gm = 1) -> 327 us
    // (gm = 6), fc2 286 -> 276 us (gm = 8); q|k|v and out-proj flat; gm = 1 is the WORST choice for the Llama shapes
    // ... except when tiles_n divides the 8 XCDs: n-fastest then hands every XCD ONE column block of W for the whole launch
    // (workgroup i -> XCD i % 8), which stays in its 4 MiB L2 — fc2 32768x1024x4096 in the c3 bench: 1172 TF n-fastest,
    // 1121 TF grouped (profiles/history/r02/r02_ab_tile_group.txt)
    if (tiles_n <= 8 && 8 % tiles_n == 0 && !vly_tile_order_m_fast(M, N, K, tiles_m, tiles_n)) return 1;
    const double conc = 32.0 * wg_per_cu;
    int gm = (int)(sqrt(conc * BN / BM) + 0.5);
    if (gm < 1) gm = 1;
    if (gm >= tiles_m || tiles_m <= gm + gm / 2) return tiles_m;       // a ragged last group would be worse than one group
    const int groups = (tiles_m + gm - 1) / gm;                           // balance the groups: 11 -> 6 + 5
    return (tiles_m + groups - 1) / groups;
}

namespace {

constexpr int BK = 64;            // K tile (elements) = 128 bytes per row = 8 chunks of 16 B

// One K tile (BK = 64 = two MFMA K steps) of a wave's MI x NI fragments from a 128-byte-row LDS stage: K step 1's fragments
// are requested BETWEEN K step 0's MFMAs, each into the registers the finished MFMAs freed — one exposed LDS round trip per
// K tile instead of the six the compiler's own order ({2 reads, lgkmcnt(0), 4 MFMAs} x 6) leaves; +1.0..2.7 % on the
// 2-stage loops, +5.9 % on the 3-stage 192x192 loop (interleaved A/B on cold weights, round 1; the other orders that were
// tried — source order, all reads of a step first, both steps first — are in the history of this file and DESIGN.md §4).
// ldw == VLY_LDW_PACKED64: W is stored as [K/64][ceil(N/64)][64 rows][64 k] blocks (vly_pack_weight_bf16) — every K
// tile of a weight panel is one contiguous run in HBM instead of one 128-byte line out of each 2*K-byte row
VLY_DEVICE uint32_t w_row_off(int n, int ldw) {
    return ldw < 0 ? (uint32_t)(n >> 6) * 4096u + (uint32_t)(n & 63) * 64u : (uint32_t)n * (uint32_t)ldw;
}
template <int MI, int NI>
VLY_DEVICE void mma_ktile(f32x4 (&acc)[MI][NI], const char* pa, const char* pw, int sw0, int sw1) {
    bf16x8 a0[MI], w0[NI], a1[MI], w1[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) w0[j] = *(const bf16x8*)(pw + j * 2048 + sw0);
#pragma unroll
    for (int i = 0; i < MI; ++i) a0[i] = *(const bf16x8*)(pa + i * 2048 + sw0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = mfma16(w0[j], a0[i], acc[i][j]);
        if (i == 0) {
#pragma unroll
            for (int j = 0; j < NI; ++j) w1[j] = *(const bf16x8*)(pw + j * 2048 + sw1);
        }
        a1[i] = *(const bf16x8*)(pa + i * 2048 + sw1);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = mfma16(w1[j], a1[i], acc[i][j]);
    __builtin_amdgcn_sched_barrier(0);
}

// (The 32x32x16 MFMA variants of these loops — half the operand reads per flop — measured 7-10 % SLOWER in rounds 1 and 2,
// with and without a conflict-free swizzle for their 32-row fragments: DESIGN.md §4; the code is in the round-2 history.)
// Chunk swizzle of the LDS stage rows, applied to the staging SOURCE address and to the fragment reads: chunk ^= row & 7,
// conflict-free for the 16-row fragments of the 16x16x32 MFMA.

// placement constants of the 4-wave loops (A/B of other placements: profiles/history/r02/r02_ab_4wave.txt)
constexpr int P8_BAR_GAP = 8;       // barrier A this many MFMAs (~140 clk, a ds_read round trip) after the last read of phase 1
constexpr int P8_RD2_START = 1;     // phase 2: reads of the next tile's K step 0 after MFMA 1, 3, 5, ...
constexpr int P8_RD2_STRIDE = 2;
constexpr int P4_M0_LEAD = 2;       // the M0 write of a piece sits this many MFMAs before its buffer_load
// the persistent kernel's tile boundary (round 5; DESIGN.md §4.1 "the tile boundary").  The rolled instantiations — bf16 outputs, every
// epilogue but the fused RoPE, no split-K — keep their accumulators BY NAME (mfma16_lit below) and issue a tile's first K step with
// C = 0 (no 256 v_accvgpr_write per tile: +2.0..2.2 % on the K = 1024 shapes, profiles/r05/r05_boundary_ab_1_switches.txt); on top:
//   VLY_P4_ROLL / VLY_P4_ROLL_MASK  the ROLLING epilogue: the finished tile's epilogue runs row by row INSIDE the next tile's first K
//                    tile (`boundary` in gemm_p4_kernel), for the epilogues of the mask
//   VLY_P4_LIT=0     accumulators as C++ values everywhere (the round-4 kernel; A/B builds)
//   VLY_P4_DROPSTORE diagnostic: every store of the bf16 epilogue is issued out of range (dropped by the hardware)
//   VLY_P4_TIMING    anatomy builds: s_memtime stamps at the seams of every tile (tools/p4_boundary_times.py)
// (measured and removed — commit b4a1225 holds both: whole-line epilogue stores (8 rows x 128 B per instruction by a row_ror:8 exchange)
//  and a four-phase start skew of an XCD's workgroups, alone and together: -1.8 .. +1.1 %, profiles/r05/r05_boundary_ab_4_fullline_skew.txt;
//  waiting for the loads in flight BEFORE the epilogue's first store so that the next K tile's counted wait
//  does not also wait for the stores' acknowledgements — -0.7 .. +0.9 %, noise: the stores throttle the epilogue at their ISSUE;
//  the zero-free first step with the accumulators as loop-carried C++ values tied "+a" — +2 % until any restructuring of the loop
//  made hipcc resolve a phi in arch VGPRs: 86 .. 446 spilled registers under three forms of pinning)
#ifndef VLY_P4_ROLL
#define VLY_P4_ROLL 1
#endif
#ifndef VLY_P4_ROLL_MASK
#define VLY_P4_ROLL_MASK 0x09       // bit e: epilogue e (VLY_EPI_*) takes the rolled boundary (plain + ReLU); the others keep their epilogue phase
#endif                              // (still with the accumulators by name and the zero-free first K step)
#ifndef VLY_P4_LIT
#define VLY_P4_LIT 1                // 0: accumulators as C++ values everywhere (the round-4 kernel)
#endif
#ifndef VLY_P4_CHAIN
#define VLY_P4_CHAIN 0              // 1: the by-name instantiations issue the two K steps of a block BACK TO BACK (see `CHAIN` in gemm_p4_kernel)
#endif
#ifndef VLY_P4_LATE
#define VLY_P4_LATE 0               // 1: barrier B and the next K tile's first fragment reads late in phase 2 (see ktile)
#endif
#ifndef VLY_P4_DROPSTORE
#define VLY_P4_DROPSTORE 0
#endif
#ifndef VLY_P4_TIMING
#define VLY_P4_TIMING 0
#endif
#ifndef VLY_EXPERIMENTAL
#define VLY_EXPERIMENTAL 0          // 1: libvalley_hip_exp.so — also carries tile hint 297
#endif

// One phase of the 4-wave loop (PIPE 8): the MI x NI MFMAs of one 32-wide K step on fragments that are already in
// registers, with up to three lists of other instructions (the NI + MI fragment reads of the NEXT step — W fragments
// first, then A, the order the next phase consumes them —, the LDS-DMA pieces of a later K tile, a barrier) placed at fixed
// MFMA indices.  Every non-MFMA instruction is fenced so the compiler cannot bunch them up in front of the MFMAs (it
// does: ISA of the first version had nine ds_reads + lgkmcnt(0) ahead of the first MFMA of each phase).  One wave per
// SIMD: nothing else hides a gap in this wave's MFMA stream.
// f1(k), k < N1, goes after MFMA number S1 + k * D1 (row-major over the MI x NI MFMAs); f2 / f3 likewise
// ---- accumulators by NAME (round 5, the rolled kernels).  Block b of a wave's tile IS a[4b : 4b + 3]: every MFMA and every read of
// the accumulators is an asm statement that spells the registers out (the block number is an immediate operand, printed into the
// register range), and no C++ value ever holds an accumulator.  hipcc allocates the accumulation registers only because of the one
// clobber list at the top of the kernel (VLY_ALL_AGPRS) and — having no value of its own there, and no free accumulation register
// to spill into — never emits a v_accvgpr_* of its own; tools/isa_blocks.py --agpr-audit checks exactly that after every build.
// Why: as loop-carried C++ values (which the rolling epilogue needs: a block is re-used by the next tile the moment it is read) the
// 256 accumulators put the allocator at exactly 256 of 256 registers, and every copy it inserts at a merge point or in front of a
// tied asm operand is a spill (what three forms of pinning still left: 86 .. 446 spilled registers).
// Wait states are ours now (nothing inside or around an asm statement is padded): an MFMA's D is read (v_accvgpr_read, or the next
// K step's MFMA taking it as C) no sooner than a fragment row = 8 MFMAs later everywhere below, far beyond the 12 states an
// 8-pass MFMA needs; the MFMA operands come from ds_read (waited for by explicit lgkmcnt waits), never from a VALU write.
#if VLY_FP16
#define VLY_MFMA16_NAME "v_mfma_f32_16x16x32_f16"
#else
#define VLY_MFMA16_NAME "v_mfma_f32_16x16x32_bf16"
#endif
VLY_DEVICE void mfma16_lit(int blk, const bf16x8& a, const bf16x8& b) {          // a[4 blk ..] += A . B   (blk: constant after unrolling)
    asm volatile(VLY_MFMA16_NAME " a[%2:%3], %0, %1, a[%2:%3]" ::"v"(a), "v"(b), "i"(4 * blk), "i"(4 * blk + 3));
}
VLY_DEVICE void mfma16_lit_zero(int blk, const bf16x8& a, const bf16x8& b) {     // a[4 blk ..]  = A . B
    asm volatile(VLY_MFMA16_NAME " a[%2:%3], %0, %1, 0" ::"v"(a), "v"(b), "i"(4 * blk), "i"(4 * blk + 3));
}
VLY_DEVICE f32x4 acc_read_lit(int blk) {
    f32x4 v;
    asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\tv_accvgpr_read_b32 %3, a[%7]"
                 : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3])
                 : "i"(4 * blk), "i"(4 * blk + 1), "i"(4 * blk + 2), "i"(4 * blk + 3));
    return v;
}
#define VLY_A8(n) "a" #n "0", "a" #n "1", "a" #n "2", "a" #n "3", "a" #n "4", "a" #n "5", "a" #n "6", "a" #n "7", "a" #n "8", "a" #n "9"
#define VLY_ALL_AGPRS                                                                                                                  \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", VLY_A8(1), VLY_A8(2), VLY_A8(3), VLY_A8(4), VLY_A8(5), VLY_A8(6), VLY_A8(7), \
        VLY_A8(8), VLY_A8(9), VLY_A8(10), VLY_A8(11), VLY_A8(12), VLY_A8(13), VLY_A8(14), VLY_A8(15), VLY_A8(16), VLY_A8(17), VLY_A8(18),   \
        VLY_A8(19), VLY_A8(20), VLY_A8(21), VLY_A8(22), VLY_A8(23), VLY_A8(24), "a250", "a251", "a252", "a253", "a254", "a255"

// MODE: 0 = builtin accumulate into acc[][]; 2 / 3 = accumulators by name: accumulate / first K step of a tile (C = 0)
template <int MI, int NI, int N1, int S1, int D1, int N2, int S2, int D2, int N3, int S3, int D3, int N4, int S4, int D4, int MODE = 0,
          typename F1, typename F2, typename F3, typename F4, typename ACC>
VLY_DEVICE void phase_4w4(ACC& acc, const bf16x8 (&af)[MI], const bf16x8 (&wf)[NI], F1&& f1, F2&& f2, F3&& f3, F4&& f4) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < N4; ++k)
        if (S4 + k * D4 < 0) f4(k);                                  // a slot in front of the first MFMA
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            if constexpr (MODE == 3) mfma16_lit_zero(i * NI + j, wf[j], af[i]);
            else if constexpr (MODE == 2) mfma16_lit(i * NI + j, wf[j], af[i]);
            else acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);
            const int t = i * NI + j;
            if (N1 > 0 && t >= S1 && (t - S1) % D1 == 0 && (t - S1) / D1 < N1) {
                __builtin_amdgcn_sched_barrier(0);
                f1((t - S1) / D1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (N3 > 0 && t >= S3 && (t - S3) % D3 == 0 && (t - S3) / D3 < N3) {
                __builtin_amdgcn_sched_barrier(0);
                f3((t - S3) / D3);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (N4 > 0 && t >= S4 && (t - S4) % D4 == 0 && (t - S4) / D4 < N4) {
                __builtin_amdgcn_sched_barrier(0);
                f4((t - S4) / D4);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (N2 > 0 && t >= S2 && (t - S2) % D2 == 0 && (t - S2) / D2 < N2) {
                __builtin_amdgcn_sched_barrier(0);
                f2((t - S2) / D2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // whatever did not fit the MFMA slots (small MI x NI)
#pragma unroll
    for (int k = 0; k < N1; ++k)
        if (S1 + k * D1 >= MI * NI) f1(k);
#pragma unroll
    for (int k = 0; k < N3; ++k)
        if (S3 + k * D3 >= MI * NI) f3(k);
#pragma unroll
    for (int k = 0; k < N2; ++k)
        if (S2 + k * D2 >= MI * NI) {
            if (k < N4 && S4 + k * D4 >= MI * NI) f4(k);
            f2(k);
        }
    __builtin_amdgcn_sched_barrier(0);
}
// T MFMAs issued by mf(t) with the same four hook lists as phase_4w4 (order behind MFMA t: f1, f3, f4, f2) — the chain-order phases
template <int T, int N1, int S1, int D1, int N2, int S2, int D2, int N3, int S3, int D3, int N4, int S4, int D4, typename MF, typename F1, typename F2,
          typename F3, typename F4>
VLY_DEVICE void phase_seq(MF&& mf, F1&& f1, F2&& f2, F3&& f3, F4&& f4) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < N4; ++k)
        if (S4 + k * D4 < 0) f4(k);
#pragma unroll
    for (int t = 0; t < T; ++t) {
        mf(t);
        if (N1 > 0 && t >= S1 && (t - S1) % D1 == 0 && (t - S1) / D1 < N1) {
            __builtin_amdgcn_sched_barrier(0);
            f1((t - S1) / D1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (N3 > 0 && t >= S3 && (t - S3) % D3 == 0 && (t - S3) / D3 < N3) {
            __builtin_amdgcn_sched_barrier(0);
            f3((t - S3) / D3);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (N4 > 0 && t >= S4 && (t - S4) % D4 == 0 && (t - S4) / D4 < N4) {
            __builtin_amdgcn_sched_barrier(0);
            f4((t - S4) / D4);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (N2 > 0 && t >= S2 && (t - S2) % D2 == 0 && (t - S2) / D2 < N2) {
            __builtin_amdgcn_sched_barrier(0);
            f2((t - S2) / D2);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    static_assert((N1 == 0 || S1 + (N1 - 1) * D1 < T) && (N2 == 0 || S2 + (N2 - 1) * D2 < T) && (N3 == 0 || S3 + (N3 - 1) * D3 < T) &&
                  (N4 == 0 || S4 + (N4 - 1) * D4 < T), "every hook inside the phase");
    __builtin_amdgcn_sched_barrier(0);
}
template <int MI, int NI, int N1, int S1, int D1, int N2, int S2, int D2, int N3, int S3, int D3, typename F1, typename F2, typename F3, typename ACC>
VLY_DEVICE void phase_4w3(ACC& acc, const bf16x8 (&af)[MI], const bf16x8 (&wf)[NI], F1&& f1, F2&& f2, F3&& f3) {
    phase_4w4<MI, NI, N1, S1, D1, N2, S2, D2, N3, S3, D3, 0, 0, 1, 0>(acc, af, wf, f1, f2, f3, [](int) {});
}
template <int MI, int NI, int N1, int S1, int D1, int N2, int S2, int D2, typename F1, typename F2, typename ACC>
VLY_DEVICE void phase_4w(ACC& acc, const bf16x8 (&af)[MI], const bf16x8 (&wf)[NI], F1&& f1, F2&& f2) {
    phase_4w4<MI, NI, N1, S1, D1, N2, S2, D2, 0, 0, 1, 0, 0, 1, 0>(acc, af, wf, f1, f2, [](int) {}, [](int) {});
}

// arguments of the VLY_EPI_QKV_ROPE epilogue (kernel argument by value; unused by every other instantiation)
struct RopeArgs {
    const float* cos_t;
    const float* sin_t;
    uint16_t* kc;
    uint16_t* vc;
    int S, past, heads, ctx_max;
};

template <int BM, int BN, int WM, int WN, int EPI, int OUT, int PIPE>
__global__ void __launch_bounds__((BM / WM) * (BN / WN) * 64)
gemm_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W,
            const float* __restrict__ bias, const float* __restrict__ R, void* __restrict__ Cv,
            int M, int N, int K, int lda, int ldw, int ldc, int ldr, int tiles_m, int tiles_n, int gm, int wide,
            int ksplit, void* __restrict__ Cv2, RopeArgs rp) {
    constexpr int NW = (BM / WM) * (BN / WN);
    constexpr int NT = NW * 64;
    constexpr int MI = WM / 16, NI = WN / 16;
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;
    // glds passes per tile.  A tile height that is not a multiple of NT / 8 rows (224 rows on 512 threads = 3.5 passes)
    // takes a last pass in which only the waves whose 64-slot run still lies inside the tile load (wave-uniform guard);
    // such tiles use the vmcnt(0) loops only (the counted-vmcnt loops assume the same number of loads in every wave)
    constexpr int PA = (BM * 8 + NT - 1) / NT, PW = BN * 8 / NT;
    constexpr bool A_RAGGED = BM * 8 % NT != 0;
    static_assert(BN * 8 % NT == 0 && BM * 8 % 64 == 0, "tile/threads mismatch");
    static_assert(!A_RAGGED || PIPE == 0 || PIPE == 4, "ragged A staging needs a vmcnt(0) loop");

    // bf16 outputs leave through LDS (see the epilogue): the C tile image may be larger than the K-loop stages
    constexpr int BNO = EPI == VLY_EPI_SWIGLU ? BN / 2 : BN;          // output columns of the tile
    constexpr int C_ROW = BNO * 2 + 16;                               // bytes per LDS row of the C image (+16: bank shift)
    constexpr int K_BYTES = (PIPE == 6 || PIPE == 7 ? 3 : 2) * STAGE;
    constexpr int SMEM_BYTES = (OUT == VLY_OUT_BF16 && BM * C_ROW > K_BYTES) ? BM * C_ROW : K_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[SMEM_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;

    // ---- XCD-aware tile assignment (bijective for any grid size) ---------------------------
    // ksplit == 2: the grid holds every tile twice; workgroups [0, tiles) reduce the first half of K into C,
    // workgroups [tiles, 2*tiles) the second half into C2 (no bias / residual there) and the consumer adds the
    // two bf16 partials (vly_add2_*norm).  For the N = 4096 Llama projections at M = 1312 this turns 176-224
    // one-per-CU workgroups into 352-448 that pair up on the CUs.
    const int nwg = ksplit == 2 ? (int)gridDim.x >> 1 : (int)gridDim.x;
    const int part = (ksplit == 2 && (int)blockIdx.x >= nwg) ? 1 : 0;
    const int bid = (int)blockIdx.x - part * nwg;
    if (part) { Cv = Cv2; bias = nullptr; R = nullptr; }
    const int xcd = bid & 7, qd = nwg >> 3, rm = nwg & 7;
    const int swz = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (bid >> 3);
    // tile order inside the XCD-contiguous run: groups of gm m-tiles x all n-tiles, m fastest inside a group
    // (vly_tile_group_height: gm = 1 n-fastest — an A row-panel stays in the XCD's L2 while the W panels stream;
    // gm = tiles_m m-fastest; in between, the workgroups an XCD runs together form a near-square block of tiles)
    const int gsz = gm * tiles_n, grp = swz / gsz, first = grp * gm;
    const int gh = min(gm, tiles_m - first), rr = swz - grp * gsz;
    const int m0 = (first + rr % gh) * BM;
    const int n0 = (rr / gh) * BN;

    const int wm0 = (wave / (BN / WN)) * WM, wn0 = (wave % (BN / WN)) * WN;
    // a wave whose whole WM x WN slab lies past the edge of the problem (the lower half of the last 256-row tile at
    // M = 2688 or F*257) keeps staging and meeting the barriers but skips its fragment reads and MFMAs: the tile then
    // costs its SIMD partner's share only, and the last round of a launch is shorter
    const bool wave_live = __builtin_amdgcn_readfirstlane((m0 + wm0 < M && n0 + wn0 < N) ? 1 : 0) != 0;
    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int nk = K / BK;
    const uint32_t wk = ldw < 0 ? (uint32_t)((N + 63) >> 6) * 4096u : (uint32_t)BK;           // W elements per K tile
    if (ksplit == 2) {
        const int h0 = nk >> 1;
        if (part) { A += (size_t)h0 * BK; W += (size_t)h0 * wk; nk -= h0; } else nk = h0;
    }

    if constexpr (PIPE == 0) {
        // ================= 2-stage loop: whole K tiles, vmcnt(0) + one barrier per tile ================
        uint32_t offA[PA], offW[PW];
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int s = p * NT + tid, row = s >> 3, cp = s & 7;
            const int gr = min(m0 + row, M - 1);
            offA[p] = (uint32_t)gr * (uint32_t)lda + (uint32_t)((cp ^ (row & 7)) << 3);
        }
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            const int s = p * NT + tid, row = s >> 3, cp = s & 7;
            const int gr = min(n0 + row, N - 1);
            offW[p] = w_row_off(gr, ldw) + (uint32_t)((cp ^ (row & 7)) << 3);
        }
        auto stage = [&](int kt, int buf) {
            char* sA = smem + buf * STAGE;
            char* sW = sA + A_BYTES;
            const int k0 = kt * BK;
#pragma unroll
            for (int p = 0; p < PA; ++p)
                if (!A_RAGGED || p + 1 < PA || (p * NT + wave * 64) < BM * 8)
                    glds16(A + offA[p] + k0, sA + (p * NT + wave * 64) * 16);
#pragma unroll
            for (int p = 0; p < PW; ++p) glds16(W + offW[p] + kt * wk, sW + (p * NT + wave * 64) * 16);
        };
        // row & 7 == l15 & 7 for every fragment row (all bases are multiples of 16)
        const int rdA = (wm0 + l15) * 128, rdW = (wn0 + l15) * 128;
        const int sw0 = ((0 + g) ^ (l15 & 7)) << 4, sw1 = ((4 + g) ^ (l15 & 7)) << 4;

        stage(0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            __syncthreads();                              // tile kt landed; buffer (kt+1)&1 is free
            if (kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
            const char* sA = smem + (kt & 1) * STAGE;
            const char* sW = sA + A_BYTES;
            if (!wave_live) continue;
            mma_ktile<MI, NI>(acc, sA + rdA, sW + rdW, sw0, sw1);
        }
    } else if constexpr (PIPE == 4) {
        // ================= role-split over the 2-stage full-tile buffers (8 waves) ==========================
        // Same staging as PIPE 0 (whole K tiles, 128-byte LDS rows -> full-line global reads), but the K tile
        // is computed in two phases (k-step 0, k-step 1) and wave group 1 (waves 4-7, the SIMD partners of
        // waves 0-3) runs one phase behind group 0: R0 = {issue loads of tile t+1, read k-step-0 fragments},
        // M0, R1 = {read k-step-1 fragments}, M1 — one raw barrier after every phase, MFMA phases at
        // s_setprio 1, vmcnt(0) once per tile before the barrier that precedes the first read of tile t+1.
        static_assert(NW == 8, "role-split pipeline needs two waves per SIMD");
        uint32_t offA[PA], offW[PW];
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int s = p * NT + tid, row = s >> 3, cp = s & 7;
            offA[p] = (uint32_t)min(m0 + row, M - 1) * (uint32_t)lda + (uint32_t)((cp ^ (row & 7)) << 3);
        }
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            const int s = p * NT + tid, row = s >> 3, cp = s & 7;
            offW[p] = w_row_off(min(n0 + row, N - 1), ldw) + (uint32_t)((cp ^ (row & 7)) << 3);
        }
        auto stage = [&](int kt, int buf) {
            char* sA = smem + buf * STAGE;
            char* sW = sA + A_BYTES;
            const int k0 = kt * BK;
#pragma unroll
            for (int p = 0; p < PA; ++p)
                if (!A_RAGGED || p + 1 < PA || (p * NT + wave * 64) < BM * 8)
                    glds16(A + offA[p] + k0, sA + (p * NT + wave * 64) * 16);
#pragma unroll
            for (int p = 0; p < PW; ++p) glds16(W + offW[p] + kt * wk, sW + (p * NT + wave * 64) * 16);
        };
        const int rdA = (wm0 + l15) * 128, rdW = A_BYTES + (wn0 + l15) * 128;
        const int sw0 = ((0 + g) ^ (l15 & 7)) << 4, sw1 = ((4 + g) ^ (l15 & 7)) << 4;
        const int grp = wave >> 2;

        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                               // tile 0 visible to everyone
        if (grp == 1) __builtin_amdgcn_s_barrier();                 // group 1 starts one phase late
        for (int kt = 0; kt < nk; ++kt) {
            const char* cur = smem + (kt & 1) * STAGE;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                // ---------------- R phase
                if (kk == 0 && kt + 1 < nk) stage(kt + 1, (kt + 1) & 1);
                const int sw = kk ? sw1 : sw0;
                bf16x8 af[MI], wf[NI];
                if (wave_live) {
                {
#pragma unroll
                for (int j = 0; j < NI; ++j) wf[j] = *(const bf16x8*)(cur + rdW + j * 2048 + sw);
#pragma unroll
                for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(cur + rdA + i * 2048 + sw);
                }
                }
                if (kk == 1 && grp == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my loads of tile kt+1
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                // ---------------- M phase
                __builtin_amdgcn_s_setprio(1);
                if (wave_live) {
                {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);
                }
                }
                __builtin_amdgcn_s_setprio(0);
                if (kk == 1 && grp == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            }
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();
    } else if constexpr (PIPE == 6) {
        // ================= 3-stage loop: whole K tiles, two tiles in flight, counted vmcnt ===================
        // Little's law check: one 64 KB tile in flight per CU sustains ~32 B/clk only if the loaded L2/HBM
        // latency stays under ~2000 cycles.  Tiles whose stage is <= 53 KB can keep THREE stages in the
        // 160 KB LDS, i.e. two K tiles (2x the bytes) in flight, with one raw barrier per K tile.
        constexpr int NS = 3;
        static_assert(NS * STAGE <= 160 * 1024, "LDS budget");
        uint32_t offA[PA], offW[PW];
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int s = p * NT + tid, row = s >> 3, cp = s & 7;
            offA[p] = (uint32_t)min(m0 + row, M - 1) * (uint32_t)lda + (uint32_t)((cp ^ (row & 7)) << 3);
        }
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            const int s = p * NT + tid, row = s >> 3, cp = s & 7;
            offW[p] = w_row_off(min(n0 + row, N - 1), ldw) + (uint32_t)((cp ^ (row & 7)) << 3);
        }
        auto stage = [&](int kt, int buf) {
            char* sA = smem + buf * STAGE;
            char* sW = sA + A_BYTES;
            const int k0 = kt * BK;
#pragma unroll
            for (int p = 0; p < PA; ++p)
                if (!A_RAGGED || p + 1 < PA || (p * NT + wave * 64) < BM * 8)
                    glds16(A + offA[p] + k0, sA + (p * NT + wave * 64) * 16);
#pragma unroll
            for (int p = 0; p < PW; ++p) glds16(W + offW[p] + kt * wk, sW + (p * NT + wave * 64) * 16);
        };
        const int rdA = (wm0 + l15) * 128, rdW = A_BYTES + (wn0 + l15) * 128;
        const int sw0 = ((0 + g) ^ (l15 & 7)) << 4, sw1 = ((4 + g) ^ (l15 & 7)) << 4;

        stage(0, 0);
        if (nk > 1) stage(1, 1);
        int buf = 0;
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PA + PW) : "memory");   // tile kt landed, kt+1 may fly
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (kt + 2 < nk) stage(kt + 2, buf == 0 ? 2 : buf - 1);                           // (kt+2) % 3
            const char* cur = smem + buf * STAGE;
            if (wave_live) {
            mma_ktile<MI, NI>(acc, cur + rdA, cur + rdW, sw0, sw1);
            }
            buf = buf == 2 ? 0 : buf + 1;
        }
    } else if constexpr (PIPE == 8) {
        // ================= 4 waves, one per SIMD, 128 x 128 per wave; two whole-K-tile buffers (128-byte rows) ============
        // LDS-pipe arithmetic per 256 x 256 x 64 K tile (MFMA time at peak: 2062 clk): fragment reads are
        // waves x (WM + WN) x 128 B at 256 B/clk — 16 waves of 64 x 64: 1024 clk, 8 waves of 128 x 64: 768 clk, 4 waves of
        // 128 x 128: 512 clk.  The many-wave tiles buy latency hiding with LDS traffic (MFMA busy 56-60 %, PMC); this loop
        // has a quarter / half of their reads and hides latency in software (fragments of the next K step are read between
        // the MFMAs of the current one, across K tiles too).
        // What bounds it is the L1 -> LDS path: 64 B/clk per CU = one 1 KB LDS-DMA piece per 16 clk per CU = per 64 clk for
        // each of the four waves, i.e. one piece per ~4 MFMAs; a wave that issues faster stalls in the issue and, being alone
        // on its SIMD, feeds no MFMAs meanwhile (measured: 16 pieces at every 2nd MFMA -8 % against every 4th; a version of
        // the loop WITHOUT the pieces runs 26-30 % faster; 64-byte-row half tiles (PIPE 9) cost twice the path time per piece).
        // So the 16 pieces of a K tile are spread as thin as the buffers allow — the buffer of tile kt is released as soon as
        // its last fragments are read, in the middle of phase 1, not at its end:
        //   phase 1 (MFMAs of K step 0 of tile kt): reads of K step 1 first | lgkmcnt(0) + barrier A: buffer kt & 1 is free |
        //            the first N1 pieces of tile kt+2 into it
        //   vmcnt(N1) + barrier B: tile kt+1 has landed everywhere
        //   phase 2 (MFMAs of K step 1): reads of K step 0 of tile kt+1 | the other 16 - N1 pieces of tile kt+2.
        // 256 accumulator + 128 fragment registers per lane.
        static_assert(NW == 4 && !A_RAGGED, "one wave per SIMD");
        // slot arithmetic (T = MFMAs per phase): reads of phase 1 after MFMA 0 .. MI+NI-1, barrier A eight MFMAs (~140 clk, a
        // ds_read round trip) later, then one piece every STRIDE MFMAs until the end of phase 2.  256 x 256: barrier after
        // MFMA 24, 7 pieces after 26, 32 .. 62, 9 after 4, 10 .. 52 of phase 2; A/B of other placements: r02_ab_4wave.txt
        constexpr int T = MI * NI, BAR_AT = MI + NI + P8_BAR_GAP, GL1_START = BAR_AT + 2;
        static_assert(GL1_START < T, "phase too short for the early release");
        constexpr int NS = PA + PW;
        constexpr int GL_STRIDE = (2 * T - GL1_START) / NS;                          // 102 / 16 = 6
        constexpr int N1 = (T - GL1_START + GL_STRIDE - 1) / GL_STRIDE, N2 = NS - N1;   // pieces in phase 1 / phase 2
        constexpr int GL2_START = GL1_START + N1 * GL_STRIDE - T;
        static_assert(GL_STRIDE >= 1 && N1 >= 0 && N2 >= 0 && GL2_START + (N2 - 1) * GL_STRIDE < T, "piece schedule");
        uint32_t voA[PA], voW[PW];                                  // per-lane BYTE offsets
#pragma unroll
        for (int q = 0; q < PA; ++q) {
            const int sl = q * NT + tid, row = sl >> 3, cp = sl & 7;
            voA[q] = ((uint32_t)min(m0 + row, M - 1) * (uint32_t)lda + (uint32_t)((cp ^ (row & 7)) << 3)) * 2u;
        }
#pragma unroll
        for (int q = 0; q < PW; ++q) {
            const int sl = q * NT + tid, row = sl >> 3, cp = sl & 7;
            voW[q] = (w_row_off(min(n0 + row, N - 1), ldw) + (uint32_t)((cp ^ (row & 7)) << 3)) * 2u;
        }
        const __amdgpu_buffer_rsrc_t rsA = vly_rsrc(A), rsW = vly_rsrc(W);
        auto piece = [&](int kt, int buf, int q) {
            char* st = smem + buf * STAGE;
            if (q < PA) bglds16<0>(rsA, voA[q < PA ? q : 0], (uint32_t)kt * (BK * 2u), st + (q * NT + wave * 64) * 16);
            else bglds16<0>(rsW, voW[q >= PA ? q - PA : 0], (uint32_t)kt * wk * 2u, st + A_BYTES + ((q - PA) * NT + wave * 64) * 16);
        };
        const int rdA = (wm0 + l15) * 128, rdW = A_BYTES + (wn0 + l15) * 128;
        const int sw0 = ((0 + g) ^ (l15 & 7)) << 4, sw1 = ((4 + g) ^ (l15 & 7)) << 4;
        bf16x8 a0[MI], w0[NI], a1[MI], w1[NI];
        auto rd_step1 = [&](const char* st) {
            return [&, st](int k) {
                if (k < NI) w1[k < NI ? k : 0] = *(const bf16x8*)(st + rdW + k * 2048 + sw1);
                else a1[k >= NI ? k - NI : 0] = *(const bf16x8*)(st + rdA + (k - NI) * 2048 + sw1);
            };
        };
        auto rd_step0 = [&](const char* st) {
            return [&, st](int k) {
                if (k < NI) w0[k < NI ? k : 0] = *(const bf16x8*)(st + rdW + k * 2048 + sw0);
                else a0[k >= NI ? k - NI : 0] = *(const bf16x8*)(st + rdA + (k - NI) * 2048 + sw0);
            };
        };
        auto none = [](int) {};
        auto bar_a = [](int) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        };

#pragma unroll
        for (int q = 0; q < NS; ++q) piece(0, 0, q);
        if (nk > 1) {
#pragma unroll
            for (int q = 0; q < NS; ++q) piece(1, 1, q);
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NS) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (wave_live) {
            auto r0 = rd_step0(smem);
#pragma unroll
            for (int k = 0; k < MI + NI; ++k) r0(k);
        }
        int kt = 0;
        if (nk == 1) {                                              // K = 64: one tile, nothing to overlap
            if (wave_live) {
                phase_4w<MI, NI, MI + NI, 0, 1, 0, 0, 1>(acc, a0, w0, rd_step1(smem), none);
                phase_4w<MI, NI, 0, 0, 1, 0, 0, 1>(acc, a1, w1, none, none);
            }
        } else {
        for (; kt + 2 < nk; ++kt) {
            const char* cur = smem + (kt & 1) * STAGE;
            const char* nxt = smem + ((kt + 1) & 1) * STAGE;
            if (wave_live)
                phase_4w3<MI, NI, MI + NI, 0, 1, N1, GL1_START, GL_STRIDE, 1, BAR_AT, 1>(
                    acc, a0, w0, rd_step1(cur), [&](int q) { piece(kt + 2, kt & 1, q); }, bar_a);
            else {
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int q = 0; q < N1; ++q) piece(kt + 2, kt & 1, q);
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N1) : "memory");
            __builtin_amdgcn_s_barrier();
            if (wave_live)
                phase_4w<MI, NI, MI + NI, P8_RD2_START, P8_RD2_STRIDE, N2, GL2_START, GL_STRIDE>(
                    acc, a1, w1, rd_step0(nxt), [&](int q) { piece(kt + 2, kt & 1, N1 + q); });
            else {
#pragma unroll
                for (int q = 0; q < N2; ++q) piece(kt + 2, kt & 1, N1 + q);
            }
        }
        {   // kt = nk - 2, nk - 1: nothing left to request
            const char* cur = smem + (kt & 1) * STAGE;
            const char* nxt = smem + ((kt + 1) & 1) * STAGE;
            if (wave_live) phase_4w<MI, NI, MI + NI, 0, 1, 0, 0, 1>(acc, a0, w0, rd_step1(cur), none);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (wave_live) {
                phase_4w<MI, NI, MI + NI, P8_RD2_START, P8_RD2_STRIDE, 0, 0, 1>(acc, a1, w1, rd_step0(nxt), none);
                phase_4w<MI, NI, MI + NI, 0, 1, 0, 0, 1>(acc, a0, w0, rd_step1(nxt), none);
                phase_4w<MI, NI, 0, 0, 1, 0, 0, 1>(acc, a1, w1, none, none);
            }
        }
        }
    } else if constexpr (PIPE == 7) {
        // ================= role-split (as PIPE 4) over THREE full-tile stages (as PIPE 6) =====================
        static_assert(NW == 8, "role-split pipeline needs two waves per SIMD");
        static_assert(3 * STAGE <= 160 * 1024, "LDS budget");
        uint32_t offA[PA], offW[PW];
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int s = p * NT + tid, row = s >> 3, cp = s & 7;
            offA[p] = (uint32_t)min(m0 + row, M - 1) * (uint32_t)lda + (uint32_t)((cp ^ (row & 7)) << 3);
        }
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            const int s = p * NT + tid, row = s >> 3, cp = s & 7;
            offW[p] = w_row_off(min(n0 + row, N - 1), ldw) + (uint32_t)((cp ^ (row & 7)) << 3);
        }
        auto stage = [&](int kt, int buf) {
            char* sA = smem + buf * STAGE;
            char* sW = sA + A_BYTES;
            const int k0 = kt * BK;
#pragma unroll
            for (int p = 0; p < PA; ++p)
                if (!A_RAGGED || p + 1 < PA || (p * NT + wave * 64) < BM * 8)
                    glds16(A + offA[p] + k0, sA + (p * NT + wave * 64) * 16);
#pragma unroll
            for (int p = 0; p < PW; ++p) glds16(W + offW[p] + kt * wk, sW + (p * NT + wave * 64) * 16);
        };
        auto wait_next = [&](int kt) {                              // my loads of tile kt+1 landed; kt+2 may fly
            if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(PA + PW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        const int rdA = (wm0 + l15) * 128, rdW = A_BYTES + (wn0 + l15) * 128;
        const int sw0 = ((0 + g) ^ (l15 & 7)) << 4, sw1 = ((4 + g) ^ (l15 & 7)) << 4;
        const int grp = wave >> 2;

        stage(0, 0);
        if (nk > 1) stage(1, 1);
        wait_next(-1);
        __builtin_amdgcn_s_barrier();                               // tile 0 visible to everyone
        if (grp == 1) __builtin_amdgcn_s_barrier();                 // group 1 starts one phase late
        int buf = 0;
        for (int kt = 0; kt < nk; ++kt) {
            const char* cur = smem + buf * STAGE;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                // ---------------- R phase
                if (kk == 0 && kt + 2 < nk) stage(kt + 2, buf == 0 ? 2 : buf - 1);
                const int sw = kk ? sw1 : sw0;
                bf16x8 af[MI], wf[NI];
                if (wave_live) {
                {
#pragma unroll
                for (int j = 0; j < NI; ++j) wf[j] = *(const bf16x8*)(cur + rdW + j * 2048 + sw);
#pragma unroll
                for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(cur + rdA + i * 2048 + sw);
                }
                }
                if (kk == 1 && grp == 1) wait_next(kt);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                // ---------------- M phase
                __builtin_amdgcn_s_setprio(1);
                if (wave_live) {
                {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[i][j] = mfma16(wf[j], af[i], acc[i][j]);
                }
                }
                __builtin_amdgcn_s_setprio(0);
                if (kk == 1 && grp == 0) wait_next(kt);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            }
            buf = buf == 2 ? 0 : buf + 1;
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();
    } else {
        static_assert(PIPE == 0, "unknown K-loop variant");   // (the half-tile loops of rounds 1-2, tile hints +10 / +30, never won a shape and are gone)
    }

    // ---- epilogue: lane holds C[m][n .. n+3], m = .. + l15, n = .. + 4*g ----------------------
    // bf16 outputs go through LDS: a fragment store covers 16 rows x 32 contiguous bytes (a quarter of each
    // 128-byte line per instruction; tools/probes/store_pattern.hip: 3.0 TB/s for the 67 MB ViT fc1 output),
    // so the tile is first assembled in LDS and then written as full lines, 16 bytes per lane with
    // consecutive lanes along the row (4.6 TB/s).  The fused bias / activation / residual math is unchanged.
#define EPI_NH (MI % 2 == 0 ? 2 : 1)        /* an odd number of 16-row fragments per wave leaves in one pass */
#define VLY_FRAG_ROW(i) wm0 + i * 16 + l15
#define VLY_FRAG_COL(i, j) wn0 + j * 16 + g * 4
    if constexpr (OUT == VLY_OUT_BF16 && EPI != VLY_EPI_QKV_ROPE && NI % 4 == 0) {
        if (wide == 2) {
            // ---- full-line stores WITHOUT LDS: the four lanes that hold a row (g = 0 .. 3, sixteen lanes apart) trade
            // register halves with v_permlane16_swap / v_permlane32_swap (gfx950) until each holds 16 contiguous bytes:
            //   plain: per pair of 16-column blocks (j0, j1) one swap16 per packed dword — even g ends up with columns
            //          4g .. 4g+7 of j0, odd g with columns 4(g-1) .. 4g+3 of j1: 64 contiguous bytes per row and pair;
            //   SwiGLU (one dword = 2 outputs per block): a 4 x 4 transpose over four blocks, swap16 then swap32 — lane g
            //          ends up with the 8 outputs of block 4q + g.
            // swap16(x, y): even rows keep x and receive the odd row's x, odd rows receive the even row's y and keep y
            // (tools/probes/permlane_swap.hip).  No barrier, no LDS: the K-loop buffers are not touched, and the epilogue
            // needs no workgroup-wide synchronisation.  EXEC must be full at the swaps: math and swaps run for every lane,
            // only loads and stores are predicated.
            const int No = EPI == VLY_EPI_SWIGLU ? N >> 1 : N;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = m0 + wm0 + i * 16 + l15;
                uint16_t* crow = (uint16_t*)Cv + (size_t)min(m, M - 1) * ldc;
                if constexpr (EPI == VLY_EPI_SWIGLU) {
#pragma unroll
                    for (int jq = 0; jq < NI / 4; ++jq) {
                        uint32_t d[4];
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            const f32x4 v = acc[i][4 * jq + jj];
                            d[jj] = pack_h2(x_sigmoid(v[0], 1.f) * v[1], x_sigmoid(v[2], 1.f) * v[3]);
                        }
                        const auto p01 = __builtin_amdgcn_permlane16_swap(d[0], d[1], false, false);
                        const auto p23 = __builtin_amdgcn_permlane16_swap(d[2], d[3], false, false);
                        const auto q0 = __builtin_amdgcn_permlane32_swap(p01[0], p23[0], false, false);
                        const auto q1 = __builtin_amdgcn_permlane32_swap(p01[1], p23[1], false, false);
                        const u32x4 o = u32x4{q0[0], q1[0], q0[1], q1[1]};
                        const int no = ((n0 + wn0) >> 1) + (4 * jq + g) * 8;
                        if (m < M) {
                            if (no + 8 <= No) *(u32x4*)(crow + no) = o;
                            else if (no + 4 <= No) *(u32x2*)(crow + no) = u32x2{o[0], o[1]};
                        }
                    }
                } else {
#pragma unroll
                    for (int jp = 0; jp < NI / 2; ++jp) {
                        u32x2 pk[2];
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int n = n0 + wn0 + (2 * jp + jj) * 16 + g * 4;
                            f32x4 v = acc[i][2 * jp + jj];
                            const bool in = m < M && n < N;
                            if (bias && in) v += *(const f32x4*)(bias + n);
                            if constexpr (EPI == VLY_EPI_QUICK_GELU) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] = x_sigmoid(v[r], 1.702f);
                            }
                            if constexpr (EPI == VLY_EPI_RELU) {
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                            }
                            if (R && in) v += *(const f32x4*)(R + (size_t)m * ldr + n);
                            pk[jj][0] = pack_h2(v[0], v[1]);
                            pk[jj][1] = pack_h2(v[2], v[3]);
                        }
                        const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
                        const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
                        const u32x4 o = u32x4{s0[0], s1[0], s0[1], s1[1]};
                        const int n = n0 + wn0 + (2 * jp + (g & 1)) * 16 + (g & 2) * 4;
                        if (m < M) {
                            if (n + 8 <= N) *(u32x4*)(crow + n) = o;
                            else if (n + 4 <= N) *(u32x2*)(crow + n) = u32x2{o[0], o[1]};
                        }
                    }
                }
            }
            return;
        }
    }
    if constexpr (OUT == VLY_OUT_BF16) {
        if (wide) {
            __syncthreads();                                  // every wave is done with the K-loop stages
            // two halves of every wave's fragment rows: the stores of the first half are in flight while the
            // second half's bias / activation math runs (the exp of quick_gelu / SiLU is not free)
            static_assert(MI % EPI_NH == 0, "epilogue halves");
            constexpr int HR = (MI / EPI_NH) * 16;                // rows per wave-row block and half
            constexpr int CPR = BNO / 8;                          // 16-byte chunks per tile row
            const int n0o = EPI == VLY_EPI_SWIGLU ? n0 >> 1 : n0, No = EPI == VLY_EPI_SWIGLU ? N >> 1 : N;
#pragma unroll
            for (int half = 0; half < EPI_NH; ++half) {
#pragma unroll
                for (int ii = 0; ii < MI / EPI_NH; ++ii) {
                    const int i = half * (MI / EPI_NH) + ii;
                    const int row = VLY_FRAG_ROW(i), m = m0 + row;
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const int col = VLY_FRAG_COL(i, j), n = n0 + col;
                        f32x4 v = acc[i][j];
                        const bool in = m < M && n < N;
                        if (bias && in) v += *(const f32x4*)(bias + n);
                        if constexpr (EPI == VLY_EPI_QUICK_GELU) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = x_sigmoid(v[r], 1.702f);
                        }
                        if constexpr (EPI == VLY_EPI_RELU) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                        }
                        if constexpr (EPI == VLY_EPI_SWIGLU) {
                            const float o0 = x_sigmoid(v[0], 1.f) * v[1];
                            const float o1 = x_sigmoid(v[2], 1.f) * v[3];
                            *(uint32_t*)(smem + row * C_ROW + (col >> 1) * 2) = pack_h2(o0, o1);
                        } else {
                            if (R && in) v += *(const f32x4*)(R + (size_t)m * ldr + n);
                            u32x2 pk;
                            pk[0] = pack_h2(v[0], v[1]);
                            pk[1] = pack_h2(v[2], v[3]);
                            *(u32x2*)(smem + row * C_ROW + col * 2) = pk;
                        }
                    }
                }
                __syncthreads();
                for (int c = tid; c < (BM / EPI_NH) * CPR; c += NT) {
                    const int hr = c / CPR, q = c - hr * CPR;     // hr: row index inside this half's row set
                    const int row = (hr / HR) * WM + half * HR + hr % HR;
                    const int m = m0 + row, n = n0o + q * 8;
                    if (m >= M || n >= No) continue;
                    u32x4 d = *(const u32x4*)(smem + row * C_ROW + q * 16);
                    uint16_t* dst = (uint16_t*)Cv + (size_t)m * ldc + n;
                    if constexpr (EPI == VLY_EPI_QKV_ROPE) {
                        // the bf16 image of the tile is complete in LDS: rotate q / k chunks with their partner chunk 64
                        // columns away (same head: BN % 128 == 0 and n0 % 128 == 0), route k / v to the KV cache.
                        // Same arithmetic on the same bf16-rounded inputs as rope_kv_kernel -> identical bits.
                        const int Hq = rp.heads * 128, sect = n / Hq, nn = n - sect * Hq;
                        const int head = nn >> 7, dd = nn & 127;
                        const int b = m / rp.S, pos = rp.past + (m - b * rp.S);
                        if (sect < 2) {
                            const u32x4 pd = *(const u32x4*)(smem + row * C_ROW + (dd < 64 ? q + 8 : q - 8) * 16);
                            const float* cp = rp.cos_t + (size_t)pos * 64 + (dd & 63);
                            const float* sp = rp.sin_t + (size_t)pos * 64 + (dd & 63);
                            const f32x4 c0 = *(const f32x4*)cp, c1 = *(const f32x4*)(cp + 4);
                            const f32x4 s0 = *(const f32x4*)sp, s1 = *(const f32x4*)(sp + 4);
                            const float sign = dd < 64 ? -1.f : 1.f;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float x0 = h_lo(d[e]), x1 = h_hi(d[e]);
                                const float y0 = h_lo(pd[e]), y1 = h_hi(pd[e]);
                                const float cc0 = e < 2 ? c0[2 * e] : c1[2 * e - 4], cc1 = e < 2 ? c0[2 * e + 1] : c1[2 * e - 3];
                                const float ss0 = e < 2 ? s0[2 * e] : s1[2 * e - 4], ss1 = e < 2 ? s0[2 * e + 1] : s1[2 * e - 3];
                                d[e] = pack_h2(rope_rot(x0, y0, cc0, ss0, sign), rope_rot(x1, y1, cc1, ss1, sign));
                            }
                        }
                        if (sect > 0)
                            dst = (sect == 1 ? rp.kc : rp.vc) + (((size_t)b * rp.heads + head) * rp.ctx_max + pos) * 128 + dd;
                    }
                    if (n + 8 <= No) *(u32x4*)dst = d;
                    else *(u32x2*)dst = u32x2{d[0], d[1]};      // N % 4 == 0: the ragged chunk holds exactly 4 columns
                }
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = m0 + VLY_FRAG_ROW(i);
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int n = n0 + VLY_FRAG_COL(i, j);
            if (n >= N) continue;
            f32x4 v = acc[i][j];
            if (bias) {
                const f32x4 b = *(const f32x4*)(bias + n);
                v += b;
            }
            if constexpr (EPI == VLY_EPI_QUICK_GELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = x_sigmoid(v[r], 1.702f);
            }
            if constexpr (EPI == VLY_EPI_RELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if constexpr (EPI == VLY_EPI_SWIGLU) {
                const float o0 = x_sigmoid(v[0], 1.f) * v[1];
                const float o1 = x_sigmoid(v[2], 1.f) * v[3];
                const size_t o = (size_t)m * ldc + (n >> 1);
                if constexpr (OUT == VLY_OUT_BF16) {
                    *(uint32_t*)((uint16_t*)Cv + o) = pack_h2(o0, o1);
                } else {
                    *(float2*)((float*)Cv + o) = make_float2(o0, o1);
                }
            } else {
                if (R) {
                    const f32x4 rr = *(const f32x4*)(R + (size_t)m * ldr + n);
                    v += rr;
                }
                const size_t o = (size_t)m * ldc + n;
                if constexpr (OUT == VLY_OUT_BF16) {
                    u32x2 pk;
                    pk[0] = pack_h2(v[0], v[1]);
                    pk[1] = pack_h2(v[2], v[3]);
                    *(u32x2*)((uint16_t*)Cv + o) = pk;
                } else {
                    *(f32x4*)((float*)Cv + o) = v;
                }
            }
        }
    }
}

#undef VLY_FRAG_ROW
#undef VLY_FRAG_COL
#undef EPI_NH

// Read one accumulator block where it is USED: the "a" constraint keeps the value in the accumulation registers up to this
// point (left alone, hipcc copies half of the 256 accumulators into VGPRs at the top of the epilogue and spills the K loop's
// per-lane offsets to scratch to make room — a scratch reload in the K loop is a vmcnt(0) in front of the LDS-DMA stream).
VLY_DEVICE f32x4 acc_read(const f32x4& a) {
    f32x4 v;
    asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
                 : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3])
                 : "a"(a[0]), "a"(a[1]), "a"(a[2]), "a"(a[3]));
    return v;
}
// the same block as (even columns, odd columns) register pairs: the SwiGLU epilogue's gates and ups, ready for packed math
VLY_DEVICE void acc_read_pairs(const f32x4& a, f32x2& even, f32x2& odd) {
    asm volatile("v_accvgpr_read_b32 %0, %4\n\tv_accvgpr_read_b32 %1, %5\n\tv_accvgpr_read_b32 %2, %6\n\tv_accvgpr_read_b32 %3, %7"
                 : "=v"(even[0]), "=v"(odd[0]), "=v"(even[1]), "=v"(odd[1])
                 : "a"(a[0]), "a"(a[1]), "a"(a[2]), "a"(a[3]));
}

// ================= persistent 4-wave kernel: the PIPE 8 loop running THROUGH tile boundaries ==========================
// One workgroup per CU walks tiles bid, bid + G, bid + 2G, ...  Measured on the one-tile-per-workgroup kernel
// (profiles/history/r02/r02_ab_4wave.txt, "noepi"): without its epilogue the ViT fc1 GEMM (K = 1024: 16 K tiles per output tile)
// runs 213 instead of 292 us — the stores of one round, the workgroup turn-around and the first loads of the next round
// are all exposed when every CU holds ONE workgroup (128 KB of LDS, 512 registers per wave).  Here
//   * the LOAD cursor runs two K tiles ahead of the COMPUTE cursor across tile boundaries (it owns the per-lane offsets and
//     recomputes them when it enters a new tile), so the K-loop buffers never drain between tiles;
//   * the epilogue works on registers only (v_permlane16/32_swap, gemm_kernel's `wide == 2` path): it needs neither the LDS
//     the next tile's K tiles are landing in nor a barrier, and its stores are in flight while the next tile's MFMAs run;
//   * past the last tile the load cursor re-requests the last K tile (valid addresses, free buffer): one code path, the
//     vmcnt bookkeeping never changes, 128 KB of redundant L2 reads per workgroup and launch.
// vmcnt and the stores: before barrier B the wave waits for "at most N1 operations outstanding".  Loads AND stores retire
// through vmcnt in issue order (tools/probes/vmcnt_store_order.hip, round 5), so the wait covers the K tile this barrier
// publishes — and every store issued before its N1 youngest loads: the unrolled schedule waits for its epilogue's stores here,
// the rolled one (round 5, `boundary`) issues the pieces first and counts the stores in.
// Round 5: the instantiations with bf16 outputs (all epilogues but the fused RoPE, no split-K) hold their accumulators BY NAME
// (mfma16_lit: a[4 b : 4 b + 3] spelled out in asm, no C++ accumulator values) — that is what allows the zero-free first K step
// and the rolling epilogue; the others (fp32 outputs, RoPE, SK) keep the C++ accumulators and the round-4 schedule.
// SK = true (tile hints 298 / 299 of vly_gemm_bf16_streamk, round 3): the same kernel with the REMAINDER ROUND split along K.
// The tiles past the last whole round of G workgroups (G = CUs) would cost a whole round for a fraction of the chip; instead
// each of them is cut into S equal K slices (S chosen by the host so that S x remainder fills whole rounds: 164 tiles x 3 =
// 492 units = 1.92 rounds of a third of a tile), the (slice, tile) units run FIRST, slice-major, one per workgroup and round,
// then every workgroup walks its whole tiles as before.  Slices are UNIFORM on purpose: all workgroups of a round walk the
// same K range in step, so tiles of one row / column still share their A / W panel in the XCD's L2 (contiguous stream-K
// ranges — every workgroup at its own k — measured 1.1-1.8x SLOWER than no split at all: profiles/history/r03/r03_p4_streamk.jsonl).
// Slices 0 .. S-2 are CONTRIBUTORS: fp32 accumulators to the unit's slab, agent-scope release, flag.  The last slice OWNS the
// tile: it waits for the S - 1 flags (their units ran in the same or an earlier round: no deadlock while the grid is
// resident, spins are bounded) and its epilogue adds the slabs to every accumulator block it reads — the accumulators
// themselves are only ever written by MFMAs (anything else and hipcc moves them out of the AGPRs).
// Hand-off protocol as gemm_sk_kernel's (gemm_streamk.hip); the K loop, staging and epilogues are this kernel's.
struct P4Cursor {                                          // position in a workgroup's iteration stream
    int tile, k, kend, u, r;                               // u: the pool unit being walked (or >= units: whole tiles, r = next round)
};
constexpr int P4_SK_MAX_WAYS = 8;                          // most slices per tile
constexpr unsigned P4_SK_SPIN_LIMIT = 1u << 24;
constexpr int P4_SK_ERR_FLAG = 4000;                       // index of the error word in the flag area (gemm_streamk.hip's)

template <int BM, int BN, int EPI, int OUT, bool SK>
__global__ void __launch_bounds__(256)
gemm_p4_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W, const float* __restrict__ bias,
               const float* __restrict__ R, void* __restrict__ Cv, int M, int N, int K, int lda, int ldw, int ldc, int ldr,
               int tiles_m, int tiles_n, int gm, RopeArgs rp, float* __restrict__ slabs, unsigned* __restrict__ flags,
               unsigned epoch, int sk_S) {
    constexpr int WM = BM / 2, WN = BN / 2, NT = 256;
    constexpr bool SKT = SK && BM != 256;       // the owner's epilogue adds slab terms (256-row tiles fold them into the accumulators instead)
    constexpr int MI = WM / 16, NI = WN / 16;
    constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;
    constexpr int PA = BM * 8 / NT, PW = BN * 8 / NT, NS = PA + PW;
    static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0 && NI % 4 == 0, "tile/threads mismatch");
    constexpr int T = MI * NI, BAR_AT = MI + NI + P8_BAR_GAP, GL1_START = BAR_AT + 2;
    constexpr int GL_STRIDE = (2 * T - GL1_START) / NS;
    constexpr int N1 = (T - GL1_START + GL_STRIDE - 1) / GL_STRIDE, N2 = NS - N1;
    constexpr int GL2_START = GL1_START + N1 * GL_STRIDE - T;
    static_assert(GL1_START < T && GL_STRIDE >= 1 && N2 >= 0 && GL2_START + (N2 - 1) * GL_STRIDE < T, "piece schedule");
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
#if VLY_P4_TIMING
    // anatomy builds (tools/p4_boundary_times.py): wave 0 of the first workgroups stamps s_memtime at the seams of every tile into LDS
    // (a global store would join the vmcnt queue the loop counts on) and copies them out at the end; `slabs` carries the buffer
    __shared__ unsigned long long tstamp[64];
    int tsn = 0;
#define VLY_STAMP()                                                                            \
    do {                                                                                       \
        if (!SK && threadIdx.x == 0 && tsn < 64) tstamp[tsn] = __builtin_readcyclecounter();  \
        ++tsn;                                                                                 \
    } while (0)
#else
#define VLY_STAMP() do {} while (0)
#endif

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    const int ntiles = tiles_m * tiles_n, G = (int)gridDim.x;
    const int nk = K / BK;
    __builtin_assume(nk >= 2);                                       // launcher: K >= 128 (a zero-trip K loop would make the accumulators a phi)
    const uint32_t wk = ldw < 0 ? (uint32_t)((N + 63) >> 6) * 4096u : (uint32_t)BK;
    // tile index -> (m0, n0): the mapping of gemm_kernel (XCD-contiguous runs, groups of gm m-tiles); tile t runs on
    // workgroup t % G and G % 8 == 0 whenever a workgroup owns more than one tile, so t & 7 is still its XCD
    auto tile_origin = [&](int t, int& m0, int& n0) {
        const int xcd = t & 7, qd = ntiles >> 3, rm = ntiles & 7;
        const int swz = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (t >> 3);
        const int gsz = gm * tiles_n, grp = swz / gsz, first = grp * gm;
        const int gh = min(gm, tiles_m - first), rr = swz - grp * gsz;
        m0 = (first + rr % gh) * BM;
        n0 = (rr / gh) * BN;
    };
    const __amdgpu_buffer_rsrc_t rsA = vly_rsrc(A), rsW = vly_rsrc(W);
    // ---- split-K schedule of the remainder round (SK): unit u = slice * rem8 + p is slice u / rem8 of pool tile full * G + p
    // (p < rem; rem8 = rem rounded up to 8, so that unit u, run by workgroup u % G, sits on tile p's XCD p & 7)
    [[maybe_unused]] const int sk_full = ntiles / G, sk_rem = ntiles - sk_full * G, sk_rem8 = (sk_rem + 7) & ~7;
    [[maybe_unused]] const int sk_units = sk_S * sk_rem8, sk_nks = (nk + sk_S - 1) / sk_S;      // K tiles per slice (host: every slice non-empty)
    [[maybe_unused]] auto cur_unit = [&](P4Cursor& c) {                  // c.u -> the next live unit at or after it, or the whole tiles
        while (c.u < sk_units && c.u % sk_rem8 >= sk_rem) c.u += G;
        if (c.u < sk_units) {
            const int sl = c.u / sk_rem8;
            c.tile = sk_full * G + (c.u - sl * sk_rem8);
            c.k = sl * sk_nks;
            c.kend = min(nk, c.k + sk_nks);
            return true;
        }
        if (c.r < sk_full) {
            c.tile = c.r * G + (int)blockIdx.x; ++c.r; c.k = 0; c.kend = nk;
            return true;
        }
        return false;
    };
    [[maybe_unused]] auto cur_next_seg = [&](P4Cursor& c) {              // false: the stream is used up (the cursor stays where it is)
        P4Cursor n = c;
        if (n.u < sk_units) n.u += G;
        if (!cur_unit(n)) return false;
        c = n;
        return true;
    };
    // ---- load cursor
    uint32_t voA[PA], voW[PW];
    auto set_offsets = [&](int m0, int n0) {
#pragma unroll
        for (int q = 0; q < PA; ++q) {
            const int sl = q * NT + tid, row = sl >> 3, cp = sl & 7;
            voA[q] = ((uint32_t)min(m0 + row, M - 1) * (uint32_t)lda + (uint32_t)((cp ^ (row & 7)) << 3)) * 2u;
        }
#pragma unroll
        for (int q = 0; q < PW; ++q) {
            const int sl = q * NT + tid, row = sl >> 3, cp = sl & 7;
            voW[q] = (w_row_off(min(n0 + row, N - 1), ldw) + (uint32_t)((cp ^ (row & 7)) << 3)) * 2u;
        }
    };
    int lt = (int)blockIdx.x, lk = 0;                                // tile / K tile the load cursor points at
    [[maybe_unused]] P4Cursor sc;                                    // (SK: the same cursor as a segment walker)
    if constexpr (SK) {
        sc.u = (int)blockIdx.x;
        sc.r = 0;
        if (!cur_unit(sc)) return;                                   // (uniform: the whole workgroup; nothing to do)
        lt = sc.tile;
        lk = sc.k;
    }
    // A piece = M0 (its LDS destination) + one buffer_load ... lds.  Written as two asm statements so that the M0 write
    // can sit in an EARLIER MFMA gap than the load (a 16x16x32 MFMA hides ~3 other issue slots; the builtin form puts
    // s_add m0 + s_nop + buffer_load into one gap).  Nothing else in this kernel touches M0 between the two.
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (uint32_t)wave * 1024u;
    auto piece_m0 = [&](int buf, int q) {
        const uint32_t dst = lds0 + (uint32_t)buf * STAGE + (q < PA ? (uint32_t)q * (NT * 16) : (uint32_t)A_BYTES + (uint32_t)(q - PA) * (NT * 16));
        asm volatile("s_mov_b32 m0, %0" ::"s"(dst) : "memory");
    };
    auto piece_ld = [&](int q) {
        if (q < PA) asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voA[q < PA ? q : 0]), "s"(rsA), "s"((uint32_t)lk * (BK * 2u)) : "memory");
        else asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voW[q >= PA ? q - PA : 0]), "s"(rsW), "s"((uint32_t)lk * wk * 2u) : "memory");
    };
    auto piece = [&](int buf, int q) {                               // both at once (prologue, dead waves)
        piece_m0(buf, q);
        asm volatile("s_nop 0" ::: "memory");
        piece_ld(q);
    };
    auto advance_load = [&]() {                                      // past the last tile: stay on its last K tile
        if constexpr (SK) {
            if (sc.k + 1 < sc.kend) { lk = ++sc.k; return; }
            if (!cur_next_seg(sc)) return;
            lt = sc.tile;
            lk = sc.k;
        } else {
            if (lk + 1 < nk) { ++lk; return; }
            if (lt + G >= ntiles) return;
            lt += G;
            lk = 0;
        }
        int m0, n0;
        tile_origin(lt, m0, n0);
        set_offsets(m0, n0);
    };
    int cm0, cn0;                                                    // compute cursor
    int ct = lt;
    [[maybe_unused]] P4Cursor cc;
    if constexpr (SK) cc = sc;
    tile_origin(ct, cm0, cn0);
    set_offsets(cm0, cn0);

    const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
    const int rdA = (wm0 + l15) * 128, rdW = A_BYTES + (wn0 + l15) * 128;
    const int sw0 = ((0 + g) ^ (l15 & 7)) << 4, sw1 = ((4 + g) ^ (l15 & 7)) << 4;
    bf16x8 a0[MI], w0[NI], a1[MI], w1[NI];
    auto rd_step1 = [&](const char* st) {
        return [&, st](int k) {
            if (k < NI) w1[k < NI ? k : 0] = *(const bf16x8*)(st + rdW + k * 2048 + sw1);
            else a1[k >= NI ? k - NI : 0] = *(const bf16x8*)(st + rdA + (k - NI) * 2048 + sw1);
        };
    };
    auto rd_step0 = [&](const char* st) {
        return [&, st](int k) {
            if (k < NI) w0[k < NI ? k : 0] = *(const bf16x8*)(st + rdW + k * 2048 + sw0);
            else a0[k >= NI ? k - NI : 0] = *(const bf16x8*)(st + rdA + (k - NI) * 2048 + sw0);
        };
    };
    // the BUILTIN waitcnt (0xc07f = lgkmcnt(0)): unlike an asm statement the compiler's own wait insertion sees it, and drops
    // the ten-odd "lgkmcnt(14)" it otherwise puts in front of the MFMA rows of a phase (each one an issue slot of the wave)
    auto bar_a = [](int) {
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
    };

    // ---- CHAIN (round 6, -DVLY_P4_CHAIN=1; the by-name instantiations): the two K steps of a block are issued BACK TO BACK on its
    // accumulator instead of 64 MFMAs apart.  tools/probes/mfma_order.hip: 2.37 against 2.23 PFLOP/s sustained with nothing but MFMAs —
    // the chip is power-limited and the chained form is the cheaper one (the second MFMA takes the first one's result as forwarded C).
    // Per block the products are added in the same order as before (K step 0, then 1, K tile by K tile): bit-identical results.
    // A K tile = phase 1: block rows 0 .. CH-1, row by row (all W fragments of the tile + the A fragments of those rows resident; the
    // A fragments of rows CH .. MI-1 are read meanwhile); phase 2: rows CH .. MI-1 COLUMN by column, so that a column's two W fragments
    // are free when its chains are through and take the next K tile's at once, while the first rows' A fragments are re-read too.
    // Same 32 fragment reads, barriers and pieces per K tile; no rolling boundary (it is built around the phase order).
    constexpr bool CHAIN = VLY_P4_CHAIN != 0 && VLY_P4_LIT != 0 && !SK && OUT == VLY_OUT_BF16 && EPI != VLY_EPI_QKV_ROPE;
    constexpr int CH = MI / 2, CT1 = CH * NI * 2, CT2 = (MI - CH) * NI * 2;                      // rows and MFMAs of the two phases
    constexpr int CR1 = 2 * (MI - CH), CBAR = CR1 + P8_BAR_GAP, CGL1 = CBAR + 2;                 // phase 1: reads, barrier A, first piece
    constexpr int CGLS = (CT1 + CT2 - CGL1) / NS, CN1 = (CT1 - CGL1 + CGLS - 1) / CGLS, CN2 = NS - CN1, CGL2 = CGL1 + CN1 * CGLS - CT1;
    static_assert(!CHAIN || (CGL1 < CT1 && CGLS >= 1 && CN2 >= 0 && CGL2 >= 0 && CGL2 + (CN2 - 1) * CGLS < CT2 && 1 + 2 * (2 * CH - 1) < CT2),
                  "chain piece schedule");
    auto chain_preload = [&](const char* st) {                       // what a K tile's phase 1 starts from: every W fragment, A rows 0 .. CH-1
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            w0[j] = *(const bf16x8*)(st + rdW + j * 2048 + sw0);
            w1[j] = *(const bf16x8*)(st + rdW + j * 2048 + sw1);
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            a0[i] = *(const bf16x8*)(st + rdA + i * 2048 + sw0);
            a1[i] = *(const bf16x8*)(st + rdA + i * 2048 + sw1);
        }
    };

    // ---- prologue: the first two K tiles of the stream
#pragma unroll
    for (int q = 0; q < NS; ++q) piece(0, q);
    advance_load();
#pragma unroll
    for (int q = 0; q < NS; ++q) piece(1, q);
    advance_load();
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NS) : "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (CHAIN) chain_preload(smem);
    else {
        auto r0 = rd_step0(smem);
#pragma unroll
        for (int k = 0; k < MI + NI; ++k) r0(k);
    }
    // C through a buffer descriptor that ends with row M - 1: a 16-byte store whose row is past M (the padded rows of the last
    // m-tile) or whose column block was flagged invalid (offset + 2 GB) is out of range and dropped by the hardware — no
    // exec masking, no 64-bit address arithmetic: one v_add + one buffer_store per 16 bytes (round 2's epilogue spent 12 of
    // its 84 instructions per 8 outputs on the two range checks).  launch_p4 guarantees M * ldc * 2 < 2^31 and No % 8 == 0.
    [[maybe_unused]] const __amdgpu_buffer_rsrc_t rsC =
        __builtin_amdgcn_make_buffer_rsrc(Cv, 0, (uint32_t)M * (uint32_t)ldc * (OUT == VLY_OUT_BF16 ? 2u : 4u), 0x00020000);
    int buf = 0;                                                     // buffer of the K tile being computed
    constexpr bool LIT = VLY_P4_LIT != 0 && !SK && OUT == VLY_OUT_BF16 && EPI != VLY_EPI_QKV_ROPE;     // the accumulators by name (mfma16_lit)
    constexpr bool ROLL = VLY_P4_ROLL != 0 && LIT && !CHAIN && ((VLY_P4_ROLL_MASK >> EPI) & 1) != 0;
    constexpr int NST_ = EPI == VLY_EPI_SWIGLU ? NI / 4 : NI / 2;    // 16-byte stores per fragment row of the bf16 epilogue
    static_assert(!ROLL || N1 + MI * NST_ <= 63, "vmcnt is a 6-bit count");
    if constexpr (LIT) asm volatile("" ::: VLY_ALL_AGPRS);           // the kernel owns a0 .. a255 (this is what makes the descriptor allocate them)
    f32x4 acc[MI][NI];                                               // (untouched, and optimised away, when LIT)
    bool wave_live = false;                                          // this wave's slab of the tile being computed lies inside the problem
    // first_c: the first K tile of a tile.  relaxed (ROLL): the first K tile after a boundary — the stream of vector-memory operations
    // is then [pieces of the K tile this barrier publishes] [the epilogue's stores] [N1 pieces], and vmcnt retires loads AND stores in
    // issue order (tools/probes/vmcnt_store_order.hip: 0 violations in 16.8 M trials), so "all but the N1 + stores youngest" waits for
    // the pieces and for none of the stores: their drain (~10 k clk for the 32 MB every CU bursts at once) gets two more K tiles.
    auto ktile = [&](auto first_c, bool relaxed = false) {       // one K tile = two phases
        constexpr bool FIRST = decltype(first_c)::value;
        const char* cur = smem + buf * STAGE;
        const char* nxt = smem + (buf ^ 1) * STAGE;
        // (a dead wave's fragment registers are never used; its reads are skipped with its MFMAs)
        __builtin_amdgcn_s_waitcnt(0xc07f);                      // the fragments of this phase: read >= 33 MFMAs ago (VLY_P4_LATE: >= 11)
        if constexpr (CHAIN) {
            if (wave_live) {
                // phase 1: rows 0 .. CH-1, row-major pairs; reads: a0 / a1 of rows CH .. MI-1 (the last reads of this buffer)
                phase_seq<CT1, CR1, 0, 1, CN1, CGL1, CGLS, 1, CBAR, 1, CN1, CGL1 - P4_M0_LEAD, CGLS>(
                    [&](int t) {
                        const int i = t / (2 * NI), j = (t % (2 * NI)) >> 1, st = t & 1;
                        if (FIRST && st == 0) mfma16_lit_zero(i * NI + j, w0[j], a0[i]);
                        else if (st == 0) mfma16_lit(i * NI + j, w0[j], a0[i]);
                        else mfma16_lit(i * NI + j, w1[j], a1[i]);
                    },
                    [&](int k) {
                        const int i = CH + (k >> 1);
                        if (k & 1) a1[i < MI ? i : 0] = *(const bf16x8*)(cur + rdA + i * 2048 + sw1);
                        else a0[i < MI ? i : 0] = *(const bf16x8*)(cur + rdA + i * 2048 + sw0);
                    },
                    [&](int q) { piece_ld(q); }, bar_a, [&](int q) { piece_m0(buf, q); });
            } else {
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int q = 0; q < CN1; ++q) piece(buf, q);
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(CN1) : "memory");
            __builtin_amdgcn_s_barrier();
            if (wave_live) {
                // phase 2: rows CH .. MI-1, column-major pairs; reads from the NEXT K tile: a0 / a1 of rows 0 .. CH-1 (one per two MFMAs), and
                // a column's w0 / w1 the moment its last chain is through
                constexpr int CW = 2 * (MI - CH);                    // MFMAs per column
                phase_seq<CT2, 2 * CH, 1, 2, CN2, CGL2, CGLS, NI, CW - 1, CW, CN2, CGL2 - P4_M0_LEAD, CGLS>(
                    [&](int t) {
                        const int j = t / CW, i = CH + ((t % CW) >> 1), st = t & 1;
                        if (FIRST && st == 0) mfma16_lit_zero(i * NI + j, w0[j], a0[i]);
                        else if (st == 0) mfma16_lit(i * NI + j, w0[j], a0[i]);
                        else mfma16_lit(i * NI + j, w1[j], a1[i]);
                    },
                    [&](int k) {
                        const int i = k >> 1;
                        if (k & 1) a1[i < MI ? i : 0] = *(const bf16x8*)(nxt + rdA + i * 2048 + sw1);
                        else a0[i < MI ? i : 0] = *(const bf16x8*)(nxt + rdA + i * 2048 + sw0);
                    },
                    [&](int q) { piece_ld(CN1 + q); },
                    [&](int j) {
                        w0[j < NI ? j : 0] = *(const bf16x8*)(nxt + rdW + j * 2048 + sw0);
                        w1[j < NI ? j : 0] = *(const bf16x8*)(nxt + rdW + j * 2048 + sw1);
                    },
                    [&](int q) { piece_m0(buf, CN1 + q); });
            } else {
#pragma unroll
                for (int q = 0; q < CN2; ++q) piece(buf, CN1 + q);
            }
            advance_load();
            buf ^= 1;
            return;
        }
        if (wave_live)
            phase_4w4<MI, NI, MI + NI, 0, 1, N1, GL1_START, GL_STRIDE, 1, BAR_AT, 1, N1, GL1_START - P4_M0_LEAD, GL_STRIDE, LIT ? (FIRST ? 3 : 2) : 0>(
                acc, a0, w0, rd_step1(cur), [&](int q) { piece_ld(q); }, bar_a, [&](int q) { piece_m0(buf, q); });
        else {
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int q = 0; q < N1; ++q) piece(buf, q);
        }
#if VLY_P4_LATE
        // Round 6: barrier B (the next K tile has landed everywhere) moves from the phase boundary INTO phase 2, and the reads of the next
        // K tile's first fragments behind it.  A buffer's fragments are then read in [B2_AT of the tile before, 16 of its own] — 0.35 of
        // a K tile instead of 0.75 — and its refill (from barrier A) has 1.6 K tiles to land instead of 1.3: the K loop's period was the
        // staging round trip (~2390 cycles: latency + 64 KB at ~36 B/clk per CU, profiles/r06/r06_p32_ablation_*.txt), not the MFMAs.
        constexpr int B2_AT = T - (MI + NI) - 12;
        constexpr int NB2 = B2_AT <= GL2_START ? 0 : (B2_AT - GL2_START + GL_STRIDE - 1) / GL_STRIDE < N2 ? (B2_AT - GL2_START + GL_STRIDE - 1) / GL_STRIDE : N2;
        static_assert(B2_AT > 0 && B2_AT + 1 + MI + NI <= T, "late barrier");
        auto bar_b = [&](int) {
            if (ROLL && relaxed) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N1 + MI * NST_ + NB2) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N1 + NB2) : "memory");
            __builtin_amdgcn_s_barrier();
        };
        if (wave_live)
            phase_4w4<MI, NI, MI + NI, B2_AT + 1, 1, N2, GL2_START, GL_STRIDE, 1, B2_AT, 1, N2, GL2_START - P4_M0_LEAD, GL_STRIDE, LIT ? 2 : 0>(
                acc, a1, w1, rd_step0(nxt), [&](int q) { piece_ld(N1 + q); }, bar_b, [&](int q) { piece_m0(buf, N1 + q); });
        else {
#pragma unroll
            for (int q = 0; q < NB2; ++q) piece(buf, N1 + q);
            bar_b(0);
#pragma unroll
            for (int q = NB2; q < N2; ++q) piece(buf, N1 + q);
        }
#else
        if (ROLL && relaxed) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N1 + MI * NST_) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N1) : "memory");
        __builtin_amdgcn_s_barrier();
        if (wave_live)
            phase_4w4<MI, NI, MI + NI, P8_RD2_START, P8_RD2_STRIDE, N2, GL2_START, GL_STRIDE, 0, 0, 1, N2, GL2_START - P4_M0_LEAD, GL_STRIDE, LIT ? 2 : 0>(
                acc, a1, w1, rd_step0(nxt), [&](int q) { piece_ld(N1 + q); }, [](int) {}, [&](int q) { piece_m0(buf, N1 + q); });
        else {
#pragma unroll
            for (int q = 0; q < N2; ++q) piece(buf, N1 + q);
        }
#endif
        advance_load();
        buf ^= 1;
    };
    [[maybe_unused]] int nfol = 0;                                   // SK owner: its epilogue adds the slabs of workgroups (x, j + 1 .. j + nfol)
    [[maybe_unused]] bool contributor = false;
    // SK owners (nfol = 1): the chain's running sum for the blocks of the current fragment row sits in sk_ld[]; the epilogue
    // re-issues a block's load for row i + 1 as soon as it has consumed it (sk_next), so the reads of one row fly under
    // the arithmetic of the row before.  (First version: one dependent load per block at its point of use = 128 exposed
    // round trips per tile, +93 us on the 13B gate|up; second: two followers x a whole row ahead = 96 registers, spilled.)
    [[maybe_unused]] f32x4 sk_ld[NI];
    [[maybe_unused]] auto sk_next = [&](int i, int j0, int n) {  // loads of blocks j0 .. j0 + n - 1 of fragment row i
        if (nfol == 0 || i >= MI) return;
        uint32_t lo = (uint32_t)((i * NI * NT + tid) * 16);      // (opaque: see the contributor's stores)
        asm volatile("" : "+v"(lo));
        const char* sl = (const char*)(slabs + (size_t)(cc.u - sk_rem8) * (BM * BN)) + lo;
#pragma unroll
        for (int j = 0; j < NI; ++j)
            if (j >= j0 && j < j0 + n) sk_ld[j] = *(const f32x4*)(sl + j * (NT * 16));
    };
    [[maybe_unused]] auto slab_term = [&](int, int j) { return nfol ? sk_ld[j] : f32x4{0.f, 0.f, 0.f, 0.f}; };
    // ---- the bf16 epilogue as (setup, row) so that the rolling boundary can weave its rows into the next tile's first K tile
    constexpr int NST = EPI == VLY_EPI_SWIGLU ? NI / 4 : NI / 2;                    // 16-byte stores per fragment row
    [[maybe_unused]] const int No = EPI == VLY_EPI_SWIGLU ? N >> 1 : N;
    [[maybe_unused]] uint32_t vo[NST];                           // byte offset of this lane's store s in fragment row i (advanced per row)
    [[maybe_unused]] const uint32_t rstep = (uint32_t)ldc * 32u; // 16 rows further down
    [[maybe_unused]] f32x4 bv[NI];                               // the bias of this lane's 4 columns per block: once per tile
    [[maybe_unused]] u32x4 held[NST];                            // a row's packed outputs, when their stores are deferred (boundary row 0)
    [[maybe_unused]] auto epi_setup = [&](int em0, int en0) {    // (em0, en0): origin of the tile being written
        if constexpr (OUT == VLY_OUT_BF16 && EPI != VLY_EPI_QKV_ROPE) {
            {
                const uint32_t rowb = (uint32_t)(em0 + wm0 + l15) * (uint32_t)ldc * 2u;
#pragma unroll
                for (int s = 0; s < NST; ++s) {
                    const int n = EPI == VLY_EPI_SWIGLU ? ((en0 + wn0) >> 1) + (4 * s + g) * 8
                                                        : en0 + wn0 + (2 * s + (g & 1)) * 16 + (g & 2) * 4;
                    vo[s] = rowb + (n + 8 <= No && !VLY_P4_DROPSTORE ? (uint32_t)n * 2u : 0x80000000u);
                }
            }
            if constexpr (EPI != VLY_EPI_SWIGLU) {
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int n = en0 + wn0 + j * 16 + g * 4;
                    bv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (bias) bv[j] = *(const f32x4*)(bias + min(n, N - 4));           // uniform branch, clamped (dropped at the store)
                }
            }
        }
    };
    // hook(k), k = 0 .. 2 NI - 1: sixteen evenly spread points of a row at which the rolling boundary issues ONE MFMA of the next
    // tile each (an MFMA occupies the matrix core for ~16 clk and the wave's issue for one slot: placed between ~12 VALU
    // instructions it runs entirely in their shadow; sixteen back to back would block the wave's issue for 240 clk)
    [[maybe_unused]] auto epi_row = [&](int i, auto&& hook, auto defer_c) {    // fragment row i of the finished tile: read, activate, pack, store
        constexpr bool DEFER = decltype(defer_c)::value;         // keep the row's 16-byte pieces in held[] (epi_flush stores them)
        if constexpr (OUT == VLY_OUT_BF16 && EPI != VLY_EPI_QKV_ROPE) {
            if constexpr (EPI == VLY_EPI_SWIGLU) {
#pragma unroll
                for (int jq = 0; jq < NI / 4; ++jq) {
                    // gate = even columns, up = odd columns (row-interleaved weights); the four blocks of one 16-byte store go
                    // through every step of x_sigmoid2(gate, 1) * up together (see the quick_gelu branch)
                    f32x2 gt[4], up[4], e[4];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) {
                        if constexpr (SKT) {
                            const f32x4 v = acc_read(acc[i][4 * jq + jj]) + slab_term(i, 4 * jq + jj);
                            gt[jj] = f32x2{v[0], v[2]};
                            up[jj] = f32x2{v[1], v[3]};
                        } else if constexpr (LIT) {
                            const f32x4 v = acc_read_lit(i * NI + 4 * jq + jj);
                            gt[jj] = f32x2{v[0], v[2]};
                            up[jj] = f32x2{v[1], v[3]};
                        } else acc_read_pairs(acc[i][4 * jq + jj], gt[jj], up[jj]);
                    }
                    if constexpr (SKT) sk_next(i + 1, 4 * jq, 4);
                    hook(8 * jq + 0);
#pragma unroll
                    for (int q = 0; q < 4; ++q) e[q] = gt[q] * -1.4426950408889634f;
                    hook(8 * jq + 1);
#pragma unroll
                    for (int q = 0; q < 2; ++q) e[q] = f32x2{__builtin_amdgcn_exp2f(e[q][0]), __builtin_amdgcn_exp2f(e[q][1])};
                    hook(8 * jq + 2);
#pragma unroll
                    for (int q = 2; q < 4; ++q) e[q] = f32x2{__builtin_amdgcn_exp2f(e[q][0]), __builtin_amdgcn_exp2f(e[q][1])};
                    hook(8 * jq + 3);
#pragma unroll
                    for (int q = 0; q < 4; ++q) e[q] += 1.f;
#pragma unroll
                    for (int q = 0; q < 2; ++q) e[q] = f32x2{__builtin_amdgcn_rcpf(e[q][0]), __builtin_amdgcn_rcpf(e[q][1])};
                    hook(8 * jq + 4);
#pragma unroll
                    for (int q = 2; q < 4; ++q) e[q] = f32x2{__builtin_amdgcn_rcpf(e[q][0]), __builtin_amdgcn_rcpf(e[q][1])};
                    hook(8 * jq + 5);
#pragma unroll
                    for (int q = 0; q < 4; ++q) gt[q] *= e[q];
#pragma unroll
                    for (int q = 0; q < 4; ++q) gt[q] *= up[q];
                    hook(8 * jq + 6);
                    uint32_t d[4];
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) d[jj] = pack_h2(gt[jj][0], gt[jj][1]);
                    const auto p01 = __builtin_amdgcn_permlane16_swap(d[0], d[1], false, false);
                    const auto p23 = __builtin_amdgcn_permlane16_swap(d[2], d[3], false, false);
                    const auto q0 = __builtin_amdgcn_permlane32_swap(p01[0], p23[0], false, false);
                    const auto q1 = __builtin_amdgcn_permlane32_swap(p01[1], p23[1], false, false);
                    if constexpr (DEFER) held[jq] = u32x4{q0[0], q1[0], q0[1], q1[1]};
                    else {
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{q0[0], q1[0], q0[1], q1[1]}, rsC, vo[jq], 0, 0);
                        vo[jq] += rstep;
                    }
                    hook(8 * jq + 7);
                }
            } else {
#pragma unroll
                for (int jp = 0; jp < NI / 2; ++jp) {
                    // the four register pairs of two blocks go through every step TOGETHER: between a packed op and the
                    // transcendental that consumes it (and back) the hardware wants a wait state, which independent
                    // work fills (one chain at a time cost 487 s_nop per tile)
                    f32x4 v0, v1;
                    if constexpr (LIT) {
                        v0 = acc_read_lit(i * NI + 2 * jp) + bv[2 * jp];
                        v1 = acc_read_lit(i * NI + 2 * jp + 1) + bv[2 * jp + 1];
                    } else {
                        v0 = acc_read(acc[i][2 * jp]) + bv[2 * jp];
                        v1 = acc_read(acc[i][2 * jp + 1]) + bv[2 * jp + 1];
                    }
                    if constexpr (SKT) {
                        v0 += slab_term(i, 2 * jp);
                        v1 += slab_term(i, 2 * jp + 1);
                        sk_next(i + 1, 2 * jp, 2);
                    }
                    f32x2 x[4] = {{v0[0], v0[1]}, {v0[2], v0[3]}, {v1[0], v1[1]}, {v1[2], v1[3]}};
                    hook(4 * jp + 0);
                    if constexpr (EPI == VLY_EPI_QUICK_GELU) {
                        constexpr float c = -1.4426950408889634f * 1.702f;           // x_sigmoid2's arithmetic, step by step
                        f32x2 e[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) e[q] = x[q] * c;
#pragma unroll
                        for (int q = 0; q < 4; ++q) e[q] = f32x2{__builtin_amdgcn_exp2f(e[q][0]), __builtin_amdgcn_exp2f(e[q][1])};
                        hook(4 * jp + 1);
#pragma unroll
                        for (int q = 0; q < 4; ++q) e[q] += 1.f;
#pragma unroll
                        for (int q = 0; q < 4; ++q) e[q] = f32x2{__builtin_amdgcn_rcpf(e[q][0]), __builtin_amdgcn_rcpf(e[q][1])};
                        hook(4 * jp + 2);
#pragma unroll
                        for (int q = 0; q < 4; ++q) x[q] *= e[q];
                    } else {
                        hook(4 * jp + 1);
                    }
                    if constexpr (EPI == VLY_EPI_RELU) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) x[q] = f32x2{fmaxf(x[q][0], 0.f), fmaxf(x[q][1], 0.f)};
                    }
                    u32x2 pk[2];
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        pk[jj][0] = pack_h2(x[2 * jj][0], x[2 * jj][1]);
                        pk[jj][1] = pack_h2(x[2 * jj + 1][0], x[2 * jj + 1][1]);
                    }
                    if constexpr (EPI != VLY_EPI_QUICK_GELU) hook(4 * jp + 2);
                    const auto s0 = __builtin_amdgcn_permlane16_swap(pk[0][0], pk[1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(pk[0][1], pk[1][1], false, false);
                    if constexpr (DEFER) held[jp] = u32x4{s0[0], s1[0], s0[1], s1[1]};
                    else {
                        __builtin_amdgcn_raw_buffer_store_b128(u32x4{s0[0], s1[0], s0[1], s1[1]}, rsC, vo[jp], 0, 0);
                        vo[jp] += rstep;
                    }
                    hook(4 * jp + 3);
                }
            }
        }
    };
    [[maybe_unused]] auto epi_flush = [&]() {                    // the deferred row's stores
#pragma unroll
        for (int q = 0; q < NST; ++q) {
            __builtin_amdgcn_raw_buffer_store_b128(held[q], rsC, vo[q], 0, 0);
            vo[q] += rstep;
        }
    };
    // ---- ROLL: the tile boundary = the NEXT tile's first K tile, computed fragment row by fragment row, with the FINISHED tile's
    // epilogue woven in.  What an epilogue costs (profiles/r05/r05_boundary_ab_1.txt, r05_store_burst.txt): the memory system takes a
    // 32 MB burst of stores (every CU's 128 KB at once) at ~6 TB/s = 5.4 us whatever the lane -> address pattern, so a plain epilogue is
    // bound by the ISSUE of its 32 stores per wave (~5.4 k clk, the wave stalled in the store queue), an activation epilogue by its
    // VALU (~10 k clk, stores hidden behind it); either way the matrix cores idle.  The accumulators cannot be double-buffered (256
    // of 512 registers), but a block is free the moment the epilogue has READ it: row r of the epilogue (8 blocks read, activated,
    // packed, stored) is followed by row r of the next tile's K tile 0 — 8 MFMAs with C = 0 into the blocks just read, then the 8
    // of K step 1 — so the tile's first 128 MFMAs execute in the shadow of the stores' issue stalls / the activation's VALU.
    // Per block the order of accumulation is unchanged (K step 0, then 1): results are bit-identical to the phase order.
    // Buffers and counters: the step-0 fragments of the new tile's K tile 0 were read by the last regular phase (they are kept across
    // the boundary now); its step-1 fragments are read first thing, then barrier A releases that buffer and the 16 pieces of K tile 2
    // are spread over the rows as in a regular K tile.  Every wave waits for its own pieces of K tile 1 (vmcnt(0)) BEFORE its first
    // store — no store is ever older than a load someone still waits for — and barrier A publishes them: no barrier B in here.  The
    // next K tile's step-0 fragments are re-read row by row (A) and behind the last row's K step 0 (W).
    [[maybe_unused]] auto boundary = [&](int em0, int en0) {
        if constexpr (ROLL) {
            // (a wave whose slab of the NEW tile lies outside the problem computes this one K tile anyway: its accumulators are never
            // stored — the descriptor drops them — and a branch per row would be eight 256-register merge points for the allocator)
            const char* cur = smem + buf * STAGE;                // K tile 0 of the new tile
            const char* nxt = smem + (buf ^ 1) * STAGE;          // its K tile 1
            constexpr int SLOTS = 2 * NI;                        // MFMAs per fragment row
            static_assert(NS <= SLOTS, "one piece per hook point of row 0");
            __builtin_amdgcn_s_waitcnt(0xc07f);                  // step-0 fragments (read by the last phase) have landed
            {
                auto r1 = rd_step1(cur);
#pragma unroll
                for (int k = 0; k < MI + NI; ++k) r1(k);
            }
            if (!wave_live) {                                    // this wave sat the finished tile out (its slab lay outside the problem):
                auto r0c = rd_step0(cur);                        // the dead branch of the last phase did not read the new tile's first fragments
#pragma unroll
                for (int k = 0; k < MI + NI; ++k) r0c(k);
            }
            epi_setup(em0, en0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my pieces of K tile 1, and the bias
            if constexpr (EPI != VLY_EPI_SWIGLU) {               // (hipcc's own wait for the bias loads lands HERE, not behind the pieces below)
#pragma unroll
                for (int j = 0; j < NI; ++j) asm volatile("" : "+v"(bv[j]));
            }
            bar_a(0);                                            // everyone has read K tile 0 and waited for its pieces of K tile 1
            auto r0n = rd_step0(nxt);
            // MFMA number 16 rr + t of the boundary (row rr: t < 8 K step 0 with C = 0, t >= 8 K step 1) and what hangs behind it: the
            // reload of a0[rr] once the row's K step 0 is through, and behind the LAST row's K step 1 the reload of w0
            auto emit = [&](int rr, int t) {
                const int j = t % NI;
                __builtin_amdgcn_sched_barrier(0);
                if (t < NI) mfma16_lit_zero(rr * NI + j, w0[j], a0[rr]);
                else mfma16_lit(rr * NI + j, w1[j], a1[rr]);
                __builtin_amdgcn_sched_barrier(0);
                if (t == NI) {                                   // a0[rr] was last used by this row's K step 0
                    r0n(NI + rr);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (rr == MI - 1 && t >= NI) {                   // last row: w0[j] is free once its K step 0 is through
                    r0n(t - NI);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            // row 0 carries the NS pieces of K tile 2 at its hook points and keeps its stores back until they are all issued: no store
            // of this tile is older than a piece the next K tile's counted wait waits for (ktile, `relaxed`)
            epi_row(0, [&](int k) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < NS; ++q)
                    if (q * SLOTS / NS == k) piece(buf, q);      // SLOTS = 16 hook points, NS = 14 .. 16 pieces
                __builtin_amdgcn_sched_barrier(0);
            }, std::true_type{});
            __builtin_amdgcn_sched_barrier(0);
            epi_flush();
            // epilogue row r carries the MFMAs of row r - 1 (whose blocks it has read) at its sixteen hook points
#pragma unroll
            for (int r = 1; r < MI; ++r) {
                __builtin_amdgcn_sched_barrier(0);
                epi_row(r, [&](int k) { emit(r - 1, k); }, std::false_type{});
            }
#pragma unroll
            for (int t = 0; t < SLOTS; ++t) emit(MI - 1, t);
            __builtin_amdgcn_sched_barrier(0);
            advance_load();
            buf ^= 1;
        }
    };
    if constexpr (ROLL) {
        // ---- the rolled schedule: first K tile of the first tile, then { K tiles 1 .. nk - 1 ; boundary = epilogue + next tile's K tile 0 }
        // per tile, and the last tile's epilogue on its own.  One loop, one back edge, the accumulators written by asm MFMAs only.
        wave_live = __builtin_amdgcn_readfirstlane((cm0 + wm0 < M && cn0 + wn0 < N) ? 1 : 0) != 0;
        VLY_STAMP();
        ktile(std::true_type{});
        bool after_boundary = false;
        for (;;) {
            VLY_STAMP();                                             // (stamps: K loop from its second K tile | boundary | ...)
            ktile(std::false_type{}, after_boundary);
            for (int kt = 2; kt < nk; ++kt) ktile(std::false_type{});
            after_boundary = true;
            VLY_STAMP();
            if (ct + G >= ntiles) break;
            const int em0 = cm0, en0 = cn0;
            ct += G;
            tile_origin(ct, cm0, cn0);
            boundary(em0, en0);
            wave_live = __builtin_amdgcn_readfirstlane((cm0 + wm0 < M && cn0 + wn0 < N) ? 1 : 0) != 0;
        }
        epi_setup(cm0, cn0);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            __builtin_amdgcn_sched_barrier(0);
            epi_row(i, [](int) {}, std::false_type{});
        }
        VLY_STAMP();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if VLY_P4_TIMING
        VLY_STAMP();
        if (threadIdx.x == 0 && blockIdx.x < 64 && slabs) {
            unsigned long long* o = (unsigned long long*)slabs + (size_t)blockIdx.x * 65;
            o[0] = (unsigned long long)tsn;
            for (int q = 0; q < 64 && q < tsn; ++q) o[1 + q] = tstamp[q];
        }
#endif
        return;
    }
    for (;;) {
        VLY_STAMP();                                                 // (stamps of the unrolled schedule: K loop | epilogue | K loop | ...)
        wave_live = __builtin_amdgcn_readfirstlane((cm0 + wm0 < M && cn0 + wn0 < N) ? 1 : 0) != 0;
        if constexpr (!LIT) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        nfol = 0;
        contributor = false;
        {
            int kt = SK ? cc.k : 0;
            const int kend = SK ? cc.kend : nk;
            if constexpr (LIT) {                                     // (first K step with C = 0)
                ktile(std::true_type{});
                while (++kt < kend) ktile(std::false_type{});
            } else {
                do ktile(std::false_type{}); while (++kt < kend);
            }
        }
        VLY_STAMP();
        if constexpr (SK) {
            const bool pool = cc.u < sk_units;
            contributor = pool && cc.kend < nk;                      // slices 0 .. S-2
            const bool chained = pool && cc.k > 0;                   // slices 1 .. S-1 take over the running sum of slice - 1
            if (chained) {
                // Consume side of Guideline 16 R1: one lane polls the one flag relaxed, one agent acquire, barrier, plain loads.
                // The slices of a tile form a CHAIN — slice s publishes (its accumulators + the slab of slice s - 1) — so that
                // the owner's epilogue adds ONE slab whatever S is: it has 32 registers to spare for that, not 32 (S - 1).
                if (tid == 0) {
                    unsigned spins = 0;
                    while (__hip_atomic_load(flags + (cc.u - sk_rem8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                        __builtin_amdgcn_s_sleep(8);
                        if (++spins > P4_SK_SPIN_LIMIT) {            // never hang the GPU: flag the failure
                            __hip_atomic_store(flags + P4_SK_ERR_FLAG, 0xDEADu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            break;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                __syncthreads();
                nfol = 1;
                if constexpr (BM == 256) {
                    // 256-row tiles (round 4, hint 297): the chain's running sum goes INTO the accumulators, by the matrix cores —
                    // D = I_k . P_k + C with the exact f32-input MFMA (v_mfma_f32_16x16x4_f32, 1.0 x p: one rounding, that of
                    // acc + p) — so the owner's epilogue is the plain one.  With 256 accumulators per lane the epilogue had no 32
                    // registers for the slab terms of a fragment row (hipcc spilled them behind vmcnt(0) waits: 3-8 % SLOWER than
                    // no split, round 3); here the terms pass through registers the K loop's fragments have just left, four
                    // dwords per block: lane (g, l15) needs P[4 k + g][l15], which the contributor's lane (k, l15) stored as
                    // element g of its float4.  The accumulators are still only ever written by MFMAs.  256 x 32 cycles = 4 us per
                    // owner tile, once per workgroup and launch.
                    if (!contributor) {
                        const char* prevb = (const char*)(slabs + (size_t)(cc.u - sk_rem8) * (BM * BN));
                        float idk[4];
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) idk[kk] = (l15 == 4 * kk + g) ? 1.f : 0.f;
                        asm volatile("s_nop 1" : "+v"(idk[0]), "+v"(idk[1]), "+v"(idk[2]), "+v"(idk[3]));       // (VALU write -> MFMA operand)
#pragma unroll
                        for (int i = 0; i < MI; ++i) {
                            __builtin_amdgcn_sched_barrier(0);
                            uint32_t lo = (uint32_t)((i * NI * NT + (tid & ~63) + l15) * 16 + g * 4);     // (opaque: see the stores below)
                            asm volatile("" : "+v"(lo));
#pragma unroll
                            for (int jh = 0; jh < NI; jh += 4) {     // four blocks (16 loads) at a time
                                float pv[4][4];
#pragma unroll
                                for (int j = 0; j < 4; ++j)
#pragma unroll
                                    for (int kk = 0; kk < 4; ++kk) pv[j][kk] = *(const float*)(prevb + lo + (jh + j) * (NT * 16) + kk * 256);
                                // (asm with the accumulator block tied "+a": the builtin lets the allocator put the result in NEW
                                // accumulation registers, and on the SwiGLU instantiation it then spills accumulators to scratch; k
                                // outer, block inner: four independent accumulators between two MFMAs of one chain)
#pragma unroll
                                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                                    for (int j = 0; j < 4; ++j)
                                        asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i][jh + j]) : "v"(idk[kk]), "v"(pv[j][kk]));
                            }
                        }
                        asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");      // the last MFMAs' results before any v_accvgpr_read of the epilogue
                        nfol = 0;
                    }
                }
            }
            if (contributor) {
                // publish the partial sums (lane-linear float4 image, fully coalesced); no epilogue.
                // (The lane offset goes through an asm statement per fragment row: left to itself hipcc computes the 64 block
                // addresses of a lane ONCE, outside the persistent loop, and spills 200 registers to hold them.)
                // Publish side of R1: WRITE-THROUGH (sc1) 16-byte stores — a release fence would write back the XCD's whole
                // L2 —, every wave drains its stores, barrier, one lane stores the flag.
                const __amdgpu_buffer_rsrc_t rsS =
                    __builtin_amdgcn_make_buffer_rsrc(slabs + (size_t)cc.u * (BM * BN), 0, (uint32_t)(BM * BN * 4), 0x00020000);
                const float* prev = slabs + (size_t)(cc.u - sk_rem8) * (BM * BN);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    __builtin_amdgcn_sched_barrier(0);
                    uint32_t lo = (uint32_t)((i * NI * NT + tid) * 16);
                    asm volatile("" : "+v"(lo));
                    f32x4 pv[NI];
                    if (chained) {
#pragma unroll
                        for (int j = 0; j < NI; ++j) pv[j] = *(const f32x4*)((const char*)prev + lo + j * (NT * 16));
                    } else {
#pragma unroll
                        for (int j = 0; j < NI; ++j) pv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc_read(acc[i][j]) + pv[j]), rsS, lo + j * (NT * 16), 0, 16);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_store(flags + cc.u, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                nfol = 0;
            }
            if constexpr (BM == 256) {
                // where the three paths (owner / owner after the add / contributor) meet, the accumulators ARE in the accumulation
                // registers; say so (left alone, hipcc resolves the merge in VGPRs on the SwiGLU instantiation: 353 spills)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) asm volatile("" : "+a"(acc[i][j]));
            }
        }
        if constexpr (SKT) sk_next(0, 0, NI);
        // ---- epilogue on registers; lane holds C[m][n .. n+3], m = .. + l15, n = .. + 4*g.  The next tile's first fragments
        // are NOT kept across it (they are re-read below): 64 more registers for the epilogue, one LDS round trip per tile
        if (!(SK && contributor)) {
        if constexpr (OUT == VLY_OUT_BF16 && EPI == VLY_EPI_QKV_ROPE) {
            // RoPE + KV append on registers: a wave's 128 columns are ONE head (cn0 + wn0 is a multiple of 128),
            // the rotation partner of column c < 64 is c + 64 = block j + 4 of the SAME lane.  Same arithmetic
            // on the same bf16-rounded projections as rope_kv_kernel -> identical bits (tests compare them).
            static_assert(NI == 8, "one head per wave");
            static_assert(!SK, "the fused RoPE epilogue has no stream-K form");
            const int nb = cn0 + wn0, Hq = rp.heads * 128, sect = nb / Hq, head = (nb - sect * Hq) >> 7;   // wave-uniform
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                __builtin_amdgcn_sched_barrier(0);                   // row by row: keeps the accumulator reads from piling up
                const int m = cm0 + wm0 + i * 16 + l15;
                uint16_t* crow = (uint16_t*)Cv + (size_t)min(m, M - 1) * ldc;
                const int mc = min(m, M - 1), bq = mc / rp.S, pos = rp.past + (mc - bq * rp.S);
                u32x2 pk[NI];
#pragma unroll
                for (int jl = 0; jl < 4; ++jl) {
                    const f32x4 lo = acc_read(acc[i][jl]), hi = acc_read(acc[i][jl + 4]);
                    const uint32_t l0 = pack_h2(lo[0], lo[1]), l1 = pack_h2(lo[2], lo[3]);
                    const uint32_t h0 = pack_h2(hi[0], hi[1]), h1 = pack_h2(hi[2], hi[3]);
                    if (sect < 2) {
                        const f32x4 c = *(const f32x4*)(rp.cos_t + (size_t)pos * 64 + jl * 16 + g * 4);
                        const f32x4 sn = *(const f32x4*)(rp.sin_t + (size_t)pos * 64 + jl * 16 + g * 4);
                        const float xl[4] = {h_lo(l0), h_hi(l0), h_lo(l1),
                                             h_hi(l1)};
                        const float xh[4] = {h_lo(h0), h_hi(h0), h_lo(h1),
                                             h_hi(h1)};
                        pk[jl][0] = pack_h2(rope_rot(xl[0], xh[0], c[0], sn[0], -1.f), rope_rot(xl[1], xh[1], c[1], sn[1], -1.f));
                        pk[jl][1] = pack_h2(rope_rot(xl[2], xh[2], c[2], sn[2], -1.f), rope_rot(xl[3], xh[3], c[3], sn[3], -1.f));
                        pk[jl + 4][0] = pack_h2(rope_rot(xh[0], xl[0], c[0], sn[0], 1.f), rope_rot(xh[1], xl[1], c[1], sn[1], 1.f));
                        pk[jl + 4][1] = pack_h2(rope_rot(xh[2], xl[2], c[2], sn[2], 1.f), rope_rot(xh[3], xl[3], c[3], sn[3], 1.f));
                    } else {
                        pk[jl] = u32x2{l0, l1};
                        pk[jl + 4] = u32x2{h0, h1};
                    }
                }
                uint16_t* drow = sect == 0 ? crow + nb
                                           : (sect == 1 ? rp.kc : rp.vc) + (((size_t)bq * rp.heads + head) * rp.ctx_max + pos) * 128;
#pragma unroll
                for (int jp = 0; jp < NI / 2; ++jp) {
                    const auto s0 = __builtin_amdgcn_permlane16_swap(pk[2 * jp][0], pk[2 * jp + 1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(pk[2 * jp][1], pk[2 * jp + 1][1], false, false);
                    const u32x4 o = u32x4{s0[0], s1[0], s0[1], s1[1]};
                    const int dd = (2 * jp + (g & 1)) * 16 + (g & 2) * 4;
                    if (m < M && nb < N) *(u32x4*)(drow + dd) = o;
                }
            }
        } else if constexpr (OUT == VLY_OUT_BF16) {
            epi_setup(cm0, cn0);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                __builtin_amdgcn_sched_barrier(0);                   // row by row: keeps the accumulator reads from piling up
                epi_row(i, [](int) {}, std::false_type{});
            }
        }
        if constexpr (OUT == VLY_OUT_F32) {                            // 16 bytes per lane already: 64 contiguous bytes per row and block
            static_assert(EPI == VLY_EPI_NONE, "fp32 outputs carry no activation");
            f32x4 bv[NI];
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int n = cn0 + wn0 + j * 16 + g * 4;
                bv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (bias) bv[j] = *(const f32x4*)(bias + min(n, N - 4));
            }
            const int ncol = cn0 + wn0 + g * 4;                      // column of block 0; block j adds 16 j
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                __builtin_amdgcn_sched_barrier(0);
                [[maybe_unused]] f32x4 skc[NI];                      // (SK owners: this row's slab terms; the next row's loads leave now)
                if constexpr (SKT) {
#pragma unroll
                    for (int j = 0; j < NI; ++j) skc[j] = slab_term(i, j);
                    sk_next(i + 1, 0, NI);
                }
                const int m = cm0 + wm0 + i * 16 + l15;
                if (m < M) {
                    float* crow = (float*)Cv + (size_t)m * ldc + ncol;
                    if (R) {
                        const float* rrow = R + (size_t)m * ldr + ncol;
#pragma unroll
                        for (int j = 0; j < NI; ++j)
                            if (ncol + j * 16 < N) {
                                f32x4 v = acc_read(acc[i][j]) + bv[j] + *(const f32x4*)(rrow + j * 16);
                                if constexpr (SKT) v += skc[j];
                                *(f32x4*)(crow + j * 16) = v;
                            }
                    } else {
#pragma unroll
                        for (int j = 0; j < NI; ++j)
                            if (ncol + j * 16 < N) {
                                f32x4 v = acc_read(acc[i][j]) + bv[j];
                                if constexpr (SKT) v += skc[j];
                                *(f32x4*)(crow + j * 16) = v;
                            }
                    }
                }
            }
        }
        }
        if constexpr (SK) {
            if (!cur_next_seg(cc)) break;
            ct = cc.tile;
        } else {
            if (ct + G >= ntiles) break;
            ct += G;
        }
        tile_origin(ct, cm0, cn0);
        if constexpr (CHAIN) chain_preload(smem + buf * STAGE);
        else {   // K step 0 of the next tile's first K tile: it landed before the last barrier B (same buffer rotation)
            auto r0 = rd_step0(smem + buf * STAGE);
#pragma unroll
            for (int k = 0; k < MI + NI; ++k) r0(k);
        }
    }
    VLY_STAMP();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // no LDS-DMA may outlive the workgroup's LDS allocation
#if VLY_P4_TIMING
    VLY_STAMP();
    if (!SK && threadIdx.x == 0 && blockIdx.x < 64 && slabs) {
        unsigned long long* o = (unsigned long long*)slabs + (size_t)blockIdx.x * 65;
        o[0] = (unsigned long long)tsn;
        for (int q = 0; q < 64 && q < tsn; ++q) o[1 + q] = tstamp[q];
    }
#endif
}
#undef VLY_STAMP

#if VLY_P4_TIMING
static void* vlydbg_p4_buffer() {
    static void* buf = [] { void* p = nullptr; (void)hipMalloc(&p, 64 * 65 * 8); (void)hipMemset(p, 0, 64 * 65 * 8); return p; }();
    return buf;
}
extern "C" int vlydbg_p4_timing_read(unsigned long long* host) {        // 64 workgroups x (count, 64 stamps)
    return (int)hipMemcpy(host, vlydbg_p4_buffer(), 64 * 65 * 8, hipMemcpyDeviceToHost);
}
#endif
struct P4SkArgs {                                          // split-K remainder form (vly_gemm_bf16_streamk, tile hints 298 / 299)
    float* slabs;
    unsigned* flags;
    unsigned epoch;
    int slab_cap;                                          // slabs the workspace holds
};

template <int BM, int BN, bool SK = false>
int launch_p4(const void* A, const void* W, const float* bias, const float* R, void* C, int M, int N, int K, int lda, int ldw,
              int ldc, int ldr, int epi, int out, hipStream_t st, const RopeArgs* rope = nullptr, const P4SkArgs* sk = nullptr) {
    const int tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
    const int gm = vly_tile_group_height(M, N, K, tm, tn, BM, BN, 1);
    // bf16 outputs leave as 16-byte buffer stores clipped by the descriptor: aligned rows, whole 8-column chunks, < 2 GB
    const int No = epi == VLY_EPI_SWIGLU ? N >> 1 : N;
    const int vec_ok = (out == VLY_OUT_BF16 && ldc % 8 == 0 && ((uintptr_t)C & 15) == 0 && !R && No % 8 == 0 &&
                        (size_t)M * (size_t)ldc * 2 < ((size_t)1 << 31)) ? 1 : 0;
    if (out == VLY_OUT_BF16 && !vec_ok) return 1;                       // caller falls back to the one-tile-per-workgroup kernel
    static const int cus = [] {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (getenv("VLY_P4_GRID")) n = atoi(getenv("VLY_P4_GRID"));
        return n > 0 ? n / 8 * 8 : 256;                                // a multiple of the XCD count (tile -> XCD mapping)
    }();
    const int tiles = tm * tn;
    // SK: slices per remainder tile — the S that packs S x rem8 units into the fewest rounds per slice (1 = no split), within
    // the slab capacity, at least two K tiles per slice, +6 % of a tile per extra slice for the hand-off chain and the shallower
    // loop (measured: 7B gate|up on 192-row tiles 206 / 216 / 242 us at S = 2 / 4 / 8, profiles/history/r03/r03_p4_splitk_slices.jsonl)
    int sk_S = 1;
    if constexpr (SK) {
        const int rem = tiles % cus, rem8 = (rem + 7) & ~7, nk = K / BK;
        float best = 1e30f;
        for (int S = 1; S <= P4_SK_MAX_WAYS && rem > 0; ++S) {
            if (S > 1 && ((S - 1) * rem8 > sk->slab_cap || nk / S < 2 || (S - 1) * ((nk + S - 1) / S) >= nk)) continue;
            const float cost = (float)((S * rem8 + cus - 1) / cus) / (float)S + 0.06f * (float)(S - 1);
            if (cost < best - 1e-6f) { best = cost; sk_S = S; }
        }
        static const int pin = [] { const char* e = getenv("VLY_P4_SK_S"); return e ? atoi(e) : 0; }();      // (A/B runs)
        if (pin > 0 && rem > 0 && (pin == 1 || ((pin - 1) * rem8 <= sk->slab_cap && nk / pin >= 2))) sk_S = pin;
    }
    dim3 grid(SK || tiles >= cus ? cus : tiles), block(256);           // SK: every CU takes its share of the units
#if VLY_P4_TIMING
    P4SkArgs dbg{(float*)vlydbg_p4_buffer(), nullptr, 0u, 0};
    if (!sk) sk = &dbg;
#endif
#define VLY_P4_LAUNCH(E, O)                                                                                                  \
    hipLaunchKernelGGL((gemm_p4_kernel<BM, BN, E, O, SK>), grid, block, 0, st, (const uint16_t*)A, (const uint16_t*)W, bias, R, C, M, \
                       N, K, lda, ldw, ldc, ldr, tm, tn, gm, rope ? *rope : RopeArgs{}, sk ? sk->slabs : nullptr,                 \
                       sk ? sk->flags : nullptr, sk ? sk->epoch : 0u, sk_S)
    if (epi == VLY_EPI_NONE && out == VLY_OUT_BF16) VLY_P4_LAUNCH(VLY_EPI_NONE, VLY_OUT_BF16);
    else if (epi == VLY_EPI_NONE && out == VLY_OUT_F32) VLY_P4_LAUNCH(VLY_EPI_NONE, VLY_OUT_F32);
    else if (epi == VLY_EPI_QUICK_GELU && out == VLY_OUT_BF16) VLY_P4_LAUNCH(VLY_EPI_QUICK_GELU, VLY_OUT_BF16);
    else if (epi == VLY_EPI_SWIGLU && out == VLY_OUT_BF16) VLY_P4_LAUNCH(VLY_EPI_SWIGLU, VLY_OUT_BF16);
    else if (epi == VLY_EPI_RELU && out == VLY_OUT_BF16) VLY_P4_LAUNCH(VLY_EPI_RELU, VLY_OUT_BF16);
    else if (!SK && epi == VLY_EPI_QKV_ROPE && out == VLY_OUT_BF16 && rope) {
        if constexpr (!SK) VLY_P4_LAUNCH(VLY_EPI_QKV_ROPE, VLY_OUT_BF16);
    } else {
        vly_set_error("vly_gemm_bf16: unsupported epilogue/out_dtype combination (%d,%d) for the persistent tiles", epi, out);
        return -22;
    }
#undef VLY_P4_LAUNCH
    return vly_check_launch("vly_gemm_bf16");
}

template <int BM, int BN, int WM, int WN, int PIPE>
int launch_tile(const void* A, const void* W, const float* bias, const float* R, void* C, int M, int N, int K,
                int lda, int ldw, int ldc, int ldr, int epi, int out, hipStream_t st, void* C2 = nullptr,
                const RopeArgs* rope = nullptr) {
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    const int tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
    constexpr int STAGE_B = (BM + BN) * 128 * ((PIPE == 6 || PIPE == 7) ? 3 : 2);
    const int gm = vly_tile_group_height(M, N, K, tm, tn, BM, BN, STAGE_B <= 80 * 1024 ? 2 : 1);
    // full-line stores through LDS need 16-byte aligned output rows (VLY_EPILOGUE=frag: A/B switch for measurements)
    static const bool frag_only = getenv("VLY_EPILOGUE") && !strcmp(getenv("VLY_EPILOGUE"), "frag");
    // VLY_EPILOGUE=lds keeps the LDS image for the 4-wave tiles too (A/B); their default is the register-swap epilogue
    static const bool lds_only = getenv("VLY_EPILOGUE") && !strcmp(getenv("VLY_EPILOGUE"), "lds");
    int wide = (out == VLY_OUT_BF16 && ldc % 8 == 0 && ((uintptr_t)C & 15) == 0 && !frag_only) ? 1 : 0;
    if (wide && PIPE == 8 && epi != VLY_EPI_QKV_ROPE && !lds_only) wide = 2;
    const int ksplit = C2 ? 2 : 1;
    dim3 grid(tm * tn * ksplit), block(NT);
#define VLY_GEMM_LAUNCH(E, O)                                                                         \
    hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN, E, O, PIPE>), grid, block, 0, st, (const uint16_t*)A,   \
                       (const uint16_t*)W, bias, R, C, M, N, K, lda, ldw, ldc, ldr, tm, tn, gm, wide, ksplit, C2, rope ? *rope : RopeArgs{})
    if (epi == VLY_EPI_NONE && out == VLY_OUT_BF16) VLY_GEMM_LAUNCH(VLY_EPI_NONE, VLY_OUT_BF16);
    else if (epi == VLY_EPI_NONE && out == VLY_OUT_F32) VLY_GEMM_LAUNCH(VLY_EPI_NONE, VLY_OUT_F32);
    else if (epi == VLY_EPI_QUICK_GELU && out == VLY_OUT_BF16) VLY_GEMM_LAUNCH(VLY_EPI_QUICK_GELU, VLY_OUT_BF16);
    else if (epi == VLY_EPI_SWIGLU && out == VLY_OUT_BF16) VLY_GEMM_LAUNCH(VLY_EPI_SWIGLU, VLY_OUT_BF16);
    else if (epi == VLY_EPI_RELU && out == VLY_OUT_BF16) VLY_GEMM_LAUNCH(VLY_EPI_RELU, VLY_OUT_BF16);
    else if (epi == VLY_EPI_QKV_ROPE && out == VLY_OUT_BF16 && rope) {
        // a head's two halves must sit in one tile (BN % 128 == 0) and the epilogue works on the LDS image (wide);
        // PIPE 1 / 3 (half-tile loops) are not instantiated for it
        if constexpr (BN % 128 == 0) {
            if (!wide) { vly_set_error("vly_gemm_bf16_qkv_rope: qkv rows must be 16-byte aligned"); return -22; }
            VLY_GEMM_LAUNCH(VLY_EPI_QKV_ROPE, VLY_OUT_BF16);
        } else {
            vly_set_error("vly_gemm_bf16_qkv_rope: this tile is not a multiple of 128 columns wide");
            return -22;
        }
    }
    else {
        vly_set_error("vly_gemm_bf16: unsupported epilogue/out_dtype combination (%d,%d)", epi, out);
        return -22;
    }
#undef VLY_GEMM_LAUNCH
    return vly_check_launch("vly_gemm_bf16");
}

}  // namespace

// Tile choice: 256x256 (8 waves, 1 block/CU) has the best inner-loop efficiency but needs enough
// tiles to fill 256 CUs; 128x128 (4 waves, 2 blocks/CU) quantises better on small problems.
static int pick_tile(int M, int N) {
    auto tiles = [&](int bm, int bn) { return (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn); };
    // enough 256 x 256 tiles for (nearly) every CU: the persistent 4-wave kernel — 1.3-1.5 PFLOP/s in its loop against
    // 0.6-1.2 of the others (same MFMA and K order as every other tile, so the static choice stays bit-identical across batch sizes)
    if (tiles(256, 256) >= 192) return 197;
    // modelled time = waves-of-blocks * per-tile work / relative efficiency
    auto cost = [&](int bm, int bn, int per_cu, double eff) {
        const long t = tiles(bm, bn);
        const long slots = 256L * per_cu;
        const long rounds = (t + slots - 1) / slots;
        return (double)rounds * per_cu * bm * bn / eff;
    };
    const double c256 = cost(256, 256, 1, 1.0);
    const double c128 = cost(128, 128, 2, 0.72);
    const double c2x1 = cost(256, 128, 1, 0.88);
    if (c256 <= c128 && c256 <= c2x1) return 1;
    return c2x1 <= c128 ? 3 : 2;
}

// The split-K-remainder form of the persistent kernel, for vly_gemm_bf16_streamk (gemm_streamk.hip): tile 297 / 298 / 299 = 256 / 224 / 192
// rows x 256 columns.  Returns 1 when the shape cannot take the persistent kernel's vector stores (the caller reports it).
__attribute__((visibility("hidden"))) int valley_p4_streamk(int tile, const void* A, const void* W, const float* bias, const float* R,
                                                            void* C, int M, int N, int K, int lda, int ldw, int ldc, int ldr, int epi,
                                                            int out, void* ws, size_t ws_bytes, unsigned epoch, hipStream_t st) {
    constexpr size_t FLAG_BYTES = 16384;                               // gemm_streamk.hip's workspace layout: flags, then slabs
    int dev = 0, cus = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    cus = cus > 0 ? cus / 8 * 8 : 256;
    const int bm = tile == 297 ? 256 : tile == 298 ? 224 : 192;
    if (!ws || ws_bytes < FLAG_BYTES + (size_t)bm * 256 * 4) {
        vly_set_error("vly_gemm_bf16_streamk: workspace too small (%zu bytes)", ws_bytes);
        return -22;
    }
    int cap = (int)((ws_bytes - FLAG_BYTES) / ((size_t)bm * 256 * 4));  // one slab per contributor unit; their flags share the 4000 words
    if (cap > P4_SK_ERR_FLAG - 8) cap = P4_SK_ERR_FLAG - 8;
    if (K < 128 || K % BK) { vly_set_error("vly_gemm_bf16_streamk: the persistent tiles need K >= 128"); return -22; }
    const P4SkArgs sk{(float*)((char*)ws + FLAG_BYTES), (unsigned*)ws, epoch, cap};
    int rc;
    // (256-row tiles, hint 297, round 4: the owner folds the chain's sum into its accumulators with f32 MFMAs before a plain
    // epilogue — with 256 accumulators per lane the epilogue has no 32 registers for slab terms; see the kernel)
    if (tile == 297) {
#if VLY_EXPERIMENTAL
        rc = launch_p4<256, 256, true>(A, W, bias, R, C, M, N, K, lda, ldw, ldc, ldr, epi, out, st, nullptr, &sk);
#else
        vly_set_error("vly_gemm_bf16_streamk: tile 297 (split-K remainder on 256-row tiles: measured behind 197) is in libvalley_hip_exp.so only");
        return -22;
#endif
    } else if (tile == 298) rc = launch_p4<224, 256, true>(A, W, bias, R, C, M, N, K, lda, ldw, ldc, ldr, epi, out, st, nullptr, &sk);
    else rc = launch_p4<192, 256, true>(A, W, bias, R, C, M, N, K, lda, ldw, ldc, ldr, epi, out, st, nullptr, &sk);
    if (rc == 1) {
        vly_set_error("vly_gemm_bf16_streamk: tile %d needs 16-byte aligned rows of whole 8-column chunks and no residual for bf16 outputs", tile);
        return -22;
    }
    return rc;
}

extern "C" int vly_gemm_tile_for(int M, int N) { return pick_tile(M, N); }

// gemm_p32.hip: the persistent kernel on the 32x32x16 MFMA (tile hint 397); 1 = the problem does not fit it
int valley_p32_gemm(const void* A, const void* W, const float* bias, const float* R, void* C, int M, int N, int K, int lda, int ldw, int ldc, int epi,
                    int out, hipStream_t st, int deep);
// gemm_p16.hip: the persistent kernel on the 16x16x32 MFMA in chains of two, parked whole-line stores (tile hint 497); 1 = does not fit
int valley_p16_gemm(const void* A, const void* W, const float* bias, const float* R, void* C, int M, int N, int K, int lda, int ldw, int ldc, int epi,
                    int out, hipStream_t st);

static int run_tile(int t, int tile_hint, const void* A, const void* W, const float* bias, const float* residual, void* C,
                    int M, int N, int K, int lda, int ldw, int ldc, int ldr, int epilogue, int out_dtype, hipStream_t st,
                    void* C2, const RopeArgs* rope = nullptr) {
#define VLY_TILE_ARGS A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, st, C2, rope
#define VLY_TILE_ARGS_RAW A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, st, C2, rope
#ifdef VLY_FEW_TILES                                        /* fast A/B builds (tools/ab_lib.py): only the tiles under study */
    switch (t) {
        case 497: {
            const int rc = rope || C2 ? 1 : valley_p16_gemm(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, epilogue, out_dtype, st);
            return rc == 1 ? run_tile(197, tile_hint, VLY_TILE_ARGS_RAW) : rc;
        }
        case 397:
        case 398: {
            const int rc = rope || C2 ? 1 : valley_p32_gemm(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, epilogue, out_dtype, st, t == 398);
            return rc == 1 ? run_tile(197, tile_hint, VLY_TILE_ARGS_RAW) : rc;
        }
        case 9: return launch_tile<256, 256, 64, 64, 0>(VLY_TILE_ARGS);
        case 97: return launch_tile<256, 256, 128, 128, 8>(VLY_TILE_ARGS);
        case 98: return launch_tile<224, 256, 112, 128, 8>(VLY_TILE_ARGS);
        case 197: return launch_p4<256, 256>(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, st);
        case 198: return launch_p4<224, 256>(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, st);
        case 199: return launch_p4<192, 256>(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, st);
        default: vly_set_error("vly_gemm_bf16: tile_hint %d is not in this VLY_FEW_TILES build", tile_hint); return -22;
    }
#else
    switch (t) {                                          // 1..9: 2-stage loop; 51..55: role split over the 2-stage buffers; 7x / 8x: three stages
        case 1: return launch_tile<256, 256, 128, 64, 0>(VLY_TILE_ARGS);
        case 2: return launch_tile<128, 128, 64, 64, 0>(VLY_TILE_ARGS);
        case 3: return launch_tile<256, 128, 64, 64, 0>(VLY_TILE_ARGS);
        case 4: return launch_tile<128, 256, 64, 64, 0>(VLY_TILE_ARGS);
        case 5: return launch_tile<192, 256, 96, 64, 0>(VLY_TILE_ARGS);
        case 73: return launch_tile<256, 128, 64, 64, 6>(VLY_TILE_ARGS);
        case 74: return launch_tile<128, 256, 64, 64, 6>(VLY_TILE_ARGS);
        case 76: return launch_tile<192, 192, 96, 48, 6>(VLY_TILE_ARGS);
        case 6: return launch_tile<192, 192, 96, 48, 0>(VLY_TILE_ARGS);
        case 83: return launch_tile<256, 128, 64, 64, 7>(VLY_TILE_ARGS);
        case 84: return launch_tile<128, 256, 64, 64, 7>(VLY_TILE_ARGS);
        case 86: return launch_tile<192, 192, 96, 48, 7>(VLY_TILE_ARGS);
        // 128x192: 40 KB per stage, two 2-stage workgroups per CU (each covers the other's prologue / epilogue)
        case 7: return launch_tile<128, 192, 64, 48, 0>(VLY_TILE_ARGS);
        case 8: return launch_tile<192, 128, 96, 32, 0>(VLY_TILE_ARGS);
        // 16 waves (4 x 4): four waves per SIMD hide LDS / barrier latency without the role split
        case 9: return launch_tile<256, 256, 64, 64, 0>(VLY_TILE_ARGS);
        case 93: return launch_tile<256, 128, 64, 32, 6>(VLY_TILE_ARGS);
        case 94: return launch_tile<128, 256, 32, 64, 6>(VLY_TILE_ARGS);
        case 497: {                                         // persistent, 16x16x32 in chains of two, parked whole-line stores (gemm_p16.hip)
            const int rc = rope || C2 ? 1 : valley_p16_gemm(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, epilogue, out_dtype, st);
            return rc == 1 ? run_tile(197, tile_hint, VLY_TILE_ARGS_RAW) : rc;
        }
        case 397:                                           // persistent on the 32x32x16 MFMA (gemm_p32.hip); what it does not take goes to 197
        case 398: {                                         // (398: its DEEP form — four resident fragment sets, burst stores)
            const int rc = rope || C2 ? 1 : valley_p32_gemm(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, epilogue, out_dtype, st, t == 398);
            return rc == 1 ? run_tile(197, tile_hint, VLY_TILE_ARGS_RAW) : rc;
        }
        // 4 waves x (128 x 128): a quarter of the 16-wave tile's LDS fragment traffic (PIPE 8 comment)
        case 197:                                           // persistent: one workgroup per CU walks the tiles (gemm_p4_kernel)
        case 198:
        case 199:
            if (C2) { vly_set_error("vly_gemm_bf16: tile_hint %d does not take the split-K pair", tile_hint); return -22; }
            if (K < 2 * BK) return run_tile(t - 100, tile_hint, VLY_TILE_ARGS_RAW);
            {
                const int rc = t == 197 ? launch_p4<256, 256>(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, st, rope)
                               : t == 198 ? launch_p4<224, 256>(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, st, rope)
                                          : launch_p4<192, 256>(A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, st, rope);
                // 1: bf16 rows that are not 16-byte aligned, or bf16 + residual -> the LDS / fragment epilogues of tile 97 / 98
                return rc == 1 ? run_tile(t - 100, tile_hint, VLY_TILE_ARGS_RAW) : rc;
            }
        case 97:
        case 98:                                            // 224 x 256 (M = 2688 = 12 x 224), 112 x 128 per wave
        case 99:                                            // 192 x 256 (M = 2688 = 14 x 192; 2688 x 27648: 1512 tiles = 5.9 rounds of 0.75)
            if (t == 97) return launch_tile<256, 256, 128, 128, 8>(VLY_TILE_ARGS);
            if (t == 98) return launch_tile<224, 256, 112, 128, 8>(VLY_TILE_ARGS);
            return launch_tile<192, 256, 96, 128, 8>(VLY_TILE_ARGS);
        case 51: return launch_tile<256, 256, 128, 64, 4>(VLY_TILE_ARGS);
        case 53: return launch_tile<256, 128, 64, 64, 4>(VLY_TILE_ARGS);
        case 54: return launch_tile<128, 256, 64, 64, 4>(VLY_TILE_ARGS);
        case 55: return launch_tile<192, 256, 96, 64, 4>(VLY_TILE_ARGS);
        default: vly_set_error("vly_gemm_bf16: bad tile_hint %d", tile_hint); return -22;
    }
#endif
#undef VLY_TILE_ARGS
#undef VLY_TILE_ARGS_RAW
}

extern "C" int vly_gemm_bf16(const void* A, const void* W, const float* bias, const float* residual, void* C,
                             int M, int N, int K, int lda, int ldw, int ldc, int ldr, int epilogue,
                             int out_dtype, int tile_hint, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0) { vly_set_error("vly_gemm_bf16: empty problem"); return -22; }
    const bool wpacked = ldw == VLY_LDW_PACKED64;
    if (K % BK || lda % 8 || (ldw % 8 || (ldw <= 0 && !wpacked)) || N % 4 || ldc % 2 || (epilogue == VLY_EPI_SWIGLU && N % 8) ||
        ((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)C & 7) ||
        (residual && (ldr % 4 || ((uintptr_t)residual & 15))) || (bias && ((uintptr_t)bias & 15))) {
        vly_set_error("vly_gemm_bf16: unsupported shape/alignment M=%d N=%d K=%d lda=%d ldw=%d ldc=%d ldr=%d",
                      M, N, K, lda, ldw, ldc, ldr);
        return -22;
    }
    if ((size_t)M * lda >= (1ull << 31) || (wpacked ? (size_t)((N + 63) / 64 * 64) * K : (size_t)N * ldw) >= (1ull << 31)) {
        vly_set_error("vly_gemm_bf16: operand exceeds 2^31 elements (32-bit byte offsets)");
        return -22;
    }
    if (epilogue == VLY_EPI_SWIGLU && residual) { vly_set_error("vly_gemm_bf16: SWIGLU takes no residual"); return -22; }
    hipStream_t st = (hipStream_t)stream;
    const int t = tile_hint ? tile_hint : pick_tile(M, N);
    return run_tile(t, tile_hint, A, W, bias, residual, C, M, N, K, lda, ldw, ldc, ldr, epilogue, out_dtype, st, nullptr);
}

extern "C" int vly_gemm_bf16_splitk2(const void* A, const void* W, const float* bias, void* C0, void* C1, int M, int N, int K,
                                     int lda, int ldw, int ldc, int tile_hint, void* stream) {
    const bool wpacked = ldw == VLY_LDW_PACKED64;
    if (M <= 0 || N <= 0 || K < 2 * BK || K % BK || lda % 8 || ldw % 8 || (ldw <= 0 && !wpacked) || N % 4 || ldc % 2 || !C0 ||
        !C1 || C0 == C1 ||
        ((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)C0 & 7) || ((uintptr_t)C1 & 7) || (bias && ((uintptr_t)bias & 15))) {
        vly_set_error("vly_gemm_bf16_splitk2: unsupported shape/alignment M=%d N=%d K=%d lda=%d ldw=%d ldc=%d", M, N, K, lda, ldw, ldc);
        return -22;
    }
    if ((size_t)M * lda >= (1ull << 31) || (wpacked ? (size_t)((N + 63) / 64 * 64) * K : (size_t)N * ldw) >= (1ull << 31)) {
        vly_set_error("vly_gemm_bf16_splitk2: operand exceeds 2^31 elements (32-bit byte offsets)");
        return -22;
    }
    const int t = tile_hint ? tile_hint : 8;
    return run_tile(t, tile_hint, A, W, bias, nullptr, C0, M, N, K, lda, ldw, ldc, 0, VLY_EPI_NONE, VLY_OUT_BF16,
                    (hipStream_t)stream, C1);
}

extern "C" int vly_gemm_bf16_qkv_rope(const void* A, const void* W, void* qkv, void* kcache, void* vcache, const float* cos_table,
                                      const float* sin_table, int M, int H, int K, int lda, int ldw, int ldc, int S, int heads,
                                      int past_len, int ctx_max, int tile_hint, void* stream) {
    const bool wpacked = ldw == VLY_LDW_PACKED64;
    const int N = 3 * H;
    if (M <= 0 || H <= 0 || K <= 0 || K % BK || H != heads * 128 || S <= 0 || M % S || past_len < 0 || past_len + S > ctx_max ||
        lda % 8 || (ldw % 8 || (ldw <= 0 && !wpacked)) || ldc % 8 || ldc < N || !qkv || !kcache || !vcache || !cos_table || !sin_table ||
        ((uintptr_t)A & 15) || ((uintptr_t)W & 15) || ((uintptr_t)qkv & 15) || ((uintptr_t)kcache & 15) || ((uintptr_t)vcache & 15) ||
        ((uintptr_t)cos_table & 15) || ((uintptr_t)sin_table & 15)) {
        vly_set_error("vly_gemm_bf16_qkv_rope: unsupported shape/alignment M=%d H=%d K=%d S=%d heads=%d past=%d ctx_max=%d lda=%d ldw=%d ldc=%d",
                      M, H, K, S, heads, past_len, ctx_max, lda, ldw, ldc);
        return -22;
    }
    if ((size_t)M * lda >= (1ull << 31) || (wpacked ? (size_t)((N + 63) / 64 * 64) * K : (size_t)N * ldw) >= (1ull << 31)) {
        vly_set_error("vly_gemm_bf16_qkv_rope: operand exceeds 2^31 elements (32-bit byte offsets)");
        return -22;
    }
    const RopeArgs rp{cos_table, sin_table, (uint16_t*)kcache, (uint16_t*)vcache, S, past_len, heads, ctx_max};
    const int t = tile_hint ? tile_hint : pick_tile(M, N);
    return run_tile(t, tile_hint, A, W, nullptr, nullptr, qkv, M, N, K, lda, ldw, ldc, 0, VLY_EPI_QKV_ROPE, VLY_OUT_BF16,
                    (hipStream_t)stream, nullptr, &rp);
}
