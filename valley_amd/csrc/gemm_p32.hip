// Persistent GEMM on the 32x32x16 matrix-core instruction (round 6):  C[M,N] = epi(A[M,K] . W[N,K]^T + bias), 16-bit in / out.
//
// Why a second persistent kernel: gemm_p4_kernel (gemm_bf16.hip) is built from v_mfma_f32_16x16x32 and runs at 93 % of what that
// instruction sustains on random operands at the power-limited clock (1.40 of 1.5 PFLOP/s; the 32x32x16 form sustains 1.9:
// profiles/history/r01/r01_mfma_shapes_probe.txt) — half the operand-register reads per flop, half the MFMA issues.  The
// instruction shape was the cap.  What changes with it, and what this kernel is built around:
//   * a wave's 128 x 128 block of C is 16 blocks of 32 x 32 = a[16 b : 16 b + 15], held BY NAME (asm MFMAs spelling the
//     register range; no C++ value is ever an accumulator) — as in gemm_p4_kernel's rolled instantiations;
//   * a K tile (BK = 64) is 4 K steps of 16 MFMAs; the fragments of a K step are 4 + 4 ds_read_b128 = 32 registers, two sets
//     (64 registers, 128 in gemm_p4_kernel).  With 32-cycle MFMAs a gap hides ~5 other instructions: every fragment read, LDS-DMA
//     piece, address computation and store below sits at a fixed MFMA index (hooks), at most a handful per gap;
//   * ONE barrier per K tile: in the middle of K step 2 every wave has read the last fragments of the K tile's buffer and waited
//     (counted vmcnt) for its own LDS-DMA pieces of the next K tile — the barrier publishes that tile and frees this buffer, the 16
//     pieces of K tile kt + 2 follow one per two MFMAs;
//   * the LDS rows are 128 B (8 chunks of 16 B); chunk ^= (row >> 1) & 7 on the DMA SOURCE address and on the ds_read — the
//     conflict-free form for the 32-row fragments (a ds_read_b128 lane group covers rows {0-3, 12-15, 20-27} or {4-11, 16-19,
//     28-31} of one chunk column: the 16 (row & 1, (row >> 1) & 7) pairs of either group are distinct);
//   * bias goes in through the matrix cores: the first MFMA of a block is D = biasfrag . ones + 0, the bias split into three
//     16-bit pieces at k = 0, 1, 2 (hi + lo + lo2 is the fp32 value exactly; products with 1.0 are exact) — no VALU add, no
//     bias registers in the epilogue; it is loaded by one more LDS-DMA piece;
//   * the finished tile leaves through LDS: each wave turns a 32 x 128 block row around in its private 8 KB (ds_write_b64 of the
//     packed lane-owned quads, ds_read_b128 of whole rows), so every store instruction writes 4 rows x 256 B (plain) or 8 rows x
//     128 B (SwiGLU) — whole 128-byte lines — and the 16-byte pieces are PARKED in registers (32 x 4 = 128 VGPRs, the registers
//     the smaller fragment sets left free) and stored four per K tile under the first eight K tiles of the NEXT tile: the store
//     path that bounded gemm_p4_kernel's tile boundary (16 B/clk per CU for partial lines, ~1 TB/s per XCD, a
//     wave stalled in the in-order store queue issues nothing) sees a trickle instead of a 32 MB burst.
// Operands as gemm_bf16.hip: both K-contiguous, W plain [N][ldw] or block-packed (VLY_LDW_PACKED64); tile order = XCD-contiguous
// runs, groups of gm m-tiles (vly_tile_group_height).  Summation order differs from the 16x16x32 kernels (K step of 16, bias
// first): results agree to fp32 rounding, not bit for bit.
// Algorithmic work: 2 M N K flop per launch; HBM floor (M K + N K) 2 + M N' 2 bytes.
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>
#include "gemm_persist.hpp"
#include "../../include/valley_hip.h"

namespace {
using namespace vlyp;
typedef TileMap P32Map;

constexpr int BK = 64;

#if VLY_FP16
#define VLY_MFMA32_NAME "v_mfma_f32_32x32x16_f16"
#else
#define VLY_MFMA32_NAME "v_mfma_f32_32x32x16_bf16"
#endif
// A/B switches (tools/ab_lib.py builds variants with -D...)
#ifndef VLY_P32_PIECE_START
#define VLY_P32_PIECE_START 13      // first LDS-DMA piece of a group: this many MFMAs into K step 2 (the barrier sits at 12)
#endif
#ifndef VLY_P32_PIECE_STRIDE
#define VLY_P32_PIECE_STRIDE 2      // MFMAs between two pieces
#endif
#ifndef VLY_P32_BAR_AT
#define VLY_P32_BAR_AT 12
#endif
#ifndef VLY_P32_HEAD
#define VLY_P32_HEAD 8              // the parked stores of a tile leave under this many K tiles of the next one
#endif
#ifndef VLY_P32_TIMING
#define VLY_P32_TIMING 0
#endif
#ifndef VLY_P32_ABLATE
#define VLY_P32_ABLATE 0            // diagnostic builds (wrong results): 1 no LDS-DMA pieces, 2 no fragment reads, 4 no barrier, 8 no MFMAs
#endif

// a[16 blk .. 16 blk + 15] (+)= Wfrag . Afrag : D[i = n][j = m], lane l supplies W[n = l & 31][k = 8 (l >> 5) ..] and A[m = l & 31][same k],
// and holds D[n = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][m = l & 31] in register r
VLY_DEVICE void mfma32(int blk, const bf16x8& w, const bf16x8& a) {
    asm volatile(VLY_MFMA32_NAME " a[%2:%3], %0, %1, a[%2:%3]" ::"v"(w), "v"(a), "i"(16 * blk), "i"(16 * blk + 15));
}
VLY_DEVICE void mfma32_zero(int blk, const bf16x8& w, const bf16x8& a) {
    asm volatile(VLY_MFMA32_NAME " a[%2:%3], %0, %1, 0" ::"v"(w), "v"(a), "i"(16 * blk), "i"(16 * blk + 15));
}
// (s_nop 1 opens each statement: hipcc reuses the registers of a just-issued ds_write / store as these outputs, and nothing pads an asm
//  statement — without it the last lanes of the store's data were overwritten before the store had read them: round 6, NaNs in lanes 60-63)
VLY_DEVICE void acc_read16(int blk, float (&v)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q)
        asm volatile("s_nop 1\n\tv_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\tv_accvgpr_read_b32 %3, a[%7]"
                     : "=v"(v[4 * q]), "=v"(v[4 * q + 1]), "=v"(v[4 * q + 2]), "=v"(v[4 * q + 3])
                     : "i"(16 * blk + 4 * q), "i"(16 * blk + 4 * q + 1), "i"(16 * blk + 4 * q + 2), "i"(16 * blk + 4 * q + 3));
}
// 16 MFMAs of one K step; hook(t) runs behind MFMA t.  MODE 1: C = 0 (first K step of a tile without bias)
template <int MODE, typename H>
VLY_DEVICE void step32(const bf16x8 (&fa)[4], const bf16x8 (&fw)[4], H&& hook) {
#pragma unroll
    for (int bi = 0; bi < 4; ++bi) {
#pragma unroll
        for (int bj = 0; bj < 4; ++bj) {
            __builtin_amdgcn_sched_barrier(0);
            if constexpr ((VLY_P32_ABLATE & 8) != 0) asm volatile("" ::"v"(fw[bj]), "v"(fa[bi]));
            else if constexpr (MODE == 1) mfma32_zero(bi * 4 + bj, fw[bj], fa[bi]);
            else mfma32(bi * 4 + bj, fw[bj], fa[bi]);
            __builtin_amdgcn_sched_barrier(0);
            hook(bi * 4 + bj);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// DEEP = false: the finished tile is parked in registers and trickles out under the next tile (64 fragment registers, one barrier per
// K tile, the buffer's read window = the whole K tile).  DEEP = true: all four K steps' fragment sets are resident (128 registers, none
// left to park in: the tile's stores leave in one burst) and the next K tile is read in a compressed burst — K steps 0, 1 during this
// tile's step 2, step 2 during step 3, step 3 during the next tile's step 0 — so a buffer is free ~0.2 K tiles after its K tile began
// and its refill has ~1.3 K tiles to land instead of 1.0 (two barriers per K tile: F = buffer free, L = next tile landed).
template <int EPI, bool DEEP>
__global__ void __launch_bounds__(256)
gemm_p32_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W, const float* __restrict__ bias, void* __restrict__ Cv,
                int M, int N, int K, int lda, int ldw, int ldc, P32Map mp, unsigned long long* __restrict__ tstamps) {
    constexpr int BM = 256, BN = 256, NT = 256;
    constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;           // one K tile: 64 KB
    constexpr int SCR = 8192;                                             // a wave's private turn-around area (and its bias landing strip)
    constexpr int PA = BM * 8 / NT, PW = BN * 8 / NT, NS = PA + PW;       // 16 LDS-DMA pieces of 1 KB per wave and K tile
    constexpr bool SWI = EPI == VLY_EPI_SWIGLU;
    constexpr int NPIECE = SWI ? 16 : 32;                                 // 16-byte pieces per lane and tile
    constexpr int NPARK = DEEP ? 1 : NPIECE;                              // ... parked in registers
    constexpr int HEAD = VLY_P32_HEAD, SPK = NPIECE / HEAD;               // stores per K tile while the parked tile drains
    static_assert(NPIECE % HEAD == 0, "head");
    constexpr int PG0 = VLY_P32_PIECE_START, PGS = VLY_P32_PIECE_STRIDE, BAR_AT = VLY_P32_BAR_AT;
    // g: MFMA index counted from K step 2 (g = 0 .. 63: steps 2, 3, then 0, 1 of the next K tile).  Piece q at g = PG0 + q PGS.
    constexpr int PG_LAST = PG0 + (NS - 1) * PGS;
    static_assert(PG0 > BAR_AT && PG_LAST + 1 + SPK * 2 < 64, "piece schedule");
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE + 4 * SCR];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int ntiles = mp.tiles_m * mp.tiles_n, G = (int)gridDim.x;
    const int nk = K / BK;
    const uint32_t wk = ldw < 0 ? (uint32_t)((N + 63) >> 6) * 4096u : (uint32_t)BK;
    const int No = SWI ? N >> 1 : N;

    auto tile_origin = [&](int t, int& m0, int& n0) { vlyp::tile_origin<BM, BN>(mp, ntiles, t, m0, n0); };
    // Operands through descriptors that END with the last row: a piece whose row lies past M (N) is out of range and reads as zero —
    // no per-row clamp, so a lane's source offset is LINEAR in the piece number: one register per operand (gemm_p4_kernel keeps 14-16).
    // (Rows past N of a block-packed W land in later blocks or past the end: finite garbage or zero in columns that are never stored.)
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(A), 0, (uint32_t)M * (uint32_t)lda * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(W), 0, ldw < 0 ? (uint32_t)((N + 63) >> 6) * 4096u * (uint32_t)nk * 2u : (uint32_t)N * (uint32_t)ldw * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, bias ? (uint32_t)N * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(Cv, 0, (uint32_t)M * (uint32_t)ldc * 2u, 0x00020000);

    // ---- load cursor: (lt, lk) = tile / K tile of the group of pieces being issued; per-lane source offset of its tile
    const int prow = tid >> 3, pswz = ((tid & 7) ^ ((tid >> 4) & 7)) << 3;      // row of this lane's slot in piece 0; swizzled chunk (elements)
    const uint32_t sA32 = (uint32_t)lda * 64u, sW32 = ldw < 0 ? 4096u : (uint32_t)ldw * 64u;       // bytes between the rows of two pieces
    const uint32_t kW = wk * 2u;                                          // bytes between two K tiles of W
    auto offA = [&](int m0) { return (__umul24((uint32_t)(m0 + prow), (uint32_t)lda) + (uint32_t)pswz) * 2u; };
    auto offW = [&](int n0) { return (w_row_off32(n0 + prow, ldw) + (uint32_t)pswz) * 2u; };
    int lt = (int)blockIdx.x, lk = 0;
    int lm0, ln0;                                                        // origin of the load cursor's tile
    tile_origin(lt, lm0, ln0);
    uint32_t vA = offA(lm0), vW = offW(ln0);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t ldsw = lds0 + (uint32_t)wave * 1024u;
    auto piece = [&](int buf, int q) {                                   // M0 = LDS destination, then one buffer_load ... lds
        if constexpr ((VLY_P32_ABLATE & 1) != 0) return;
        if ((VLY_P32_ABLATE & 16) != 0 && q >= PA) return;               // (A pieces only)
        if ((VLY_P32_ABLATE & 32) != 0 && q < PA) return;                // (W pieces only)
        const uint32_t dst = ldsw + (uint32_t)buf * STAGE + (q < PA ? (uint32_t)q * 4096u : (uint32_t)A_BYTES + (uint32_t)(q - PA) * 4096u);
        const uint32_t off = q < PA ? vA + ((uint32_t)q * sA32 + (uint32_t)lk * (BK * 2u)) : vW + ((uint32_t)(q - PA) * sW32 + (uint32_t)lk * kW);
        if (q < PA) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(dst), "v"(off), "s"(rsA) : "memory");
        else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(dst), "v"(off), "s"(rsW) : "memory");
    };
    // the bias of the load cursor's tile: 256 floats into this wave's scratch (every wave its own copy: no cross-wave hand-off)
    const uint32_t scr0 = lds0 + 2u * STAGE + (uint32_t)wave * SCR;
    auto bias_piece = [&]() {
        const uint32_t vb = (uint32_t)(ln0 + 4 * lane) * 4u;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(scr0), "v"(vb), "s"(rsB) : "memory");
    };

    // ---- compute cursor
    int ct = lt, cm0 = lm0, cn0 = ln0;
    int nm0 = 0, nn0 = 0;                                                // origin of the tile after the compute cursor's (set by the load cursor)
    const int wm0 = (wave >> 1) * 128, wn0 = (wave & 1) * 128;
    const int rdA = (wave >> 1) * 16384, rdW = A_BYTES + (wave & 1) * 16384;
    int fo[4];                                                           // fragment read offset of K step s inside a 32-row block
#pragma unroll
    for (int s = 0; s < 4; ++s) fo[s] = l31 * 128 + (((2 * s + h) ^ ((lane >> 1) & 7)) << 4);
    bf16x8 fa0[4], fw0[4], fa1[4], fw1[4];                               // fragment sets of even / odd K steps (DEEP: of K steps 0, 1)
    [[maybe_unused]] bf16x8 fa2[4], fw2[4], fa3[4], fw3[4];              // DEEP: K steps 2, 3
    auto rd = [&](bf16x8 (&fa)[4], bf16x8 (&fw)[4], const char* st, int o) {
        return [&, st, o](int k) {
            if constexpr ((VLY_P32_ABLATE & 2) != 0) return;
            if (k < 4) fw[k < 4 ? k : 0] = *(const bf16x8*)(st + rdW + k * 4096 + o);
            else if (k < 8) fa[k >= 4 && k < 8 ? k - 4 : 0] = *(const bf16x8*)(st + rdA + (k - 4) * 4096 + o);
        };
    };

    // ---- the parked tile: 16-byte pieces, their common lane offset, stores by asm (nothing of this kernel's vector memory traffic
    // is visible to hipcc's own vmcnt bookkeeping: every wait is ours)
    u32x4 park[NPARK];
#pragma unroll
    for (int p = 0; p < NPARK; ++p) park[p] = u32x4{0u, 0u, 0u, 0u};
    uint32_t vo_park = 0x80000000u;                                      // (nothing parked yet: out of range, dropped by the hardware)
    const uint32_t pstep = (uint32_t)ldc * (SWI ? 16u : 8u);             // bytes between the rows of two consecutive parked pieces
    auto park_store = [&](int p) {
        const uint32_t off = vo_park + (uint32_t)p * pstep;
        asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(park[p]), "v"(off), "s"(rsC) : "memory");     // (§5.7: the data registers are read late)
    };
    float braw[4] = {0.f, 0.f, 0.f, 0.f};                                // bias[n0 + wn0 + 32 bj + l31] of the NEXT tile to start
    auto bias_read = [&]() {
        const char* scr = smem + 2 * STAGE + wave * SCR;
#pragma unroll
        for (int bj = 0; bj < 4; ++bj) braw[bj] = *(const float*)(scr + (wn0 + bj * 32 + l31) * 4);
    };

#if VLY_P32_TIMING
    // anatomy builds (tools/p32_times.py): lane 0 of the first 64 workgroups stamps s_memtime at the seams of every tile (the LDS is full:
    // the stamps go to memory by asm stores, which join the in-order vmcnt queue in front of the next K tile's pieces)
    int tsn = 0;
#define VLY_STAMP()                                                                                                   \
    do {                                                                                                              \
        if (threadIdx.x == 0 && blockIdx.x < 64 && tsn < 64 && tstamps) {                                             \
            const unsigned long long now = __builtin_readcyclecounter();                                             \
            unsigned long long* dst = tstamps + (size_t)blockIdx.x * 65 + 1 + tsn;                                    \
            asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst), "v"(now) : "memory");                         \
        }                                                                                                             \
        ++tsn;                                                                                                        \
    } while (0)
#else
#define VLY_STAMP() do {} while (0)
#endif

    asm volatile("" ::: VLY_ALL_AGPRS);                                  // the kernel owns a0 .. a255
    int buf = 0;
    bool wave_live = false;
    bool crossing = false;                                               // the load cursor enters a new tile at its next advance

    // ---- prologue: group 0 whole, pieces 0 .. NPRE-1 of group 1 (the rest follows in K tile 0's steps 0 / 1)
    constexpr int NPRE = (64 - PG0 + PGS - 1) / PGS < NS ? (64 - PG0 + PGS - 1) / PGS : NS;     // pieces issued in steps 2, 3 (g < 64 - 32)
    constexpr int NFIRST = (32 - PG0 + PGS - 1) / PGS < NS ? (32 - PG0 + PGS - 1) / PGS : NS;   // pieces with g < 32: issued in steps 2 / 3
    if (bias) bias_piece();
#pragma unroll
    for (int q = 0; q < NS; ++q) piece(0, q);
    // advance to group 1
    auto advance_begin = [&]() {                                         // scalar part of the cursor's advance; sets `crossing`
        crossing = false;
        if (lk + 1 < nk) { ++lk; return; }
        if (lt + G >= ntiles) return;                                    // past the last tile: stay on its last K tile
        lt += G;
        lk = 0;
        tile_origin(lt, lm0, ln0);
        nm0 = lm0;
        nn0 = ln0;
        crossing = true;
    };
    advance_begin();
    if (crossing) { vA = offA(lm0); vW = offW(ln0); }
    // DEEP: pieces of a group at MFMA t of step 0 (from DF0, after barrier F), every t of ... see ktile_deep; ND01 of them before barrier L
    constexpr int DBARF = 11, DP0 = 12, ND0 = (16 - DP0 + 1) / 2, ND01 = ND0 + 8;
    if constexpr (DEEP) {
#pragma unroll
        for (int q = 0; q < NS; ++q) piece(1, q);
        advance_begin();
        if (crossing) { vA = offA(lm0); vW = offW(ln0); }
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NS) : "memory");
        __builtin_amdgcn_s_barrier();
        if (bias) bias_read();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            rd(fa0, fw0, smem, fo[0])(k);
            rd(fa1, fw1, smem, fo[1])(k);
            rd(fa2, fw2, smem, fo[2])(k);
        }
    } else {
#pragma unroll
        for (int q = 0; q < NFIRST; ++q) piece(1, q);
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NFIRST) : "memory");
        __builtin_amdgcn_s_barrier();
        if (bias) bias_read();
        auto r0 = rd(fa0, fw0, smem, fo[0]);
#pragma unroll
        for (int k = 0; k < 8; ++k) r0(k);
    }

    // ---- one K tile.  FIRST: the tile's first (C = 0, or the bias MFMAs in front).  ST0 >= 0: park[ST0 .. ST0 + SPK - 1] leave.
    [[maybe_unused]] auto ktile = [&](auto first_c, auto st0_c, bool last) {
        constexpr bool FIRST = decltype(first_c)::value;
        constexpr int ST0 = decltype(st0_c)::value;
        const char* cur = smem + buf * STAGE;
        const char* nxt = smem + (buf ^ 1) * STAGE;
        if (wave_live) {
            // ---- K step 0 (set 0): reads of step 1; pieces with g >= 32 of the group that started in the previous K tile
            bf16x8 bfrag[4];
            if constexpr (FIRST) {
                if (bias) {
                    // D = biasfrag . ones: bias[n] = hi + lo + lo2 at k = 0, 1, 2 of the W-side operand (lanes 0-31), ones at the same k on the A side
                    bf16x8 ones;
                    {
                        const uint32_t one = (uint32_t)f2h(1.0f);
                        const u32x4 o = h == 0 ? u32x4{one | (one << 16), one, 0u, 0u} : u32x4{0u, 0u, 0u, 0u};
                        ones = __builtin_bit_cast(bf16x8, o);
                    }
#pragma unroll
                    for (int bj = 0; bj < 4; ++bj) {
                        const float x = braw[bj];
                        const uint32_t hi = f2h(x);
                        const float r1 = x - h2f((uint16_t)hi);
                        const uint32_t lo = f2h(r1);
                        const float r2 = r1 - h2f((uint16_t)lo);
                        const uint32_t lo2 = f2h(r2);
                        const u32x4 b = h == 0 ? u32x4{hi | (lo << 16), lo2, 0u, 0u} : u32x4{0u, 0u, 0u, 0u};
                        bfrag[bj] = __builtin_bit_cast(bf16x8, b);
                    }
                    asm volatile("s_nop 4" : "+v"(ones), "+v"(bfrag[0]), "+v"(bfrag[1]), "+v"(bfrag[2]), "+v"(bfrag[3]));     // VALU write -> MFMA operand
#pragma unroll
                    for (int b = 0; b < 16; ++b) {
                        __builtin_amdgcn_sched_barrier(0);
                        mfma32_zero(b, bfrag[b & 3], ones);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            auto h0 = [&](int t) {
                rd(fa1, fw1, cur, fo[1])(t);
#pragma unroll
                for (int q = 0; q < NS; ++q)
                    if (PG0 + q * PGS == 32 + t) piece(buf ^ 1, q);
                if (32 + t == PG_LAST + 1) {                              // the group is out: its tile's bias rides behind it
                    if (bias && lk == 0 && !(lt == ct)) bias_piece();
                }
                if constexpr (ST0 >= 0) {
#pragma unroll
                    for (int s = 0; s < SPK; ++s)
                        if (32 + t == PG_LAST + 2 + 2 * s) park_store(ST0 + s);
                }
            };
            if constexpr (FIRST) {
                if (bias) step32<0>(fa0, fw0, h0);
                else step32<1>(fa0, fw0, h0);
            } else step32<0>(fa0, fw0, h0);
            // ---- K step 1 (set 1): reads of step 2; the load cursor advances (new offsets one per gap when it enters a new tile)
            auto h1 = [&](int t) {
                rd(fa0, fw0, cur, fo[2])(t);
#pragma unroll
                for (int q = 0; q < NS; ++q)
                    if (PG0 + q * PGS == 48 + t) piece(buf ^ 1, q);
                if (48 + t == PG_LAST + 1) {
                    if (bias && lk == 0 && !(lt == ct)) bias_piece();
                }
                if constexpr (ST0 >= 0) {
#pragma unroll
                    for (int s = 0; s < SPK; ++s)
                        if (48 + t == PG_LAST + 2 + 2 * s) park_store(ST0 + s);
                }
            };
            step32<0>(fa1, fw1, h1);
        } else {
            // a wave whose slab of the tile lies outside the problem: pieces, stores and the barrier only
#pragma unroll
            for (int q = NFIRST; q < NS; ++q) piece(buf ^ 1, q);
            if (bias && lk == 0 && !(lt == ct)) bias_piece();
            if constexpr (ST0 >= 0) {
#pragma unroll
                for (int s = 0; s < SPK; ++s) park_store(ST0 + s);
            }
        }
        // ---- the cursor moves on to the group of K tile kt + 2
        advance_begin();
        if (crossing) { vA = offA(lm0); vW = offW(ln0); }
        if (wave_live) {
            // ---- K step 2 (set 0): reads of step 3; barrier; first pieces of the new group into THIS buffer
            auto h2 = [&](int t) {
                rd(fa1, fw1, cur, fo[3])(t);
                if (t == BAR_AT) {
                    if constexpr (ST0 >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(SPK) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    if constexpr ((VLY_P32_ABLATE & 4) == 0) __builtin_amdgcn_s_barrier();
                }
#pragma unroll
                for (int q = 0; q < NS; ++q)
                    if (PG0 + q * PGS == t) piece(buf, q);
            };
            step32<0>(fa0, fw0, h2);
            // ---- K step 3 (set 1): reads of the NEXT K tile's step 0 (the buffer the barrier just published)
            auto h3 = [&](int t) {
                rd(fa0, fw0, nxt, fo[0])(t);
#pragma unroll
                for (int q = 0; q < NS; ++q)
                    if (PG0 + q * PGS == 16 + t) piece(buf, q);
                if (t == 9 && last && bias) bias_read();                  // the next tile's bias landed with its first K tile
            };
            step32<0>(fa1, fw1, h3);
        } else {
            if constexpr (ST0 >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(SPK) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int q = 0; q < NFIRST; ++q) piece(buf, q);
            if (last) {                                                   // this wave may be live in the next tile: its first fragments, its bias
                if (bias) bias_read();
                auto r0 = rd(fa0, fw0, nxt, fo[0]);
#pragma unroll
                for (int k = 0; k < 8; ++k) r0(k);
            }
        }
        buf ^= 1;
    };
    // ---- DEEP: one K tile with all four fragment sets resident.  after_burst: this wave's drain stores (NPIECE of them) sit in the queue
    // between the last pieces of the group barrier L waits for and the pieces issued since — count them in instead of waiting for them
    bool after_burst = false;
    auto bias_mfmas = [&]() {
        bf16x8 bfrag[4], ones;
        int h = lane >> 5;
        asm volatile("" : "+v"(h));                                      // (not hoisted out of the persistent loop)
        {
            const uint32_t one = (uint32_t)f2h(1.0f);
            const u32x4 o = h == 0 ? u32x4{one | (one << 16), one, 0u, 0u} : u32x4{0u, 0u, 0u, 0u};
            ones = __builtin_bit_cast(bf16x8, o);
        }
#pragma unroll
        for (int bj = 0; bj < 4; ++bj) {
            const float x = braw[bj];
            const uint32_t hi = f2h(x);
            const float r1 = x - h2f((uint16_t)hi);
            const uint32_t lo = f2h(r1);
            const float r2 = r1 - h2f((uint16_t)lo);
            const uint32_t lo2 = f2h(r2);
            const u32x4 b = h == 0 ? u32x4{hi | (lo << 16), lo2, 0u, 0u} : u32x4{0u, 0u, 0u, 0u};
            bfrag[bj] = __builtin_bit_cast(bf16x8, b);
        }
        asm volatile("s_nop 4" : "+v"(ones), "+v"(bfrag[0]), "+v"(bfrag[1]), "+v"(bfrag[2]), "+v"(bfrag[3]));     // VALU write -> MFMA operand
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            __builtin_amdgcn_sched_barrier(0);
            mfma32_zero(b, bfrag[b & 3], ones);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    [[maybe_unused]] auto ktile_deep = [&](auto first_c, bool last) {
        constexpr bool FIRST = decltype(first_c)::value;
        const char* cur = smem + buf * STAGE;
        const char* nxt = smem + (buf ^ 1) * STAGE;
        auto wait_landed = [&]() {
            if (after_burst) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(ND01 + NPIECE) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(ND01) : "memory");
            after_burst = false;
        };
        if (wave_live) {
            if constexpr (FIRST) {
                if (bias) bias_mfmas();
            }
            // ---- step 0: this K tile's step-3 fragments (the last reads of its buffer), barrier F, the next group starts
            auto h0 = [&](int t) {
                rd(fa3, fw3, cur, fo[3])(t);
                if (t == DBARF) {
                    __builtin_amdgcn_s_waitcnt(0xc07f);
                    __builtin_amdgcn_s_barrier();
                }
#pragma unroll
                for (int q = 0; q < ND0; ++q)
                    if (t == DP0 + 2 * q) piece(buf, q);
            };
            if constexpr (FIRST) {
                if (bias) step32<0>(fa0, fw0, h0);
                else step32<1>(fa0, fw0, h0);
            } else step32<0>(fa0, fw0, h0);
            // ---- step 1: eight more pieces; barrier L: the next K tile has landed everywhere
            auto h1 = [&](int t) {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (t == 2 * q) piece(buf, ND0 + q);
                if (t == 15) {
                    wait_landed();
                    __builtin_amdgcn_s_barrier();
                }
            };
            step32<0>(fa1, fw1, h1);
            // ---- step 2: the next K tile's steps 0 and 1 into the sets steps 0 and 1 have left; the rest of the pieces
            auto h2 = [&](int t) {
                if (t < 8) rd(fa0, fw0, nxt, fo[0])(t);
                else rd(fa1, fw1, nxt, fo[1])(t - 8);
#pragma unroll
                for (int q = ND01; q < NS; ++q)
                    if (t == 1 + 2 * (q - ND01)) piece(buf, q);
                if (t == 1 + 2 * (NS - ND01)) {
                    if (bias && lk == 0 && !(lt == ct)) bias_piece();     // the group is out: its tile's bias rides behind it
                }
            };
            step32<0>(fa2, fw2, h2);
            // ---- step 3: the next K tile's step 2
            auto h3 = [&](int t) {
                rd(fa2, fw2, nxt, fo[2])(t);
                if (t == 9 && last && bias) bias_read();
            };
            step32<0>(fa3, fw3, h3);
        } else {
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int q = 0; q < ND01; ++q) piece(buf, q);
            wait_landed();
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int q = ND01; q < NS; ++q) piece(buf, q);
            if (bias && lk == 0 && !(lt == ct)) bias_piece();
            if (last) {                                                   // this wave may be live in the next tile
                if (bias) bias_read();
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    rd(fa0, fw0, nxt, fo[0])(k);
                    rd(fa1, fw1, nxt, fo[1])(k);
                    rd(fa2, fw2, nxt, fo[2])(k);
                }
            }
        }
        advance_begin();
        if (crossing) { vA = offA(lm0); vW = offW(ln0); }
        buf ^= 1;
    };
    (void)NPRE;

    // ---- the finished tile: accumulators -> activation -> 16-bit -> this wave's LDS area -> whole rows -> park[]
    char* const scr = smem + 2 * STAGE + wave * SCR;
    const int wb_plain = l31 * 256 + ((l31 & 15) << 4) + 8 * h;          // write base: row l31, chunk position ^ (l31 & 15), half h
    const int wb_swi = l31 * 128 + ((l31 & 7) << 4) + 4 * h;
    const int rb_plain = (lane >> 4) * 256 + (((lane & 15) ^ (lane >> 4)) << 4);
    const int rb_swi = (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4);
    auto drain = [&](int em0, int en0) {
        if (wave_live) {
            // (the lane bases go through an asm statement per tile: left alone hipcc computes the sixteen write and eight read addresses of a
            // lane ONCE, outside the persistent loop, and holds two dozen registers for them for the whole kernel)
            int wb_p = wb_plain, wb_s = wb_swi, rb_p = rb_plain, rb_s = rb_swi;
            asm volatile("" : "+v"(wb_p), "+v"(wb_s), "+v"(rb_p), "+v"(rb_s));
            {
                // lane offset of piece 0: row em0 + wm0 + (lane >> 4 | lane >> 3), 16-byte column chunk lane & 15 | lane & 7
                const int row = em0 + wm0 + (SWI ? lane >> 3 : lane >> 4);
                const int col = SWI ? ((en0 + wn0) >> 1) + (lane & 7) * 8 : en0 + wn0 + (lane & 15) * 8;
                vo_park = col + 8 <= No ? ((uint32_t)row * (uint32_t)ldc + (uint32_t)col) * 2u : 0x80000000u;
            }
            auto emit = [&](int pp, const u32x4& v) {                     // piece pp of the tile: parked, or (DEEP) stored at once
                if constexpr (DEEP) {
                    const uint32_t off = vo_park + (uint32_t)pp * pstep;
                    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(off), "s"(rsC) : "memory");
                } else park[pp < NPARK ? pp : 0] = v;
            };
#pragma unroll
            for (int bi = 0; bi < 4; ++bi) {
#pragma unroll
                for (int bj = 0; bj < 4; ++bj) {
                    __builtin_amdgcn_sched_barrier(0);
                    float v[16];
                    acc_read16(bi * 4 + bj, v);
                    if constexpr (SWI) {
                        // gate = even columns, up = odd: quad q gives outputs (8 q + 4 h) / 2 + {0, 1} of this block's 16
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x2 gt = x_sigmoid2(f32x2{v[4 * q], v[4 * q + 2]}, 1.f) * f32x2{v[4 * q + 1], v[4 * q + 3]};
                            const uint32_t d = pack_h2(gt[0], gt[1]);
                            *(uint32_t*)(scr + (wb_s ^ ((bj * 2 + (q >> 1)) << 4)) + (q & 1) * 8) = d;
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            f32x2 x0 = {v[4 * q], v[4 * q + 1]}, x1 = {v[4 * q + 2], v[4 * q + 3]};
                            if constexpr (EPI == VLY_EPI_QUICK_GELU) {
                                x0 = x_sigmoid2(x0, 1.702f);
                                x1 = x_sigmoid2(x1, 1.702f);
                            } else if constexpr (EPI == VLY_EPI_RELU) {
                                x0 = f32x2{fmaxf(x0[0], 0.f), fmaxf(x0[1], 0.f)};
                                x1 = f32x2{fmaxf(x1[0], 0.f), fmaxf(x1[1], 0.f)};
                            }
                            *(u32x2*)(scr + (wb_p ^ ((bj * 4 + q) << 4))) = u32x2{pack_h2(x0[0], x0[1]), pack_h2(x1[0], x1[1])};
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (SWI) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) emit(bi * 4 + i, *(const u32x4*)(scr + rb_s + i * 1024));
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) emit(bi * 8 + i, *(const u32x4*)(scr + (rb_p ^ ((i & 3) << 6)) + i * 1024));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (DEEP) after_burst = true;
        } else vo_park = 0x80000000u;
    };

    // ---- the schedule: per tile { first K tile; HEAD - 1 more with parked stores; the rest rolled; drain }
    for (;;) {
        wave_live = __builtin_amdgcn_readfirstlane((cm0 + wm0 < M && cn0 + wn0 < N) ? 1 : 0) != 0;
        VLY_STAMP();
        if constexpr (DEEP) {
            ktile_deep(std::true_type{}, false);
            for (int kt = 1; kt < HEAD; ++kt) ktile_deep(std::false_type{}, false);
            VLY_STAMP();
            for (int kt = HEAD; kt < nk - 1; ++kt) ktile_deep(std::false_type{}, false);
            ktile_deep(std::false_type{}, true);
        } else {
            ktile(std::true_type{}, std::integral_constant<int, 0>{}, false);
            static_for<HEAD - 1>([&](auto i) { ktile(std::false_type{}, std::integral_constant<int, (decltype(i)::value + 1) * SPK>{}, false); });
            VLY_STAMP();
            for (int kt = HEAD; kt < nk - 1; ++kt) ktile(std::false_type{}, std::integral_constant<int, -1>{}, false);
            ktile(std::false_type{}, std::integral_constant<int, -1>{}, true);
        }
        VLY_STAMP();
        drain(cm0, cn0);
        VLY_STAMP();
        if (ct + G >= ntiles) break;
        ct += G;
        cm0 = nm0;
        cn0 = nn0;
    }
    // ---- the last tile's stores, and no LDS-DMA may outlive the workgroup's LDS allocation
    if constexpr (!DEEP) {
#pragma unroll
        for (int p = 0; p < NPARK; ++p) park_store(p);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if VLY_P32_TIMING
    if (threadIdx.x == 0 && blockIdx.x < 64 && tstamps) tstamps[(size_t)blockIdx.x * 65] = (unsigned long long)tsn;
#endif
}
#undef VLY_STAMP

}  // namespace

#if VLY_P32_TIMING
static void* vlydbg_p32_buffer() {
    static void* buf = [] { void* p = nullptr; (void)hipMalloc(&p, 64 * 65 * 8); (void)hipMemset(p, 0, 64 * 65 * 8); return p; }();
    return buf;
}
extern "C" int vlydbg_p32_timing_read(unsigned long long* host) { return (int)hipMemcpy(host, vlydbg_p32_buffer(), 64 * 65 * 8, hipMemcpyDeviceToHost); }
#endif

// Tile hint 397 of vly_gemm_bf16.  Returns 1 when the problem does not fit this kernel (the caller falls back to hint 197):
// 16-bit outputs, no residual, 16-byte aligned rows of whole 8-column chunks, K >= (HEAD + 1) K tiles, < 2^16 tiles.
__attribute__((visibility("hidden"))) int valley_p32_gemm(const void* A, const void* W, const float* bias, const float* R, void* C, int M, int N,
                                                          int K, int lda, int ldw, int ldc, int epi, int out, hipStream_t st, int deep) {
    constexpr int BM = 256, BN = 256;
    const int No = epi == VLY_EPI_SWIGLU ? N >> 1 : N;
    if (out != VLY_OUT_BF16 || R || ldc % 8 || ((uintptr_t)C & 15) || No % 8 || (size_t)M * (size_t)ldc * 2 >= ((size_t)1 << 31) ||
        K % BK || K / BK < VLY_P32_HEAD + 1 || (epi != VLY_EPI_NONE && epi != VLY_EPI_QUICK_GELU && epi != VLY_EPI_SWIGLU && epi != VLY_EPI_RELU))
        return 1;
    const int tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
    if ((long)tm * tn >= 65536 || lda >= (1 << 24) || (ldw > 0 && ldw >= (1 << 24)) || M >= (1 << 24) || N >= (1 << 24)) return 1;
    const P32Map mp = make_tile_map(M, N, K, BM, BN);
    const int cus = persistent_grid_cus(), tiles = tm * tn;
    dim3 grid(tiles >= cus ? cus : tiles), block(256);
    unsigned long long* ts = nullptr;
#if VLY_P32_TIMING
    ts = (unsigned long long*)vlydbg_p32_buffer();
#endif
#define VLY_P32_LAUNCH(E, D)                                                                                                       \
    hipLaunchKernelGGL((gemm_p32_kernel<E, D>), grid, block, 0, st, (const uint16_t*)A, (const uint16_t*)W, bias, C, M, N, K, lda, ldw, ldc, \
                       mp, ts)
    if (deep) {
        if (epi == VLY_EPI_NONE) VLY_P32_LAUNCH(VLY_EPI_NONE, true);
        else if (epi == VLY_EPI_QUICK_GELU) VLY_P32_LAUNCH(VLY_EPI_QUICK_GELU, true);
        else if (epi == VLY_EPI_SWIGLU) VLY_P32_LAUNCH(VLY_EPI_SWIGLU, true);
        else VLY_P32_LAUNCH(VLY_EPI_RELU, true);
    } else {
        if (epi == VLY_EPI_NONE) VLY_P32_LAUNCH(VLY_EPI_NONE, false);
        else if (epi == VLY_EPI_QUICK_GELU) VLY_P32_LAUNCH(VLY_EPI_QUICK_GELU, false);
        else if (epi == VLY_EPI_SWIGLU) VLY_P32_LAUNCH(VLY_EPI_SWIGLU, false);
        else VLY_P32_LAUNCH(VLY_EPI_RELU, false);
    }
#undef VLY_P32_LAUNCH
    return vly_check_launch("vly_gemm_bf16");
}
