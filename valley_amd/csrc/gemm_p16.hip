// Persistent GEMM on v_mfma_f32_16x16x32 issued as CHAINS OF TWO (round 6, tile hint 497):  C[M,N] = epi(A[M,K] . W[N,K]^T + bias).
//
// Where it comes from.  Round 5 left gemm_p4_kernel (gemm_bf16.hip) at 1.40 PFLOP/s "93 % of the 16x16x32 ceiling" and VERDICT r5 asked
// for the 32x32x16 instruction.  Measured this round (profiles/r06/):
//   * tools/probes/mfma_order.hip, one wave per SIMD owning all 256 accumulation registers, asm-pinned issue order, random operands,
//     settled clocks: 32x32x16 sustains 1.92-1.96 PFLOP/s whatever the order; 16x16x32 2.15 (no operand repeats) / 2.23 (one operand
//     held for 8 MFMAs = gemm_p4_kernel's order) / 2.37 with the two K steps of a block issued BACK TO BACK on one accumulator.
//     The chip is power-limited: the 32x32x16 form moves twice the accumulator bytes per flop and clocks lower — the shape was not the cap;
//   * gemm_p32.hip (the 32x32x16 kernel, hints 397 / 398) runs the K loop in the same ~2390 cycles per K tile as gemm_p4_kernel at
//     8 % lower clock; its ablations (r06_p32_ablation_*.txt, r06_p32_energy.txt) put the energy of a GEMM at ~60 % matrix cores, ~19 %
//     LDS -> register fragment reads, ~18 % L2 -> LDS staging; removing cycles (its DEEP form, -5 %) gave the time back as clock.
// So this kernel spends its effort where the joules are: the cheapest MFMA order (chains of two), and no exposed store phase.
//   * a wave's 128 x 128 block of C = 64 blocks of 16 x 16 = a[4 b : 4 b + 3] by name; a K tile (BK = 64) is walked ROW by row of
//     blocks: row i = 8 blocks x { K step 0, K step 1 } back to back on the same accumulator.  Resident: the W fragments of the whole
//     K tile (16 x 4 registers) and the A fragments of two rows (4 x 4): 80 registers (gemm_p4_kernel: 128), same 32 ds_read_b128
//     per K tile.  Row i + 1's A fragments are read during row i; the NEXT K tile's W fragments during row 7, each pair into the
//     registers its column's chain has just left;
//   * ONE barrier per K tile (row 6): every wave has read the buffer's last fragments and waited (counted vmcnt) for its own LDS-DMA
//     pieces of the next K tile; the 16 pieces of K tile kt + 2 follow, one per four MFMAs; source offsets are linear in the piece
//     number (descriptors that end with the last row: no clamps) — one register per operand;
//   * the finished tile: accumulators -> + bias (fp32, from the wave's LDS strip, landed by one more LDS-DMA piece) -> activation ->
//     16 bit -> the wave's private LDS area, one 16-row block row at a time (ds_write_b64 of the lane's 4 columns; ds_read_b128 of
//     whole rows) -> 32 (SwiGLU: 16) 16-byte pieces PARKED in registers, stored four (two) per K tile under the first eight K tiles
//     of the next tile as whole 128-byte lines (4 rows x 256 B | 8 rows x 128 B per instruction).  The store path that bounded
//     gemm_p4_kernel's tile boundary sees a trickle; the drain itself is VALU + LDS only.
// Operands as gemm_bf16.hip (both K-contiguous; W plain or VLY_LDW_PACKED64), tile order as gemm_p4_kernel (XCD-contiguous runs, groups
// of gm m-tiles).  Per block the products are added in the same K order as in gemm_p4_kernel; bias is added in fp32 before the activation.
// Algorithmic work: 2 M N K flop per launch; HBM floor (M K + N K) 2 + M N' 2 bytes.
#include <cstdlib>
#include <cstring>
#include "gemm_persist.hpp"
#include "../../include/valley_hip.h"

namespace {
using namespace vlyp;

constexpr int BK = 64;

#if VLY_FP16
#define VLY_MFMA16C_NAME "v_mfma_f32_16x16x32_f16"
#else
#define VLY_MFMA16C_NAME "v_mfma_f32_16x16x32_bf16"
#endif
#ifndef VLY_P16_BAR_AT
#define VLY_P16_BAR_AT 10           // the barrier sits behind this MFMA of row 6
#endif
#ifndef VLY_P16_PIECE_STRIDE
#define VLY_P16_PIECE_STRIDE 4      // MFMAs between two LDS-DMA pieces
#endif
#ifndef VLY_P16_HEAD
#define VLY_P16_HEAD 8              // the parked stores of a tile leave under this many K tiles of the next one
#endif
#ifndef VLY_P16_NPARK
#define VLY_P16_NPARK 16            // 16-byte pieces of a finished tile that wait in registers (of 32; SwiGLU: of 16)
#endif
#ifndef VLY_P16_TIMING
#define VLY_P16_TIMING 0
#endif
#ifndef VLY_P16_CHAIN
#define VLY_P16_CHAIN 1             // 0: the two K steps of a block are NOT adjacent (j inner, K step outer inside a row) — A/B of the order
#endif

// a[4 blk .. 4 blk + 3] (+)= Wfrag . Afrag: lane (g = l >> 4, r = l & 15) supplies W[n = r][k = 8 g ..], A[m = r][same k] and holds
// D[n = 4 g + e][m = r], e = 0 .. 3.  Back-to-back MFMAs on the same block need no wait states (the result is forwarded as C).
VLY_DEVICE void mfma16c(int blk, const bf16x8& w, const bf16x8& a) {
    asm volatile(VLY_MFMA16C_NAME " a[%2:%3], %0, %1, a[%2:%3]" ::"v"(w), "v"(a), "i"(4 * blk), "i"(4 * blk + 3));
}
VLY_DEVICE void mfma16c_zero(int blk, const bf16x8& w, const bf16x8& a) {
    asm volatile(VLY_MFMA16C_NAME " a[%2:%3], %0, %1, 0" ::"v"(w), "v"(a), "i"(4 * blk), "i"(4 * blk + 3));
}
// (s_nop 1 opens the statement: hipcc reuses the registers of a just-issued ds_write / store as these outputs and pads nothing inside
//  or in front of an asm statement — gemm_p32.hip's lanes 60-63)
VLY_DEVICE f32x4 acc_read4(int blk) {
    f32x4 v;
    asm volatile("s_nop 1\n\tv_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\tv_accvgpr_read_b32 %3, a[%7]"
                 : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3])
                 : "i"(4 * blk), "i"(4 * blk + 1), "i"(4 * blk + 2), "i"(4 * blk + 3));
    return v;
}

template <int EPI>
__global__ void __launch_bounds__(256)
gemm_p16_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ W, const float* __restrict__ bias, void* __restrict__ Cv,
                int M, int N, int K, int lda, int ldw, int ldc, TileMap mp, unsigned long long* __restrict__ tstamps) {
    constexpr int BM = 256, BN = 256, NT = 256;
    constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;           // one K tile: 64 KB
    constexpr int SCR = 8192;                                             // per wave: [0, 4 KB) turn-around of a block row, [4 KB, 6 KB) two bias strips
    constexpr int PA = BM * 8 / NT, NS = 2 * PA;                          // 16 LDS-DMA pieces of 1 KB per wave and K tile
    constexpr bool SWI = EPI == VLY_EPI_SWIGLU;
    constexpr int NPIECE = SWI ? 16 : 32;                                 // 16-byte pieces per lane and tile
    // ... of which the LAST NPARK are parked in registers and trickle out under the next tile; the first NPIECE - NPARK (plain epilogues:
    // block rows 0-3) are stored while the rest of the tile is still being drained.  (All 32 parked = 128 registers beside 80 of fragments:
    // hipcc then spills into the accumulation registers it does not know are live — tools/agpr_audit.py.)
    constexpr int NPARK = VLY_P16_NPARK < NPIECE ? VLY_P16_NPARK : NPIECE, NNOW = NPIECE - NPARK;
    constexpr int HEAD = VLY_P16_HEAD, SPK = NPARK / HEAD;
    static_assert(NPARK % HEAD == 0 && SPK >= 1 && SPK <= 4, "head");
    constexpr int BAR_AT = VLY_P16_BAR_AT, PGS = VLY_P16_PIECE_STRIDE;
    // G: MFMA index counted from row 6 (G = 0 .. 127: rows 6, 7, then 0 .. 5 of the next K tile).  Piece q at G = BAR_AT + 1 + q PGS.
    constexpr int PG0 = BAR_AT + 1, PG_LAST = PG0 + (NS - 1) * PGS;
    constexpr int NFIRST = (32 - PG0 + PGS - 1) / PGS;                    // pieces issued in rows 6 / 7 (G < 32)
    static_assert(PG_LAST + 2 + 4 * SPK < 128 - 16 && NFIRST >= 1 && NFIRST < NS, "piece schedule");
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE + 4 * SCR];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r16 = lane & 15, g = lane >> 4;
    const int ntiles = mp.tiles_m * mp.tiles_n, G = (int)gridDim.x;
    const int nk = K / BK;
    const uint32_t wk = ldw < 0 ? (uint32_t)((N + 63) >> 6) * 4096u : (uint32_t)BK;
    const int No = SWI ? N >> 1 : N;

    // Operands through descriptors that END with the last row: a piece whose row lies past M (N) is out of range and reads as zero — no
    // per-row clamp, a lane's source offset is linear in the piece number.  (Rows past N of a block-packed W land in later blocks or
    // past the end: finite garbage or zero in columns that are never stored.)
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(A), 0, (uint32_t)M * (uint32_t)lda * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(W), 0, ldw < 0 ? (uint32_t)((N + 63) >> 6) * 4096u * (uint32_t)nk * 2u : (uint32_t)N * (uint32_t)ldw * 2u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bias), 0, bias ? (uint32_t)N * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc(Cv, 0, (uint32_t)M * (uint32_t)ldc * 2u, 0x00020000);

    // ---- load cursor: (lt, lk) = tile / K tile of the group of pieces being issued.  A lane's source offset is LINEAR in the piece number
    // (no clamps: the descriptor's range check on the register offset drops rows past the end), so it costs one register per operand;
    // the gap behind a 16-cycle MFMA holds ~3 other instructions: a piece is { M0, the scalar part of its offset } in one gap and
    // { one v_add, buffer_load } in the next
    const int prow = tid >> 3, pswz = ((tid & 7) ^ ((tid >> 3) & 7)) << 3;      // this lane's slot in piece 0: row, swizzled chunk (elements)
    const uint32_t sA32 = (uint32_t)lda * 64u, sW32 = ldw < 0 ? 4096u : (uint32_t)ldw * 64u;       // bytes between the rows of two pieces
    const uint32_t kW = wk * 2u;
    auto offA = [&](int m0) { return (__umul24((uint32_t)(m0 + prow), (uint32_t)lda) + (uint32_t)pswz) * 2u; };
    auto offW = [&](int n0) { return (w_row_off32(n0 + prow, ldw) + (uint32_t)pswz) * 2u; };
    int lt = (int)blockIdx.x, lk = 0, lpar = 0;                          // (lpar: parity of the cursor's tile = its bias strip)
    int lm0, ln0;
    tile_origin<BM, BN>(mp, ntiles, lt, lm0, ln0);
    uint32_t vA = offA(lm0), vW = offW(ln0);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const uint32_t ldsw = lds0 + (uint32_t)wave * 1024u;
    uint32_t soff = 0;                                                   // scalar part of the NEXT piece's offset (set with its M0)
    auto piece_m0 = [&](int buf, int q) {
        const uint32_t dst = ldsw + (uint32_t)buf * STAGE + (q < PA ? (uint32_t)q * 4096u : (uint32_t)A_BYTES + (uint32_t)(q - PA) * 4096u);
        soff = q < PA ? (uint32_t)q * sA32 + (uint32_t)lk * (BK * 2u) : (uint32_t)(q - PA) * sW32 + (uint32_t)lk * kW;
        asm volatile("s_mov_b32 m0, %0" ::"s"(dst) : "memory");
    };
    auto piece_ld = [&](int q) {
        const uint32_t off = (q < PA ? vA : vW) + soff;
        if (q < PA) asm volatile("buffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(off), "s"(rsA) : "memory");
        else asm volatile("buffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(off), "s"(rsW) : "memory");
    };
    auto piece = [&](int buf, int q) {                                   // both at once (prologue, dead waves)
        piece_m0(buf, q);
        asm volatile("s_nop 0" ::: "memory");
        piece_ld(q);
    };
    const uint32_t scr0 = lds0 + 2u * STAGE + (uint32_t)wave * SCR;
    auto bias_piece = [&]() {                                            // 256 floats of the cursor's tile into this wave's strip `lpar`
        const uint32_t vb = (uint32_t)(ln0 + 4 * lane) * 4u;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(scr0 + 4096u + (uint32_t)lpar * 1024u), "v"(vb), "s"(rsB)
                     : "memory");
    };

    // ---- compute cursor
    int ct = lt, cm0 = lm0, cn0 = ln0, cpar = 0;
    int nm0 = 0, nn0 = 0;
    const int wm0 = (wave >> 1) * 128, wn0 = (wave & 1) * 128;
    const int rdA = (wave >> 1) * 16384, rdW = A_BYTES + (wave & 1) * 16384;
    int fo[2];                                                           // fragment read offset of K step s inside a 16-row block
#pragma unroll
    for (int s = 0; s < 2; ++s) fo[s] = r16 * 128 + (((4 * s + g) ^ (r16 & 7)) << 4);
    bf16x8 wf[8][2], af[2][2];                                           // W: the whole K tile; A: two rows of blocks
    auto rd_a = [&](int set, const char* st, int i, int s) { af[set][s] = *(const bf16x8*)(st + rdA + i * 2048 + fo[s]); };
    auto rd_w = [&](const char* st, int j, int s) { wf[j][s] = *(const bf16x8*)(st + rdW + j * 2048 + fo[s]); };

    // ---- the parked tile
    u32x4 park[NPARK];
#pragma unroll
    for (int p = 0; p < NPARK; ++p) park[p] = u32x4{0u, 0u, 0u, 0u};
    uint32_t vo_park = 0x80000000u;                                      // (nothing parked yet: out of range, dropped by the hardware)
    const uint32_t pstep = (uint32_t)ldc * (SWI ? 16u : 8u);             // bytes between the rows of two consecutive parked pieces
    auto park_store = [&](int p) {                                       // parked piece p = piece NNOW + p of the tile
        const uint32_t off = vo_park + (uint32_t)(NNOW + p) * pstep;
        asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(park[p]), "v"(off), "s"(rsC) : "memory");     // (§5.7: data registers are read late)
    };

#if VLY_P16_TIMING
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
    bool crossing = false;
    auto advance = [&]() {                                               // the load cursor moves on one K tile (scalar unit + two VALU at a crossing)
        crossing = false;
        if (lk + 1 < nk) { ++lk; return; }
        if (lt + G >= ntiles) return;                                    // past the last tile: stay on its last K tile
        lt += G;
        lk = 0;
        lpar ^= 1;
        tile_origin<BM, BN>(mp, ntiles, lt, lm0, ln0);
        nm0 = lm0;
        nn0 = ln0;
        crossing = true;
        vA = offA(lm0);
        vW = offW(ln0);
    };

    // ---- prologue: the first tile's bias, group 0 whole, the first NFIRST pieces of group 1
    if (bias) bias_piece();
#pragma unroll
    for (int q = 0; q < NS; ++q) piece(0, q);
    advance();
#pragma unroll
    for (int q = 0; q < NFIRST; ++q) piece(1, q);
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NFIRST) : "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        rd_w(smem, j, 0);
        rd_w(smem, j, 1);
    }
    rd_a(0, smem, 0, 0);
    rd_a(0, smem, 0, 1);

    // ---- one K tile.  FIRST: a block's first MFMA takes C = 0.  ST0 >= 0: park[ST0 .. ST0 + SPK - 1] leave.
    auto ktile = [&](auto first_c, auto st0_c, bool last) {
        constexpr bool FIRST = decltype(first_c)::value;
        constexpr int ST0 = decltype(st0_c)::value;
        const char* cur = smem + buf * STAGE;
        const char* nxt = smem + (buf ^ 1) * STAGE;
        // what hangs behind MFMA G of the cycle that starts with row 6 (see PG0): pieces, the bias piece behind the last one, the parked stores
        auto vmem_at = [&](int Gi, int dstbuf) {
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                if (PG0 + q * PGS - 1 == Gi) piece_m0(dstbuf, q);
                if (PG0 + q * PGS == Gi) piece_ld(q);
            }
            if (Gi == PG_LAST + 1) {
                if (bias && lk == 0 && lt != ct) bias_piece();           // the group is out: its tile's bias rides behind it
            }
            if constexpr (ST0 >= 0) {
#pragma unroll
                for (int s = 0; s < SPK; ++s)
                    if (Gi == PG_LAST + 3 + 4 * s) park_store(ST0 + s);
            }
        };
        if (wave_live) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int set = i & 1;
#pragma unroll
                for (int t = 0; t < 16; ++t) {
                    const int j = VLY_P16_CHAIN ? t >> 1 : t & 7, s = VLY_P16_CHAIN ? t & 1 : t >> 3;
                    __builtin_amdgcn_sched_barrier(0);
                    if (FIRST && s == 0) mfma16c_zero(i * 8 + j, wf[j][0], af[set][0]);
                    else mfma16c(i * 8 + j, wf[j][s], af[set][s]);
                    __builtin_amdgcn_sched_barrier(0);
                    // ---- hooks behind MFMA (i, t)
                    if (t < 2) {                                         // the next row's A fragments (row 7: row 0 of the next K tile)
                        if (i < 7) rd_a(set ^ 1, cur, i + 1, t);
                        else rd_a(set ^ 1, nxt, 0, t);
                    }
                    if (i == 7) {                                        // the next K tile's W fragments, into the registers the finished chains left
                        if (VLY_P16_CHAIN) {
                            rd_w(nxt, t >> 1, t & 1);                    // (MFMA t was the last reader of this fragment)
                        } else if (t >= 8) {
                            rd_w(nxt, t - 8, 0);                         // (K step 0 of column t - 8 is long done; step 1 just issued: read behind the row)
                        }
                    }
                    if (i == 6 && t == BAR_AT) {
                        if constexpr (ST0 >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(SPK) : "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_waitcnt(0xc07f);
                        __builtin_amdgcn_s_barrier();
                    }
                    if (i >= 6) vmem_at((i - 6) * 16 + t, buf);          // rows 6, 7: the new group into THIS buffer
                    else vmem_at(32 + i * 16 + t, buf ^ 1);              // rows 0 .. 5: the group that started in the previous K tile
                    if (i == 5 && t == 0) advance();                     // (every piece of the group is out by row 4)
                }
                if (!VLY_P16_CHAIN && i == 7) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) rd_w(nxt, j, 1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        } else {
            // a wave whose slab of the tile lies outside the problem: pieces, stores and the barrier only
#pragma unroll
            for (int q = NFIRST; q < NS; ++q) piece(buf ^ 1, q);
            if (bias && lk == 0 && lt != ct) bias_piece();
            if constexpr (ST0 >= 0) {
#pragma unroll
                for (int s = 0; s < SPK; ++s) park_store(ST0 + s);
            }
            advance();
            if constexpr (ST0 >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(SPK) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int q = 0; q < NFIRST; ++q) piece(buf, q);
            if (last) {                                                   // this wave may be live in the next tile: its first fragments
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    rd_w(nxt, j, 0);
                    rd_w(nxt, j, 1);
                }
                rd_a(0, nxt, 0, 0);
                rd_a(0, nxt, 0, 1);
            }
        }
        buf ^= 1;
    };

    // ---- the finished tile: accumulators -> + bias -> activation -> 16 bit -> this wave's LDS area -> whole rows -> park[]
    char* const scr = smem + 2 * STAGE + wave * SCR;
    const int wb_plain = r16 * 256 + (r16 << 4) + (g & 1) * 8;           // write base: row r16, chunk position ^ r16; + (column chunk << 4) by XOR
    const int wb_swi = r16 * 128 + ((r16 & 7) << 4) + 4 * g;
    const int rb_plain = (lane >> 4) * 256 + (((lane & 15) ^ (lane >> 4)) << 4);
    const int rb_swi = (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4);
    auto drain = [&](int em0, int en0) {
        if (wave_live) {
            // (the lane bases go through an asm statement per tile: left alone hipcc computes the write and read addresses of a lane ONCE,
            // outside the persistent loop, and holds two dozen registers for them for the whole kernel)
            int wb_p = wb_plain, wb_s = wb_swi, rb_p = rb_plain, rb_s = rb_swi, bo = 4096 + cpar * 1024 + (wn0 + 4 * g) * 4;
            asm volatile("" : "+v"(wb_p), "+v"(wb_s), "+v"(rb_p), "+v"(rb_s), "+v"(bo));
            {
                const int row = em0 + wm0 + (SWI ? lane >> 3 : lane >> 4);
                const int col = SWI ? ((en0 + wn0) >> 1) + (lane & 7) * 8 : en0 + wn0 + (lane & 15) * 8;
                vo_park = col + 8 <= No ? ((uint32_t)row * (uint32_t)ldc + (uint32_t)col) * 2u : 0x80000000u;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_barrier(0);
                f32x4 bv[8];                                             // the bias of the lane's 4 columns per block (zeros without one)
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[j] = bias ? *(const f32x4*)(scr + bo + j * 64) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (j == 4) {
                        __builtin_amdgcn_sched_barrier(0);              // (two halves: sixteen registers of bias at a time)
#pragma unroll
                        for (int jj = 4; jj < 8; ++jj) bv[jj] = bias ? *(const f32x4*)(scr + bo + jj * 64) : f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                    f32x4 v = acc_read4(i * 8 + j);
                    v += bv[j];
                    if constexpr (SWI) {
                        // gate = even columns, up = odd: the lane's four columns give outputs (16 j + 4 g) / 2 + {0, 1}
                        const f32x2 gt = x_sigmoid2(f32x2{v[0], v[2]}, 1.f) * f32x2{v[1], v[3]};
                        *(uint32_t*)(scr + (wb_s ^ (j << 4))) = pack_h2(gt[0], gt[1]);
                    } else {
                        f32x2 x0 = {v[0], v[1]}, x1 = {v[2], v[3]};
                        if constexpr (EPI == VLY_EPI_QUICK_GELU) {
                            x0 = x_sigmoid2(x0, 1.702f);
                            x1 = x_sigmoid2(x1, 1.702f);
                        } else if constexpr (EPI == VLY_EPI_RELU) {
                            x0 = f32x2{fmaxf(x0[0], 0.f), fmaxf(x0[1], 0.f)};
                            x1 = f32x2{fmaxf(x1[0], 0.f), fmaxf(x1[1], 0.f)};
                        }
                        *(u32x2*)(scr + (wb_p ^ ((2 * j + (g >> 1)) << 4))) = u32x2{pack_h2(x0[0], x0[1]), pack_h2(x1[0], x1[1])};
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                auto emit = [&](int pp, const u32x4& v) {                 // piece pp of the tile: stored at once, or parked
                    if (pp < NNOW) {
                        const uint32_t off = vo_park + (uint32_t)pp * pstep;
                        asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen\n\ts_nop 1" ::"v"(v), "v"(off), "s"(rsC) : "memory");
                    } else park[pp >= NNOW ? pp - NNOW : 0] = v;
                };
                if constexpr (SWI) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) emit(i * 2 + q, *(const u32x4*)(scr + rb_s + q * 1024));
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) emit(i * 4 + q, *(const u32x4*)(scr + (rb_p ^ (q << 6)) + q * 1024));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        } else vo_park = 0x80000000u;
    };

    // ---- the schedule: per tile { first K tile; HEAD - 1 more with parked stores; the rest rolled; drain }
    for (;;) {
        wave_live = __builtin_amdgcn_readfirstlane((cm0 + wm0 < M && cn0 + wn0 < N) ? 1 : 0) != 0;
        VLY_STAMP();
        ktile(std::true_type{}, std::integral_constant<int, 0>{}, false);
        static_for<HEAD - 1>([&](auto i) { ktile(std::false_type{}, std::integral_constant<int, (decltype(i)::value + 1) * SPK>{}, false); });
        VLY_STAMP();
        for (int kt = HEAD; kt < nk - 1; ++kt) ktile(std::false_type{}, std::integral_constant<int, -1>{}, false);
        ktile(std::false_type{}, std::integral_constant<int, -1>{}, true);
        VLY_STAMP();
        drain(cm0, cn0);
        VLY_STAMP();
        if (ct + G >= ntiles) break;
        ct += G;
        cm0 = nm0;
        cn0 = nn0;
        cpar ^= 1;
    }
    // ---- the last tile's stores, and no LDS-DMA may outlive the workgroup's LDS allocation
#pragma unroll
    for (int p = 0; p < NPARK; ++p) park_store(p);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if VLY_P16_TIMING
    if (threadIdx.x == 0 && blockIdx.x < 64 && tstamps) tstamps[(size_t)blockIdx.x * 65] = (unsigned long long)tsn;
#endif
}
#undef VLY_STAMP

}  // namespace

#if VLY_P16_TIMING
static void* vlydbg_p16_buffer() {
    static void* buf = [] { void* p = nullptr; (void)hipMalloc(&p, 64 * 65 * 8); (void)hipMemset(p, 0, 64 * 65 * 8); return p; }();
    return buf;
}
extern "C" int vlydbg_p32_timing_read(unsigned long long* host) { return (int)hipMemcpy(host, vlydbg_p16_buffer(), 64 * 65 * 8, hipMemcpyDeviceToHost); }
#endif

// Tile hint 497 of vly_gemm_bf16.  Returns 1 when the problem does not fit this kernel (the caller falls back to hint 197):
// 16-bit outputs, no residual, 16-byte aligned rows of whole 8-column chunks, K >= (HEAD + 1) K tiles, < 2^16 tiles.
__attribute__((visibility("hidden"))) int valley_p16_gemm(const void* A, const void* W, const float* bias, const float* R, void* C, int M, int N,
                                                          int K, int lda, int ldw, int ldc, int epi, int out, hipStream_t st) {
    constexpr int BM = 256, BN = 256;
    const int No = epi == VLY_EPI_SWIGLU ? N >> 1 : N;
    if (out != VLY_OUT_BF16 || R || ldc % 8 || ((uintptr_t)C & 15) || No % 8 || (size_t)M * (size_t)ldc * 2 >= ((size_t)1 << 31) ||
        K % BK || K / BK < VLY_P16_HEAD + 1 || (epi != VLY_EPI_NONE && epi != VLY_EPI_QUICK_GELU && epi != VLY_EPI_SWIGLU && epi != VLY_EPI_RELU))
        return 1;
    const int tm = (M + BM - 1) / BM, tn = (N + BN - 1) / BN;
    if ((long)tm * tn >= 65536 || lda >= (1 << 24) || (ldw > 0 && ldw >= (1 << 24)) || M >= (1 << 24) || N >= (1 << 24)) return 1;
    const TileMap mp = make_tile_map(M, N, K, BM, BN);
    const int cus = persistent_grid_cus(), tiles = tm * tn;
    dim3 grid(tiles >= cus ? cus : tiles), block(256);
    unsigned long long* ts = nullptr;
#if VLY_P16_TIMING
    ts = (unsigned long long*)vlydbg_p16_buffer();
#endif
#define VLY_P16_LAUNCH(E)                                                                                                          \
    hipLaunchKernelGGL((gemm_p16_kernel<E>), grid, block, 0, st, (const uint16_t*)A, (const uint16_t*)W, bias, C, M, N, K, lda, ldw, ldc, \
                       mp, ts)
    if (epi == VLY_EPI_NONE) VLY_P16_LAUNCH(VLY_EPI_NONE);
    else if (epi == VLY_EPI_QUICK_GELU) VLY_P16_LAUNCH(VLY_EPI_QUICK_GELU);
    else if (epi == VLY_EPI_SWIGLU) VLY_P16_LAUNCH(VLY_EPI_SWIGLU);
    else VLY_P16_LAUNCH(VLY_EPI_RELU);
#undef VLY_P16_LAUNCH
    return vly_check_launch("vly_gemm_bf16");
}
