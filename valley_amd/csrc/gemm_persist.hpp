// Shared by the round-6 persistent GEMMs (gemm_p32.hip: 32x32x16 MFMA, gemm_p16.hip: 16x16x32 in chains of two): the clobber list that
// makes a kernel own a0 .. a255, the tile -> origin map on the scalar unit, a compile-time loop.
#pragma once
#include <type_traits>
#include <utility>
#include "common.hpp"

#define VLY_A8(n) "a" #n "0", "a" #n "1", "a" #n "2", "a" #n "3", "a" #n "4", "a" #n "5", "a" #n "6", "a" #n "7", "a" #n "8", "a" #n "9"
#define VLY_ALL_AGPRS                                                                                                                  \
    "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", VLY_A8(1), VLY_A8(2), VLY_A8(3), VLY_A8(4), VLY_A8(5), VLY_A8(6), VLY_A8(7), \
        VLY_A8(8), VLY_A8(9), VLY_A8(10), VLY_A8(11), VLY_A8(12), VLY_A8(13), VLY_A8(14), VLY_A8(15), VLY_A8(16), VLY_A8(17), VLY_A8(18),   \
        VLY_A8(19), VLY_A8(20), VLY_A8(21), VLY_A8(22), VLY_A8(23), VLY_A8(24), "a250", "a251", "a252", "a253", "a254", "a255"

int vly_tile_group_height(int M, int N, int K, int tiles_m, int tiles_n, int BM, int BN, int wg_per_cu);

namespace vlyp {

// x / d for x, d < 2^16 with mg = floor(2^32 / d) + 1 (host; 0 for d = 1): exact, one s_mul_hi_u32 — the tile -> origin map stays on the scalar unit
VLY_DEVICE int udiv_magic(int x, unsigned mg) { return mg ? (int)__builtin_amdgcn_readfirstlane((int)__umulhi((unsigned)x, mg)) : x; }

struct TileMap {                    // tile order (gemm_bf16.hip's: XCD-contiguous runs, groups of gm m-tiles), divisions by multiplication
    int tiles_m, tiles_n, gm, gsz, ghl;             // gsz = gm * tiles_n; ghl = height of the last group
    unsigned mg_gsz, mg_gm, mg_ghl;
};
inline TileMap make_tile_map(int M, int N, int K, int BM, int BN) {
    TileMap mp;
    mp.tiles_m = (M + BM - 1) / BM;
    mp.tiles_n = (N + BN - 1) / BN;
    mp.gm = vly_tile_group_height(M, N, K, mp.tiles_m, mp.tiles_n, BM, BN, 1);
    mp.gsz = mp.gm * mp.tiles_n;
    const int groups = (mp.tiles_m + mp.gm - 1) / mp.gm;
    mp.ghl = mp.tiles_m - (groups - 1) * mp.gm;
    auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) / (unsigned long long)d) + 1ull); };
    mp.mg_gsz = magic(mp.gsz);
    mp.mg_gm = magic(mp.gm);
    mp.mg_ghl = magic(mp.ghl);
    return mp;
}
// tile t of ntiles -> origin (scalar unit only)
template <int BM, int BN>
VLY_DEVICE void tile_origin(const TileMap& mp, int ntiles, int t, int& m0, int& n0) {
    const int xcd = t & 7, qd = ntiles >> 3, rm = ntiles & 7;
    const int swz = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (t >> 3);
    const int grp = udiv_magic(swz, mp.mg_gsz), first = grp * mp.gm;
    const int rr = swz - grp * mp.gsz;
    const bool lastg = first + mp.gm > mp.tiles_m;
    const int gh = lastg ? mp.ghl : mp.gm;
    const int c = udiv_magic(rr, lastg ? mp.mg_ghl : mp.mg_gm);
    m0 = (first + rr - c * gh) * BM;
    n0 = c * BN;
}

template <int... I, typename F>
VLY_DEVICE void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
VLY_DEVICE void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

VLY_DEVICE uint32_t w_row_off32(int n, int ldw) {
    return ldw < 0 ? (uint32_t)(n >> 6) * 4096u + (uint32_t)(n & 63) * 64u : __umul24((uint32_t)n, (uint32_t)ldw);
}

inline int persistent_grid_cus() {
    static const int cus = [] {
        int dev = 0, n = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (getenv("VLY_P4_GRID")) n = atoi(getenv("VLY_P4_GRID"));
        return n > 0 ? n / 8 * 8 : 256;
    }();
    return cus;
}

}  // namespace vlyp
