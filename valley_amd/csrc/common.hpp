// Shared device helpers for the gfx950 kernels (wave64, MFMA 16x16x32 on the 16-bit storage type: bf16 or fp16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) short bf16x8;   // one MFMA A/B operand (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;    // one 16x16 accumulator fragment
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define VLY_DEVICE __device__ __forceinline__

// ---- the 16-bit storage type ----------------------------------------------------------------------------------------
// Every "half" tensor of the C ABI (GEMM operands, activations, KV cache) is bf16 in libvalley_hip.so and IEEE fp16 in
// libvalley_hip_f16.so, which is this same source tree compiled with -DVLY_FP16=1 (VALLEY_PRECISION=fp16 on the host):
// fp16 is the reference's own inference dtype (valley/inference/run_valley.py:39, valley_model.py:430), runs on the same
// MFMA at the same rate (v_mfma_f32_16x16x32_f16) and keeps 3 more mantissa bits.  The kernels never touch the bits
// themselves: they go through h2f / f2h / h_lo / h_hi / pack_h2 / mfma16 below; vly_storage_dtype() reports which
// library one holds.  Accumulation, softmax / norm statistics and the residual stream are fp32 in both.
#ifndef VLY_FP16
#define VLY_FP16 0
#endif
typedef __attribute__((ext_vector_type(2))) float vly_f32x2;
#if VLY_FP16
typedef __attribute__((ext_vector_type(2))) _Float16 vly_h2;
typedef __attribute__((ext_vector_type(8))) _Float16 vly_h8;
VLY_DEVICE float h2f(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
VLY_DEVICE float h_lo(uint32_t w) { return (float)__builtin_bit_cast(vly_h2, w)[0]; }      // element 0 / 1 of a packed pair
VLY_DEVICE float h_hi(uint32_t w) { return (float)__builtin_bit_cast(vly_h2, w)[1]; }
VLY_DEVICE uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }      // round-to-nearest-even
#else
typedef __attribute__((ext_vector_type(2))) __bf16 vly_h2;
VLY_DEVICE float h2f(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
VLY_DEVICE float h_lo(uint32_t w) { return __uint_as_float(w << 16); }
VLY_DEVICE float h_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
VLY_DEVICE uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
#endif
// fp32 pair -> packed storage pair, round-to-nearest-even: ONE v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 on gfx950 (a hand-written
// integer RNE costs ~6 VALU ops per value, which showed up in the attention softmax and in every GEMM epilogue).
VLY_DEVICE uint32_t pack_h2(float lo, float hi) {
    const vly_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, vly_h2));
}

// x * sigmoid(a*x) = x / (1 + 2^(-a*log2(e)*x)) with one v_exp_f32 and one v_rcp_f32 (a full IEEE division costs ~10 VALU
// instructions, which showed up as 4-5 us of un-overlapped epilogue on the 8224 x 4096 quick_gelu GEMM); 1 ulp, far below
// the bf16 rounding of the result.  quick_gelu: a = 1.702 (hf activations.py QuickGELU); SiLU: a = 1.  The scale and
// log2(e) are ONE constant (round 3: the two multiplies of -a*x and of __expf were 16 of the 75 instructions per 8
// outputs of the persistent kernel's epilogue).  Every kernel evaluates exactly this expression, scalar or packed.
typedef __attribute__((ext_vector_type(2))) float f32x2;
VLY_DEVICE float x_sigmoid(float x, float a) {
    return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(x * (-1.4426950408889634f * a)));
}
// two values at once: v_pk_mul_f32 / v_pk_add_f32 around the two transcendentals (same bits as the scalar form)
VLY_DEVICE f32x2 x_sigmoid2(f32x2 x, float a) {
    const f32x2 t = x * (-1.4426950408889634f * a);
    f32x2 e = {__builtin_amdgcn_exp2f(t[0]), __builtin_amdgcn_exp2f(t[1])};
    e += 1.f;
    const f32x2 r = {__builtin_amdgcn_rcpf(e[0]), __builtin_amdgcn_rcpf(e[1])};
    return x * r;
}

// Rotate-half RoPE of one element pair, written ONE way for every kernel that rotates (vly_rope_kv, the fused
// prefill and decode attention kernels), so that the cached keys are bit-identical whichever path wrote them:
//   low half  (d <  64): x*cos - partner*sin        high half (d >= 64): x*cos + partner*sin
VLY_DEVICE float rope_rot(float x, float partner, float c, float s, float sign) {
    return fmaf(x, c, sign * (partner * s));
}

// D(16x16) += A(16x32) * B(32x16).  Lane l supplies A[row = l&15][k = 8*(l>>4) .. +7] and
// B[k = 8*(l>>4) .. +7][col = l&15]; it receives D[row = 4*(l>>4) + r][col = l&15], r = 0..3.
VLY_DEVICE f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {            // (bf16x8 = 8 storage elements = 4 VGPRs, either dtype)
#if VLY_FP16
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(vly_h8, a), __builtin_bit_cast(vly_h8, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}

// Expressions whose rounding must not depend on the kernel they are inlined into.  hipcc (-ffp-contract=fast) turns a*b + c into an
// fma wherever its vectoriser has not already paired the multiplies: the SAME source line was two roundings in one kernel (v_pk_mul +
// v_add) and one in another (v_mul + v_fmac), and the persistent decode step disagreed with the launches it replaces in 0.4 % of
// steps (one RMSNorm sum a last bit apart -> one 16-bit activation rounded the other way).  Pinned here: never contracted.
VLY_DEVICE float vly_sumsq4(float x, float y, float z, float w) {
#pragma clang fp contract(off)
    return x * x + y * y + z * z + w * w;
}
VLY_DEVICE float vly_mul_add(float a, float b, float c) {       // a * b + c, two roundings
#pragma clang fp contract(off)
    return a * b + c;
}
// Eight products of two packed 16-byte operands as a chain of eight fmas, low element first — the GEMVs' inner step.  With
// contraction allowed, hipcc may fold two fmas on the halves of one fp16 register into ONE v_dot2_f32_f16 (other intermediate
// rounding) in one kernel and not in another: round 4's persistent decode step then differed from the launches on the fp16
// build only.  Pinned: explicit fmas, never re-associated.
VLY_DEVICE float vly_dot8(const u32x4& w, const u32x4& a) {
#pragma clang fp contract(off)
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        s = __builtin_fmaf(h_lo(w[i]), h_lo(a[i]), s);
        s = __builtin_fmaf(h_hi(w[i]), h_hi(a[i]), s);
    }
    return s;
}

// Butterfly reductions over the 64 lanes: v (+)= v[lane ^ 32], ^ 16, ^ 8, ^ 4, ^ 2, ^ 1 — every lane ends with the total, summed in
// that fixed order.  __shfl_xor compiles to ds_bpermute (an LDS-crossbar round trip per step, ~12 issue slots + 6 lgkmcnt waits per
// sum); the same partners are reachable without the LDS: the two gfx950 swaps for 32 and 16, then DPP inside rows of 16 — a
// rotation by 8 IS xor 8, a rotation by 4 reaches lane i ^ 4 or (i ^ 4) ^ 8, which hold equal values after the xor-8 step, the quad
// permutes are xor 2 and xor 1.  Same operands in every add: the result is BIT-identical to the __shfl_xor form (round 4; the decode
// GEMVs reduce two sums per row pair and wave).
template <typename OP>
VLY_DEVICE float wave_reduce(float v, OP op) {
    {
        const uint32_t x = __float_as_uint(v);
        const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);     // r[0] = {lo, lo}, r[1] = {hi, hi}
        v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    {
        const uint32_t x = __float_as_uint(v);
        const auto r = __builtin_amdgcn_permlane16_swap(x, x, false, false);     // rows {0, 0, 2, 2} and {1, 1, 3, 3}
        v = op(__uint_as_float(r[0]), __uint_as_float(r[1]));
    }
    v = op(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x128, 0xf, 0xf, false)));   // row_ror:8
    v = op(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x124, 0xf, 0xf, false)));   // row_ror:4
    v = op(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x4e, 0xf, 0xf, false)));    // quad_perm [2,3,0,1]
    v = op(v, __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0xb1, 0xf, 0xf, false)));    // quad_perm [1,0,3,2]
    return v;
}
// the __shfl_xor forms (tests/c_abi and A/B builds: -DVLY_SHFL_REDUCE=1 restores them everywhere — same partners, same order of
// additions, so a build with the switch is the bit-identity witness of the DPP forms)
#ifndef VLY_SHFL_REDUCE
#define VLY_SHFL_REDUCE 0
#endif
VLY_DEVICE float wave_sum_shfl(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
VLY_DEVICE float wave_max_shfl(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
VLY_DEVICE float wave_sum(float v) {
#if VLY_SHFL_REDUCE
    return wave_sum_shfl(v);
#else
    return wave_reduce(v, [](float a, float b) { return a + b; });
#endif
}
VLY_DEVICE float wave_max(float v) {
#if VLY_SHFL_REDUCE
    return wave_max_shfl(v);
#else
    return wave_reduce(v, [](float a, float b) { return fmaxf(a, b); });
#endif
}

// async global -> LDS copy of 16 bytes per lane; LDS destination = wave-uniform base + lane*16
VLY_DEVICE void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// the same with a cache-policy immediate (gfx942+: 1 = sc0, 2 = nt, 16 = sc1) — for A/B builds of the staging loads
template <int CPOL>
VLY_DEVICE void glds16_cp(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, CPOL);
}

// LDS-DMA through a buffer descriptor: address = descriptor base + voff (per lane, loop-invariant) + soff (SGPR, the K
// position) — no per-piece 64-bit VALU address arithmetic, one SALU (m0) + one VMEM issue slot per piece.
VLY_DEVICE __amdgpu_buffer_rsrc_t vly_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0xFFFFFFFFu, 0x00020000);
}
template <int CPOL>
VLY_DEVICE void bglds16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, void* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, CPOL);
}

void vly_set_error(const char* fmt, ...);
int vly_check_launch(const char* what);
int vly_tile_order_m_fast(int M, int N, int K, int tiles_m, int tiles_n);
