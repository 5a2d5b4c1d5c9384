// The dense product W S of the fused H = 256 right-hand sides on the fp16 matrix cores with fp32-grade results (rounds 3-5).
//
// Every fp32 operand is split into TWO fp16 pieces, x = x0 + x1 + r with 11 + 11 significand bits (|r| <= 2^-21 |x| for the
// round-toward-zero pieces of S, 2^-23 |x| for the round-to-nearest pieces of W), and the three partial products  s1 w0,  s0 w1,
// s0 w0  are accumulated in fp32 by v_mfma_f32_32x32x16_f16, small terms first.  fp16 has 5 exponent bits, so every S ROW is
// multiplied by a power of two (exact; formed by the wave that folds the row, which holds it in registers) and every OUTPUT ROW of
// the weights (row n of the B operand: W[n][:], resp. W^T[n][:] for gS = gZ W) by a power of two of its own; the accumulator is
// multiplied back (exact) when it is dumped, by 1 / (weight-row scale) and then by 1 / (S-row scale).
//
// Round 5 - WHERE the scale puts the row.  Rounds 3-4 brought the largest magnitude of a row into [0.5, 1): the fp16 range
// BELOW that is 2^-24 (the smallest subnormal), so an element 2^-e below its row's maximum kept min(22, 24 - e) bits and ONE
// outlier (a weight 2^12 above the rest, a dominant S channel against small weights) cost every other element of its scale
// group up to 10 bits - measured by the round-4 review (emulation: 1.2e-5 of sum |s w| at x 2^8 for one weight of a globally
// scaled W, 1.8e-4 at x 2^12) and reproduced by tools/micro/split_emul.py.  The row maximum now goes into [2^14, 2^15) - fp16's
// largest finite value is 65504; round-toward-zero and values below 2^15 cannot overflow, the fp32 accumulator sees at most
// 3 x 256 x 2^30 - which moves the floor to 2^-39 of the row's maximum:
//
//   GUARANTEE (what the tests hold the kernels to, tests/test_gpu_split_range.py):  an element 2^-e below the largest magnitude of
//   its row (S) / its output row (W) is represented with min(21, 38 - e) bits; for operand rows whose elements that matter lie
//   within 2^17 of the row maximum every output obeys  |out - exact| <= 2e-6 sum_k |s_k w_ok|  (the fp32 fma chain's own bound at
//   K = 256 is 256 x 2^-24 = 1.5e-5 of the same sum, its typical error 2e-7); beyond that the error degrades as
//   2^-38 (max_k |s_k| sum_k |w_ok| + max_k |w_ok| sum_k |s_k|), i.e. it stays below the fp32 chain's bound until the spread
//   inside ONE row exceeds 2^21 AND the small elements carry the sum.
//   tools/micro/split_emul.py (exact accumulation, max error / sum |s w|; fp32 chain | round 3-4 | now):
//     one weight x 2^12: 1.4e-6 | 2.0e-5 | 2.1e-7     one output row x 2^12: 2.2e-7 | 4.8e-5 | 5.1e-8
//     S channel x 2^12 against a weight column x 2^-12: 2.2e-7 | 7.9e-5 | 5.1e-8      log-normal weights: 9.4e-7 | 1.1e-5 | 3.7e-7
//
// Measured against fp64 on 10^6 x 256 x 256 (tools/micro/gemm_split_lab.hip, profiles/r03_gemm_split_lab.txt), max error over
// sum |s w|:  fp32 MFMA chain 2.3e-7 | three bf16 pieces, 6 products (rounds 1-2) 2.0e-7 | two fp16 pieces, 3 products 1.9e-7.
// Half the matrix-pipe time of the bf16 form, two weight planes instead of three (the L2 -> CU weight stream and the registers
// that hold resident k-steps shrink by a third), 24 instead of 38 VALU per split.
#pragma once
#include "common.h"

namespace ndcn {

typedef unsigned u32x4_s16 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2_s16 __attribute__((ext_vector_type(2)));
typedef float f32x4_s16 __attribute__((ext_vector_type(4)));

constexpr int kS16Planes = 2;
constexpr int kS16Bytes = 8 * 16 * kS16Planes * 1024;         // packed weights: [n-tile 8][k-step 16][plane 2][lane 64][8 fp16]
// behind the planes: float unscale[256] = 1 / (scale of B-operand row n), i.e. per OUTPUT column of the product
constexpr int kS16TailBytes = 256 * 4;                       // (+ 8 words behind them in the forward image: the range guard's verdict per n-tile)
constexpr int kS16GuardBits = 19;                              // a non-zero element more than 2^19 below its row's maximum keeps < 19 bits: outside the 2e-6 guarantee
constexpr int kS16GuardCount = 4;                              // ... and a weight row with this many of them leaves for the fp32 matrix cores (rhs_fused.hip)
constexpr int kS16GuardBytes = 32;                             // the guard's verdict per n-tile, behind the unscale factors of a packed image
constexpr int kS16Top = 15;                                    // the scale brings a row's largest magnitude into [2^(top-1), 2^top)

// wave-uniform maximum of an unsigned value (|x| bit patterns order like magnitudes; a NaN pattern wins)
__device__ __forceinline__ unsigned s16_wave_umax(unsigned v) {
    unsigned t;
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false); v = v > t ? v : t;      // quad_perm [1,0,3,2]
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, false); v = v > t ? v : t;      // quad_perm [2,3,0,1]
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false); v = v > t ? v : t;     // row_ror:4
    t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false); v = v > t ? v : t;     // row_ror:8
    const unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const unsigned c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

// |x| bits of the largest magnitude -> the bits of the power of two `scale` with scale * max in [2^14, 2^15), and of 1 / scale.
// The exponent of the scale is clamped to [-126, 126] (rows below 2^-112 land lower in the fp16 range, rows near the top of
// fp32 cannot exist as a sum of finite terms); rows holding Inf / NaN are left alone (their pieces come out Inf / NaN).
__device__ __forceinline__ void s16_scale_bits(unsigned max_bits, unsigned &scale_bits, unsigned &unscale_bits) {
    const unsigned eb = max_bits >> 23;                          // biased exponent (sign bit is clear)
    const unsigned want = 253u + (unsigned)kS16Top - eb;         // biased exponent of the scale
    const unsigned sb = eb == 255u ? 127u : (want > 253u ? 253u : (want < 1u ? 1u : want));
    scale_bits = sb << 23;
    unscale_bits = (254u - sb) << 23;
}

// (the WHOLE vector is bit-cast, then its elements are read: hipcc 7.2 compiles __builtin_bit_cast(unsigned, v.y) of an
// ext-vector ELEMENT as a read of element 0 - rounds 3-4 therefore took the row maximum over every fourth column only, which the
// [0.5, 1) scale target forgave (an underestimated maximum still fits fp16 there) and tests/test_gpu_split_range.py does not)
__device__ __forceinline__ unsigned s16_row_max_bits(f32x4_s16 v) {
    const u32x4_s16 u = __builtin_bit_cast(u32x4_s16, v) & 0x7fffffffu;
    const unsigned ab = u.x > u.y ? u.x : u.y, cd = u.z > u.w ? u.z : u.w;
    return ab > cd ? ab : cd;
}

// 8 consecutive fp32 of one S row, times the row's scale -> two fp16x8 pieces (high: round toward zero, low: the exact
// remainder rounded toward zero): x * sc = A0 + A1 + r
__device__ __forceinline__ void s16_split8(f32x4_s16 r0, f32x4_s16 r1, float sc, u32x4_s16 &A0, u32x4_s16 &A1) {
    const f32x2_s16 x[4] = {{r0.x, r0.y}, {r0.z, r0.w}, {r1.x, r1.y}, {r1.z, r1.w}};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x2_s16 xs = x[q] * sc;
        const f16x2 h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(xs.x, xs.y));
        const f32x2_s16 hf = {(float)h.x, (float)h.y};
        const f32x2_s16 ra = xs - hf;
        A0[q] = __builtin_bit_cast(unsigned, h);
        A1[q] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ra.x, ra.y));
    }
}

// 4 consecutive fp32 of one S row -> the two fp16x4 pieces (element for element the arithmetic of s16_split8)
typedef unsigned u32x2_s16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void s16_split4(f32x4_s16 r, float sc, u32x2_s16 &A0, u32x2_s16 &A1) {
    const f32x2_s16 x[2] = {{r.x, r.y}, {r.z, r.w}};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const f32x2_s16 xs = x[q] * sc;
        const f16x2 h = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(xs.x, xs.y));
        const f32x2_s16 hf = {(float)h.x, (float)h.y};
        const f32x2_s16 ra = xs - hf;
        A0[q] = __builtin_bit_cast(unsigned, h);
        A1[q] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(ra.x, ra.y));
    }
}

__device__ __forceinline__ void s16_mfma(float __attribute__((ext_vector_type(16))) &c, u32x4_s16 a, u32x4_s16 b) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

}  // namespace ndcn
