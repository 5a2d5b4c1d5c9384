// The dense product W S of the fused H = 256 right-hand sides on the fp16 matrix cores with fp32-grade results (round 3).
//
// Every fp32 operand is split error-free into TWO fp16 pieces, x = x0 + x1 + r with 11 + 11 significand bits (|r| <= 2^-22 |x|),
// and the three partial products  s1 w0,  s0 w1,  s0 w0  are accumulated in fp32 by v_mfma_f32_32x32x16_f16, small terms first.
// fp16 has 5 exponent bits, so each S row is multiplied by a power of two (exact) that brings its largest magnitude into
// [0.5, 1) - formed by the wave that folds the row, which holds it in registers - and the weights by one global power of two;
// the accumulator row is multiplied back (exact) when it is dumped.  fp16 subnormals are honoured by the MFMA (measured: a row
// whose elements span 5 decades loses nothing), so an element 2^-14 below its row's maximum keeps an ABSOLUTE accuracy of
// 2^-25 of that maximum - the scale of the fp32 accumulation's own rounding.
// Measured against fp64 on 10^6 x 256 x 256 (tools/micro/gemm_split_lab.hip, profiles/r03_gemm_split_lab.txt), max error over
// sum |s w|:  fp32 MFMA chain 2.3e-7 | three bf16 pieces, 6 products (rounds 1-2) 2.0e-7 | two fp16 pieces, 3 products 1.9e-7
// (rows spanning 16 decades: 3.4e-7 | 2.5e-7 | 1.8e-7).  Half the matrix-pipe time, two weight planes instead of three (the L2 ->
// CU weight stream and the registers that hold resident k-steps shrink by a third), 24 instead of 38 VALU per split.
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
// behind the planes: float {weight scale, 1 / weight scale}

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

// |x| bits of the largest magnitude -> the bits of the power of two `scale` with scale * max in [0.5, 1), and of 1 / scale.
// Huge rows (>= 2^126) are brought to <= 4, rows holding Inf / NaN are left alone (their pieces come out Inf / NaN).
__device__ __forceinline__ void s16_scale_bits(unsigned max_bits, unsigned &scale_bits, unsigned &unscale_bits) {
    const unsigned eb = max_bits >> 23;                          // biased exponent (sign bit is clear)
    const unsigned sb = eb <= 252u ? 253u - eb : (eb == 255u ? 127u : 1u);
    scale_bits = sb << 23;
    unscale_bits = (254u - sb) << 23;
}

__device__ __forceinline__ unsigned s16_row_max_bits(f32x4_s16 v) {
    const unsigned a = __builtin_bit_cast(unsigned, v.x) & 0x7fffffffu, b = __builtin_bit_cast(unsigned, v.y) & 0x7fffffffu;
    const unsigned c = __builtin_bit_cast(unsigned, v.z) & 0x7fffffffu, d = __builtin_bit_cast(unsigned, v.w) & 0x7fffffffu;
    const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
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
