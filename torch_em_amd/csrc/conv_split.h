// conv_split.h -- operand splitting shared by the split-precision MFMA convolution kernels
// (conv_bf16x3.hip: one patch per workgroup; conv_pp.hip: ping-pong teams).
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/tem_hip.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define BCK 16   // input channels per staged chunk = K of one MFMA
#define BLS 20   // LDS floats per halo voxel: 16 hi bf16 (32 B) + 16 lo bf16 (32 B) + 16 B pad

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    bf16x2 v = {(__bf16)a, (__bf16)b};  // v_cvt_pk_bf16_f32, round-to-nearest-even
    return __builtin_bit_cast(unsigned, v);
}
// (hi, lo) split of two floats: hi = bf16(x), lo = bf16(x - hi)
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
    hi = pk_bf16(a, b);
    const float ra = a - __builtin_bit_cast(float, hi << 16);
    const float rb = b - __builtin_bit_cast(float, hi & 0xffff0000u);
    lo = pk_bf16(ra, rb);
}
// fp16 counterparts (forward of NORMALISED activations only: |x| << 65504).  x = h + l with two fp16 terms carries 22
// mantissa bits, so hi*hi + hi*lo + lo*hi ("fp16x3") is fp32-class (~2^-22 per product) at HALF the MFMAs of bf16x6.
// The lo term of an O(1) operand is O(2^-12) and that of a 0.05-sized weight is 2^-16: deep in fp16's subnormal range
// (spacing 6e-8), where it would keep only a few bits.  Both lo planes are therefore stored SCALED by 2^12 (exact), the
// cross products hi*lo' + lo'*hi accumulate in their own fp32 accumulators and the epilogue adds them back times 2^-12.
#define F16_LO_SCALE 4096.f
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
template <bool F16>
__device__ __forceinline__ unsigned pk16(float a, float b) {
    if (F16) {
        half2_t v = {(_Float16)a, (_Float16)b};  // round-to-nearest-even
        return __builtin_bit_cast(unsigned, v);
    }
    return pk_bf16(a, b);
}
template <bool F16>
__device__ __forceinline__ float lo16(unsigned h) {
    if (F16) return (float)__builtin_bit_cast(half2_t, h).x;
    return __builtin_bit_cast(float, h << 16);
}
template <bool F16>
__device__ __forceinline__ float hi16(unsigned h) {
    if (F16) return (float)__builtin_bit_cast(half2_t, h).y;
    return __builtin_bit_cast(float, h & 0xffff0000u);
}
template <bool F16>
__device__ __forceinline__ floatx16 mfma16(uint4 a, uint4 b, floatx16 c) {
    if (F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ float act_apply_b(float v, int act) {
    if (act == TEM_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == TEM_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    return v;
}

// "fp16x3, prescaled" (TEM_WL_F16X3S): instead of scaling only the lo planes (which forces the cross products into a
// second accumulator), BOTH terms of an operand carry one power-of-two prescale: activations x 2^5, weights x 2^7.  The lo
// term of an O(1) activation is then 2^-6 and that of a 0.05-sized weight 2^-8 -- normal fp16 numbers -- and where it
// does drop into the subnormal range (|x^| < 2^-7, |w| < 2^-9) its absolute error is <= 2^-25 in scaled units, i.e.
// <= 1e-9 of an O(1) activation / 2.4e-10 of a weight: below the fp32 rounding of the typical terms of the same dot
// product.  All three products hi*hi + hi*lo + lo*hi share ONE accumulator; the epilogue multiplies by 2^-12.
// MEASURED AND REJECTED as the default: the matrix core aligns the small cross products to the large accumulator and
// truncates, a one-sided error of ~1e-5 per output; harmless per element, but the GroupNorm backward sums it over all
// voxels (tests/test_gpu_unet.py: 4e-3 on the first norm's bias gradient vs 5e-5 for the two-accumulator layout).
// Range: |x^| <= 2000 (an InstanceNorm output is bounded by sqrt(voxels); clamped beyond), |w| <= 500 (clamped).
#define F16_A_PRESCALE 32.f
#define F16_W_PRESCALE 128.f
#define F16_PRESCALE_INV (1.f / 4096.f)
