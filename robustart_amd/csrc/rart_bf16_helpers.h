// bf16 packing / 1-bit ReLU-mask helpers shared by the fused Bottleneck kernels (gfx950).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;   // (arrays of HIP's uint4 struct end up in scratch; ext vectors do not)

namespace rart_bf16 {
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) short i16x2_t;
// two floats -> packed bf16 pair, hardware round-to-nearest-even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  f32x2_t f = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
}
// ReLU on a packed pair with 16-bit integer ops (a bf16 is > 0 exactly when its bits, read as int16, are > 0)
__device__ __forceinline__ uint32_t relu_bf16x2(uint32_t w) {
  const i16x2_t z = {0, 0};
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, w), z));
}
// bits (2*pair, 2*pair+1) of `byte` -> 0xFFFF / 0 halves
__device__ __forceinline__ uint32_t halves_from_bits(uint32_t byte, uint32_t pair) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_sbfe(byte, 2u * pair, 1u);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_sbfe(byte, 2u * pair + 1u, 1u);
  return __builtin_amdgcn_perm(hi, lo, 0x07060100u);
}
// a packed bf16 pair -> 2 bits (value > 0)
__device__ __forceinline__ uint32_t bits_from_halves(uint32_t w) {
  const i16x2_t z = {0, 0}, one = {1, 1};
  const uint32_t t = __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_elementwise_max(__builtin_bit_cast(i16x2_t, w), z), one));
  return (t | (t >> 15)) & 3u;
}
// eight packed bf16 values -> their sign byte (bit j = value j > 0)
__device__ __forceinline__ uint32_t sign_byte(uint4 v) {
  return bits_from_halves(v.x) | (bits_from_halves(v.y) << 2) | (bits_from_halves(v.z) << 4) | (bits_from_halves(v.w) << 6);
}
}  // namespace rart_bf16
